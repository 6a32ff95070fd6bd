"""GPU parity tests, kernel by kernel, through the C ABI (ctypes) against the CPU oracle /
plain fp32 math on the same seeded inputs.  Tolerances: f32 kernels 2e-4 absolute on O(1..10)
values (different summation order only); bf16 kernels are checked against fp32 math on the
SAME bf16-rounded operands, so what is left is fp32-accumulation order plus one final
rounding to bf16 (half an ulp = 2^-9 relative)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from dfd_clip_amd import capi as c
    c.load_library()
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    assert c.load_library().dfd_device_check() == 0, c.load_library().dfd_last_error()
    return c


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * scale)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def assert_close(got, want, atol, rtol=0.0, msg=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    err = (got - want).abs()
    lim = atol + rtol * want.abs()
    assert torch.isfinite(got).all(), f"{msg}: non-finite output"
    assert (err <= lim).all(), f"{msg}: max err {err.max().item():.3e} (limit {lim.min().item():.3e}) at {err.argmax().item()}"


@pytest.mark.parametrize("rows,cols", [(1, 128), (7, 768), (1000, 768), (33, 1024), (5, 256)])
def test_layernorm(capi, rows, cols):
    x = rnd(rows, cols, seed=1, scale=3.0) + 0.5
    g, b = 1 + 0.1 * rnd(cols, seed=2), 0.1 * rnd(cols, seed=3)
    want = F.layer_norm(x, (cols,), g, b, 1e-5)
    xd, gd, bd = x.cuda(), g.cuda(), b.cuda()
    out = torch.empty_like(xd)
    capi.layernorm(xd, gd, bd, out)
    assert_close(out, want, 2e-5, msg="f32")
    outb = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
    capi.layernorm(xd, gd, bd, outb)
    assert_close(outb, want, 1e-5, rtol=2 ** -8, msg="bf16")
    capi.layernorm(xd, gd, bd, xd)  # in place
    assert_close(xd, want, 2e-5, msg="in place")


@pytest.mark.parametrize("rows,cols", [(1, 128), (7, 768), (1000, 768), (33, 1024), (257, 2048)])
@pytest.mark.parametrize("delta_dtype", [torch.float32, torch.bfloat16])
def test_add_layernorm(capi, rows, cols, delta_dtype):
    """x += delta in place, y = LayerNorm(x): the deferred residual of the bf16 encoder path."""
    x = rnd(rows, cols, seed=1, scale=3.0) + 0.5
    d = rnd(rows, cols, seed=4).to(delta_dtype)
    g, b = 1 + 0.1 * rnd(cols, seed=2), 0.1 * rnd(cols, seed=3)
    x_new = x + d.float()
    want = F.layer_norm(x_new, (cols,), g, b, 1e-5)
    for out_dtype, tol in ((torch.float32, dict(atol=2e-5)), (torch.bfloat16, dict(atol=1e-5, rtol=2 ** -8))):
        xd = x.clone().cuda()
        out = torch.empty(rows, cols, device="cuda", dtype=out_dtype)
        capi.add_layernorm(xd, d.cuda(), g.cuda(), b.cuda(), out)
        assert torch.equal(xd.cpu(), x_new), "x must hold the exact fp32 sum"
        assert_close(out, want, msg=str(out_dtype), **tol)
    with pytest.raises(capi.DfdError):
        xd = x.clone().cuda()
        capi.add_layernorm(xd, d.cuda(), g.cuda(), b.cuda(), xd)  # y aliasing x is refused
    # two deltas, and the non-storing form ln_2 uses
    d2 = rnd(rows, cols, seed=5).to(delta_dtype)
    x2 = (x + d.float()) + d2.float()
    xd = x.clone().cuda()
    out = torch.empty(rows, cols, device="cuda")
    capi.add_layernorm(xd, d.cuda(), g.cuda(), b.cuda(), out, delta2=d2.cuda())
    assert torch.equal(xd.cpu(), x2)
    assert_close(out, F.layer_norm(x2, (cols,), g, b, 1e-5), 2e-5, msg="two deltas")
    xd = x.clone().cuda()
    capi.add_layernorm(xd, d.cuda(), g.cuda(), b.cuda(), out, store_x=False)
    assert torch.equal(xd.cpu(), x), "store_x=False must leave x untouched"
    assert_close(out, want, 2e-5, msg="no store")


@pytest.mark.parametrize("res,patch,width", [(32, 16, 128), (224, 16, 256), (224, 14, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_embed_is_conv1_plus_cls_plus_pos(capi, res, patch, width, dtype):
    n = 3
    P = (res // patch) ** 2
    tokens = P + 1
    frames = rnd(n, 3, res, res, seed=4)
    w = rnd(width, 3, patch, patch, seed=5, scale=(3 * patch * patch) ** -0.5)
    cls, pos = rnd(width, seed=6), rnd(tokens, width, seed=7)
    fr, wr = (frames, w) if dtype == torch.float32 else (bf16_round(frames), bf16_round(w))
    y = F.conv2d(fr, wr, None, stride=patch).reshape(n, width, -1).permute(0, 2, 1)
    want = torch.cat([cls.view(1, 1, -1).expand(n, 1, width), y], dim=1) + pos
    kreal = 3 * patch * patch
    kpad = (kreal + 31) // 32 * 32
    patches = torch.zeros(n * P, kpad, device="cuda", dtype=dtype)
    capi.patchify(frames.cuda(), patches, res, patch)
    wp = torch.zeros(width, kpad)
    wp[:, :kreal] = w.reshape(width, kreal)
    x = torch.zeros(n * tokens, width, device="cuda")
    capi.gemm(patches, wp.to(dtype).cuda(), x, None, capi.EPI_PATCH_EMBED, pos=pos.cuda(), cls=cls.cuda(), tokens=tokens)
    assert_close(x.view(n, tokens, width), want, 2e-4 if dtype == torch.float32 else 2e-3, msg="patch embed")


@pytest.mark.parametrize("rows,cols", [(5, 128), (1000, 768), (333, 1024), (7, 2048)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm2_is_two_layernorms(capi, rows, cols, dtype):
    """dfd_layernorm2 (ln_pre + the first block's ln_1 in one pass over the rows) against two dfd_layernorm calls: the same
    bits in x and in the output."""
    x0 = rnd(rows, cols, seed=61).cuda()
    ga, ba, gb, bb = (rnd(cols, seed=62 + i, scale=0.5).cuda() + (1.0 if i % 2 == 0 else 0.0) for i in range(4))
    xa, ya = x0.clone(), torch.empty(rows, cols, device="cuda", dtype=dtype)
    capi.layernorm(xa, ga, ba, xa)
    capi.layernorm(xa, gb, bb, ya)
    xb, yb = x0.clone(), torch.empty(rows, cols, device="cuda", dtype=dtype)
    capi.layernorm2(xb, ga, ba, gb, bb, yb)
    assert torch.equal(xa, xb) and torch.equal(ya, yb)
    if dtype == torch.float32:
        with pytest.raises(capi.DfdError):
            capi.layernorm2(xb, ga, ba, gb, bb, xb)  # y aliasing x is refused


@pytest.mark.parametrize("n,res,patch", [(3, 32, 16), (5, 224, 16), (2, 224, 32), (2, 64, 8)])
def test_patchify_strip_kernel_matches_the_general_one(capi, n, res, patch):
    """bf16 patches of 4-aligned patch sizes come from the strip kernel (whole image rows in, one contiguous run out,
    transposed through LDS); the f32 output takes the general kernel: same values, rounded once."""
    frames = rnd(n, 3, res, res, seed=41).cuda()
    P, kk = (res // patch) ** 2, 3 * patch * patch
    a = torch.zeros(n * P, kk, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(n * P, kk, device="cuda", dtype=torch.float32)
    capi.patchify(frames, a, res, patch)
    capi.patchify(frames, b, res, patch)
    assert torch.equal(a, b.to(torch.bfloat16))
    want = frames.view(n, 3, res // patch, patch, res // patch, patch).permute(0, 2, 4, 1, 3, 5).reshape(n * P, kk)
    assert torch.equal(b, want)


@pytest.mark.parametrize("M,N,K", [(5, 8, 32), (300, 200, 64), (591, 384, 128), (1000, 2304, 768), (257, 768, 3072)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(capi, M, N, K, dtype):
    a, w, bias = rnd(M, K, seed=8), rnd(N, K, seed=9, scale=K ** -0.5), rnd(N, seed=10, scale=0.1)
    ar, wr = (a, w) if dtype == torch.float32 else (bf16_round(a), bf16_round(w))
    ref = ar.double() @ wr.double().T + bias.double()
    atol = 1e-4 if dtype == torch.float32 else 1e-4
    rtol = 1e-5 if dtype == torch.float32 else 2 ** -8
    ad, wd, bd = a.to(dtype).cuda(), w.to(dtype).cuda(), bias.cuda()
    c = torch.empty(M, N, device="cuda", dtype=dtype)
    capi.gemm(ad, wd, c, bd, capi.EPI_BIAS)
    assert_close(c, ref, atol, rtol, "bias")
    capi.gemm(ad, wd, c, bd, capi.EPI_BIAS_QUICKGELU)
    assert_close(c, ref * torch.sigmoid(1.702 * ref), atol, rtol, "quickgelu")
    x0 = rnd(M, N, seed=11)
    x = x0.clone().cuda()
    capi.gemm(ad, wd, x, bd, capi.EPI_BIAS_RESIDUAL)
    assert_close(x, x0.double() + ref, 2e-4, 1e-5, "residual")
    if dtype == torch.bfloat16:  # bf16 operands, f32 output
        cf = torch.empty(M, N, device="cuda")
        capi.gemm(ad, wd, cf, bd, capi.EPI_BIAS)
        assert_close(cf, ref, 1e-4, 1e-5, "bf16->f32")


@pytest.mark.parametrize("M,N", [(300 * 256 + 77, 256), (131 * 256 + 5, 768), (94560, 768)])
def test_gemm_large_m(capi, M, N):
    """Large-M bf16 GEMMs on the tuned 256x256 kernel (more tiles than CUs, ragged last row tile):
    checked on a sample of rows plus the whole tail region, every epilogue with a bf16 store."""
    K = 128
    g = torch.Generator(device="cuda").manual_seed(M)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    rows = torch.cat([torch.randint(0, M, (4096,), device="cuda", generator=g), torch.arange(M - 70000 if M > 70000 else 0, M, device="cuda")])
    ref = a[rows].float() @ w.float().T + bias
    c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    capi.gemm(a, w, c, bias, capi.EPI_BIAS)
    assert torch.isfinite(c.float()).all(), "a tile was not written"
    assert_close(c[rows], ref, 1e-4, 2 ** -8, "bias")
    capi.gemm(a, w, c, bias, capi.EPI_BIAS_QUICKGELU)
    assert_close(c[rows], ref * torch.sigmoid(1.702 * ref), 1e-4, 2 ** -8, "quickgelu")
    if N % 768 == 0:  # QKV export epilogue with both passes on the K / V column tiles
        tokens, T = 197, 3
        Mq = M // tokens * tokens
        D = N // 3
        tpos = torch.randn(T, D, device="cuda", generator=g)
        ke = torch.full((Mq // tokens * (tokens - 1), D), float("nan"), device="cuda", dtype=torch.bfloat16)
        ve = torch.full_like(ke, float("nan"))
        cq = torch.full((Mq, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm(a[:Mq], w, cq, bias, capi.EPI_QKV_EXPORT, pos=tpos, k_export=ke, v_export=ve, tokens=tokens, frames_per_clip=T)
        full = a[:Mq].float() @ w.float().T + bias
        assert_close(cq, full, 1e-4, 2 ** -8, "qkv")
        fv = full.view(Mq // tokens, tokens, 3, D)
        pos_f = tpos[torch.arange(Mq // tokens, device="cuda") % T].view(-1, 1, D)
        assert_close(ke.view(-1, tokens - 1, D), fv[:, 1:, 1] + pos_f, 1e-4, 2 ** -8, "k export")
        assert_close(ve.view(-1, tokens - 1, D), fv[:, 1:, 2] + pos_f, 1e-4, 2 ** -8, "v export")
        # K and V blocks alone (qkv_first=1, what the last tapped layer runs): identical exports and columns
        ke2, ve2 = torch.full_like(ke, float("nan")), torch.full_like(ve, float("nan"))
        cq2 = torch.full((Mq, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm(a[:Mq], w[D:], cq2[:, D:], bias[D:], capi.EPI_QKV_EXPORT, pos=tpos, k_export=ke2, v_export=ve2, tokens=tokens,
                  frames_per_clip=T, qkv_first=1)
        assert torch.equal(ke2, ke) and torch.equal(ve2, ve) and torch.equal(cq2[:, D:], cq[:, D:])
        assert torch.isnan(cq2[:, :D].float()).all(), "the query block must not be written"


# (N, K) of every encoder GEMM of ViT-B/16 and ViT-L/14 (reference clip/model.py:186, :197, :208-212, :277) plus
# odd step counts; M >= 1024 puts them on the tuned kernel, whose steady-state K loop needs K/64 >= 3
TUNED_SHAPES = [(768, 768), (2304, 768), (3072, 768), (768, 3072), (1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096),
                (768, 192), (1024, 640), (256, 320)]


@pytest.mark.parametrize("N,K", TUNED_SHAPES)
@pytest.mark.parametrize("M", [1024 + 256 * 7 + 77])
def test_gemm_tuned_kernel_every_epilogue(capi, M, N, K):
    """The tuned large-M bf16 kernel at the real K depths (its steady-state loop, odd and even step counts),
    every epilogue, against fp64 on the same bf16-rounded operands: what is left is f32 accumulation order
    and one rounding of the result (half a bf16 ulp = 2^-9; bar rtol 2^-8 + tiny atol)."""
    g = torch.Generator(device="cuda").manual_seed(N * 7 + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    ref = a.double() @ w.double().T + bias.double()
    RT = 2 ** -8
    c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    capi.gemm(a, w, c, bias, capi.EPI_BIAS)
    # K a multiple of 128 and >= 384: the ping-pong kernel (257); other depths: the round-2 persistent kernel (256)
    assert capi.gemm_last_path() == (257 if K % 128 == 0 and K >= 384 and N % 256 == 0 else 256), "this shape must run on a tuned kernel"
    assert_close(c, ref, 1e-4, RT, "bias")
    c.fill_(float("nan"))
    capi.gemm(a, w, c, bias, capi.EPI_BIAS_QUICKGELU)
    assert_close(c, ref * torch.sigmoid(1.702 * ref), 1e-4, RT, "quickgelu")
    cf = torch.full((M, N), float("nan"), device="cuda")
    capi.gemm(a, w, cf, bias, capi.EPI_BIAS)  # f32 store of the same product
    assert_close(cf, ref, 1e-4, 1e-5, "bias, f32 out")
    x0 = torch.randn(M, N, device="cuda", generator=g)
    x = x0.clone()
    capi.gemm(a, w, x, bias, capi.EPI_BIAS_RESIDUAL)  # f32 residual stream read-modify-write
    assert_close(x, x0.double() + ref, 2e-4, 1e-5, "residual")
    # adapter output: bf16 C = residual + acc + pos[frame % T] rounded once (no bias)
    P_, T_ = 16, 3
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    pos = torch.randn(T_, N, device="cuda", generator=g)
    cr = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    capi.gemm(a, w, cr, None, capi.EPI_RESIDUAL_POS, pos=pos, tokens=P_ + 1, frames_per_clip=T_, residual=res)
    frame = (torch.arange(M, device="cuda") // P_) % T_
    assert_close(cr, res.double() + (ref - bias.double()) + pos[frame].double(), 1e-4, RT, "residual + pos")
    if N % 768 == 0 or N % 1024 == 0:
        if N in (2304, 3072) and N // 3 % 256 == 0:  # q | k | v projection with K/V export (+ temporal pos), CLS dropped
            tokens, T = 7, 3
            Mq = M // tokens * tokens
            D = N // 3
            tpos = torch.randn(T, D, device="cuda", generator=g)
            ke = torch.full((Mq // tokens * (tokens - 1), D), float("nan"), device="cuda", dtype=torch.bfloat16)
            ve = torch.full_like(ke, float("nan"))
            cq = torch.full((Mq, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            capi.gemm(a[:Mq], w, cq, bias, capi.EPI_QKV_EXPORT, pos=tpos, k_export=ke, v_export=ve, tokens=tokens, frames_per_clip=T)
            assert_close(cq, ref[:Mq], 1e-4, RT, "qkv")
            fv = ref[:Mq].view(Mq // tokens, tokens, 3, D)
            pos_f = tpos[torch.arange(Mq // tokens, device="cuda") % T].view(-1, 1, D).double()
            assert_close(ke.view(-1, tokens - 1, D), fv[:, 1:, 1] + pos_f, 1e-4, RT, "k export")
            assert_close(ve.view(-1, tokens - 1, D), fv[:, 1:, 2] + pos_f, 1e-4, RT, "v export")
            ke2, ve2 = torch.full_like(ke, float("nan")), torch.full_like(ve, float("nan"))
            cq2 = torch.full((Mq, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            capi.gemm(a[:Mq], w[D:], cq2[:, D:], bias[D:], capi.EPI_QKV_EXPORT, pos=tpos, k_export=ke2, v_export=ve2, tokens=tokens,
                      frames_per_clip=T, qkv_first=1)
            assert torch.equal(ke2, ke) and torch.equal(ve2, ve) and torch.equal(cq2[:, D:], cq[:, D:])


@pytest.mark.parametrize("M,N,K,epi", [(94560, 768, 768, "bias"), (20000, 2304, 768, "qkv"), (70000, 3072, 768, "gelu"), (5000, 1024, 4096, "bias")])
def test_persistent_gemm_variants_are_bit_identical(capi, M, N, K, epi):
    """The persistent kernel's knobs change scheduling, never arithmetic: 224- vs 256-row tiles, non-temporal output
    stores, compute units left to other streams and the automatic choice all give the same bits, at sizes with
    several rounds of tiles per CU (ragged last row panel, last round partly empty)."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    kw, e = {}, capi.EPI_BIAS
    tokens, T = 197, 5
    if epi == "gelu":
        e = capi.EPI_BIAS_QUICKGELU
    if epi == "qkv":
        e = capi.EPI_QKV_EXPORT
        M = M // tokens * tokens
        a = a[:M]
        D = N // 3
        kw = dict(pos=torch.randn(T, D, device="cuda", generator=g), tokens=tokens, frames_per_clip=T)

    expect_path = [257]  # the ping-pong kernel serves all four shapes; variant 1 hands them to the round-2 persistent kernel

    def run(**opts):
        c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ex = {}
        if epi == "qkv":
            ex = dict(k_export=torch.full((M // tokens * (tokens - 1), N // 3), float("nan"), device="cuda", dtype=torch.bfloat16))
            ex["v_export"] = torch.full_like(ex["k_export"], float("nan"))
        capi.gemm(a, w, c, bias, e, **kw, **ex, **opts)
        assert capi.gemm_last_path() == expect_path[0]
        return [c] + list(ex.values())

    base = run(tile_blocks=8)
    rows = torch.randint(0, M, (2048,), device="cuda", generator=g)
    ref = a[rows].double() @ w.double().T + bias.double()
    if epi == "gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    assert_close(base[0][rows], ref, 1e-4, 2 ** -8, "256-row tiles")
    assert all(torch.isfinite(t.float()).all() for t in base), "a tile or an export row was not written"
    for opts in (dict(tile_blocks=7), dict(), dict(tile_blocks=8, stream_out=True), dict(tile_blocks=7, stream_out=True, spare_cus=32),
                 dict(spare_cus=100), dict(spare_cus=32, spare_if_free=True), dict(spare_cus=120, spare_if_free=True)):
        got = run(**opts)
        for x, y in zip(base, got):
            assert torch.equal(x, y), opts
    # the round-2 persistent kernel (same tile, one instruction stream for all waves) gives the same bits
    capi.gemm_set_variant(1)
    expect_path[0] = 256
    try:
        for opts in (dict(tile_blocks=8), dict(tile_blocks=7, stream_out=True)):
            got = run(**opts)
            for x, y in zip(base, got):
                assert torch.equal(x, y), ("persistent", opts)
    finally:
        capi.gemm_set_variant(0)


@pytest.mark.parametrize("M", [1024, 1024 + 96, 256 * 40 + 8, 256 * 300 + 200])
@pytest.mark.parametrize("N,K", [(256, 384), (768, 768), (1024, 4096)])
def test_pingpong_gemm_ragged_panels_and_few_tiles(capi, M, N, K):
    """The ping-pong kernel counts its vector-memory operations by hand.  Shapes that stress the counts: fewer tiles than
    compute units (every workgroup's first tile is its last), a ragged last row panel (stores of rows beyond M are dropped
    by the buffer unit and must not be counted by the next tile's waits), the shortest K it serves (head and tail of the K
    loop back to back), many tiles per workgroup.  Against fp64, and bit for bit against the round-2 kernel, three times
    (a race would not repeat)."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    ref = a.double() @ w.double().T + bias.double()
    capi.gemm_set_variant(1)
    try:
        want = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm(a, w, want, bias, capi.EPI_BIAS_QUICKGELU)
        assert capi.gemm_last_path() == 256
    finally:
        capi.gemm_set_variant(0)
    assert_close(want, ref * torch.sigmoid(1.702 * ref), 1e-4, 2 ** -8, "round-2 kernel")
    for rep in range(3):
        for opts in (dict(), dict(tile_blocks=8, stream_out=True), dict(tile_blocks=7)):
            c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            capi.gemm(a, w, c, bias, capi.EPI_BIAS_QUICKGELU, **opts)
            assert capi.gemm_last_path() == 257
            assert torch.equal(c, want), (rep, opts)


@pytest.mark.parametrize("M,N,K,epi", [(94560, 3072, 768, "gelu"), (50000, 2304, 768, "bias"), (94560, 768, 3072, "bias")])
def test_pingpong_dynamic_tile_handout(capi, M, N, K, epi):
    """Opt-in (variant 3): tiles after a workgroup's first are drawn from per-XCD counters (gemm256e.hip).  Which workgroup
    computes a tile must not matter: the dynamic order and the static order (the default) give the same bits — alone;
    with another stream's kernels holding compute units while the GEMM runs (the situation the hand-out exists for: late
    workgroups take fewer tiles); with two streams launching GEMMs at once (each stream has counters of its own); and
    launch after launch (the counters are never reset, each launch is told where they stand)."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    e = capi.EPI_BIAS_QUICKGELU if epi == "gelu" else capi.EPI_BIAS

    def run(**kw):
        c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm(a, w, c, bias, e, stream_out=True, **kw)
        return c

    want = run()
    assert capi.gemm_last_path() == 257
    rows = torch.randint(0, M, (1024,), device="cuda", generator=g)
    ref = a[rows].double() @ w.double().T + bias.double()
    assert_close(want[rows], ref * torch.sigmoid(1.702 * ref) if epi == "gelu" else ref, 1e-4, 2 ** -8, "static order")
    capi.gemm_set_variant(3)
    try:
        _dynamic_runs(run, want)
    finally:
        capi.gemm_set_variant(0)


def _dynamic_runs(run, want):
    if True:  # (kept as a block: the body is the hand-out exercised three ways)
        for _ in range(5):
            assert torch.equal(run(), want)
        assert torch.equal(run(spare_cus=40), want)
        # other work on the chip while the GEMM runs: a few hundred small kernels on a second stream
        side = torch.cuda.Stream()
        junk = torch.randn(64, 1 << 16, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(300):
                junk = torch.sin(junk) * 1.0001
        got = [run() for _ in range(4)]
        torch.cuda.synchronize()
        for c in got:
            assert torch.equal(c, want)
        # two streams launching GEMMs concurrently
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(3):
            for st in (s1, s2):
                with torch.cuda.stream(st):
                    outs.append(run())
        torch.cuda.synchronize()
        for c in outs:
            assert torch.equal(c, want)


@pytest.mark.parametrize("M,N,K", [(1024 + 96, 768, 256), (30 * 196 * 3 + 17, 768, 256), (9000, 768, 768), (5000, 1024, 384)])
@pytest.mark.parametrize("in_place,p", [(False, 0.0), (True, 0.0), (False, 0.5)])
def test_pingpong_residual_pos_equals_relaunching(capi, M, N, K, in_place, p):
    """The adapter's second Linear (reference models.py:795-875: C = residual + dropout(a . W^T) + pos[frame % T], rounded
    once) on the ping-pong kernel — including its short-K form (K = 256: the shipped adapters' x) — against fp64 and,
    bit for bit (same f32 sums, same mask, same single rounding), against the one-workgroup-per-tile kernel."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    P_, T_ = 196, 3
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    pos = torch.randn(T_, N, device="cuda", generator=g)
    rng = torch.tensor([12345, 7], device="cuda", dtype=torch.int64)
    drop = capi.Dropout(rng, 1001, p) if p > 0 else None

    def run():
        c = res.clone() if in_place else torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm(a, w, c, None, capi.EPI_RESIDUAL_POS, pos=pos, tokens=P_ + 1, frames_per_clip=T_, residual=None if in_place else res, drop=drop)
        return c

    capi.gemm_set_variant(1)
    try:
        want = run()
        assert capi.gemm_last_path() == 256
    finally:
        capi.gemm_set_variant(0)
    if p == 0:
        frame = (torch.arange(M, device="cuda") // P_) % T_
        ref = res.double() + a.double() @ w.double().T + pos[frame].double()
        assert_close(want, ref, 1e-4, 2 ** -8, "residual + pos")
    else:
        kept = (want.float() - res.float() - pos[(torch.arange(M, device="cuda") // P_) % T_]).abs() > 1e-2
        assert 0.4 < kept.float().mean().item() < 0.6, "about half of the elements keep their (doubled) product"
    for _ in range(2):
        got = run()
        assert capi.gemm_last_path() == 257
        assert torch.equal(got, want)


@pytest.mark.parametrize("res,patch,width", [(224, 16, 768), (224, 14, 1024)])
def test_patch_embed_tuned_kernel(capi, res, patch, width):
    """PATCH_EMBED epilogue of the tuned kernel (M = frames*P >= 1024): conv1 + CLS row + positional embedding at
    ViT-B/16's K = 768 and ViT-L/14's K = 588 padded to 640 (reference clip/model.py:277-291)."""
    n = 8
    P = (res // patch) ** 2
    tokens = P + 1
    g = torch.Generator(device="cuda").manual_seed(res + patch)
    frames = torch.randn(n, 3, res, res, device="cuda", generator=g)
    w = torch.randn(width, 3, patch, patch, device="cuda", generator=g) * (3 * patch * patch) ** -0.5
    cls, pos = torch.randn(width, device="cuda", generator=g), torch.randn(tokens, width, device="cuda", generator=g)
    fr, wr = frames.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()
    y = F.conv2d(fr.cpu(), wr.cpu(), None, stride=patch).reshape(n, width, -1).permute(0, 2, 1).cuda()
    want = torch.cat([cls.view(1, 1, -1).expand(n, 1, width).double(), y], dim=1) + pos.double()
    kreal = 3 * patch * patch
    kpad = (kreal + 63) // 64 * 64
    patches = torch.zeros((n * P + 255) // 256 * 256, kpad, device="cuda", dtype=torch.bfloat16)
    capi.patchify(frames, patches, res, patch)
    wp = torch.zeros(width, kpad, device="cuda")
    wp[:, :kreal] = w.reshape(width, kreal)
    x = torch.full((n * tokens, width), float("nan"), device="cuda")
    capi.gemm(patches, wp.to(torch.bfloat16), x, None, capi.EPI_PATCH_EMBED, m=n * P, pos=pos, cls=cls, tokens=tokens)
    assert capi.gemm_last_path() == 256
    assert_close(x.view(n, tokens, width), want, 2e-4, 1e-5, "patch embed")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_qkv_export_layout(capi, dtype):
    """K/V column blocks land in [frames*P, D] with the CLS row dropped and pos[frame % T] added."""
    n, tokens, D, T = 6, 5, 128, 3
    M = n * tokens
    a, w, bias = rnd(M, D, seed=12), rnd(3 * D, D, seed=13, scale=D ** -0.5), rnd(3 * D, seed=14, scale=0.1)
    tpos = rnd(T, D, seed=15)
    ar, wr = (a, w) if dtype == torch.float32 else (bf16_round(a), bf16_round(w))
    ref = (ar.double() @ wr.double().T + bias.double()).float().view(n, tokens, 3, D)
    c = torch.empty(M, 3 * D, device="cuda", dtype=dtype)
    ke = torch.zeros(n * (tokens - 1), D, device="cuda", dtype=dtype)
    ve = torch.zeros_like(ke)
    capi.gemm(a.to(dtype).cuda(), w.to(dtype).cuda(), c, bias.cuda(), capi.EPI_QKV_EXPORT, pos=tpos.cuda(), k_export=ke,
              v_export=ve, tokens=tokens, frames_per_clip=T)
    tol = dict(atol=1e-4, rtol=1e-5) if dtype == torch.float32 else dict(atol=1e-4, rtol=2 ** -8)
    assert_close(c.view(n, tokens, 3, D), ref, msg="qkv", **tol)
    pos_f = tpos[torch.arange(n) % T].view(n, 1, D)
    assert_close(ke.view(n, tokens - 1, D), ref[:, 1:, 1] + pos_f, msg="k export", **tol)
    assert_close(ve.view(n, tokens - 1, D), ref[:, 1:, 2] + pos_f, msg="v export", **tol)
    c2 = torch.empty_like(c)
    capi.gemm(a.to(dtype).cuda(), w.to(dtype).cuda(), c2, bias.cuda(), capi.EPI_QKV_EXPORT, tokens=tokens)  # no export
    assert torch.equal(c, c2)
    # K | V blocks only (general kernel at this size)
    ke2, ve2, c3 = torch.zeros_like(ke), torch.zeros_like(ve), torch.zeros_like(c)
    wd, bd = w.to(dtype).cuda(), bias.cuda()
    capi.gemm(a.to(dtype).cuda(), wd[D:], c3[:, D:], bd[D:], capi.EPI_QKV_EXPORT, pos=tpos.cuda(), k_export=ke2, v_export=ve2,
              tokens=tokens, frames_per_clip=T, qkv_first=1)
    assert torch.equal(ke2, ke) and torch.equal(ve2, ve) and torch.equal(c3[:, D:], c[:, D:]) and (c3[:, :D] == 0).all()


@pytest.mark.parametrize("n,tokens,heads", [(2, 5, 2), (3, 197, 4), (1, 257, 2), (2, 50, 12), (64, 197, 12), (45, 200, 12),
                                            (43, 205, 12), (48, 210, 12), (33, 257, 16), (47, 230, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_attention(capi, n, tokens, heads, dtype):
    D = heads * 64
    qkv = rnd(n * tokens, 3 * D, seed=16)
    qkv[:, :D] *= 2.0  # peaky softmax rows
    q = qkv if dtype == torch.float32 else bf16_round(qkv)
    t = q.view(n, tokens, 3, heads, 64)
    aff = torch.einsum("nqhc,nkhc->nqkh", t[:, :, 0] / 8.0, t[:, :, 1]).softmax(dim=-2)
    want = torch.einsum("nqlh,nlhc->nqhc", aff, t[:, :, 2]).reshape(n * tokens, D)
    out = torch.empty(n * tokens, D, device="cuda", dtype=dtype)
    capi.attention_fwd(qkv.to(dtype).cuda(), out, n, tokens, heads)
    if dtype == torch.float32:
        assert_close(out, want, 2e-5, 1e-5, "attention f32")
    else:  # P is rounded to bf16 before the PV product in the MFMA kernel
        assert_close(out, want, 2e-2, 2 ** -7, "attention bf16")


@pytest.mark.parametrize("n,tokens,heads", [(64, 197, 12), (45, 200, 12), (43, 205, 12), (40, 257, 16), (45, 288, 12), (50, 226, 12)])
def test_encoder_attention_persistent_kernel_matches_per_item_kernel(capi, n, tokens, heads):
    """>= 512 (frame, head) items of 193..208 tokens take the persistent kernel (loader wave + LDS-DMA, two barriers
    per item); the same frames in chunks below that threshold — and every other token count — take the
    one-workgroup-per-item kernel: same arithmetic, same bits, whatever the launch geometry."""
    D = heads * 64
    qkv = rnd(n * tokens, 3 * D, seed=23).to(torch.bfloat16).cuda()
    qkv[:, :D] *= 2.0
    whole = torch.empty(n * tokens, D, device="cuda", dtype=torch.bfloat16)
    capi.attention_fwd(qkv, whole, n, tokens, heads)
    parts = torch.empty_like(whole)
    step = 16  # at most 256 items per call
    for f0 in range(0, n, step):
        f1 = min(n, f0 + step)
        capi.attention_fwd(qkv[f0 * tokens:f1 * tokens], parts[f0 * tokens:f1 * tokens], f1 - f0, tokens, heads)
    assert torch.equal(whole, parts)
    again = torch.empty_like(whole)
    capi.attention_fwd(qkv, again, n, tokens, heads)
    assert torch.equal(whole, again)


def test_encoder_attention_257_tokens_stress(capi):
    """ViT-L/14's shape at BASELINE configs[3] size (240 frames x 16 heads = 3,840 items of 257 tokens) through the
    persistent 257-token kernel (attention_mfma_xrow.hip), three launches on fresh inputs, each compared bit for bit with
    the per-item kernel.  This is the case that showed round 3's store hazard (a 16-byte-per-lane store with a register
    in its scalar-offset field going out with the next instruction's result in its first data register: sporadic, 32
    elements of a tile at a time; csrc/attention_common.hpp, tools/isa_lint.py)."""
    n, tokens, heads = 240, 257, 16
    D = heads * 64
    for seed in (31, 32, 33):
        qkv = rnd(n * tokens, 3 * D, seed=seed).to(torch.bfloat16).cuda()
        whole = torch.empty(n * tokens, D, device="cuda", dtype=torch.bfloat16)
        capi.attention_fwd(qkv, whole, n, tokens, heads)
        parts = torch.empty_like(whole)
        step = 16  # 256 items per call: below the persistent kernel's threshold
        for f0 in range(0, n, step):
            capi.attention_fwd(qkv[f0 * tokens:(f0 + step) * tokens], parts[f0 * tokens:(f0 + step) * tokens], step, tokens, heads)
        bad = (whole != parts)
        assert not bad.any(), f"seed {seed}: {int(bad.sum())} elements differ, first at {bad.nonzero()[0].tolist()}"


@pytest.mark.parametrize("B,T,P,heads", [(2, 4, 4, 2), (3, 3, 196, 4), (2, 8, 196, 12), (1, 5, 256, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_attention(capi, B, T, P, heads, dtype):
    D, S = heads * 64, T * P
    k, v = rnd(B, S, heads, 64, seed=17), rnd(B, S, heads, 64, seed=18)
    q = rnd(B, 1, heads, 128, seed=19)
    m = torch.ones(B, T, dtype=torch.bool)
    if B > 1:
        m[1, T - max(1, T // 4):] = False
    kr, vr = (k, v) if dtype == torch.float32 else (bf16_round(k), bf16_round(v))
    # oracle attention with identity projections: feed q through in_proj = I
    w = {"p.attn.in_proj.weight": torch.eye(2 * D), "p.attn.in_proj.bias": torch.zeros(2 * D),
         "p.attn.out_proj.weight": torch.eye(D), "p.attn.out_proj.bias": torch.zeros(D)}
    want = ref_cpu.decoder_attention(q.reshape(B, 1, 2 * D), kr, vr, m.repeat_interleave(P, dim=-1), w, "p.", heads, T)
    splits = 3
    ws = torch.empty(capi.decoder_attn_workspace_bytes(B, heads, 64, splits) // 4, device="cuda")
    mix = torch.empty(B, D, device="cuda")
    stats = torch.empty(B, heads, 2, device="cuda")
    capi.decoder_attn_fwd(q.reshape(B, 2 * D).cuda(), k.to(dtype).reshape(B, S, D).cuda(), v.to(dtype).reshape(B, S, D).cuda(),
                          m.to(torch.uint8).cuda(), mix, stats, ws, splits, B, T, P, heads)
    assert_close(mix, want.reshape(B, D), 2e-5, 1e-4, "decoder attention")
    # (max, sumexp) of the softmax branch
    s = torch.einsum("bhc,bshc->bsh", q[:, 0, :, :64] / 8.0, kr).masked_fill(~m.repeat_interleave(P, dim=-1).unsqueeze(-1), -math.inf)
    assert_close(stats[..., 0], s.max(dim=1).values, 1e-5, 1e-5, "row max")
    assert_close(stats[..., 1], (s - s.max(dim=1, keepdim=True).values).exp().sum(dim=1), 1e-4, 1e-4, "sumexp")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("modes", [0, 3])
def test_decoder_attention_reads_keys_and_values_in_place(capi, dtype, modes):
    """dfd_kv_layout_t: keys / values read straight out of a q|k|v activation [frames, tokens, 3D] (CLS row skipped,
    row stride 3D) with the temporal positional embedding added on the fly must give, bit for bit, what the dense
    f32 export (k + pos) gives: forward mix / stats, attn_mode scores and weights, backward dq / dpos."""
    B, T, P, H = 3, 4, 20, 4
    D, tok, S = H * 64, P + 1, T * P
    qkv = rnd(B * T, tok, 3 * D, seed=51).to(dtype).cuda()
    pos = rnd(T, D, seed=52, scale=0.3).cuda()
    kview, vview = qkv[:, 1:, D:2 * D], qkv[:, 1:, 2 * D:]
    pb = pos.repeat(B, 1).view(B * T, 1, D)
    kd, vd = (kview.float() + pb).contiguous().view(B * S, D), (vview.float() + pb).contiguous().view(B * S, D)
    q = rnd(B, 2 * D, seed=53).cuda()
    m = torch.ones(B, T, dtype=torch.uint8)
    m[1, T - 1:] = 0
    m = m.cuda()
    dmix = rnd(B, D, seed=54).cuda()
    splits = 2
    f32 = dict(device="cuda", dtype=torch.float32)

    def run(k, v, p):
        ws = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4, **f32)
        mix, mix_s, stats = torch.empty(B, D, **f32), torch.empty(B, D, **f32), torch.empty(B, H, 2, **f32)
        sc = aw = dsc = None
        if modes:
            sc, aw, dsc = torch.empty(B, H, S, **f32), torch.empty(B, H, S, **f32), torch.empty(B, H, S, **f32)
            capi.decoder_attn_modes_fwd(q, k, m, modes, sc, aw, B, T, P, H, pos=p)
        capi.decoder_attn_fwd(q, k, v, m, mix, stats, ws, splits, B, T, P, H, mix_softmax=mix_s, ext_weights=aw, pos=p)
        if modes:
            capi.decoder_attn_modes_bwd(sc, v, dmix, modes, torch.empty(B, H, S, **f32), dsc, B, T, P, H, pos=p)
        ws2 = torch.empty(capi.decoder_attn_bwd_workspace_bytes(B, T, H) // 4, **f32)
        dq, dpos = torch.empty(B, 2 * D, **f32), torch.empty(T, D, **f32)
        capi.decoder_attn_bwd(q, k, v, m, dmix, None if modes else mix_s, None if modes else stats, dq, dpos, ws2, B, T, P, H,
                              ext_weights=aw, ext_dscores=dsc, pos=p)
        return [t for t in (mix, None if modes else stats, sc, aw, dq, dpos) if t is not None]

    bits = lambda t: t.view(torch.int32)  # bitwise: a fully padded frame's "frame" softmax is NaN on both sides
    got, want = run(kview, vview, pos), run(kd, vd, None)
    for a, b in zip(got, want):
        assert torch.equal(bits(a), bits(b))
    # a dense tensor with `pos` given takes the same path
    got2 = run(kview.contiguous().view(B * S, D), vview.contiguous().view(B * S, D), pos)
    for a, b in zip(got2, want):
        assert torch.equal(bits(a), bits(b))
    # rows that would not be 16-byte aligned are refused, not misread
    if modes == 0:
        odd = torch.zeros(B * T, tok, 3 * D + 2, device="cuda", dtype=dtype)
        ws = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4, **f32)
        with pytest.raises(capi.DfdError):
            capi.decoder_attn_fwd(q, odd[:, 1:, D:2 * D], odd[:, 1:, 2 * D:2 * D + D], m, torch.empty(B, D, **f32),
                                  torch.empty(B, H, 2, **f32), ws, splits, B, T, P, H, pos=pos)


@pytest.mark.parametrize("B,N,K", [(1, 8, 128), (2, 256, 128), (16, 1536, 768), (16, 768, 3072), (9, 100, 64)])
def test_linear_rows(capi, B, N, K):
    x, w, b = rnd(B, K, seed=20), rnd(N, K, seed=21, scale=K ** -0.5), rnd(N, seed=22, scale=0.1)
    ref = x.double() @ w.double().T + b.double()
    y = torch.empty(B, N, device="cuda")
    capi.linear_rows(x.cuda(), w.cuda(), b.cuda(), y)
    assert_close(y, ref, 2e-5, 1e-5, "bias")
    capi.linear_rows(x.cuda(), w.cuda(), b.cuda(), y, capi.EPI_BIAS_QUICKGELU)
    assert_close(y, ref * torch.sigmoid(1.702 * ref), 2e-5, 1e-5, "quickgelu")
    y0 = rnd(B, N, seed=23)
    y = y0.clone().cuda()
    capi.linear_rows(x.cuda(), w.cuda(), b.cuda(), y, capi.EPI_BIAS_RESIDUAL)
    assert_close(y, y0.double() + ref, 2e-5, 1e-5, "residual")


@pytest.mark.parametrize("B,D,od", [(2, 128, 2), (16, 768, 2), (3, 1024, 140)])
def test_head(capi, B, D, od):
    x, g, b, proj = rnd(B, D, seed=24, scale=2.0), 1 + 0.1 * rnd(D, seed=25), 0.1 * rnd(D, seed=26), rnd(D, od, seed=27, scale=D ** -0.5)
    feat_ref = F.layer_norm(x, (D,), g, b, 1e-5)
    z = feat_ref @ proj
    feat = torch.empty(B, D, device="cuda")
    raw = torch.empty(B, od, device="cuda")
    logits = torch.empty(B, od, device="cuda")
    capi.head_fwd(x.cuda(), g.cuda(), b.cuda(), proj.cuda(), feat, raw, logits)
    assert_close(feat, feat_ref, 2e-5, msg="feature")
    assert_close(raw, z, 2e-5, 1e-5, "raw logits")
    assert_close(logits, ref_cpu.normalise_logits(z), 5e-5, 1e-5, "logits")
    assert_close(logits.norm(dim=-1), torch.full((B,), 5.0), 1e-4, msg="norm 5")


@pytest.mark.parametrize("B,N,K", [(1, 8, 128), (2, 256, 128), (16, 1536, 768), (16, 768, 3072), (16, 3072, 768), (9, 100, 64), (20, 768, 768)])
def test_linear_rows_t(capi, B, N, K):
    """Row-streaming form on the transposed weight: same results as x @ W^T + b, all three epilogues."""
    x, w, b = rnd(B, K, seed=28), rnd(N, K, seed=29, scale=K ** -0.5), rnd(N, seed=30, scale=0.1)
    ref = x.double() @ w.double().T + b.double()
    wt = w.T.contiguous().cuda()
    ws = torch.empty(capi.linear_rows_t_workspace_bytes(B, N, K) // 4, device="cuda")
    y = torch.empty(B, N, device="cuda")
    capi.linear_rows_t(x.cuda(), wt, b.cuda(), y, ws)
    assert_close(y, ref, 2e-5, 1e-5, "bias")
    capi.linear_rows_t(x.cuda(), wt, b.cuda(), y, ws, capi.EPI_BIAS_QUICKGELU)
    assert_close(y, ref * torch.sigmoid(1.702 * ref), 2e-5, 1e-5, "quickgelu")
    y0 = rnd(B, N, seed=31)
    y = y0.clone().cuda()
    capi.linear_rows_t(x.cuda(), wt, None, y, ws, capi.EPI_BIAS_RESIDUAL)
    assert_close(y, y0.double() + ref - b.double(), 2e-5, 1e-5, "residual, no bias")
    y3, res = torch.empty(B, N, device="cuda"), y0.cuda()
    capi.linear_rows_t(x.cuda(), wt, None, y3, ws, capi.EPI_BIAS_RESIDUAL, residual=res)  # residual as a separate input
    assert torch.equal(y3, y) and torch.equal(res, y0.cuda())
    y2 = torch.empty(B, N, device="cuda")
    capi.linear_rows_t(x.cuda(), wt, b.cuda(), y2, ws)
    capi.linear_rows_t(x.cuda(), wt, b.cuda(), y, ws)
    assert torch.equal(y, y2)  # deterministic
