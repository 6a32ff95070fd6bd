"""fp8 (OCP e4m3) encoder GEMMs — BASELINE.json configs[4] — through the C ABI (`dfd_gemm_fp8`): the block-scaled matrix
cores on e4m3 operands against fp64 on the SAME e4m3 values (so what is left is f32 accumulation order and one rounding of
the output), every epilogue; then the encoder-level policy (static per-tensor activation scales from a calibration batch,
per-output-channel weight scales) against the bf16 path: logits, AUROC and rank correlation on synthetic clips."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from dfd_clip_amd import capi as c
    c.load_library()
    return c


def e4m3(t):
    """float tensor (CPU) -> (e4m3 bytes on the GPU, the values those bytes encode as f64 on the GPU)."""
    q = t.to(torch.float8_e4m3fn)
    return q.view(torch.uint8).cuda(), q.to(torch.float64).cuda()


def assert_close(got, want, atol, rtol, msg):
    """atol may be a tensor (per-element bound)."""
    got, want = got.double(), want.double()
    err = (got - want).abs()
    lim = atol + rtol * want.abs()
    assert torch.isfinite(got).all(), f"{msg}: non-finite output"
    assert (err <= lim).all(), f"{msg}: max err {err.max().item():.3e} at {err.argmax().item()}"


@pytest.mark.parametrize("M,N,K", [(1024, 256, 256), (2893, 768, 768), (3000, 3072, 1024), (1500, 1024, 4096), (94560, 768, 768)])
def test_gemm_fp8_exact_integers(capi, M, N, K):
    """Small integers are exact in e4m3 and their products / sums exact in f32: any mistake in the operand layout
    (which k a lane's 32 bytes stand for, row <-> column of the accumulator) shows as an O(1) error.  Asymmetric
    operands; every row and column checked."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    w[:, ::7] += 1.0  # asymmetric in k
    a8, af = e4m3(a)
    w8, wf = e4m3(w)
    cs = torch.ones(N, device="cuda")
    c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    capi.gemm_fp8(a8, w8, c, cs)
    rows = torch.cat([torch.arange(0, min(M, 600)), torch.arange(max(0, M - 300), M)]).cuda()
    ref = af[rows] @ wf.T
    assert_close(c[rows], ref, 0.0, 2 ** -8, "integer product")
    assert torch.isfinite(c.float()).all(), "a tile was not written"


@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (3072, 1024), (1024, 4096)])
def test_gemm_fp8_epilogues(capi, N, K):
    M = 1024 + 256 * 7 + 77
    g = torch.Generator().manual_seed(N * 3 + K)
    a8, af = e4m3(torch.randn(M, K, generator=g) * 4.0)
    w8, wf = e4m3(torch.randn(N, K, generator=g) * 8.0)
    cs = (torch.rand(N, generator=g) * 0.02 + 0.001).cuda()  # activation scale x per-row weight scale
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    ref = (af @ wf.T) * cs.double() + bias.double()
    RT = 2 ** -8
    # The instruction does not sum its 128 products as an f32 chain: it aligns them to the largest one and keeps about
    # 12 bits below it (measured with tools/lab/f8_accum_probe.py: excess error <= 2.6e-4 of sum|a_k w_k| = 2^-12, up to
    # 2-5 % of the largest product; small integers, which need no alignment, are exact — the test above).  Against the e4m3
    # quantisation noise of real operands (2^-4 per element, ~4e-3 of sum|a_k w_k| at K = 768) that is negligible, but it
    # is what bounds this comparison: per element 2^-11 of sum|a_k w_k|, on top of the output rounding.
    AT = 2.0 ** -11 * (af.abs() @ wf.abs().T) * cs.double() + 1e-4
    c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    capi.gemm_fp8(a8, w8, c, cs, bias)
    assert_close(c, ref, AT, RT, "bias")
    c.fill_(float("nan"))
    capi.gemm_fp8(a8, w8, c, cs, bias, capi.EPI_BIAS_QUICKGELU)
    gelu = ref * torch.sigmoid(1.702 * ref)
    assert_close(c, gelu, 1.1 * AT, RT, "quickgelu")  # |g'| <= 1.1
    # e4m3 output (c_fc -> c_proj): value * out_inv_scale, saturated; compare after decoding, one e4m3 rounding (2^-4 rel)
    out_scale = float(gelu.abs().max()) / 448.0
    c8 = torch.zeros(M, N, device="cuda", dtype=torch.uint8)
    capi.gemm_fp8(a8, w8, c8, cs, bias, capi.EPI_BIAS_QUICKGELU, out_inv_scale=1.0 / out_scale)
    dec = c8.view(torch.float8_e4m3fn).double() * out_scale
    assert_close(dec, gelu, 1.1 * AT + 2.0 ** -10 * out_scale, 2 ** -4, "quickgelu -> e4m3")  # + half a subnormal step (2^-9)
    if N % 768 == 0 and N // 3 % 256 == 0:  # q | k | v with export
        tokens, T = 7, 3
        Mq = M // tokens * tokens
        D = N // 3
        tpos = torch.randn(T, D, generator=g).cuda()
        ke = torch.full((Mq // tokens * (tokens - 1), D), float("nan"), device="cuda", dtype=torch.bfloat16)
        ve = torch.full_like(ke, float("nan"))
        cq = torch.full((Mq, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm_fp8(a8[:Mq], w8, cq, cs, bias, capi.EPI_QKV_EXPORT, pos=tpos, k_export=ke, v_export=ve, tokens=tokens, frames_per_clip=T)
        assert_close(cq, ref[:Mq], AT[:Mq], RT, "qkv")
        fv = ref[:Mq].view(Mq // tokens, tokens, 3, D)
        av = AT[:Mq].view(Mq // tokens, tokens, 3, D)
        pos_f = tpos[torch.arange(Mq // tokens, device="cuda") % T].view(-1, 1, D).double()
        assert_close(ke.view(-1, tokens - 1, D), fv[:, 1:, 1] + pos_f, av[:, 1:, 1], RT, "k export")
        assert_close(ve.view(-1, tokens - 1, D), fv[:, 1:, 2] + pos_f, av[:, 1:, 2], RT, "v export")


def test_gemm_fp8_rejects_unserved_shapes(capi):
    a = torch.zeros(1024, 192, device="cuda", dtype=torch.uint8)
    w = torch.zeros(256, 192, device="cuda", dtype=torch.uint8)
    c = torch.zeros(1024, 256, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(capi.DfdError, match="not served"):
        capi.gemm_fp8(a, w, c, torch.ones(256, device="cuda"))
