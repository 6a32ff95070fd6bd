"""Train-mode dropout on the HIP path (reference `src/models.py:163`, `:294`, `:304` decoder; `:804-912` adapter;
every `configs/deepfake/*.yaml` sets `dropout: 0.5`).  torch.nn.Dropout's masks are not reproducible across
implementations, so the checks are: (1) the kernel's mask equals the oracle's numpy restatement of the same
counter-based generator bit for bit; (2) keep rate and scaling within binomial bounds; (3) eval() is unchanged
bit for bit and p = 0 is the identity; (4) the same seed gives the same step, another seed or step another mask;
(5) losses and gradients of a train-mode step equal autograd of the CPU oracle GIVEN the same masks; (6) HIP-graph
replay draws fresh masks each step and equals the eager path."""
import copy

import numpy as np
import pytest
import torch

from oracle import dropout_mask, ref_cpu
from tests.cases import build_case, oracle_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from dfd_clip_amd import capi as c
    c.load_library()
    return c


def rng_state(seed, step):
    return torch.tensor([seed, step], dtype=torch.int64, device="cuda")


@pytest.mark.parametrize("n", [1, 7, 8, 1000, 4099, 1 << 20])
@pytest.mark.parametrize("p", [0.5, 0.05, 0.1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_kernel_equals_oracle_mask(capi, n, p, dtype):
    seed, step, site = 0x1234_5678_9ABC, 41, 7
    x = torch.ones(n, device="cuda", dtype=dtype)
    y = torch.empty(n, device="cuda", dtype=torch.float32)
    capi.dropout(x, y, capi.Dropout(rng_state(seed, step), site, p))
    want = dropout_mask.multiplier(n, p, seed, step, site)
    assert np.array_equal(y.cpu().numpy(), want)
    if n >= 1 << 20:
        keep = float((want > 0).mean())
        sigma = (p * (1 - p) / n) ** 0.5
        assert abs(keep - (1 - p)) < 5 * sigma + 2e-5, (keep, p)  # 2e-5: p is quantised to 1/65536
        assert abs(float(want.mean()) - 1.0) < 5 * sigma / (1 - p) + 1e-6  # unbiased: E[mask * scale] = 1
    # another site, step or seed draws another mask; the same descriptor the same one
    if n >= 1000:
        for other in ((seed, step, site + 1), (seed, step + 1, site), (seed + 1, step, site)):
            assert not np.array_equal(dropout_mask.multiplier(n, p, *other[:2], other[2]), want)
    y2 = torch.empty_like(y)
    capi.dropout(x, y2, capi.Dropout(rng_state(seed, step), site, p))
    assert torch.equal(y, y2)
    # p = 0 is the identity, bit for bit, in place
    z = torch.randn(n, device="cuda").to(dtype)
    z0 = z.clone()
    capi.dropout(z, z, capi.Dropout(rng_state(seed, step), site, 0.0))
    assert torch.equal(z, z0)


def test_fused_sites_use_the_same_mask(capi):
    """QuickGELU (forward and backward), the head's drop_post and the GEMM's RESIDUAL_POS epilogue draw exactly
    the mask of `dfd_dropout` for their site / element numbering."""
    B, D, p, seed, step = 5, 256, 0.5, 99, 3
    st = rng_state(seed, step)
    u = torch.randn(B, 4 * D, device="cuda")
    out, plain = torch.empty_like(u), torch.empty_like(u)
    capi.quickgelu(u, plain)
    capi.quickgelu(u, out, drop=capi.Dropout(st, 4, p))
    m = torch.from_numpy(dropout_mask.multiplier(u.numel(), p, seed, step, 4)).view_as(u).cuda()
    assert torch.equal(out, plain * m)
    du = torch.randn_like(u)
    g_plain, g = torch.empty_like(u), torch.empty_like(u)
    capi.quickgelu(u, g_plain, du=du)
    capi.quickgelu(u, g, du=du, drop=capi.Dropout(st, 4, p))
    assert torch.equal(g, g_plain * m)
    # head: video_feature = drop_post(LayerNorm(x)), logits from the dropped feature
    x = torch.randn(B, D, device="cuda")
    gam, bet, proj = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda"), torch.randn(D, 2, device="cuda") * D ** -0.5
    f0, r0, l0 = torch.empty(B, D, device="cuda"), torch.empty(B, 2, device="cuda"), torch.empty(B, 2, device="cuda")
    f1, r1, l1 = torch.empty_like(f0), torch.empty_like(r0), torch.empty_like(l0)
    capi.head_fwd(x, gam, bet, proj, f0, r0, l0)
    capi.head_fwd(x, gam, bet, proj, f1, r1, l1, drop=capi.Dropout(st, 250, p))
    mh = torch.from_numpy(dropout_mask.multiplier(B * D, p, seed, step, 250)).view(B, D).cuda()
    assert torch.equal(f1, f0 * mh)
    torch.testing.assert_close(r1, (f0 * mh) @ proj, atol=2e-5, rtol=1e-5)
    # GEMM epilogue (both kernels): C = residual + dropout(acc) + pos
    for M, N, K in ((300, 256, 64), (1024 + 333, 512, 128)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        capi.gemm(a, w, c, None, capi.EPI_RESIDUAL_POS, tokens=17, frames_per_clip=1, residual=res, drop=capi.Dropout(st, 1001, p))
        mg = torch.from_numpy(dropout_mask.multiplier(M * N, p, seed, step, 1001)).view(M, N).cuda()
        want = res.double() + (a.double() @ w.double().T) * mg.double()
        err = (c.double() - want).abs()
        assert (err <= 1e-4 + 2 ** -8 * want.abs()).all(), (M, N, K, float(err.max()))


def make(case, precision="fp32", p=0.5):
    from dfd_clip_amd.detector import Detector
    cfg = case["cfg"].clone()
    cfg.dropout = p
    det = Detector(cfg, case["T"], None, precision=precision)
    det.load_state_dict(case["sd"])
    return det.cuda()


@pytest.mark.parametrize("name", ["tiny", "tiny_adapter_nln"])
def test_eval_is_unchanged_bit_for_bit(name):
    case = build_case(name)
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    d0, d5 = make(case, "bf16", 0.0).eval(), make(case, "bf16", 0.5).eval()
    with torch.no_grad():
        a = d0(x, [y], m, single_task=0)[1][0]
        b = d5(x, [y], m, single_task=0)[1][0]
    assert torch.equal(a, b)
    d0.train()  # p = 0 in train(): the dropout-free kernels run, same numbers as eval
    with torch.no_grad():
        assert torch.equal(d0(x, [y], m, train=True, single_task=0)[1][0], a)
    d5.train()
    with torch.no_grad():
        assert not torch.equal(d5(x, [y], m, train=True, single_task=0)[1][0], a)


def step_once(det, case):
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    det.zero_grad(set_to_none=True)
    tl, logits, other = det(x, [y], m, train=True, single_task=0)
    (tl[0].mean() + sum(other.values())).backward()
    return tl[0].detach().clone(), {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name", ["tiny", "tiny_adapter_ln"])
def test_same_seed_same_step(name):
    case = build_case(name)
    a, b, c = make(case).train(), make(case).train(), make(case).train()
    a.seed_dropout(7)
    b.seed_dropout(7)
    c.seed_dropout(8)
    la, ga = step_once(a, case)
    lb, gb = step_once(b, case)
    lc, _ = step_once(c, case)
    assert torch.equal(la, lb) and all(torch.equal(ga[k], gb[k]) for k in ga)
    assert not torch.equal(la, lc), "another seed must draw other masks"
    la2, _ = step_once(a, case)
    assert not torch.equal(la2, la), "the next step must draw other masks"
    lb2, _ = step_once(b, case)
    assert torch.equal(la2, lb2)


@pytest.mark.parametrize("name", ["tiny", "tiny_global", "tiny_adapter_nln", "tiny_adapter_gl", "tiny_adapter_legacy"])
def test_train_step_matches_oracle_given_the_masks(name):
    """fp32 path, dropout 0.5: per-sample losses and every trainable gradient equal autograd of the CPU oracle
    run with the product's masks (regenerated by oracle/dropout_mask.py from seed / step / site)."""
    case = build_case(name)
    det = make(case, "fp32", 0.5).train()
    seed = 20240607
    det.seed_dropout(seed)
    for step in range(2):  # the second step checks that the step counter reaches the kernels
        loss, grads = step_once(det, case)
        w = {k: v.clone() for k, v in case["sd"].items()}
        params = {k: v.requires_grad_(True) for k, v in w.items() if not k.startswith("encoder.")}
        drop = dropout_mask.make_dropper(0.5, seed, step)
        losses, _ = ref_cpu.detector_forward_eval(w, case["x"], [case["y"]], case["m"], single_task=0, drop=drop, **oracle_kwargs(case))
        losses[0].mean().backward()
        np.testing.assert_allclose(loss.cpu().numpy(), losses[0].detach().numpy(), atol=2e-3, rtol=1e-3)
        for k, pr in params.items():
            if pr.grad is None:
                continue
            got = grads[k].cpu()
            scale = max(pr.grad.abs().max().item(), 1e-6)
            assert (got - pr.grad).abs().max().item() <= 2e-3 * scale + 1e-6, (step, k)
        assert any(k.startswith("adapter.") for k in grads) == (case["cfg"].adapter.type != "none")


def test_graph_replay_draws_fresh_masks_and_equals_eager():
    case = build_case("small")
    e, g = make(case, "bf16").train(), make(case, "bf16").train()
    g.static_graphs = True
    for d in (e, g):
        d.seed_dropout(3)
    seen = []
    for step in range(4):
        le, ge = step_once(e, case)
        lg, gg = step_once(g, case)
        assert torch.equal(le, lg), step
        for k in ge:
            assert torch.equal(ge[k], gg[k]), (step, k)
        seen.append(le)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[2], seen[3])


def test_full_size_adapter_dropout_runs_and_regularises():
    """ViT-B/16-sized adapter + decoder in train mode at p = 0.5 (tuned-GEMM epilogue mask path, bf16): finite
    loss and gradients; the dropped forward differs from eval; two seeds differ."""
    from dfd_clip_amd.config import ConfigNode
    from dfd_clip_amd.detector import Detector
    from dfd_clip_amd.weights import random_state_dict, synthetic_clips
    from tests.cases import make_config
    cfg = make_config("ViT-B/16", decode_mode="index", decode_indices=[10, 11], adapter__type="normal", adapter__frozen=0,
                      adapter__struct={"type": "768-x-768-nln", "x": 256})
    cfg.dropout = 0.5
    B, T = 2, 4
    det = Detector(cfg, T, None, precision="bf16")
    det.load_state_dict(random_state_dict(cfg, T, seed=0))
    det = det.cuda().train()
    det.seed_dropout(1)
    x, m, y = synthetic_clips(B, T, 224, seed=5)
    x, m, y = x.cuda(), m.cuda(), y.cuda()
    tl, logits, _ = det(x, [y], m, train=True, single_task=0)
    tl[0].mean().backward()
    assert torch.isfinite(tl[0]).all()
    for n, p in det.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
    det.eval()
    with torch.no_grad():
        ev = det(x, [y], m, single_task=0)[1][0]
    assert not torch.equal(ev, logits[0].detach())
