"""GPU parity of the decoder backward (the path's only gradients: the encoder is frozen) through the
C ABI: kernel by kernel against torch autograd of the CPU oracle, then the whole train-step
contract (reference src/trainer.py:147-177: forward(train=True) -> mean loss -> backward -> SGD
momentum 0.95, wd 0.01) against gradients and parameters produced by the reference itself
(tests/golden)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu
from tests.cases import EXTRA_INPUTS, build_case, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from dfd_clip_amd import capi as c
    c.load_library()
    assert torch.cuda.is_available()
    return c


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32) * scale)


def close(got, want, atol, rtol=0.0, msg=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    err = (got - want).abs()
    assert torch.isfinite(got).all(), msg
    assert (err <= atol + rtol * want.abs()).all(), f"{msg}: max err {err.max().item():.3e}, ref scale {want.abs().max().item():.3e}"


@pytest.mark.parametrize("B,T,P,heads", [(2, 4, 4, 2), (3, 3, 196, 4), (2, 8, 196, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_attention_backward(capi, B, T, P, heads, dtype):
    D, S = heads * 64, T * P
    k = rnd(B, S, heads, 64, seed=1).to(dtype).float()
    v = rnd(B, S, heads, 64, seed=2).to(dtype).float()
    q = rnd(B, 1, heads, 128, seed=3).requires_grad_(True)
    pos = torch.zeros(T, 1, heads, 64, requires_grad=True)  # K/V = exported + pos: d/dpos = sum of dK + dV
    m = torch.ones(B, T, dtype=torch.bool)
    if B > 1:
        m[1, T - max(1, T // 4):] = False
    w = {"p.attn.in_proj.weight": torch.eye(2 * D), "p.attn.in_proj.bias": torch.zeros(2 * D),
         "p.attn.out_proj.weight": torch.eye(D), "p.attn.out_proj.bias": torch.zeros(D)}
    kk = (k.view(B, T, P, heads, 64) + pos).flatten(1, 2).requires_grad_(True)
    vv = (v.view(B, T, P, heads, 64) + pos).flatten(1, 2).requires_grad_(True)
    kk.retain_grad(), vv.retain_grad()
    out = ref_cpu.decoder_attention(q.reshape(B, 1, 2 * D), kk, vv, m.repeat_interleave(P, dim=-1), w, "p.", heads, T)
    dmix = rnd(B, D, seed=4)
    (out.reshape(B, D) * dmix).sum().backward()

    kd, vd = k.to(dtype).reshape(B, S, D).cuda(), v.to(dtype).reshape(B, S, D).cuda()
    qd, md = q.detach().reshape(B, 2 * D).cuda(), m.to(torch.uint8).cuda()
    splits = 2
    ws = torch.empty(capi.decoder_attn_workspace_bytes(B, heads, 64, splits) // 4, device="cuda")
    mix, mix_s, stats = torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda"), torch.empty(B, heads, 2, device="cuda")
    capi.decoder_attn_fwd(qd, kd, vd, md, mix, stats, ws, splits, B, T, P, heads, mix_softmax=mix_s)
    close(mix, out.reshape(B, D), 2e-5, 1e-4, "forward")
    ws2 = torch.empty(capi.decoder_attn_bwd_workspace_bytes(B, T, heads) // 4, device="cuda")
    dq, dpos = torch.empty(B, 2 * D, device="cuda"), torch.empty(T, D, device="cuda")
    dk, dv = torch.empty(B, S, D, device="cuda"), torch.empty(B, S, D, device="cuda")
    capi.decoder_attn_bwd(qd, kd, vd, md, dmix.cuda(), mix_s, stats, dq, dpos, ws2, B, T, P, heads, dk=dk, dv=dv)
    close(dq, q.grad.reshape(B, 2 * D), 2e-5, 2e-4, "dq")
    close(dk, kk.grad.reshape(B, S, D), 2e-6, 2e-4, "dk")
    close(dv, vv.grad.reshape(B, S, D), 2e-6, 2e-4, "dv")
    close(dpos, pos.grad.reshape(T, D), 5e-5, 2e-4, "dpos")
    dq2, dpos2 = torch.empty_like(dq), torch.empty_like(dpos)
    capi.decoder_attn_bwd(qd, kd, vd, md, dmix.cuda(), mix_s, stats, dq2, dpos2, ws2, B, T, P, heads)  # without dK/dV
    assert torch.equal(dq, dq2) and torch.equal(dpos, dpos2)


@pytest.mark.parametrize("modes", ["frame", "temporal", "frame+temporal"])
@pytest.mark.parametrize("B,T,P,heads,dtype", [(2, 4, 4, 2, torch.float32), (3, 5, 196, 4, torch.bfloat16),
                                               (2, 30, 196, 12, torch.bfloat16), (2, 8, 256, 16, torch.float32)])
def test_decoder_attention_modes(capi, modes, B, T, P, heads, dtype):
    """op_mode.attn_mode (reference src/models.py:107-115): grouped softmaxes in the softmax branch,
    forward and backward against autograd of the oracle.  "temporal" also runs with padded frames
    (weight 0); "frame" would be NaN there, in the reference as well, so it gets full clips."""
    D, S = heads * 64, T * P
    attn_mode = tuple(modes.split("+"))
    bits = sum(capi.ATTN_MODE_BITS[a] for a in attn_mode)
    k = rnd(B, S, heads, 64, seed=1).to(dtype).float()
    v = rnd(B, S, heads, 64, seed=2).to(dtype).float()
    q = rnd(B, 1, heads, 128, seed=3).requires_grad_(True)
    pos = torch.zeros(T, 1, heads, 64, requires_grad=True)
    m = torch.ones(B, T, dtype=torch.bool)
    if modes == "temporal":
        m[1, T - max(1, T // 4):] = False
    w = {"p.attn.in_proj.weight": torch.eye(2 * D), "p.attn.in_proj.bias": torch.zeros(2 * D),
         "p.attn.out_proj.weight": torch.eye(D), "p.attn.out_proj.bias": torch.zeros(D)}
    kk = (k.view(B, T, P, heads, 64) + pos).flatten(1, 2).requires_grad_(True)
    vv = (v.view(B, T, P, heads, 64) + pos).flatten(1, 2).requires_grad_(True)
    kk.retain_grad(), vv.retain_grad()
    out = ref_cpu.decoder_attention(q.reshape(B, 1, 2 * D), kk, vv, m.repeat_interleave(P, dim=-1), w, "p.", heads, T,
                                    attn_mode=attn_mode)
    dmix = rnd(B, D, seed=4)
    (out.reshape(B, D) * dmix).sum().backward()

    kd, vd = k.to(dtype).reshape(B, S, D).cuda(), v.to(dtype).reshape(B, S, D).cuda()
    qd, md = q.detach().reshape(B, 2 * D).cuda(), m.to(torch.uint8).cuda()
    new = lambda *shape: torch.empty(*shape, device="cuda")
    sc, aw = new(B, heads, S), new(B, heads, S)
    capi.decoder_attn_modes_fwd(qd, kd, md, bits, sc, aw, B, T, P, heads)
    # each enabled mode contributes weights that sum to T (frame) or P (temporal) per (clip, head)
    want_sum = (T if "frame" in attn_mode else 0) + (P if "temporal" in attn_mode else 0)
    close(aw.sum(-1), torch.full((B, heads), float(want_sum)), 1e-3 * want_sum, msg="weights sum")
    splits = 3
    ws = new(capi.decoder_attn_workspace_bytes(B, heads, 64, splits) // 4)
    mix, stats = new(B, D), new(B, heads, 2)
    capi.decoder_attn_fwd(qd, kd, vd, md, mix, stats, ws, splits, B, T, P, heads, ext_weights=aw)
    scale = out.abs().max().item()
    close(mix, out.reshape(B, D), 2e-5 * max(1.0, scale), 1e-4, "forward")
    dsc = new(B, heads, S)
    capi.decoder_attn_modes_bwd(sc, vd, dmix.cuda(), bits, new(B, heads, S), dsc, B, T, P, heads)
    ws2 = new(capi.decoder_attn_bwd_workspace_bytes(B, T, heads) // 4)
    dq, dpos, dk, dv = new(B, 2 * D), new(T, D), new(B, S, D), new(B, S, D)
    capi.decoder_attn_bwd(qd, kd, vd, md, dmix.cuda(), None, None, dq, dpos, ws2, B, T, P, heads, dk=dk, dv=dv,
                          ext_weights=aw, ext_dscores=dsc)
    gs = max(1.0, q.grad.abs().max().item())
    close(dq, q.grad.reshape(B, 2 * D), 5e-5 * gs, 2e-4, "dq")
    close(dk, kk.grad.reshape(B, S, D), 5e-6 * gs, 2e-4, "dk")
    close(dv, vv.grad.reshape(B, S, D), 5e-6 * gs, 2e-4, "dv")
    close(dpos, pos.grad.reshape(T, D), 1e-4 * gs, 2e-4, "dpos")


@pytest.mark.parametrize("B,N,K", [(2, 8, 128), (16, 1536, 768), (16, 768, 3072), (9, 100, 64)])
def test_linear_backward(capi, B, N, K):
    x, w, dy = rnd(B, K, seed=5), rnd(N, K, seed=6, scale=K ** -0.5), rnd(B, N, seed=7)
    dw, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    capi.linear_rows_bwd_weight(dy.cuda(), x.cuda(), dw, db)
    close(dw, dy.double().T @ x.double(), 2e-5, 1e-5, "dW")
    close(db, dy.double().sum(0), 2e-5, 1e-5, "db")
    wt = torch.empty(K, N, device="cuda")
    capi.transpose(w.cuda(), wt)
    assert torch.equal(wt.cpu(), w.T.contiguous())
    dx = torch.empty(B, K, device="cuda")
    capi.linear_rows(dy.cuda(), wt, None, dx)
    close(dx, dy.double() @ w.double(), 2e-5, 1e-5, "dx")


@pytest.mark.parametrize("B,D", [(2, 128), (16, 768), (5, 1024)])
def test_layernorm_and_gelu_backward(capi, B, D):
    x = (rnd(B, D, seed=8, scale=2.0) + 0.3).requires_grad_(True)
    g = (1 + 0.1 * rnd(D, seed=9)).requires_grad_(True)
    b = (0.1 * rnd(D, seed=10)).requires_grad_(True)
    dy = rnd(B, D, seed=11)
    (F.layer_norm(x, (D,), g, b, 1e-5) * dy).sum().backward()
    dx0 = rnd(B, D, seed=12)
    dx = dx0.clone().cuda()
    dg, dbt, xh = torch.empty(D, device="cuda"), torch.empty(D, device="cuda"), torch.empty(B, D, device="cuda")
    capi.layernorm_bwd(x.detach().cuda(), g.detach().cuda(), dy.cuda(), dx, dg, dbt, xh, accumulate_dx=True)
    close(dx, dx0 + x.grad, 2e-5, 1e-5, "dx (accumulated)")
    close(dg, g.grad, 2e-5, 1e-5, "dgamma")
    close(dbt, b.grad, 2e-5, 1e-5, "dbeta")
    capi.layernorm_bwd(x.detach().cuda(), g.detach().cuda(), dy.cuda(), dx, dg, dbt, xh)
    close(dx, x.grad, 2e-5, 1e-5, "dx")
    u = rnd(B, 4 * D, seed=13, scale=2.0).requires_grad_(True)
    du = rnd(B, 4 * D, seed=14)
    (ref_cpu.quick_gelu(u) * du).sum().backward()
    out = torch.empty(B, 4 * D, device="cuda")
    capi.quickgelu(u.detach().cuda(), out)
    close(out, ref_cpu.quick_gelu(u), 2e-6, 1e-5, "gelu")
    capi.quickgelu(u.detach().cuda(), out, du=du.cuda())
    close(out, u.grad, 2e-6, 1e-5, "gelu'")


@pytest.mark.parametrize("B,D,od", [(2, 128, 2), (16, 768, 2), (3, 256, 10)])
def test_head_backward(capi, B, D, od):
    feat = rnd(B, D, seed=15).requires_grad_(True)
    proj = rnd(D, od, seed=16, scale=D ** -0.5).requires_grad_(True)
    dl, dfe = rnd(B, od, seed=17), rnd(B, D, seed=18)
    z = feat @ proj
    ((ref_cpu.normalise_logits(z) * dl).sum() + (feat * dfe).sum()).backward()
    dz, df, dp = torch.empty(B, od, device="cuda"), torch.empty(B, D, device="cuda"), torch.empty(D, od, device="cuda")
    capi.head_bwd(z.detach().cuda(), dl.cuda(), proj.detach().cuda(), feat.detach().cuda(), dfe.cuda(), dz, df, dp)
    close(df, feat.grad, 2e-5, 1e-4, "dfeat")
    close(dp, proj.grad, 2e-5, 1e-4, "dproj")


def make_detector(case, precision):
    from dfd_clip_amd.detector import Detector
    det = Detector(case["cfg"], case["T"], None, precision=precision)
    det.load_state_dict(case["sd"])
    return det.to("cuda")


@pytest.mark.parametrize("name", ["tiny", "tiny_nopos", "tiny_augq", "tiny_adapter_nln", "tiny_adapter_ln", "tiny_adapter_gl", "tiny_adapter_legacy", "small", "small14", "vitb16_cfg1",
                                  "tiny_global", "tiny_attnmode"])
def test_train_step_contract_matches_reference(name):
    """fp32 path: gradients of every decoder parameter after backward(mean loss), then two SGD steps
    on the same batch, against the reference's own autograd / optimizer results."""
    case = build_case(name)
    g = load_golden(name)
    det = make_detector(case, "fp32")
    det.train()
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    opt = det.configure_optimizers(0.01)
    step_losses = []
    for step in range(2):
        opt.zero_grad()
        task_losses, task_logits, other = det(x, [y], m, train=True, single_task=0)
        loss = task_losses[0].mean() + sum(other.values())
        loss.backward()
        if step == 0:
            checked = 0
            for pn, p in det.named_parameters():
                assert (p.grad is None) == pn.startswith("encoder."), pn
                if p.grad is None:
                    continue
                gr = p.grad.detach().float().cpu()
                if "grad0." + pn in g.files:
                    want = torch.from_numpy(g["grad0." + pn])
                    scale = max(want.abs().max().item(), 1e-6)
                    assert (gr - want).abs().max().item() <= 1e-3 * scale + 2e-7, (pn, (gr - want).abs().max().item(), scale)
                else:
                    np.testing.assert_allclose(gr.norm().item(), g["grad0." + pn + ".norm"], rtol=1e-3)
                    np.testing.assert_allclose(gr.flatten()[:64].numpy(), g["grad0." + pn + ".head"], rtol=2e-3,
                                               atol=2e-4 * max(float(g["grad0." + pn + ".norm"]), 1e-6) / gr.numel() ** 0.5)
                checked += 1
            assert checked > 20
        step_losses.append(loss.item())
        opt.step()
    np.testing.assert_allclose(step_losses, g["step_losses"], atol=2e-4)
    for pn, p in det.named_parameters():
        if not p.requires_grad:
            continue
        t = p.detach().float().cpu()
        if "after2." + pn in g.files:
            np.testing.assert_allclose(t.numpy(), g["after2." + pn], atol=2e-5, rtol=0, err_msg=pn)
        else:
            np.testing.assert_allclose(t.flatten()[:64].numpy(), g["after2." + pn + ".head"], atol=2e-5, rtol=0, err_msg=pn)


def test_bf16_train_step_gradients_close_to_reference():
    """bf16 path: same contract; gradients agree with the reference to bf16 accuracy (cosine > 0.999)."""
    case = build_case("vitb16_cfg1")
    g = load_golden("vitb16_cfg1")
    det = make_detector(case, "bf16")
    det.train()
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    task_losses, _, _ = det(x, [y], m, train=True, single_task=0)
    task_losses[0].mean().backward()
    params = dict(det.named_parameters())
    for pn in ("decoder.proj0x2", "decoder.class_embedding", "decoder.ln_post.weight", "decoder.ln_pre.bias",
               "decoder.transformer.resblocks.0.attn.in_proj.bias", "decoder.transformer.resblocks.5.mlp.c_fc.bias"):
        got = params[pn].grad.float().cpu().flatten()
        want = torch.from_numpy(g["grad0." + pn]).flatten()
        cos = F.cosine_similarity(got, want, dim=0).item()
        print(f"{pn}: cosine {cos:.6f}")
        assert cos > 0.999, (pn, cos)
    pos = params["decoder.positional_embedding"].grad.float().cpu()
    np.testing.assert_allclose(pos.norm().item(), g["grad0.decoder.positional_embedding.norm"], rtol=2e-2)


@pytest.mark.parametrize("R,Ma,Nb", [(40, 8, 12), (777, 128, 32), (5000, 256, 768), (94080, 768, 256)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_at_b(capi, R, Ma, Nb, dtype):
    """Weight-gradient primitive C = A^T B over R rows, deterministic split-K."""
    a, b = rnd(R, Ma, seed=40), rnd(R, Nb, seed=41)
    ar, br = (a, b) if dtype == torch.float32 else (a.bfloat16().float(), b.bfloat16().float())
    want = ar.double().T @ br.double()
    ws = torch.empty(capi.gemm_at_b_workspace_bytes(R, Ma, Nb, dtype) // 4 + 64, device="cuda")
    c = torch.empty(Ma, Nb, device="cuda")
    capi.gemm_at_b(a.to(dtype).cuda(), b.to(dtype).cuda(), c, ws)
    close(c, want, 1e-3 * R ** 0.5, 1e-4, "A^T B")
    c2 = torch.empty_like(c)
    capi.gemm_at_b(a.to(dtype).cuda(), b.to(dtype).cuda(), c2, ws)
    assert torch.equal(c, c2)


@pytest.mark.parametrize("joint", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adapter_norm_gelu_backward(capi, joint, dtype):
    """Adapter middle stage, forward and backward, at the shipped width (x = 256: the register-resident joint
    kernels and the one-wave-per-row kernels) for the three modes: 0 GELU(LN_row), 1 GELU(LN_joint), 2 LN_row(GELU)."""
    frames, P, x = 5, 196, 256
    for mode in ((1,) if joint else (0, 2)):
        a = rnd(frames, P, x, seed=50, scale=1.5).to(dtype).float().requires_grad_(True)
        shape = (P, x) if joint else (x,)
        w = (1 + 0.1 * rnd(*shape, seed=51)).requires_grad_(True)
        b = (0.1 * rnd(*shape, seed=52)).requires_grad_(True)
        dy = rnd(frames, P, x, seed=53).to(dtype).float()
        y = F.layer_norm(F.gelu(a), shape, w, b, 1e-5) if mode == 2 else F.gelu(F.layer_norm(a, shape, w, b, 1e-5))
        (y * dy).sum().backward()
        ad, dyd = a.detach().to(dtype).cuda(), dy.to(dtype).cuda()
        yd = torch.empty_like(ad)
        capi.adapter_norm_gelu(ad, yd, w.detach().cuda(), b.detach().cuda(), frames, P, x, mode)
        close(yd, y, 2e-5 if dtype == torch.float32 else 2e-2, 1e-3 if dtype == torch.float32 else 2e-2, f"forward mode {mode}")
        da = torch.empty_like(ad)
        dw, db = torch.empty(*shape, device="cuda"), torch.empty(*shape, device="cuda")
        ws = torch.empty(capi.adapter_norm_gelu_bwd_workspace_bytes(frames, P, x, mode) // 4 + 4, device="cuda")
        capi.adapter_norm_gelu_bwd(ad, dyd, da, w.detach().cuda(), b.detach().cuda(), dw, db, ws, frames, P, x, mode)
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        close(da, a.grad, tol, 1e-3 if dtype == torch.float32 else 2e-2, f"da mode {mode}")
        close(dw, w.grad, 1e-3 if dtype == torch.float32 else 5e-2, 1e-3, f"dweight mode {mode}")
        close(db, b.grad, 1e-3 if dtype == torch.float32 else 5e-2, 1e-3, f"dbias mode {mode}")


@pytest.mark.parametrize("name", ["tiny_ema", "tiny_rank", "tiny_pmask"])
def test_training_extras_match_reference(name):
    """ema_frame (models.py:572-578), temporal ranking loss (:684-704) and the random patch mask (:511-544, same
    numpy seed as the reference run): eval logits, train losses, auxiliary losses and gradients vs the reference."""
    case = build_case(name)
    g = load_golden(name)
    det = make_detector(case, "fp32")
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    det.eval()
    with torch.no_grad():
        losses, logits = det(x, [y], m, single_task=0)
    np.testing.assert_allclose(logits[0].cpu().numpy(), g["logits"], atol=1e-4)
    np.testing.assert_allclose(losses[0].cpu().numpy(), g["losses"], atol=1e-4)
    det.train()
    speed = torch.tensor(EXTRA_INPUTS["speed"], device="cuda")
    np.random.seed(EXTRA_INPUTS["np_seed"])
    tl, tz, other = det(x, [y], m, EXTRA_INPUTS["comp"], speed, train=True, single_task=0)
    np.testing.assert_allclose(tl[0].detach().cpu().numpy(), g["train_task_loss"], atol=1e-4)
    for k_, v_ in other.items():
        np.testing.assert_allclose(v_.item(), g["other." + k_], atol=1e-5, err_msg=k_)
    assert {("other." + k_) for k_ in other} == {f for f in g.files if f.startswith("other.")}
    (tl[0].mean() + sum(other.values())).backward()
    checked = 0
    for pn, p in det.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.detach().float().cpu()
        if "grad0." + pn in g.files:
            want = torch.from_numpy(g["grad0." + pn])
            scale = max(want.abs().max().item(), 1e-6)
            assert (gr - want).abs().max().item() <= 1e-3 * scale + 2e-7, (pn, (gr - want).abs().max().item(), scale)
        else:
            np.testing.assert_allclose(gr.norm().item(), g["grad0." + pn + ".norm"], rtol=1e-3)
        checked += 1
    assert checked > 20


@pytest.mark.parametrize("name", ["small", "tiny_adapter_nln", "tiny_adapter_gl", "tiny_adapter_legacy"])
def test_static_graphs_match_eager_training(name):
    """Opt-in HIP-graph replay of the decoder's — and, with a trainable CompInvAdapter, the adapter's — forward / backward
    kernels (`Detector.static_graphs`): losses, gradients and SGD steps equal the eager path bit for bit, and later
    steps see new inputs."""
    import copy
    case = build_case(name)
    det_e = make_detector(case, "bf16")
    det_g = copy.deepcopy(det_e)
    det_g.static_graphs = True
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    opt_e, opt_g = det_e.configure_optimizers(0.01), det_g.configure_optimizers(0.01)
    for step in range(4):
        xs = x if step % 2 == 0 else x.flip(0)  # new data every step: the graphs must read the current inputs
        ys = y if step % 2 == 0 else y.flip(0)
        ms = m if step % 2 == 0 else m.flip(0)
        out = []
        for det, opt in ((det_e, opt_e), (det_g, opt_g)):
            det.train()
            opt.zero_grad(set_to_none=True)
            losses, logits, other = det(xs, [ys], ms, train=True, single_task=0)
            loss = losses[0].mean() + sum(other.values())
            loss.backward()
            out.append((losses[0].detach().clone(), logits[0].detach().clone(),
                        {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}))
            opt.step()
        (le, ge_logits, ge), (lg, gg_logits, gg) = out
        assert torch.equal(le, lg) and torch.equal(ge_logits, gg_logits), f"step {step}: forward differs"
        assert ge.keys() == gg.keys()
        for n in ge:
            assert torch.equal(ge[n], gg[n]), f"step {step}: gradient of {n} differs"
    for (n, pe), (_, pg) in zip(det_e.named_parameters(), det_g.named_parameters()):
        assert torch.equal(pe, pg), n
    if det_g.adapter is not None:
        from dfd_clip_amd import adapter as amod
        assert len(amod._GRAPHS.get(det_g.adapter) or {}) == 1 and not det_g.adapter._graphs_failed, "the adapter must have replayed ONE graph"
    # eval / no_grad calls keep working (eager) on the graphed model
    det_g.eval()
    with torch.no_grad():
        a = det_g(x, [y], m, single_task=0)[1][0]
        b = det_e.eval()(x, [y], m, single_task=0)[1][0]
    assert torch.equal(a, b)


def test_graphs_survive_an_epoch_boundary_and_an_eval_pass():
    """train(full batch) -> train(short last batch) -> eval(other batch sizes) -> train(full batch): the K/V buffers of
    the training signatures stay allocated at their addresses (inference keeps sets of its own), so the second epoch
    replays the graphs captured in the first instead of capturing again or going eager; with more signatures than the
    graph cache holds, the least recently used entry makes room (round 2 stopped capturing for good)."""
    from dfd_clip_amd import decoder as dmod
    case = build_case("small")
    det = make_detector(case, "bf16")
    det.static_graphs = True
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    big = (torch.cat([x, x.flip(0)]), torch.cat([m, m.flip(0)]), torch.cat([y, y.flip(0)]))  # "full" batch: 4 clips
    opt = det.configure_optimizers(0.01)

    def train_on(xs, ms, ys):
        det.train()
        opt.zero_grad(set_to_none=True)
        losses, logits, other = det(xs, [ys], ms, train=True, single_task=0)
        (losses[0].mean() + sum(other.values())).backward()
        opt.step()
        return logits[0].detach().clone()

    def graphs():
        return dmod._GRAPHS.get(det.decoder) or {}

    train_on(*big)
    train_on(x[:1], m[:1], y[:1])  # the epoch's short last batch
    keys_epoch1 = list(graphs().keys())
    assert len(keys_epoch1) == 2 and not det.decoder._graphs_failed
    det.eval()
    with torch.no_grad():
        for n in (3, 1, 4):
            det(big[0][:n], [big[2][:n]], big[1][:n], single_task=0)
    a = train_on(*big)
    train_on(x[:1], m[:1], y[:1])
    assert list(graphs().keys())[-2:] == keys_epoch1 or set(graphs().keys()) == set(keys_epoch1), "epoch 2 must reuse epoch 1's graphs"
    assert len(graphs()) == 2
    # same step on a fresh eager model from the same parameters: graphs replayed on re-found buffers compute the same
    det.decoder.max_graphs = 2
    for n in (2, 3):  # two more signatures: the cache evicts instead of refusing
        train_on(big[0][:n], big[1][:n], big[2][:n])
    assert len(graphs()) == 2 and not det.decoder._graphs_failed
    assert all(k not in graphs() for k in keys_epoch1)
    b = train_on(*big)  # captured again after its eviction
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


@pytest.mark.parametrize("graphs,name", [(False, "small"), (True, "small"), (False, "tiny_adapter_nln"), (False, "tiny_adapter_gl")])
def test_pipelined_encoder_matches_eager_training(graphs, name):
    """`Detector.pipeline_encoder` (frozen encoder on its own stream, two alternating K/V export sets, step N+1's
    encoder pass overlapping step N's backward / optimizer): six steps with changing batches give the same losses,
    gradients and parameters as the plain single-stream path, bit for bit."""
    import copy
    case = build_case(name)
    det_e = make_detector(case, "bf16")
    det_p = copy.deepcopy(det_e)
    det_p.pipeline_encoder, det_p.inputs_ready, det_p.static_graphs = True, True, graphs
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    batches = [(x, m, y), (x.flip(0), m.flip(0), y.flip(0)), (x.roll(1, 1), m, y), (x * 0.5, m, y.flip(0)), (x, m, y), (x.flip(1), m, y)]
    batches = [tuple(t.contiguous() for t in b) for b in batches]
    torch.cuda.synchronize()  # inputs_ready: every batch is complete in device memory before the loop
    opt_e, opt_p = det_e.configure_optimizers(0.01), det_p.configure_optimizers(0.01)
    for step, (xs, ms, ys) in enumerate(batches):
        res = []
        for det, opt in ((det_e, opt_e), (det_p, opt_p)):
            det.train()
            opt.zero_grad(set_to_none=True)
            losses, logits, other = det(xs, [ys], ms, train=True, single_task=0)
            (losses[0].mean() + sum(other.values())).backward()
            res.append((losses[0].detach().clone(), {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}))
            opt.step()
        assert torch.equal(res[0][0], res[1][0]), f"step {step}: losses differ"
        for n in res[0][1]:
            assert torch.equal(res[0][1][n], res[1][1][n]), f"step {step}: gradient of {n} differs"
    for (n, pe), (_, pp) in zip(det_e.named_parameters(), det_p.named_parameters()):
        assert torch.equal(pe, pp), n
    det_p.eval()
    with torch.no_grad():
        assert torch.equal(det_p(x, [y], m, single_task=0)[1][0], det_e.eval()(x, [y], m, single_task=0)[1][0])


def test_pipelined_encoder_pass_replays_as_one_graph():
    """static_graphs + pipeline_encoder + K/V in place on RECURRING input buffers (two batches alternating, as a loader
    with recycled pinned/device buffers hands them out): from the second sighting of an (input, K/V set) pair the
    frozen encoder's pass is one HIP graph launch.  Eight steps give the plain path's losses, gradients and parameters
    bit for bit, and the graphs were really captured and replayed."""
    import copy
    case = build_case("small")
    det_e = make_detector(case, "bf16")
    det_p = copy.deepcopy(det_e)
    det_p.pipeline_encoder, det_p.inputs_ready, det_p.static_graphs = True, True, True
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    a, b = (x.contiguous(), m, y), ((x * 0.5).flip(0).contiguous(), m.flip(0).contiguous(), y.flip(0).contiguous())
    torch.cuda.synchronize()
    opt_e, opt_p = det_e.configure_optimizers(0.01), det_p.configure_optimizers(0.01)
    for step in range(8):
        xs, ms, ys = (a, b)[step % 2] if step < 6 else a  # ... and the same batch three times in a row at the end
        res = []
        for det, opt in ((det_e, opt_e), (det_p, opt_p)):
            det.train()
            det.zero_grad()
            losses, logits, other = det(xs, [ys], ms, train=True, single_task=0)
            (losses[0].mean() + sum(other.values())).backward()
            res.append((losses[0].detach().clone(), {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}))
            opt.step()
        assert torch.equal(res[0][0], res[1][0]), f"step {step}: losses differ"
        for n in res[0][1]:
            assert torch.equal(res[0][1][n], res[1][1][n]), f"step {step}: gradient of {n} differs"
    assert det_p._enc_graphs_failed is None and len(det_p._enc_graphs) >= 2, (det_p._enc_graphs_failed, len(det_p._enc_graphs))
    assert not det_e._enc_graphs
    for (n, pe), (_, pp) in zip(det_e.named_parameters(), det_p.named_parameters()):
        assert torch.equal(pe, pp), n
    # new encoder weights: the captured passes are stale (they carry the old weight buffers as addresses) and must not be
    # replayed — the next steps follow the new weights, again bit for bit with the plain path
    sd = {k: (v * 1.01 if k.endswith("mlp.c_fc.weight") else v) for k, v in det_e.encoder.state_dict().items()}
    for det in (det_e, det_p):
        det.encoder.load_state_dict(sd)
    for step in range(3):
        res = []
        for det, opt in ((det_e, opt_e), (det_p, opt_p)):
            det.zero_grad()
            losses, _, other = det(a[0], [a[2]], a[1], train=True, single_task=0)
            (losses[0].mean() + sum(other.values())).backward()
            res.append(losses[0].detach().clone())
            opt.step()
        assert torch.equal(res[0], res[1]), f"after new encoder weights, step {step}: losses differ"


@pytest.mark.parametrize("graphs", [False, True])
def test_pipelined_encoder_with_changing_batch_size(graphs):
    """The pipelined path keeps persistent K/V buffer sets per batch shape; a shape change (the last batch of an epoch)
    reallocates them while earlier steps may still be in flight.  Steps of 2, 1, 2, 2, 1 clips without host
    synchronisation give the plain path's losses and parameters, bit for bit."""
    import copy
    case = build_case("small")
    det_e = make_detector(case, "bf16")
    det_p = copy.deepcopy(det_e)
    det_p.pipeline_encoder, det_p.inputs_ready, det_p.static_graphs = True, True, graphs
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    batches = [(x, m, y), (x[:1], m[:1], y[:1]), (x.flip(0), m.flip(0), y.flip(0)), (x * 0.5, m, y), (x[1:], m[1:], y[1:])]
    batches = [tuple(t.contiguous() for t in b) for b in batches]
    torch.cuda.synchronize()

    def run(det):
        opt = det.configure_optimizers(0.01)
        det.train()
        out = []
        for xs, ms, ys in batches:
            opt.zero_grad(set_to_none=True)
            losses, _, other = det(xs, [ys], ms, train=True, single_task=0)
            (losses[0].mean() + sum(other.values())).backward()
            opt.step()
            out.append(losses[0].detach())
        return out

    lp = run(det_p)
    torch.cuda.synchronize()
    le = run(det_e)
    for a, b in zip(lp, le):
        assert torch.equal(a, b)
    for (n, pe), (_, pp) in zip(det_e.named_parameters(), det_p.named_parameters()):
        assert torch.equal(pe, pp), n


@pytest.mark.parametrize("kv_in_place", [True, False])
def test_pipelined_encoder_full_size_trainable_positional_embedding(kv_in_place):
    """kv_in_place: the default hand-over (the decoder reads K/V and the live positional embedding on the caller's
    stream) and the export path (the projection's epilogue adds a snapshot of it on the encoder stream).
    ADVICE r1 (high): the pipelined encoder stream must not read the TRAINABLE temporal positional embedding
    while the previous step's optimizer is writing it.  Full size (ViT-B/16, 16 clips x 30 frames, bf16), large
    learning rate on the positional embedding, six back-to-back steps with NO host synchronisation (the host runs
    several steps ahead of the device, as in bench.py), graphs + pipelining + inputs_ready — against the plain
    single-stream path run afterwards: losses and every parameter bit-identical."""
    import copy
    from dfd_clip_amd.detector import Detector
    from dfd_clip_amd.weights import random_state_dict
    from tests.cases import make_config
    cfg = make_config("ViT-B/16", decode_mode="index", decode_indices=[6, 7, 8, 9, 10, 11])
    B, T = 16, 30
    det_e = Detector(cfg, T, None, precision="bf16")
    det_e.load_state_dict(random_state_dict(cfg, T, seed=0))
    det_e = det_e.cuda().train()
    det_e.kv_in_place = kv_in_place
    det_p = copy.deepcopy(det_e)
    det_p.pipeline_encoder, det_p.inputs_ready, det_p.static_graphs = True, True, True
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, T, 3, 224, 224, device="cuda", generator=g)
    m = torch.ones(B, T, dtype=torch.bool, device="cuda")
    m[5, 25:] = False
    y = torch.arange(B, device="cuda") % 2
    batches = [(x, m, y), (x.flip(0).contiguous(), m.flip(0).contiguous(), y.flip(0).contiguous()), (x * 0.5, m, y)] * 2
    torch.cuda.synchronize()  # inputs_ready: every batch is complete in device memory before the loops

    def run(det):
        pos = det.decoder.positional_embedding
        rest = [p for p in det.parameters() if p.requires_grad and p is not pos]
        opt = torch.optim.SGD([{"params": [pos], "lr": 2.0}, {"params": rest, "lr": 0.01}], momentum=0.95, weight_decay=0.01)
        losses = []
        for xs, ms, ys in batches:
            opt.zero_grad(set_to_none=True)
            tl, _, other = det(xs, [ys], ms, train=True, single_task=0)
            (tl[0].mean() + sum(other.values())).backward()
            opt.step()
            losses.append(tl[0].detach())
        return losses

    lp = run(det_p)
    torch.cuda.synchronize()
    le = run(det_e)
    torch.cuda.synchronize()
    moved = (det_e.decoder.positional_embedding - random_state_dict(cfg, T, seed=0)["decoder.positional_embedding"].cuda()).abs().max().item()
    assert moved > 1e-2, "the positional embedding must really move in this test"
    for step, (a, b) in enumerate(zip(lp, le)):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), f"step {step}: losses differ (max {float((a - b).abs().max()):.3e})"
    for (n, pe), (_, pp) in zip(det_e.named_parameters(), det_p.named_parameters()):
        assert torch.equal(pe, pp), n
