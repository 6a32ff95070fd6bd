"""`FusedSGD` (dfd-clip_amd/optim.py, csrc/optim.hip): `Detector.configure_optimizers`' SGD (reference src/models.py:740-754;
momentum 0.95, weight decay; lr AND momentum moved every step by OneCycleLR, src/trainer.py:55-60) as one HIP launch that
also keeps the decoder's transposed weight copies current — against torch.optim.SGD on the same tensors."""
import copy

import pytest
import torch

from tests.cases import build_case

pytestmark = pytest.mark.gpu


class _Mirrors:
    """Stand-in for the Decoder's mirror protocol: every 2-D parameter has a transposed copy."""

    def __init__(self):
        self.t, self.written = {}, 0

    def mirror_for(self, p):
        if id(p) not in self.t:
            self.t[id(p)] = p.detach().t().contiguous()
        return self.t[id(p)]

    def current_mirror(self, p):
        return self.t.get(id(p))

    def mirrors_written(self, pairs):
        self.written += len(pairs)


def test_fused_sgd_equals_torch_sgd_step_by_step():
    from dfd_clip_amd.optim import FusedSGD
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(768, 768), (3072, 768), (768,), (30, 1, 12, 64), (1,), (2, 768), (1537, 33), (5,)]
    ref = [torch.nn.Parameter(torch.randn(*s, device="cuda", generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    mirrors = _Mirrors()
    o_ref = torch.optim.SGD(ref, lr=0.01, momentum=0.95, weight_decay=0.01)
    o_mine = FusedSGD(mine, lr=0.01, momentum=0.95, weight_decay=0.01, mirrors=mirrors)
    sched_r = torch.optim.lr_scheduler.OneCycleLR(o_ref, max_lr=0.25, total_steps=6)  # cycles momentum 0.85 <-> 0.95 as well
    sched_m = torch.optim.lr_scheduler.OneCycleLR(o_mine, max_lr=0.25, total_steps=6)
    for step in range(5):
        for i, (a, b) in enumerate(zip(ref, mine)):
            if i == 4 and step < 2:      # a parameter that gets its first gradient late (its first step initialises its velocity)
                a.grad = b.grad = None
                continue
            if i == 7 and step == 3:     # ... and one that misses a step: skipped, velocity kept
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, device="cuda", generator=g)
            a.grad, b.grad = gr.clone(), gr.clone()
        v_before = [p._version for p in mine]
        o_ref.step()
        o_mine.step()
        sched_r.step()
        sched_m.step()
        for i, (a, b) in enumerate(zip(ref, mine)):
            torch.testing.assert_close(b, a, rtol=0, atol=1e-6 * max(1.0, a.abs().max().item()), msg=f"step {step} param {i}")
            if b.grad is not None:
                assert b._version > v_before[i], "caches keyed on the version counter must see the update"
            if b.dim() == 2:
                assert torch.equal(mirrors.t[id(b)], b.detach().t()), f"step {step}: transposed copy of param {i} is stale"
            sa, sb = o_ref.state[a].get("momentum_buffer"), o_mine.state[b].get("momentum_buffer")
            if sa is not None:
                torch.testing.assert_close(sb, sa, rtol=0, atol=1e-6 * max(1.0, sa.abs().max().item()))
    assert mirrors.written > 0
    assert o_mine.param_groups[0]["lr"] == pytest.approx(o_ref.param_groups[0]["lr"])
    # state_dict round trip: a resumed optimizer continues with the same velocities (no first-step re-initialisation)
    sd = copy.deepcopy(o_mine.state_dict())
    resumed_p = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    o_res = FusedSGD(resumed_p, lr=0.01, momentum=0.95, weight_decay=0.01)
    o_res.load_state_dict(sd)
    for a, b, c in zip(ref, mine, resumed_p):
        gr = torch.randn(a.shape, device="cuda", generator=g)
        a.grad, b.grad, c.grad = gr.clone(), gr.clone(), gr.clone()
    o_ref.step(), o_mine.step(), o_res.step()
    for a, b, c in zip(ref, mine, resumed_p):
        assert torch.equal(b, c), "resumed from state_dict"
        torch.testing.assert_close(b, a, rtol=0, atol=1e-6 * max(1.0, a.abs().max().item()))


@pytest.mark.parametrize("graphs", [False, True])
def test_detector_training_with_fused_sgd_keeps_the_decoders_transposes_current(graphs):
    """Through `Detector.configure_optimizers`: the optimizer is the fused one, the decoder launches no transpose after an
    update (its copies are rewritten by the optimizer's launch), and switching to another optimizer mid-run is caught up
    with (eagerly, before a graph replay too) instead of training on stale copies."""
    from dfd_clip_amd import capi
    from dfd_clip_amd.optim import FusedSGD
    from tests.test_hip_detector import make_detector
    case = build_case("small")
    det = make_detector(case, "bf16").train()
    twin = copy.deepcopy(det)
    det.static_graphs = twin.static_graphs = graphs
    x, m, y = case["x"].cuda(), case["m"].cuda(), case["y"].cuda()
    opt = det.configure_optimizers(0.05)
    assert isinstance(opt, FusedSGD)
    opt_t = torch.optim.SGD([p for p in twin.parameters() if p.requires_grad], lr=0.05, momentum=0.95, weight_decay=twin.weight_decay)
    calls = {"n": 0}
    real = capi.transpose

    def counting(src, dst):
        calls["n"] += 1
        return real(src, dst)

    capi.transpose = counting
    try:
        for step in range(4):
            for d, o in ((det, opt), (twin, opt_t)):
                o.zero_grad(set_to_none=True)
                c0 = calls["n"]
                losses, logits, other = d(x, [y], m, train=True, single_task=0)
                (losses[0].mean() + sum(other.values())).backward()
                o.step()
                if d is det:
                    mine = calls["n"] - c0  # transposes launched by this model's forward / backward / optimizer
            if step == 0:
                assert mine > 0  # the copies are created once
            else:
                assert mine == 0, "the fused optimizer keeps the copies current: no transpose after step 0"
        for (n, a), (_, b) in zip(det.named_parameters(), twin.named_parameters()):
            if a.requires_grad:
                torch.testing.assert_close(a, b, rtol=0, atol=2e-5 * max(1.0, b.abs().max().item()), msg=n)
        # hand the model to a plain torch optimizer: the decoder must notice that its copies went stale
        plain = torch.optim.SGD([p for p in det.parameters() if p.requires_grad], lr=0.05)
        for d, o in ((det, plain), (twin, torch.optim.SGD([p for p in twin.parameters() if p.requires_grad], lr=0.05))):
            o.zero_grad(set_to_none=True)
            losses, logits, other = d(x, [y], m, train=True, single_task=0)
            (losses[0].mean() + sum(other.values())).backward()
            o.step()
        det.eval(), twin.eval()
        with torch.no_grad():
            la = det(x, [y], m, single_task=0)[1][0]
            lb = twin(x, [y], m, single_task=0)[1][0]
        assert (la - lb).abs().max().item() < 1e-3
    finally:
        capi.transpose = real
