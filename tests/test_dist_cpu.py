"""CPU checks of the multi-GPU host logic with world_size-2 gloo process groups: gradient averaging
equals the single-process gradient of the concatenated batch, metric gathering keeps rank order
and drops padded tails, parameter broadcast, scheduler stepping convention, AUROC helper."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from dfd_clip_amd import dist as ddist
from dfd_clip_amd import harness


class StubDetector(nn.Module):
    """Test stand-in with the Detector call contract (CPU): frozen 'encoder', trainable 'decoder'."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.encoder = nn.Linear(12, 8)
        for p in self.encoder.parameters():
            p.requires_grad = False
        self.decoder = nn.Linear(8, 2)
        self.unused = nn.Parameter(torch.zeros(3))  # never touched by forward: find_unused_parameters case
        self.out_dim = [2]

    def predict(self, x, m, *a, **k):
        with torch.no_grad():
            f = self.encoder(x.flatten(2).mean(dim=1))
        z = self.decoder(f)
        return [5 * z / (z.norm(dim=-1, keepdim=True) + 1e-10)], {}

    def forward(self, x, y, m, comp=None, speed=None, train=False, single_task=None):
        logits, _ = self.predict(x, m)
        losses = [torch.nn.functional.cross_entropy(logits[0], y[0], reduction="none")]
        return (losses, logits, {}) if train else (losses, logits)


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn, port):
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fn, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return dict(ret)


def _data(n=8):
    g = torch.Generator().manual_seed(3)
    return torch.randn(n, 4, 12, generator=g), torch.arange(n) % 2, torch.ones(n, 4, dtype=torch.bool)


def _train_two_ranks(rank, world):
    x, y, m = _data()
    model = StubDetector()
    with torch.no_grad():  # desynchronise, then broadcast must repair it
        model.decoder.weight.add_(rank * 1.0)
    ddist.broadcast_parameters(model)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1, momentum=0.95, weight_decay=0.01)
    sched = harness.make_one_cycle(opt, 0.1, max_steps=4)
    sl = slice(rank * 4, rank * 4 + 4)
    for _ in range(2):
        harness.train_step(model, opt, [(x[sl], y[sl], m[sl], None, None, 0)], sched)
    return model.decoder.weight.detach().clone().numpy(), opt.param_groups[0]["lr"], sched.last_epoch


def test_gradient_allreduce_matches_single_process_large_batch():
    out = run2(_train_two_ranks, 29611)
    np.testing.assert_array_equal(out[0][0], out[1][0])  # ranks stay in lock step
    x, y, m = _data()
    model = StubDetector()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1, momentum=0.95, weight_decay=0.01)
    sched = harness.make_one_cycle(opt, 0.1, max_steps=4, num_processes=2)
    for _ in range(2):
        # mean over 8 samples == average of the two ranks' means over 4
        harness.train_step(model, opt, [(x, y, m, None, None, 0)], None)
        harness.step_scheduler(sched, 2)
    np.testing.assert_allclose(out[0][0], model.decoder.weight.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert out[0][1] == pytest.approx(opt.param_groups[0]["lr"]) and out[0][2] == 4  # 2 steps x 2 processes


def _gather_two_ranks(rank, world):
    # rank 0 holds samples 0..3, rank 1 holds 4..6 plus one padded duplicate
    vals = torch.arange(4, dtype=torch.float32) + 4 * rank
    probs = torch.stack([1 - vals / 10, vals / 10], dim=1)
    a, b = ddist.gather_for_metrics((vals, probs), valid=4 if rank == 0 else 3)
    x, y, m = _data(8)
    model = StubDetector().eval()
    sl = slice(rank * 4, rank * 4 + 4)
    res = harness.evaluate(model, [(x[sl], y[sl], m[sl], 0)])
    return a.numpy(), b.numpy(), res["accuracy"], res["roc_auc"], res["probs"].numpy()


def test_gather_for_metrics_and_distributed_evaluate():
    out = run2(_gather_two_ranks, 29612)
    for r in range(2):
        np.testing.assert_array_equal(out[r][0], np.arange(7, dtype=np.float32))
        assert out[r][1].shape == (7, 2)
    x, y, m = _data(8)
    ref = harness.evaluate(StubDetector().eval(), [(x, y, m, 0)])
    assert out[0][2] == pytest.approx(ref["accuracy"]) and out[0][3] == pytest.approx(ref["roc_auc"])
    np.testing.assert_allclose(out[1][4], ref["probs"].numpy(), rtol=1e-6)


def test_auroc_matches_sklearn_and_reference_dummy_pair():
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(7)
    y = rng.integers(0, 2, 256)
    s = np.round(rng.random(256), 2)  # ties on purpose
    assert harness.binary_auroc(y, s) == pytest.approx(roc_auc_score(y, s), abs=1e-12)
    model = StubDetector().eval()
    g = torch.Generator().manual_seed(5)
    videos = [([torch.randn(4, 12, generator=g) for _ in range(n)], [i % 2] * n, [torch.ones(4, dtype=torch.bool)] * n)
              for i, n in enumerate([3, 0, 5, 2])]
    res = harness.infer_videos(model, videos, batch_size=2, modality="video", device="cpu")
    assert res["labels"].tolist() == [0, 0, 1] and res["probs"].shape == (3, 2)  # empty video skipped, ragged chunks
    want = roc_auc_score(res["labels"].tolist() + [0, 1], res["probs"][:, 1].tolist() + [0.0, 1.0])
    assert res["roc_auc"] == pytest.approx(round(want, 3))
    res_c = harness.infer_videos(model, videos, batch_size=2, modality="clip", device="cpu")
    assert res_c["probs"].shape == (10, 2)
