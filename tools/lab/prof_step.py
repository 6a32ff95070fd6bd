#!/usr/bin/env python3
"""Lab: cProfile of a few train steps (host side) of the headline configuration."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--no-secondary", "--no-cpu-baseline"]
import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
det, cfg, sd, layers = bench.build_model(args, dev)
x = torch.randn(16, 30, 3, 224, 224, device=dev)
m = torch.ones(16, 30, dtype=torch.bool, device=dev)
y = torch.arange(16, device=dev) % 2
det.static_graphs, det.pipeline_encoder, det.inputs_ready = True, True, True
det.train()
opt = det.configure_optimizers(0.01 / 25)


def step():
    det.zero_grad(set_to_none=True)
    losses, _, other = det(x, [y], m, train=True, single_task=0)
    (losses[0].mean() + sum(other.values())).backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
