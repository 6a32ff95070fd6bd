#!/usr/bin/env python3
"""Lab: cProfile of a few pipelined train steps with a CompInvAdapter (where does the host wait?)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dfd_clip_amd.config import ConfigNode, default_detector_config  # noqa: E402
from dfd_clip_amd.detector import Detector  # noqa: E402
from dfd_clip_amd.weights import random_state_dict  # noqa: E402

cfg = default_detector_config()
cfg.architecture = "ViT-B/16"; cfg.decode_mode = "index"; cfg.decode_indices = [6, 7, 8, 9, 10, 11]; cfg.out_dim = [2]; cfg.losses = ["auc_roc"]
cfg.adapter = ConfigNode({"type": "normal", "frozen": 0, "struct": {"type": "768-x-768-nln", "x": 256}})
T, B = 30, 16
det = Detector(cfg, T, None, precision="bf16"); det.load_state_dict(random_state_dict(cfg, T, seed=0)); det = det.cuda().train()
det.static_graphs = det.pipeline_encoder = det.inputs_ready = True
x = torch.randn(B, T, 3, 224, 224, device="cuda"); m = torch.ones(B, T, dtype=torch.bool, device="cuda"); y = torch.arange(B, device="cuda") % 2
opt = det.configure_optimizers(0.001)


def step():
    opt.zero_grad(set_to_none=True)
    losses, _, other = det(x, [y], m, train=True, single_task=0)
    (losses[0].mean() + sum(other.values())).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
