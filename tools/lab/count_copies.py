#!/usr/bin/env python3
"""Lab: where do the device-to-device copies of a train step come from?  One eager (no HIP graph) step under the
torch profiler; aten::copy_ / clone calls grouped by shape and by the innermost frame of this repo."""
import os
import sys
import collections

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from dfd_clip_amd.config import default_detector_config
from dfd_clip_amd.detector import Detector
from dfd_clip_amd.weights import random_state_dict

cfg = default_detector_config(); cfg.architecture = "ViT-B/16"; cfg.decode_mode = "index"; cfg.decode_indices = [6, 7, 8, 9, 10, 11]
cfg.out_dim = [2]; cfg.losses = ["auc_roc"]
T, B = 30, 16
det = Detector(cfg, T, None, precision="bf16"); det.load_state_dict(random_state_dict(cfg, T, seed=0)); det = det.cuda().train()
det.static_graphs = os.environ.get("G", "0") == "1"; det.pipeline_encoder = True; det.inputs_ready = True
x = torch.randn(B, T, 3, 224, 224, device="cuda"); m = torch.ones(B, T, dtype=torch.bool, device="cuda"); y = torch.arange(B, device="cuda") % 2
opt = det.configure_optimizers(0.001)


def step():
    det.zero_grad(set_to_none=True)
    losses, _, other = det(x, [y], m, train=True, single_task=0)
    (losses[0].mean() + sum(other.values())).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy"):
        where = "?"
        for fr in (ev.stack or []):
            if "dfd_clip_amd" in fr or "bench" in fr or "count_copies" in fr:
                where = fr.split("/")[-1]
                break
        cnt[(ev.name, str(ev.input_shapes)[:60], where)] += 1
tot = 0
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:16s} {shp:62s} {where}")
    if name == "aten::copy_":
        tot += n
print("aten::copy_ calls in one step:", tot)
