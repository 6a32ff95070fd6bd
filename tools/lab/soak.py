"""Lab: 300 pipelined + graphed train steps at the headline shape; checks that the loss stays finite and decreases
on a fixed batch, and that device memory does not grow after the warm-up."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dfd_clip_amd.config import default_detector_config  # noqa: E402
from dfd_clip_amd.detector import Detector  # noqa: E402
from dfd_clip_amd.weights import random_state_dict  # noqa: E402

cfg = default_detector_config()
cfg.architecture = "ViT-B/16"
cfg.decode_mode = "index"
cfg.decode_indices = [6, 7, 8, 9, 10, 11]
cfg.out_dim = [2]
cfg.losses = ["auc_roc"]
T, B = 30, 16
det = Detector(cfg, T, None, precision="bf16")
det.load_state_dict(random_state_dict(cfg, T, seed=0))
det = det.cuda().train()
det.static_graphs = det.pipeline_encoder = det.inputs_ready = True
x = torch.randn(B, T, 3, 224, 224, device="cuda")
m = torch.ones(B, T, dtype=torch.bool, device="cuda")
y = torch.arange(B, device="cuda") % 2
opt = det.configure_optimizers(0.003)
losses, mem = [], []
for step in range(300):
    opt.zero_grad(set_to_none=True)
    l, _, other = det(x, [y], m, train=True, single_task=0)
    loss = l[0].mean() + sum(other.values())
    loss.backward()
    opt.step()
    if step % 50 == 0 or step == 299:
        losses.append(loss.item())
        mem.append(torch.cuda.memory_allocated() / 2 ** 20)
print("loss", [round(v, 4) for v in losses])
print("MiB ", [round(v) for v in mem])
assert all(v == v and abs(v) < 1e4 for v in losses), "loss not finite"
assert losses[-1] < losses[0], "loss did not decrease on a fixed batch"
assert mem[-1] <= mem[1] * 1.01 + 1, "device memory keeps growing"
print("soak ok")
