#!/usr/bin/env python3
"""Lab: ViT-L/14 fp8 vs bf16 on 256 synthetic 2-frame clips — AUROC / Spearman / logit drift by calibration margin and by
which projections run on e4m3 operands."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.cases import build_case  # noqa: E402
from tests.test_hip_fp8 import _auroc, _make, _spearman  # noqa: E402

case = build_case("vitl14")
T, res, n = case["T"], case["res"], 256
rng = np.random.default_rng(4321)
x = torch.from_numpy(rng.standard_normal((n, T, 3, res, res), dtype=np.float32))
m = torch.ones(n, T, dtype=torch.bool)
m[3::7, T - 1:] = False
y = np.random.default_rng(7).integers(0, 2, n)


def run(det):
    p, lg = [], []
    with torch.no_grad():
        for i in range(0, n, 32):
            logits, _ = det.predict(x[i:i + 32].cuda(), m[i:i + 32].cuda())
            p.append(logits[0].softmax(dim=-1)[:, 1].cpu())
            lg.append(logits[0].float().cpu())
    return torch.cat(p).numpy(), torch.cat(lg)


p16, l16 = run(_make(case, "bf16"))
for margin in (1.0, 2.0, 4.0):
    for ncal in (16, 64):
        det = _make(case, "fp8")
        det.calibrate_fp8(x[:ncal].cuda(), margin=margin)
        p8, l8 = run(det)
        print(f"margin {margin} calib {ncal}: AUROC {_auroc(y, p16):.4f} / {_auroc(y, p8):.4f}  spearman {_spearman(p16, p8):.5f}  "
              f"max|dlogit| {(l16 - l8).abs().max().item():.3f} mean|dlogit| {(l16 - l8).abs().mean().item():.3f}", flush=True)
        del det
