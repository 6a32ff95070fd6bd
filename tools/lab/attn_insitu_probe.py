#!/usr/bin/env python3
"""Runs ViT-L/14 B8xT30 predict() with every encoder-attention call shadowed by the per-item kernel (the same frames in
launches of < 512 items) and reports the calls whose outputs differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402
from dfd_clip_amd.detector import Detector  # noqa: E402
from dfd_clip_amd.weights import random_state_dict  # noqa: E402
from tests.cases import make_config  # noqa: E402

cfg = make_config("ViT-L/14", decode_mode="stride", decode_stride=2)
B, T = 8, 30
det = Detector(cfg, T, None, precision="bf16")
det.load_state_dict(random_state_dict(cfg, T, seed=0))
det = det.cuda().eval()
det.pipeline = False if hasattr(det, "pipeline") else None
g = torch.Generator(device="cuda").manual_seed(11)
x = torch.randn(B, T, 3, 224, 224, device="cuda", generator=g)
m = torch.ones(B, T, dtype=torch.bool, device="cuda")

real = capi.attention_fwd
calls = []


def shadow(qkv, out, n, tokens, heads, *a, **k):
    r = real(qkv, out, n, tokens, heads, *a, **k)
    if n * heads >= 512 and tokens == 257:
        torch.cuda.synchronize()
        ref = torch.empty_like(out)
        step = max(1, 511 // heads)
        for f0 in range(0, n, step):
            f1 = min(n, f0 + step)
            real(qkv[f0 * tokens:f1 * tokens], ref[f0 * tokens:f1 * tokens], f1 - f0, tokens, heads, *a, **k)
        torch.cuda.synchronize()
        bad = (out != ref)
        calls.append((len(calls), n, int(bad.sum()), tuple(qkv.shape), qkv.stride(), tuple(out.shape), out.stride(),
                      float(qkv.float().abs().max())))
        if bad.any() and sum(1 for c in calls if c[2]) <= 3:
            idx = bad.nonzero()
            rows = sorted(set((idx[:, 0] % tokens).tolist()))
            print("call", len(calls) - 1, "n", n, "bad", int(bad.sum()), "token rows", rows[:20], "cols", sorted(set((idx[:, 1] % 64).tolist()))[:16],
                  "heads", sorted(set((idx[:, 1] // 64).tolist())), "frames", sorted(set((idx[:, 0] // tokens).tolist()))[:10])
            i0 = idx[0]
            print("   first:", i0.tolist(), float(out[i0[0], i0[1]]), float(ref[i0[0], i0[1]]))
    return r


capi.attention_fwd = shadow
import dfd_clip_amd.encoder as enc  # noqa: E402
if hasattr(enc, "capi"):
    enc.capi.attention_fwd = shadow
with torch.no_grad():
    det.predict(x, m)
print("attention calls shadowed:", len(calls), " differing:", sum(1 for c in calls if c[2]))
for c in calls[:4]:
    print(c)
