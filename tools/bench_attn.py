#!/usr/bin/env python3
"""Microbenchmark of the encoder attention kernel through the C ABI."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 480
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
capi.load_library()
tok, H = 197, 12
qkv = torch.randn(frames * tok, 3 * H * 64, device="cuda").to(torch.bfloat16)
out = torch.empty(frames * tok, H * 64, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    capi.attention_fwd(qkv, out, frames, tok, H)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    capi.attention_fwd(qkv, out, frames, tok, H)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attention frames={frames}: {ms * 1e3:.1f} us  {4.0 * frames * H * tok * tok * 64 / ms / 1e9:.0f} TFLOP/s  {(frames * tok * 4 * H * 64 * 2) / ms / 1e6:.0f} GB/s")
