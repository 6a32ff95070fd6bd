#!/usr/bin/env python3
"""Microbenchmark of dfd_layernorm (f32 rows in, bf16 rows out) through the C ABI."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 480 * 197
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 768
iters = 20
capi.load_library()
x = torch.randn(rows, cols, device="cuda")
g = torch.randn(cols, device="cuda")
b = torch.randn(cols, device="cuda")
y = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    capi.layernorm(x, g, b, y, 1e-5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    capi.layernorm(x, g, b, y, 1e-5)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"layernorm rows={rows} cols={cols}: {ms * 1e3:.1f} us  {rows * cols * 6 / ms / 1e6:.0f} GB/s")
