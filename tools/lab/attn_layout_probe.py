#!/usr/bin/env python3
"""Lab probe: the encoder attention kernel on the row-major qkv activation ([token][3*heads*64], an item's K is 197
pieces of 128 B at a 4,608 B stride) against the same items stored contiguously (heads = 1, ld = 192: 75 KB per
item in one run).  Same kernel, same bytes, same arithmetic; only the DRAM access pattern differs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

capi.load_library()
tok, H, frames = 197, 12, 480


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


qkv = torch.randn(frames * tok, 3 * H * 64, device="cuda").to(torch.bfloat16)
out = torch.empty(frames * tok, H * 64, device="cuda", dtype=torch.bfloat16)
a = timed(lambda: capi.attention_fwd(qkv, out, frames, tok, H))
qkv1 = torch.randn(frames * H * tok, 3 * 64, device="cuda").to(torch.bfloat16)
out1 = torch.empty(frames * H * tok, 64, device="cuda", dtype=torch.bfloat16)
b = timed(lambda: capi.attention_fwd(qkv1, out1, frames * H, tok, 1))
byt = frames * tok * 4 * H * 64 * 2
print(f"row-major qkv (128 B pieces, 4608 B stride): {a:.1f} us  {byt / a / 1e6:.2f} TB/s")
print(f"item-contiguous qkv (heads=1, ld=192):       {b:.1f} us  {byt / b / 1e6:.2f} TB/s")
