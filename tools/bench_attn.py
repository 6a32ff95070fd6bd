#!/usr/bin/env python3
"""Microbenchmark of the encoder attention kernel through the C ABI.
usage: bench_attn.py [frames] [iters] [tokens] [heads]   (defaults: ViT-B/16 at B16xT30; ViT-L/14 at B8xT30 = 240 10 257 16).
Also times the same frames in chunks of < 512 items, which take the one-workgroup-per-item kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 480
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tok = int(sys.argv[3]) if len(sys.argv) > 3 else 197
H = int(sys.argv[4]) if len(sys.argv) > 4 else 12
capi.load_library()
qkv = torch.randn(frames * tok, 3 * H * 64, device="cuda").to(torch.bfloat16)
out = torch.empty(frames * tok, H * 64, device="cuda", dtype=torch.bfloat16)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def chunks():
    step = max(1, 511 // H)
    for f0 in range(0, frames, step):
        f1 = min(frames, f0 + step)
        capi.attention_fwd(qkv[f0 * tok:f1 * tok], out[f0 * tok:f1 * tok], f1 - f0, tok, H)


byt = frames * tok * 4 * H * 64 * 2
for name, fn in (("one launch", lambda: capi.attention_fwd(qkv, out, frames, tok, H)), ("per-item kernel (chunked launches)", chunks)):
    ms = timed(fn)
    print(f"attention frames={frames} tokens={tok} heads={H} {name}: {ms * 1e3:.1f} us  {4.0 * frames * H * tok * tok * 64 / ms / 1e9:.0f} TFLOP/s  {byt / ms / 1e6:.0f} GB/s")
