#!/usr/bin/env python3
"""Microbenchmark of the uint8 ingest kernel vs f32 patchify (B16xT30 frames)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)
capi.load_library()
n, res, patch = 480, 224, 16
out = torch.empty(n * 196, 768, device="cuda", dtype=torch.bfloat16)


def timeit(fn, iters=10):
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


f32 = torch.randn(n, 3, res, res, device="cuda")
us = timeit(lambda: capi.patchify(f32, out, res, patch))
print(f"patchify f32 224x224            : {us:8.1f} us  {(f32.numel() * 4 + out.numel() * 2) / us / 1e3:7.0f} GB/s")
for (h, w), aa in [((224, 224), False), ((256, 256), False), ((256, 256), True), ((360, 640), True), ((448, 448), True), ((150, 150), True)]:
    u8 = torch.randint(0, 256, (n, 3, h, w), device="cuda", dtype=torch.uint8)
    us = timeit(lambda: capi.preprocess_u8(u8, out, res, patch, MEAN, STD, antialias=aa))
    print(f"ingest u8 {h}x{w} antialias={int(aa)}      : {us:8.1f} us  {(u8.numel() + out.numel() * 2) / us / 1e3:7.0f} GB/s")
