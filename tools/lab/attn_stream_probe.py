#!/usr/bin/env python3
"""Lab probe: the persistent attention kernel with its compute removed (private builds, tools/lab/build/
libattn_lab{1,2}.so: 1 = staging + output stores, 2 = staging only) against the product kernel: is the item
stream itself (LDS-DMA of K, V, Q per item, one barrier per item) slower than the memory system allows?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

frames, tok, H = 480, 197, 12
qkv = torch.randn(frames * tok, 3 * H * 64, device="cuda").to(torch.bfloat16)
o = torch.empty(frames * tok, H * 64, device="cuda", dtype=torch.bfloat16)
for name in ["libattn_stamps.so", "libattn_lab1.so", "libattn_lab2.so"]:
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "lab", "build", name))
    lib.dfd_attention_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    run = lambda: lib.dfd_attention_fwd(qkv.data_ptr(), qkv.stride(0), o.data_ptr(), o.stride(0), 1, frames, tok, H, 64, 0.125, None)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    rd, wr = frames * tok * 3 * H * 64 * 2, frames * tok * H * 64 * 2
    print(f"{name:22s} {us:7.1f} us   reads {rd / us / 1e6:.2f} TB/s  (+ writes {wr / us / 1e6:.2f} TB/s where stored)")
