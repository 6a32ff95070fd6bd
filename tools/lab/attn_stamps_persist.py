#!/usr/bin/env python3
"""Diagnostic for the persistent attention kernel (ATTN_STAMPS=1 private build): cycles waves 0 and 1 of a
workgroup spend per stage, summed over the items the workgroup processed (lab only)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

out = os.path.join(ROOT, "tools", "lab", "build", "libattn_stamps.so")
lib = ctypes.CDLL(out)
frames, tok, H = 480, 197, 12
qkv = torch.randn(frames * tok, 3 * H * 64, device="cuda").to(torch.bfloat16)
o = torch.empty(frames * tok, H * 64, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(256 * 2, 8, device="cuda")
lib.dfd_attn_set_debug(ctypes.c_void_p(dbg.data_ptr()))
lib.dfd_attention_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
for _ in range(3):
    lib.dfd_attention_fwd(qkv.data_ptr(), qkv.stride(0), o.data_ptr(), o.stride(0), 1, frames, tok, H, 64, 0.125, None)
torch.cuda.synchronize()
m = dbg.view(256, 2, 8).mean(dim=0)
items = frames * H / 256
names = ["vmcnt+barrier wait", "stage + q issue", "QK", "softmax", "PV", "store + q copy", "total"]
for w in range(2):
    print(f"wave {w}: cycles per item (avg over {items:.1f} items per workgroup)")
    for n, v in zip(names, m[w].tolist()):
        print(f"  {n:22s} {v / items:9.0f}")
