#!/usr/bin/env python3
"""Diagnostic: build the attention kernel with ATTN_STAMPS=1 into a private library and print the average
cycles wave 0 of a workgroup spends per stage (lab only; not part of the product)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

csrc = os.path.join(ROOT, "dfd-clip_amd", "csrc")
out = os.path.join(ROOT, "tools", "lab", "build", "libattn_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if "--build" in sys.argv:
    srcs = [os.path.join(csrc, f) for f in ("attention_mfma.hip", "attention.hip", "capi.hip")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DATTN_STAMPS=1", "-o", out] + srcs)
    sys.exit(0)
lib = ctypes.CDLL(out)
frames, tok, H = 480, 197, 12
qkv = torch.randn(frames * tok, 3 * H * 64, device="cuda").to(torch.bfloat16)
o = torch.empty(frames * tok, H * 64, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(frames * H, 12, device="cuda")
lib.dfd_attn_set_debug(ctypes.c_void_p(dbg.data_ptr()))
lib.dfd_attention_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
for _ in range(3):
    lib.dfd_attention_fwd(qkv.data_ptr(), qkv.stride(0), o.data_ptr(), o.stride(0), 1, frames, tok, H, 64, 0.125, None)
torch.cuda.synchronize()
m = dbg.mean(dim=0).tolist()
names = ["load issue->landed", "LDS writes", "barrier", "q load (2 blocks)", "QK", "softmax", "PV", "store", "total"]
print("attention stamps, avg cycles per workgroup (wave 0):")
for n, v in zip(names, m):
    print(f"  {n:22s} {v:9.0f}")
