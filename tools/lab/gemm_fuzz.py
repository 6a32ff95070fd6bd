#!/usr/bin/env python3
"""Lab: random shapes / epilogues / scheduling knobs through the ping-pong GEMM against round 2's persistent kernel (or the
relaunching one where that does not serve the epilogue), bit for bit; every configuration twice.  A race in the hand-counted
waits shows as a mismatch that comes and goes."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dfd_clip_amd import capi  # noqa: E402

capi.load_library()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for it in range(n_cfg):
    N = 256 * rnd.choice([1, 1, 2, 3, 4, 9, 12, 16])
    K = 128 * rnd.choice([3, 4, 5, 6, 7, 8, 12, 16, 24, 32])  # K / 64 even and >= 6
    M = rnd.choice([1024, 1100, 2893, 5000, 9999, 20000, 20001, 47001, 94560])
    if M * N > 300e6:
        M = 20000
    epi = rnd.choice(["bias", "bias", "gelu", "respos", "respos_drop", "fp8", "fp8_8"])
    if epi.startswith("respos") and rnd.random() < 0.5:
        K = 256
    opts = dict(stream_out=rnd.random() < 0.7)
    if rnd.random() < 0.3:
        opts["spare_cus"] = rnd.choice([8, 32, 100])
    if rnd.random() < 0.4 and not epi.startswith(("respos", "fp8")):
        opts["tile_blocks"] = rnd.choice([7, 8])
    g = torch.Generator(device="cuda").manual_seed(it)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    if epi.startswith("fp8"):
        if (K // 128) % 2 or K < 768:
            K = 1024
        a = (torch.randn(M, K, device="cuda", generator=g) * 4).to(torch.float8_e4m3fn).view(torch.uint8)
        w = (torch.randn(N, K, device="cuda", generator=g) * 8).to(torch.float8_e4m3fn).view(torch.uint8)
        cs = torch.rand(N, device="cuda", generator=g) * 0.02 + 0.001
        out8 = epi == "fp8_8"

        def run():
            c = torch.zeros(M, N, device="cuda", dtype=torch.uint8) if out8 else torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            capi.gemm_fp8(a, w, c, cs, bias, capi.EPI_BIAS_QUICKGELU, out_inv_scale=0.25 if out8 else 0.0, **opts)
            return c
    else:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        if epi.startswith("respos"):
            res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
            pos = torch.randn(3, N, device="cuda", generator=g)
            rng = torch.tensor([99, it], device="cuda", dtype=torch.int64)
            drop = capi.Dropout(rng, 1003, 0.3) if epi.endswith("drop") else None

            def run():
                c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                capi.gemm(a, w, c, None, capi.EPI_RESIDUAL_POS, pos=pos, tokens=197, frames_per_clip=3, residual=res, drop=drop, **opts)
                return c
        else:
            e = capi.EPI_BIAS_QUICKGELU if epi == "gelu" else capi.EPI_BIAS

            def run():
                c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
                capi.gemm(a, w, c, bias, e, **opts)
                return c
    capi.gemm_set_variant(1)
    want = run()
    capi.gemm_set_variant(rnd.choice([0, 0, 3]))
    got = [run(), run()]
    path = capi.gemm_last_path()
    capi.gemm_set_variant(0)
    torch.cuda.synchronize()
    ok = all(torch.equal(x, want) for x in got) and bool(torch.isfinite(want.float()).all())
    bad += not ok
    print(f"{it:3d} M={M} N={N} K={K} {epi:11s} {opts} path {path}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
