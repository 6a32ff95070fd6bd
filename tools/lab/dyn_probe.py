#!/usr/bin/env python3
"""Lab: one ping-pong GEMM launch with the dynamic tile hand-out against the static order (variant 2), small then large."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dfd_clip_amd import capi  # noqa: E402

capi.load_library()
for M, N, K in ((2893, 768, 768), (20000, 2304, 768), (94560, 3072, 768)):
    g = torch.Generator(device="cuda").manual_seed(M)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    outs = []
    for variant in (0, 3, 3, 3):
        capi.gemm_set_variant(variant)
        c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        capi.gemm(a, w, c, bias, capi.EPI_BIAS, stream_out=True)
        torch.cuda.synchronize()
        outs.append(c)
        print(M, N, K, "variant", variant, "path", capi.gemm_last_path(), "finite", bool(torch.isfinite(c.float()).all()),
              "equal to static", bool(torch.equal(c, outs[0])), flush=True)
    capi.gemm_set_variant(0)
