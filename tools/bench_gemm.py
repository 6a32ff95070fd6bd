#!/usr/bin/env python3
"""Microbenchmark of the encoder GEMM shapes through the C ABI (random data, HIP-event timed)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=480)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--pad", action="store_true", help="round M up to a multiple of 256 (what the encoder passes: padded workspaces)")
    ap.add_argument("--shape", type=int, nargs=3, action="append", metavar=("M", "N", "K"),
                    help="time this shape with the plain bias epilogue instead of the encoder's (repeatable), e.g. --shape 4096 4096 4096")
    args = ap.parse_args()
    capi.load_library()
    if args.shape:
        for M, N, K in args.shape:
            a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)  # uniform [-1, 1): the guide's GEMM figures are quoted on it
            w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
            bias = torch.zeros(N, device="cuda")
            c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            for _ in range(5):
                capi.gemm(a, w, c, bias, capi.EPI_BIAS)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                capi.gemm(a, w, c, bias, capi.EPI_BIAS)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            print(f"plain     M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s")
        return
    M = args.frames * 197
    if args.pad:
        M = (M + 255) // 256 * 256
    D = 768
    dev = "cuda"
    shapes = [("qkv", 3 * D, D, capi.EPI_QKV_EXPORT, torch.bfloat16), ("out_proj", D, D, capi.EPI_BIAS_RESIDUAL, torch.float32),
              ("c_fc", 4 * D, D, capi.EPI_BIAS_QUICKGELU, torch.bfloat16), ("c_proj", D, 4 * D, capi.EPI_BIAS_RESIDUAL, torch.float32),
              ("out_proj/d", D, D, capi.EPI_BIAS, torch.bfloat16), ("c_proj/d", D, 4 * D, capi.EPI_BIAS, torch.bfloat16)]
    Mp = (M + 255) // 256 * 256
    if args.pad:  # padded M is only valid for the plain epilogues (what the encoder uses with padded rows)
        shapes = [("qkv/plain", 3 * D, D, capi.EPI_BIAS, torch.bfloat16)] + [s_ for s_ in shapes if s_[3] in (capi.EPI_BIAS, capi.EPI_BIAS_QUICKGELU)]
    for name, N, K, epi, cdt in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) * 0.1
        c = torch.zeros(M, N, device=dev, dtype=cdt)
        kw = dict(tokens=197) if epi == capi.EPI_QKV_EXPORT else {}
        if args.check:
            c.zero_()
            capi.gemm(a, w, c, bias, epi, **kw)
            rows = torch.randint(0, M, (512,), device=dev)
            ref = a[rows].float() @ w.float().T + bias
            if epi == capi.EPI_BIAS_QUICKGELU:
                ref = ref * torch.sigmoid(1.702 * ref)
            err = (c[rows].float() - ref).abs().max().item()
            print(f"{name}: max err on 512 sampled rows = {err:.3e} (ref scale {ref.abs().max().item():.2f})")
        for _ in range(3):
            capi.gemm(a, w, c, bias, epi, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            capi.gemm(a, w, c, bias, epi, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print(f"{name:9s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
