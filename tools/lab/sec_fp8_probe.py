#!/usr/bin/env python3
"""Lab: the fp8 ViT-L/14 B8 forward leg of bench.py's `secondary`, alone or after the bf16 leg, with per-pass times."""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--no-secondary", "--no-cpu-baseline"]
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
legs = [("bf16", 8), ("fp8", 8)] if os.environ.get("AFTER_BF16") else [("fp8", 8)]
if os.environ.get("AFTER_HEADLINE"):  # a ViT-B/16 train loop first, as bench.py's headline does, and the model stays alive
    hd, _, _, _ = bench.build_model(args, dev)
    hd.static_graphs, hd.pipeline_encoder, hd.inputs_ready = True, True, True
    hd.train()
    hx = torch.randn(16, 30, 3, 224, 224, device=dev)
    hm = torch.ones(16, 30, dtype=torch.bool, device=dev)
    hy = torch.arange(16, device=dev) % 2
    hopt = hd.configure_optimizers(0.01 / 25)
    mode = os.environ["AFTER_HEADLINE"]
    for _ in range(8):
        hd.zero_grad()
        losses, _, other = hd(hx, [hy], hm, train=True, single_task=0)
        (losses[0].mean() + sum(other.values())).backward()
        hopt.step()
    torch.cuda.synchronize()
    if mode == "infer" or mode == "both":
        hd.eval()
        with torch.no_grad():
            for _ in range(6):
                hd.predict(hx, hm)
        torch.cuda.synchronize()
    print("headline phase done:", mode)
for prec, clips in legs:
    a = copy.copy(args)
    a.arch, a.precision, a.clips, a.adapter = "ViT-L/14", prec, clips, "none"
    det, _, _, _ = bench.build_model(a, dev)
    det.eval()
    det.static_graphs, det.pipeline_encoder, det.inputs_ready = True, True, True
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.randn(clips, args.frames, 3, 224, 224, device=dev, generator=g)
    m = torch.ones(clips, args.frames, dtype=torch.bool, device=dev)
    with torch.no_grad():
        if prec == "fp8":
            det.calibrate_fp8(x[:2])
        times = []
        for i in range(24):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            det.predict(x, m)
            torch.cuda.synchronize()
            times.append(round((time.perf_counter() - t0) * 1e3, 1))
        print(prec, "per-pass ms with a sync after each:", times, "graphs", len(det._enc_graphs), det._enc_graphs_failed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            det.predict(x, m)
        torch.cuda.synchronize()
        print(prec, "8 passes back to back:", round((time.perf_counter() - t0) / 8 * 1e3, 2), "ms per pass")
    del det, x, m
    torch.cuda.empty_cache()
