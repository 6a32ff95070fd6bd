#!/usr/bin/env python3
"""Microbenchmark of the decoder's cross-attention forward (one query per clip over T*P exported keys / values)
through the C ABI: K and V are streamed once, so the figure of merit is bytes / time.  Sweeps the split count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

capi.load_library()
B, T, P, H = 16, 30, 196, 12
D = H * 64
S = T * P
k = torch.randn(B * S, D, device="cuda").to(torch.bfloat16)
v = torch.randn(B * S, D, device="cuda").to(torch.bfloat16)
q = torch.randn(B, 2 * D, device="cuda")
mask = torch.ones(B, T, dtype=torch.uint8, device="cuda")
mix, stats = torch.empty(B, D, device="cuda"), torch.empty(B, H, 2, device="cuda")
splits_list = [int(a) for a in sys.argv[1:]] or [24, 48, 96, 192]
# the same rows read in place out of a q|k|v activation [frames, tokens, 3D] (CLS row skipped, positional embedding added
# on the fly): the default hand-over without an adapter
qkv = torch.randn(B * T, P + 1, 3 * D, device="cuda").to(torch.bfloat16)
kview, vview, pos = qkv[:, 1:, D:2 * D], qkv[:, 1:, 2 * D:], torch.randn(T, D, device="cuda")
for splits in splits_list:
    ws = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4, device="cuda")
    run = lambda: capi.decoder_attn_fwd(q, kview, vview, mask, mix, stats, ws, splits, B, T, P, H, pos=pos)
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
    print(f"decoder_attn_fwd IN PLACE B={B} S={S} splits={splits}: {us:.1f} us (partial + combine)  {2 * B * S * D * 2 / us / 1e6:.2f} TB/s")
for splits in splits_list:
    ws = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4, device="cuda")
    run = lambda: capi.decoder_attn_fwd(q, k, v, mask, mix, stats, ws, splits, B, T, P, H)
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
    byt = 2 * B * S * D * 2
    print(f"decoder_attn_fwd B={B} S={S} splits={splits}: {us:.1f} us (partial + combine)  {byt / us / 1e6:.2f} TB/s")

# backward (dq and the positional-embedding gradient; K/V gradients not requested), dense export and in place
mix_s = torch.empty(B, D, device="cuda")
ws = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, 48) // 4, device="cuda")
capi.decoder_attn_fwd(q, k, v, mask, mix, stats, ws, 48, B, T, P, H, mix_softmax=mix_s)
ws_b = torch.empty(capi.decoder_attn_bwd_workspace_bytes(B, T, H) // 4, device="cuda")
dmix, dq, dpos = torch.randn(B, D, device="cuda"), torch.empty(B, 2 * D, device="cuda"), torch.empty(T, D, device="cuda")
for name, (kk, vv, pp) in {"dense": (k, v, None), "IN PLACE": (kview, vview, pos)}.items():
    run = lambda: capi.decoder_attn_bwd(q, kk, vv, mask, dmix, mix_s, stats, dq, dpos, ws_b, B, T, P, H, pos=pp)
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
    print(f"decoder_attn_bwd {name} B={B} S={S}: {us:.1f} us (kernel + reduce)  {2 * B * S * D * 2 / us / 1e6:.2f} TB/s")
