#!/usr/bin/env python3
"""Where does the persistent attention kernel differ from the per-item one?  attn_probe.py [frames] [tokens] [heads]
Prints, per (frame, head), the 32-query blocks with mismatching elements (and the extra row for 257 tokens)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tok = int(sys.argv[2]) if len(sys.argv) > 2 else 257
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
capi.load_library()
torch.manual_seed(5)
qkv = torch.randn(n * tok, 3 * H * 64, device="cuda")
scale = float(os.environ.get("PROBE_QK_SCALE", "1"))  # > 1: peaky softmax rows (lab tool only)
qkv[:, :2 * H * 64] *= scale
qkv = qkv.to(torch.bfloat16)
whole = torch.empty(n * tok, H * 64, device="cuda", dtype=torch.bfloat16)
capi.attention_fwd(qkv, whole, n, tok, H)
parts = torch.empty_like(whole)
step = max(1, 511 // H)
for f0 in range(0, n, step):
    f1 = min(n, f0 + step)
    capi.attention_fwd(qkv[f0 * tok:f1 * tok], parts[f0 * tok:f1 * tok], f1 - f0, tok, H)
torch.cuda.synchronize()
bad = (whole != parts).view(n, tok, H, 64)
print("mismatching elements:", int(bad.sum()), "of", bad.numel(), " max abs diff", float((whole.float() - parts.float()).abs().max()),
      " in rows 0..255:", int(bad[:, :256].sum()), " in the last row:", int(bad[:, 256:].sum()) if tok > 256 else 0)
shown = 0
for f in range(n):
    for hd in range(H):
        b = bad[f, :, hd]
        if b.any() and shown < 40:
            rows = b.any(dim=1).nonzero().flatten().tolist()
            blocks = sorted({r // 32 for r in rows})
            cols = b.any(dim=0).nonzero().flatten().tolist()
            print(f"frame {f} head {hd}: {len(rows)} rows, blocks {blocks}, rows {rows[:6]}.., cols {cols[:4]}..{cols[-1]} ({len(cols)})")
            shown += 1
per_frame = bad.view(n, -1).any(dim=1).nonzero().flatten().tolist()
print("frames with mismatches:", per_frame[:50])
if len(sys.argv) > 4 and bad.any():
    idx = bad.nonzero()
    f, t, hd, d = [int(x) for x in idx[0]]
    W = whole.view(n, tok, H, 64)
    P = parts.view(n, tok, H, 64)
    print(f"first bad item: frame {f} head {hd}")
    for (ff, tt, hh, dd) in idx[(idx[:, 0] == f) & (idx[:, 2] == hd)][:16].tolist():
        got, want = float(W[ff, tt, hh, dd]), float(P[ff, tt, hh, dd])
        where = (P[f, :, hd, :] == W[ff, tt, hh, dd]).nonzero()[:4].tolist()
        print(f"  token {tt} d {dd}: got {got:+.5f} want {want:+.5f}; got-value occurs in the item's reference at {where}")
