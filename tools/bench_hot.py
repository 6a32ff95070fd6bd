#!/usr/bin/env python3
"""One pass over the encoder's hot kernels at the B16xT30 shapes (for PMC collection under rocprofv3):
the four GEMMs (as the encoder launches them: persistent kernel, streaming output stores; the q|k|v projection
once plain and once with the K/V export epilogue), both add-LayerNorm forms, the attention kernel and the decoder's
cross-attention forward / backward reading K and V in place out of the q|k|v activation, a few launches each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dfd_clip_amd import capi  # noqa: E402

capi.load_library()
frames, tok, H, D = 480, 197, 12, 768
M = frames * tok
dev = "cuda"
bf = torch.bfloat16
h = torch.randn(M, D, device=dev).to(bf)
u = torch.randn(M, 4 * D, device=dev).to(bf)
x = torch.randn(M, D, device=dev)
d1 = torch.randn(M, D, device=dev).to(bf)
d2 = torch.randn(M, D, device=dev).to(bf)
g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
wq = (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(bf)
wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(bf)
wf = (torch.randn(4 * D, D, device=dev) * D ** -0.5).to(bf)
wp = (torch.randn(D, 4 * D, device=dev) * (4 * D) ** -0.5).to(bf)
bq, bo, bf_, bp = (torch.randn(n, device=dev) * 0.1 for n in (3 * D, D, 4 * D, D))
qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
mix = torch.empty(M, D, device=dev, dtype=bf)
hh = torch.empty(M, D, device=dev, dtype=bf)
tpos = torch.randn(30, D, device=dev)
ke = torch.empty(frames * (tok - 1), D, device=dev, dtype=bf)
ve = torch.empty_like(ke)
B, T, P = frames // 30, 30, tok - 1
q3 = qkv.view(frames, tok, 3 * D)
kview, vview = q3[:, 1:, D:2 * D], q3[:, 1:, 2 * D:]
dq_in = torch.randn(B, 2 * D, device=dev)
mask = torch.ones(B, T, dtype=torch.uint8, device=dev)
splits = 48
ws_f = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4, device=dev)
ws_b = torch.empty(capi.decoder_attn_bwd_workspace_bytes(B, T, H) // 4, device=dev)
dmix_o, mix_s, stats = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, H, 2, device=dev)
dmix, dq, dpos = torch.randn(B, D, device=dev), torch.empty(B, 2 * D, device=dev), torch.empty(T, D, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    capi.gemm(h, wq, qkv, bq, capi.EPI_BIAS, stream_out=True)
    capi.decoder_attn_fwd(dq_in, kview, vview, mask, dmix_o, stats, ws_f, splits, B, T, P, H, mix_softmax=mix_s, pos=tpos)
    capi.decoder_attn_bwd(dq_in, kview, vview, mask, dmix, mix_s, stats, dq, dpos, ws_b, B, T, P, H, pos=tpos)
    capi.gemm(h, wq, qkv, bq, capi.EPI_QKV_EXPORT, tokens=tok, pos=tpos, k_export=ke, v_export=ve, frames_per_clip=30, stream_out=True)
    capi.attention_fwd(qkv, mix, frames, tok, H)
    capi.gemm(mix, wo, d1, bo, capi.EPI_BIAS, stream_out=True)
    capi.add_layernorm(x, d1, g, b, hh, store_x=False)
    capi.gemm(hh, wf, u, bf_, capi.EPI_BIAS_QUICKGELU, stream_out=True)
    capi.gemm(u, wp, d2, bp, capi.EPI_BIAS, stream_out=True)
    capi.add_layernorm(x, d1, g, b, hh, delta2=d2)
    x.mul_(0.5)  # keep the stream bounded over the iterations
torch.cuda.synchronize()
print("done")
