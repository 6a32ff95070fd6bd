"""Lab probe: how exactly does v_mfma_scale_f32_16x16x128_f8f6f4 sum its 128 products?  Compares dfd_gemm_fp8 (f32 output
via bf16? no: uses wide-range e4m3 operands and unit column scales) against fp64 on the same e4m3 values and relates the
error to the largest product of each dot product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dfd_clip_amd import capi
capi.load_library()
torch.manual_seed(0)
M, N = 1024, 256
for K, sa, sw in ((256, 4.0, 8.0), (768, 4.0, 8.0), (768, 1.0, 1.0), (4096, 1.0, 0.05)):
    a = (torch.randn(M, K) * sa).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K) * sw).to(torch.float8_e4m3fn)
    af, wf = a.double().cuda(), w.double().cuda()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    capi.gemm_fp8(a.view(torch.uint8).cuda(), w.view(torch.uint8).cuda(), c, torch.ones(N, device="cuda"))
    ref = af @ wf.T
    rows = slice(0, 64)
    prod = (af[rows, None, :] * wf[None, :, :]).abs()           # [64, N, K]
    maxp = prod.amax(dim=-1)
    sump = prod.sum(dim=-1)
    # per 128-block maxima (the instruction's depth)
    blk = prod.view(64, N, K // 128, 128).amax(dim=-1).sum(dim=-1)
    err = (c[rows].double() - ref[rows]).abs()
    bf = ref[rows].abs() * 2 ** -9
    excess = (err - bf).clamp_min(0)
    print(f"K={K} sa={sa} sw={sw}: max|ref|={ref.abs().max():.1f}  max err={err.max():.3e}  max excess over bf16 rounding={excess.max():.3e}  "
          f"max excess/maxprod={float((excess / maxp).max()):.3e}  excess/sum_of_block_max={float((excess / blk).max()):.3e}  "
          f"excess/sum|prod|={float((excess / sump).max()):.3e}")
