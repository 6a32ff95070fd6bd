// Device-side frame ingest: uint8 video frames -> normalised patch rows (or f32 frames).
//
// Replaces the dataloader-side `Detector._transform` of the reference (src/models.py:756-768:
// Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ConvertImageDtype(float32) -> Normalize) for
// uint8 tensors, fused with the patch extraction that feeds the patch-embed GEMM
// (clip/model.py:277-279), so a clip crosses PCIe as 1 byte/sample instead of 4 and never
// exists as an f32 frame in HBM.
//
// Arithmetic restated (torchvision's tensor path delegates to ATen):
//   resize   : bicubic, align_corners=False.  antialias=0: upsample_bicubic2d (4 taps per axis,
//              A=-0.75, source index clamped to the border).  antialias=1:
//              _upsample_bicubic2d_aa (A=-0.5, support 2*max(scale,1), taps normalised,
//              horizontal pass then vertical pass).  Computed in f32, horizontal first.
//   u8 grid  : round-half-even, clamp to [0,255] (torchvision rounds a resized uint8 image back
//              to uint8 before ConvertImageDtype); skipped when no resize is needed.
//   convert  : v / 255, then (v - mean[c]) / std[c]   (IEEE f32 divisions, as ATen does).
//
// One 256-thread block per (frame, patch): the filter taps are computed once and shared by the three
// channels; per channel the source window of the patch is staged in LDS as f32, filtered horizontally into a
// second LDS buffer, then vertically into registers.  Without a resize the patch-row form takes a gather
// kernel instead (8 pixels per thread, 16-byte stores).
// HBM traffic: u8 source read once (neighbouring blocks share window edges through L2) +
// output written once; the kernel is bound by that stream.
#include "common.hpp"

namespace {

struct PreArgs {
  const uint8_t* src;
  void* out;
  int n_frames, in_h, in_w;    // source frames [n, 3, in_h, in_w]
  int rs_h, rs_w;              // size after Resize
  int top, left;               // CenterCrop origin in the resized image
  int res, patch, grid_w;      // output res x res, patch size, patches per side
  int antialias, identity;     // identity: rs == in (no filtering, no re-quantisation)
  int layout;                  // 0: frames [n,3,res,res]; 1: patch rows [n*P, kpad]
  int kpad;
  int win;                     // LDS window side (rows and cols) reserved per block
  int max_taps;
  float scale_y, scale_x;      // in / rs
  float mean[3], stdv[3];
};

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// PIL-style antialias filter (A = -0.5), aten/native/UpSample.h: bicubic aa_filter
__device__ __forceinline__ float aa_filter(float x) {
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}

// Tap list of one output coordinate `o` (in resized-image coordinates): first source index
// (may be < 0 for the plain bicubic: clamped on read), count and weights.
__device__ void make_taps(int o, float scale, int in_size, int antialias, int* start, int* count, float* w, int max_taps) {
  if (!antialias) {
    const float real = scale * ((float)o + 0.5f) - 0.5f;
    const float fl = floorf(real);
    const float t = real - fl;
    *start = (int)fl - 1;
    *count = 4;
    const float A = -0.75f;
    w[0] = cubic2(t + 1.f, A);
    w[1] = cubic1(t, A);
    w[2] = cubic1(1.f - t, A);
    w[3] = cubic2(2.f - t, A);
    return;
  }
  const float support = scale >= 1.f ? 2.f * scale : 2.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * ((float)o + 0.5f);
  int xmin = (int)(center - support + 0.5f);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5f);
  if (xmax > in_size) xmax = in_size;
  int n = xmax - xmin;
  if (n > max_taps) n = max_taps;
  float total = 0.f;
  for (int j = 0; j < n; ++j) {
    const float v = aa_filter(((float)(j + xmin) - center + 0.5f) * invscale);
    w[j] = v;
    total += v;
  }
  const float inv = total != 0.f ? 1.f / total : 0.f;
  for (int j = 0; j < n; ++j) w[j] *= inv;
  *start = xmin;
  *count = n;
}

template <typename OutT>
__global__ __launch_bounds__(256) void preprocess_u8_kernel(PreArgs a) {
  extern __shared__ float lds[];
  const int p = a.patch;
  const int P = a.grid_w * a.grid_w;
  const int pidx = blockIdx.x % P;
  const int n = blockIdx.x / P;
  const int py = pidx / a.grid_w, px = pidx % a.grid_w;
  const int tid = threadIdx.x;

  // LDS carve-up: window [win][win] | horizontal result [win][p] | taps (shared by the three channels)
  float* win = lds;
  float* hbuf = win + a.win * a.win;
  float* wx = hbuf + a.win * p;               // [p][max_taps]
  float* wy = wx + p * a.max_taps;            // [p][max_taps]
  int* sx = reinterpret_cast<int*>(wy + p * a.max_taps);  // [p] start, [p] count
  int* cx = sx + p;
  int* sy = cx + p;
  int* cy = sy + p;

  const int oy0 = a.top + py * p, ox0 = a.left + px * p;
  auto emit = [&](int c, int i, int j, int e, float o) {
    if (a.layout == 0) {
      static_cast<OutT*>(a.out)[(((int64_t)n * 3 + c) * a.res + (py * p + i)) * a.res + px * p + j] = from_f32<OutT>(o);
    } else {
      static_cast<OutT*>(a.out)[((int64_t)n * P + pidx) * a.kpad + c * p * p + e] = from_f32<OutT>(o);
    }
  };
  if (a.identity) {
    // no resize: plain gather of the patch
    for (int c = 0; c < 3; ++c) {
      const uint8_t* img = a.src + ((int64_t)n * 3 + c) * a.in_h * a.in_w;
      for (int e = tid; e < p * p; e += 256) {
        const int i = e / p, j = e % p;
        const float v = (float)img[(int64_t)(oy0 + i) * a.in_w + ox0 + j];
        emit(c, i, j, e, (v / 255.f - a.mean[c]) / a.stdv[c]);
      }
    }
  } else {
    if (tid < p) {
      make_taps(ox0 + tid, a.scale_x, a.in_w, a.antialias, &sx[tid], &cx[tid], wx + tid * a.max_taps, a.max_taps);
    } else if (tid >= 64 && tid < 64 + p) {
      const int t = tid - 64;
      make_taps(oy0 + t, a.scale_y, a.in_h, a.antialias, &sy[t], &cy[t], wy + t * a.max_taps, a.max_taps);
    }
    __syncthreads();
    // window bounds = union of the tap ranges (they are monotone in the output coordinate)
    const int x_lo = sx[0], x_hi = sx[p - 1] + cx[p - 1];
    const int y_lo = sy[0], y_hi = sy[p - 1] + cy[p - 1];
    const int ww = x_hi - x_lo, wh = y_hi - y_lo;  // host sized a.win >= both
    for (int c = 0; c < 3; ++c) {
      const uint8_t* img = a.src + ((int64_t)n * 3 + c) * a.in_h * a.in_w;
      const float mean = a.mean[c], stdv = a.stdv[c];
      if (c > 0) __syncthreads();  // the previous channel's vertical pass is done with hbuf
      for (int e = tid; e < wh * ww; e += 256) {
        const int r = e / ww, q = e % ww;
        int yy = y_lo + r, xx = x_lo + q;
        yy = yy < 0 ? 0 : (yy >= a.in_h ? a.in_h - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= a.in_w ? a.in_w - 1 : xx);
        win[r * a.win + q] = (float)img[(int64_t)yy * a.in_w + xx];
      }
      __syncthreads();
      // horizontal pass: hbuf[r][j] = sum_k wx[j][k] * win[r][sx[j] - x_lo + k]
      for (int e = tid; e < wh * p; e += 256) {
        const int r = e / p, j = e % p;
        const float* wrow = win + r * a.win + (sx[j] - x_lo);
        const float* wj = wx + j * a.max_taps;
        float acc = 0.f;
        if (!a.antialias) {
          acc = wrow[0] * wj[0] + wrow[1] * wj[1] + wrow[2] * wj[2] + wrow[3] * wj[3];
        } else {
          acc = wrow[0] * wj[0];
          for (int k = 1; k < cx[j]; ++k) acc += wrow[k] * wj[k];
        }
        hbuf[r * p + j] = acc;
      }
      __syncthreads();
      for (int e = tid; e < p * p; e += 256) {
        const int i = e / p, j = e % p;
        const float* wi = wy + i * a.max_taps;
        const int r0 = sy[i] - y_lo;
        float acc;
        if (!a.antialias) {
          acc = hbuf[r0 * p + j] * wi[0] + hbuf[(r0 + 1) * p + j] * wi[1] + hbuf[(r0 + 2) * p + j] * wi[2] +
                hbuf[(r0 + 3) * p + j] * wi[3];
        } else {
          acc = hbuf[r0 * p + j] * wi[0];
          for (int k = 1; k < cy[i]; ++k) acc += hbuf[(r0 + k) * p + j] * wi[k];
        }
        float v = rintf(acc);                        // back onto the uint8 grid (round half even)
        v = fminf(fmaxf(v, 0.f), 255.f);
        emit(c, i, j, e, (v / 255.f - mean) / stdv);
      }
    }
  }
  if (a.layout == 1) {
    OutT* row = static_cast<OutT*>(a.out) + ((int64_t)n * P + pidx) * a.kpad;
    for (int k = 3 * p * p + tid; k < a.kpad; k += 256) row[k] = from_f32<OutT>(0.f);
  }
}

// No-resize fast path for patch rows (patch % 8 == 0, width % 8 == 0): one thread per 8 consecutive output
// columns = 8 horizontally adjacent pixels of one patch row.  A wave writes 1 KiB of a patch row contiguously
// (16 B per lane) and reads 8-byte pieces that neighbouring patches complete to full cache lines through L2.
template <typename OutT>
__global__ __launch_bounds__(256) void preprocess_identity8_kernel(PreArgs a) {
  const int p = a.patch;
  const int P = a.grid_w * a.grid_w;
  const int per_row = 3 * p * p / 8;  // 8-column groups of real data per patch row
  const int64_t total = (int64_t)a.n_frames * P * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int g8 = (int)(idx % per_row);
    const int64_t prow = idx / per_row;
    const int pidx = (int)(prow % P);
    const int64_t n = prow / P;
    const int k = g8 * 8;
    const int c = k / (p * p), r = k % (p * p);
    const int i = r / p, j = r % p;
    const int y = a.top + (pidx / a.grid_w) * p + i, x = a.left + (pidx % a.grid_w) * p + j;
    const uint8_t* src = a.src + ((n * 3 + c) * a.in_h + y) * (int64_t)a.in_w + x;
    uint8_t px[8];
    if ((reinterpret_cast<uintptr_t>(src) & 7) == 0) {
      *reinterpret_cast<uint2*>(px) = *reinterpret_cast<const uint2*>(src);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) px[e] = src[e];
    }
    const float mean = a.mean[c], stdv = a.stdv[c];
    OutT* dst = static_cast<OutT*>(a.out) + prow * a.kpad + k;
    if constexpr (sizeof(OutT) == 2) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(((float)px[e] / 255.f - mean) / stdv);
      *reinterpret_cast<bf16x8*>(dst) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = ((float)px[e] / 255.f - mean) / stdv;
    }
  }
}

// torchvision.transforms.functional._compute_resized_output_size for an int size
static void resized_size(int h, int w, int size, int* nh, int* nw) {
  const int s = h < w ? h : w, l = h < w ? w : h;
  const int new_s = size, new_l = (int)((int64_t)size * l / s);
  if (h <= w) { *nh = new_s; *nw = new_l; } else { *nh = new_l; *nw = new_s; }
}

// Python round() of x/2 for non-negative x: half to even
static int round_half_even_div2(int x) {
  const int q = x / 2;
  return (x & 1) ? (q + (q & 1)) : q;
}

}  // namespace

extern "C" int dfd_preprocess_geometry(int in_h, int in_w, int res, int* rs_h, int* rs_w, int* top, int* left) {
  DFD_REQUIRE(in_h > 0 && in_w > 0 && res > 0 && rs_h && rs_w && top && left, "dfd_preprocess_geometry: bad argument");
  resized_size(in_h, in_w, res, rs_h, rs_w);
  *top = round_half_even_div2(*rs_h - res);
  *left = round_half_even_div2(*rs_w - res);
  return DFD_OK;
}

extern "C" int dfd_preprocess_u8(const uint8_t* frames, int n_frames, int in_h, int in_w, int res, int patch,
                                 int antialias, const float* mean3, const float* std3, void* out, int out_dtype,
                                 int layout, int kpad, void* stream) {
  DFD_REQUIRE(frames && out && mean3 && std3, "dfd_preprocess_u8: null pointer");
  DFD_REQUIRE(n_frames >= 0 && in_h > 0 && in_w > 0 && res > 0 && patch > 0 && res % patch == 0 && patch <= 64,
              "dfd_preprocess_u8: bad shape (n=%d in=%dx%d res=%d patch=%d)", n_frames, in_h, in_w, res, patch);
  DFD_REQUIRE(out_dtype == DFD_F32 || out_dtype == DFD_BF16, "dfd_preprocess_u8: out_dtype=%d", out_dtype);
  DFD_REQUIRE(layout == 0 || (layout == 1 && kpad >= 3 * patch * patch), "dfd_preprocess_u8: layout=%d kpad=%d", layout, kpad);
  DFD_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "dfd_preprocess_u8: zero std");
  if (n_frames == 0) return DFD_OK;
  PreArgs a;
  a.src = frames;
  a.out = out;
  a.n_frames = n_frames;
  a.in_h = in_h;
  a.in_w = in_w;
  resized_size(in_h, in_w, res, &a.rs_h, &a.rs_w);
  a.top = round_half_even_div2(a.rs_h - res);
  a.left = round_half_even_div2(a.rs_w - res);
  a.res = res;
  a.patch = patch;
  a.grid_w = res / patch;
  a.antialias = antialias ? 1 : 0;
  a.identity = (a.rs_h == in_h && a.rs_w == in_w) ? 1 : 0;
  a.layout = layout;
  a.kpad = kpad;
  a.scale_y = (float)in_h / (float)a.rs_h;
  a.scale_x = (float)in_w / (float)a.rs_w;
  const float smax = a.scale_y > a.scale_x ? a.scale_y : a.scale_x;
  const float support = (a.antialias && smax >= 1.f) ? 2.f * smax : 2.f;
  a.max_taps = a.antialias ? (int)(2.f * support) + 3 : 4;
  a.win = (int)((float)patch * smax) + 2 * (int)(support + 1.f) + 4;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean3[i]; a.stdv[i] = std3[i]; }
  const size_t lds_bytes = sizeof(float) * ((size_t)a.win * a.win + (size_t)a.win * patch + 2 * (size_t)patch * a.max_taps) +
                           sizeof(int) * 4 * (size_t)patch;
  DFD_REQUIRE(lds_bytes <= 150 * 1024, "dfd_preprocess_u8: downscale %.2fx with patch %d needs %zu B of LDS (> 150 KB)", (double)smax, patch, lds_bytes);
  const int64_t blocks = (int64_t)n_frames * a.grid_w * a.grid_w;
  DFD_REQUIRE(blocks < (1ll << 31), "dfd_preprocess_u8: too many frames");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.identity && layout == 1 && patch % 8 == 0 && kpad == 3 * patch * patch && (kpad * (out_dtype == DFD_F32 ? 4 : 2)) % 16 == 0 &&
      dfd_aligned16(out)) {
    const int64_t total = (int64_t)n_frames * a.grid_w * a.grid_w * (3 * patch * patch / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 16 ? (total + 255) / 256 : 65536 * 16);
    if (out_dtype == DFD_F32)
      hipLaunchKernelGGL(preprocess_identity8_kernel<float>, dim3(grid), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(preprocess_identity8_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, a);
    DFD_CHECK_LAUNCH("dfd_preprocess_u8");
    return DFD_OK;
  }
  if (out_dtype == DFD_F32) {
    if (lds_bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&preprocess_u8_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(preprocess_u8_kernel<float>, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
  } else {
    if (lds_bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&preprocess_u8_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(preprocess_u8_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
  }
  DFD_CHECK_LAUNCH("dfd_preprocess_u8");
  return DFD_OK;
}
