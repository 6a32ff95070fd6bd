// LayerNorm (fp32 statistics) and patchify — the HBM-bound row kernels of the encoder.
//
// dfd_layernorm: one 64-lane wave per row; the row lives in registers (float4 per lane per
//   256-column slab), so x is read once and y written once: algorithmic bytes per row =
//   cols * (4 + sizeof(y)).  Two-pass statistics in registers (mean, then centred sum of
//   squares), biased variance, eps inside the sqrt: the reference's nn.LayerNorm on fp32
//   (clip/model.py:157-163).
// dfd_patchify: frames [N,3,R,R] f32 -> patch rows [N*P, kpad] so that the patch conv
//   (clip/model.py:264, :277) is a plain A·Wᵀ GEMM with W = conv1.weight.view(D, 3*p*p).
#include "common.hpp"

// e4m3 output (the A operand of an fp8 GEMM, dfd_gemm_fp8): stored value = e4m3(y * inv_scale), saturated at +-448
struct fp8_t {
  unsigned char v;
};

template <typename OutT>
__device__ __forceinline__ void store_row4(OutT* p, f32x4 o, float inv_scale) {
  if constexpr (sizeof(OutT) == 4) {
    *reinterpret_cast<f32x4*>(p) = o;
  } else if constexpr (sizeof(OutT) == 2) {
    bf16x4 ob;
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = (bf16_t)o[j];
    *reinterpret_cast<bf16x4*>(p) = ob;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __builtin_fminf(__builtin_fmaxf(o[j] * inv_scale, -448.0f), 448.0f);
    unsigned pk = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0u, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], pk, true);
    *reinterpret_cast<unsigned*>(p) = pk;
  }
}

template <typename OutT, int SLABS>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, OutT* __restrict__ y,
                                                             int64_t ldy, int64_t rows, int cols, float eps, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  f32x4 v[SLABS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SLABS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < SLABS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  OutT* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < SLABS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      store_row4(yr + c, o, inv_scale);
    }
  }
}

template <typename OutT>
static int launch_ln(const float* x, int64_t ldx, const float* g, const float* b, void* y, int64_t ldy, int64_t rows,
                     int cols, float eps, float inv_scale, hipStream_t st) {
  const int slabs = (cols + 255) / 256;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  OutT* yo = static_cast<OutT*>(y);
#define LN_CASE(S)                                                                                     \
  case S:                                                                                              \
    hipLaunchKernelGGL((layernorm_rows_kernel<OutT, S>), grid, block, 0, st, x, ldx, g, b, yo, ldy, rows, cols, eps, inv_scale); \
    break;
  switch (slabs) {
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    LN_CASE(9) LN_CASE(10) LN_CASE(11) LN_CASE(12) LN_CASE(13) LN_CASE(14) LN_CASE(15) LN_CASE(16)
    default:
      dfd_set_error("dfd_layernorm: cols=%d > 4096 unsupported", cols);
      return DFD_ERR_INVALID_ARG;
  }
#undef LN_CASE
  DFD_CHECK_LAUNCH("dfd_layernorm");
  return DFD_OK;
}

// Two LayerNorms back to back over the same rows in one pass: x <- LN_a(x) (f32, in place), y = LN_b(x).  The
// encoder's ln_pre followed by the first block's ln_1 (clip/model.py:292, :221): the row stays in registers between
// the two, so x is read once instead of twice.  Same arithmetic, in the same order, as two dfd_layernorm calls.
template <typename OutT, int SLABS>
__global__ __launch_bounds__(256) void layernorm2_rows_kernel(float* __restrict__ x, int64_t ldx, const float* __restrict__ ga,
                                                              const float* __restrict__ ba, const float* __restrict__ gb,
                                                              const float* __restrict__ bb, OutT* __restrict__ y, int64_t ldy,
                                                              int64_t rows, int cols, float eps, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = x + row * ldx;
  f32x4 v[SLABS];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SLABS; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < cols) {
        if (pass == 0) v[i] = *reinterpret_cast<const f32x4*>(xr + c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < SLABS; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    const float* gamma = pass == 0 ? ga : gb;
    const float* beta = pass == 0 ? ba : bb;
#pragma unroll
    for (int i = 0; i < SLABS; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < cols) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
        if (pass == 0) {
          *reinterpret_cast<f32x4*>(xr + c) = o;
          v[i] = o;
        } else {
          store_row4(y + row * ldy + c, o, inv_scale);
        }
      }
    }
  }
}

template <typename OutT>
static int launch_ln2(float* x, int64_t ldx, const float* ga, const float* ba, const float* gb, const float* bb, void* y, int64_t ldy,
                      int64_t rows, int cols, float eps, float inv_scale, hipStream_t st) {
  const int slabs = (cols + 255) / 256;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  OutT* yo = static_cast<OutT*>(y);
#define LN2_CASE(S)                                                                                    \
  case S:                                                                                              \
    hipLaunchKernelGGL((layernorm2_rows_kernel<OutT, S>), grid, block, 0, st, x, ldx, ga, ba, gb, bb, yo, ldy, rows, cols, eps, inv_scale); \
    break;
  switch (slabs) {
    LN2_CASE(1) LN2_CASE(2) LN2_CASE(3) LN2_CASE(4) LN2_CASE(5) LN2_CASE(6) LN2_CASE(7) LN2_CASE(8)
    default:
      dfd_set_error("dfd_layernorm2: cols=%d > 2048 unsupported", cols);
      return DFD_ERR_INVALID_ARG;
  }
#undef LN2_CASE
  DFD_CHECK_LAUNCH("dfd_layernorm2");
  return DFD_OK;
}

extern "C" int dfd_layernorm2(float* x, int64_t ldx, const float* gamma_a, const float* beta_a, const float* gamma_b,
                              const float* beta_b, void* y, int64_t ldy, int y_dtype, int64_t rows, int cols, float eps,
                              float y_inv_scale, void* stream) {
  DFD_REQUIRE(x && gamma_a && beta_a && gamma_b && beta_b && y, "dfd_layernorm2: null pointer");
  DFD_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "dfd_layernorm2: cols=%d must be a multiple of 4, <= 2048", cols);
  DFD_REQUIRE(ldx >= cols && ldy >= cols && ldx % 4 == 0 && ldy % 4 == 0, "dfd_layernorm2: bad leading dimension (ldx=%lld ldy=%lld)", (long long)ldx, (long long)ldy);
  DFD_REQUIRE(dfd_aligned16(x) && dfd_aligned16(gamma_a) && dfd_aligned16(beta_a) && dfd_aligned16(gamma_b) && dfd_aligned16(beta_b) &&
                  ((uintptr_t)y & (y_dtype == DFD_FP8 ? 3 : 7)) == 0,
              "dfd_layernorm2: pointers must be 16-byte aligned");
  DFD_REQUIRE(y_dtype == DFD_F32 || y_dtype == DFD_BF16 || (y_dtype == DFD_FP8 && y_inv_scale > 0.f), "dfd_layernorm2: y_dtype=%d (fp8 needs y_inv_scale > 0)", y_dtype);
  DFD_REQUIRE(static_cast<void*>(x) != y, "dfd_layernorm2: y must not alias x");
  if (rows == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (y_dtype == DFD_F32) return launch_ln2<float>(x, ldx, gamma_a, beta_a, gamma_b, beta_b, y, ldy, rows, cols, eps, 1.f, st);
  if (y_dtype == DFD_FP8) return launch_ln2<fp8_t>(x, ldx, gamma_a, beta_a, gamma_b, beta_b, y, ldy, rows, cols, eps, y_inv_scale, st);
  return launch_ln2<bf16_t>(x, ldx, gamma_a, beta_a, gamma_b, beta_b, y, ldy, rows, cols, eps, 1.f, st);
}

extern "C" int dfd_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                             int y_dtype, int64_t rows, int cols, float eps, float y_inv_scale, void* stream) {
  DFD_REQUIRE(x && gamma && beta && y, "dfd_layernorm: null pointer");
  DFD_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 4096, "dfd_layernorm: cols=%d must be a multiple of 4, <= 4096", cols);
  DFD_REQUIRE(ldx >= cols && ldy >= cols && ldx % 4 == 0 && ldy % 4 == 0, "dfd_layernorm: bad leading dimension (ldx=%lld ldy=%lld)", (long long)ldx, (long long)ldy);
  DFD_REQUIRE(dfd_aligned16(x) && dfd_aligned16(gamma) && dfd_aligned16(beta) && ((uintptr_t)y & (y_dtype == DFD_FP8 ? 3 : 7)) == 0, "dfd_layernorm: pointers must be 16-byte aligned");
  DFD_REQUIRE(y_dtype == DFD_F32 || y_dtype == DFD_BF16 || (y_dtype == DFD_FP8 && y_inv_scale > 0.f), "dfd_layernorm: y_dtype=%d (fp8 needs y_inv_scale > 0)", y_dtype);
  if (rows == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (y_dtype == DFD_F32) return launch_ln<float>(x, ldx, gamma, beta, y, ldy, rows, cols, eps, 1.f, st);
  if (y_dtype == DFD_FP8) return launch_ln<fp8_t>(x, ldx, gamma, beta, y, ldy, rows, cols, eps, y_inv_scale, st);
  return launch_ln<bf16_t>(x, ldx, gamma, beta, y, ldy, rows, cols, eps, 1.f, st);
}

// ---- residual add + LayerNorm ----------------------------------------------------------------
// x[row] += delta[row] (f32 residual stream, updated in place), y[row] = LayerNorm(x[row]).
// The bf16 encoder path lets out_proj / c_proj write their output as a bf16 `delta` with a plain
// store epilogue instead of a read-modify-write of the f32 stream inside the GEMM (a 128 MB burst
// per wave of tiles with every CU in its epilogue at once), and folds the add into the LayerNorm
// that follows — exactly torch autocast's dataflow (Linear output in bf16, added to the fp32 stream).
// Bytes per row: cols * (4 + sizeof(delta) + 4 + sizeof(y)).
template <typename DeltaT, typename OutT, int SLABS>
__global__ __launch_bounds__(256) void add_layernorm_rows_kernel(float* __restrict__ x, int64_t ldx,
                                                                 const DeltaT* __restrict__ delta, int64_t ldd,
                                                                 const DeltaT* __restrict__ delta2, int store_x,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, OutT* __restrict__ y,
                                                                 int64_t ldy, int64_t rows, int cols, float eps, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* xr = x + row * ldx;
  const DeltaT* dr = delta + row * ldd;
  f32x4 v[SLABS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SLABS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + c);
      auto add = [&](const DeltaT* p) {
        if constexpr (sizeof(DeltaT) == 4) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(p + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] += d[j];
        } else {
          const bf16x4 d = *reinterpret_cast<const bf16x4*>(p + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] += (float)d[j];
        }
      };
      add(dr);                                             // (x + delta) + delta2: the order of the two
      if (delta2 != nullptr) add(delta2 + row * ldd);      // separate adds it replaces
      if (store_x) *reinterpret_cast<f32x4*>(xr + c) = v[i];
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < SLABS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  OutT* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < SLABS; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      store_row4(yr + c, o, inv_scale);
    }
  }
}

template <typename DeltaT, typename OutT>
static int launch_add_ln(float* x, int64_t ldx, const void* delta, int64_t ldd, const void* delta2, int store_x, const float* g,
                         const float* b, void* y, int64_t ldy, int64_t rows, int cols, float eps, float inv_scale, hipStream_t st) {
  const int slabs = (cols + 255) / 256;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const DeltaT* dd = static_cast<const DeltaT*>(delta);
  const DeltaT* dd2 = static_cast<const DeltaT*>(delta2);
  OutT* yo = static_cast<OutT*>(y);
#define ALN_CASE(S)                                                                                                   \
  case S:                                                                                                             \
    hipLaunchKernelGGL((add_layernorm_rows_kernel<DeltaT, OutT, S>), grid, block, 0, st, x, ldx, dd, ldd, dd2, store_x, g, b,  \
                       yo, ldy, rows, cols, eps, inv_scale);                                                          \
    break;
  switch (slabs) {
    ALN_CASE(1) ALN_CASE(2) ALN_CASE(3) ALN_CASE(4) ALN_CASE(5) ALN_CASE(6) ALN_CASE(7) ALN_CASE(8)
    default:
      dfd_set_error("dfd_add_layernorm: cols=%d > 2048 unsupported", cols);
      return DFD_ERR_INVALID_ARG;
  }
#undef ALN_CASE
  DFD_CHECK_LAUNCH("dfd_add_layernorm");
  return DFD_OK;
}

extern "C" int dfd_add_layernorm(float* x, int64_t ldx, const void* delta, const void* delta2, int64_t ldd, int delta_dtype,
                                 int store_x, const float* gamma, const float* beta, void* y, int64_t ldy, int y_dtype,
                                 int64_t rows, int cols, float eps, float y_inv_scale, void* stream) {
  DFD_REQUIRE(x && delta && gamma && beta && y, "dfd_add_layernorm: null pointer");
  DFD_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "dfd_add_layernorm: cols=%d must be a multiple of 4, <= 2048", cols);
  DFD_REQUIRE(ldx >= cols && ldd >= cols && ldy >= cols && ldx % 4 == 0 && ldd % 4 == 0 && ldy % 4 == 0,
              "dfd_add_layernorm: bad leading dimension (ldx=%lld ldd=%lld ldy=%lld)", (long long)ldx, (long long)ldd, (long long)ldy);
  DFD_REQUIRE(dfd_aligned16(x) && dfd_aligned16(gamma) && dfd_aligned16(beta) && ((uintptr_t)y & (y_dtype == DFD_FP8 ? 3 : 7)) == 0 && ((uintptr_t)delta & 7) == 0,
              "dfd_add_layernorm: pointers must be 16-byte aligned (8 for bf16, 4 for fp8 operands)");
  DFD_REQUIRE((delta_dtype == DFD_F32 || delta_dtype == DFD_BF16) && (y_dtype == DFD_F32 || y_dtype == DFD_BF16 || (y_dtype == DFD_FP8 && y_inv_scale > 0.f)),
              "dfd_add_layernorm: delta_dtype=%d y_dtype=%d (fp8 needs y_inv_scale > 0)", delta_dtype, y_dtype);
  DFD_REQUIRE(static_cast<const void*>(x) != y && delta != y && delta2 != y, "dfd_add_layernorm: y must not alias x or a delta");
  DFD_REQUIRE(!delta2 || ((uintptr_t)delta2 & 7) == 0, "dfd_add_layernorm: delta2 must be 8-byte aligned");
  if (rows == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (delta_dtype == DFD_F32) {
    if (y_dtype == DFD_F32) return launch_add_ln<float, float>(x, ldx, delta, ldd, delta2, store_x, gamma, beta, y, ldy, rows, cols, eps, 1.f, st);
    if (y_dtype == DFD_FP8) return launch_add_ln<float, fp8_t>(x, ldx, delta, ldd, delta2, store_x, gamma, beta, y, ldy, rows, cols, eps, y_inv_scale, st);
    return launch_add_ln<float, bf16_t>(x, ldx, delta, ldd, delta2, store_x, gamma, beta, y, ldy, rows, cols, eps, 1.f, st);
  }
  if (y_dtype == DFD_F32) return launch_add_ln<bf16_t, float>(x, ldx, delta, ldd, delta2, store_x, gamma, beta, y, ldy, rows, cols, eps, 1.f, st);
  if (y_dtype == DFD_FP8) return launch_add_ln<bf16_t, fp8_t>(x, ldx, delta, ldd, delta2, store_x, gamma, beta, y, ldy, rows, cols, eps, y_inv_scale, st);
  return launch_add_ln<bf16_t, bf16_t>(x, ldx, delta, ldd, delta2, store_x, gamma, beta, y, ldy, rows, cols, eps, 1.f, st);
}

// ---- patchify -----------------------------------------------------------------------------
// Output column k = c*p*p + i*p + j of patch (gy, gx) reads frame pixel (c, gy*p + i, gx*p + j).
//
// Strip form (bf16 output, p % 4 == 0, no K padding: ViT-B/16, ViT-B/32): one workgroup per (frame, patch row gy).
// Its input is 3*p image rows read as whole rows (16 bytes per lane, fully coalesced); its output, the grid_w patch
// rows of that strip, is ONE contiguous run (grid_w * 3*p*p bf16).  The transpose goes through LDS: a patch's
// 3*p*p bf16 at stride 3*p*p*2 + 32 bytes (the 8-byte writes of neighbouring patches land 8 banks apart).
__global__ __launch_bounds__(256) void patchify_strip_kernel(const float* __restrict__ frames, bf16_t* __restrict__ out, int res,
                                                             int patch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char strip[];
  const int grid_w = res / patch;
  const int kk = 3 * patch * patch;
  const int pstride = kk * 2 + 32;
  const int n = blockIdx.x / grid_w, gy = blockIdx.x % grid_w;
  const int qrow = res >> 2;  // 16-byte quads per image row
  const int nq = 3 * patch * qrow;
  for (int idx = threadIdx.x; idx < nq; idx += 256) {
    const int rw = idx / qrow, xq = idx - rw * qrow;  // rw = c*patch + i
    const int c = rw / patch, i = rw - c * patch;
    const f32x4 v = *reinterpret_cast<const f32x4*>(frames + (((int64_t)n * 3 + c) * res + (gy * patch + i)) * res + xq * 4);
    const int x = xq * 4, gx = x / patch, j = x - gx * patch;
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x4*>(strip + gx * pstride + (rw * patch + j) * 2) = o;
  }
  __syncthreads();
  const int cpp = kk >> 3;  // 16-byte chunks per patch
  bf16_t* dst = out + ((int64_t)n * grid_w * grid_w + (int64_t)gy * grid_w) * kk;
  for (int idx = threadIdx.x; idx < grid_w * cpp; idx += 256) {
    const int pp = idx / cpp, ch = idx - pp * cpp;
    *reinterpret_cast<bf16x8*>(dst + (int64_t)idx * 8) = *reinterpret_cast<const bf16x8*>(strip + pp * pstride + ch * 16);
  }
}

// General form: one thread per 4 consecutive output columns of one patch row.  Reads are contiguous along j
// (p pixels = 64 B for p=16), writes fully coalesced.
template <typename OutT>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ frames, OutT* __restrict__ out,
                                                       int n_frames, int res, int patch, int kpad) {
  const int grid_w = res / patch;
  const int P = grid_w * grid_w;
  const int kq = kpad >> 2;  // column quads per row
  const int64_t total = (int64_t)n_frames * P * kq;
  const int kreal = 3 * patch * patch;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % kq);
    const int64_t row = idx / kq;
    const int p = (int)(row % P);
    const int n = (int)(row / P);
    const int gy = p / grid_w, gx = p % grid_w;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = q * 4 + e;
      if (k < kreal) {
        const int c = k / (patch * patch);
        const int r = k % (patch * patch);
        const int i = r / patch, j = r % patch;
        v[e] = frames[(((int64_t)n * 3 + c) * res + (gy * patch + i)) * res + gx * patch + j];
      } else {
        v[e] = 0.f;
      }
    }
    OutT* o = out + row * kpad + q * 4;
    if constexpr (sizeof(OutT) == 4) {
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      bf16x4 ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) ob[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x4*>(o) = ob;
    }
  }
}

extern "C" int dfd_patchify(const float* frames, void* patches, int out_dtype, int n_frames, int res, int patch,
                            int kpad, void* stream) {
  DFD_REQUIRE(frames && patches, "dfd_patchify: null pointer");
  DFD_REQUIRE(n_frames >= 0 && res > 0 && patch > 0 && res % patch == 0, "dfd_patchify: res=%d patch=%d", res, patch);
  DFD_REQUIRE(kpad % 4 == 0 && kpad >= 3 * patch * patch, "dfd_patchify: kpad=%d too small or not a multiple of 4", kpad);
  DFD_REQUIRE(out_dtype == DFD_F32 || out_dtype == DFD_BF16, "dfd_patchify: out_dtype=%d", out_dtype);
  DFD_REQUIRE(dfd_aligned16(patches), "dfd_patchify: output must be 16-byte aligned");
  if (n_frames == 0) return DFD_OK;
  const int P = (res / patch) * (res / patch);
  const int64_t total = (int64_t)n_frames * P * (kpad / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int kk = 3 * patch * patch;
  const size_t strip_lds = (size_t)(res / patch) * (kk * 2 + 32);
  if (out_dtype == DFD_BF16 && patch % 4 == 0 && kpad == kk && strip_lds <= 64 * 1024 && dfd_aligned16(frames) &&
      (int64_t)n_frames * (res / patch) < (int64_t)0x7fffffff) {
    hipLaunchKernelGGL(patchify_strip_kernel, dim3((unsigned)(n_frames * (res / patch))), dim3(256), strip_lds, st, frames,
                       static_cast<bf16_t*>(patches), res, patch);
    DFD_CHECK_LAUNCH("dfd_patchify(strip)");
    return DFD_OK;
  }
  if (out_dtype == DFD_F32)
    hipLaunchKernelGGL((patchify_kernel<float>), dim3(blocks), dim3(256), 0, st, frames, static_cast<float*>(patches), n_frames, res, patch, kpad);
  else
    hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, frames, static_cast<bf16_t*>(patches), n_frames, res, patch, kpad);
  DFD_CHECK_LAUNCH("dfd_patchify");
  return DFD_OK;
}
