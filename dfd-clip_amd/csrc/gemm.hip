// C = epilogue(A[M,K] · W[N,K]ᵀ) on the gfx950 matrix cores — the encoder's four GEMM shapes
// (QKV, out_proj, c_fc, c_proj; reference clip/model.py:186, :197, :208-212) plus the patch
// conv as a GEMM (clip/model.py:277).
//
// gemm128: general-shape kernel.  128x128 output tile per 256-thread workgroup (4 waves as
//   2x2, each wave a 64x64 patch = 2x2 MFMA 32x32 tiles), K consumed 64 BYTES per row per step
//   (32 bf16 / 16 f32), LDS rows padded to 80 B, double-buffered LDS with the next tile's
//   global loads in flight across the MFMA block.  bf16 operands use v_mfma_f32_32x32x16_bf16;
//   f32 operands use v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain — the 1e-3 parity path).
//   Any M, N; K % 32 == 0.  Out-of-range rows load zeros and are not stored.
//
// Both A and W are K-contiguous, so both MFMA operands are plain 16-byte row reads: lane
// (r = lane&31, h = lane>>5) holds A[row r][k-slice h] and W[row r][k-slice h].
#include "gemm_args.hpp"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 64;   // K bytes per row per step
constexpr int ROW_STRIDE = 80;  // padded LDS row
constexpr int TILE_BYTES = BM * ROW_STRIDE;
constexpr int EPI_SPLITK_SLAB = 100;  // internal: raw f32 accumulators to slab blockIdx.z of C

template <typename CT> __device__ __forceinline__ void store_c(void* C, int64_t off, float v) {
  static_cast<CT*>(C)[off] = from_f32<CT>(v);
}

template <typename CT, int EPI>
__device__ __forceinline__ void epilogue_elem(const GemmArgs& a, int64_t m, int n, float acc) {
  if (m >= a.M || n >= a.N) return;
  if constexpr (EPI == DFD_EPI_BIAS) {
    store_c<CT>(a.C, m * a.ldc + n, acc + (a.bias ? a.bias[n] : 0.f));
  } else if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) {
    store_c<CT>(a.C, m * a.ldc + n, quick_gelu(acc + (a.bias ? a.bias[n] : 0.f)));
  } else if constexpr (EPI == DFD_EPI_BIAS_RESIDUAL) {
    float* c = static_cast<float*>(a.C) + m * a.ldc + n;
    *c = *c + (acc + (a.bias ? a.bias[n] : 0.f));
  } else if constexpr (EPI == DFD_EPI_PATCH_EMBED) {
    const int P = a.tokens - 1;
    const int64_t frame = m / P;
    const int p = (int)(m - frame * P);
    float* c = static_cast<float*>(a.C);
    c[(frame * a.tokens + 1 + p) * a.ldc + n] = acc + a.pos[(int64_t)(1 + p) * a.N + n];
    if (p == 0) c[(frame * a.tokens) * a.ldc + n] = a.cls[n] + a.pos[n];
  } else if constexpr (EPI == DFD_EPI_RESIDUAL_POS) {
    const int64_t frame = m / (a.tokens - 1);
    const float p = a.pos ? a.pos[(frame % a.frames_per_clip) * a.N + n] : 0.f;
    CT* c = static_cast<CT*>(a.C) + m * a.ldc + n;
    const CT* rsd = a.residual ? static_cast<const CT*>(a.residual) + m * a.ldc + n : c;
    *c = from_f32<CT>(to_f32(*rsd) + dfd_drop_one(a.drop, (uint64_t)m * a.N + n, acc) + p);
  } else if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
    const float v = acc + (a.bias ? a.bias[n] : 0.f);
    store_c<CT>(a.C, m * a.ldc + n, v);
    const int D = a.N / (3 - a.qkv_first);
    const int ne = n + a.qkv_first * D;  // column in the full [q | k | v] numbering
    if (a.k_export != nullptr && ne >= D) {
      const int64_t frame = m / a.tokens;
      const int tok = (int)(m - frame * a.tokens);
      if (tok > 0) {
        const bool is_v = ne >= 2 * D;
        const int cc = ne - (is_v ? 2 * D : D);
        const float e = v + (a.pos ? a.pos[(frame % a.frames_per_clip) * D + cc] : 0.f);
        store_c<CT>(is_v ? a.v_export : a.k_export, (frame * (a.tokens - 1) + tok - 1) * D + cc, e);
      }
    }
  }
}

template <typename T, typename CT, int EPI>
__global__ __launch_bounds__(256) void gemm128_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];  // A0 A1 B0 B1
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  constexpr int KSTEP = ROW_BYTES / (int)sizeof(T);  // elements of K per step
  // split-K (gridDim.z > 1): slice z covers K steps [z*per, (z+1)*per) and writes its own f32 slab of C
  const int nk_all = a.K / KSTEP;
  const int per = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int kt0 = (int)blockIdx.z * per;
  const int nk = min(per, nk_all - kt0) > 0 ? min(per, nk_all - kt0) : 0;

  // staging assignment: 512 16-byte chunks per matrix tile, 2 per thread
  const unsigned char* Ab = static_cast<const unsigned char*>(a.A);
  const unsigned char* Wb = static_cast<const unsigned char*>(a.W);
  uint4 ra[2], rb[2];
  int srow[2], scol[2];
  bool va[2], vb[2];
  const unsigned char* pa[2];
  const unsigned char* pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    srow[i] = c >> 2;
    scol[i] = (c & 3) * 16;
    va[i] = (m0 + srow[i]) < a.M;
    vb[i] = (n0 + srow[i]) < a.N;
    pa[i] = Ab + ((m0 + srow[i]) * a.lda) * (int64_t)sizeof(T) + scol[i];
    pb[i] = Wb + ((int64_t)(n0 + srow[i]) * a.ldw) * (int64_t)sizeof(T) + scol[i];
  }
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i] = va[i] ? *reinterpret_cast<const uint4*>(pa[i] + (int64_t)(kt0 + kt) * ROW_BYTES) : uint4{0, 0, 0, 0};
      rb[i] = vb[i] ? *reinterpret_cast<const uint4*>(pb[i] + (int64_t)(kt0 + kt) * ROW_BYTES) : uint4{0, 0, 0, 0};
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<uint4*>(smem + buf * TILE_BYTES + srow[i] * ROW_STRIDE + scol[i]) = ra[i];
      *reinterpret_cast<uint4*>(smem + (2 + buf) * TILE_BYTES + srow[i] * ROW_STRIDE + scol[i]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const unsigned char* As = smem + buf * TILE_BYTES + (wm * 64 + r) * ROW_STRIDE;
    const unsigned char* Bs = smem + (2 + buf) * TILE_BYTES + (wn * 64 + r) * ROW_STRIDE;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ROW_STRIDE + s * 32 + h * 16);
          fb[i] = *reinterpret_cast<const bf16x8*>(Bs + i * 32 * ROW_STRIDE + s * 32 + h * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // f32: lane half h owns k-slice [8h, 8h+8) of the 16-wide step; MFMA step s pairs k = s and
      // k = 8 + s (any pairing is valid as long as A and W use the same one).
      f32x4 fa[2][2], fb[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          fa[i][q] = *reinterpret_cast<const f32x4*>(As + i * 32 * ROW_STRIDE + h * 32 + q * 16);
          fb[i][q] = *reinterpret_cast<const f32x4*>(Bs + i * 32 * ROW_STRIDE + h * 32 + q * 16);
        }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s >> 2][s & 3], fb[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int n = n0 + wn * 64 + j * 32 + r;
        if constexpr (EPI == EPI_SPLITK_SLAB) {
          if (m < a.M && n < a.N) static_cast<float*>(a.C)[((int64_t)blockIdx.z * a.M + m) * a.ldc + n] = acc[i][j][e];
        } else {
          epilogue_elem<CT, EPI>(a, m, n, acc[i][j][e]);
        }
      }
}

template <typename T, typename CT>
int launch_gemm128(const GemmArgs& a, int epi, hipStream_t st) {
  const dim3 grid((a.N + BN - 1) / BN, (unsigned)((a.M + BM - 1) / BM)), block(256);
#define EPI_CASE(E)                                                                    \
  case E:                                                                              \
    hipLaunchKernelGGL((gemm128_kernel<T, CT, E>), grid, block, 0, st, a);             \
    break;
  switch (epi) {
    EPI_CASE(DFD_EPI_BIAS)
    EPI_CASE(DFD_EPI_BIAS_QUICKGELU)
    EPI_CASE(DFD_EPI_QKV_EXPORT)
    EPI_CASE(DFD_EPI_RESIDUAL_POS)
    default:
      if constexpr (sizeof(CT) == 4) {
        switch (epi) {
          EPI_CASE(DFD_EPI_BIAS_RESIDUAL)
          EPI_CASE(DFD_EPI_PATCH_EMBED)
          default:
            dfd_set_error("dfd_gemm: unknown epilogue %d", epi);
            return DFD_ERR_INVALID_ARG;
        }
      } else {
        dfd_set_error("dfd_gemm: epilogue %d needs an f32 C", epi);
        return DFD_ERR_INVALID_ARG;
      }
  }
#undef EPI_CASE
  DFD_CHECK_LAUNCH("dfd_gemm");
  return DFD_OK;
}

// ---- C = Aᵀ·B for tall operands (weight gradients): transposes + split-K on the kernel above ----------
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst, int64_t R,
                                                            int C, int64_t Rp) {
  __shared__ T tile[32][33];
  const int c0 = blockIdx.x * 32;
  const int64_t r0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) tile[i][tx] = (r0 + i < R && c0 + tx < C) ? src[(r0 + i) * lds_ + c0 + tx] : from_f32<T>(0.f);
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < Rp) dst[(int64_t)(c0 + i) * Rp + r0 + tx] = tile[tx][i];
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n, int splits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * n + i];
  out[i] = s;
}

void at_b_plan(int64_t R, int Ma, int Nb, int64_t* Rp, int* splits) {
  const int tiles = ((Ma + BM - 1) / BM) * ((Nb + BN - 1) / BN);
  int sp = (1024 + tiles - 1) / tiles;                       // ~4 workgroups per CU in flight
  const int64_t steps = (R + 31) / 32;
  if (sp > steps / 8) sp = (int)(steps / 8 > 0 ? steps / 8 : 1);  // at least 8 K steps per slice
  if (sp > 512) sp = 512;
  *splits = sp;
  *Rp = (R + 31) / 32 * 32;
}

template <typename T>
int launch_at_b(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t R, int Ma, int Nb, void* workspace,
                hipStream_t st) {
  int64_t Rp;
  int splits;
  at_b_plan(R, Ma, Nb, &Rp, &splits);
  T* At = static_cast<T*>(workspace);
  T* Bt = At + (int64_t)Ma * Rp;
  size_t off = (size_t)(Ma + Nb) * Rp * sizeof(T);
  off = (off + 255) / 256 * 256;
  float* slabs = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + off);
  const dim3 tb(256);
  hipLaunchKernelGGL((transpose_pad_kernel<T>), dim3((Ma + 31) / 32, (unsigned)(Rp / 32)), tb, 0, st, static_cast<const T*>(A), lda, At, R, Ma, Rp);
  hipLaunchKernelGGL((transpose_pad_kernel<T>), dim3((Nb + 31) / 32, (unsigned)(Rp / 32)), tb, 0, st, static_cast<const T*>(B), ldb, Bt, R, Nb, Rp);
  GemmArgs a{};
  a.A = At; a.W = Bt; a.C = slabs; a.lda = Rp; a.ldw = Rp; a.ldc = Nb; a.M = Ma; a.N = Nb; a.K = (int)Rp;
  const dim3 grid((Nb + BN - 1) / BN, (Ma + BM - 1) / BM, splits);
  hipLaunchKernelGGL((gemm128_kernel<T, float, EPI_SPLITK_SLAB>), grid, tb, 0, st, a);
  const int64_t n = (int64_t)Ma * Nb;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), tb, 0, st, slabs, C, n, splits);
  DFD_CHECK_LAUNCH("dfd_gemm_at_b");
  return DFD_OK;
}

}  // namespace

// epilogue-specific argument checks + copy of dfd_gemm_extra into the kernels' argument block
static int fill_extra(GemmArgs& a, int epilogue, const dfd_gemm_extra* extra, int c_dtype, int64_t ldc, int64_t M, int N) {
  if (epilogue == DFD_EPI_PATCH_EMBED) {
    DFD_REQUIRE(extra && extra->pos && extra->cls && extra->tokens > 1, "dfd_gemm: PATCH_EMBED needs extra.pos, extra.cls, extra.tokens");
    DFD_REQUIRE(M % (extra->tokens - 1) == 0, "dfd_gemm: PATCH_EMBED M=%lld is not a whole number of frames", (long long)M);
    DFD_REQUIRE(c_dtype == DFD_F32 && ldc >= N, "dfd_gemm: PATCH_EMBED writes an f32 token matrix");
  } else {
    DFD_REQUIRE(ldc >= N, "dfd_gemm: ldc=%lld < N", (long long)ldc);
  }
  if (epilogue == DFD_EPI_QKV_EXPORT) {
    DFD_REQUIRE(extra && extra->tokens > 1 && (extra->qkv_first == 0 || extra->qkv_first == 1) && N % (3 - extra->qkv_first) == 0,
                "dfd_gemm: QKV_EXPORT needs extra.tokens, qkv_first in {0, 1} and N a multiple of the column blocks present");
    DFD_REQUIRE(M % extra->tokens == 0, "dfd_gemm: QKV_EXPORT M=%lld is not a whole number of frames", (long long)M);
    DFD_REQUIRE(!extra->k_export == !extra->v_export, "dfd_gemm: QKV_EXPORT needs both k_export and v_export or neither");
    DFD_REQUIRE(!extra->pos || extra->frames_per_clip > 0, "dfd_gemm: QKV_EXPORT with pos needs frames_per_clip");
  }
  if (epilogue == DFD_EPI_RESIDUAL_POS) {
    DFD_REQUIRE(extra && extra->tokens > 1, "dfd_gemm: RESIDUAL_POS needs extra.tokens (patches per frame + 1)");
    DFD_REQUIRE(!extra->pos || extra->frames_per_clip > 0, "dfd_gemm: RESIDUAL_POS with pos needs frames_per_clip");
  }
  if (extra) {
    a.pos = extra->pos; a.cls = extra->cls; a.k_export = extra->k_export; a.v_export = extra->v_export;
    a.residual = epilogue == DFD_EPI_RESIDUAL_POS ? extra->residual : nullptr;
    a.tokens = extra->tokens; a.frames_per_clip = extra->frames_per_clip > 0 ? extra->frames_per_clip : 1;
    a.qkv_first = epilogue == DFD_EPI_QKV_EXPORT ? extra->qkv_first : 0;
    a.stream_out = (extra->flags & DFD_GEMM_STREAM_OUT) ? 1 : 0;
    a.spare_cus = (int)((extra->flags >> DFD_GEMM_SPARE_CUS_SHIFT) & 0xff);
    a.spare_if_free = (extra->flags & DFD_GEMM_SPARE_IF_FREE) ? 1 : 0;
    a.tile_rows = 32 * (int)((extra->flags >> DFD_GEMM_TILE_BLOCKS_SHIFT) & 0xf);
    DFD_REQUIRE(a.tile_rows == 0 || a.tile_rows == 224 || a.tile_rows == 256, "dfd_gemm: tile blocks must be 0, 7 or 8");
    if (epilogue == DFD_EPI_RESIDUAL_POS && extra->drop_rng && extra->drop_p > 0.f) {
      DFD_REQUIRE(extra->drop_p < 1.f, "dfd_gemm: drop_p=%f", (double)extra->drop_p);
      const dfd_dropout_t dd{extra->drop_rng, extra->drop_site, extra->drop_p};
      a.drop = dfd_make_drop(&dd);
    }
  }
  return DFD_OK;
}

static thread_local int g_last_path = 0;
static thread_local int g_gemm_variant = 0;
extern "C" int dfd_gemm_last_path(void) { return g_last_path; }
extern "C" int dfd_gemm_set_variant(int variant) {
  const int old = g_gemm_variant;
  g_gemm_variant = variant;
  return old;
}

extern "C" int dfd_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, int ab_dtype, void* C, int64_t ldc,
                        int c_dtype, const float* bias, int epilogue, const dfd_gemm_extra* extra, int64_t M, int N,
                        int K, void* stream) {
  DFD_REQUIRE(A && W && C, "dfd_gemm: null pointer");
  DFD_REQUIRE(M >= 0 && N > 0 && K > 0, "dfd_gemm: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  DFD_REQUIRE(ab_dtype == DFD_F32 || ab_dtype == DFD_BF16, "dfd_gemm: ab_dtype=%d", ab_dtype);
  DFD_REQUIRE(c_dtype == DFD_F32 || c_dtype == DFD_BF16, "dfd_gemm: c_dtype=%d", c_dtype);
  DFD_REQUIRE(!(ab_dtype == DFD_F32 && c_dtype == DFD_BF16), "dfd_gemm: f32 operands with bf16 output unsupported");
  DFD_REQUIRE(K % 32 == 0, "dfd_gemm: K=%d must be a multiple of 32", K);
  const int esz = ab_dtype == DFD_F32 ? 4 : 2;
  DFD_REQUIRE(lda >= K && ldw >= K && (lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "dfd_gemm: lda=%lld ldw=%lld must be >= K and 16-byte multiples", (long long)lda, (long long)ldw);
  DFD_REQUIRE(dfd_aligned16(A) && dfd_aligned16(W), "dfd_gemm: A and W must be 16-byte aligned");
  GemmArgs a{};
  a.A = A; a.W = W; a.C = C; a.bias = bias;
  a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  { const int rc = fill_extra(a, epilogue, extra, c_dtype, ldc, M, N); if (rc != DFD_OK) return rc; }
  if (M == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ab_dtype == DFD_BF16) {
    // a q|k|v projection without an export is a plain biased store
    const int epi_p = epilogue == DFD_EPI_QKV_EXPORT && a.k_export == nullptr ? DFD_EPI_BIAS : epilogue;
    a.no_dynamic = g_gemm_variant != 3;
    int rc = g_gemm_variant == 1 ? 1 : dfd_gemm256e_try(a, c_dtype, epi_p, st);  // ping-pong K loop (K a multiple of 128, >= 384)
    if (rc == 0) g_last_path = 257;
    if (rc <= 0) return rc;
    rc = dfd_gemm256p_try(a, c_dtype, epi_p, st);
    if (rc == 0) g_last_path = 256;
    if (rc <= 0) return rc;
    rc = dfd_gemm256_try(a, c_dtype, epilogue, st);
    if (rc == 0) g_last_path = 256;
    if (rc <= 0) return rc;
  }
  g_last_path = 128;
  if (ab_dtype == DFD_F32) return launch_gemm128<float, float>(a, epilogue, st);
  if (c_dtype == DFD_BF16) return launch_gemm128<bf16_t, bf16_t>(a, epilogue, st);
  return launch_gemm128<bf16_t, float>(a, epilogue, st);
}

extern "C" int dfd_gemm_fp8(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int c_dtype,
                            const float* col_scale, const float* bias, float out_inv_scale, int epilogue,
                            const dfd_gemm_extra* extra, int64_t M, int N, int K, void* stream) {
  DFD_REQUIRE(A && W && C && col_scale, "dfd_gemm_fp8: null pointer");
  DFD_REQUIRE(M >= 0 && N > 0 && K > 0, "dfd_gemm_fp8: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  DFD_REQUIRE(c_dtype == DFD_BF16 || c_dtype == DFD_FP8, "dfd_gemm_fp8: c_dtype=%d", c_dtype);
  DFD_REQUIRE(epilogue == DFD_EPI_BIAS || epilogue == DFD_EPI_BIAS_QUICKGELU || epilogue == DFD_EPI_QKV_EXPORT,
              "dfd_gemm_fp8: epilogue %d not served (BIAS, BIAS_QUICKGELU, QKV_EXPORT)", epilogue);
  DFD_REQUIRE(c_dtype == DFD_BF16 || (epilogue != DFD_EPI_QKV_EXPORT && out_inv_scale > 0.f), "dfd_gemm_fp8: fp8 output needs out_inv_scale > 0 and a plain / QuickGELU epilogue");
  DFD_REQUIRE(lda >= K && ldw >= K && ldc >= N, "dfd_gemm_fp8: leading dimensions");
  GemmArgs a{};
  a.A = A; a.W = W; a.C = C; a.bias = bias; a.col_scale = col_scale; a.out_inv_scale = out_inv_scale;
  a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  { const int rc = fill_extra(a, epilogue, extra, c_dtype, ldc, M, N); if (rc != DFD_OK) return rc; }
  if (M == 0) return DFD_OK;
  const int epi_p = epilogue == DFD_EPI_QKV_EXPORT && a.k_export == nullptr ? DFD_EPI_BIAS : epilogue;
  a.no_dynamic = g_gemm_variant != 3;
  int rc = g_gemm_variant == 1 ? 1 : dfd_gemm256e_f8_try(a, c_dtype, epi_p, static_cast<hipStream_t>(stream));
  if (rc == 1) rc = dfd_gemm256p_f8_try(a, c_dtype, epi_p, static_cast<hipStream_t>(stream));
  if (rc == 1) {
    dfd_set_error("dfd_gemm_fp8: shape not served (needs M >= 1024, N %% 256 == 0, K %% 128 == 0, K >= 256, 16-byte aligned rows; got M=%lld N=%d K=%d)",
                  (long long)M, N, K);
    return DFD_ERR_INVALID_ARG;
  }
  return rc;
}

// gemm_tn.hip: bf16 operands read through the transposing LDS load, no HBM transposes
bool dfd_gemm_tn_supported(const void* A, int64_t lda, const void* B, int64_t ldb, int Ma, int Nb);
int dfd_gemm_tn_splits(int64_t R, int Ma, int Nb);
int dfd_gemm_tn_launch(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t R, int Ma, int Nb,
                       float* slabs, hipStream_t st);

extern "C" size_t dfd_gemm_at_b_workspace(int64_t R, int Ma, int Nb, int dtype) {
  if (R <= 0 || Ma <= 0 || Nb <= 0) return 0;
  int64_t Rp;
  int splits;
  at_b_plan(R, Ma, Nb, &Rp, &splits);
  const size_t esz = dtype == DFD_F32 ? 4 : 2;
  size_t bytes = (size_t)(Ma + Nb) * Rp * esz;
  bytes = (bytes + 255) / 256 * 256;
  bytes += (size_t)splits * Ma * Nb * sizeof(float) + 256;
  if (dtype == DFD_BF16 && Ma % 128 == 0 && Nb % 128 == 0) {
    const size_t tn = (size_t)dfd_gemm_tn_splits(R, Ma, Nb) * Ma * Nb * sizeof(float) + 256;
    if (tn > bytes) bytes = tn;
  }
  return bytes;
}

extern "C" int dfd_gemm_at_b(const void* A, int64_t lda, const void* B, int64_t ldb, int dtype, float* C, int64_t R, int Ma,
                             int Nb, void* workspace, void* stream) {
  DFD_REQUIRE(A && B && C && workspace, "dfd_gemm_at_b: null pointer");
  DFD_REQUIRE(R > 0 && Ma > 0 && Nb > 0 && lda >= Ma && ldb >= Nb, "dfd_gemm_at_b: bad shape");
  DFD_REQUIRE(R < (int64_t)1 << 31, "dfd_gemm_at_b: R too large");
  DFD_REQUIRE(dtype == DFD_F32 || dtype == DFD_BF16, "dfd_gemm_at_b: dtype=%d", dtype);
  DFD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "dfd_gemm_at_b: workspace must be 256-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == DFD_BF16 && dfd_gemm_tn_supported(A, lda, B, ldb, Ma, Nb))
    return dfd_gemm_tn_launch(A, lda, B, ldb, C, R, Ma, Nb, static_cast<float*>(workspace), st);
  if (dtype == DFD_F32) return launch_at_b<float>(A, lda, B, ldb, C, R, Ma, Nb, workspace, st);
  return launch_at_b<bf16_t>(A, lda, B, ldb, C, R, Ma, Nb, workspace, st);
}
