// C[Ma, Nb] (f32) = Aᵀ·B for tall row-major bf16 operands A [R, Ma], B [R, Nb]: the weight gradient
// of a Linear applied to R = B·T·patches rows (adapter training, reference models.py:783-940 under
// trainer.py:157-165).  The contraction index is the ROW index of both operands, so neither is
// K-contiguous; instead of transposing them in HBM the tiles are staged row-major in LDS by
// global_load_lds and read back through gfx950's transposing LDS read (ds_read_b64_tr_b16), which
// hands each lane 4 consecutive contraction rows of one column — the v_mfma_f32_16x16x32_bf16 operand
// layout (2 reads per 16x32 fragment).  A and B use the same row→k mapping, so any consistent order
// of the contraction is fine.
//
// Tile TT(m) x TT(n) x 64 rows per step, TT = 128 or 256.  128: 4 waves (2x2) of 64x64 (64 accumulator registers), two
// 32 KB LDS slots, 2 workgroups per CU.  256 (both extents multiples of 256: the adapter's 768 x 256 and 256 x 768):
// 8 waves (4x2) of 64x128 (128 accumulator registers; sixteen 64x64 waves would have 128 registers each and spill), two
// 64 KB slots, one workgroup per CU — every operand row is then read Nb/256 resp. Ma/256 times instead of Nb/128,
// Ma/128, and half as many f32 slabs are written and reduced per output element.
// LDS image: plain 2*TT-byte rows with the 16-byte chunk
// XOR of cdna_hip_programming.md T10(b), off(row, ch) = 2*TT*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3))) (the XOR
// touches the low four chunk bits only, so it stays inside a 256-byte half of a wide row),
// applied on the global source address (LDS-DMA destinations are lane-linear): conflict-free for the
// transposed reads.  Split over R: each grid.z slice accumulates its rows into its own f32 slab, a
// fixed-order slab reduce follows (deterministic).  Rows past R come from a 16-byte zero constant.
//
// Bytes: every A row is read Nb/TT times and every B row Ma/TT times (L2/MALL absorbs most).
#include "common.hpp"

namespace {

constexpr int TR = 64;             // contraction rows per step
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef short short4v __attribute__((ext_vector_type(4)));

__device__ const uint4 kZero16 = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

struct TnArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* slabs;
  int64_t lda, ldb, R;
  int Ma, Nb, steps_per_split;
};

// one operand tile: 64 rows x TT columns starting at (r0, c0); wave w stages rows (64/NW)w .. in pieces of 1 KB
// (1024 / ROWB rows: a lane moves 16 bytes)
template <int TT>
__device__ __forceinline__ void stage(const bf16_t* __restrict__ src, int64_t ld, int64_t R, int64_t r0, int c0,
                                      unsigned char* lds_tile, int wave, int lane) {
  constexpr int ROWB = TT * 2, NW = TT == 128 ? 4 : 8, RPW = TR / NW, CPR = ROWB / 16, RPI = 64 / CPR;
#pragma unroll
  for (int i = 0; i < RPW / RPI; ++i) {
    const int row = RPW * wave + RPI * i + lane / CPR;
    const int cpos = lane % CPR;
    const int ch = cpos ^ swz(row);
    const int64_t gr = r0 + row;
    const void* g = gr < R ? static_cast<const void*>(src + gr * ld + c0 + 8 * ch) : static_cast<const void*>(&kZero16);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(lds_tile + (RPW * wave + RPI * i) * ROWB), 16, 0, 0);
  }
}

// 16(col) x 32(row) operand fragment of the tile at columns col0.., rows k0..k0+31
template <int ROWB>
__device__ __forceinline__ bf16x8 frag(const unsigned char* lds_tile, int col0, int k0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int c0 = col0 >> 3;  // first 16-byte chunk of the 16 columns
  const int r1 = k0 + 8 * g + q, r2 = r1 + 4;
  const unsigned char* a1 = lds_tile + ROWB * r1 + 16 * ((c0 + (p >> 1)) ^ swz(r1)) + 8 * (p & 1);
  const unsigned char* a2 = lds_tile + ROWB * r2 + 16 * ((c0 + (p >> 1)) ^ swz(r2)) + 8 * (p & 1);
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)a1);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)a2);
  short8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
  return __builtin_bit_cast(bf16x8, v);
}

template <int TT>
__global__ __launch_bounds__(TT == 128 ? 256 : 512, TT == 128 ? 2 : 1) void gemm_tn_bf16_kernel(TnArgs a) {
  constexpr int ROWB = TT * 2;       // bytes per LDS row
  constexpr int HALF = TR * ROWB;    // one operand's tile
  constexpr int SLOT = 2 * HALF;     // A tile | B tile
  constexpr int WN = 2;              // waves along n (TT/64 along m)
  constexpr int WTN = TT / WN;       // columns of a wave's tile: 64 or 128
  constexpr int NJ = WTN / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 slots x (A tile | B tile)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * TT, n0 = blockIdx.x * TT;
  const int64_t total_steps = (a.R + TR - 1) / TR;
  const int64_t s_begin = (int64_t)blockIdx.z * a.steps_per_split;
  int64_t s_end = s_begin + a.steps_per_split;
  if (s_end > total_steps) s_end = total_steps;

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    stage<TT>(a.A, a.lda, a.R, s_begin * TR, m0, lds, wave, lane);
    stage<TT>(a.B, a.ldb, a.R, s_begin * TR, n0, lds + HALF, wave, lane);
  }
  for (int64_t s = s_begin; s < s_end; ++s) {
    const int cur = (int)((s - s_begin) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the current slot have landed
    __syncthreads();                     // ... everyone's have, and everyone is done reading the other slot
    if (s + 1 < s_end) {
      unsigned char* nxt = lds + (cur ^ 1) * SLOT;
      stage<TT>(a.A, a.lda, a.R, (s + 1) * TR, m0, nxt, wave, lane);
      stage<TT>(a.B, a.ldb, a.R, (s + 1) * TR, n0, nxt + HALF, wave, lane);
    }
    const unsigned char* At = lds + cur * SLOT;
    const unsigned char* Bt = At + HALF;
    // 256-wide tiles: the 48 fragment addresses of a step are rebuilt from an opaque copy of the lane id — hoisted out of
    // the loop (they are loop-invariant) they would not fit beside 128 accumulator registers
    int lq = lane;
    if (TT == 256) asm volatile("" : "+v"(lq));
#pragma unroll
    for (int kk = 0; kk < TR; kk += 32) {
      bf16x8 fa[4], fb[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag<ROWB>(At, 64 * wm + 16 * i, kk, lq);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = frag<ROWB>(Bt, WTN * wn + 16 * j, kk, lq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  // C fragment: lane holds rows 4*(lane/16)+e of the A-side index, column lane%16 of the B-side index
  float* slab = a.slabs + (int64_t)blockIdx.z * a.Ma * a.Nb;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 64 * wm + 16 * i + 4 * (lane >> 4) + e;
        const int n = n0 + WTN * wn + 16 * j + (lane & 15);
        slab[(int64_t)m * a.Nb + n] = acc[i][j][e];
      }
}

// out[i] = Σ_z slabs[z][i] in a FIXED order: lane group g (of ZG) sums the slabs z ≡ g (mod ZG) in ascending z, the
// groups are then added in ascending g.  (One thread walking all slabs was latency-bound: 85 dependent loads, 23 us for
// 67 MB; four independent walks per output quadruple the loads in flight.)
constexpr int ZG = 4;
__global__ __launch_bounds__(64 * ZG) void tn_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n, int splits) {
  __shared__ f32x4 part[ZG][64];
  const int t = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = ((int64_t)blockIdx.x * 64 + t) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    for (int z = g; z < splits; z += ZG) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * n + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
  }
  part[g][t] = s;
  __syncthreads();
  if (g == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < ZG; ++k) {
      const f32x4 v = part[k][t];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
    *reinterpret_cast<f32x4*>(out + i) = s;
  }
}

}  // namespace

// plan shared with dfd_gemm_at_b_workspace (gemm.hip)
bool dfd_gemm_tn_supported(const void* A, int64_t lda, const void* B, int64_t ldb, int Ma, int Nb) {
  return Ma % 128 == 0 && Nb % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0 && dfd_aligned16(A) && dfd_aligned16(B);
}

static int tn_tile(int Ma, int Nb) { return (Ma % 256 == 0 && Nb % 256 == 0) ? 256 : 128; }

int dfd_gemm_tn_splits(int64_t R, int Ma, int Nb) {
  const int TT = tn_tile(Ma, Nb);
  const int tiles = (Ma / TT) * (Nb / TT);
  const int64_t steps = (R + TR - 1) / TR;
  int sp = ((TT == 128 ? 512 : 256) + tiles - 1) / tiles;  // 128-wide tiles: ~2 workgroups per CU; 256-wide: one
  if (sp > steps / 4) sp = (int)(steps / 4 > 0 ? steps / 4 : 1);
  if (sp > 256) sp = 256;
  return sp;
}

int dfd_gemm_tn_launch(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t R, int Ma, int Nb,
                       float* slabs, hipStream_t st) {
  TnArgs a;
  a.A = static_cast<const bf16_t*>(A);
  a.B = static_cast<const bf16_t*>(B);
  a.slabs = slabs;
  a.lda = lda;
  a.ldb = ldb;
  a.R = R;
  a.Ma = Ma;
  a.Nb = Nb;
  const int splits = dfd_gemm_tn_splits(R, Ma, Nb);
  const int64_t steps = (R + TR - 1) / TR;
  a.steps_per_split = (int)((steps + splits - 1) / splits);
  if (tn_tile(Ma, Nb) == 256) {
    constexpr int LDS = 2 * 2 * TR * 512;  // two slots of (A tile | B tile), 512-byte rows
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(gemm_tn_bf16_kernel<256>, dim3(Nb / 256, Ma / 256, splits), dim3(512), LDS, st, a);
  } else {
    constexpr int LDS = 2 * 2 * TR * 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_bf16_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(gemm_tn_bf16_kernel<128>, dim3(Nb / 128, Ma / 128, splits), dim3(256), LDS, st, a);
  }
  DFD_CHECK_LAUNCH("dfd_gemm_at_b(tn)");
  const int64_t n = (int64_t)Ma * Nb;
  hipLaunchKernelGGL(tn_slab_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(64 * ZG), 0, st, slabs, C, n, splits);
  DFD_CHECK_LAUNCH("dfd_gemm_at_b(reduce)");
  return DFD_OK;
}
