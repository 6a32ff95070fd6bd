// bf16 MFMA self-attention for the encoder: out = softmax(q kᵀ · scale) v per (frame, head),
// head_dim 64, tokens <= 32*NB (197 -> NB = 7, 257 -> NB = 9) (reference clip/model.py:188-195).
//
// One 256-thread workgroup (4 waves, one per SIMD; two workgroups per CU) per (frame, head):
//   * K [keys][64] goes to LDS row-major with the 16-byte chunk c of row r at position
//     c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 for the 32-row MFMA operand); V goes to LDS
//     TRANSPOSED, Vt[d][key] with a (2*keys + 8)-byte row stride (conflict-free ds_read_b64); pad keys
//     are zero-filled;
//   * a wave owns 32-query blocks.  Sᵀ = K·Qᵀ is formed with v_mfma_f32_32x32x16_bf16 (K fragment
//     as the A operand, Q fragment straight from global memory as B), so a lane holds one query
//     column and all of its 32*NB scores stay in registers: the softmax is a plain max / exp2 /
//     sum over registers plus ONE cross-lane exchange (lane ^ 32) — no online rescaling, and the
//     [N, tokens, tokens, heads] affinity tensor of the reference is never materialised;
//   * the exponentiated scores, converted pairwise to bf16, are already the B operand of the
//     second product Oᵀ = Vᵀ·Pᵀ (accumulator-as-operand, k order permuted the same way on the
//     Vt fragment), so P never touches LDS; a lane ends with 4 consecutive output channels of
//     its query per register group and stores them as 8-byte pieces.
#include "common.hpp"
#include <stdlib.h>

#ifndef ATTN_STAMPS
#define ATTN_STAMPS 0  // diagnostic build: wave 0 of each workgroup writes cycle stamps to `out`-adjacent debug memory
#endif

namespace {

constexpr int HD = 64;
#if ATTN_STAMPS
__device__ float* g_attn_dbg = nullptr;
__device__ __forceinline__ unsigned long long astamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define ASTAMP(v) const unsigned long long v = astamp()
#define PSTAMP(v) const unsigned long long v = astamp()
#else
#define ASTAMP(v)
#define PSTAMP(v)
#endif

typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int NB> struct Lds {
  static constexpr int KEYS = NB * 32;
  // bytes per Vt row.  32 lanes read rows r = 0..31 at one key offset with ds_read_b64 (banks mod 64
  // dwords): r * (VSTRIDE/4) must hit 32 distinct even banks, i.e. VSTRIDE/4 = 2 * odd
  static constexpr int VSTRIDE = KEYS * 2 + 8;
  static constexpr int K_BYTES = KEYS * 128;
  static constexpr int BYTES = K_BYTES + HD * VSTRIDE;
};

template <int NB>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const bf16_t* __restrict__ qkv, int64_t ld_qkv,
                                                           bf16_t* __restrict__ out, int64_t ld_out, int tokens,
                                                           int heads, float scale_log2e) {
  using L = Lds<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vt = smem + L::K_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int frame = blockIdx.x / heads, head = blockIdx.x % heads;
  const int D = heads * HD;
  const bf16_t* base = qkv + (int64_t)frame * tokens * ld_qkv + head * HD;

  // ---- stage K (swizzled rows) and V (transposed); 8 lanes cover one 128-byte row ------------------
  // All of the workgroup's 2*NB global loads per thread are issued before the first LDS write: with
  // only two workgroups per CU, serialising load -> write per iteration exposes NB HBM latencies.
  constexpr int IT = L::KEYS * 8 / 256;  // = NB
  bf16x8 kreg[IT], vreg[IT];
  ASTAMP(t0);
  // Q fragments of the wave's first 32-query block travel with the K/V staging loads (their latency used to
  // be exposed in front of every block: 14 % of the kernel)
  auto load_q = [&](bf16x8 (&qf)[4], int qb) {
    const int qq = qb * 32 + r;
    const int qc = qq < tokens ? qq : tokens - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qc * ld_qkv + s * 16 + h * 8);
  };
  bf16x8 qf[4], qn[4];
  load_q(qf, wave);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = tid + it * 256;
    const int key = c >> 3, ch = c & 7;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    kreg[it] = z;
    vreg[it] = z;
    if (key < tokens) {
      const bf16_t* src = base + (int64_t)key * ld_qkv + ch * 8;
      kreg[it] = *reinterpret_cast<const bf16x8*>(src + D);
      vreg[it] = *reinterpret_cast<const bf16x8*>(src + 2 * D);
    }
  }
#if ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  ASTAMP(t1);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = tid + it * 256;
    const int key = c >> 3, ch = c & 7;
    *reinterpret_cast<bf16x8*>(Ks + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[it];
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16_t*>(Vt + (ch * 8 + e) * L::VSTRIDE + key * 2) = vreg[it][e];
  }
  ASTAMP(t2);
  __syncthreads();
  ASTAMP(t3);
#if ATTN_STAMPS
  unsigned long long a_q = 0, a_qk = 0, a_sm = 0, a_pv = 0, a_st = 0;
#endif

  const int ksw = (r >> 1) & 7;
  for (int qb = wave; qb < NB; qb += 4) {
    ASTAMP(s0);
    const int q = qb * 32 + r;

#if ATTN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    ASTAMP(s1);
    // ---- Sᵀ[key][q] for all NB key blocks ---------------------------------------------------------
    // K fragments are double-buffered: the reads for d-slice s+1 are issued before the MFMAs of slice s,
    // so the matrix pipe never waits for an LDS read issued in the same group.
    f32x16 S[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
    bf16x8 kfa[NB], kfb[NB];
    auto read_k = [&](bf16x8 (&kf)[NB], int s) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
        kf[kb] = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * 128 + (((2 * s + h) ^ ksw) << 4));
    };
    auto mma_k = [&](const bf16x8 (&kf)[NB], int s) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb], qf[s], S[kb], 0, 0, 0);
    };
    read_k(kfa, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_k(kfb, 1);
    mma_k(kfa, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_k(kfa, 2);
    mma_k(kfb, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_k(kfb, 3);
    mma_k(kfa, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_k(kfb, 3);
    __builtin_amdgcn_sched_barrier(0);

    ASTAMP(s2);
    // ---- softmax over the key axis: registers + one exchange with lane ^ 32 -----------------------
    // only the last key block can hold keys >= tokens (the launcher picks NB = ceil(tokens / 32))
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (key >= tokens) S[NB - 1][e] = -INFINITY;
    }
    // The softmax is the VALU-bound part of the kernel (two waves per SIMD): v_max3 without the NaN
    // canonicalisation fmaxf() drags in, packed f32 multiply-add and packed row sums.
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) mx = vmax3(mx, S[kb][e], S[kb][e + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sl2 = scale_log2e;
    float nmc = -mx * scale_log2e;
    asm volatile("" : "+s"(sl2));  // opaque scalars: the vector expression below packs into v_pk_fma_f32
    f32x16 lv;
#pragma unroll
    for (int e = 0; e < 16; ++e) lv[e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2v t = __builtin_elementwise_fma(f32x2v{S[kb][e], S[kb][e + 1]}, f32x2v{sl2, sl2}, f32x2v{nmc, nmc});
        S[kb][e] = __builtin_amdgcn_exp2f(t[0]);
        S[kb][e + 1] = __builtin_amdgcn_exp2f(t[1]);
      }
      lv += S[kb];
    }
    float l = ((lv[0] + lv[1]) + (lv[2] + lv[3])) + ((lv[4] + lv[5]) + (lv[6] + lv[7])) +
              (((lv[8] + lv[9]) + (lv[10] + lv[11])) + ((lv[12] + lv[13]) + (lv[14] + lv[15])));
    l += __shfl_xor(l, 32, 64);

    if (qb + 4 < NB) load_q(qn, qb + 4);  // next block's Q fragments fly during the PV product
    ASTAMP(s3);
    // ---- Oᵀ[d][q] = Σ_key Vt[d][key] · Pᵀ[key][q]; P fragments come straight from the S registers ------
    // Vt fragments (element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3) of the block: two runs of 4
    // keys) are read one (block, slice) step ahead of the MFMAs that use them.
    f32x16 O[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
    auto read_v = [&](bf16x8 (&vf)[2], int step) {
      const int kb = step >> 1, sl = step & 1;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned char* vp = Vt + (dt * 32 + r) * L::VSTRIDE + (kb * 32 + 16 * sl + 4 * h) * 2;
        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vp);
        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vp + 16);
        vf[dt] = bf16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      }
    };
    auto mma_v = [&](const bf16x8 (&vf)[2], int step) {
      const int kb = step >> 1, sl = step & 1;
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)S[kb][8 * sl + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pf, O[dt], 0, 0, 0);
    };
    bf16x8 vfa[2], vfb[2];
    read_v(vfa, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int step = 0; step < 2 * NB; step += 2) {
      read_v(vfb, step + 1);
      mma_v(vfa, step);
      __builtin_amdgcn_sched_barrier(0);
      if (step + 2 < 2 * NB) read_v(vfa, step + 2);
      mma_v(vfb, step + 1);
      __builtin_amdgcn_sched_barrier(0);
    }

    ASTAMP(s4);
    if (q < tokens) {
      const float inv = 1.0f / l;
      bf16_t* op = out + ((int64_t)frame * tokens + q) * ld_out + head * HD;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(O[dt][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * h) = o;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = qn[s];
#if ATTN_STAMPS
    {
      const unsigned long long s5 = astamp();
      a_q += s1 - s0; a_qk += s2 - s1; a_sm += s3 - s2; a_pv += s4 - s3; a_st += s5 - s4;
    }
#endif
  }
#if ATTN_STAMPS
  if (tid == 0 && g_attn_dbg) {
    float* d = g_attn_dbg + (int64_t)blockIdx.x * 12;
    const unsigned long long te = astamp();
    d[0] = (float)(t1 - t0); d[1] = (float)(t2 - t1); d[2] = (float)(t3 - t2);
    d[3] = (float)a_q; d[4] = (float)a_qk; d[5] = (float)a_sm; d[6] = (float)a_pv; d[7] = (float)a_st; d[8] = (float)(te - t0);
  }
#endif
}

// ---- persistent variant (NB <= 8) --------------------------------------------------------------------
// One 512-thread workgroup per CU walks (frame, head) items.  K and V of an item go to LDS by LDS-DMA
// (global_load_lds, no register round trip, swizzle applied on the source address; rows of pad keys come
// from a 16-byte zero constant), double-buffered: item i+1 is in flight while item i is computed, so
// the HBM latency that the 4-wave kernel above pays in front of every item (a quarter of its time) is
// hidden and there is one workgroup barrier per item.  Wave w owns query block w (one wave idles when
// NB = 7).  V stays row-major [key][64] and its MFMA operand (d rows, 8 keys per lane: two runs of 4
// keys) is read with the transposing LDS read ds_read_b64_tr_b16 — no transposed 2-byte stores.
// LDS image of V: 128-byte rows, 16-byte chunk c of key k at position c ^ (((k >> 1) & 1) << 2):
// the 4 keys x 64 bytes a 32-lane half reads hit 64 distinct banks.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ const uint4 kAttnZero16 = {0u, 0u, 0u, 0u};

template <int NB>
__global__ __launch_bounds__(512) void attn_mfma_persist_kernel(const bf16_t* __restrict__ qkv, int64_t ld_qkv,
                                                                bf16_t* __restrict__ out, int64_t ld_out, int tokens,
                                                                int heads, int n_items, float scale_log2e) {
  constexpr int KEYS = NB * 32;
  constexpr int HALF = KEYS * 128;        // K image, then V image
  constexpr int BUF = 2 * HALF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int D = heads * HD;
  const int ksw = (r >> 1) & 7;

  // stage K and V of `item` into buffer `b`: 2*KEYS/8 one-KiB pieces, dealt round-robin to the 8 waves.
  // The per-lane source offsets do not depend on the item: computed once (-1 = pad key -> zero constant).
  constexpr int PIECES = KEYS / 8;  // per operand
  constexpr int MYP = (2 * PIECES + 7) / 8;
  int64_t soff[MYP];
#pragma unroll
  for (int i = 0; i < MYP; ++i) {
    const int piece = wave + 8 * i;
    const int isv = piece >= PIECES;
    const int pp = isv ? piece - PIECES : piece;
    const int key = pp * 8 + (lane >> 3), pos = lane & 7;
    const int ch = isv ? (pos ^ (((key >> 1) & 1) << 2)) : (pos ^ ((key >> 1) & 7));
    soff[i] = (piece < 2 * PIECES && key < tokens) ? (int64_t)key * ld_qkv + (isv ? 2 * D : D) + ch * 8 : -1;
  }
  auto stage = [&](int item, int b) {
    const int frame = item / heads, head = item - frame * heads;
    const bf16_t* base = qkv + (int64_t)frame * tokens * ld_qkv + head * HD;
    unsigned char* kb_ = smem + b * BUF;
#pragma unroll
    for (int i = 0; i < MYP; ++i) {
      const int piece = wave + 8 * i;
      if (piece < 2 * PIECES) {
        const int isv = piece >= PIECES;
        const int pp = isv ? piece - PIECES : piece;
        const void* g = soff[i] >= 0 ? static_cast<const void*>(base + soff[i]) : static_cast<const void*>(&kAttnZero16);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(kb_ + isv * HALF + pp * 1024), 16, 0, 0);
      }
    }
  };

  const int qb = wave;
  const int q = qb * 32 + r;
  auto load_q = [&](bf16x8 (&qv)[4], int it) {
    if (qb >= NB) return;
    const int fr_ = it / heads, hd_ = it - fr_ * heads;
    const bf16_t* b_ = qkv + (int64_t)fr_ * tokens * ld_qkv + hd_ * HD;
    const int qc = q < tokens ? q : tokens - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qv[s] = *reinterpret_cast<const bf16x8*>(b_ + (int64_t)qc * ld_qkv + s * 16 + h * 8);
  };
  int item = blockIdx.x;
  bf16x8 qf[4], qn[4];
#if ATTN_STAMPS
  unsigned long long a_w = 0, a_i = 0, a_qk = 0, a_sm = 0, a_pv = 0, a_st = 0;
  const unsigned long long t_begin = astamp();
#endif
  if (item < n_items) {
    stage(item, 0);
    load_q(qf, item);
  }
  for (int n = 0; item < n_items; ++n, item += gridDim.x) {
    const int cur = n & 1;
    const int frame = item / heads, head = item - frame * heads;
    const unsigned char* Ks = smem + cur * BUF;
    const unsigned char* Vs = Ks + HALF;
    // my pieces of this item and my Q were issued one item ago (letting the previous item's stores stay in
    // flight with a counted vmcnt + raw s_barrier measured slower: 172 vs 159 us)
    PSTAMP(p0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // everyone's pieces have landed and everyone is done reading the other buffer
    PSTAMP(p1);
    if (item + (int)gridDim.x < n_items) {
      stage(item + gridDim.x, cur ^ 1);
      load_q(qn, item + gridDim.x);  // the next item's Q fragments fly during this item's compute
    }
    PSTAMP(p2);
    if (qb >= NB) continue;

    // ---- Sᵀ[key][q] for all NB key blocks (as in the kernel above) ----------------------------------
    f32x16 S[NB];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 kfa[NB], kfb[NB];
    auto read_k = [&](bf16x8 (&kf)[NB], int s) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
        kf[kb] = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * 128 + (((2 * s + h) ^ ksw) << 4));
    };
    auto mma_k = [&](const bf16x8 (&kf)[NB], int s) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)  // first slice: C = inline constant 0, no accumulator clearing
        S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb], qf[s], s == 0 ? zero16 : S[kb], 0, 0, 0);
    };
    read_k(kfa, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_k(kfb, 1);
    mma_k(kfa, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_k(kfa, 2);
    mma_k(kfb, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_k(kfb, 3);
    mma_k(kfa, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_k(kfb, 3);
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(p3);

    // ---- softmax over the key axis ------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (key >= tokens) S[NB - 1][e] = -INFINITY;
    }
    // The softmax is the VALU-bound part of the kernel (two waves per SIMD): v_max3 without the NaN
    // canonicalisation fmaxf() drags in, packed f32 multiply-add and packed row sums.
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) mx = vmax3(mx, S[kb][e], S[kb][e + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sl2 = scale_log2e;
    float nmc = -mx * scale_log2e;
    asm volatile("" : "+s"(sl2));  // opaque scalars: the vector expression below packs into v_pk_fma_f32
    f32x16 lv;
#pragma unroll
    for (int e = 0; e < 16; ++e) lv[e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2v t = __builtin_elementwise_fma(f32x2v{S[kb][e], S[kb][e + 1]}, f32x2v{sl2, sl2}, f32x2v{nmc, nmc});
        S[kb][e] = __builtin_amdgcn_exp2f(t[0]);
        S[kb][e + 1] = __builtin_amdgcn_exp2f(t[1]);
      }
      lv += S[kb];
    }
    float l = ((lv[0] + lv[1]) + (lv[2] + lv[3])) + ((lv[4] + lv[5]) + (lv[6] + lv[7])) +
              (((lv[8] + lv[9]) + (lv[10] + lv[11])) + ((lv[12] + lv[13]) + (lv[14] + lv[15])));
    l += __shfl_xor(l, 32, 64);
    PSTAMP(p4);

    // ---- Oᵀ[d][q] = Σ_key V[key][d] · Pᵀ[key][q]: V operand through the transposing read ------------
    // lane (r, h): d row r of tile dt, keys base + 4h + {0..3} and base + 8 + 4h + {0..3}.  In its 16-lane
    // group (d columns 16*(r>>4) ..), lane 4q+p supplies the address of key row q, d columns 4p..4p+3.
    f32x16 O[2];
    const int tq = (lane & 15) >> 2, tp = lane & 3, rr = (lane >> 4) & 1;
    auto read_v = [&](bf16x8 (&vf)[2], int step) {
      const int kbase = (step >> 1) * 32 + 16 * (step & 1) + 4 * h + tq;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int c = 4 * dt + 2 * rr + (tp >> 1);
        const int k1 = kbase, k2 = kbase + 8;
        const unsigned char* a1 = Vs + k1 * 128 + ((c ^ (((k1 >> 1) & 1) << 2)) << 4) + 8 * (tp & 1);
        const unsigned char* a2 = Vs + k2 * 128 + ((c ^ (((k2 >> 1) & 1) << 2)) << 4) + 8 * (tp & 1);
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)a1);
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)a2);
        short8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        vf[dt] = __builtin_bit_cast(bf16x8, v);
      }
    };
    auto mma_v = [&](const bf16x8 (&vf)[2], int step) {
      const int kb = step >> 1, sl = step & 1;
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)S[kb][8 * sl + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pf, step == 0 ? zero16 : O[dt], 0, 0, 0);
    };
    bf16x8 vfa[2], vfb[2];
    read_v(vfa, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int step = 0; step < 2 * NB; step += 2) {
      read_v(vfb, step + 1);
      mma_v(vfa, step);
      __builtin_amdgcn_sched_barrier(0);
      if (step + 2 < 2 * NB) read_v(vfa, step + 2);
      mma_v(vfb, step + 1);
      __builtin_amdgcn_sched_barrier(0);
    }

    PSTAMP(p5);
    if (q < tokens) {
      const float inv = 1.0f / l;
      bf16_t* op = out + ((int64_t)frame * tokens + q) * ld_out + head * HD;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(O[dt][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * g + 4 * h) = o;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = qn[s];
#if ATTN_STAMPS
    {
      const unsigned long long p6 = astamp();
      a_w += p1 - p0; a_i += p2 - p1; a_qk += p3 - p2; a_sm += p4 - p3; a_pv += p5 - p4; a_st += p6 - p5;
    }
#endif
  }
#if ATTN_STAMPS
  if (lane == 0 && g_attn_dbg && wave < 2) {
    float* d = g_attn_dbg + ((int64_t)blockIdx.x * 2 + wave) * 8;
    d[0] = (float)a_w; d[1] = (float)a_i; d[2] = (float)a_qk; d[3] = (float)a_sm; d[4] = (float)a_pv; d[5] = (float)a_st;
    d[6] = (float)(astamp() - t_begin);
  }
#endif
}

template <int NB>
int launch_persist(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads, float scale,
                   hipStream_t st) {
  constexpr int BYTES = 2 * 2 * NB * 32 * 128;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_persist_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
  }
  const int n_items = n_frames * heads;
  const int grid = n_items < ncu ? n_items : ncu;
  hipLaunchKernelGGL((attn_mfma_persist_kernel<NB>), dim3(grid), dim3(512), BYTES, st, static_cast<const bf16_t*>(qkv), ld_qkv,
                     static_cast<bf16_t*>(out), ld_out, tokens, heads, n_items, scale * 1.4426950408889634f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_attention_fwd(mfma, persistent): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}

template <int NB>
int launch(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads, float scale,
           hipStream_t st) {
  using L = Lds<NB>;
  static_assert(L::VSTRIDE % 16 == 8, "Vt row stride must be 8 mod 16 bytes");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, L::BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_mfma_kernel<NB>), dim3(n_frames * heads), dim3(256), L::BYTES, st,
                     static_cast<const bf16_t*>(qkv), ld_qkv, static_cast<bf16_t*>(out), ld_out, tokens, heads,
                     scale * 1.4426950408889634f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_attention_fwd(mfma): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}

}  // namespace

#if ATTN_STAMPS
extern "C" void dfd_attn_set_debug(float* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &p, sizeof(p)); }
#endif

int dfd_attention_mfma_try(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads,
                           float scale, hipStream_t st) {
  if ((ld_qkv % 8) != 0 || (ld_out % 4) != 0) return 1;
  // NB = ceil(tokens / 32) exactly: the kernel masks only its last key block
  static const bool persist = getenv("DFD_ATTN_PERSIST") == nullptr || getenv("DFD_ATTN_PERSIST")[0] != '0';
  if (tokens > 6 * 32 && tokens <= 7 * 32) {
    if (persist && n_frames * heads >= 512) return launch_persist<7>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
    return launch<7>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  }
  if (tokens > 8 * 32 && tokens <= 9 * 32) return launch<9>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  return 1;
}
