// Pieces shared by the persistent encoder-attention kernels (attention_mfma.hip: 193..224 tokens, loader wave;
// attention_mfma_xrow.hip: 257 tokens, eight query blocks + one extra row): LDS images and their LDS-DMA staging, the
// register softmax and the second product.  head_dim 64 everywhere (reference clip/model.py:188-195).
#pragma once
#include "common.hpp"

namespace {

constexpr int HD = 64;

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// LDS-DMA in 1 KB pieces (8 rows): lane (sub = lane >> 3, pos = lane & 7) moves 16-byte chunk pos^swizzle of row
// 8*piece + sub; the swizzle of the K and Q images, ((row >> 1) & 7), depends on the piece only through its parity.
// attn_stage_k / _v issue pieces p0, p0 + dp, .. < np (every piece has live lanes): all of them from a loader wave
// (p0 = 0, dp = 1), or every eighth when the eight compute waves share the job (dp even: one parity per wave).
__device__ __forceinline__ void attn_stage_k(__amdgpu_buffer_rsrc_t srd, unsigned char* kimg, uint32_t sbase, uint32_t ldq,
                                             uint32_t Db, int lane, int np, int tokens, int p0 = 0, int dp = 1) {
  const int sub = lane >> 3, pos = lane & 7;
  const uint32_t row0 = (uint32_t)sub * ldq;
  const uint32_t k_even = row0 + Db + ((pos ^ ((sub >> 1) & 3)) << 4), k_odd = row0 + Db + ((pos ^ (((sub >> 1) & 3) | 4)) << 4);
  for (int p = p0; p < np; p += dp) {
    if (8 * p + sub < tokens)  // lanes past the image write nothing
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(kimg + p * 1024), 16, ((p & 1) ? k_odd : k_even) + p * 8 * ldq, sbase, 0, 0);
  }
}
__device__ __forceinline__ void attn_stage_v(__amdgpu_buffer_rsrc_t srd, unsigned char* vimg, uint32_t sbase, uint32_t ldq,
                                             uint32_t Db, int lane, int np, int tokens, int p0 = 0, int dp = 1) {
  const int sub = lane >> 3, pos = lane & 7;
  const uint32_t v_any = (uint32_t)sub * ldq + 2 * Db + ((pos ^ (((sub >> 1) & 1) << 2)) << 4);
  for (int p = p0; p < np; p += dp) {
    if (8 * p + sub < tokens)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(vimg + p * 1024), 16, v_any + p * 8 * ldq, sbase, 0, 0);
  }
}

// ---- softmax over the key axis of Sᵀ[key][q] (lane = query column; key of element e of block kb, lane half h:
// 32 kb + (e & 3) + 8 (e >> 2) + 4 h): registers + one exchange with lane ^ 32.  Returns the row sum; S holds the
// exponentials.  SHORT: the last block has at most 8 keys, so only its elements 0..3 can be live — the other twelve are
// not exponentiated (elements 4..7 are cleared for the one step of the second product that still runs).
// The softmax is the VALU-bound part of these kernels (two waves per SIMD): v_max3 without the NaN canonicalisation
// fmaxf() drags in, packed f32 multiply-add and packed row sums.
struct AttnNoTick {
  __device__ __forceinline__ void operator()(int) const {}
};
// tick(kb) runs behind the exponentials of block kb (a kernel without a loader wave spreads its LDS-DMA there).
template <int NB, bool SHORT, typename Tick = AttnNoTick>
__device__ __forceinline__ float attn_softmax(f32x16 (&S)[NB], int tokens, int h, float scale_log2e, Tick tick = Tick()) {
  constexpr int LE = SHORT ? 4 : 16;
#pragma unroll
  for (int e = 0; e < LE; ++e) {
    const int key = (NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (key >= tokens) S[NB - 1][e] = -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int e = 0; e < (kb == NB - 1 ? LE : 16); e += 2) mx = vmax3(mx, S[kb][e], S[kb][e + 1]);
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
    for (int e = 0; e < (kb == NB - 1 ? LE : 16); e += 2) {
      const f32x2v t = __builtin_elementwise_fma(f32x2v{S[kb][e], S[kb][e + 1]}, f32x2v{sl2, sl2}, f32x2v{nmc, nmc});
      S[kb][e] = __builtin_amdgcn_exp2f(t[0]);
      S[kb][e + 1] = __builtin_amdgcn_exp2f(t[1]);
    }
    if (SHORT && kb == NB - 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) lv[e] += S[kb][e];
#pragma unroll
      for (int e = 4; e < 8; ++e) S[kb][e] = 0.f;
    } else {
      lv += S[kb];
    }
    tick(kb);
  }
  float l = ((lv[0] + lv[1]) + (lv[2] + lv[3])) + ((lv[4] + lv[5]) + (lv[6] + lv[7])) +
            (((lv[8] + lv[9]) + (lv[10] + lv[11])) + ((lv[12] + lv[13]) + (lv[14] + lv[15])));
  l += __shfl_xor(l, 32, 64);
  return l;
}

// ---- V operand of Oᵀ[d][q] = Σ_key V[key][d] · Pᵀ[key][q] for one 16-key step, through the transposing LDS read.
// V image: [key][64] bf16, 16-byte chunk c of key k at position c ^ (((k >> 1) & 1) << 2).  Lane (r, h): d row r of
// tile dt, keys base + 4h + {0..3} and base + 8 + 4h + {0..3}; in its 16-lane group (d columns 16*(r>>4) ..), lane
// 4q+p supplies the address of key row q, d columns 4p..4p+3.
__device__ __forceinline__ void attn_read_v(bf16x8 (&vf)[2], const unsigned char* Vs, int lane, int step) {
  const int h = lane >> 5, tq = (lane & 15) >> 2, tp = lane & 3, rr = (lane >> 4) & 1;
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
}

// The second product over all 16-key steps with a live key; the P operand comes straight out of the S registers
// (accumulator-as-operand, converted pairwise to bf16), V fragments are read one step ahead of the MFMAs that use them.
struct AttnNoRider {
  __device__ __forceinline__ void operator()(const bf16x8 (&)[2], int) const {}
};
// rider(vf, step) runs behind the MFMAs of a step with the V fragments still in registers (a second, independent
// accumulation over the same V — the 257-token kernel's extra row).
template <int NB, bool SHORT, typename Rider = AttnNoRider>
__device__ __forceinline__ void attn_pv(const f32x16 (&S)[NB], const unsigned char* Vs, int lane, f32x16 (&O)[2], Rider rider = Rider()) {
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto mma_v = [&](const bf16x8 (&vf)[2], int step) {
    const int kb = step >> 1, sl = step & 1;
    bf16x8 pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)S[kb][8 * sl + j];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pf, step == 0 ? zero16 : O[dt], 0, 0, 0);
  };
  bf16x8 vfa[2], vfb[2];
  attn_read_v(vfa, Vs, lane, 0);
  __builtin_amdgcn_sched_barrier(0);
  constexpr int NS = SHORT ? 2 * NB - 1 : 2 * NB;  // 16-key steps with a live key
#pragma unroll
  for (int step = 0; step < NS; step += 2) {
    if (step + 1 < NS) attn_read_v(vfb, Vs, lane, step + 1);
    mma_v(vfa, step);
    rider(vfa, step);
    __builtin_amdgcn_sched_barrier(0);
    if (step + 2 < NS) attn_read_v(vfa, Vs, lane, step + 2);
    if (step + 1 < NS) {
      mma_v(vfb, step + 1);
      rider(vfb, step + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One 16-byte store per lane, scalar-offset field = 0.  With a REGISTER in that field the compiler's hazard recognizer
// leaves out the wait states it otherwise puts between a store of more than 8 bytes per lane and a VALU write of the
// store's data registers (LLVM, GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction is not
// using a register in the soffset field") — and on gfx950 the store had not always read its data by then: it went out
// with the next instruction's result in its first data register for the lanes read last (round 3: 32 elements of a
// 32 x 64 tile, sporadically, more often on the second wave of a SIMD).  tools/isa_lint.py checks every kernel for it.
__device__ __forceinline__ void attn_store_line(v4i_t d, __amdgpu_buffer_rsrc_t srd, uint32_t off) {
  __builtin_amdgcn_raw_buffer_store_b128(d, srd, off, 0, 0);
}

template <int N> __device__ __forceinline__ void attn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
