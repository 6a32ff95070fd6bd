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
#include "attention_common.hpp"

namespace {

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
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int c = tid + it * 256;
    const int key = c >> 3, ch = c & 7;
    *reinterpret_cast<bf16x8*>(Ks + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[it];
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16_t*>(Vt + (ch * 8 + e) * L::VSTRIDE + key * 2) = vreg[it][e];
  }
  __syncthreads();

  const int ksw = (r >> 1) & 7;
  for (int qb = wave; qb < NB; qb += 4) {
    const int q = qb * 32 + r;
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
  }
}

// ---- persistent variant (tokens <= 208) -----------------------------------------------------------------
// One 512-thread workgroup per CU walks (frame, head) items; every byte of an item moves as whole 128-byte
// lines.  K, V and the Q rows of an item go to LDS by LDS-DMA (buffer_load ... lds: no register round trip, the
// bank swizzle applied on the SOURCE address), double-buffered: item i+1 is in flight while item i is computed,
// and there is one workgroup barrier per item.  Wave w owns query block w (one wave only stages when NB = 7).
//   LDS: [K0][V0][K1][V1] images of `tokens` 128-byte rows (no pad rows: the MFMA reads of key rows
//   tokens .. 32*NB-1 fall into whatever follows — K scores of those keys are overwritten with -inf, their P is
//   exactly 0 and what V reads there is finite: real bf16 data or the zeros the kernel starts with), then two
//   4 KB Q images per wave.  The output tile of a wave (32 queries x 64 channels) is transposed through the Q
//   image it has just consumed and leaves as 16-byte-per-lane, whole-line stores.
// V stays row-major [key][64] and its MFMA operand (d rows, 8 keys per lane: two runs of 4 keys) is read with the
// transposing LDS read ds_read_b64_tr_b16.  V image: 16-byte chunk c of key k at position c ^ (((k >> 1) & 1) << 2):
// the 4 keys x 64 bytes a 32-lane half reads hit 64 distinct banks.  K and Q images: chunk c of row r at
// c ^ ((r >> 1) & 7).
// Items are dealt so that the workgroups of one XCD (blockIdx % 8) walk the heads of the same few frames at the
// same time: their 128-byte pieces of a qkv row are neighbours in memory.
template <int NB>
__device__ __forceinline__ void attn_stage_q(__amdgpu_buffer_rsrc_t srd, unsigned char* qimg, uint32_t sbase, uint32_t ldq, int lane,
                                             int tokens) {
  const int sub = lane >> 3, pos = lane & 7;
  const uint32_t q_even = (pos ^ ((sub >> 1) & 3)) << 4, q_odd = (pos ^ (((sub >> 1) & 3) | 4)) << 4;
#pragma unroll
  for (int p = 0; p < 4 * NB; ++p) {  // query rows past the frame repeat its last row (never stored)
    const uint32_t row = (uint32_t)min(8 * p + sub, tokens - 1);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(qimg + p * 1024), 16, row * ldq + ((p & 1) ? q_odd : q_even), sbase, 0, 0);
  }
}

// s_waitcnt vmcnt(n) for a run-time n (the loader wave only): at most n of my memory instructions still in flight
__device__ __forceinline__ void wait_vm_dyn(int n) {
  switch (n < 0 ? 0 : n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
    case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
    case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
    case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
    case 51: asm volatile("s_waitcnt vmcnt(51)" ::: "memory"); break;
    case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 53: asm volatile("s_waitcnt vmcnt(53)" ::: "memory"); break;
    case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
    case 55: asm volatile("s_waitcnt vmcnt(55)" ::: "memory"); break;
    case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
  }
}

// SHORT: the last key block holds at most 8 keys (197 = 6*32 + 5, 257 = 8*32 + 1), i.e. only elements 0..3 of its score
// registers can be live — the other twelve are neither exponentiated nor multiplied (a tenth of the softmax, which bounds the
// kernel, and one of the 2*NB steps of the second product).
template <int NB, bool SHORT>
__global__ __launch_bounds__(512) void attn_mfma_persist_kernel(const bf16_t* __restrict__ qkv, uint32_t ldq /* bytes */,
                                                                bf16_t* __restrict__ out, uint32_t ldo /* bytes */,
                                                                int tokens, int heads, int n_frames, uint32_t qkv_bytes,
                                                                uint32_t out_bytes, float scale_log2e) {
  constexpr int QIMG = 32 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));  // opaque: per-item addresses are rebuilt from it instead of living in registers
  const int r = lane & 31, h = lane >> 5;
  const uint32_t Db = (uint32_t)heads * HD * 2;
  const int ksw = (r >> 1) & 7;
  const int img = tokens * 128;
  const int np = (tokens * 8 + 63) >> 6;  // 1 KB pieces per K or V image (the last one partial)
  const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
  unsigned char* const qs0 = smem + 4 * img + (wave < NB ? wave : 0) * QIMG;  // my Q image of buffer 0; buffer 1 is NB * QIMG further

  // a finite start state for the rows the images do not cover
  for (int i = tid * 16; i < 4 * img + 2 * NB * QIMG; i += 512 * 16) *reinterpret_cast<v4i_t*>(smem + i) = v4i_t{0, 0, 0, 0};
  __syncthreads();

  const int qb = wave;  // compute waves 0 .. NB-1 own one 32-query block each
  // item n of this workgroup
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  auto item_of = [&](int n, int& fr_, int& hd_) -> bool {
    const int li = n * per_xcd + slot;
    const int fl = li / heads;
    hd_ = li - fl * heads;
    fr_ = fl * 8 + xcd;
    return fr_ < n_frames;
  };
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };  // no release fence: see below

  // ---- the loader wave ----------------------------------------------------------------------------------
  // The spare wave issues every LDS-DMA of the workgroup, so the compute waves never queue behind the 78 one-KB
  // transfers of an item (a sixth of their time when each wave staged its share).  Two barriers per item:
  //   top of item n    : K(n) and Q(n) are in; the other Q image is free          -> Q(n+1)
  //   middle of item n : V(n) is in; the waves hold their scores, K(n) is dead    -> K(n+2) over it, V(n+1)
  // so an item's transfers are issued while the compute waves work (28 during their QK product, 50 during the
  // softmax and PV), each with a lead of one item or more, and the memory pipe of the CU never drains.
  // vmcnt counts in order: to know that a group has landed, allow exactly the instructions issued after it.
  if (wave == NB) {
    int f1, h1, f2, h2;
    auto sbase_of = [&](int f, int hh) { return (uint32_t)f * tokens * ldq + hh * (HD * 2); };
    int issued = 0, mark_kq = 0, mark_v = 0;
    bool have = item_of(0, f1, h1);
    if (have) {
      attn_stage_k(srdQ, smem, sbase_of(f1, h1), ldq, Db, lane, np, tokens);
      attn_stage_q<NB>(srdQ, smem + 4 * img, sbase_of(f1, h1), ldq, lane, tokens);
      issued += np + 4 * NB;
      mark_kq = issued;
      if (item_of(1, f2, h2)) {
        attn_stage_k(srdQ, smem + 2 * img, sbase_of(f2, h2), ldq, Db, lane, np, tokens);
        issued += np;
      }
      attn_stage_v(srdQ, smem + img, sbase_of(f1, h1), ldq, Db, lane, np, tokens);
      issued += np;
      mark_v = issued;
    }
    for (int n = 0; have; ++n) {
      const int cur = n & 1;
      wait_vm_dyn(issued - mark_kq);
      barrier();
      const bool have1 = item_of(n + 1, f1, h1);
      if (have1) {
        attn_stage_q<NB>(srdQ, smem + 4 * img + (cur ^ 1) * NB * QIMG, sbase_of(f1, h1), ldq, lane, tokens);
        issued += 4 * NB;
        mark_kq = issued;  // K(n+1) went out before it
      }
      wait_vm_dyn(issued - mark_v);
      barrier();
      if (have1 && item_of(n + 2, f2, h2)) {
        attn_stage_k(srdQ, smem + cur * 2 * img, sbase_of(f2, h2), ldq, Db, lane, np, tokens);
        issued += np;
      }
      if (have1) {
        attn_stage_v(srdQ, smem + (cur ^ 1) * 2 * img + img, sbase_of(f1, h1), ldq, Db, lane, np, tokens);
        issued += np;
        mark_v = issued;
      }
      have = have1;
    }
    return;
  }
  if (wave > NB) return;

  // ---- the compute waves --------------------------------------------------------------------------------
  int frame, head, nframe = 0, nhead = 0;
  bool have = item_of(0, frame, head);
  for (int n = 0; have; ++n) {
    const int cur = n & 1;
    const unsigned char* Ks = smem + cur * 2 * img;
    const unsigned char* Vs = Ks + img;
    unsigned char* Qs = qs0 + cur * NB * QIMG;
    // The output stores of the previous item stay in flight: nothing here waits for them (a raw s_barrier;
    // __syncthreads() would add the workgroup release fence, i.e. a wait for those stores).
    barrier();
    const bool have_next = item_of(n + 1, nframe, nhead);
    {
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qs + r * 128 + (((2 * s + h) ^ ksw) << 4));

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
    barrier();  // every wave has read its K fragments (the loader may overwrite the K image); V is in

    // ---- softmax over the key axis, then Oᵀ[d][q] = Σ_key V[key][d] · Pᵀ[key][q] (attention_common.hpp) -------------
    const float l = attn_softmax<NB, SHORT>(S, tokens, h, scale_log2e);
    f32x16 O[2];
    attn_pv<NB, SHORT>(S, Vs, lane, O);
    // ---- output: [q][64] bf16 through my (consumed) Q image, then whole 128-byte lines --------------
    {
      const float inv = 1.0f / l;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(O[dt][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(Qs + r * 128 + (((4 * dt + g) ^ ksw) << 4) + 8 * h) = o;
        }
      const uint32_t obase = (uint32_t)frame * tokens * ldo + head * (HD * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 8 * i + (lane >> 3), pos = lane & 7;
        const v4i_t d = *reinterpret_cast<const v4i_t*>(Qs + rl * 128 + ((pos ^ ((rl >> 1) & 7)) << 4));
        const int qq = qb * 32 + rl;
        // (the item's base goes into the lane offset, not into the scalar-offset field: see attn_store_line())
        const uint32_t off = qq < tokens ? obase + (uint32_t)qq * ldo + pos * 16 : 0xffffffffu;  // rows past the frame: dropped
        attn_store_line(d, srdO, off);
      }
    }
    }
    have = have_next;
    frame = nframe;
    head = nhead;
  }
}

// 1 = not served (shape outside the persistent kernel's LDS budget or 32-bit offsets)
template <int NB, bool SHORT>
int launch_persist(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads, float scale,
                   hipStream_t st) {
  const int64_t lds = (int64_t)4 * tokens * 128 + 2 * NB * 4096;
  const int64_t qkv_bytes = ((int64_t)n_frames * tokens - 1) * ld_qkv * 2 + (int64_t)3 * heads * HD * 2;
  const int64_t out_bytes = ((int64_t)n_frames * tokens - 1) * ld_out * 2 + (int64_t)heads * HD * 2;
  if (lds > 160 * 1024 || qkv_bytes > (int64_t)0xfffffff0 || out_bytes > (int64_t)0xfffffff0 || (ld_qkv % 8) != 0 || (ld_out % 8) != 0) return 1;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_persist_kernel<NB, SHORT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int grid = ncu & ~7;  // a whole number of workgroups per XCD
  if (grid < 8) return 1;
  hipLaunchKernelGGL((attn_mfma_persist_kernel<NB, SHORT>), dim3(grid), dim3(512), (size_t)lds, st, static_cast<const bf16_t*>(qkv),
                     (uint32_t)(ld_qkv * 2), static_cast<bf16_t*>(out), (uint32_t)(ld_out * 2), tokens, heads, n_frames,
                     (uint32_t)qkv_bytes, (uint32_t)out_bytes, scale * 1.4426950408889634f);
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


int dfd_attention_mfma_xrow_try(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads,
                                float scale, hipStream_t st);  // attention_mfma_xrow.hip

int dfd_attention_mfma_try(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads,
                           float scale, hipStream_t st) {
  if ((ld_qkv % 8) != 0 || (ld_out % 4) != 0) return 1;
  // NB = ceil(tokens / 32) exactly: the kernels mask only their last key block
  if (tokens > 6 * 32 && tokens <= 7 * 32) {
    if (n_frames * heads >= 512) {
      const int rc = tokens <= 6 * 32 + 8 ? launch_persist<7, true>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st)
                                          : launch_persist<7, false>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
      if (rc <= 0) return rc;
    }
    return launch<7>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  }
  if (tokens > 8 * 32 && tokens <= 9 * 32) {
    const int rc = dfd_attention_mfma_xrow_try(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);  // 257 tokens, >= 512 items
    if (rc <= 0) return rc;
    return launch<9>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  }
  return 1;
}
