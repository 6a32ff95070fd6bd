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
#else
#define ASTAMP(v)
#endif

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
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) mx = fmaxf(mx, fmaxf(S[kb][e], S[kb][e + 1]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * scale_log2e;
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf(S[kb][e], scale_log2e, -mc));
        S[kb][e] = p;
        l += p;
      }
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
  if (tokens > 6 * 32 && tokens <= 7 * 32) return launch<7>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  if (tokens > 8 * 32 && tokens <= 9 * 32) return launch<9>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  return 1;
}
