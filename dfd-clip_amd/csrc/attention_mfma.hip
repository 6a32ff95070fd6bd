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

namespace {

constexpr int HD = 64;

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
    const int qc = q < tokens ? q : tokens - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qc * ld_qkv + s * 16 + h * 8);

    // ---- Sᵀ[key][q] for all NB key blocks ---------------------------------------------------------
    f32x16 S[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) S[kb][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * 128 + (((2 * s + h) ^ ksw) << 4));
        S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], S[kb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the K-fragment reads of later k-steps from being hoisted (spills)
    }

    // ---- softmax over the key axis: registers + one exchange with lane ^ 32 -----------------------
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= tokens) S[kb][e] = -INFINITY;
        mx = fmaxf(mx, S[kb][e]);
      }
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

    // ---- Oᵀ[d][q] = Σ_key Vt[d][key] · Pᵀ[key][q]; P fragments come straight from the S registers ------
    f32x16 O[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)S[kb][8 * s + j];
        // element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3) of the block: two runs of 4 keys
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned char* vp = Vt + (dt * 32 + r) * L::VSTRIDE + (kb * 32 + 16 * s + 4 * h) * 2;
          const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vp);
          const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vp + 16);
          const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, O[dt], 0, 0, 0);
        }
        if (s == 1) __builtin_amdgcn_sched_barrier(0);
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
  }
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

int dfd_attention_mfma_try(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads,
                           float scale, hipStream_t st) {
  if ((ld_qkv % 8) != 0 || (ld_out % 4) != 0) return 1;
  if (tokens > 32 && tokens <= 7 * 32) return launch<7>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  if (tokens > 7 * 32 && tokens <= 9 * 32) return launch<9>(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
  return 1;
}
