// Encoder self-attention, out = softmax(q kᵀ / √d) v per (frame, head), head_dim = 64
// (reference clip/model.py:188-195).  The [N, tokens, tokens, heads] affinity tensor the
// reference materialises (894 MB per layer at B·T = 480) never exists here.
//
// attn_rows_kernel: general kernel for both dtypes (the f32 parity path and the fallback for
//   shapes the MFMA kernel does not take).  One workgroup per (frame, head); K and V of that
//   head staged once in LDS in their storage dtype; one query row per thread held in
//   registers; keys streamed from LDS as wave-wide broadcasts; online softmax in fp32.
// The bf16 MFMA kernel lives in attention_mfma.hip.
#include "common.hpp"

namespace {

constexpr int HD = 64;

template <typename T> struct Vec16;  // 16-byte vector of T
template <> struct Vec16<float> { using type = f32x4; static constexpr int N = 4; };
template <> struct Vec16<bf16_t> { using type = bf16x8; static constexpr int N = 8; };

template <typename T>
__global__ __launch_bounds__(256) void attn_rows_kernel(const T* __restrict__ qkv, int64_t ld_qkv, T* __restrict__ out,
                                                        int64_t ld_out, int tokens, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ks = reinterpret_cast<T*>(smem_raw);
  T* Vs = Ks + (size_t)tokens * HD;
  using V16 = typename Vec16<T>::type;
  constexpr int VN = Vec16<T>::N;
  constexpr int CHUNKS = HD / VN;  // 16-byte chunks per head row
  const int frame = blockIdx.x / heads, head = blockIdx.x % heads;
  const int D = heads * HD;
  const T* base = qkv + (int64_t)frame * tokens * ld_qkv + head * HD;

  for (int c = threadIdx.x; c < tokens * CHUNKS; c += blockDim.x) {
    const int row = c / CHUNKS, ch = c % CHUNKS;
    const T* src = base + (int64_t)row * ld_qkv + ch * VN;
    *reinterpret_cast<V16*>(Ks + row * HD + ch * VN) = *reinterpret_cast<const V16*>(src + D);
    *reinterpret_cast<V16*>(Vs + row * HD + ch * VN) = *reinterpret_cast<const V16*>(src + 2 * D);
  }
  __syncthreads();

  for (int qi = threadIdx.x; qi < tokens; qi += blockDim.x) {
    float q[HD], acc[HD];
    const T* qp = base + (int64_t)qi * ld_qkv;
#pragma unroll
    for (int ch = 0; ch < CHUNKS; ++ch) {
      const V16 v = *reinterpret_cast<const V16*>(qp + ch * VN);
#pragma unroll
      for (int e = 0; e < VN; ++e) q[ch * VN + e] = to_f32(v[e]) * scale;
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    float mx = -INFINITY, l = 0.f;
    for (int j = 0; j < tokens; ++j) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int ch = 0; ch < CHUNKS; ++ch) {
        const V16 kv = *reinterpret_cast<const V16*>(Ks + j * HD + ch * VN);
#pragma unroll
        for (int e = 0; e < VN; e += 2) {
          s0 = fmaf(q[ch * VN + e], to_f32(kv[e]), s0);
          s1 = fmaf(q[ch * VN + e + 1], to_f32(kv[e + 1]), s1);
        }
      }
      const float s = s0 + s1;
      if (s > mx) {
        const float alpha = __expf(mx - s);
        l *= alpha;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] *= alpha;
        mx = s;
      }
      const float p = __expf(s - mx);
      l += p;
#pragma unroll
      for (int ch = 0; ch < CHUNKS; ++ch) {
        const V16 vv = *reinterpret_cast<const V16*>(Vs + j * HD + ch * VN);
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[ch * VN + e] = fmaf(p, to_f32(vv[e]), acc[ch * VN + e]);
      }
    }
    const float inv = 1.0f / l;
    T* op = out + ((int64_t)frame * tokens + qi) * ld_out + head * HD;
#pragma unroll
    for (int ch = 0; ch < CHUNKS; ++ch) {
      V16 o;
#pragma unroll
      for (int e = 0; e < VN; ++e) o[e] = from_f32<T>(acc[ch * VN + e] * inv);
      *reinterpret_cast<V16*>(op + ch * VN) = o;
    }
  }
}

}  // namespace

// MFMA kernel (attention_mfma.hip); returns 1 when the shape is not eligible
int dfd_attention_mfma_try(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens,
                           int heads, float scale, hipStream_t st);

extern "C" int dfd_attention_fwd(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int dtype, int n_frames,
                                 int tokens, int heads, int head_dim, float scale, void* stream) {
  DFD_REQUIRE(qkv && out, "dfd_attention_fwd: null pointer");
  DFD_REQUIRE(head_dim == HD, "dfd_attention_fwd: head_dim=%d, only 64 is supported", head_dim);
  DFD_REQUIRE(n_frames >= 0 && tokens > 0 && heads > 0, "dfd_attention_fwd: bad shape");
  DFD_REQUIRE(dtype == DFD_F32 || dtype == DFD_BF16, "dfd_attention_fwd: dtype=%d", dtype);
  const int esz = dtype == DFD_F32 ? 4 : 2;
  DFD_REQUIRE(ld_qkv >= 3 * heads * HD && ld_out >= heads * HD && (ld_qkv * esz) % 16 == 0 && (ld_out * esz) % 16 == 0,
              "dfd_attention_fwd: bad leading dimensions");
  DFD_REQUIRE(dfd_aligned16(qkv) && dfd_aligned16(out), "dfd_attention_fwd: pointers must be 16-byte aligned");
  const size_t lds = (size_t)2 * tokens * HD * esz;
  DFD_REQUIRE(lds <= 160 * 1024, "dfd_attention_fwd: tokens=%d needs %zu B of LDS (> 160 KiB)", tokens, lds);
  if (n_frames == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == DFD_BF16) {
    const int rc = dfd_attention_mfma_try(qkv, ld_qkv, out, ld_out, n_frames, tokens, heads, scale, st);
    if (rc <= 0) return rc;
  }
  const dim3 grid(n_frames * heads), block(256);
  if (dtype == DFD_F32) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_rows_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_rows_kernel<float>), grid, block, lds, st, static_cast<const float*>(qkv), ld_qkv,
                       static_cast<float*>(out), ld_out, tokens, heads, scale);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_rows_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_rows_kernel<bf16_t>), grid, block, lds, st, static_cast<const bf16_t*>(qkv), ld_qkv,
                       static_cast<bf16_t*>(out), ld_out, tokens, heads, scale);
  }
  DFD_CHECK_LAUNCH("dfd_attention_fwd");
  return DFD_OK;
}
