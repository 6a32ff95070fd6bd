// dfd_dropout — out = dropout(in) elementwise over n logical elements (mask definition: dropout.hpp).
// Forward and backward are the same map (y = m·s·x, dx = m·s·dy), so one entry point serves both; the mask
// is regenerated from (seed, step, site, element index), never stored.  Used where no producing kernel can
// carry the mask (the decoder's drop_pre, gradients entering a dropped tensor); the larger sites are fused
// into their producers (dfd_quickgelu, dfd_head_fwd, dfd_gemm RESIDUAL_POS).
#include "dropout.hpp"

namespace {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void dropout_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n, DfdDrop d) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // group of 8 elements
  const int64_t e0 = g * 8;
  if (e0 >= n) return;
  float v[8];
  if (e0 + 8 <= n) {
    if constexpr (sizeof(TI) == 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(in + e0), b = *reinterpret_cast<const f32x4*>(in + e0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    } else {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(in + e0);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
    }
    dfd_drop_eight(d, (uint64_t)e0, v);
    if constexpr (sizeof(TO) == 4) {
      f32x4 a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
      *reinterpret_cast<f32x4*>(out + e0) = a;
      *reinterpret_cast<f32x4*>(out + e0 + 4) = b;
    } else {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x8*>(out + e0) = o;
    }
  } else {
    for (int j = 0; j < 8; ++j) v[j] = e0 + j < n ? to_f32(in[e0 + j]) : 0.f;
    dfd_drop_eight(d, (uint64_t)e0, v);
    for (int j = 0; j < 8 && e0 + j < n; ++j) out[e0 + j] = from_f32<TO>(v[j]);
  }
}

}  // namespace

extern "C" int dfd_dropout(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, const dfd_dropout_t* drop,
                           void* stream) {
  DFD_REQUIRE(in && out && n >= 0, "dfd_dropout: bad arguments");
  DFD_REQUIRE((in_dtype == DFD_F32 || in_dtype == DFD_BF16) && (out_dtype == DFD_F32 || out_dtype == DFD_BF16), "dfd_dropout: dtype");
  DFD_REQUIRE(dfd_aligned16(in) && dfd_aligned16(out), "dfd_dropout: buffers must be 16-byte aligned");
  DFD_REQUIRE(drop && drop->p >= 0.f && drop->p < 1.f, "dfd_dropout: needs a dropout descriptor with 0 <= p < 1");
  DFD_REQUIRE(drop->p == 0.f || drop->rng_state, "dfd_dropout: p > 0 needs rng_state");
  if (n == 0) return DFD_OK;
  const DfdDrop d = dfd_make_drop(drop);
  const dim3 grid((unsigned)(((n + 7) / 8 + 255) / 256)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (in_dtype == DFD_F32 && out_dtype == DFD_F32)
    hipLaunchKernelGGL((dropout_kernel<float, float>), grid, block, 0, st, static_cast<const float*>(in), static_cast<float*>(out), n, d);
  else if (in_dtype == DFD_F32)
    hipLaunchKernelGGL((dropout_kernel<float, bf16_t>), grid, block, 0, st, static_cast<const float*>(in), static_cast<bf16_t*>(out), n, d);
  else if (out_dtype == DFD_F32)
    hipLaunchKernelGGL((dropout_kernel<bf16_t, float>), grid, block, 0, st, static_cast<const bf16_t*>(in), static_cast<float*>(out), n, d);
  else
    hipLaunchKernelGGL((dropout_kernel<bf16_t, bf16_t>), grid, block, 0, st, static_cast<const bf16_t*>(in), static_cast<bf16_t*>(out), n, d);
  DFD_CHECK_LAUNCH("dfd_dropout");
  return DFD_OK;
}
