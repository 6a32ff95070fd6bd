// Pieces shared by the decoder's forward (decoder.hip) and backward (decoder_bwd.hip) kernels: 8-element row pieces
// (one lane's share of a 64-wide head), the exchanges inside an 8-lane head group, tanh.
#pragma once
#include "common.hpp"

namespace {

template <typename T> struct Ld8;
template <> struct Ld8<float> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  }
};
template <> struct Ld8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
  }
};

__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}
// x[u] = this lane's partial sum for row u (u = 0..7); returns the 8-lane group's total for row `sub` (the lane's
// index in its group): 7 exchanges instead of the 24 of eight butterflies
__device__ __forceinline__ float group8_reduce_scatter(const float (&x)[8], int sub) {
  float y[4], z[2];
  const bool h4 = (sub & 4) != 0, h2 = (sub & 2) != 0, h1 = (sub & 1) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = (h4 ? x[i + 4] : x[i]) + __shfl_xor(h4 ? x[i] : x[i + 4], 4, 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) z[i] = (h2 ? y[i + 2] : y[i]) + __shfl_xor(h2 ? y[i] : y[i + 2], 2, 64);
  return (h1 ? z[1] : z[0]) + __shfl_xor(h1 ? z[0] : z[1], 1, 64);
}

// four rows per trip: x[u] = this lane's partial sum for row u (u = 0..3); returns the group's total for row
// `sub & 3` (lanes j and j + 4 finish the same row): 7 exchanges instead of the 12 of four butterflies
__device__ __forceinline__ float group8_reduce_scatter4(const float (&x)[4], int sub) {
  float y[4], z[2];
  const bool h2 = (sub & 2) != 0, h1 = (sub & 1) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = x[i] + __shfl_xor(x[i], 4, 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) z[i] = (h2 ? y[i + 2] : y[i]) + __shfl_xor(h2 ? y[i] : y[i + 2], 2, 64);
  return (h1 ? z[1] : z[0]) + __shfl_xor(h1 ? z[0] : z[1], 1, 64);
}

// eight consecutive elements as loaded (bf16: 16 bytes, f32: 32 bytes), converted where they are used
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
  __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e] : b[e - 4]; }
};
template <> struct Raw8<bf16_t> {
  bf16x8 a;
  __device__ __forceinline__ void load(const bf16_t* p) { a = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ float get(int e) const { return (float)a[e]; }
};

__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf, abs error ~1e-7 around 0
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f / (e + 1.0f);
}

}  // namespace
