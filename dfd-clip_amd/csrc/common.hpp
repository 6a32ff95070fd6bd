// Shared device/host helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dfdclip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short8 __attribute__((ext_vector_type(8)));

#define DFD_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------
void dfd_set_error(const char* fmt, ...);

#define DFD_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      dfd_set_error(__VA_ARGS__);         \
      return DFD_ERR_INVALID_ARG;         \
    }                                     \
  } while (0)

#define DFD_CHECK_LAUNCH(name)                                            \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      dfd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return DFD_ERR_LAUNCH;                                              \
    }                                                                     \
  } while (0)

static inline bool dfd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// u * sigmoid(1.702 u) = u / (1 + 2^(-1.702*log2(e)*u)); v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an
// IEEE division sequence.  Every kernel uses exactly this operation order (gemm256.hip spells it out in
// packed form), so the tuned and the general GEMM agree bit for bit.
#define DFD_QUICKGELU_SCALE (-2.45546696f)
__device__ __forceinline__ float quick_gelu(float u) {
  return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * DFD_QUICKGELU_SCALE));
}

// floor(n / d) for 0 <= n < 2^31 as one multiply-high and a shift: mul = ceil(2^(31+s) / d), s = ceil(log2 d)
struct FastDiv {
  uint32_t mul, shift, one;
  static FastDiv make(uint32_t d) {
    if (d <= 1) return FastDiv{0u, 0u, 1u};
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    const uint64_t m = ((1ull << (31 + s)) + d - 1) / d;
    return FastDiv{(uint32_t)m, s - 1, 0u};
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return one ? n : (__umulhi(n, mul) >> shift); }
};

// Where the decoder's keys / values live (dfd_kv_layout_t of the C ABI, resolved): element (frame f, patch p,
// channel c) at base + f*frame_stride + p*row_stride + c; `pos` [T, D] f32 (may be NULL) is added to every key and
// value row of frame f % T as it is read.
struct KvLayout {
  int64_t row_stride, frame_stride;
  const float* pos;
};
static inline KvLayout dfd_kv_layout(const dfd_kv_layout_t* l, int patches, int D) {
  if (l == nullptr) return KvLayout{(int64_t)D, (int64_t)patches * D, nullptr};
  return KvLayout{l->row_stride, l->frame_stride, l->pos};
}
