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
