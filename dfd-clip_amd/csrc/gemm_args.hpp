// Argument block shared by the GEMM kernels (gemm.hip: general shapes; gemm256.hip: tuned bf16).
#pragma once
#include "dropout.hpp"

struct GemmArgs {
  const void* A;
  const void* W;
  void* C;
  const float* bias;
  const float* pos;
  const float* cls;
  void* k_export;
  void* v_export;
  const void* residual;  // RESIDUAL_POS: source of the residual (NULL = C itself, in place)
  int64_t lda, ldw, ldc, M;
  int N, K, tokens, frames_per_clip;
  int qkv_first;  // QKV_EXPORT: first column block present (0 = q, 1 = k)
  DfdDrop drop;   // RESIDUAL_POS: dropout on the accumulator (element index row*N + col); thr16 == 0: none
};

// tuned bf16 kernels: 0 = launched, <0 = error, 1 = shape / epilogue not eligible
int dfd_gemm256_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);   // gemm256.hip: one workgroup per tile
int dfd_gemm256p_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);  // gemm256p.hip: persistent, bf16 C
