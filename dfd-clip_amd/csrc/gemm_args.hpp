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
  int stream_out; // DFD_GEMM_STREAM_OUT: non-temporal output stores
  int spare_cus;  // persistent kernel: compute units left free for other streams
  int spare_if_free;  // DFD_GEMM_SPARE_IF_FREE: ... only where that costs no extra round of tiles
  int tile_rows;  // persistent kernel: 0 = choose, 224 / 256 = force that tile height (lab, tests)
  DfdDrop drop;   // RESIDUAL_POS: dropout on the accumulator (element index row*N + col); thr16 == 0: none
  FastDiv div_tokens, div_frames;  // persistent kernel, QKV_EXPORT: row -> (frame, token), frame -> frame % T
  const float* col_scale;          // fp8 operands: per output column, activation scale x weight-row scale
  float out_inv_scale;             // fp8 output: stored value = e4m3(result * out_inv_scale)
  // gemm256e.hip, dynamic tile hand-out (set by its launcher): eight monotonic counters, one 128-byte line per XCD label
  // (blockIdx & 7), and the value each held when this launch was enqueued; NULL = tiles are dealt statically
  uint32_t* sched;
  uint32_t sched_base[8];
  int no_dynamic;  // tests / A-B: 1 = keep the static order
};

// tuned bf16 kernels: 0 = launched, <0 = error, 1 = shape / epilogue not eligible
int dfd_gemm256_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);   // gemm256.hip: one workgroup per tile
int dfd_gemm256p_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);  // gemm256p.hip: persistent, bf16 C
int dfd_gemm256p_f8_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);  // same kernel, e4m3 operands (dfd_gemm_fp8)
int dfd_gemm256e_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);     // gemm256e.hip: persistent, ping-pong K loop (tried first)
int dfd_gemm256e_f8_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);  // same kernel, e4m3 operands
