// Ablation lab for gemm256.hip: the same source compiled with LAB_* switches under different
// entry names; times each variant on the four encoder shapes.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../dfd-clip_amd/csrc/gemm_args.hpp"

void dfd_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

#define DECL(n) int n(const GemmArgs&, int, int, hipStream_t);
DECL(lab_full) DECL(lab_same) DECL(lab_same_noepi) DECL(lab_noepi) DECL(lab_noglds) DECL(lab_nods) DECL(lab_nobar) DECL(lab_mfma_only) DECL(lab_noglds_nods) DECL(lab_nostore) DECL(lab_nogelu) DECL(lab_nostore_nogelu) DECL(lab_same_nostore)

int main() {
  const int64_t M = 480 * 197;
  struct Shape { const char* name; int N, K, epi, cdt; } shapes[] = {
      {"c_fc", 3072, 768, DFD_EPI_BIAS_QUICKGELU, DFD_BF16}, {"qkv_plain", 2304, 768, DFD_EPI_BIAS, DFD_BF16}, {"c_proj/d", 768, 3072, DFD_EPI_BIAS, DFD_BF16}};
  struct Var { const char* name; int (*fn)(const GemmArgs&, int, int, hipStream_t); } vars[] = {
      {"full", lab_full}, {"no_store", lab_nostore}, {"no_gelu", lab_nogelu}, {"no_store_no_gelu", lab_nostore_nogelu}, {"same_tile_no_store", lab_same_nostore}, {"same_tile", lab_same}, {"same_tile_noepi", lab_same_noepi}, {"no_epilogue", lab_noepi}, {"no_glds", lab_noglds}, {"no_dsread", lab_nods},
      {"no_barrier", lab_nobar}, {"no_glds_no_dsread", lab_noglds_nods}, {"mfma_only", lab_mfma_only}};
  for (auto& sh : shapes) {
    void *A, *W, *C; float* bias;
    hipMalloc(&A, M * sh.K * 2); hipMalloc(&W, (size_t)sh.N * sh.K * 2); hipMalloc(&C, M * sh.N * 4); hipMalloc(&bias, sh.N * 4);
    std::vector<unsigned short> h((size_t)M * sh.K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);  // random bf16 in +-[0.0078, 0.0156)
    hipMemcpy(A, h.data(), M * sh.K * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, sh.N * 4); hipMemset(C, 0, M * sh.N * 4);
    GemmArgs a{}; a.A = A; a.W = W; a.C = C; a.bias = bias; a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.M = M; a.N = sh.N; a.K = sh.K;
    for (int i = 0; i < 20; ++i) lab_full(a, sh.cdt, sh.epi, 0);  // warm-up: clocks, caches, lazy code load (the first
    hipDeviceSynchronize();                                       // variant measured used to read ~10 % low without it)
    for (auto& v : vars) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) v.fn(a, sh.cdt, sh.epi, 0);
      hipEventRecord(e0, 0);
      const int it = 20;
      for (int i = 0; i < it; ++i) v.fn(a, sh.cdt, sh.epi, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
      printf("%-7s %-18s %.3f ms  %6.0f TF\n", sh.name, v.name, ms, 2.0 * M * sh.N * sh.K / ms / 1e9);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(bias);
  }
  return 0;
}
