// Ablation lab for gemm256.hip: the same source compiled with LAB_* switches under different
// entry names; times each variant on the four encoder shapes.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../dfd-clip_amd/csrc/gemm_args.hpp"

void dfd_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

#define DECL(n) int n(const GemmArgs&, int, int, hipStream_t);
DECL(lab_full) DECL(lab_same) DECL(lab_same_noepi) DECL(lab_noepi) DECL(lab_noglds) DECL(lab_nods) DECL(lab_nobar) DECL(lab_mfma_only) DECL(lab_noglds_nods) DECL(dfd_gemm256p_try) DECL(labp_nostore) DECL(labp_noepi) DECL(labp_a1) DECL(labp_a3) DECL(labp_a16) DECL(labp_a18) DECL(labp_nt) DECL(lab_nt) DECL(lab_nostore) DECL(lab_nogelu) DECL(lab_nostore_nogelu) DECL(lab_same_nostore)

int main() {
  const int64_t M = 480 * 197;
  struct Shape { const char* name; int N, K, epi, cdt; } shapes[] = {
      {"c_fc", 3072, 768, DFD_EPI_BIAS_QUICKGELU, DFD_BF16}, {"qkv_plain", 2304, 768, DFD_EPI_BIAS, DFD_BF16}, {"c_proj/d", 768, 3072, DFD_EPI_BIAS, DFD_BF16}, {"out_proj/d", 768, 768, DFD_EPI_BIAS, DFD_BF16}};
  struct Var { const char* name; int (*fn)(const GemmArgs&, int, int, hipStream_t); } vars[] = {
      {"full", lab_full}, {"persistent", dfd_gemm256p_try}, {"full", lab_full}, {"persistent", dfd_gemm256p_try}, {"full_nt", lab_nt}, {"P_nt", labp_nt}, {"P_aux1_sc0", labp_a1}, {"P_aux3_sc0nt", labp_a3}, {"P_aux16_sc1", labp_a16}, {"P_aux18_sc1nt", labp_a18}, {"full_nt", lab_nt}, {"P_nt", labp_nt}, {"persistent", dfd_gemm256p_try}, {"P_no_store", labp_nostore}, {"P_no_epilogue", labp_noepi}, {"no_store", lab_nostore}, {"no_epilogue", lab_noepi}, {"mfma_only", lab_mfma_only}};
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
    {  // the persistent kernel must reproduce the relaunching kernel bit for bit (same MFMA order, same epilogue arithmetic)
      std::vector<unsigned short> c0((size_t)M * sh.N), c1((size_t)M * sh.N);
      hipMemset(C, 0xff, M * sh.N * 2); lab_full(a, sh.cdt, sh.epi, 0); hipDeviceSynchronize();
      hipMemcpy(c0.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
      hipMemset(C, 0xff, M * sh.N * 2); int rc = dfd_gemm256p_try(a, sh.cdt, sh.epi, 0); hipError_t e = hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
      size_t bad = 0; for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
      printf("%-7s persistent vs full: rc=%d sync=%s mismatching elements = %zu of %zu\n", sh.name, rc, hipGetErrorString(e), bad, c0.size());
      fflush(stdout);
    }
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
