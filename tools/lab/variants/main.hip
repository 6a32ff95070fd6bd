// Lab: scheduling variants of the persistent GEMM's K step (generated copies of gemm256p.hip, see gen in tools/lab) vs the product
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../../dfd-clip_amd/csrc/gemm_args.hpp"
int dfd_gemm256p_try_v1(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);
int dfd_gemm256p_try_v2(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);
int dfd_gemm256p_try_v3(const GemmArgs& a, int c_dtype, int epi, hipStream_t st);
void dfd_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int main() {
  const int64_t M = 480 * 197;
  struct Shape { const char* name; int N, K, epi; } shapes[] = {{"c_fc", 3072, 768, DFD_EPI_BIAS_QUICKGELU}, {"qkv", 2304, 768, DFD_EPI_BIAS}, {"c_proj", 768, 3072, DFD_EPI_BIAS}};
  typedef int (*fn_t)(const GemmArgs&, int, int, hipStream_t);
  struct Var { const char* name; fn_t fn; } vars[] = {{"product", dfd_gemm256p_try}, {"v1 W in P0+P1", dfd_gemm256p_try_v1}, {"v2 W with A P3", dfd_gemm256p_try_v2},
                                                      {"v3 = product", dfd_gemm256p_try_v3}, {"product", dfd_gemm256p_try}, {"v1 W in P0+P1", dfd_gemm256p_try_v1},
                                                      {"v2 W with A P3", dfd_gemm256p_try_v2}, {"v3 = product", dfd_gemm256p_try_v3}};
  for (auto& sh : shapes) {
    void *A, *W, *C; float* bias;
    hipMalloc(&A, M * sh.K * 2); hipMalloc(&W, (size_t)sh.N * sh.K * 2); hipMalloc(&C, M * sh.N * 2); hipMalloc(&bias, sh.N * 4);
    std::vector<unsigned short> h((size_t)M * sh.K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    hipMemcpy(A, h.data(), M * sh.K * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, sh.N * 4);
    GemmArgs a{}; a.A = A; a.W = W; a.C = C; a.bias = bias; a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.M = M; a.N = sh.N; a.K = sh.K; a.stream_out = 1;
    std::vector<unsigned short> c0((size_t)M * sh.N), c1((size_t)M * sh.N);
    vars[0].fn(a, DFD_BF16, sh.epi, 0); hipDeviceSynchronize(); hipMemcpy(c0.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
    for (int vi = 1; vi < 4; ++vi) {
      hipMemset(C, 0xff, M * sh.N * 2); vars[vi].fn(a, DFD_BF16, sh.epi, 0); hipError_t e = hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
      size_t bad = 0; for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
      printf("%-8s %-16s sync=%s mismatches=%zu\n", sh.name, vars[vi].name, hipGetErrorString(e), bad);
    }
    for (auto& v : vars) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) v.fn(a, DFD_BF16, sh.epi, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) v.fn(a, DFD_BF16, sh.epi, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
      printf("%-8s %-16s %.3f ms  %6.0f TF\n", sh.name, v.name, ms, 2.0 * M * sh.N * sh.K / ms / 1e9);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(bias);
  }
  return 0;
}
