// Lab: the product's two tuned bf16 GEMM kernels (gemm256.hip = one workgroup per tile, gemm256p.hip = persistent)
// and the four-wave lab variant (gemm256q_lab.hip, "Q")
// on the four ViT-B/16 encoder shapes at B16xT30, interleaved rounds in one process; bitwise comparison first.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../dfd-clip_amd/csrc/gemm_args.hpp"

int dfd_gemm256q_launch(const GemmArgs& a, int epi, hipStream_t st);  // gemm256q_lab.hip
void dfd_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

int main() {
  const int64_t M = 480 * 197;
  struct Shape { const char* name; int N, K, epi; } shapes[] = {
      {"c_fc", 3072, 768, DFD_EPI_BIAS_QUICKGELU}, {"qkv_plain", 2304, 768, DFD_EPI_BIAS}, {"c_proj/d", 768, 3072, DFD_EPI_BIAS}, {"out_proj/d", 768, 768, DFD_EPI_BIAS}};
  struct Var { const char* name; int persistent, rows, stream; } vars[] = {
      {"relaunch", 0, 0, 0}, {"relaunch_nt", 0, 0, 1}, {"P256", 1, 256, 0}, {"P256_nt", 1, 256, 1}, {"P224_nt", 1, 224, 1}, {"Pauto_nt", 1, 0, 1},
      {"Q256_nt", 2, 256, 1}, {"Q224_nt", 2, 224, 1}, {"relaunch_nt", 0, 0, 1}, {"P256_nt", 1, 256, 1}, {"P224_nt", 1, 224, 1}, {"Q256_nt", 2, 256, 1},
      {"Q224_nt", 2, 224, 1}};
  for (auto& sh : shapes) {
    void *A, *W, *C; float* bias;
    hipMalloc(&A, M * sh.K * 2); hipMalloc(&W, (size_t)sh.N * sh.K * 2); hipMalloc(&C, M * sh.N * 2); hipMalloc(&bias, sh.N * 4);
    std::vector<unsigned short> h((size_t)M * sh.K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);  // random bf16 in +-[0.0078, 0.0156)
    hipMemcpy(A, h.data(), M * sh.K * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, sh.N * 4);
    GemmArgs a{}; a.A = A; a.W = W; a.C = C; a.bias = bias; a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.M = M; a.N = sh.N; a.K = sh.K;
    auto run = [&](const Var& v) {
      a.tile_rows = v.rows; a.stream_out = v.stream;
      if (v.persistent == 2) return dfd_gemm256q_launch(a, sh.epi, 0);
      return v.persistent ? dfd_gemm256p_try(a, DFD_BF16, sh.epi, 0) : dfd_gemm256_try(a, DFD_BF16, sh.epi, 0);
    };
    for (int i = 0; i < 20; ++i) run(vars[0]);
    hipDeviceSynchronize();
    std::vector<unsigned short> c0((size_t)M * sh.N), c1((size_t)M * sh.N);
    hipMemset(C, 0xff, M * sh.N * 2); run(vars[0]); hipDeviceSynchronize();
    hipMemcpy(c0.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
    for (int vi : {3, 4, 6, 7}) {
      hipMemset(C, 0xff, M * sh.N * 2); int rc = run(vars[vi]); hipError_t e = hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
      size_t bad = 0; for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
      printf("%-10s %-9s vs relaunch: rc=%d sync=%s mismatching elements = %zu of %zu\n", sh.name, vars[vi].name, rc, hipGetErrorString(e), bad, c0.size());
    }
    for (auto& v : vars) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) run(v);
      hipEventRecord(e0, 0);
      const int it = 20;
      for (int i = 0; i < it; ++i) run(v);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
      printf("%-10s %-12s %.3f ms  %6.0f TF\n", sh.name, v.name, ms, 2.0 * M * sh.N * sh.K / ms / 1e9);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(bias);
  }
  return 0;
}
