// Lab: the product's two tuned bf16 GEMM kernels (gemm256.hip = one workgroup per tile, gemm256p.hip = persistent)
// and the four-wave lab variant (gemm256q_lab.hip, "Q")
// on the four ViT-B/16 encoder shapes at B16xT30, interleaved rounds in one process; bitwise comparison first.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../dfd-clip_amd/csrc/gemm_args.hpp"

int dfd_gemm256q_launch(const GemmArgs& a, int epi, hipStream_t st);  // gemm256q_lab.hip
int dfd_gemm256e_launch(const GemmArgs& a, int epi, int depth, hipStream_t st);  // gemm256e_lab.hip (ping-pong K loop)
void dfd_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

int main(int argc, char** argv) {
  const int64_t MB = 480 * 197;
  struct Shape { const char* name; int64_t M; int N, K, epi; } shapes[] = {
      {"c_fc", MB, 3072, 768, DFD_EPI_BIAS_QUICKGELU}, {"qkv_plain", MB, 2304, 768, DFD_EPI_BIAS}, {"c_proj/d", MB, 768, 3072, DFD_EPI_BIAS},
      {"out_proj/d", MB, 768, 768, DFD_EPI_BIAS}, {"sq4096", 4096, 4096, 4096, DFD_EPI_BIAS}, {"sq8192", 8192, 8192, 8192, DFD_EPI_BIAS},
      {"L14_c_fc", 240 * 257, 4096, 1024, DFD_EPI_BIAS_QUICKGELU}};
  // persistent: 0 relaunching, 1 P (product), 2 Q (four waves), 3 E (ping-pong; `depth` segments of prefetch)
  struct Var { const char* name; int persistent, rows, stream, depth; } vars[] = {
      {"relaunch_nt", 0, 0, 1, 0}, {"P256_nt", 1, 256, 1, 0}, {"E6", 3, 256, 1, 6}, {"E6_nostores", 3, 256, 1, 86}, {"E6_noepilogue", 3, 256, 1, 166},
      {"E6_nostag", 3, 256, 1, 46}, {"E6", 3, 256, 1, 6}, {"E6_nostores", 3, 256, 1, 86}, {"E6_noepilogue", 3, 256, 1, 166}, {"P256_nt", 1, 256, 1, 0}};
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // argv[1]: index of the single shape to run
  int si = -1;
  for (auto& sh : shapes) {
    ++si;
    if (only >= 0 && si != only) continue;
    const int64_t M = sh.M;
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
      if (v.persistent == 3) return dfd_gemm256e_launch(a, sh.epi, v.depth, 0);
      return v.persistent ? dfd_gemm256p_try(a, DFD_BF16, sh.epi, 0) : dfd_gemm256_try(a, DFD_BF16, sh.epi, 0);
    };
    for (int i = 0; i < 20; ++i) run(vars[0]);
    hipDeviceSynchronize();
    std::vector<unsigned short> c0((size_t)M * sh.N), c1((size_t)M * sh.N);
    hipMemset(C, 0xff, M * sh.N * 2); run(vars[0]); hipDeviceSynchronize();
    hipMemcpy(c0.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
    for (int vi : {2}) {
      hipMemset(C, 0xff, M * sh.N * 2); int rc = run(vars[vi]); hipError_t e = hipDeviceSynchronize();
      hipMemcpy(c1.data(), C, M * sh.N * 2, hipMemcpyDeviceToHost);
      size_t bad = 0; for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
      if (bad && getenv("LAB_DIAG")) {  // where inside the 256x256 tiles, and in which tiles, do the results differ?
        size_t rb[16] = {0}, cb[16] = {0}; int shown = 0; size_t ntile_bad = 0, first_round = 0; const int tn = sh.N / 256;
        std::vector<unsigned char> tb((size_t)((M + 255) / 256) * tn, 0);
        for (size_t i = 0; i < c0.size(); ++i) if (c0[i] != c1[i]) {
          const size_t r = i / sh.N, c = i % sh.N; rb[(r % 256) / 16]++; cb[(c % 256) / 16]++; tb[(r / 256) * tn + c / 256] = 1;
          if (shown < 6) { printf("   (%zu,%zu) tile (%zu,%zu) ref %04x got %04x\n", r, c, r / 256, c / 256, c0[i], c1[i]); ++shown; }
        }
        for (size_t t = 0; t < tb.size(); ++t) ntile_bad += tb[t];
        printf("   tiles with mismatches: %zu of %zu; by 16-row block:", ntile_bad, tb.size());
        for (int k = 0; k < 16; ++k) printf(" %zu", rb[k]);
        printf("; by 16-col block:");
        for (int k = 0; k < 16; ++k) printf(" %zu", cb[k]);
        printf("\n");
      }
      printf("%-10s %-10s vs relaunch: rc=%d sync=%s mismatching elements = %zu of %zu\n", sh.name, vars[vi].name, rc, hipGetErrorString(e), bad, c0.size());
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
