// Tuned bf16 GEMM for the encoder's large shapes (placeholder until the 256x256 kernel lands).
#include "common.hpp"
struct GemmArgs;
int dfd_gemm256_try(const GemmArgs&, int, int, hipStream_t) { return 1; }
