// Fused SGD step over a list of f32 parameters (reference src/models.py:740-754: torch.optim.SGD with momentum 0.95 and
// the config's weight decay over the trainable parameters; src/trainer.py:157-177 calls it once per step).  ONE launch
// instead of torch's ten multi-tensor launches, and for the decoder's Linear weights the launch also rewrites the
// transposed f32 copy the row-streaming linear kernels read (dfd_linear_rows_t): the 20 transposes a training step
// otherwise needs after every update disappear.
//
// Arithmetic, per element, in torch's order (torch/optim/sgd.py, foreach path):
//     g   = grad + wd * p                (one fused multiply-add)
//     buf = first step ? g : momentum * buf + g      (product rounded, then the sum)
//     p   = p - lr * buf                 (one fused multiply-add)
// Workgroups are dealt to parameters through a table in device memory (binary search on the first block of each entry):
// an entry is either a flat run of 1,024 elements per block or, for a mirrored [rows, cols] weight, 32 x 32 tiles whose
// transposed image goes through LDS so that both the parameter and its mirror are written in whole row segments.
#include "common.hpp"

namespace {

struct SgdEntry {      // mirrors dfd_sgd_param of the C ABI
  float* p;
  const float* g;
  float* buf;
  float* mirror;       // [cols, rows] transposed copy, or NULL
  int64_t numel;
  int32_t rows, cols;  // only read when mirror != NULL
  int64_t first_block;
};

__global__ __launch_bounds__(256) void sgd_step_kernel(const SgdEntry* __restrict__ table, int n, float lr, float momentum, float wd, int first) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.x;
  int lo = 0, hi = n - 1;  // last entry whose first_block <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid;
    else hi = mid - 1;
  }
  const SgdEntry e = table[lo];
  const int64_t lb = b - e.first_block;
  auto update = [&](int64_t i) {
    const float p = e.p[i];
    const float g = __builtin_fmaf(wd, p, e.g[i]);
    float m = g;
    if (!first) m = momentum * e.buf[i] + g;
    e.buf[i] = m;
    const float np = __builtin_fmaf(-lr, m, p);
    e.p[i] = np;
    return np;
  };
  if (e.mirror == nullptr) {
    const int64_t i0 = lb * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * 256;
      if (i < e.numel) update(i);
    }
    return;
  }
  const int tiles_c = (e.cols + 31) / 32;
  const int tr = (int)(lb / tiles_c), tc = (int)(lb - (int64_t)tr * tiles_c);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads, four rows each
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
    if (r < e.rows && c < e.cols) tile[ty + 8 * k][tx] = update((int64_t)r * e.cols + c);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tc * 32 + ty + 8 * k, r = tr * 32 + tx;  // mirror row = parameter column
    if (r < e.rows && c < e.cols) e.mirror[(int64_t)c * e.rows + r] = tile[tx][ty + 8 * k];
  }
}

}  // namespace

static_assert(sizeof(SgdEntry) == sizeof(dfd_sgd_param), "dfd_sgd_param layout");

extern "C" int64_t dfd_sgd_blocks(int64_t numel, int rows, int cols, int mirrored) {
  if (mirrored) return (int64_t)((rows + 31) / 32) * ((cols + 31) / 32);
  return (numel + 1023) / 1024;
}

extern "C" int dfd_sgd_step(const dfd_sgd_param* table_dev, int n, int64_t total_blocks, float lr, float momentum, float weight_decay,
                            int first_step, void* stream) {
  DFD_REQUIRE(table_dev != nullptr && n > 0, "dfd_sgd_step: empty table");
  DFD_REQUIRE(total_blocks > 0 && total_blocks < ((int64_t)1 << 31), "dfd_sgd_step: total_blocks = %lld", (long long)total_blocks);
  hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const SgdEntry*>(table_dev), n, lr, momentum, weight_decay, first_step);
  DFD_CHECK_LAUNCH("dfd_sgd_step");
  return DFD_OK;
}
