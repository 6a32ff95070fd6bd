// CompInvAdapter middle stage (reference src/models.py:823-875): y = GELU_erf(LayerNorm(a)) over a
// [frames, patches, x] tensor.  "nln" normalises the whole (patches, x) slab of a frame jointly with a
// [patches, x] affine (models.py:831); "ln"/"z0" normalise each row of x (models.py:847, :860).
// HBM-bound: one read and one write of the tensor (slab kept in registers between the passes when it
// fits; the statistics pass re-reads from L2 otherwise).  fp32 statistics, two-pass variance.
#include "common.hpp"

namespace {

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <typename T> __device__ __forceinline__ float ld(const T* p) { return to_f32(*p); }

// one workgroup per frame (joint) — slab = patches * x elements
template <typename T>
__global__ __launch_bounds__(1024) void adapter_nln_kernel(const T* __restrict__ a, T* __restrict__ y,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           int slab, float eps) {
  __shared__ float sc[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const T* ap = a + (int64_t)blockIdx.x * slab;
  T* yp = y + (int64_t)blockIdx.x * slab;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sc[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sc[i];
    return t;
  };
  float s = 0.f;
  for (int i = tid; i < slab; i += blockDim.x) s += ld(ap + i);
  const float mean = block_sum(s) / (float)slab;
  float q = 0.f;
  for (int i = tid; i < slab; i += blockDim.x) { const float d = ld(ap + i) - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum(q) / (float)slab + eps);
  for (int i = tid; i < slab; i += blockDim.x) yp[i] = from_f32<T>(gelu_erf((ld(ap + i) - mean) * rstd * w[i] + b[i]));
}

// one wave per row of x
template <typename T>
__global__ __launch_bounds__(256) void adapter_ln_kernel(const T* __restrict__ a, T* __restrict__ y, const float* __restrict__ w,
                                                         const float* __restrict__ b, int64_t rows, int x, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* ap = a + row * x;
  T* yp = y + row * x;
  float s = 0.f;
  for (int i = lane; i < x; i += 64) s += ld(ap + i);
  const float mean = wave_sum(s) / (float)x;
  float q = 0.f;
  for (int i = lane; i < x; i += 64) { const float d = ld(ap + i) - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)x + eps);
  for (int i = lane; i < x; i += 64) yp[i] = from_f32<T>(gelu_erf((ld(ap + i) - mean) * rstd * w[i] + b[i]));
}

}  // namespace

extern "C" int dfd_adapter_norm_gelu(const void* a, void* y, int dtype, const float* weight, const float* bias, int frames,
                                     int patches, int x, int joint, float eps, void* stream) {
  DFD_REQUIRE(a && y && weight && bias, "dfd_adapter_norm_gelu: null pointer");
  DFD_REQUIRE(frames >= 0 && patches > 0 && x > 0, "dfd_adapter_norm_gelu: bad shape");
  DFD_REQUIRE(dtype == DFD_F32 || dtype == DFD_BF16, "dfd_adapter_norm_gelu: dtype=%d", dtype);
  if (frames == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (joint) {
    if (dtype == DFD_F32)
      hipLaunchKernelGGL((adapter_nln_kernel<float>), dim3(frames), dim3(1024), 0, st, static_cast<const float*>(a),
                         static_cast<float*>(y), weight, bias, patches * x, eps);
    else
      hipLaunchKernelGGL((adapter_nln_kernel<bf16_t>), dim3(frames), dim3(1024), 0, st, static_cast<const bf16_t*>(a),
                         static_cast<bf16_t*>(y), weight, bias, patches * x, eps);
  } else {
    const int64_t rows = (int64_t)frames * patches;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == DFD_F32)
      hipLaunchKernelGGL((adapter_ln_kernel<float>), grid, dim3(256), 0, st, static_cast<const float*>(a), static_cast<float*>(y),
                         weight, bias, rows, x, eps);
    else
      hipLaunchKernelGGL((adapter_ln_kernel<bf16_t>), grid, dim3(256), 0, st, static_cast<const bf16_t*>(a),
                         static_cast<bf16_t*>(y), weight, bias, rows, x, eps);
  }
  DFD_CHECK_LAUNCH("dfd_adapter_norm_gelu");
  return DFD_OK;
}
