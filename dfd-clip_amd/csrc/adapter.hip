// CompInvAdapter middle stage (reference src/models.py:823-875): y = GELU_erf(LayerNorm(a)) over a
// [frames, patches, x] tensor.  "nln" normalises the whole (patches, x) slab of a frame jointly with a
// [patches, x] affine (models.py:831); "ln"/"z0" normalise each row of x (models.py:847, :860).
// HBM-bound: one read and one write of the tensor (slab kept in registers between the passes when it
// fits; the statistics pass re-reads from L2 otherwise).  fp32 statistics, two-pass variance.
#include "common.hpp"

namespace {

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <typename T> __device__ __forceinline__ float ld(const T* p) { return to_f32(*p); }

template <typename T> struct Ld8;
template <> struct Ld8<float> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
  }
};
template <> struct Ld8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
  }
};
template <typename T> struct St8;
template <> struct St8<float> {
  static __device__ __forceinline__ void store(float* p, const float* o) {
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
  }
};
template <> struct St8<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)o[e];
    *reinterpret_cast<bf16x8*>(p) = r;
  }
};

// one workgroup per frame (joint) — slab = patches * x elements
template <typename TA, typename T>
__global__ __launch_bounds__(1024) void adapter_nln_kernel(const TA* __restrict__ a, T* __restrict__ y,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           int slab, float eps) {
  __shared__ float sc[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const TA* ap = a + (int64_t)blockIdx.x * slab;
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

// Register-resident form of the joint ("nln") kernel: the frame's (patches, x) slab is read ONCE, 8
// elements (16 bytes of bf16) per load, held in registers across the two statistics passes and the
// affine + GELU pass.  Needs slab % 8 == 0 and slab <= 1024 * 8 * NCH.
template <typename TA, typename T, int NCH>
__global__ __launch_bounds__(1024) void adapter_nln_reg_kernel(const TA* __restrict__ a, T* __restrict__ y,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               int slab, float eps) {
  __shared__ float sc[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TA* ap = a + (int64_t)blockIdx.x * slab;
  T* yp = y + (int64_t)blockIdx.x * slab;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sc[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sc[i];
    return t;
  };
  const int nchunk = slab >> 3;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 1024;
    if (ch < nchunk) {
      Ld8<TA>::load(ap + (int64_t)ch * 8, v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e];
    }
  }
  const float mean = block_sum(s) / (float)slab;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (tid + c * 1024 < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(block_sum(q) / (float)slab + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 1024;
    if (ch < nchunk) {
      float ww[8], bb[8], o[8];
      Ld8<float>::load(w + (int64_t)ch * 8, ww);
      Ld8<float>::load(b + (int64_t)ch * 8, bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = gelu_erf((v[c][e] - mean) * rstd * ww[e] + bb[e]);
      St8<T>::store(yp + (int64_t)ch * 8, o);
    }
  }
}

// one wave per row of x
template <typename TA, typename T>
__global__ __launch_bounds__(256) void adapter_ln_kernel(const TA* __restrict__ a, T* __restrict__ y, const float* __restrict__ w,
                                                         const float* __restrict__ b, int64_t rows, int x, float eps,
                                                         int gelu_first) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TA* ap = a + row * x;
  T* yp = y + row * x;
  // gelu_first ("768-x-768" / "legacy-768-x-768", models.py:795-821): y = LayerNorm(GELU(a)); else GELU(LayerNorm(a))
  auto in = [&](int i) { const float v = ld(ap + i); return gelu_first ? gelu_erf(v) : v; };
  float s = 0.f;
  for (int i = lane; i < x; i += 64) s += in(i);
  const float mean = wave_sum(s) / (float)x;
  float q = 0.f;
  for (int i = lane; i < x; i += 64) { const float d = in(i) - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)x + eps);
  for (int i = lane; i < x; i += 64) {
    const float z = (in(i) - mean) * rstd * w[i] + b[i];
    yp[i] = from_f32<T>(gelu_first ? z : gelu_erf(z));
  }
}

__device__ __forceinline__ float gelu_erf_grad(float z) {
  return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}

// x == 256 fast paths of the per-row modes (0: GELU(LN(a)), 2: LN(GELU(a))): one wave per row, each lane owns 4
// consecutive elements loaded once with one 8- or 16-byte access.
template <typename T> struct Ld4;
template <> struct Ld4<float> {
  static __device__ __forceinline__ void load(const float* p, float* o) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e];
  }
  static __device__ __forceinline__ void store(float* p, const float* o) { *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]}; }
};
template <> struct Ld4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (float)a[e];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (bf16_t)o[e];
    *reinterpret_cast<bf16x4*>(p) = r;
  }
};

template <typename TA, typename T>
__global__ __launch_bounds__(256) void adapter_row256_kernel(const TA* __restrict__ a, T* __restrict__ y,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             int64_t rows, float eps, int gelu_first) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[4], ww[4], bb[4], o[4];
  Ld4<TA>::load(a + row * 256 + lane * 4, v);
  Ld4<float>::load(w + lane * 4, ww);
  Ld4<float>::load(b + lane * 4, bb);
  if (gelu_first) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
  }
  const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / 256.0f) + eps);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float z = (v[e] - mean) * rstd * ww[e] + bb[e];
    o[e] = gelu_first ? z : gelu_erf(z);
  }
  Ld4<T>::store(y + row * 256 + lane * 4, o);
}

// backward group pass for the same modes and width: da and the row's (mean, rstd)
template <typename TA, typename T>
__global__ __launch_bounds__(256) void adapter_bwd_row256_kernel(const TA* __restrict__ a, const T* __restrict__ dy,
                                                                 T* __restrict__ da, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ stats,
                                                                 int64_t rows, float eps, int gelu_first) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float av[4], v[4], dd[4], ww[4], bb[4], g[4], o[4];
  Ld4<TA>::load(a + row * 256 + lane * 4, av);
  Ld4<T>::load(dy + row * 256 + lane * 4, dd);
  Ld4<float>::load(w + lane * 4, ww);
  Ld4<float>::load(b + lane * 4, bb);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = gelu_first ? gelu_erf(av[e]) : av[e];
  const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / 256.0f) + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float nh = (v[e] - mean) * rstd;
    v[e] = nh;
    g[e] = dd[e] * (gelu_first ? 1.0f : gelu_erf_grad(nh * ww[e] + bb[e])) * ww[e];
    s1 += g[e];
    s2 += g[e] * nh;
  }
  const float m1 = wave_sum(s1) * (1.0f / 256.0f), m2 = wave_sum(s2) * (1.0f / 256.0f);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float du = rstd * (g[e] - m1 - v[e] * m2);
    o[e] = gelu_first ? du * gelu_erf_grad(av[e]) : du;
  }
  Ld4<T>::store(da + row * 256 + lane * 4, o);
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// ---- backward, pass 1: one workgroup per normalisation group (frame slab for "nln", row for "ln") -------
// da = rstd * (g - mean(g) - nhat * mean(g * nhat)),  g = dy * gelu'(nhat*w + b) * w;  saves (mean, rstd)
template <typename TA, typename T>
__global__ __launch_bounds__(1024) void adapter_bwd_group_kernel(const TA* __restrict__ a, const T* __restrict__ dy,
                                                                 T* __restrict__ da, const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ stats,
                                                                 int group, int affine_period, float eps, int gelu_first) {
  __shared__ float sc[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const TA* ap = a + (int64_t)blockIdx.x * group;
  const T* dp = dy + (int64_t)blockIdx.x * group;
  T* op = da + (int64_t)blockIdx.x * group;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sc[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sc[i];
    return t;
  };
  // gelu_first: the normalised quantity is u = GELU(a); dL/du = LN backward of dy*w, dL/da = dL/du * GELU'(a)
  auto in = [&](int i) { const float v = ld(ap + i); return gelu_first ? gelu_erf(v) : v; };
  float s = 0.f;
  for (int i = tid; i < group; i += blockDim.x) s += in(i);
  const float mean = block_sum(s) / (float)group;
  float q = 0.f;
  for (int i = tid; i < group; i += blockDim.x) { const float d = in(i) - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum(q) / (float)group + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int i = tid; i < group; i += blockDim.x) {
    const int e = i % affine_period;
    const float nh = (in(i) - mean) * rstd;
    const float g = ld(dp + i) * (gelu_first ? 1.0f : gelu_erf_grad(nh * w[e] + b[e])) * w[e];
    s1 += g;
    s2 += g * nh;
  }
  const float m1 = block_sum(s1) / (float)group, m2 = block_sum(s2) / (float)group;
  for (int i = tid; i < group; i += blockDim.x) {
    const int e = i % affine_period;
    const float nh = (in(i) - mean) * rstd;
    const float g = ld(dp + i) * (gelu_first ? 1.0f : gelu_erf_grad(nh * w[e] + b[e])) * w[e];
    const float du = rstd * (g - m1 - nh * m2);
    op[i] = from_f32<T>(gelu_first ? du * gelu_erf_grad(ld(ap + i)) : du);
  }
  if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

// Register-resident form of the group pass for the joint ("nln") mode: `a` is read once with 16-byte
// loads and kept normalised in registers; dy / weight / bias are streamed twice (second time from L2).
template <typename TA, typename T, int NCH>
__global__ __launch_bounds__(1024) void adapter_bwd_group_reg_kernel(const TA* __restrict__ a, const T* __restrict__ dy,
                                                                     T* __restrict__ da, const float* __restrict__ w,
                                                                     const float* __restrict__ b, float* __restrict__ stats,
                                                                     int group, float eps) {
  __shared__ float sc[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TA* ap = a + (int64_t)blockIdx.x * group;
  const T* dp = dy + (int64_t)blockIdx.x * group;
  T* op = da + (int64_t)blockIdx.x * group;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sc[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sc[i];
    return t;
  };
  const int nchunk = group >> 3;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 1024;
    if (ch < nchunk) {
      Ld8<TA>::load(ap + (int64_t)ch * 8, v[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[c][e];
    }
  }
  const float mean = block_sum(s) / (float)group;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (tid + c * 1024 < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(block_sum(q) / (float)group + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 1024;
    if (ch < nchunk) {
      float dd[8], ww[8], bb[8];
      Ld8<T>::load(dp + (int64_t)ch * 8, dd);
      Ld8<float>::load(w + (int64_t)ch * 8, ww);
      Ld8<float>::load(b + (int64_t)ch * 8, bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float nh = (v[c][e] - mean) * rstd;
        v[c][e] = nh;  // keep the normalised value
        const float g = dd[e] * gelu_erf_grad(nh * ww[e] + bb[e]) * ww[e];
        s1 += g;
        s2 += g * nh;
      }
    }
  }
  const float m1 = block_sum(s1) / (float)group, m2 = block_sum(s2) / (float)group;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = tid + c * 1024;
    if (ch < nchunk) {
      float dd[8], ww[8], bb[8], o[8];
      Ld8<T>::load(dp + (int64_t)ch * 8, dd);
      Ld8<float>::load(w + (int64_t)ch * 8, ww);
      Ld8<float>::load(b + (int64_t)ch * 8, bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float nh = v[c][e];
        const float g = dd[e] * gelu_erf_grad(nh * ww[e] + bb[e]) * ww[e];
        o[e] = rstd * (g - m1 - nh * m2);
      }
      St8<T>::store(op + (int64_t)ch * 8, o);
    }
  }
  if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

// ---- backward, pass 2: affine gradients.  Thread per affine element e, block-row per chunk of groups;
// partial sums go to slab blockIdx.y and are added in fixed order by adapter_bwd_affine_reduce_kernel.
template <typename TA, typename T>
__global__ __launch_bounds__(256) void adapter_bwd_affine_kernel(const TA* __restrict__ a, const T* __restrict__ dy,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 const float* __restrict__ stats, float* __restrict__ part,
                                                                 int64_t groups, int group, int affine, int groups_per_slab,
                                                                 int gelu_first) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= affine) return;
  const int reps = group / affine;  // rows of an affine period inside one group (1 for nln, 1 for ln)
  const int64_t g0 = (int64_t)blockIdx.y * groups_per_slab;
  const int64_t g1 = g0 + groups_per_slab < groups ? g0 + groups_per_slab : groups;
  const float we = w[e], be = b[e];
  float dw = 0.f, db = 0.f;
  for (int64_t gi = g0; gi < g1; ++gi) {
    const float mean = stats[2 * gi], rstd = stats[2 * gi + 1];
    for (int r = 0; r < reps; ++r) {
      const int64_t idx = gi * group + (int64_t)r * affine + e;
      const float av = ld(a + idx);
      const float nh = ((gelu_first ? gelu_erf(av) : av) - mean) * rstd;
      const float dz = ld(dy + idx) * (gelu_first ? 1.0f : gelu_erf_grad(nh * we + be));
      dw = fmaf(dz, nh, dw);
      db += dz;
    }
  }
  part[((int64_t)blockIdx.y * 2) * affine + e] = dw;
  part[((int64_t)blockIdx.y * 2 + 1) * affine + e] = db;
}

__global__ void adapter_bwd_affine_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                 int affine, int slabs) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= affine) return;
  float s0 = 0.f, s1 = 0.f;
  for (int z = 0; z < slabs; ++z) { s0 += part[((int64_t)z * 2) * affine + e]; s1 += part[((int64_t)z * 2 + 1) * affine + e]; }
  dw[e] = s0;
  db[e] = s1;
}

int affine_slabs(int64_t groups, int affine) {
  // enough (element, chunk) threads to cover the chip; at most 512 slabs
  int64_t want = (262144 + affine - 1) / affine;
  if (want > groups) want = groups;
  if (want > 512) want = 512;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace

extern "C" size_t dfd_adapter_norm_gelu_bwd_workspace(int frames, int patches, int x, int joint) {
  if (frames <= 0 || patches <= 0 || x <= 0) return 0;
  const bool jt = joint == 1;
  const int64_t groups = jt ? frames : (int64_t)frames * patches;
  const int affine = jt ? patches * x : x;
  return (size_t)groups * 2 * sizeof(float) + (size_t)affine_slabs(groups, affine) * 2 * affine * sizeof(float);
}

extern "C" int dfd_adapter_norm_gelu_bwd(const void* a, int a_dtype, const void* dy, void* da, int dtype, const float* weight,
                                         const float* bias, float* dweight, float* dbias, void* workspace, int frames,
                                         int patches, int x, int joint, float eps, void* stream) {
  DFD_REQUIRE(a && dy && da && weight && bias && dweight && dbias && workspace, "dfd_adapter_norm_gelu_bwd: null pointer");
  DFD_REQUIRE(frames > 0 && patches > 0 && x > 0, "dfd_adapter_norm_gelu_bwd: bad shape");
  DFD_REQUIRE(da != dy && da != a, "dfd_adapter_norm_gelu_bwd: da must not alias dy or a (both are re-read by the affine pass)");
  DFD_REQUIRE(dtype == DFD_F32 || dtype == DFD_BF16, "dfd_adapter_norm_gelu_bwd: dtype=%d", dtype);
  DFD_REQUIRE(a_dtype == dtype || a_dtype == DFD_F32, "dfd_adapter_norm_gelu_bwd: a_dtype=%d must be dtype or f32", a_dtype);
  DFD_REQUIRE(joint >= 0 && joint <= 2, "dfd_adapter_norm_gelu_bwd: mode=%d", joint);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int gelu_first = joint == 2;
  joint = joint == 1;
  const int64_t groups = joint ? frames : (int64_t)frames * patches;
  const int group = joint ? patches * x : x;
  const int affine = group;
  float* stats = static_cast<float*>(workspace);
  float* part = stats + groups * 2;
  const int slabs = affine_slabs(groups, affine);
  const int gps = (int)((groups + slabs - 1) / slabs);
  const int threads = joint ? 1024 : 64;
  const bool reg_ok = joint && group % 8 == 0 && group <= 1024 * 8 * 7 && dfd_aligned16(a) && dfd_aligned16(dy) &&
                      dfd_aligned16(da) && dfd_aligned16(weight) && dfd_aligned16(bias);
  const bool row256 = !joint && group == 256 && dfd_aligned16(a) && dfd_aligned16(dy) && dfd_aligned16(da) &&
                      dfd_aligned16(weight) && dfd_aligned16(bias);
#define ADP_LAUNCH(TA, T)                                                                                                 \
  if (row256)                                                                                                             \
    hipLaunchKernelGGL((adapter_bwd_row256_kernel<TA, T>), dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, st,             \
                       static_cast<const TA*>(a), static_cast<const T*>(dy), static_cast<T*>(da), weight, bias, stats,      \
                       groups, eps, gelu_first);                                                                          \
  else if (reg_ok)                                                                                                        \
    hipLaunchKernelGGL((adapter_bwd_group_reg_kernel<TA, T, 7>), dim3((unsigned)groups), dim3(1024), 0, st,                  \
                       static_cast<const TA*>(a), static_cast<const T*>(dy), static_cast<T*>(da), weight, bias, stats,      \
                       group, eps);                                                                                       \
  else                                                                                                                    \
    hipLaunchKernelGGL((adapter_bwd_group_kernel<TA, T>), dim3((unsigned)groups), dim3(threads), 0, st,                      \
                       static_cast<const TA*>(a), static_cast<const T*>(dy), static_cast<T*>(da), weight, bias, stats,      \
                       group, affine, eps, gelu_first);                                                                   \
  hipLaunchKernelGGL((adapter_bwd_affine_kernel<TA, T>), dim3((affine + 255) / 256, slabs), dim3(256), 0, st,                \
                     static_cast<const TA*>(a), static_cast<const T*>(dy), weight, bias, stats, part, groups, group, affine, \
                     gps, gelu_first);
  if (dtype == DFD_F32) { ADP_LAUNCH(float, float) }
  else if (a_dtype == DFD_F32) { ADP_LAUNCH(float, bf16_t) }
  else { ADP_LAUNCH(bf16_t, bf16_t) }
#undef ADP_LAUNCH
  hipLaunchKernelGGL(adapter_bwd_affine_reduce_kernel, dim3((affine + 255) / 256), dim3(256), 0, st, part, dweight, dbias, affine, slabs);
  DFD_CHECK_LAUNCH("dfd_adapter_norm_gelu_bwd");
  return DFD_OK;
}

extern "C" int dfd_adapter_norm_gelu(const void* a, int a_dtype, void* y, int dtype, const float* weight, const float* bias, int frames,
                                     int patches, int x, int joint, float eps, void* stream) {
  DFD_REQUIRE(a && y && weight && bias, "dfd_adapter_norm_gelu: null pointer");
  DFD_REQUIRE(frames >= 0 && patches > 0 && x > 0, "dfd_adapter_norm_gelu: bad shape");
  DFD_REQUIRE(dtype == DFD_F32 || dtype == DFD_BF16, "dfd_adapter_norm_gelu: dtype=%d", dtype);
  DFD_REQUIRE(a_dtype == dtype || a_dtype == DFD_F32, "dfd_adapter_norm_gelu: a_dtype=%d must be dtype or f32", a_dtype);
  DFD_REQUIRE(a_dtype == dtype || a != y, "dfd_adapter_norm_gelu: in place needs one dtype");
  DFD_REQUIRE(joint >= 0 && joint <= 2, "dfd_adapter_norm_gelu: mode=%d", joint);
  if (frames == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int gelu_first = joint == 2;
  const int64_t rows = (int64_t)frames * patches;
  const dim3 grid((unsigned)((rows + 3) / 4));
  const int slab_all = patches * x;
  // register-resident slab: 16-byte loads need slab % 8 == 0 and aligned bases; 7 chunks x 1024 threads x 8
  const bool reg_ok = slab_all % 8 == 0 && slab_all <= 1024 * 8 * 7 && dfd_aligned16(a) && dfd_aligned16(y) &&
                      dfd_aligned16(weight) && dfd_aligned16(bias);
  const bool row_ok = dfd_aligned16(a) && dfd_aligned16(y) && dfd_aligned16(weight) && dfd_aligned16(bias);
#define FWD_LAUNCH(TA, T)                                                                                              \
  do {                                                                                                                 \
    if (joint == 1 && reg_ok)                                                                                          \
      hipLaunchKernelGGL((adapter_nln_reg_kernel<TA, T, 7>), dim3(frames), dim3(1024), 0, st, static_cast<const TA*>(a), \
                         static_cast<T*>(y), weight, bias, patches * x, eps);                                          \
    else if (joint == 1)                                                                                               \
      hipLaunchKernelGGL((adapter_nln_kernel<TA, T>), dim3(frames), dim3(1024), 0, st, static_cast<const TA*>(a),        \
                         static_cast<T*>(y), weight, bias, patches * x, eps);                                          \
    else if (joint != 1 && x == 256 && row_ok)                                                                         \
      hipLaunchKernelGGL((adapter_row256_kernel<TA, T>), grid, dim3(256), 0, st, static_cast<const TA*>(a),              \
                         static_cast<T*>(y), weight, bias, rows, eps, gelu_first);                                     \
    else                                                                                                               \
      hipLaunchKernelGGL((adapter_ln_kernel<TA, T>), grid, dim3(256), 0, st, static_cast<const TA*>(a), static_cast<T*>(y), \
                         weight, bias, rows, x, eps, gelu_first);                                                      \
  } while (0)
  if (dtype == DFD_F32) FWD_LAUNCH(float, float);
  else if (a_dtype == DFD_F32) FWD_LAUNCH(float, bf16_t);
  else FWD_LAUNCH(bf16_t, bf16_t);
#undef FWD_LAUNCH
  DFD_CHECK_LAUNCH("dfd_adapter_norm_gelu");
  return DFD_OK;
}
