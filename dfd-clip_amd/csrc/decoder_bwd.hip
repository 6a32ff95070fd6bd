// Temporal decoder backward kernels.  The encoder is frozen (reference src/models.py:440, :501), so
// the only gradients of the path are the decoder's (reference src/models.py:81-361); they are what
// `accelerator.backward` produces in the reference's train step (src/trainer.py:157-165).
//
// dfd_decoder_attn_bwd — gradient of the single-query two-branch attention.  Streams K and V of one
//   (clip, frame) per workgroup exactly once (same thread layout as the forward: heads*8 threads
//   per key row, 8 channels each), recomputes the weights from the forward's (max, sumexp) and
//   produces dq and, since the exported K/V are `encoder_kv + positional_embedding`, the
//   positional-embedding gradient  Σ_{clip, patch} (dK + dV)  per frame — without ever writing
//   dK / dV (they are only materialised when the caller asks for them: adapter training).
// dfd_linear_rows_bwd_weight, dfd_transpose_f32, dfd_layernorm_bwd, dfd_quickgelu, dfd_head_bwd —
//   the [B, D]-row pieces of the backward.
#include "dropout.hpp"
#include "decoder_common.hpp"

namespace {

constexpr int HD = 64;

__device__ __forceinline__ float sgn(float x) { return (float)(x > 0.f) - (float)(x < 0.f); }

// grid (T, B).  part layout per (b, t): [heads][128] dq (softmax query | CoDA query), then [heads*64] dpos.
template <typename T, typename G, int MAXT>
__global__ __launch_bounds__(MAXT) void decoder_attn_bwd_kernel(const float* __restrict__ q, const T* __restrict__ k,
                                                                const T* __restrict__ v,
                                                                const uint8_t* __restrict__ frame_mask,
                                                                const float* __restrict__ dmix,
                                                                const float* __restrict__ mix_s,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ ext_w,
                                                                const float* __restrict__ ext_ds, float* __restrict__ part,
                                                                G* __restrict__ dk_out, G* __restrict__ dv_out,
                                                                int T_frames, int patches, int heads, int R, KvLayout lay) {
  extern __shared__ float red[];  // [R][tpr][24]
  const int tpr = heads * 8;
  const int t = blockIdx.x, b = blockIdx.y;
  const int rs = threadIdx.x / tpr, tr = threadIdx.x % tpr;
  const int hd = tr >> 3, sub = tr & 7;
  const int D = heads * HD;
  const int S = T_frames * patches;
  float* dst = part + ((int64_t)b * T_frames + t) * (3 * D);
  const bool valid = frame_mask[(int64_t)b * T_frames + t] != 0;

  float qs[8], qc[8], dm[8], dqs[8], dqc[8], sv[8];
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;
  {
    const float* qp = q + ((int64_t)b * heads + hd) * (2 * HD) + sub * 8;
    const float* dp = dmix + (int64_t)b * D + hd * HD + sub * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qs[e] = qp[e];
      qc[e] = qp[HD + e];
      dm[e] = dp[e];
      dqs[e] = dqc[e] = sv[e] = 0.f;
    }
  }
  // Σ_i a_i·(dmix·v_i)/2 = (dmix · mix_softmax)/2
  // attn_mode: softmax-branch weights and score gradients come from the grouped-softmax passes (ext_*)
  const bool ext = ext_w != nullptr;
  float dlt = 0.f, M = 0.f, invL = 1.f;
  if (!ext) {
    const float* mp = mix_s + (int64_t)b * D + hd * HD + sub * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) dlt = fmaf(dm[e], mp[e], dlt);
    dlt = 0.5f * group8_sum(dlt);
    M = stats[((int64_t)b * heads + hd) * 2];
    invL = 1.0f / stats[((int64_t)b * heads + hd) * 2 + 1];
  }

  if (valid) {
    const int s0 = t * patches;
    // this block's frame: rows j*row_stride of frame b*T + t (KvLayout), plus the frame's positional embedding
    const T* kb = k + ((int64_t)b * T_frames + t) * lay.frame_stride + hd * HD + sub * 8;
    const T* vb = v + ((int64_t)b * T_frames + t) * lay.frame_stride + hd * HD + sub * 8;
    float pe[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pe[e] = 0.f;
    if (lay.pos != nullptr) Ld8<float>::load(lay.pos + (int64_t)t * D + hd * HD + sub * 8, pe);
    // Four rows per trip, all eight loads issued before the first use (the trip is a latency chain otherwise).
    // Every lane forms its 8-channel share of the four dot products of the four rows; a reduce-scatter over the
    // 8-lane head group leaves lanes j and j+4 with the totals of row j, so a row's transcendentals and divisions
    // (the VALU bulk here) run on two lanes instead of eight; its three gradient scalars travel back to the group
    // for the per-channel accumulation.
    constexpr int UN = 4;
    const int lane_base = (threadIdx.x & 63) & ~7;
    float a1o = 0.f, a2o = 0.f, a3o = 0.f;  // sums over the rows lanes 0..3 of the group finished
    for (int j0 = rs; j0 < patches; j0 += UN * R) {
      Raw8<T> kq[UN], vq[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int j = min(j0 + u * R, patches - 1);
        kq[u].load(kb + (int64_t)j * lay.row_stride);
        vq[u].load(vb + (int64_t)j * lay.row_stride);
      }
      float ps[UN], pt[UN], pl[UN], pw[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        float s = 0.f, tt = 0.f, l1 = 0.f, dw = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float kk = kq[u].get(e) + pe[e], vv = vq[u].get(e) + pe[e];
          s = fmaf(qs[e] * 0.125f, kk, s);
          tt = fmaf(qc[e] * 0.125f, kk, tt);
          l1 += fabsf(qc[e] - kk);
          dw = fmaf(dm[e], vv, dw);
        }
        ps[u] = s; pt[u] = tt; pl[u] = l1; pw[u] = dw;
      }
      const float s = group8_reduce_scatter4(ps, sub), tt = group8_reduce_scatter4(pt, sub);
      const float l1 = group8_reduce_scatter4(pl, sub), dw = group8_reduce_scatter4(pw, sub);
      // my row of this trip (shared with lane sub ^ 4)
      const int jm = j0 + (sub & 3) * R;
      const bool ok = jm < patches;
      const int64_t wi = ((int64_t)b * heads + hd) * S + s0 + min(jm, patches - 1);
      const float aw = ext ? ext_w[wi] : __expf(s - M) * invL;      // softmax(-branch) weight
      const float g = 2.0f / (1.0f + __expf(l1 * 0.125f));         // 2·sigmoid(−l1/8)
      const float e2 = __expf(2.0f * tt);
      const float th = 1.0f - 2.0f / (e2 + 1.0f);                  // tanh
      const float w = ok ? 0.5f * (aw + th * g) : 0.f;
      const float dc = 0.5f * dw;
      const float ds = ok ? (ext ? ext_ds[wi] : aw * (0.5f * dw - dlt)) : 0.f;
      const float dt = ok ? dc * g * (1.0f - th * th) : 0.f;
      const float dL = ok ? -(dc * th) * g * (1.0f - 0.5f * g) * 0.125f : 0.f;
      if (sub < 4) {  // each row once
        a1o += ds;
        a2o += dt;
        a3o += w;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const float dsu = __shfl(ds, lane_base + u, 64), dtu = __shfl(dt, lane_base + u, 64), dLu = __shfl(dL, lane_base + u, 64);
        float dkk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float kk = kq[u].get(e) + pe[e];
          const float sg = sgn(qc[e] - kk) * dLu;
          dqs[e] = fmaf(dsu * 0.125f, kk, dqs[e]);
          dqc[e] = fmaf(dtu * 0.125f, kk, dqc[e]) + sg;
          sv[e] += sg;
          dkk[e] = (dsu * qs[e] + dtu * qc[e]) * 0.125f - sg;
        }
        if (dk_out != nullptr) {
          const float wu = __shfl(w, lane_base + u, 64);
          const int j = j0 + u * R;
          if (j < patches) {
            G* ko = dk_out + ((int64_t)b * S + s0 + j) * D + hd * HD + sub * 8;
            G* vo = dv_out + ((int64_t)b * S + s0 + j) * D + hd * HD + sub * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { ko[e] = from_f32<G>(dkk[e]); vo[e] = from_f32<G>(wu * dm[e]); }
          }
        }
      }
    }
    a1 = group8_sum(a1o);
    a2 = group8_sum(a2o);
    a3 = group8_sum(a3o);
  } else if (dk_out != nullptr) {
    for (int j = rs; j < patches; j += R) {
      G* ko = dk_out + ((int64_t)b * S + t * patches + j) * D + hd * HD + sub * 8;
      G* vo = dv_out + ((int64_t)b * S + t * patches + j) * D + hd * HD + sub * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { ko[e] = from_f32<G>(0.f); vo[e] = from_f32<G>(0.f); }
    }
  }
  // Σ_j (dK_j + dV_j) for this thread's 8 channels
  float dp[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dp[e] = (a1 * qs[e] + a2 * qc[e]) * 0.125f - sv[e] + a3 * dm[e];

  float* mine = red + ((size_t)rs * tpr + tr) * 24;
#pragma unroll
  for (int e = 0; e < 8; ++e) { mine[e] = dqs[e]; mine[8 + e] = dqc[e]; mine[16 + e] = dp[e]; }
  __syncthreads();
  if (rs == 0) {
    float acc[24];
#pragma unroll
    for (int e = 0; e < 24; ++e) acc[e] = 0.f;
    for (int r2 = 0; r2 < R; ++r2) {
      const float* o = red + ((size_t)r2 * tpr + tr) * 24;
#pragma unroll
      for (int e = 0; e < 24; ++e) acc[e] += o[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dst[hd * 128 + sub * 8 + e] = acc[e];
      dst[hd * 128 + HD + sub * 8 + e] = acc[8 + e];
      dst[2 * D + hd * HD + sub * 8 + e] = acc[16 + e];
    }
  }
}

// Backward of decoder_modes_fwd_kernel: with p the grouped-softmax weights of one mode and
// dp = dmix·v/2 (the branch average), ds = Σ_modes p·(dp − Σ_group p·dp).
__global__ __launch_bounds__(256) void decoder_modes_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ dwv,
                                                                float* __restrict__ dscores, int modes, int T_frames,
                                                                int patches) {
  extern __shared__ float sm[];  // [S] scores | [S] dp | [S] ds
  const int S = T_frames * patches;
  float* sc = sm;
  float* dp = sm + S;
  float* ds = sm + 2 * S;
  const int64_t base = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * S;
  for (int i = threadIdx.x; i < S; i += 256) { sc[i] = scores[base + i]; dp[i] = 0.5f * dwv[base + i]; ds[i] = 0.f; }
  __syncthreads();
  if (modes & 1) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = wave; t < T_frames; t += 4) {
      const float* row = sc + t * patches;
      const float* dr = dp + t * patches;
      float m = -INFINITY;
      for (int p = lane; p < patches; p += 64) m = fmaxf(m, row[p]);
      m = wave_max(m);
      float l = 0.f, dd = 0.f;
      for (int p = lane; p < patches; p += 64) {
        const float e = __expf(row[p] - m);
        l += e;
        dd = fmaf(e, dr[p], dd);
      }
      l = wave_sum(l);
      dd = wave_sum(dd) / l;
      for (int p = lane; p < patches; p += 64) ds[t * patches + p] += __expf(row[p] - m) / l * (dr[p] - dd);
    }
    __syncthreads();
  }
  if (modes & 2) {
    for (int p = threadIdx.x; p < patches; p += 256) {
      float m = -INFINITY;
      for (int t = 0; t < T_frames; ++t) m = fmaxf(m, sc[t * patches + p]);
      float l = 0.f, dd = 0.f;
      for (int t = 0; t < T_frames; ++t) {
        const float e = __expf(sc[t * patches + p] - m);
        l += e;
        dd = fmaf(e, dp[t * patches + p], dd);
      }
      dd /= l;
      for (int t = 0; t < T_frames; ++t) {
        const int i = t * patches + p;
        // a padded frame has weight exactly 0; keep its gradient 0 rather than 0·(dp − D)
        const float w = __expf(sc[i] - m) / l;
        ds[i] += (w == 0.f) ? 0.f : w * (dp[i] - dd);
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < S; i += 256) dscores[base + i] = ds[i];
}

// dq[b, :] = Σ_t part[b, t, 0:2D];  dpos[t, :] = Σ_b part[b, t, 2D:3D]
__global__ void decoder_attn_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dq,
                                               float* __restrict__ dpos, int B, int T_frames, int D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = B * 2 * D;
  if (idx < nq) {
    const int b = idx / (2 * D), c = idx % (2 * D);
    float s = 0.f;
    for (int t = 0; t < T_frames; ++t) s += part[((int64_t)b * T_frames + t) * (3 * D) + c];
    dq[idx] = s;
  } else if (dpos != nullptr && idx < nq + T_frames * D) {
    const int i2 = idx - nq;
    const int t = i2 / D, c = i2 % D;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[((int64_t)b * T_frames + t) * (3 * D) + 2 * D + c];
    dpos[i2] = s;
  }
}

// dW[n, k] = Σ_b dy[b, n] x[b, k];  db[n] = Σ_b dy[b, n].  8 output rows per workgroup.
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                const float* __restrict__ x, int64_t ldx,
                                                                float* __restrict__ dW, float* __restrict__ db, int B,
                                                                int N, int K) {
  __shared__ float sdy[8][64];
  const int n0 = blockIdx.x * 8;
  for (int i = threadIdx.x; i < 8 * B; i += 256) {
    const int r = i / B, b = i % B;
    sdy[r][b] = (n0 + r < N) ? dy[(int64_t)b * lddy + n0 + r] : 0.f;
  }
  __syncthreads();
  for (int kk = threadIdx.x * 4; kk < K; kk += 1024) {
    f32x4 acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (int64_t)b * ldx + kk);
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += x4 * sdy[r][b];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (n0 + r < N) *reinterpret_cast<f32x4*>(dW + (int64_t)(n0 + r) * K + kk) = acc[r];
  }
  if (db != nullptr && threadIdx.x < 8 && n0 + threadIdx.x < N) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sdy[threadIdx.x][b];
    db[n0 + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(int64_t)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) dst[(int64_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// One workgroup per row: dx = [dx +] rstd·(g·dy − mean(g·dy) − x̂·mean(g·dy·x̂)); writes x̂ for the column pass.
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ dy, int64_t lddy,
                                                                 float* __restrict__ dx, int64_t lddx,
                                                                 float* __restrict__ xhat, int cols, float eps, int accumulate) {
  __shared__ float sc[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (int64_t)b * ldx;
  const float* dyr = dy + (int64_t)b * lddy;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sc[wave] = v;
    __syncthreads();
    return sc[0] + sc[1] + sc[2] + sc[3];
  };
  float s = 0.f;
  for (int c = tid; c < cols; c += 256) s += xr[c];
  const float mean = block_sum(s) / (float)cols;
  float qv = 0.f;
  for (int c = tid; c < cols; c += 256) { const float d = xr[c] - mean; qv += d * d; }
  const float rstd = rsqrtf(block_sum(qv) / (float)cols + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int c = tid; c < cols; c += 256) {
    const float xh = (xr[c] - mean) * rstd;
    const float g = dyr[c] * gamma[c];
    s1 += g;
    s2 += g * xh;
  }
  const float m1 = block_sum(s1) / (float)cols;
  const float m2 = block_sum(s2) / (float)cols;
  for (int c = tid; c < cols; c += 256) {
    const float xh = (xr[c] - mean) * rstd;
    const float g = dyr[c] * gamma[c];
    const float r = rstd * (g - m1 - xh * m2);
    float* o = dx + (int64_t)b * lddx + c;
    *o = accumulate ? *o + r : r;
    xhat[(int64_t)b * cols + c] = xh;
  }
}

__global__ void layernorm_bwd_cols_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ xhat,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int cols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float g = 0.f, bb = 0.f;
  for (int b = 0; b < rows; ++b) {
    const float d = dy[(int64_t)b * lddy + c];
    g = fmaf(d, xhat[(int64_t)b * cols + c], g);
    bb += d;
  }
  dgamma[c] = g;
  dbeta[c] = bb;
}

// du == nullptr: out = g(u);  else out = du · g'(u), g(u) = u·σ(1.702u)
__global__ void quickgelu_kernel(const float* __restrict__ u, const float* __restrict__ du, float* __restrict__ out, int64_t n,
                                 DfdDrop d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = u[i];
  const float sg = 1.0f / (1.0f + __expf(-1.702f * x));
  const float r = du ? du[i] * (sg + 1.702f * x * sg * (1.0f - sg)) : x * sg;
  out[i] = dfd_drop_one(d, (uint64_t)i, r);  // y = m·s·g(u); dL/du = m·s·dL/dy·g'(u)
}

// Head backward: logits = 5 z/(‖z‖+ε), z = f·P, f = LN(x).  One workgroup per clip computes dz and
// df = dz·Pᵀ (+ dfeat_ext); dP and the LayerNorm gradients are finished by the generic kernels.
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ dlogits,
                                                       const float* __restrict__ proj, const float* __restrict__ dfeat_ext,
                                                       float* __restrict__ dz, float* __restrict__ df, int D, int out_dim) {
  extern __shared__ float sh[];  // [out_dim] dz
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* z = raw + (int64_t)b * out_dim;
  const float* dyv = dlogits + (int64_t)b * out_dim;
  float n2 = 0.f, dot = 0.f;
  for (int o = 0; o < out_dim; ++o) { n2 = fmaf(z[o], z[o], n2); dot = fmaf(dyv[o], z[o], dot); }
  const float n = sqrtf(n2), ne = n + 1e-10f;
  for (int o = tid; o < out_dim; o += 256) {
    const float d = 5.0f * dyv[o] / ne - (n > 0.f ? 5.0f * dot * z[o] / (ne * ne * n) : 0.f);
    sh[o] = d;
    dz[(int64_t)b * out_dim + o] = d;
  }
  __syncthreads();
  for (int c = tid; c < D; c += 256) {
    float s = dfeat_ext ? dfeat_ext[(int64_t)b * D + c] : 0.f;
    for (int o = 0; o < out_dim; ++o) s = fmaf(sh[o], proj[(int64_t)c * out_dim + o], s);
    df[(int64_t)b * D + c] = s;
  }
}

// dP[c, o] = Σ_b f[b, c] dz[b, o]
__global__ void head_dproj_kernel(const float* __restrict__ feat, const float* __restrict__ dz, float* __restrict__ dproj, int B,
                                  int D, int out_dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * out_dim) return;
  const int c = idx / out_dim, o = idx % out_dim;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s = fmaf(feat[(int64_t)b * D + c], dz[(int64_t)b * out_dim + o], s);
  dproj[idx] = s;
}

int rows_per_block(int heads) {
  const int tpr = heads * 8;
  int R = (256 + tpr - 1) / tpr;
  while ((tpr * R) % 64 != 0) ++R;
  return R;
}

}  // namespace

extern "C" size_t dfd_decoder_attn_bwd_workspace(int B, int T, int heads, int d) {
  if (B <= 0 || T <= 0 || heads <= 0 || d != HD) return 0;
  return (size_t)B * T * 3 * heads * HD * sizeof(float);
}

extern "C" int dfd_decoder_attn_bwd(const float* q, const void* k, const void* v, int kv_dtype, const dfd_kv_layout_t* layout,
                                    const uint8_t* frame_mask,
                                    const float* dmix, const float* mix_softmax, const float* stats,
                                    const float* ext_weights, const float* ext_dscores, float* dq, float* dpos, void* dk,
                                    void* dv, int dkv_dtype, void* workspace, int B, int T, int patches, int heads, int d,
                                    void* stream) {
  DFD_REQUIRE(q && k && v && frame_mask && dmix && dq && workspace, "dfd_decoder_attn_bwd: null pointer");
  DFD_REQUIRE(!ext_weights == !ext_dscores, "dfd_decoder_attn_bwd: ext_weights and ext_dscores go together");
  DFD_REQUIRE(ext_weights || (mix_softmax && stats), "dfd_decoder_attn_bwd: mix_softmax and stats are required without ext_weights");
  DFD_REQUIRE(d == HD, "dfd_decoder_attn_bwd: head dim %d, only 64 is supported", d);
  DFD_REQUIRE(B >= 0 && T > 0 && patches > 0 && heads > 0 && heads * HD <= 1024, "dfd_decoder_attn_bwd: bad shape");
  DFD_REQUIRE(kv_dtype == DFD_F32 || kv_dtype == DFD_BF16, "dfd_decoder_attn_bwd: kv_dtype=%d", kv_dtype);
  DFD_REQUIRE(!dk == !dv, "dfd_decoder_attn_bwd: dk and dv must both be given or both be NULL");
  DFD_REQUIRE(!dk || dkv_dtype == DFD_F32 || dkv_dtype == DFD_BF16, "dfd_decoder_attn_bwd: dkv_dtype=%d", dkv_dtype);
  DFD_REQUIRE(dfd_aligned16(k) && dfd_aligned16(v), "dfd_decoder_attn_bwd: k and v must be 16-byte aligned");
  if (layout != nullptr) {
    const int per16 = kv_dtype == DFD_F32 ? 4 : 8;
    DFD_REQUIRE(layout->row_stride >= heads * HD && layout->frame_stride > 0 && layout->row_stride % per16 == 0 &&
                    layout->frame_stride % per16 == 0 && (!layout->pos || dfd_aligned16(layout->pos)),
                "dfd_decoder_attn_bwd: key/value layout: row stride %lld, frame stride %lld (elements; 16-byte aligned rows)",
                (long long)layout->row_stride, (long long)layout->frame_stride);
  }
  if (B == 0) return DFD_OK;
  const KvLayout lay = dfd_kv_layout(layout, patches, heads * HD);
  const int R = rows_per_block(heads);
  const int threads = heads * 8 * R;
  const size_t lds = (size_t)threads * 24 * sizeof(float);
  const int D = heads * HD;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const dim3 grid(T, B), block(threads);
#define BWD_LAUNCH1(KT, GT, MT)                                                                                              \
  do {                                                                                                                      \
    if (lds > 64 * 1024)                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decoder_attn_bwd_kernel<KT, GT, MT>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                      \
    hipLaunchKernelGGL((decoder_attn_bwd_kernel<KT, GT, MT>), grid, block, lds, st, q, static_cast<const KT*>(k),              \
                       static_cast<const KT*>(v), frame_mask, dmix, mix_softmax, stats, ext_weights, ext_dscores, part,      \
                       static_cast<GT*>(dk), static_cast<GT*>(dv), T, patches, heads, R, lay);                                    \
  } while (0)
  // blocks of <= 512 threads (every even head count) get the 256-register budget: no spills with two rows in flight
#define BWD_LAUNCH(KT, GT)                                                                                                  \
  do {                                                                                                                      \
    if (threads <= 512) BWD_LAUNCH1(KT, GT, 512); else BWD_LAUNCH1(KT, GT, 1024);                                           \
  } while (0)
  const bool gb = dk && dkv_dtype == DFD_BF16;
  if (kv_dtype == DFD_F32) { if (gb) BWD_LAUNCH(float, bf16_t); else BWD_LAUNCH(float, float); }
  else { if (gb) BWD_LAUNCH(bf16_t, bf16_t); else BWD_LAUNCH(bf16_t, float); }
#undef BWD_LAUNCH1
#undef BWD_LAUNCH
  DFD_CHECK_LAUNCH("dfd_decoder_attn_bwd");
  const int total = B * 2 * D + (dpos ? T * D : 0);
  hipLaunchKernelGGL(decoder_attn_bwd_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, dq, dpos, B, T, D);
  DFD_CHECK_LAUNCH("dfd_decoder_attn_bwd(reduce)");
  return DFD_OK;
}

int dfd_decoder_rowdot(const float* a, int a_stride, const void* X, int kv_dtype, const dfd_kv_layout_t* layout,
                       const uint8_t* frame_mask, float* out, float scale, float fill, int B, int T, int patches, int heads,
                       hipStream_t st);  // decoder.hip

extern "C" int dfd_decoder_attn_modes_bwd(const float* scores, const void* v, int kv_dtype, const dfd_kv_layout_t* layout,
                                          const float* dmix, int modes, float* dwv_workspace, float* dscores, int B, int T,
                                          int patches, int heads, int d, void* stream) {
  DFD_REQUIRE(scores && v && dmix && dwv_workspace && dscores, "dfd_decoder_attn_modes_bwd: null pointer");
  DFD_REQUIRE(d == HD, "dfd_decoder_attn_modes_bwd: head dim %d, only 64 is supported", d);
  DFD_REQUIRE(B >= 0 && T > 0 && patches > 0 && heads > 0 && heads * HD <= 1024, "dfd_decoder_attn_modes_bwd: bad shape");
  DFD_REQUIRE(modes >= 1 && modes <= 3, "dfd_decoder_attn_modes_bwd: modes=%d (bit 0 frame, bit 1 temporal)", modes);
  DFD_REQUIRE(kv_dtype == DFD_F32 || kv_dtype == DFD_BF16, "dfd_decoder_attn_modes_bwd: kv_dtype=%d", kv_dtype);
  DFD_REQUIRE(dfd_aligned16(v) && dfd_aligned16(dmix), "dfd_decoder_attn_modes_bwd: pointers must be 16-byte aligned");
  const size_t lds = (size_t)3 * T * patches * sizeof(float);
  DFD_REQUIRE(lds <= 150 * 1024, "dfd_decoder_attn_modes_bwd: T*patches=%d too large for one LDS pass", T * patches);
  if (B == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // dL/d(weight of key s) = dmix · v_s  (weights multiply v in the mix, models.py:144)
  const int rc = dfd_decoder_rowdot(dmix, HD, v, kv_dtype, layout, nullptr, dwv_workspace, 1.0f, 0.f, B, T, patches, heads, st);
  if (rc != DFD_OK) return rc;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decoder_modes_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(decoder_modes_bwd_kernel, dim3(heads, B), dim3(256), lds, st, scores, dwv_workspace, dscores, modes, T, patches);
  DFD_CHECK_LAUNCH("dfd_decoder_attn_modes_bwd");
  return DFD_OK;
}

extern "C" int dfd_linear_rows_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dW, float* db,
                                          int B, int N, int K, void* stream) {
  DFD_REQUIRE(dy && x && dW, "dfd_linear_rows_bwd_weight: null pointer");
  DFD_REQUIRE(B > 0 && B <= 64 && N > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0, "dfd_linear_rows_bwd_weight: bad shape B=%d N=%d K=%d", B, N, K);
  DFD_REQUIRE(dfd_aligned16(x) && dfd_aligned16(dW), "dfd_linear_rows_bwd_weight: x and dW must be 16-byte aligned");
  hipLaunchKernelGGL(linear_bwd_weight_kernel, dim3((N + 7) / 8), dim3(256), 0, static_cast<hipStream_t>(stream), dy, lddy, x,
                     ldx, dW, db, B, N, K);
  DFD_CHECK_LAUNCH("dfd_linear_rows_bwd_weight");
  return DFD_OK;
}

extern "C" int dfd_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream) {
  DFD_REQUIRE(src && dst && rows > 0 && cols > 0, "dfd_transpose_f32: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, dst, rows, cols);
  DFD_CHECK_LAUNCH("dfd_transpose_f32");
  return DFD_OK;
}

extern "C" int dfd_layernorm_bwd(const float* x, int64_t ldx, const float* gamma, const float* dy, int64_t lddy, float* dx,
                                 int64_t lddx, float* dgamma, float* dbeta, float* xhat_ws, int rows, int cols, float eps,
                                 int accumulate_dx, void* stream) {
  DFD_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && xhat_ws, "dfd_layernorm_bwd: null pointer");
  DFD_REQUIRE(rows > 0 && cols > 0, "dfd_layernorm_bwd: bad shape");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(layernorm_bwd_rows_kernel, dim3(rows), dim3(256), 0, st, x, ldx, gamma, dy, lddy, dx, lddx, xhat_ws, cols,
                     eps, accumulate_dx);
  DFD_CHECK_LAUNCH("dfd_layernorm_bwd(rows)");
  hipLaunchKernelGGL(layernorm_bwd_cols_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, dy, lddy, xhat_ws, dgamma, dbeta,
                     rows, cols);
  DFD_CHECK_LAUNCH("dfd_layernorm_bwd(cols)");
  return DFD_OK;
}

extern "C" int dfd_quickgelu(const float* u, const float* du, float* out, int64_t n, const dfd_dropout_t* drop, void* stream) {
  DFD_REQUIRE(u && out && n >= 0, "dfd_quickgelu: bad arguments");
  DFD_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f && (drop->p == 0.f || drop->rng_state)), "dfd_quickgelu: bad dropout descriptor");
  if (n == 0) return DFD_OK;
  hipLaunchKernelGGL(quickgelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), u, du,
                     out, n, dfd_make_drop(drop));
  DFD_CHECK_LAUNCH("dfd_quickgelu");
  return DFD_OK;
}

extern "C" int dfd_head_bwd(const float* raw_logits, const float* dlogits, const float* proj, const float* feat,
                            const float* dfeat_ext, float* dz, float* dfeat, float* dproj, int B, int D, int out_dim, void* stream) {
  DFD_REQUIRE(raw_logits && dlogits && proj && feat && dz && dfeat && dproj, "dfd_head_bwd: null pointer");
  DFD_REQUIRE(B > 0 && D > 0 && out_dim > 0 && out_dim <= 4096, "dfd_head_bwd: bad shape");
  hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(256), out_dim * sizeof(float), static_cast<hipStream_t>(stream), raw_logits,
                     dlogits, proj, dfeat_ext, dz, dfeat, D, out_dim);
  DFD_CHECK_LAUNCH("dfd_head_bwd");
  hipLaunchKernelGGL(head_dproj_kernel, dim3((D * out_dim + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), feat, dz,
                     dproj, B, D, out_dim);
  DFD_CHECK_LAUNCH("dfd_head_bwd(dproj)");
  return DFD_OK;
}
