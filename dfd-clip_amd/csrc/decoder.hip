// Temporal decoder forward kernels (reference src/models.py:99-146, :160-176, :323-361, :551-553).
//
// The decoder has ONE query token per clip, so everything except the K/V pass is a handful of
// [B, D] row operations streaming ~6.5 M fp32 weights per block; the K/V pass is a pure HBM
// stream: per clip and layer 2·S·D·sizeof(kv) bytes, read exactly once.
//
// dfd_decoder_attn_fwd — single-query two-branch attention.  K/V rows are [heads*64] wide.
//   A row is covered by heads*8 threads, 8 channels (16 B of bf16 / 32 B of f32) each, so a
//   head's three reductions (q_s·k, q_c·k, ‖q_c−k‖₁) are 3-step xor-shuffles inside 8-lane
//   groups.  R rows are processed side by side per workgroup; each thread keeps an online
//   softmax state (max, sum, acc_s[8]) and the CoDA accumulator acc_c[8] for its channels.
//   Workgroups split S; partial states are merged by decoder_attn_combine_kernel, which also
//   emits (max, sumexp) per (clip, head) for the backward pass.
// dfd_linear_rows — y[B,N] = x[B,K]·W[N,K]ᵀ + b: one wave per output column streams the weight
//   row once per 8 clips; x stays L1/L2 resident.
// dfd_head_fwd — ln_post + projection + 5·z/(‖z‖+1e-10).
#include "dropout.hpp"
#include "decoder_common.hpp"

namespace {

constexpr int HD = 64;
constexpr int PART = 2 + 2 * HD;  // floats per (clip, split, head): m, l, acc_s[64], acc_c[64]

template <typename T, int MAXT, bool POS>
__global__ __launch_bounds__(MAXT, (MAXT <= 512 ? (POS ? 2 : 3) : 4)) void decoder_attn_partial_kernel(const float* __restrict__ q, const T* __restrict__ k,
                                                                    const T* __restrict__ v,
                                                                    const uint8_t* __restrict__ frame_mask,
                                                                    const float* __restrict__ ext_w,
                                                                    float* __restrict__ ws, int splits, int T_frames,
                                                                    int patches, int heads, int R, KvLayout lay,
                                                                    FastDiv div_patches) {
  extern __shared__ float red[];  // [R][tpr][18], then (POS) the positional embedding of this block's frames
  const int tpr = heads * 8;
  const int b = blockIdx.y, split = blockIdx.x;
  const int rs = threadIdx.x / tpr, tr = threadIdx.x % tpr;
  const int hd = tr >> 3, sub = tr & 7;
  const int S = T_frames * patches;
  const int D = heads * HD;
  const int per = (S + splits - 1) / splits;
  const int s_begin = split * per;
  const int s_end = min(S, s_begin + per);
  // POS: the rows of this block lie in frames f0 .. f1 of the clip; their positional embeddings are staged in LDS
  // once (held in registers they cost the kernel half its occupancy)
  [[maybe_unused]] float* const posl = red + (size_t)blockDim.x * 18;
  [[maybe_unused]] const int f0 = (int)div_patches.div((uint32_t)min(s_begin, S - 1));
  if constexpr (POS) {
    const int f1 = (int)div_patches.div((uint32_t)(max(s_end, s_begin + 1) - 1 < S ? max(s_end, s_begin + 1) - 1 : S - 1));
    const int n4 = (f1 - f0 + 1) * (D >> 2);
    for (int i = threadIdx.x; i < n4; i += blockDim.x)
      reinterpret_cast<f32x4*>(posl)[i] = reinterpret_cast<const f32x4*>(lay.pos + (int64_t)f0 * D)[i];
    __syncthreads();
  }

  float qs[8], qc[8];
  {
    const float* qp = q + ((int64_t)b * heads + hd) * (2 * HD) + sub * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qs[e] = qp[e] * 0.125f;
      qc[e] = qp[HD + e];
    }
  }
  float mx = -INFINITY, l = 0.f, as[8], ac[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { as[e] = 0.f; ac[e] = 0.f; }

  // row s of clip b = patch (s % patches) of frame b*T + s / patches (KvLayout: dense export or the qkv activation in place)
  const T* kb = k + (int64_t)b * T_frames * lay.frame_stride + hd * HD + sub * 8;
  const T* vb = v + (int64_t)b * T_frames * lay.frame_stride + hd * HD + sub * 8;
  [[maybe_unused]] const float* const pb = posl + hd * HD + sub * 8;
  const uint8_t* mb = frame_mask + (int64_t)b * T_frames;
  const int lane_base = (threadIdx.x & 63) & ~7;
  // Eight rows per trip, one per lane of the 8-lane head group, all sixteen loads issued before the first use (the
  // loop is a latency chain otherwise).  Every lane forms its 8-channel share of the three dot products of all eight
  // rows; a reduce-scatter over the group leaves lane j with the totals of row j, so the row's transcendentals
  // (exp, tanh, sigmoid: the VALU bulk of this kernel) run ONCE per row instead of once per lane; the row weights
  // then travel back to the group for the V accumulation.  The softmax is rescaled once per trip.
  constexpr int UN = 8;
  float l_own = 0.f;  // softmax weights of the rows this lane finished; summed over the group at the end
  for (int s0 = s_begin + rs; s0 < s_end; s0 += UN * R) {
    Raw8<T> kr[UN], vr[UN];
    [[maybe_unused]] int po[UN];  // LDS offset of the row's frame in the staged positional embeddings
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int s = min(min(s0 + u * R, s_end - 1), S - 1);  // clamp: rows past the end are loaded but not used
      const int tf = (int)div_patches.div((uint32_t)s);
      const int64_t off = (int64_t)tf * lay.frame_stride + (int64_t)(s - tf * patches) * lay.row_stride;
      kr[u].load(kb + off);
      vr[u].load(vb + off);
      po[u] = (tf - f0) * D;
    }
    float pds[UN], pdc[UN], pl1[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      float ds = 0.f, dc = 0.f, l1 = 0.f;
      [[maybe_unused]] float pe[8];
      if constexpr (POS) Ld8<float>::load(pb + po[u], pe);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float kk = POS ? kr[u].get(e) + pe[e] : kr[u].get(e);
        ds = fmaf(qs[e], kk, ds);
        dc = fmaf(qc[e] * 0.125f, kk, dc);
        l1 += fabsf(qc[e] - kk);
      }
      pds[u] = ds;
      pdc[u] = dc;
      pl1[u] = l1;
    }
    const float ds = group8_reduce_scatter(pds, sub), dc = group8_reduce_scatter(pdc, sub), l1 = group8_reduce_scatter(pl1, sub);
    // my row of this trip
    const int s_me = s0 + sub * R;
    const bool ok = s_me < s_end && mb[div_patches.div((uint32_t)min(s_me, S - 1))] != 0;
    float p;
    if (ext_w != nullptr) {
      // attn_mode: the softmax-branch weight was computed by the grouped-softmax pass
      p = ok ? ext_w[((int64_t)b * heads + hd) * S + s_me] : 0.f;
    } else {
      const float m_new = fmaxf(mx, group8_max(ok ? ds : -INFINITY));
      if (m_new != mx) {  // uniform over the group; exp(-inf) = 0 the first time
        const float alpha = __expf(mx - m_new);
        l_own *= alpha;
#pragma unroll
        for (int e = 0; e < 8; ++e) as[e] *= alpha;
        mx = m_new;
      }
      p = ok ? __expf(ds - mx) : 0.f;
      l_own += p;
    }
    const float gate = 2.0f / (1.0f + __expf(l1 * 0.125f));  // 2·sigmoid(−l1/√d)
    const float c = ok ? fast_tanh(dc) * gate : 0.f;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const float pu = __shfl(p, lane_base + u, 64), cu = __shfl(c, lane_base + u, 64);
      [[maybe_unused]] float pe[8];
      if constexpr (POS) Ld8<float>::load(pb + po[u], pe);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float vv = POS ? vr[u].get(e) + pe[e] : vr[u].get(e);
        as[e] = fmaf(pu, vv, as[e]);
        ac[e] = fmaf(cu, vv, ac[e]);
      }
    }
  }
  l = group8_sum(l_own);
  if (ext_w != nullptr) {  // weights are final: neutral softmax state, unit normaliser counted once
    mx = 0.f;
    l = (split == 0 && rs == 0) ? 1.f : 0.f;
  }
  // merge the R row slots through LDS
  float* mine = red + ((size_t)rs * tpr + tr) * 18;
  mine[0] = mx;
  mine[1] = l;
#pragma unroll
  for (int e = 0; e < 8; ++e) { mine[2 + e] = as[e]; mine[10 + e] = ac[e]; }
  __syncthreads();
  if (rs == 0) {
    float M = mx;
    for (int r2 = 1; r2 < R; ++r2) M = fmaxf(M, red[((size_t)r2 * tpr + tr) * 18]);
    float L = 0.f, As[8], Ac[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { As[e] = 0.f; Ac[e] = 0.f; }
    for (int r2 = 0; r2 < R; ++r2) {
      const float* o = red + ((size_t)r2 * tpr + tr) * 18;
      const float w = (o[0] == -INFINITY) ? 0.f : __expf(o[0] - M);
      L = fmaf(o[1], w, L);
#pragma unroll
      for (int e = 0; e < 8; ++e) { As[e] = fmaf(o[2 + e], w, As[e]); Ac[e] += o[10 + e]; }
    }
    float* dst = ws + (((int64_t)b * splits + split) * heads + hd) * PART;
    if (sub == 0) { dst[0] = M; dst[1] = L; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { dst[2 + sub * 8 + e] = As[e]; dst[2 + HD + sub * 8 + e] = Ac[e]; }
  }
}

__global__ void decoder_attn_combine_kernel(const float* __restrict__ ws, float* __restrict__ mix,
                                            float* __restrict__ mix_softmax, float* __restrict__ stats, int splits,
                                            int heads) {
  extern __shared__ float cw[];  // [heads][splits] rescale weights exp(m_split − M)
  __shared__ float sM[16], sL[16];
  const int b = blockIdx.x;
  const int hd = threadIdx.x / HD, c = threadIdx.x % HD;
  const float* base = ws + (int64_t)b * splits * heads * PART;
  // stage 1: one wave-parallel pass over the (split, head) states instead of a serial chain per thread
  if (c < 64) {
    float M = -INFINITY;
    for (int s = c; s < splits; s += 64) M = fmaxf(M, base[((int64_t)s * heads + hd) * PART]);
    M = wave_max(M);
    float L = 0.f;
    for (int s = c; s < splits; s += 64) {
      const float* o = base + ((int64_t)s * heads + hd) * PART;
      const float w = (o[0] == -INFINITY) ? 0.f : __expf(o[0] - M);
      cw[hd * splits + s] = w;
      L = fmaf(o[1], w, L);
    }
    L = wave_sum(L);
    if (c == 0) { sM[hd] = M; sL[hd] = L; }
  }
  __syncthreads();
  const float L = sL[hd];
  float As = 0.f, Ac = 0.f;
  const float* col = base + (int64_t)hd * PART + 2 + c;
#pragma unroll 8
  for (int s = 0; s < splits; ++s) {
    const float* o = col + (int64_t)s * heads * PART;
    As = fmaf(o[0], cw[hd * splits + s], As);
    Ac += o[HD];
  }
  // (softmax branch + CoDA branch) / n_act, n_act = 2 (models.py:140-142).  All keys masked:
  // L == 0 -> NaN, as the reference's softmax over all -inf.
  mix[(int64_t)b * heads * HD + threadIdx.x] = 0.5f * (As / L) + 0.5f * Ac;
  if (mix_softmax != nullptr) mix_softmax[(int64_t)b * heads * HD + threadIdx.x] = As / L;
  if (c == 0) {
    stats[((int64_t)b * heads + hd) * 2 + 0] = sM[hd];
    stats[((int64_t)b * heads + hd) * 2 + 1] = L;
  }
}

// out[b, h, s] = scale * Σ_c a[b, h, c] · X[b, s, h*64 + c]  (a: `a_stride` floats per head); keys of
// padded frames get `fill` when a mask is given.  Same lane mapping as the partial kernel.
template <typename T>
__global__ __launch_bounds__(1024) void decoder_rowdot_kernel(const float* __restrict__ a, int a_stride,
                                                              const T* __restrict__ X,
                                                              const uint8_t* __restrict__ frame_mask,
                                                              float* __restrict__ out, float scale, float fill, int splits,
                                                              int T_frames, int patches, int heads, int R, KvLayout lay,
                                                              FastDiv div_patches) {
  const int tpr = heads * 8;
  const int b = blockIdx.y, split = blockIdx.x;
  const int rs = threadIdx.x / tpr, tr = threadIdx.x % tpr;
  const int hd = tr >> 3, sub = tr & 7;
  const int S = T_frames * patches;
  const int D = heads * HD;
  const int per = (S + splits - 1) / splits;
  const int s_begin = split * per;
  const int s_end = min(S, s_begin + per);
  float av[8];
  {
    const float* ap = a + ((int64_t)b * heads + hd) * a_stride + sub * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) av[e] = ap[e] * scale;
  }
  const T* xb = X + (int64_t)b * T_frames * lay.frame_stride + hd * HD + sub * 8;
  const float* pb = lay.pos ? lay.pos + hd * HD + sub * 8 : nullptr;
  for (int s = s_begin + rs; s < s_end; s += R) {
    float xx[8];
    const int tf = (int)div_patches.div((uint32_t)s);
    Ld8<T>::load(xb + (int64_t)tf * lay.frame_stride + (int64_t)(s - tf * patches) * lay.row_stride, xx);
    if (pb != nullptr) {
      float pe[8];
      Ld8<float>::load(pb + (int64_t)tf * D, pe);
#pragma unroll
      for (int e = 0; e < 8; ++e) xx[e] += pe[e];
    }
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d = fmaf(av[e], xx[e], d);
    d = group8_sum(d);
    if (sub == 0) {
      const bool ok = frame_mask == nullptr || frame_mask[(int64_t)b * T_frames + tf] != 0;
      out[((int64_t)b * heads + hd) * S + s] = ok ? d : fill;
    }
  }
}

// attn_mode (models.py:107-115): the softmax branch is a sum of grouped softmaxes over the scores
// viewed [T, P]: "frame" normalises over the patches of each frame, "temporal" over the frames at
// each patch position.  One block per (head, clip); scores and weights live in LDS.
// A group whose keys are all padded gives NaN, as the reference's softmax over all -inf does.
__global__ __launch_bounds__(256) void decoder_modes_fwd_kernel(const float* __restrict__ scores, float* __restrict__ weights,
                                                                int modes, int T_frames, int patches) {
  extern __shared__ float sm[];  // [S] scores | [S] weights
  const int S = T_frames * patches;
  float* sc = sm;
  float* aw = sm + S;
  const int64_t base = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * S;
  for (int i = threadIdx.x; i < S; i += 256) { sc[i] = scores[base + i]; aw[i] = 0.f; }
  __syncthreads();
  if (modes & 1) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = wave; t < T_frames; t += 4) {
      const float* row = sc + t * patches;
      float m = -INFINITY;
      for (int p = lane; p < patches; p += 64) m = fmaxf(m, row[p]);
      m = wave_max(m);
      float l = 0.f;
      for (int p = lane; p < patches; p += 64) l += __expf(row[p] - m);
      l = wave_sum(l);
      for (int p = lane; p < patches; p += 64) aw[t * patches + p] += __expf(row[p] - m) / l;
    }
    __syncthreads();
  }
  if (modes & 2) {
    for (int p = threadIdx.x; p < patches; p += 256) {
      float m = -INFINITY;
      for (int t = 0; t < T_frames; ++t) m = fmaxf(m, sc[t * patches + p]);
      float l = 0.f;
      for (int t = 0; t < T_frames; ++t) l += __expf(sc[t * patches + p] - m);
      for (int t = 0; t < T_frames; ++t) aw[t * patches + p] += __expf(sc[t * patches + p] - m) / l;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < S; i += 256) weights[base + i] = aw[i];
}

template <int EPI>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t ldy, int B, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* wr = W + (int64_t)n * K;
  for (int b0 = 0; b0 < B; b0 += 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int kk = lane * 4; kk < K; kk += 256) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + kk);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (b0 + i < B) {
          const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (int64_t)(b0 + i) * ldx + kk);
          acc[i] = fmaf(w4[0], x4[0], fmaf(w4[1], x4[1], fmaf(w4[2], x4[2], fmaf(w4[3], x4[3], acc[i]))));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = wave_sum(acc[i]);
    if (lane < 8 && b0 + lane < B) {
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) r = (lane == i) ? acc[i] : r;
      r += bias ? bias[n] : 0.f;
      float* yp = y + (int64_t)(b0 + lane) * ldy + n;
      if constexpr (EPI == DFD_EPI_BIAS) *yp = r;
      else if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) *yp = quick_gelu(r);
      else *yp = *yp + r;
    }
  }
}

__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ proj,
                                                   float* __restrict__ feat, float* __restrict__ raw,
                                                   float* __restrict__ logits, int D, int out_dim, float eps, DfdDrop drop) {
  extern __shared__ float sh[];  // [D] feature, [out_dim] z, [8] scratch
  float* f = sh;
  float* z = sh + D;
  float* scratch = z + out_dim;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (int64_t)b * ldx;
  float s = 0.f;
  for (int c = tid; c < D; c += 256) s += xr[c];
  s = wave_sum(s);
  if (lane == 0) scratch[wave] = s;
  __syncthreads();
  const float mean = (scratch[0] + scratch[1] + scratch[2] + scratch[3]) / (float)D;
  __syncthreads();
  float qv = 0.f;
  for (int c = tid; c < D; c += 256) { const float d = xr[c] - mean; qv += d * d; }
  qv = wave_sum(qv);
  if (lane == 0) scratch[wave] = qv;
  __syncthreads();
  const float rstd = rsqrtf((scratch[0] + scratch[1] + scratch[2] + scratch[3]) / (float)D + eps);
  for (int c = tid; c < D; c += 256) {
    float o = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    o = dfd_drop_one(drop, (uint64_t)b * D + c, o);  // drop_post (models.py:342): the projection sees the dropped feature
    f[c] = o;
    feat[(int64_t)b * D + c] = o;
  }
  __syncthreads();
  for (int o = wave; o < out_dim; o += 4) {
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a = fmaf(f[c], proj[(int64_t)c * out_dim + o], a);
    a = wave_sum(a);
    if (lane == 0) z[o] = a;
  }
  __syncthreads();
  if (wave == 0) {
    float n2 = 0.f;
    for (int o = lane; o < out_dim; o += 64) n2 += z[o] * z[o];
    n2 = wave_sum(n2);
    const float inv = 5.0f / (sqrtf(n2) + 1e-10f);
    for (int o = lane; o < out_dim; o += 64) {
      raw[(int64_t)b * out_dim + o] = z[o];
      logits[(int64_t)b * out_dim + o] = z[o] * inv;
    }
  }
}

// ---- row-streaming linear: y[b, n] = Σ_k x[b, k] · Wt[k, n] ------------------------------------------
// Wt is the [K, N] transpose of an nn.Linear weight, so a wave reads 1 KB of one Wt row per
// instruction (64 lanes x float4, fully coalesced) and the B <= 16 activations x[b, k] of that row
// are wave-uniform scalars.  K is split over workgroups AND over the 4 waves of a workgroup; the waves'
// sums are added in LDS, each workgroup writes one slab and linear_t_reduce_kernel adds the slabs in a fixed
// order (deterministic) and applies bias / QuickGELU / residual.
constexpr int LT_TILE = 256;  // output columns per workgroup

__global__ __launch_bounds__(256) void linear_t_partial_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const float* __restrict__ Wt, float* __restrict__ part, int B,
                                                               int N, int K, int rows_per_wave) {
  __shared__ f32x4 red[4][16][64];  // [wave][clip][lane]: 64 KB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x * LT_TILE + lane * 4;
  const int k0 = (blockIdx.y * 4 + wave) * rows_per_wave;
  const int k1 = min(K, k0 + rows_per_wave);
  f32x4 acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (j < N) {
    const float* wp = Wt + j;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {  // 8 independent 1 KB row reads in flight per wave
      f32x4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const f32x4*>(wp + (int64_t)(k + u) * N);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int b = 0; b < 16; ++b)
          if (b < B) acc[b] += w[u] * x[(int64_t)b * ldx + k + u];
    }
    for (; k < k1; ++k) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wp + (int64_t)k * N);
#pragma unroll
      for (int b = 0; b < 16; ++b)
        if (b < B) acc[b] += w * x[(int64_t)b * ldx + k];
    }
  }
  // add the 4 waves' partial sums in a fixed order; wave w finishes clips 4w .. 4w+3
#pragma unroll
  for (int b = 0; b < 16; ++b) red[wave][b][lane] = acc[b];
  __syncthreads();
  if (j < N) {
    float* pp = part + ((int64_t)blockIdx.y * B) * N + j;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
      const int b = wave * 4 + bi;
      if (b < B) *reinterpret_cast<f32x4*>(pp + (int64_t)b * N) = (red[0][b][lane] + red[1][b][lane]) + (red[2][b][lane] + red[3][b][lane]);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void linear_t_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              const float* res, int64_t ldr, float* y, int64_t ldy, int B, int N,
                                                              int slabs) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one float4 of one row
  const int nq = N >> 2;
  if (idx >= B * nq) return;
  const int b = idx / nq, j = (idx % nq) * 4;
  f32x4 s = bias ? *reinterpret_cast<const f32x4*>(bias + j) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float* pp = part + (int64_t)b * N + j;
  const int64_t stride = (int64_t)B * N;
  int sl = 0;
  for (; sl + 8 <= slabs; sl += 8) {  // independent loads first, then a fixed-order sum
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(pp + (sl + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sl < slabs; ++sl) s += *reinterpret_cast<const f32x4*>(pp + sl * stride);
  float* yp = y + (int64_t)b * ldy + j;
  if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = quick_gelu(s[e]);
  }
  if constexpr (EPI == DFD_EPI_BIAS_RESIDUAL) s += *reinterpret_cast<const f32x4*>(res + (int64_t)b * ldr + j);  // res may be y itself
  *reinterpret_cast<f32x4*>(yp) = s;
}

// 8 K-rows per wave (32 per workgroup), all 8 row reads of a wave in flight at once
void linear_t_plan(int N, int K, int* rows_per_wave, int* ksplit) {
  (void)N;
  *rows_per_wave = 8;
  *ksplit = (K + 31) / 32;
}

int rows_per_block(int heads) {
  const int tpr = heads * 8;
  int R = (256 + tpr - 1) / tpr;
  while ((tpr * R) % 64 != 0) ++R;
  return R;
}

}  // namespace

extern "C" size_t dfd_decoder_attn_workspace(int B, int heads, int d, int splits) {
  if (B <= 0 || heads <= 0 || splits <= 0 || d != HD) return 0;
  return (size_t)B * splits * heads * PART * sizeof(float);
}

// layout checks shared by the decoder attention entry points
static int check_kv_layout(const char* who, const dfd_kv_layout_t* l, int kv_dtype, int D) {
  if (l == nullptr) return DFD_OK;
  const int per16 = kv_dtype == DFD_F32 ? 4 : 8;
  DFD_REQUIRE(l->row_stride >= D && l->frame_stride > 0 && l->row_stride % per16 == 0 && l->frame_stride % per16 == 0,
              "%s: key/value layout: row stride %lld, frame stride %lld (elements; rows must stay 16-byte aligned)", who,
              (long long)l->row_stride, (long long)l->frame_stride);
  DFD_REQUIRE(!l->pos || dfd_aligned16(l->pos), "%s: the positional embedding must be 16-byte aligned", who);
  return DFD_OK;
}

extern "C" int dfd_decoder_attn_fwd(const float* q, const void* k, const void* v, int kv_dtype, const dfd_kv_layout_t* layout,
                                    const uint8_t* frame_mask, const float* ext_weights, float* mix, float* mix_softmax,
                                    float* stats, void* workspace, int splits, int B, int T, int patches, int heads, int d,
                                    void* stream) {
  DFD_REQUIRE(q && k && v && frame_mask && mix && stats && workspace, "dfd_decoder_attn_fwd: null pointer");
  DFD_REQUIRE(d == HD, "dfd_decoder_attn_fwd: head dim %d, only 64 is supported", d);
  DFD_REQUIRE(B >= 0 && T > 0 && patches > 0 && heads > 0 && heads * HD <= 1024, "dfd_decoder_attn_fwd: bad shape");
  DFD_REQUIRE(splits > 0 && splits <= 4096, "dfd_decoder_attn_fwd: splits=%d", splits);
  DFD_REQUIRE(kv_dtype == DFD_F32 || kv_dtype == DFD_BF16, "dfd_decoder_attn_fwd: kv_dtype=%d", kv_dtype);
  DFD_REQUIRE(dfd_aligned16(k) && dfd_aligned16(v) && dfd_aligned16(q), "dfd_decoder_attn_fwd: pointers must be 16-byte aligned");
  if (int rc = check_kv_layout("dfd_decoder_attn_fwd", layout, kv_dtype, heads * HD)) return rc;
  if (B == 0) return DFD_OK;
  const KvLayout lay = dfd_kv_layout(layout, patches, heads * HD);
  const FastDiv divp = FastDiv::make((uint32_t)patches);
  const int R = rows_per_block(heads);
  const int threads = heads * 8 * R;
  DFD_REQUIRE(threads <= 1024, "dfd_decoder_attn_fwd: heads=%d needs %d threads", heads, threads);
  const int S_ = T * patches, per_ = (S_ + splits - 1) / splits;
  const int span = lay.pos ? (per_ + patches - 2) / patches + 1 : 0;  // frames a block's rows can touch
  const size_t lds = (size_t)threads * 18 * sizeof(float) + (size_t)(span < T ? span : T) * heads * HD * sizeof(float);
  DFD_REQUIRE(lds <= 160 * 1024, "dfd_decoder_attn_fwd: %zu B of LDS (use more splits)", lds);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(splits, B), block(threads);
  float* ws = static_cast<float*>(workspace);
#define PARTIAL_LAUNCH1(KT, MT, PS)                                                                                  \
  do {                                                                                                               \
    if (lds > 64 * 1024)                                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decoder_attn_partial_kernel<KT, MT, PS>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    hipLaunchKernelGGL((decoder_attn_partial_kernel<KT, MT, PS>), grid, block, lds, st, q, static_cast<const KT*>(k), \
                       static_cast<const KT*>(v), frame_mask, ext_weights, ws, splits, T, patches, heads, R, lay, divp); \
  } while (0)
#define PARTIAL_LAUNCH(KT, MT)                                                                                       \
  do {                                                                                                               \
    if (lay.pos) PARTIAL_LAUNCH1(KT, MT, true); else PARTIAL_LAUNCH1(KT, MT, false);                                 \
  } while (0)
  if (kv_dtype == DFD_F32) { if (threads <= 512) PARTIAL_LAUNCH(float, 512); else PARTIAL_LAUNCH(float, 1024); }
  else { if (threads <= 512) PARTIAL_LAUNCH(bf16_t, 512); else PARTIAL_LAUNCH(bf16_t, 1024); }
#undef PARTIAL_LAUNCH
#undef PARTIAL_LAUNCH1
  DFD_CHECK_LAUNCH("dfd_decoder_attn_fwd(partial)");
  hipLaunchKernelGGL(decoder_attn_combine_kernel, dim3(B), dim3(heads * HD), (size_t)heads * splits * sizeof(float), st, ws, mix,
                     mix_softmax, stats, splits, heads);
  DFD_CHECK_LAUNCH("dfd_decoder_attn_fwd(combine)");
  return DFD_OK;
}

// shared by the attn_mode entry points (also used from decoder_bwd.hip through dfd_decoder_rowdot)
int dfd_decoder_rowdot(const float* a, int a_stride, const void* X, int kv_dtype, const dfd_kv_layout_t* layout,
                       const uint8_t* frame_mask, float* out, float scale, float fill, int B, int T, int patches, int heads,
                       hipStream_t st) {
  const KvLayout lay = dfd_kv_layout(layout, patches, heads * HD);
  const FastDiv divp = FastDiv::make((uint32_t)patches);
  const int R = rows_per_block(heads);
  const int threads = heads * 8 * R;
  const int S = T * patches;
  const int splits = max(1, min(S / 64, max(1, 768 / max(B, 1))));
  const dim3 grid(splits, B), block(threads);
  if (kv_dtype == DFD_F32)
    hipLaunchKernelGGL((decoder_rowdot_kernel<float>), grid, block, 0, st, a, a_stride, static_cast<const float*>(X), frame_mask,
                       out, scale, fill, splits, T, patches, heads, R, lay, divp);
  else
    hipLaunchKernelGGL((decoder_rowdot_kernel<bf16_t>), grid, block, 0, st, a, a_stride, static_cast<const bf16_t*>(X),
                       frame_mask, out, scale, fill, splits, T, patches, heads, R, lay, divp);
  DFD_CHECK_LAUNCH("dfd_decoder_rowdot");
  return DFD_OK;
}

extern "C" int dfd_decoder_attn_modes_fwd(const float* q, const void* k, int kv_dtype, const dfd_kv_layout_t* layout,
                                          const uint8_t* frame_mask, int modes, float* scores, float* weights, int B, int T,
                                          int patches, int heads, int d, void* stream) {
  DFD_REQUIRE(q && k && frame_mask && scores && weights, "dfd_decoder_attn_modes_fwd: null pointer");
  DFD_REQUIRE(d == HD, "dfd_decoder_attn_modes_fwd: head dim %d, only 64 is supported", d);
  DFD_REQUIRE(B >= 0 && T > 0 && patches > 0 && heads > 0 && heads * HD <= 1024, "dfd_decoder_attn_modes_fwd: bad shape");
  DFD_REQUIRE(modes >= 1 && modes <= 3, "dfd_decoder_attn_modes_fwd: modes=%d (bit 0 frame, bit 1 temporal)", modes);
  DFD_REQUIRE(kv_dtype == DFD_F32 || kv_dtype == DFD_BF16, "dfd_decoder_attn_modes_fwd: kv_dtype=%d", kv_dtype);
  DFD_REQUIRE(dfd_aligned16(k) && dfd_aligned16(q), "dfd_decoder_attn_modes_fwd: pointers must be 16-byte aligned");
  const size_t lds = (size_t)2 * T * patches * sizeof(float);
  DFD_REQUIRE(lds <= 150 * 1024, "dfd_decoder_attn_modes_fwd: T*patches=%d too large for one LDS pass", T * patches);
  if (B == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // softmax-branch scores q_s·k/√d, -inf on padded frames (models.py:103-104); q holds [softmax | CoDA] per head
  if (int rc = check_kv_layout("dfd_decoder_attn_modes_fwd", layout, kv_dtype, heads * HD)) return rc;
  const int rc = dfd_decoder_rowdot(q, 2 * HD, k, kv_dtype, layout, frame_mask, scores, 0.125f, -INFINITY, B, T, patches, heads, st);
  if (rc != DFD_OK) return rc;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decoder_modes_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(decoder_modes_fwd_kernel, dim3(heads, B), dim3(256), lds, st, scores, weights, modes, T, patches);
  DFD_CHECK_LAUNCH("dfd_decoder_attn_modes_fwd");
  return DFD_OK;
}

extern "C" int dfd_linear_rows(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int64_t ldy,
                               int epilogue, int B, int N, int K, void* stream) {
  DFD_REQUIRE(x && W && y, "dfd_linear_rows: null pointer");
  DFD_REQUIRE(B >= 0 && B <= 64 && N > 0 && K > 0 && K % 4 == 0, "dfd_linear_rows: bad shape B=%d N=%d K=%d", B, N, K);
  DFD_REQUIRE(ldx >= K && ldx % 4 == 0 && ldy >= N, "dfd_linear_rows: bad leading dimensions");
  DFD_REQUIRE(dfd_aligned16(x) && dfd_aligned16(W), "dfd_linear_rows: x and W must be 16-byte aligned");
  if (B == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((N + 3) / 4), block(256);
  switch (epilogue) {
    case DFD_EPI_BIAS:
      hipLaunchKernelGGL((linear_rows_kernel<DFD_EPI_BIAS>), grid, block, 0, st, x, ldx, W, bias, y, ldy, B, N, K);
      break;
    case DFD_EPI_BIAS_QUICKGELU:
      hipLaunchKernelGGL((linear_rows_kernel<DFD_EPI_BIAS_QUICKGELU>), grid, block, 0, st, x, ldx, W, bias, y, ldy, B, N, K);
      break;
    case DFD_EPI_BIAS_RESIDUAL:
      hipLaunchKernelGGL((linear_rows_kernel<DFD_EPI_BIAS_RESIDUAL>), grid, block, 0, st, x, ldx, W, bias, y, ldy, B, N, K);
      break;
    default:
      dfd_set_error("dfd_linear_rows: epilogue %d unsupported", epilogue);
      return DFD_ERR_INVALID_ARG;
  }
  DFD_CHECK_LAUNCH("dfd_linear_rows");
  return DFD_OK;
}

extern "C" size_t dfd_linear_rows_t_workspace(int B, int N, int K) {
  if (B <= 0 || N <= 0 || K <= 0) return 0;
  int rpw, ks;
  linear_t_plan(N, K, &rpw, &ks);
  return (size_t)ks * (B < 16 ? B : 16) * N * sizeof(float);
}

extern "C" int dfd_linear_rows_t(const float* x, int64_t ldx, const float* Wt, const float* bias, const float* residual,
                                 int64_t ldr, float* y, int64_t ldy, int epilogue, int B, int N, int K, void* workspace,
                                 void* stream) {
  DFD_REQUIRE(x && Wt && y && workspace, "dfd_linear_rows_t: null pointer");
  DFD_REQUIRE(B >= 0 && B <= 64 && N > 0 && K > 0 && N % 4 == 0, "dfd_linear_rows_t: bad shape B=%d N=%d K=%d", B, N, K);
  DFD_REQUIRE(ldx >= K && ldy >= N && ldy % 4 == 0, "dfd_linear_rows_t: bad leading dimensions");
  DFD_REQUIRE(dfd_aligned16(Wt) && dfd_aligned16(y) && dfd_aligned16(workspace) && (!bias || dfd_aligned16(bias)),
              "dfd_linear_rows_t: Wt, y, bias and workspace must be 16-byte aligned");
  if (!residual) {  // in place
    residual = y;
    ldr = ldy;
  }
  DFD_REQUIRE(epilogue != DFD_EPI_BIAS_RESIDUAL || (dfd_aligned16(residual) && ldr >= N && ldr % 4 == 0),
              "dfd_linear_rows_t: residual rows must be 16-byte aligned");
  if (B == 0) return DFD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rpw, ks;
  linear_t_plan(N, K, &rpw, &ks);
  float* part = static_cast<float*>(workspace);
  for (int b0 = 0; b0 < B; b0 += 16) {
    const int bb = B - b0 < 16 ? B - b0 : 16;
    hipLaunchKernelGGL(linear_t_partial_kernel, dim3((N + LT_TILE - 1) / LT_TILE, ks), dim3(256), 0, st, x + (int64_t)b0 * ldx,
                       ldx, Wt, part, bb, N, K, rpw);
    DFD_CHECK_LAUNCH("dfd_linear_rows_t(partial)");
    const dim3 rg((bb * (N / 4) + 255) / 256), rb(256);
    float* yb = y + (int64_t)b0 * ldy;
    const float* rb_ = residual + (int64_t)b0 * ldr;
    switch (epilogue) {
      case DFD_EPI_BIAS:
        hipLaunchKernelGGL((linear_t_reduce_kernel<DFD_EPI_BIAS>), rg, rb, 0, st, part, bias, rb_, ldr, yb, ldy, bb, N, ks);
        break;
      case DFD_EPI_BIAS_QUICKGELU:
        hipLaunchKernelGGL((linear_t_reduce_kernel<DFD_EPI_BIAS_QUICKGELU>), rg, rb, 0, st, part, bias, rb_, ldr, yb, ldy, bb, N, ks);
        break;
      case DFD_EPI_BIAS_RESIDUAL:
        hipLaunchKernelGGL((linear_t_reduce_kernel<DFD_EPI_BIAS_RESIDUAL>), rg, rb, 0, st, part, bias, rb_, ldr, yb, ldy, bb, N, ks);
        break;
      default:
        dfd_set_error("dfd_linear_rows_t: epilogue %d unsupported", epilogue);
        return DFD_ERR_INVALID_ARG;
    }
    DFD_CHECK_LAUNCH("dfd_linear_rows_t(reduce)");
  }
  return DFD_OK;
}

extern "C" int dfd_head_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* proj,
                            float* video_feature, float* raw_logits, float* logits, int B, int D, int out_dim,
                            float eps, const dfd_dropout_t* drop_post, void* stream) {
  DFD_REQUIRE(x && gamma && beta && proj && video_feature && raw_logits && logits, "dfd_head_fwd: null pointer");
  DFD_REQUIRE(!drop_post || (drop_post->p >= 0.f && drop_post->p < 1.f && (drop_post->p == 0.f || drop_post->rng_state)),
              "dfd_head_fwd: bad dropout descriptor");
  DFD_REQUIRE(B >= 0 && D > 0 && out_dim > 0 && out_dim <= 4096 && ldx >= D, "dfd_head_fwd: bad shape");
  if (B == 0) return DFD_OK;
  const size_t lds = (size_t)(D + out_dim + 8) * sizeof(float);
  hipLaunchKernelGGL(head_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), x, ldx, gamma, beta, proj,
                     video_feature, raw_logits, logits, D, out_dim, eps, dfd_make_drop(drop_post));
  DFD_CHECK_LAUNCH("dfd_head_fwd");
  return DFD_OK;
}
