// Train-mode dropout (reference src/models.py:163, :294, :304 decoder; :804-912 adapter): counter-based masks
// generated inside the kernels that produce the dropped tensor and REGENERATED, not stored, in the backward.
//
// Mask of logical element e of a dropout site:  group g = e >> 3 draws one Philox4x32-10 block
//     counter = (g_lo, g_hi, step_lo, step_hi),  key = (seed_lo ^ site * 0x9E3779B9, seed_hi)
// whose 128 bits are eight 16-bit draws; element e uses draw (e & 7) and is KEPT iff draw >= thr16,
// thr16 = round(p * 65536); kept values are scaled by 65536 / (65536 - thr16) (exactly unbiased for the
// realised keep probability).  `seed` and `step` are read from DEVICE memory (uint64[2]) so that a captured
// HIP graph draws fresh masks on every replay: the host bumps `step` between steps.
// oracle/dropout_mask.py restates this in numpy; tests compare the two bit for bit.
#pragma once
#include "common.hpp"

struct DfdDrop {
  const uint64_t* rng;  // device: {seed, step}; nullptr = no dropout
  uint32_t site;
  uint32_t thr16;       // 0 = no dropout
  float scale;
};

static inline DfdDrop dfd_make_drop(const dfd_dropout_t* d) {
  DfdDrop o{nullptr, 0u, 0u, 1.0f};
  if (d && d->rng_state && d->p > 0.f) {
    uint32_t t = (uint32_t)(d->p * 65536.0f + 0.5f);
    if (t > 65535u) t = 65535u;
    o.rng = d->rng_state;
    o.site = d->site;
    o.thr16 = t;
    o.scale = t ? 65536.0f / (float)(65536u - t) : 1.0f;
  }
  return o;
}

__device__ __forceinline__ uint4 dfd_philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = uint4{hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0};
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

// the eight 16-bit draws of group g (element indices 8g .. 8g+7), packed two per word, low half first
__device__ __forceinline__ uint4 dfd_drop_draws(const DfdDrop& d, uint64_t g) {
  const uint64_t seed = d.rng[0], step = d.rng[1];
  const uint4 c{(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
  const uint2 k{(uint32_t)seed ^ (d.site * 0x9E3779B9u), (uint32_t)(seed >> 32)};
  return dfd_philox4x32_10(c, k);
}

__device__ __forceinline__ uint32_t dfd_draw16(const uint4& w, int j) {  // j = e & 7
  const uint32_t word = j < 4 ? (j < 2 ? w.x : w.y) : (j < 6 ? w.z : w.w);
  return (j & 1) ? (word >> 16) : (word & 0xffffu);
}

// one element (kernels that hold scattered elements: one Philox block per call)
__device__ __forceinline__ float dfd_drop_one(const DfdDrop& d, uint64_t e, float v) {
  if (d.thr16 == 0) return v;
  const uint4 w = dfd_drop_draws(d, e >> 3);
  return dfd_draw16(w, (int)(e & 7)) >= d.thr16 ? v * d.scale : 0.f;
}

// eight consecutive elements starting at e0 (e0 % 8 == 0)
__device__ __forceinline__ void dfd_drop_eight(const DfdDrop& d, uint64_t e0, float* v) {
  if (d.thr16 == 0) return;
  const uint4 w = dfd_drop_draws(d, e0 >> 3);
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t r = (j & 1) ? (ws[j >> 1] >> 16) : (ws[j >> 1] & 0xffffu);
    v[j] = r >= d.thr16 ? v[j] * d.scale : 0.f;
  }
}
