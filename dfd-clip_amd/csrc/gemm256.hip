// Tuned bf16 GEMM for the encoder's large shapes: C = epilogue(A[M,K] · W[N,K]ᵀ), M ~ 1e5,
// N in {768, 2304, 3072}, K in {768, 3072} (reference clip/model.py:186, :197, :208-212, :277).
//
// Structure (CDNA4, one 512-thread workgroup = 8 waves per CU, 2 waves per SIMD):
//   * 256x256 output tile; waves laid out 2 (M) x 4 (N), each wave a 128x64 patch = 8x4 tiles of
//     v_mfma_f32_16x16x32_bf16 (128 accumulator registers per lane);
//   * K is consumed 64 per step from two 64 KB LDS slots (A 256x64 | W 256x64 bf16).  Operands go
//     L2/HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip).  LDS rows are 128 B, so
//     every request is a FULL 128-byte cache line (64-byte rows halve the L1-miss-path rate:
//     measured 16.6 vs ~34 B/clk/CU);
//   * a step has four phases of 16 MFMAs (k-half x row-half).  Every phase first issues 4 MFMAs
//     on fragments already in registers, then the ds_read_b128s that feed the NEXT phase (second
//     W register set), then its other 12 MFMAs: no MFMA waits on an LDS read of its own phase;
//   * the next step's operands are issued in two batches (A in phase 3 of the previous step, W in
//     phase 0) and retired by ONE s_waitcnt vmcnt(0) + raw s_barrier per step, placed between
//     phases 2 and 3: phase 3 runs from registers, so after that barrier the current slot is
//     free for refilling and the other slot is readable — the loads get ~3 phases to land;
//   * the 16-byte chunk c of LDS row r sits at chunk position c ^ ((r >> 1) & 7): applied to the
//     per-lane SOURCE address of the lane-linear LDS-DMA write and again on the ds_read_b128
//     side, it makes every 16-lane group of a fragment read conflict-free;
//   * the product is formed transposed (W fragment as the MFMA A operand), so a lane ends up
//     with 4 consecutive output columns of one row: the fused epilogues store 8 B (bf16) or
//     read-modify-write 16 B (f32 residual stream) per lane per fragment;
//   * workgroup ids are remapped so that each XCD (own L2) walks a contiguous run of tiles, ordered
//     in column groups of <= 6 tiles (panel major inside a group): the A panel is shared from L2 by
//     the group's column tiles and the group's slice of W stays L2-resident across panels.
// Rows beyond M are clamped on load and masked on store; N % 256 == 0, K % 64 == 0, K >= 128.
#include <type_traits>

#include "gemm_args.hpp"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int ROWB = TK * 2;              // 128 B per LDS row = one cache line
constexpr int A_BYTES = TM * ROWB;        // 32 KB
constexpr int SLOT = (TM + TN) * ROWB;    // 64 KB
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void wg_barrier() { __builtin_amdgcn_s_barrier(); }

template <typename CT, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmArgs a, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SLOT];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware, bijective remap of the workgroup id (blocks b and b+8 share an XCD)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // Tile order inside the remapped id: column GROUPS of at most 6 tiles, and inside a group row panel
  // major / column minor.  An XCD's contiguous run of ids then stays inside one group for a long time:
  // the group's slice of W (<= 6 x 393 KB at K = 768) stays resident in that XCD's 4 MB L2 while the
  // A panels stream through, instead of all of W being re-fetched every ~3 panels.
  const int ngroups = (tiles_n + 5) / 6;
  const int gcols = (tiles_n + ngroups - 1) / ngroups;     // columns per (full) group
  const int tiles_m_all = nwg / tiles_n;
  int grp = wg / (tiles_m_all * gcols);
  grp = grp < ngroups - 1 ? grp : ngroups - 1;
  const int rem = wg - grp * tiles_m_all * gcols;
  const int cols_here = min(gcols, tiles_n - grp * gcols);
  const int tile_m = rem / cols_here, tile_n = grp * gcols + (rem - tile_m * cols_here);
  const int64_t m0 = (int64_t)tile_m * TM;
  const int n0 = tile_n * TN;

  // ---- LDS-DMA staging: 8-row pieces (1 KiB); wave w fills rows [32w, 32w+32) of A and of W ----------
  const int prow = lane >> 3;  // row inside a piece
  const int ppos = lane & 7;   // chunk position inside the row
  const unsigned char* ga[4];
  const unsigned char* gw[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = wave * 32 + p * 8 + prow;
    const int chunk = ppos ^ ((r >> 1) & 7);
    int64_t am = m0 + r;
    am = am < a.M ? am : a.M - 1;
    ga[p] = static_cast<const unsigned char*>(a.A) + (am * a.lda) * 2 + chunk * 16;
    gw[p] = static_cast<const unsigned char*>(a.W) + ((int64_t)(n0 + r) * a.ldw) * 2 + chunk * 16;
  }
  auto issue_a = [&](int kt) {
    unsigned char* d = smem + (kt & 1) * SLOT + wave * 32 * ROWB;
#pragma unroll
    for (int p = 0; p < 4; ++p) glds16(ga[p] + (int64_t)kt * ROWB, d + p * 8 * ROWB);
  };
  auto issue_w = [&](int kt) {
    unsigned char* d = smem + (kt & 1) * SLOT + A_BYTES + wave * 32 * ROWB;
#pragma unroll
    for (int p = 0; p < 4; ++p) glds16(gw[p] + (int64_t)kt * ROWB, d + p * 8 * ROWB);
  };

  // ---- fragment reads: lane (fr, fq) reads row fr of a 16-row block, chunk 4*ks + fq ----------------
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;  // rows of every block are congruent to fr mod 16
  int offA[2], offW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offA[ks] = (wr * 128 + fr) * ROWB + (((4 * ks + fq) ^ sw) << 4);
    offW[ks] = A_BYTES + (wc * 64 + fr) * ROWB + (((4 * ks + fq) ^ sw) << 4);
  }
  auto read_w = [&](bf16x8 (&w)[4], int kt, int ks) {
    const unsigned char* sb = smem + (kt & 1) * SLOT + offW[ks];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = lds_frag(sb + j * 16 * ROWB);
  };
  auto read_a = [&](bf16x8 (&f)[4], int kt, int ks, int half) {
    const unsigned char* sb = smem + (kt & 1) * SLOT + offA[ks] + half * 64 * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = lds_frag(sb + i * 16 * ROWB);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 16 MFMAs of one phase: rows [4*half, 4*half+4) x 4 column blocks; `mid` runs after the first 4
  auto phase = [&](const bf16x8 (&w)[4], const bf16x8 (&f)[4], int half, auto&& mid) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 * half][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[0], acc[4 * half][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    mid();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 1; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[i], acc[4 * half + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = a.K / TK;
  bf16x8 wA[4], wB[4], lo[4], hi[4];
  issue_a(0);
  issue_w(0);
  issue_a(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // step 0 landed (A of step 1 may be in flight)
  wg_barrier();
  read_w(wA, 0, 0);
  read_a(lo, 0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    // P0: k-half 0, rows 0-3 | prefetch rows 4-7 | issue W of step kt+1
    phase(wA, lo, 0, [&] {
      read_a(hi, kt, 0, 1);
      if (more) issue_w(kt + 1);
    });
    // P1: k-half 0, rows 4-7 | prefetch k-half 1: W (second set) and rows 0-3
    phase(wA, hi, 1, [&] {
      read_w(wB, kt, 1);
      read_a(lo, kt, 1, 0);
    });
    // P2: k-half 1, rows 0-3 | prefetch rows 4-7
    phase(wB, lo, 0, [&] { read_a(hi, kt, 1, 1); });
    // my LDS reads of this slot are complete, step kt+1 has landed: after the barrier this slot is
    // free for step kt+2 and the other slot is readable by every wave
    // (on the last step the barrier also orders every wave's slot reads before the epilogue's LDS use)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // P3: k-half 1, rows 4-7 (registers only) | prefetch step kt+1's first fragments | issue A of step kt+2
    phase(wB, hi, 1, [&] {
      if (more) {
        read_w(wA, kt + 1, 0);
        read_a(lo, kt + 1, 0, 0);
      }
      if (kt + 2 < nk) issue_a(kt + 2);
    });
  }

  // ---- LDS-staged epilogues -------------------------------------------------------------------------
  // The ring is dead now, so each wave owns 16 KB of it.  The wave parks its 128x64 patch there
  // (conflict-free XOR layouts) and reads it back row-contiguous: global traffic becomes whole
  // 128-byte (bf16) / 256-byte (f32) row segments moved 16 B per lane, instead of the 32-byte
  // segments of 8-byte stores that the accumulator layout gives directly.
  [[maybe_unused]] unsigned char* const ep = smem + wave * 16384;
  if constexpr (sizeof(CT) == 2 && (EPI == DFD_EPI_BIAS || EPI == DFD_EPI_BIAS_QUICKGELU || EPI == DFD_EPI_QKV_EXPORT)) {
    const int nb = n0 + wc * 64;
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b4[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nb + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    int D = 0, which = 0;
    if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
      D = a.N / (3 - a.qkv_first);
      which = n0 / D + a.qkv_first;  // 0 = q, 1 = k, 2 = v (tiles never straddle: D % 256 == 0)
    }
    const bool exporting = EPI == DFD_EPI_QKV_EXPORT && which > 0 && a.k_export != nullptr;
    const int passes = exporting ? 2 : 1;
    for (int pass = 0; pass < passes; ++pass) {
      // park: row = i*16 + fr, 8-byte unit u = j*4 + fq at unit position u ^ ((row & 7) << 1)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = i * 16 + fr;
        f32x4 p4[4];
        if (pass == 1) {  // exported copy = f32 value + temporal positional embedding, rounded once
          const int64_t m = m0 + wr * 128 + row;
          // 32-bit unsigned division (M < 2^31 is checked by the launcher): the 64-bit form costs ~4x the instructions
          const uint32_t frame = (uint32_t)(m < a.M ? m : a.M - 1) / (uint32_t)a.tokens;
          const float* pr = a.pos ? a.pos + (int64_t)(frame % (uint32_t)a.frames_per_clip) * D + (nb - (which - a.qkv_first) * D) + fq * 4 : nullptr;
#pragma unroll
          for (int j = 0; j < 4; ++j) p4[j] = pr ? *reinterpret_cast<const f32x4*>(pr + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = acc[i][j] + b4[j];
          if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) {
            // v * sigmoid(1.702 v) = v / (1 + 2^(-1.702*log2(e)*v)): the epilogue of this shape is VALU-bound
            // (two waves per SIMD, no MFMA left to hide behind), so the scale constants are folded into
            // one packed multiply and everything but v_exp_f32 / v_rcp_f32 stays in packed f32 ops
            float cgelu = DFD_QUICKGELU_SCALE;  // -1.702 * log2(e); opaque + in an SGPR so that the multiply packs
            asm volatile("" : "+s"(cgelu));
            const f32x4 t = v * cgelu;
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_exp2f(t[e]);
            d = d + 1.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_rcpf(d[e]);
            v = v * d;
          }
          if (pass == 1) v += p4[j];
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(ep + row * 128 + (((j * 4 + fq) ^ ((row & 7) << 1)) << 3)) = o;
        }
      }
      // drain: 16 wave-stores of 8 rows x 128 B
      bf16_t* dst = static_cast<bf16_t*>(pass == 0 ? a.C : (which == 2 ? a.v_export : a.k_export));
#pragma unroll 4
      for (int rr = 0; rr < 16; ++rr) {
        const int row = rr * 8 + (lane >> 3), c = lane & 7;
        const uint4 d = *reinterpret_cast<const uint4*>(ep + row * 128 + ((c ^ (row & 7)) << 4));
        const int64_t m = m0 + wr * 128 + row;
        if (m < a.M) {
          // stream_out: non-temporal stores, the output goes past L2 instead of evicting the operand panels the other
          // workgroups of the XCD are still reading (c_fc 0.486 -> 0.446 ms, QKV 0.361 -> 0.326 ms on MI355X)
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 dn{d.x, d.y, d.z, d.w};
          if (pass == 0) {
            u32x4* cp = reinterpret_cast<u32x4*>(dst + m * a.ldc + nb + c * 8);
            if (a.stream_out) __builtin_nontemporal_store(dn, cp);
            else *cp = dn;
          } else {
            const uint32_t frame = (uint32_t)m / (uint32_t)a.tokens;
            const int tok = (int)((uint32_t)m - frame * (uint32_t)a.tokens);
            if (tok > 0) {
              u32x4* ep_ = reinterpret_cast<u32x4*>(dst + ((int64_t)frame * (a.tokens - 1) + tok - 1) * D + (nb - (which - a.qkv_first) * D) + c * 8);
              if (a.stream_out) __builtin_nontemporal_store(dn, ep_);
              else *ep_ = dn;
            }
          }
        }
      }
    }
    return;
  } else if constexpr (EPI == DFD_EPI_RESIDUAL_POS && sizeof(CT) == 2) {
    // C(bf16) += acc + pos[frame % T]: park acc + pos as bf16?  No: the sum must be rounded once, so the
    // f32 values are parked (two passes of 64 rows) and the read-modify-write happens at drain time.
    const int nb = n0 + wc * 64;
    const int rows_per_frame = a.tokens - 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(ep + row * 256 + (((j * 4 + fq) ^ (row & 15)) << 4)) = acc[half * 4 + i][j];
      }
      // drain: lane handles 8 consecutive columns (two parked 16-byte units) of one row: 8 rows x 128 B per store
#pragma unroll 4
      for (int rr = 0; rr < 8; ++rr) {
        const int row = rr * 8 + (lane >> 3), c = lane & 7;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * c) ^ (row & 15)) << 4));
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
        const int64_t m = m0 + wr * 128 + half * 64 + row;
        if (m < a.M) {
          bf16x8* cp = reinterpret_cast<bf16x8*>(static_cast<bf16_t*>(a.C) + m * a.ldc + nb + c * 8);
          const bf16x8 old = a.residual ? *reinterpret_cast<const bf16x8*>(static_cast<const bf16_t*>(a.residual) + m * a.ldc + nb + c * 8) : *cp;
          f32x4 p0 = f32x4{0.f, 0.f, 0.f, 0.f}, p1 = p0;
          if (a.pos) {
            const float* pr = a.pos + (int64_t)(((uint32_t)m / (uint32_t)rows_per_frame) % (uint32_t)a.frames_per_clip) * a.N + nb + c * 8;
            p0 = *reinterpret_cast<const f32x4*>(pr);
            p1 = *reinterpret_cast<const f32x4*>(pr + 4);
          }
          float dv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { dv[e] = d0[e]; dv[4 + e] = d1[e]; }
          dfd_drop_eight(a.drop, (uint64_t)m * a.N + nb + c * 8, dv);  // the adapter's last nn.Dropout, before the residual add
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (bf16_t)((float)old[e] + dv[e] + p0[e]);
            o[4 + e] = (bf16_t)((float)old[4 + e] + dv[4 + e] + p1[e]);
          }
          *cp = o;
        }
      }
    }
    return;
  } else if constexpr (EPI == DFD_EPI_BIAS_RESIDUAL) {
    const int nb = n0 + wc * 64;
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b4[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nb + j * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // park 64 rows x 64 f32: 16-byte unit u = j*4 + fq of row at unit position u ^ (row & 15)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(ep + row * 256 + (((j * 4 + fq) ^ (row & 15)) << 4)) = acc[half * 4 + i][j] + b4[j];
      }
      // drain: 16 read-modify-writes of 4 rows x 256 B of the residual stream
#pragma unroll 4
      for (int rr = 0; rr < 16; ++rr) {
        const int row = rr * 4 + (lane >> 4), c = lane & 15;
        const f32x4 d = *reinterpret_cast<const f32x4*>(ep + row * 256 + ((c ^ (row & 15)) << 4));
        const int64_t m = m0 + wr * 128 + half * 64 + row;
        if (m < a.M) {
          f32x4* xp = reinterpret_cast<f32x4*>(static_cast<float*>(a.C) + m * a.ldc + nb + c * 4);
          *xp = *xp + d;
        }
      }
    }
    return;
  }
  // ---- epilogue: lane holds, per (i, j), row m = .. + i*16 + fr, columns n .. n+3 ----------------
  const int nbase = n0 + wc * 64 + fq * 4;
  f32x4 bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bias4[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nbase + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};

  int D = 0, which = 0;
  if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
    D = a.N / (3 - a.qkv_first);
    which = n0 / D + a.qkv_first;  // 0 = q, 1 = k, 2 = v (tiles never straddle: D % 256 == 0)
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = m0 + wr * 128 + i * 16 + fr;
    if (m >= a.M) continue;
    if constexpr (EPI == DFD_EPI_BIAS || EPI == DFD_EPI_BIAS_QUICKGELU || EPI == DFD_EPI_QKV_EXPORT) {
      int64_t erow = -1;
      const float* prow_pos = nullptr;
      if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
        if (which > 0 && a.k_export != nullptr) {
          const int64_t frame = m / a.tokens;
          const int tok = (int)(m - frame * a.tokens);
          if (tok > 0) {
            erow = frame * (a.tokens - 1) + tok - 1;
            if (a.pos) prow_pos = a.pos + (frame % a.frames_per_clip) * D;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nbase + j * 16;
        f32x4 v = acc[i][j] + bias4[j];
        if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        if constexpr (sizeof(CT) == 2) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(a.C) + m * a.ldc + n) = o;
        } else {
          *reinterpret_cast<f32x4*>(static_cast<float*>(a.C) + m * a.ldc + n) = v;
        }
        if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
          if (erow >= 0) {
            const int cc = n - (which - a.qkv_first) * D;
            if (prow_pos) v += *reinterpret_cast<const f32x4*>(prow_pos + cc);
            void* dst = which == 2 ? a.v_export : a.k_export;
            if constexpr (sizeof(CT) == 2) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
              *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(dst) + erow * D + cc) = o;
            } else {
              *reinterpret_cast<f32x4*>(static_cast<float*>(dst) + erow * D + cc) = v;
            }
          }
        }
      }
    } else if constexpr (EPI == DFD_EPI_BIAS_RESIDUAL) {
      float* xr = static_cast<float*>(a.C) + m * a.ldc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4* p = reinterpret_cast<f32x4*>(xr + nbase + j * 16);
        *p = *p + (acc[i][j] + bias4[j]);
      }
    } else if constexpr (EPI == DFD_EPI_PATCH_EMBED) {
      const int P = a.tokens - 1;
      const int64_t frame = m / P;
      const int p = (int)(m - frame * P);
      float* xr = static_cast<float*>(a.C) + (frame * a.tokens + 1 + p) * a.ldc;
      const float* pr = a.pos + (int64_t)(1 + p) * a.N;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nbase + j * 16;
        *reinterpret_cast<f32x4*>(xr + n) = acc[i][j] + *reinterpret_cast<const f32x4*>(pr + n);
        if (p == 0)
          *reinterpret_cast<f32x4*>(static_cast<float*>(a.C) + (frame * a.tokens) * a.ldc + n) =
              *reinterpret_cast<const f32x4*>(a.cls + n) + *reinterpret_cast<const f32x4*>(a.pos + n);
      }
    }
  }
}

template <typename CT, int EPI>
int launch256(const GemmArgs& a, hipStream_t st) {
  const int tiles_n = a.N / TN;
  const int tiles_m = (int)((a.M + TM - 1) / TM);
  hipLaunchKernelGGL((gemm256_kernel<CT, EPI>), dim3(tiles_m * tiles_n), dim3(512), 0, st, a, tiles_n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_gemm(tuned bf16): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}

}  // namespace

int dfd_gemm256_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st) {
  if (a.N % TN != 0 || a.K % 64 != 0 || a.K < 128 || a.M < 1024) return 1;
  if ((a.lda % 8) != 0 || (a.ldw % 8) != 0 || (a.ldc % 4) != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return 1;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) return 1;
  if ((int64_t)((a.M + TM - 1) / TM) * (a.N / TN) > 0x7fffffff) return 1;
  if (a.M >= ((int64_t)1 << 31)) return 1;  // the epilogues index rows with 32-bit arithmetic
  switch (epi) {
    case DFD_EPI_BIAS:
      return c_dtype == DFD_BF16 ? launch256<bf16_t, DFD_EPI_BIAS>(a, st) : launch256<float, DFD_EPI_BIAS>(a, st);
    case DFD_EPI_BIAS_QUICKGELU:
      return c_dtype == DFD_BF16 ? launch256<bf16_t, DFD_EPI_BIAS_QUICKGELU>(a, st) : launch256<float, DFD_EPI_BIAS_QUICKGELU>(a, st);
    case DFD_EPI_QKV_EXPORT:
      if ((a.N / (3 - a.qkv_first)) % TN != 0) return 1;
      if (a.pos && (reinterpret_cast<uintptr_t>(a.pos) & 15) != 0) return 1;
      return c_dtype == DFD_BF16 ? launch256<bf16_t, DFD_EPI_QKV_EXPORT>(a, st) : launch256<float, DFD_EPI_QKV_EXPORT>(a, st);
    case DFD_EPI_BIAS_RESIDUAL:
      return c_dtype == DFD_F32 ? launch256<float, DFD_EPI_BIAS_RESIDUAL>(a, st) : 1;
    case DFD_EPI_RESIDUAL_POS:
      if (a.pos && (reinterpret_cast<uintptr_t>(a.pos) & 15) != 0) return 1;
      if ((a.ldc % 8) != 0 || (reinterpret_cast<uintptr_t>(a.residual) & 15) != 0) return 1;
      return c_dtype == DFD_BF16 ? launch256<bf16_t, DFD_EPI_RESIDUAL_POS>(a, st) : 1;
    case DFD_EPI_PATCH_EMBED:
      if ((reinterpret_cast<uintptr_t>(a.pos) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.cls) & 15) != 0) return 1;
      return c_dtype == DFD_F32 ? launch256<float, DFD_EPI_PATCH_EMBED>(a, st) : 1;
    default:
      return 1;
  }
}
