// (Since round 3 the encoder's GEMMs run on gemm256e.hip, the same tile and ring with a ping-pong K loop; this kernel serves
// the K depths that one does not: K / step odd or < 6.)
// Persistent form of the tuned bf16 GEMM (gemm256.hip describes the 256x256 tile, the LDS ring, the XOR
// swizzle and the four-phase K step; all of that is unchanged here).  What changes is what happens BETWEEN
// tiles.  Ablations on MI355X (profiles/r02_gemm_ablation.txt) put the relaunching kernel's c_fc time at
// 0.50 ms of which the MFMA loop with its loads is 0.29, the per-tile load prologue 0.05 and the epilogue 0.16
// (0.07 of it the global stores: s_endpgm waits for them, so the next tile's workgroup cannot start).  So:
//   * one workgroup per CU walks a list of tiles (grid = min(CUs, tiles));
//   * the K loop is one flat sequence of 64-deep steps across tiles: the last two steps of a tile already
//     request the first two steps of the next one, so a tile never starts with an empty ring;
//   * operands come through buffer_load ... lds with ONE descriptor per matrix and 32-bit per-lane offsets
//     (no 64-bit pointer sets to rebuild per tile); stores go through buffer_store with out-of-range offsets
//     for masked rows, so every wave issues exactly S stores per tile;
//   * vmcnt retires in issue order, so the epilogue's S stores are issued AFTER the next tile's first loads
//     and the next tile's first wait is s_waitcnt vmcnt(S): the stores drain to HBM under the next tile's
//     first K steps instead of in front of them;
//   * epilogue staging lives in the 32 KB of LDS beside the 128 KB ring (4 KB per wave, 32 rows per pass).
// Epilogues with a bf16 C: BIAS, BIAS_QUICKGELU, QKV_EXPORT (reference clip/model.py:186, :197, :208-212).
#include "gemm256p_common.hpp"

namespace {

// RB = 16-row blocks per wave: 8 -> 256-row tiles; 7 -> 224-row tiles (the launcher picks the height whose tile count
// divides best over the CUs: at M = 94,560 the 256-row grid needs 4.34 / 13.01 / 17.3 rounds of tiles for N = 768 /
// 2304 / 3072, i.e. 5 / 14 / 18; 224-row tiles need 4.96 / 14.9 / 19.8 rounds of 7/8 the work).  The staging of A is
// the same for both (256 rows: the extra 32 belong to the next tile and are simply not used).
//
// F8: the operands are OCP e4m3 bytes.  A K step is still 128 bytes of every row (= 128 elements), staged, swizzled and
// read from LDS exactly as the bf16 form: the lane's two 16-byte fragment reads of a row (chunks fq and 4 + fq) are
// concatenated into ONE 32-byte operand of v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), which sums over the
// lane group's k-set {16 fq .. 16 fq + 15} U {64 + 16 fq ..} — the same set for both operands, so every k is used once.
// That instruction takes twice the cycles of the bf16 16x16x32 form for four times the depth: per step a wave issues 32
// of them (two phases of 16) in the time of the bf16 kernel's 64, on the same bytes — twice the FLOPs per byte moved.
// The accumulator is scaled per output column (col_scale[n] = activation scale x weight scale of row n) in the epilogue.
// CF8: C is stored as e4m3 of value * out_inv_scale (the c_fc output, consumed by the next fp8 GEMM).
template <int EPI, int RB, bool F8, bool CF8>
__global__ __launch_bounds__(512) void gemm256p_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  static_assert(!F8 || RB == 8, "fp8 form: 256-row tiles only");
  static_assert(!CF8 || (F8 && EPI != DFD_EPI_QKV_EXPORT), "fp8 output: fp8 operands, plain or QuickGELU epilogue");
  constexpr int ESZ = F8 ? 1 : 2;      // bytes per operand element
  constexpr int CSZ = CF8 ? 1 : 2;     // bytes per output element
  constexpr int TMU = 32 * RB;   // rows a tile uses
  constexpr int WROWS = 16 * RB; // rows per wave
  constexpr int HB = RB - 4;     // row blocks in the second half (phases P1 / P3)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + 8 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = tiles_m * tiles_n;

  // XCD-aware, bijective position of this workgroup inside one round of the grid (blocks b and b+8 share an XCD):
  // each XCD takes a contiguous run of the tile order every round
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)(uint32_t)(a.M * a.lda * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(uint32_t)((int64_t)a.N * a.ldw * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(uint32_t)(a.M * a.ldc * CSZ), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F8 ? a.col_scale : reinterpret_cast<const float*>(a.W)), 0, F8 ? a.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias ? a.bias : reinterpret_cast<const float*>(a.W)), 0,
                                                                        a.bias ? a.N * 4 : 0, 0x00020000);

  // ---- LDS-DMA staging: wave w fills rows [32w, 32w+32) of A and of W in 8-row pieces (1 KiB each) ----------
  // per-lane byte offsets of the four pieces into A / W.  They describe the tile whose steps are being REQUESTED:
  // vA switches to the next tile two steps before the end of the current one, vW one step before.
  uint32_t vA[4], vW[4];
  const uint32_t lda2 = (uint32_t)(a.lda * ESZ), ldw2 = (uint32_t)(a.ldw * ESZ);  // row pitches in bytes
  const uint32_t a_last = (uint32_t)(a.M - 1) * lda2;  // byte offset of the last valid row: rows beyond M re-read it
  // Offsets are rebuilt from an opaque copy of the lane id in plain 32-bit arithmetic (tile part scalar, lane part
  // two or three VALU ops): nothing lane-dependent has to stay live across the tile loop for them.
  auto set_a = [&](const Tile& t) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t pr = (uint32_t)(l >> 3), pp = (uint32_t)(l & 7);
    const uint32_t ch0 = (pp ^ (pr >> 1)) << 4;  // chunk swizzle ppos ^ ((row >> 1) & 7): rows 8p + prow -> (4 (p & 1)) ^ (prow >> 1)
    const uint32_t row0 = ((uint32_t)t.m0 + (uint32_t)(wave * 32)) * lda2 + pr * lda2;
#pragma unroll
    for (int p = 0; p < 4; ++p) vA[p] = min(row0 + (uint32_t)(p * 8) * lda2, a_last) + (ch0 ^ ((p & 1) << 6));
  };
  auto set_w = [&](const Tile& t) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t pr = (uint32_t)(l >> 3), pp = (uint32_t)(l & 7);
    const uint32_t ch0 = (pp ^ (pr >> 1)) << 4;
    const uint32_t row0 = ((uint32_t)t.n0 + (uint32_t)(wave * 32)) * ldw2 + pr * ldw2;
#pragma unroll
    for (int p = 0; p < 4; ++p) vW[p] = row0 + (uint32_t)(p * 8) * ldw2 + (ch0 ^ ((p & 1) << 6));
  };
  auto issue_a = [&](int kt, int slot) {
    unsigned char* d = smem + slot * SLOT + wave * 32 * ROWB;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(d + p * 8 * ROWB), 16, vA[p], kt * ROWB, 0, 0);
  };
  auto issue_w = [&](int kt, int slot) {
    unsigned char* d = smem + slot * SLOT + A_BYTES + wave * 32 * ROWB;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdW, (lds_ptr_t)(d + p * 8 * ROWB), 16, vW[p], kt * ROWB, 0, 0);
  };

  // ---- fragment reads: lane (fr, fq) reads row fr of a 16-row block, chunk 4*ks + fq ----------------
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  int offA[2], offW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offA[ks] = (wr * WROWS + fr) * ROWB + (((4 * ks + fq) ^ sw) << 4);  // WROWS/2 is a multiple of 8: the swizzle term is unchanged
    offW[ks] = A_BYTES + (wc * 64 + fr) * ROWB + (((4 * ks + fq) ^ sw) << 4);
  }
  auto read_w = [&](bf16x8 (&w)[4], int slot, int ks) {
    const unsigned char* sb = smem + slot * SLOT + offW[ks];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * ROWB);
  };
  auto read_a = [&](bf16x8 (&f)[4], int slot, int ks, int half) {
    const unsigned char* sb = smem + slot * SLOT + offA[ks] + half * 64 * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (half == 0 || i < HB) f[i] = *reinterpret_cast<const bf16x8*>(sb + i * 16 * ROWB);
  };

  f32x4 acc[RB][4];
  auto phase = [&](const bf16x8 (&w)[4], const bf16x8 (&f)[4], int half, auto&& mid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 * half][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[0], acc[4 * half][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mid();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 1; i < (half == 0 ? 4 : HB); ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[i], acc[4 * half + i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // fp8 form: one operand = both 16-byte chunks of the lane's row
  [[maybe_unused]] auto read_w8 = [&](v8i (&w)[4], int slot, int j) {
    const unsigned char* sb = smem + slot * SLOT + j * 16 * ROWB;
    const v4i x0 = *reinterpret_cast<const v4i*>(sb + offW[0]), x1 = *reinterpret_cast<const v4i*>(sb + offW[1]);
    w[j] = v8i{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  };
  [[maybe_unused]] auto read_a8 = [&](v8i (&f)[2], int slot, int q) {  // row blocks 2q, 2q+1 of this wave's eight
    const unsigned char* sb = smem + slot * SLOT + q * 32 * ROWB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const v4i x0 = *reinterpret_cast<const v4i*>(sb + offA[0] + i * 16 * ROWB), x1 = *reinterpret_cast<const v4i*>(sb + offA[1] + i * 16 * ROWB);
      f[i] = v8i{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    }
  };
  [[maybe_unused]] auto mfma8 = [&](const v8i& w, const v8i& f, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, f, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);  // e4m3 x e4m3, block scales 2^0
  };

  const int nk = a.K / (F8 ? 128 : TK);  // >= 2
  int idx = pos;                         // < ntiles: the launcher keeps G <= ntiles
  Tile cur = decode_tile(idx, tiles_m, tiles_n, TMU);
  set_a(cur);
  set_w(cur);
  int par = 0;  // ring slot of the current tile's step 0
  issue_a(0, 0);
  issue_w(0, 0);
  issue_a(1, 1);
  issue_w(1, 1);
  wait_vm<8>();  // step 0 landed, step 1 may be in flight
  __builtin_amdgcn_s_barrier();
  int s_prev = 0;  // stores of the previous epilogue still in flight when this tile's loop starts

  [[maybe_unused]] bf16x8 wA[4], wB[4], lo[4], hi[4];
  [[maybe_unused]] v8i W8[4], aE[2], aO[2];
  [[maybe_unused]] unsigned char* const ep = smem + RING + wave * STAGE;
  const int D = EPI == DFD_EPI_QKV_EXPORT ? a.N / (3 - a.qkv_first) : 0;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t srdK = srdC, srdV = srdC, srdP = srdC;
  if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
    if (a.k_export != nullptr) {
      srdP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pos ? a.pos : reinterpret_cast<const float*>(a.W)), 0,
                                               a.pos ? a.frames_per_clip * D * 4 : 0, 0x00020000);
      const int64_t erows = (a.M / a.tokens) * (a.tokens - 1);
      srdK = __builtin_amdgcn_make_buffer_rsrc(a.k_export, 0, (int)(uint32_t)(erows * D * 2), 0x00020000);
      srdV = __builtin_amdgcn_make_buffer_rsrc(a.v_export, 0, (int)(uint32_t)(erows * D * 2), 0x00020000);
    }
  }

  for (;;) {
    const int nidx = idx + G;
    const bool has_next = nidx < ntiles;
    const Tile nxt = has_next ? decode_tile(nidx, tiles_m, tiles_n, TMU) : cur;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) read_w8(W8, par, j);
      read_a8(aE, par, 0);
    } else {
      read_w(wA, par, 0);
      read_a(lo, par, 0, 0);
    }
    f32x4 b4[4];                    // bias of this wave's 64 columns
    [[maybe_unused]] f32x4 cs4[4];  // fp8: dequantisation scale of the same columns

    // bias (and column scales), requested ahead of the last step's wait so that the epilogue never waits for a load
    auto load_col_vectors = [&] {
      int lb = lane;  // opaque: keeps the (tile-invariant) lane part of the address out of the loop-carried registers
      asm volatile("" : "+v"(lb));
      const uint32_t coff = (uint32_t)((cur.n0 + wc * 64 + (lb >> 4) * 4) * 4);
      const uint32_t boff = a.bias ? coff : 0xffffffffu;  // no bias: out of range reads 0
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdB, boff, j * 64, 0));
        if constexpr (F8) cs4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdS, coff, j * 64, 0));
      }
    };
    // ... and consumed right behind that wait: keeps the compiler from placing its own (merged, hence vmcnt(0)) wait
    // at the top of the epilogue, behind the next tile's first requests
    auto pin_col_vectors = [&] {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(b4[j]));
        if constexpr (F8) asm volatile("" : "+v"(cs4[j]));
      }
    };
    auto step_wait = [&](int kt) {
      // my reads of this slot are done and step kt+1 has landed: after the barrier this slot is free and the other
      // one readable.  In a tile's first step the youngest s_prev operations are the previous epilogue's stores:
      // they stay in flight.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (kt == 0) {
        if (s_prev == 0) wait_vm<0>();
        else if (s_prev == 2 * RB) wait_vm<2 * RB>();
        else if (s_prev == RB) wait_vm<RB>();
        else wait_vm<4 * RB>();
      } else {
        wait_vm<0>();
      }
    };

    // fp8 step (128 elements deep): four phases of 8 MFMAs, phase q = row blocks 2q, 2q+1 against all four column
    // blocks (W8); the row operands alternate between two 2-block buffers (acc 128 + W8 32 + aE 16 + aO 16 registers).
    // W8 is single-buffered: the last phase walks the column blocks and re-reads each for the next step right after
    // its last use.  Same wait / barrier placement as the bf16 step: between the third and the fourth phase.
    // (the empty asm statements tie each MFMA group to its place in program order: without them the compiler sinks
    // the groups of a step's first phases below the barrier, next to their consumers, and every operand of the step is
    // live at once)
    [[maybe_unused]] auto phase8 = [&](const v8i (&f)[2], int q, auto&& mid) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[2 * q + i][j] = mfma8(W8[j], f[i], acc[2 * q + i][j]);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(acc[2 * q + i][j]));
      __builtin_amdgcn_sched_barrier(0);
      mid();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 2; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[2 * q + i][j] = mfma8(W8[j], f[i], acc[2 * q + i][j]);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int j = 2; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(acc[2 * q + i][j]));
      __builtin_amdgcn_sched_barrier(0);
    };
    [[maybe_unused]] auto kstep8 = [&](int kt, auto last_c) {
      constexpr bool last = decltype(last_c)::value;
      const int slot = (par + kt) & 1;
      if (has_next) {
        if (kt == nk - 2) set_a(nxt);
        if (kt == nk - 1) set_w(nxt);
      }
      phase8(aE, 0, [&] {
        read_a8(aO, slot, 1);
        if (kt >= 1) {
          if (!last) issue_w(kt + 1, slot ^ 1);
          else if (has_next) issue_w(0, slot ^ 1);
        }
      });
      phase8(aO, 1, [&] { read_a8(aE, slot, 2); });
      phase8(aE, 2, [&] { read_a8(aO, slot, 3); });
      if constexpr (last) load_col_vectors();
      step_wait(kt);
      if constexpr (last) pin_col_vectors();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[6 + i][j] = mfma8(W8[j], aO[i], acc[6 + i][j]);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(acc[6 + i][j]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!last) read_w8(W8, slot ^ 1, j);  // this column block's operand of the next step
        if (j == 0) {
          if constexpr (!last) read_a8(aE, slot ^ 1, 0);
          if (kt + 2 < nk) issue_a(kt + 2, slot);
          else if (has_next) issue_a(kt + 2 - nk, slot);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // one 64-deep step; `last` (compile time) = the tile's final step, which requests only the next tile's operands,
    // fetches the bias ahead of its wait (so that the epilogue never waits for a load) and reads no further fragments
    auto kstep = [&](int kt, auto last_c) {
      constexpr bool last = decltype(last_c)::value;
      const int slot = (par + kt) & 1;
      if (has_next) {  // the requests below move on to the next tile
        if (kt == nk - 2) set_a(nxt);
        if (kt == nk - 1) set_w(nxt);
      }
      // P0: k-half 0, rows 0-3 | prefetch rows 4-7 | request W of step kt+1 (a tile's step 1 is requested before its loop)
      phase(wA, lo, 0, [&] {
        read_a(hi, slot, 0, 1);
        if (kt >= 1) {
          if (!last) issue_w(kt + 1, slot ^ 1);
          else if (has_next) issue_w(0, slot ^ 1);
        }
      });
      // P1: k-half 0, rows 4-7 | prefetch k-half 1: W (second set) and rows 0-3
      phase(wA, hi, 1, [&] {
        read_w(wB, slot, 1);
        read_a(lo, slot, 1, 0);
      });
      // P2: k-half 1, rows 0-3 | prefetch rows 4-7
      phase(wB, lo, 0, [&] { read_a(hi, slot, 1, 1); });
      if constexpr (last) load_col_vectors();
      step_wait(kt);
      if constexpr (last) pin_col_vectors();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // P3: k-half 1, rows 4-7 (registers only) | first fragments of step kt+1 | request A of step kt+2
      phase(wB, hi, 1, [&] {
        if constexpr (!last) {
          read_w(wA, slot ^ 1, 0);
          read_a(lo, slot ^ 1, 0, 0);
        }
        if (kt + 2 < nk) issue_a(kt + 2, slot);
        else if (has_next) issue_a(kt + 2 - nk, slot);
      });
    };
    if constexpr (F8) {
      for (int kt = 0; kt < nk - 1; ++kt) kstep8(kt, std::false_type{});
      kstep8(nk - 1, std::true_type{});
    } else {
      for (int kt = 0; kt < nk - 1; ++kt) kstep(kt, std::false_type{});
      kstep(nk - 1, std::true_type{});
    }
    if (has_next) issue_w(1, (par + nk - 1) & 1);  // ahead of the stores below: the next tile's first wait skips them

    // ---- epilogue ------------------------------------------------------------------------------------------------
    // Every address below is rebuilt from an opaque copy of the lane id: left to itself the compiler hoists two
    // dozen tile-invariant address registers out of the tile loop and spills them (scratch traffic counts in vmcnt
    // and would drain the stores this kernel exists to leave in flight).
    int le = lane;
    asm volatile("" : "+v"(le));
    const int er = le & 15, eq = le >> 4;          // accumulator fragment: row er of a 16-row block, columns 4*eq ..
    const int drow = le >> 3, dc = le & 7;         // drain: row drow of an 8-row group, 16-byte chunk dc
    const int nb = cur.n0 + wc * 64;
    const int64_t mrow0 = (int64_t)cur.m0 + wr * WROWS + drow;  // first row this lane stores
    const int rows_left = (int)min((int64_t)0x7fffffff, a.M - mrow0);
    int which = 0;
    if constexpr (EPI == DFD_EPI_QKV_EXPORT) which = cur.n0 / D + a.qkv_first;  // 0 = q, 1 = k, 2 = v
    const bool exporting = EPI == DFD_EPI_QKV_EXPORT && which > 0 && a.k_export != nullptr;
    int stores = 2 * RB;
    {
      // (column scale and) bias once, in place: both copies of an exported tile read the same registers
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (F8) acc[i][j] = acc[i][j] * cs4[j] + b4[j];
          else acc[i][j] += b4[j];
        }
      if constexpr (EPI == DFD_EPI_QKV_EXPORT) {
        if (exporting) {
          // Exported copy of a K / V tile FIRST (its positional-embedding loads then wait only for loads, never for
          // this tile's stores): bf16(acc + bias + pos[frame % T]) -> row frame*(tokens-1) + token-1 of the export,
          // the CLS row dropped.  Eight sub-passes of 16 rows parked as f32 (4 KB); the drain adds the embedding
          // (two 16-byte loads per store, requested at the top of the sub-pass) and rounds once.
          stores = 4 * RB;
          const int ecol = nb - (which - a.qkv_first) * D + dc * 8;  // first of this lane's 8 export columns
          unsigned char* const parkf = ep + er * 256;                 // unit (j*4 + eq) ^ er of a 256-byte row
          const __amdgpu_buffer_rsrc_t srdE = which == 2 ? srdV : srdK;
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            __builtin_amdgcn_sched_barrier(0);  // keep each sub-pass's embedding loads inside it (16 registers, not 128)
            uint32_t eoff[2];
            f32x4 pe[2][2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int rloc = i * 16 + rr * 8;
              const uint32_t m = (uint32_t)min(mrow0 + rloc, a.M - 1);
              const uint32_t frame = a.div_tokens.div(m);
              const uint32_t tok = m - frame * (uint32_t)a.tokens;
              const uint32_t t = frame - a.div_frames.div(frame) * (uint32_t)a.frames_per_clip;
              eoff[rr] = (rloc < rows_left && tok > 0) ? ((frame * (uint32_t)(a.tokens - 1) + tok - 1) * (uint32_t)D + ecol) * 2 : 0xffffffffu;
              const uint32_t poff = a.pos ? (t * (uint32_t)D + ecol) * 4 : 0xffffffffu;  // no embedding: out of range reads 0
              pe[rr][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdP, poff, 0, 0));
              pe[rr][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdP, poff, 16, 0));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
              *reinterpret_cast<f32x4*>(parkf + (((j * 4 + eq) ^ er) << 4)) = acc[i][j];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int row = rr * 8 + drow;
              const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * dc) ^ row) << 4)) + pe[rr][0];
              const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * dc + 1) ^ row) << 4)) + pe[rr][1];
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o[e] = (bf16_t)x0[e];
                o[4 + e] = (bf16_t)x1[e];
              }
              store_out(__builtin_bit_cast(v4i, o), srdE, eoff[rr], a.stream_out);
            }
          }
        }
      }
      // QuickGELU on a 4-wide fragment (packed f32 arithmetic: gemm256.hip)
      auto activate = [&](f32x4 v) {
        if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) {
          float cgelu = DFD_QUICKGELU_SCALE;  // opaque + in an SGPR so that the multiply packs
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
        return v;
      };
      if constexpr (CF8) {
        // C as e4m3 of value * out_inv_scale: 4 passes of 32 rows x 64 B parked (2 KB); 8 wave-stores of 16 rows x 64 B
        stores = RB;
        const int srow = le >> 2, sc = le & 3;  // drain: row srow of a 16-row group, 16-byte chunk sc
        unsigned char* const park8 = ep + er * 64 + eq * 4;  // + ii*1024, chunk j at position j ^ ((row >> 1) & 3)
        const int psw = (er >> 1) & 3;
        const unsigned char* const dsrc8 = ep + srow * 64 + ((sc ^ ((srow >> 1) & 3)) << 4);  // + rr*1024
        const int64_t m8 = (int64_t)cur.m0 + wr * WROWS + srow;
        const int rows_left8 = (int)min((int64_t)0x7fffffff, a.M - m8);
        const uint32_t cbase8 = (uint32_t)(m8 * a.ldc + nb + sc * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * q + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 v = activate(acc[i][j]) * a.out_inv_scale;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = __builtin_fminf(__builtin_fmaxf(v[e], -448.0f), 448.0f);  // e4m3 saturates at +-448
              unsigned pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0u, false);
              pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
              *reinterpret_cast<unsigned*>(park8 + ii * 1024 + ((j ^ psw) << 4)) = pk;
            }
          }
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const v4i d = *reinterpret_cast<const v4i*>(dsrc8 + rr * 1024);
            const int rloc = q * 32 + rr * 16;
            const uint32_t off = rloc < rows_left8 ? cbase8 + (uint32_t)rloc * (uint32_t)a.ldc : 0xffffffffu;
            store_out(d, srdC, off, a.stream_out);
          }
        }
      } else {
      // C itself: 4 passes of 32 rows parked as bf16 (4 KB); 16 wave-stores of 8 rows x 128 B
      unsigned char* const park = ep + er * 128 + ((eq ^ ((er & 7) << 1)) << 3);  // + ii*2048, ^ (j << 5)
      const unsigned char* const dsrc = ep + drow * 128 + ((dc ^ drow) << 4);     // + rr*1024
      const uint32_t cbase = (uint32_t)((mrow0 * a.ldc + nb + dc * 8) * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * q + ii;
          if (i >= RB) continue;  // 224-row tiles: the last pass holds 16 rows
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = activate(acc[i][j]);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            // row ii*16 + er, 8-byte unit (j*4 + eq) ^ ((row & 7) << 1)
            // (XOR on the LDS byte address, cast back to an LDS pointer: through a generic pointer this became flat_store)
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            *reinterpret_cast<lds_bf16x4*>(((uint32_t)(uintptr_t)(lds_ptr_t)(park + ii * 2048)) ^ (uint32_t)(j << 5)) = o;
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (q * 32 + rr * 8 >= WROWS) continue;
          const v4i d = *reinterpret_cast<const v4i*>(dsrc + rr * 1024);
          const int rloc = q * 32 + rr * 8;  // row of the store relative to this lane's first row
          uint32_t off = rloc < rows_left ? cbase + (uint32_t)rloc * (uint32_t)(a.ldc * 2) : 0xffffffffu;  // out of range: dropped
          store_out(d, srdC, off, a.stream_out);
        }
      }
      }
    }
    if (!has_next) break;
    // A store whose lanes are all out of range is dropped by the buffer unit and retires at once, ahead of older loads
    // (found with gemm256e.hip): the next tile's first wait may count this epilogue's stores only if every one of them
    // was real, i.e. all of this wave's rows are inside M.
    s_prev = (int64_t)cur.m0 + wr * WROWS + WROWS <= a.M ? stores : 0;
    par = (par + nk) & 1;
    idx = nidx;
    cur = nxt;
  }
}

template <int EPI, bool F8, bool CF8>
int launch256p(const GemmArgs& a, hipStream_t st) {
  const int tiles_n = a.N / TN;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dfd_set_error("dfd_gemm(persistent): cannot query the device");
      return DFD_ERR_LAUNCH;
    }
    n_cu = prop.multiProcessorCount;
  }
  int cus = n_cu - a.spare_cus;
  cus = cus < n_cu / 2 ? n_cu / 2 : cus;
  // tile height: the one with the least (rounds of tiles) x (cost of a tile).  A 224-row tile saves the MFMA and
  // epilogue work of 32 rows but stages as many bytes as a 256-row one, and the loop is bound by that staging:
  // measured on the four ViT-B/16 shapes it costs 0.97 of a full tile, so it wins only where it saves a whole
  // round (M = 94,560: N = 768 needs 5 rounds either way -> 224; N = 2304 / 3072: 15 vs 14, 20 vs 18 -> 256)
  auto rounds = [&](int rows) {
    const int64_t tiles = ((a.M + rows - 1) / rows) * tiles_n;
    return (double)((tiles + cus - 1) / cus);
  };
  const bool use224 = !F8 && (a.tile_rows == 224 || (a.tile_rows == 0 && rounds(224) * 0.97 < rounds(256)));
  const int rows = use224 ? 224 : 256;
  const int tiles_m = (int)((a.M + rows - 1) / rows);
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  const int grid = (int)(ntiles < cus ? ntiles : cus);
  if constexpr (F8) {
    hipLaunchKernelGGL((gemm256p_kernel<EPI, 8, true, CF8>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
  } else {
    if (use224) hipLaunchKernelGGL((gemm256p_kernel<EPI, 7, false, false>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm256p_kernel<EPI, 8, false, false>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_gemm(persistent): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}

// shared eligibility: 0 = fine, 1 = not served
int check256p(const GemmArgs& a, int esz, int csz, int kstep) {
  if (a.N % TN != 0 || a.K % kstep != 0 || a.K < 2 * kstep || a.M < 1024) return 1;
  if ((a.lda * esz) % 16 != 0 || (a.ldw * esz) % 16 != 0 || (a.ldc * csz) % 16 != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.W) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return 1;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) return 1;
  const int64_t lim = (int64_t)0xfffffff0;  // buffer descriptors carry 32-bit byte offsets
  if (a.M * a.lda * esz > lim || (int64_t)a.N * a.ldw * esz > lim || a.M * a.ldc * csz > lim) return 1;
  if ((int64_t)((a.M + 223) / 224) * (a.N / TN) > 0x3fffffff || a.M >= ((int64_t)1 << 31)) return 1;
  return 0;
}

int check_export(const GemmArgs& a) {
  if ((a.N / (3 - a.qkv_first)) % TN != 0) return 1;
  if (a.pos && (reinterpret_cast<uintptr_t>(a.pos) & 15) != 0) return 1;
  if (a.k_export && (a.M / a.tokens) * (a.tokens - 1) * (int64_t)(a.N / (3 - a.qkv_first)) * 2 > (int64_t)0xfffffff0) return 1;
  return 0;
}

}  // namespace

// 0 = launched, <0 = error, 1 = shape / epilogue not served by this kernel
int dfd_gemm256p_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st) {
  if (c_dtype != DFD_BF16 || check256p(a, 2, 2, 64)) return 1;
  switch (epi) {
    case DFD_EPI_BIAS:
      return launch256p<DFD_EPI_BIAS, false, false>(a, st);
    case DFD_EPI_BIAS_QUICKGELU:
      return launch256p<DFD_EPI_BIAS_QUICKGELU, false, false>(a, st);
    case DFD_EPI_QKV_EXPORT: {
      if (check_export(a)) return 1;
      GemmArgs b = a;
      b.div_tokens = FastDiv::make((uint32_t)a.tokens);
      b.div_frames = FastDiv::make((uint32_t)a.frames_per_clip);
      return launch256p<DFD_EPI_QKV_EXPORT, false, false>(b, st);
    }
    default:
      return 1;
  }
}

// fp8 (e4m3) operands on the block-scaled matrix cores; C bf16, or e4m3 for the plain / QuickGELU epilogues
int dfd_gemm256p_f8_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st) {
  if (c_dtype != DFD_BF16 && c_dtype != DFD_FP8) return 1;
  if (!a.col_scale || (reinterpret_cast<uintptr_t>(a.col_scale) & 15) != 0) return 1;
  if (check256p(a, 1, c_dtype == DFD_FP8 ? 1 : 2, 128)) return 1;
  switch (epi) {
    case DFD_EPI_BIAS:
      return c_dtype == DFD_FP8 ? launch256p<DFD_EPI_BIAS, true, true>(a, st) : launch256p<DFD_EPI_BIAS, true, false>(a, st);
    case DFD_EPI_BIAS_QUICKGELU:
      return c_dtype == DFD_FP8 ? launch256p<DFD_EPI_BIAS_QUICKGELU, true, true>(a, st) : launch256p<DFD_EPI_BIAS_QUICKGELU, true, false>(a, st);
    case DFD_EPI_QKV_EXPORT: {
      if (c_dtype != DFD_BF16 || check_export(a)) return 1;
      GemmArgs b = a;
      b.div_tokens = FastDiv::make((uint32_t)a.tokens);
      b.div_frames = FastDiv::make((uint32_t)a.frames_per_clip);
      return launch256p<DFD_EPI_QKV_EXPORT, true, false>(b, st);
    }
    default:
      return 1;
  }
}
