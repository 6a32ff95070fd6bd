// Persistent 256x256 GEMM with a PING-PONG K loop: the encoder's GEMM since round 3 (bf16 and e4m3 operands; epilogues
// BIAS, BIAS_QUICKGELU, QKV_EXPORT with a bf16 C, e4m3 C for the first two of the fp8 form; reference clip/model.py:186,
// :197, :208-212).  gemm256p.hip (round 2) stays as the fallback for the K depths this loop does not serve.
//
// gemm256p runs the same software-pipelined instruction stream in all eight waves: the two waves of a SIMD reach their
// MFMA groups, their LDS read bursts and the one barrier per step together, and the matrix pipe waits whenever both of
// them do (measured busy: 50-53 %).  Here the two wave groups (wr = 0 / 1: one wave of each on every SIMD) run ONE
// BARRIER APART: while a group issues the 16 MFMAs of a quadrant of its 128x64 output, the other group does its LDS
// fragment reads and requests the next half tile; then they swap (cdna_hip_programming.md, "The 256^2 8-phase template").
// Per 128-byte-deep K tile (64 bf16 / 128 e4m3 elements) four phases, each
//     L: ds_read the operands of the phase's quadrant | 2 LDS-DMA pieces (one 16 KB "unit") | counted vmcnt | barrier
//     M: lgkmcnt(0) | 16 x v_mfma_f32_16x16x32_bf16 (8 x v_mfma_scale_f32_16x16x128_f8f6f4) | barrier
// A unit = one half (128 rows) of A or of W for one K tile.  The halves are INTERLEAVED so that a wave's output stays
// one contiguous block: LDS row (64 wr' + r) of A-half ha is tile row WROWS wr' + 64 ha + r, LDS row (32 wc' + c) of
// W-half hb is tile column 64 wc' + 32 hb + c.  Units are requested in the order they are read — W0, A0, W1, A1 of K
// tile 0, W0, A0, ... — one per L segment, DEPTH segments ahead of the segment that reads them (a unit's buffer is
// reused by the unit 8 later; DEPTH <= 6 keeps the request behind the other group's last read of it).  The wait at the
// end of an L segment retires the unit the NEXT L segment reads, so the other group's pieces are covered by the two
// barriers in between, and leaves 2 (DEPTH - 1) younger requests in flight: nothing in the loop waits for vmcnt(0).
// The unit sequence is flat across the tiles a workgroup walks; the epilogue's stores stay in flight under the next
// tile's first segments, whose waits count them.
//
// Two things the counted waits depend on (both found the hard way, tools/lab/gemm256e_lab.hip):
//   * a register load issued beside LDS-DMA makes the compiler wait vmcnt(0) at its first use, so the bias (and the
//     fp8 column scales) are fetched by inline-asm loads it does not see, counted by hand like everything else;
//   * a store whose lanes are ALL out of range is dropped by the buffer unit and retires at once, ahead of older loads:
//     a wait that counts such stores waits for nothing.  A wave therefore counts its previous epilogue's stores only
//     when every one of them was a real store (all of its rows inside M); otherwise it waits as if there were none.
// OPT-IN (dfd_gemm_set_variant(3); off by default): tiles after a workgroup's first handed out dynamically WITHIN an XCD —
// the workgroups that share an XCD label (blockIdx & 7) draw the label's tile list (the same XCD-contiguous runs the
// static order gives them) from one counter, so that a workgroup that starts late or runs slowly because another
// stream's kernels hold its CU (an RCCL all-reduce, say) takes fewer tiles instead of making the launch wait for its full
// share.  One returning atomic per tile, issued by wave 0 (EXEC narrowed to one lane: an out-of-range lane offset makes an
// atomic FAULT, it is not dropped like a store) at the top of a tile for the tile after it, counted in that wave's waits
// and published through LDS with explicit DS instructions; the counters are monotonic (the launcher passes the value
// each will have when the launch starts), so nothing is reset.  Captured launches keep the static order.  Measured on an
// undisturbed chip: 0.4 % slower than the static order (profiles/r03_gemm_dynamic_vs_static_ab.txt), bit-identical.
//
// Measured on MI355X against gemm256p (same process, interleaved; profiles/r03_gemm_pingpong_lab.txt): c_fc 0.411 ->
// 0.398-0.403 ms, q|k|v 0.285-0.292 -> 0.272-0.279, c_proj 0.388-0.395 -> 0.343-0.349, 8192^3 1,434 -> 1,590 TFLOP/s;
// bit-identical to it on every shape; without the stagger (same code, groups in lockstep) 10-15 % slower.
#include <mutex>
#include <unordered_map>

#include "gemm256p_common.hpp"

namespace {

constexpr int UNIT = 128 * ROWB;  // 16 KB: half of A (128 rows) or of W for one K tile
constexpr int DEPTH = 6;          // L segments between a unit's request and its read (4..6 measure the same; 6 tolerates the most latency)
constexpr int NB = 2 * (DEPTH - 1);  // requests younger than the unit a segment waits for

template <int EPI, int RB, bool F8, bool CF8>
__global__ __launch_bounds__(512) void gemm256e_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  static_assert(!F8 || RB == 8, "fp8 form: 256-row tiles only");
  static_assert(!CF8 || (F8 && EPI != DFD_EPI_QKV_EXPORT), "fp8 output: fp8 operands, plain or QuickGELU epilogue");
  constexpr int ESZ = F8 ? 1 : 2;  // bytes per operand element
  constexpr int CSZ = CF8 ? 1 : 2;  // bytes per output element
  constexpr int TMU = 32 * RB;    // rows a tile uses
  constexpr int WROWS = 16 * RB;  // rows per wave
  constexpr int HB = RB - 4;      // row blocks in the second half
  constexpr int NCV = F8 ? 2 : 1;  // column-vector loads per wave and tile (bias; fp8: + column scales)
  constexpr int S1 = CF8 ? RB : 2 * RB;  // stores of an epilogue per wave (S2: a tile that also exports)
  constexpr int S2 = 4 * RB;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + 8 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = tiles_m * tiles_n;

  // XCD-aware, bijective position of this workgroup inside one round of the grid (blocks b and b+8 share an XCD)
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)(uint32_t)(a.M * a.lda * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(uint32_t)((int64_t)a.N * a.ldw * ESZ), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(uint32_t)(a.M * a.ldc * CSZ), 0x00020000);
  // descriptors of the column vectors as four plain words: their loads are inline asm
  auto words = [](const float* p, int bytes) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    return v4i{(int)(uint32_t)u, (int)((uint32_t)(u >> 32) & 0xffffu), bytes, 0x00020000};
  };
  const v4i srdB = words(a.bias ? a.bias : reinterpret_cast<const float*>(a.W), a.bias ? a.N * 4 : 0);
  [[maybe_unused]] const v4i srdS = words(F8 ? a.col_scale : reinterpret_cast<const float*>(a.W), F8 ? a.N * 4 : 0);
  // dynamic hand-out: this label's workgroups (cnt_x of them, at positions off_x .. of every round) share counter xcd
  const int cnt_x = q8 + (xcd < r8 ? 1 : 0), off_x = pos - (bid >> 3);
  const v4i srdT = words(reinterpret_cast<const float*>(a.sched ? a.sched : reinterpret_cast<const uint32_t*>(a.W)), a.sched ? 8 * 128 : 0);

  // ---- LDS-DMA staging: wave w fills LDS rows [16w, 16w+16) of a unit in two 8-row pieces (1 KiB each) -------------
  // vA[ha][q] / vW[hb]: per-lane byte offsets of the pieces, for the tile whose units of that kind are being REQUESTED.
  // Rebuilt from an opaque copy of the lane id (nothing lane-dependent stays live across the tile loop for them).
  const uint32_t lda2 = (uint32_t)(a.lda * ESZ), ldw2 = (uint32_t)(a.ldw * ESZ);  // row pitches in bytes (ldw2 % 128 == 0)
  const uint32_t a_last = (uint32_t)(a.M - 1) * lda2;  // rows beyond M re-read the last valid row
  uint32_t vA[2][2], vW[2];
  auto set_a = [&](const Tile& t, int ha) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t pr = (uint32_t)(l >> 3), pp = (uint32_t)(l & 7);
    const uint32_t ch0 = (pp ^ (pr >> 1)) << 4;  // source chunk of LDS position pp in row 8q + pr: pp ^ ((4q + (pr >> 1)) & 7)
    const uint32_t row0 = ((uint32_t)t.m0 + (uint32_t)(WROWS * (wave >> 2) + 64 * ha + 16 * (wave & 3))) * lda2 + pr * lda2;
    vA[ha][0] = min(row0, a_last) + ch0;
    vA[ha][1] = min(row0 + 8u * lda2, a_last) + (ch0 ^ 64u);
  };
  auto set_w = [&](const Tile& t, int hb) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t pr = (uint32_t)(l >> 3), pp = (uint32_t)(l & 7);
    const uint32_t ch0 = (pp ^ (pr >> 1)) << 4;
    vW[hb] = ((uint32_t)t.n0 + (uint32_t)(64 * (wave >> 1) + 32 * hb + 16 * (wave & 1))) * ldw2 + pr * ldw2 + ch0;
  };
  // LDS ring: unit buffers [A0 s0][A0 s1][A1 s0][A1 s1][W0 s0][W0 s1][W1 s0][W1 s1] (s = K tile & 1): every fragment read
  // of an operand is one base register + an immediate < 64 KB.  kind 0: W0, 1: A0, 2: W1, 3: A1 (the order in which a
  // K tile's units are read)
  auto issue = [&](int kind, int kr) {
    unsigned char* d = smem + ((kind & 1) ? 0 : 4 * UNIT) + (2 * (kind >> 1) + (kr & 1)) * UNIT + wave * 16 * ROWB;
    if (kind & 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)d, 16, vA[kind >> 1][0], kr * ROWB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(d + 8 * ROWB), 16, vA[kind >> 1][1], kr * ROWB, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdW, (lds_ptr_t)d, 16, vW[kind >> 1], kr * ROWB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdW, (lds_ptr_t)(d + 8 * ROWB), 16, vW[kind >> 1] ^ 64u, kr * ROWB + 8 * (int)ldw2, 0, 0);
    }
  };

  // ---- fragment reads: lane (fr, fq) reads row fr of a 16-row block, 16-byte chunk 4*ks + fq ----------------
  const unsigned char* rdA[2];
  const unsigned char* rdW[2];
  {
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = (fr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cs = ((4 * ks + fq) ^ sw) << 4;
      rdA[ks] = smem + (wr * 64 + fr) * ROWB + cs;
      rdW[ks] = smem + 4 * UNIT + (wc * 32 + fr) * ROWB + cs;
    }
  }

  // operands of a quadrant: 4 row blocks of A, 2 column blocks of W, both 16-byte chunks of the K tile each (two sets of
  // W: W0 of the next K tile is read while W0 of this one is still needed).  bf16: a chunk is the operand of one
  // 16x16x32 MFMA; fp8: the two chunks of a row are ONE 32-byte operand (k-set {16 fq ..} U {64 + 16 fq ..}, the same on
  // both sides) and sit in eight consecutive registers.
  struct Opnd {
    v8i v;
  };
  Opnd Aa[4], X[2], Y[2];
  f32x4 acc[RB][4];
  auto read_op = [&](Opnd& o, const unsigned char* p0, const unsigned char* p1) {
    const v4i x0 = *reinterpret_cast<const v4i*>(p0), x1 = *reinterpret_cast<const v4i*>(p1);
    o.v = v8i{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  };
  auto read_a = [&](int slot, int ha) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ha == 0 || i < HB) read_op(Aa[i], rdA[0] + (2 * ha + slot) * UNIT + i * 16 * ROWB, rdA[1] + (2 * ha + slot) * UNIT + i * 16 * ROWB);
  };
  auto read_w = [&](Opnd (&w)[2], int slot, int hb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) read_op(w[j], rdW[0] + (2 * hb + slot) * UNIT + j * 16 * ROWB, rdW[1] + (2 * hb + slot) * UNIT + j * 16 * ROWB);
  };
  auto chunk = [](const Opnd& o, int ks) {
    return __builtin_bit_cast(bf16x8, ks == 0 ? __builtin_shufflevector(o.v, o.v, 0, 1, 2, 3) : __builtin_shufflevector(o.v, o.v, 4, 5, 6, 7));
  };
  auto quadrant = [&](const Opnd (&w)[2], int ha, int hb) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * ha + i][2 * hb + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w[j].v, Aa[i].v, acc[4 * ha + i][2 * hb + j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                                          0x7f7f7f7f);  // e4m3 x e4m3, block scales 2^0
      // (tie the group to its place: the compiler otherwise sinks these MFMAs across the barriers towards their consumers,
      // which makes every operand of several phases live at once)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[4 * ha + i][2 * hb + j]));
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < (ha == 0 ? 4 : HB); ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[4 * ha + i][2 * hb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(chunk(w[j], ks), chunk(Aa[i], ks), acc[4 * ha + i][2 * hb + j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = a.K / (F8 ? 128 : TK);  // even, >= 6
  int idx = pos;                         // < ntiles: the launcher keeps G <= ntiles
  Tile cur = decode_tile(idx, tiles_m, tiles_n, TMU);
  set_a(cur, 0);
  set_a(cur, 1);
  set_w(cur, 0);
  set_w(cur, 1);
  // prologue: units 0 .. DEPTH of the first tile (unit v: K tile v >> 2, kind v & 3)
#pragma unroll
  for (int v = 0; v <= DEPTH; ++v) issue(v & 3, v >> 2);
  wait_vm<NB>();  // units 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on
  __builtin_amdgcn_sched_barrier(0);
  int s_prev = 0;  // REAL stores of the previous epilogue that may still be in flight when this tile's loop starts

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

  const int nk_ = a.K / (F8 ? 128 : TK);
  // (the short-K form needs the next tile before a draw could return; the e4m3 forms have no register left for the draw)
  const bool dyn = !F8 && a.sched != nullptr && nk_ >= 6;
  const bool drawer = dyn && wave == 0;
  for (;;) {
    // the tile after this one: known at once when tiles are dealt statically; drawn from the label's counter otherwise
    // (requested here by wave 0, published through LDS behind the K loop's first plain wait, read by every wave after the
    // second K tile: `draw_next` below)
    int nidx = idx + G;
    bool has_next = nidx < ntiles;
    Tile nxt = has_next ? decode_tile(nidx, tiles_m, tiles_n, TMU) : cur;
    uint32_t drawn = 1;
    if (drawer) {
      // ONE lane adds: EXEC is narrowed to lane 0 around the instruction.  (Steering the other lanes out of range with an
      // offset of ~0, as the masked stores do, is not an option here: an atomic with such an offset faulted — 'memory
      // aperture violation' — instead of being dropped.)
      const uint32_t coff = (uint32_t)(xcd * 128);
      uint64_t saved_exec;
      asm volatile(
          "s_mov_b64 %1, exec\n\t"
          "s_mov_b64 exec, 1\n\t"
          "s_nop 4\n\t"
          "buffer_atomic_add %0, %2, %3, 0 offen sc0\n\t"
          "s_mov_b64 exec, %1"
          : "+v"(drawn), "=&s"(saved_exec)
          : "v"(coff), "s"(srdT)
          : "memory");
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // W0 of the tile's first K tile (unit 0: retired by the wait of the previous tile's last-but-one segment, or by the
    // prologue's).  Read here rather than in the previous tile's last segment, so that no operand is live across the
    // epilogue; retired at once, so that the other group's request for this buffer's next unit (two barriers on) finds
    // it read.
    read_w(X, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // Column vectors of this wave's 64 columns (bias; fp8: + dequantisation scales): ONE float per lane, fetched two K
    // tiles before the epilogue, parked in the wave's (idle) 4 KB of epilogue staging once landed, and read back as
    // per-fragment vectors at the top of the epilogue: 1-2 registers across the loop's tail instead of 16-32.
    float bv;
    [[maybe_unused]] float sv;
    auto load_col_vectors = [&] {
      int lb = lane;
      asm volatile("" : "+v"(lb));
      const uint32_t coff = (uint32_t)((cur.n0 + wc * 64 + lb) * 4);
      const uint32_t boff = a.bias ? coff : 0xffffffffu;  // no bias: out of range reads 0
      // (s_nop 4: the descriptor may have been restored from a spill lane by v_readlane just before; a VALU-written SGPR
      // needs five wait states before a buffer instruction reads it, and nothing pads the inside of an asm statement)
      asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=&v"(bv) : "v"(boff), "s"(srdB) : "memory");
      if constexpr (F8) asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=&v"(sv) : "v"(coff), "s"(srdS) : "memory");
    };
    auto park_col_vectors = [&] {  // behind the wait that covers them (the compiler does not know they were ever in flight)
      int lb = lane;
      asm volatile("" : "+v"(lb), "+v"(bv));
      *reinterpret_cast<float*>(ep + lb * 4) = bv;
      if constexpr (F8) {
        asm volatile("" : "+v"(sv));
        *reinterpret_cast<float*>(ep + 256 + lb * 4) = sv;
      }
    };

    // One K tile.  PAR = kt & 1 (ring slot; which register set holds W0: 0 X, 1 Y).  HEAD = kt for the tile's first two
    // K tiles, else 2.  END = nk - 1 - kt for the last four, else 4.  Everything below that depends on the position in
    // the tile is decided at compile time from those.
    // L segment p: [reads] | request unit 4 kt + p + DEPTH + 1 | wait for unit 4 kt + p + 2 | barrier.  Operations not
    // counted in NB that are younger than the awaited unit: the previous epilogue's stores while 4 kt + p <= DEPTH - 2,
    // the column-vector loads (issued ahead of K tile nk - 2) for DEPTH - 1 segments from there.
    auto ktile = [&](int kt, auto par_c, auto head_c, auto end_c) {
      constexpr int PAR = decltype(par_c)::value, HEAD = decltype(head_c)::value, END = decltype(end_c)::value;
      auto seg_tail = [&](auto p_c) {
        constexpr int p = decltype(p_c)::value;
        constexpr int c = p + DEPTH + 1, kind = c & 3, dk = c >> 2;  // unit 4 kt + c: K tile kt + dk
        constexpr bool wraps = END < dk;                           // ... which is the next tile's K tile dk - END - 1
        const int kr = wraps ? dk - END - 1 : kt + dk;
        issue(kind, kr);
        if constexpr (END == dk) {  // that was the kind's last unit of this tile: its offsets move on (no next tile: harmless re-reads)
          if (kind == 0) set_w(nxt, 0);
          else if (kind == 1) set_a(nxt, 0);
          else if (kind == 2) set_w(nxt, 1);
          else set_a(nxt, 1);
        }
        constexpr bool stores_young = HEAD < 2 && 4 * HEAD + p <= DEPTH - 2;
        constexpr bool cv_young = END <= 1 && 4 * (1 - END) + p <= DEPTH - 2;
        if constexpr (stores_young) {  // (never together with the column-vector window: nk >= 6)
          // ... and wave 0's draw of the next tile, issued at the top of the tile: as young as the stores
          if (drawer) {
            if (s_prev == 0) wait_vm<NB + 1>();
            else if (EPI != DFD_EPI_QKV_EXPORT || s_prev == S1) wait_vm<NB + S1 + 1>();
            else wait_vm<NB + S2 + 1>();
          } else {
            if (s_prev == 0) wait_vm<NB>();
            else if (EPI != DFD_EPI_QKV_EXPORT || s_prev == S1) wait_vm<NB + S1>();
            else wait_vm<NB + S2>();
          }
        } else {
          wait_vm<NB + (cv_young ? NCV : 0)>();
        }
        if constexpr (HEAD < 2 && 4 * HEAD + p == DEPTH - 1) {
          // first wait that no longer skips the draw: it has returned.  Publish it (wave 0's staging, beyond the parked
          // column vectors); the other waves read it several barriers later
          // (explicit DS instructions on the LDS byte address: a volatile access through the generic pointer makes the
          // compiler fall back to FLAT instructions for this and every other staging access — those count in vmcnt too)
          if (drawer) {
            const uint32_t slot_addr = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + RING + 1024);
            asm volatile("" : "+v"(drawn));  // (the compiler believes it was written when the atomic was issued)
            const uint32_t lane0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);  // only lane 0 drew; every lane stores its value
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(slot_addr), "v"(lane0) : "memory");
          }
        }
        if constexpr (END <= 1 && 4 * (1 - END) + p == DEPTH - 1) {
          park_col_vectors();  // the column vectors have landed
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      using P0 = std::integral_constant<int, 0>;
      using P1 = std::integral_constant<int, 1>;
      using P2 = std::integral_constant<int, 2>;
      using P3 = std::integral_constant<int, 3>;
      if constexpr (END == 1) load_col_vectors();
      read_a(PAR, 0);
      seg_tail(P0{});
      quadrant(PAR ? Y : X, 0, 0);
      read_w(PAR ? X : Y, PAR, 1);
      seg_tail(P1{});
      quadrant(PAR ? X : Y, 0, 1);
      read_a(PAR, 1);
      seg_tail(P2{});
      quadrant(PAR ? X : Y, 1, 1);
      if constexpr (END != 0) read_w(PAR ? X : Y, PAR ^ 1, 0);  // W0 of the next K tile
      seg_tail(P3{});
      quadrant(PAR ? Y : X, 1, 0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using C4 = std::integral_constant<int, 4>;
    bool short_k = false;
    if constexpr (EPI == DFD_EPI_RESIDUAL_POS) short_k = nk == 4;  // the adapter's x -> D Linear (x = 256): head and tail overlap
    if (short_k) {
      if constexpr (EPI == DFD_EPI_RESIDUAL_POS) {
        ktile(0, C0{}, C0{}, C3{});
        ktile(1, C1{}, C1{}, C2{});
        ktile(2, C0{}, C2{}, C1{});
        ktile(3, C1{}, C2{}, C0{});
      }
    } else {
      ktile(0, C0{}, C0{}, C4{});
      ktile(1, C1{}, C1{}, C4{});
      if (dyn) {  // draw_next: the n-th tile drawn on this label is position n % cnt_x of its round 1 + n / cnt_x
        uint32_t n;
        const uint32_t slot_addr = (uint32_t)(uintptr_t)(lds_ptr_t)(smem + RING + 1024);
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(n) : "v"(slot_addr) : "memory");
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n) - a.sched_base[xcd];
        const uint32_t r = n / (uint32_t)cnt_x;
        has_next = r < (uint32_t)(ntiles / G + 1);  // (also keeps a counter that ran away from overflowing the index)
        nidx = (int)((r + 1) * (uint32_t)G + (uint32_t)off_x + (n - r * (uint32_t)cnt_x));
        has_next = has_next && nidx < ntiles;
        nxt = has_next ? decode_tile(nidx, tiles_m, tiles_n, TMU) : cur;
      }
      for (int kt = 2; kt < nk - 4; kt += 2) {
        ktile(kt, C0{}, C2{}, C4{});
        ktile(kt + 1, C1{}, C2{}, C4{});
      }
      ktile(nk - 4, C0{}, C2{}, C3{});
      ktile(nk - 3, C1{}, C2{}, C2{});
      ktile(nk - 2, C0{}, C2{}, C1{});
      ktile(nk - 1, C1{}, C2{}, C0{});
    }

    // ---- epilogue (gemm256p.hip's: bias, activation, LDS-staged whole-line stores, left in flight) -------------------
    // Every address below is rebuilt from an opaque copy of the lane id (left to itself the compiler hoists two dozen
    // tile-invariant address registers out of the tile loop and spills them).
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
      f32x4 b4[4];
      [[maybe_unused]] f32x4 cs4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b4[j] = *reinterpret_cast<const f32x4*>(ep + (j * 16 + eq * 4) * 4);
        if constexpr (F8) cs4[j] = *reinterpret_cast<const f32x4*>(ep + 256 + (j * 16 + eq * 4) * 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // before the staging below overwrites them
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
              // no embedding (the raw export an adapter reads): add zeros, and issue no load — a register load beside
              // LDS-DMA costs a vmcnt(0) at its use, i.e. one memory round trip per sub-pass
              pe[rr][0] = pe[rr][1] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (a.pos != nullptr) {
                const uint32_t poff = (t * (uint32_t)D + ecol) * 4;
                pe[rr][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdP, poff, 0, 0));
                pe[rr][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdP, poff, 16, 0));
              }
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
      if constexpr (EPI == DFD_EPI_RESIDUAL_POS) {
        // C(bf16) = residual + dropout(acc) + pos[(row / rows_per_frame) % T], rounded ONCE (the adapter's second Linear,
        // reference models.py:795-875, :930-940; gemm256.hip has the one-workgroup-per-tile form of the same arithmetic).
        // RB sub-passes of 16 rows parked as f32 (4 KB); the drain reads the residual row segment and the positional
        // embedding (requested at the top of the sub-pass), adds and rounds; 2 stores of 8 rows x 128 B per sub-pass.
        unsigned char* const parkf = ep + er * 256;  // unit (j*4 + eq) ^ er of a 256-byte row
        const uint32_t cbase = (uint32_t)((mrow0 * a.ldc + nb + dc * 8) * 2);
        // The residual and the embedding of sub-pass i + 1 are requested before sub-pass i is drained (one sub-pass of
        // loads always in flight: with each load waited for where it is issued the epilogue is a chain of 2 RB memory
        // round trips, 29 us per tile).  Inline-asm loads, counted by hand like the K loop's: queue at the wait of sub-pass
        // i = [loads i][2 stores of i-1][6 loads of i+1].
        const v4i srdRw = a.residual ? words(reinterpret_cast<const float*>(a.residual), (int)(uint32_t)(a.M * a.ldc * 2))
                                     : words(reinterpret_cast<const float*>(a.C), (int)(uint32_t)(a.M * a.ldc * 2));
        const v4i srdQw = words(a.pos ? a.pos : reinterpret_cast<const float*>(a.W), a.pos ? a.frames_per_clip * a.N * 4 : 0);
        f32x4 pq[2][2][2];
        v4i oq[2][2];
        uint32_t offq[2][2], gq[2][2];
        auto request = [&](int i, int b) {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int rloc = i * 16 + rr * 8;
            const uint32_t m = (uint32_t)min(mrow0 + rloc, a.M - 1);
            const uint32_t frame = a.div_tokens.div(m);  // rows per frame = tokens - 1 (the export has no CLS row)
            const uint32_t t = frame - a.div_frames.div(frame) * (uint32_t)a.frames_per_clip;
            offq[b][rr] = rloc < rows_left ? cbase + (uint32_t)rloc * (uint32_t)(a.ldc * 2) : 0xffffffffu;
            gq[b][rr] = m * (uint32_t)(a.N >> 3) + (uint32_t)((nb + dc * 8) >> 3);  // dropout group (the launcher checks M * N / 8 < 2^32)
            const uint32_t poff = a.pos ? (t * (uint32_t)a.N + (uint32_t)(nb + dc * 8)) * 4 : 0xffffffffu;  // no embedding: out of range reads 0
            asm volatile(
                "s_nop 4\n\t"
                "buffer_load_dwordx4 %0, %3, %5, 0 offen\n\t"
                "buffer_load_dwordx4 %1, %3, %5, 0 offen offset:16\n\t"
                "buffer_load_dwordx4 %2, %4, %6, 0 offen"
                : "=&v"(pq[b][rr][0]), "=&v"(pq[b][rr][1]), "=&v"(oq[b][rr])
                : "v"(poff), "v"(offq[b][rr]), "s"(srdQw), "s"(srdRw)
                : "memory");
          }
        };
        request(0, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          const int b = i & 1;
          if (i + 1 < RB) request(i + 1, b ^ 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(parkf + (((j * 4 + eq) ^ er) << 4)) = acc[i][j];
          if (i == 0) wait_vm<6>();
          else if (i + 1 < RB) wait_vm<8>();
          else wait_vm<2>();
          asm volatile("" : "+v"(pq[b][0][0]), "+v"(pq[b][0][1]), "+v"(oq[b][0]), "+v"(pq[b][1][0]), "+v"(pq[b][1][1]), "+v"(oq[b][1]));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int row = rr * 8 + drow;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * dc) ^ row) << 4));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * 256 + (((2 * dc + 1) ^ row) << 4));
            float dv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              dv[e] = x0[e];
              dv[4 + e] = x1[e];
            }
            // the adapter's last nn.Dropout, before the residual add: element index row * N + column (a multiple of 8)
            dfd_drop_eight(a.drop, (uint64_t)gq[b][rr] << 3, dv);
            const bf16x8 ob = __builtin_bit_cast(bf16x8, oq[b][rr]);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] = (bf16_t)((float)ob[e] + dv[e] + pq[b][rr][0][e]);
              o[4 + e] = (bf16_t)((float)ob[4 + e] + dv[4 + e] + pq[b][rr][1][e]);
            }
            store_out(__builtin_bit_cast(v4i, o), srdC, offq[b][rr], a.stream_out);
          }
        }
      } else if constexpr (CF8) {
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
            // (the XOR is done on the LDS byte address and cast back to an LDS pointer: through a generic pointer the
            // compiler loses the address space and emits flat_store, which also counts in vmcnt)
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
    // a wave's stores are all real only if all of its rows are inside M (see the header); else count none
    s_prev = (int64_t)cur.m0 + wr * WROWS + WROWS <= a.M ? stores : 0;
    idx = nidx;
    cur = nxt;
  }
  wait_vm<0>();                                // units requested past the end must not land in LDS that is no longer ours
  if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the second group's extra barrier
}

// ---- host side of the dynamic hand-out -------------------------------------------------------------------------------
// One set of eight counters per stream (launches of one stream run one after the other, so they can share a set; launches
// of different streams may overlap and must not).  The counters only ever grow: a launch is told the value each will hold
// when it starts, which the host knows because a launch with grid G over `ntiles` tiles adds exactly (its workgroups'
// first tiles) + (the tiles drawn) = one draw per tile processed to the label's counter.
struct SchedSlot {
  uint32_t* dev = nullptr;
  uint32_t base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

bool sched_prepare(GemmArgs& a, hipStream_t st, int grid, int64_t ntiles) {
  static std::mutex mu;
  static std::unordered_map<hipStream_t, SchedSlot> slots;
  static uint32_t* pool = nullptr;
  static int used = 0;
  constexpr int MAX_SLOTS = 64;
  a.sched = nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return true;  // a captured launch is replayed with the same arguments: static order
  }
  std::lock_guard<std::mutex> lock(mu);
  auto it = slots.find(st);
  if (it == slots.end()) {
    if (pool == nullptr) {
      if (hipMalloc(reinterpret_cast<void**>(&pool), MAX_SLOTS * 8 * 128) != hipSuccess || hipMemset(pool, 0, MAX_SLOTS * 8 * 128) != hipSuccess) {
        (void)hipGetLastError();
        pool = nullptr;
        return true;  // no counters: static order
      }
    }
    if (used >= MAX_SLOTS) return true;
    SchedSlot sl;
    sl.dev = pool + used * 8 * 32;
    ++used;
    it = slots.emplace(st, sl).first;
  }
  SchedSlot& sl = it->second;
  a.sched = sl.dev;
  const int q8 = grid >> 3, r8 = grid & 7;
  const int64_t full = ntiles / grid, rem = ntiles % grid;  // full >= 1: the launcher keeps grid <= ntiles
  for (int x = 0; x < 8; ++x) {
    a.sched_base[x] = sl.base[x];
    const int cnt = q8 + (x < r8 ? 1 : 0);
    const int off = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8;
    int64_t last = rem - off;  // tiles of the partly filled last round that fall to this label
    last = last < 0 ? 0 : (last > cnt ? cnt : last);
    sl.base[x] += (uint32_t)(full * cnt + last);  // = first tiles (cnt) + drawn tiles ((full - 1) cnt + last): one draw per tile processed
  }
  return true;
}

template <int EPI, bool F8, bool CF8>
int launch256e(const GemmArgs& a_in, hipStream_t st) {
  GemmArgs a = a_in;
  const int tiles_n = a.N / TN;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dfd_set_error("dfd_gemm(ping-pong): cannot query the device");
      return DFD_ERR_LAUNCH;
    }
    n_cu = prop.multiProcessorCount;
  }
  int cus = n_cu - a.spare_cus;
  cus = cus < n_cu / 2 ? n_cu / 2 : cus;
  if (a.spare_cus > 0 && a.spare_if_free) {
    // a request, not an order: honoured where the rounds of tiles (at the better of the two tile heights) stay the same
    auto cost = [&](int c) {
      auto r = [&](int rows) { return (double)((((a.M + rows - 1) / rows) * tiles_n + c - 1) / c); };
      const double r256 = r(256), r224 = r(224) * 0.97;
      return (F8 || EPI == DFD_EPI_RESIDUAL_POS) ? (EPI == DFD_EPI_RESIDUAL_POS ? r224 : r256) : (r224 < r256 ? r224 : r256);
    };
    if (cost(cus) > cost(n_cu)) cus = n_cu;
  }
  // tile height: the one with the least (rounds of tiles) x (cost of a tile).  A 224-row tile saves the MFMA and
  // epilogue work of 32 rows but stages as many bytes as a 256-row one, and the loop is bound by that staging:
  // measured on the four ViT-B/16 shapes it costs 0.97 of a full tile, so it wins only where it saves a whole
  // round (M = 94,560: N = 768 needs 5 rounds either way -> 224; N = 2304 / 3072: 15 vs 14, 20 vs 18 -> 256)
  auto rounds = [&](int rows) {
    const int64_t tiles = ((a.M + rows - 1) / rows) * tiles_n;
    return (double)((tiles + cus - 1) / cus);
  };
  // (RESIDUAL_POS: 224-row tiles only — with 128 accumulator registers its read-modify-write epilogue, which also draws
  // the dropout mask, does not fit the register file without spilling, and a spill would break the counted waits)
  const bool use224 = !F8 && (EPI == DFD_EPI_RESIDUAL_POS || a.tile_rows == 224 || (a.tile_rows == 0 && rounds(224) * 0.97 < rounds(256)));
  const int rows = use224 ? 224 : 256;
  const int tiles_m = (int)((a.M + rows - 1) / rows);
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  const int grid = (int)(ntiles < cus ? ntiles : cus);
  const int nk = a.K / (F8 ? 128 : TK);
  if (!F8 && nk >= 6 && a.no_dynamic == 0) sched_prepare(a, st, grid, ntiles);
  else a.sched = nullptr;
  if constexpr (F8) {
    hipLaunchKernelGGL((gemm256e_kernel<EPI, 8, true, CF8>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
  } else {
    if (use224) hipLaunchKernelGGL((gemm256e_kernel<EPI, 7, false, false>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
    else if constexpr (EPI != DFD_EPI_RESIDUAL_POS) hipLaunchKernelGGL((gemm256e_kernel<EPI, 8, false, false>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_gemm(ping-pong): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}

// shared eligibility: 0 = fine, 1 = not served
int check256e(const GemmArgs& a, int esz, int csz, int kstep, bool short_k_ok = false) {
  const int nk = a.K / kstep;  // the loop is unrolled in pairs of K tiles around a head of two and a tail of four
  if (a.N % TN != 0 || a.K % kstep != 0 || (nk < 6 && !(short_k_ok && nk == 4)) || (nk & 1) || a.M < 1024) return 1;
  if ((a.ldw * esz) % 128 != 0) return 1;  // the second piece of a W unit is addressed as (first ^ 64) + 8 rows
  if ((a.lda * esz) % 16 != 0 || (a.ldw * esz) % 16 != 0 || (a.ldc * csz) % 16 != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.W) & 15) != 0 || (reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return 1;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) return 1;
  const int64_t lim = (int64_t)0xfffffff0;  // buffer descriptors carry 32-bit byte offsets
  if (a.M * a.lda * esz > lim || (int64_t)a.N * a.ldw * esz > lim || a.M * a.ldc * csz > lim) return 1;
  if ((int64_t)((a.M + 223) / 224) * (a.N / TN) > 0x3fffffff || a.M >= ((int64_t)1 << 31)) return 1;
  return 0;
}

int check_export_e(const GemmArgs& a) {
  if ((a.N / (3 - a.qkv_first)) % TN != 0) return 1;
  if (a.pos && (reinterpret_cast<uintptr_t>(a.pos) & 15) != 0) return 1;
  if (a.k_export && (a.M / a.tokens) * (a.tokens - 1) * (int64_t)(a.N / (3 - a.qkv_first)) * 2 > (int64_t)0xfffffff0) return 1;
  return 0;
}

}  // namespace

// 0 = launched, <0 = error, 1 = shape / epilogue not served by this kernel
int dfd_gemm256e_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st) {
  if (c_dtype != DFD_BF16 || check256e(a, 2, 2, 64, epi == DFD_EPI_RESIDUAL_POS)) return 1;
  switch (epi) {
    case DFD_EPI_BIAS:
      return launch256e<DFD_EPI_BIAS, false, false>(a, st);
    case DFD_EPI_BIAS_QUICKGELU:
      return launch256e<DFD_EPI_BIAS_QUICKGELU, false, false>(a, st);
    case DFD_EPI_QKV_EXPORT: {
      if (check_export_e(a)) return 1;
      GemmArgs b = a;
      b.div_tokens = FastDiv::make((uint32_t)a.tokens);
      b.div_frames = FastDiv::make((uint32_t)a.frames_per_clip);
      return launch256e<DFD_EPI_QKV_EXPORT, false, false>(b, st);
    }
    case DFD_EPI_RESIDUAL_POS: {
      if (check256e(a, 2, 2, 64, true) || a.tokens < 2 || a.M * (int64_t)(a.N >> 3) >= ((int64_t)1 << 32)) return 1;
      if (a.pos && ((reinterpret_cast<uintptr_t>(a.pos) & 15) != 0 || a.frames_per_clip < 1)) return 1;
      if (a.residual && (reinterpret_cast<uintptr_t>(a.residual) & 15) != 0) return 1;
      GemmArgs b = a;
      b.div_tokens = FastDiv::make((uint32_t)(a.tokens - 1));  // rows per frame
      b.div_frames = FastDiv::make((uint32_t)(a.frames_per_clip > 0 ? a.frames_per_clip : 1));
      return launch256e<DFD_EPI_RESIDUAL_POS, false, false>(b, st);
    }
    default:
      return 1;
  }
}

// fp8 (e4m3) operands on the block-scaled matrix cores; C bf16, or e4m3 for the plain / QuickGELU epilogues
int dfd_gemm256e_f8_try(const GemmArgs& a, int c_dtype, int epi, hipStream_t st) {
  if (c_dtype != DFD_BF16 && c_dtype != DFD_FP8) return 1;
  if (!a.col_scale || (reinterpret_cast<uintptr_t>(a.col_scale) & 15) != 0) return 1;
  if (check256e(a, 1, c_dtype == DFD_FP8 ? 1 : 2, 128)) return 1;
  switch (epi) {
    case DFD_EPI_BIAS:
      return c_dtype == DFD_FP8 ? launch256e<DFD_EPI_BIAS, true, true>(a, st) : launch256e<DFD_EPI_BIAS, true, false>(a, st);
    case DFD_EPI_BIAS_QUICKGELU:
      return c_dtype == DFD_FP8 ? launch256e<DFD_EPI_BIAS_QUICKGELU, true, true>(a, st) : launch256e<DFD_EPI_BIAS_QUICKGELU, true, false>(a, st);
    case DFD_EPI_QKV_EXPORT: {
      if (c_dtype != DFD_BF16 || check_export_e(a)) return 1;
      GemmArgs b = a;
      b.div_tokens = FastDiv::make((uint32_t)a.tokens);
      b.div_frames = FastDiv::make((uint32_t)a.frames_per_clip);
      return launch256e<DFD_EPI_QKV_EXPORT, true, false>(b, st);
    }
    default:
      return 1;
  }
}
