// Lab: persistent 256x256 bf16 GEMM with a PING-PONG K loop (candidate replacement for gemm256p.hip's K step).
//
// gemm256p runs the same software-pipelined instruction stream in all eight waves: the two waves of a SIMD reach their
// MFMA groups, their LDS read bursts and the one barrier per step together.  Here the two wave groups (wr = 0 / 1: one
// wave of each on every SIMD) run ONE BARRIER APART: while a group issues the 16 MFMAs of a quadrant of its 128x64
// output, the other group does its LDS fragment reads and requests the next half tile; then they swap.  Structure
// (cdna_hip_programming.md "The 256^2 8-phase template"): per 64-deep K tile four phases, each
//     L: ds_read the operands of the phase's quadrant | 2 LDS-DMA pieces (one 16 KB "unit") | counted vmcnt | barrier
//     M: lgkmcnt(0) | 16 x v_mfma_f32_16x16x32_bf16 | barrier
// A unit = one half (128 rows) of A or of W for one K tile.  The halves are INTERLEAVED so that a wave's output stays
// one contiguous 128x64 block: LDS row (64 wr' + r) of A-half ha is tile row 128 wr' + 64 ha + r, LDS row (32 wc' + c)
// of W-half hb is tile column 64 wc' + 32 hb + c.  Units are requested in the order they are read — W0, A0, W1, A1 of
// K tile 0, W0, A0, ... — one per L segment, DEPTH segments ahead of the segment that reads them; the wait at the end of
// an L segment retires the unit the NEXT L segment reads (so the other group's pieces are covered by the barriers in
// between) and leaves 2 (DEPTH - 1) younger requests in flight.  The sequence is flat across the tiles a workgroup walks.
#include "../../dfd-clip_amd/csrc/gemm256p_common.hpp"

namespace {

constexpr int UNIT = 128 * ROWB;  // 16 KB; a ring slot = [A0][A1][W0][W1]

template <int EPI, int RB, int DEPTH, int VAR>
__global__ __launch_bounds__(512) void gemm256e_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  static_assert(DEPTH >= 4 && DEPTH <= 6, "a unit's buffer is free again 8 segments after the unit before it was read");
  constexpr int TMU = 32 * RB;    // rows a tile uses
  constexpr int WROWS = 16 * RB;  // rows per wave
  constexpr int HB = RB - 4;      // row blocks in the second half
  constexpr int NB = 2 * (DEPTH - 1) - ((VAR & 1) ? 2 : 0);  // requests younger than the unit a segment waits for (VAR bit 0: wait one unit further)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + 8 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = tiles_m * tiles_n;

  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)(uint32_t)(a.M * a.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(uint32_t)((int64_t)a.N * a.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(uint32_t)(a.M * a.ldc * 2), 0x00020000);
  // the bias descriptor as four plain words: its loads are inline asm (the compiler waits vmcnt(0) for any ordinary
  // register load issued beside LDS-DMA, which would drain the ring once per tile; these it does not see)
  v4i srdB;
  {
    const uint64_t bp = reinterpret_cast<uint64_t>(a.bias ? a.bias : reinterpret_cast<const float*>(a.W));
    srdB = v4i{(int)(uint32_t)bp, (int)(uint32_t)(bp >> 32) & 0xffff, a.bias ? a.N * 4 : 0, 0x00020000};
  }

  // ---- LDS-DMA staging: wave w fills LDS rows [16w, 16w+16) of a unit in two 8-row pieces (1 KiB each) -------------
  // vA[ha][q] / vW[hb]: per-lane byte offsets of the pieces, for the tile whose units of that kind are being REQUESTED.
  // Rebuilt from an opaque copy of the lane id (nothing lane-dependent stays live across the tile loop for them).
  const uint32_t lda2 = (uint32_t)(a.lda * 2), ldw2 = (uint32_t)(a.ldw * 2);
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
  // of an operand is one base register + an immediate < 64 KB
  // kind 0: W0, 1: A0, 2: W1, 3: A1 (the order in which a K tile's units are read)
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

  // ---- fragment reads: lane (fr, fq) reads row fr of a 16-row block, chunk 4*ks + fq ----------------
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

  bf16x8 Aa[4][2], X[2][2], Y[2][2];
  f32x4 acc[RB][4];
  auto read_a = [&](int slot, int ha) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ha == 0 || i < HB) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) Aa[i][ks] = *reinterpret_cast<const bf16x8*>(rdA[ks] + (2 * ha + slot) * UNIT + i * 16 * ROWB);
      }
  };
  auto read_w = [&](bf16x8 (&w)[2][2], int slot, int hb) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) w[j][ks] = *reinterpret_cast<const bf16x8*>(rdW[ks] + (2 * hb + slot) * UNIT + j * 16 * ROWB);
  };
  auto quadrant = [&](const bf16x8 (&w)[2][2], int ha, int hb) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < (ha == 0 ? 4 : HB); ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * ha + i][2 * hb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][ks], Aa[i][ks], acc[4 * ha + i][2 * hb + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  constexpr int STORES = 2 * RB;  // per wave and tile
  const int nk = a.K / TK;        // even, >= 6
  int idx = pos;
  Tile cur = decode_tile(idx, tiles_m, tiles_n, TMU);
  set_a(cur, 0);
  set_a(cur, 1);
  set_w(cur, 0);
  set_w(cur, 1);
  // prologue: units 0 .. DEPTH of the first tile (unit v: K tile v >> 2, kind v & 3).  (Padding the queue with dropped
  // out-of-range stores so that the first tile counts like every other does NOT work: they retire at once, ahead of
  // the older loads, and a counted wait that includes them waits for nothing.)
#pragma unroll
  for (int v = 0; v <= DEPTH; ++v) issue(v & 3, v >> 2);
  wait_vm<NB>();  // units 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  read_w(X, 0, 0);
  if ((VAR & 4) == 0 && wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind from here on
  __builtin_amdgcn_sched_barrier(0);

  [[maybe_unused]] unsigned char* const ep = smem + RING + wave * STAGE;
  bool first_tile = true;  // no epilogue stores in flight at the start of its loop

  for (;;) {
    const int nidx = idx + G;
    const bool has_next = nidx < ntiles;
    const Tile nxt = has_next ? decode_tile(nidx, tiles_m, tiles_n, TMU) : cur;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b4[4];  // bias of this wave's 64 columns
    auto load_col_vectors = [&] {
      int lb = lane;
      asm volatile("" : "+v"(lb));
      const uint32_t coff = (uint32_t)((cur.n0 + wc * 64 + (lb >> 4) * 4) * 4);
      const uint32_t boff = a.bias ? coff : 0xffffffffu;  // no bias: out of range reads 0
      asm volatile(
          "buffer_load_dwordx4 %0, %4, %5, 0 offen\n\t"
          "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:64\n\t"
          "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:128\n\t"
          "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:192"
          : "=&v"(b4[0]), "=&v"(b4[1]), "=&v"(b4[2]), "=&v"(b4[3])
          : "v"(boff), "s"(srdB)
          : "memory");
    };

    // One K tile.  PAR = kt & 1 (ring slot; which register set holds W0: 0 X, 1 Y).  HEAD = kt for the tile's first two
    // K tiles, else 2.  END = nk - 1 - kt for the last four, else 4.  Everything below that depends on the position in
    // the tile is decided at compile time from those.
    // L segment p: [reads] | request unit 4 kt + p + DEPTH + 1 | wait for unit 4 kt + p + 2 | barrier.  Requests not
    // counted in NB that are younger than the awaited unit: the previous epilogue's stores while 4 kt + p <= DEPTH - 2,
    // the bias loads (issued ahead of K tile nk - 2) for DEPTH - 1 segments from there.
    auto ktile = [&](int kt, auto par_c, auto head_c, auto end_c) {
      constexpr int PAR = decltype(par_c)::value, HEAD = decltype(head_c)::value, END = decltype(end_c)::value;
      auto seg_tail = [&](auto p_c) {
        constexpr int p = decltype(p_c)::value;
        constexpr int c = p + DEPTH + 1, kind = c & 3, dk = c >> 2;  // unit 4 kt + c: K tile kt + dk
        constexpr bool wraps = END < dk;                           // ... which is the next tile's K tile dk - END - 1
        const int kr = wraps ? dk - END - 1 : kt + dk;
        if constexpr ((VAR & 2) != 0) {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        issue(kind, kr);
        if constexpr (END == dk) {  // that was the kind's last unit of this tile: its offsets move on (no next tile: harmless re-reads)
          if (kind == 0) set_w(nxt, 0);
          else if (kind == 1) set_a(nxt, 0);
          else if (kind == 2) set_w(nxt, 1);
          else set_a(nxt, 1);
        }
        constexpr bool stores_young = HEAD < 2 && 4 * HEAD + p <= DEPTH - 2;
        constexpr bool bias_young = END <= 1 && 4 * (1 - END) + p <= DEPTH - 2;
        if constexpr (stores_young) {  // (never together with the bias window: nk >= 6)
          if (first_tile || (VAR & 24) != 0) wait_vm<NB>();  // (ablations 8 / 16 issue no stores)
          else wait_vm<NB + STORES>();
        } else {
          wait_vm<NB + (bias_young ? 4 : 0)>();
        }
        if constexpr (END <= 1 && 4 * (1 - END) + p == DEPTH - 1) {
          // the bias has landed: consume it here, where the compiler's own wait for it is already satisfied (at its first
          // use in the epilogue it would be a vmcnt(0) behind the next tile's requests)
          asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
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
      read_w(PAR ? X : Y, PAR ^ 1, 0);  // W0 of the next K tile (the next tile's first at the end of this one)
      seg_tail(P3{});
      quadrant(PAR ? Y : X, 1, 0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using C4 = std::integral_constant<int, 4>;
    ktile(0, C0{}, C0{}, C4{});
    ktile(1, C1{}, C1{}, C4{});
    for (int kt = 2; kt < nk - 4; kt += 2) {
      ktile(kt, C0{}, C2{}, C4{});
      ktile(kt + 1, C1{}, C2{}, C4{});
    }
    ktile(nk - 4, C0{}, C2{}, C3{});
    ktile(nk - 3, C1{}, C2{}, C2{});
    ktile(nk - 2, C0{}, C2{}, C1{});
    ktile(nk - 1, C1{}, C2{}, C0{});

    if constexpr ((VAR & 16) != 0) {  // ablation: no epilogue at all (the accumulators are only kept alive)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
      asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
    } else {
    // ---- epilogue (as gemm256p.hip: bias, activation, LDS-staged whole-line stores left in flight) ----------------
    int le = lane;
    asm volatile("" : "+v"(le));
    const int er = le & 15, eq = le >> 4;
    const int drow = le >> 3, dc = le & 7;
    const int nb = cur.n0 + wc * 64;
    const int64_t mrow0 = (int64_t)cur.m0 + wr * WROWS + drow;
    const int rows_left = (int)min((int64_t)0x7fffffff, a.M - mrow0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += b4[j];
    auto activate = [&](f32x4 v) {
      if constexpr (EPI == DFD_EPI_BIAS_QUICKGELU) {
        float cgelu = DFD_QUICKGELU_SCALE;
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
    unsigned char* const park = ep + er * 128 + ((eq ^ ((er & 7) << 1)) << 3);
    const unsigned char* const dsrc = ep + drow * 128 + ((dc ^ drow) << 4);
    const uint32_t cbase = (uint32_t)((mrow0 * a.ldc + nb + dc * 8) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * q + ii;
        if (i >= RB) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = activate(acc[i][j]);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned char*>(reinterpret_cast<uintptr_t>(park + ii * 2048) ^ (uintptr_t)(j << 5))) = o;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (q * 32 + rr * 8 >= WROWS) continue;
        const v4i d = *reinterpret_cast<const v4i*>(dsrc + rr * 1024);
        const int rloc = q * 32 + rr * 8;
        uint32_t off = rloc < rows_left ? cbase + (uint32_t)rloc * (uint32_t)(a.ldc * 2) : 0xffffffffu;
        if constexpr ((VAR & 8) == 0) store_out(d, srdC, off, a.stream_out);  // ablation 8: everything but the stores
        else asm volatile("" ::"v"(d), "v"(off));
      }
    }
    }
    if (!has_next) break;
    first_tile = false;
    idx = nidx;
    cur = nxt;
  }
  wait_vm<0>();                                // the re-read units requested past the end must not land in LDS that is no longer ours
  if ((VAR & 4) == 0 && wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the second group's extra barrier
}

template <int EPI, int DEPTH, int VAR>
int launch256e(const GemmArgs& a, hipStream_t st) {
  const int tiles_n = a.N / TN;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DFD_ERR_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  int cus = n_cu - a.spare_cus;
  cus = cus < n_cu / 2 ? n_cu / 2 : cus;
  auto rounds = [&](int rows) {
    const int64_t tiles = ((a.M + rows - 1) / rows) * tiles_n;
    return (double)((tiles + cus - 1) / cus);
  };
  const bool use224 = a.tile_rows == 224 || (a.tile_rows == 0 && rounds(224) * 0.97 < rounds(256));
  const int rows = use224 ? 224 : 256;
  const int tiles_m = (int)((a.M + rows - 1) / rows);
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  const int grid = (int)(ntiles < cus ? ntiles : cus);
  if (use224) hipLaunchKernelGGL((gemm256e_kernel<EPI, 7, DEPTH, VAR>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
  else hipLaunchKernelGGL((gemm256e_kernel<EPI, 8, DEPTH, VAR>), dim3(grid), dim3(512), 0, st, a, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? DFD_OK : DFD_ERR_LAUNCH;
}

}  // namespace

// 0 = launched, <0 = error, 1 = shape not served.  depth: 4..6; + 10 x variant bits (1: wait a unit further, 2: keep the
// fragment reads ahead of the requests, 4: no stagger between the wave groups)
int dfd_gemm256e_launch(const GemmArgs& a, int epi, int depth, hipStream_t st) {
  const int nk = a.K / TK;
  if (a.N % TN != 0 || a.K % TK != 0 || nk < 6 || (nk & 1) || a.M < 1024 || (a.ldw * 2) % 128 != 0) return 1;
#define E_CASE(D, V)                                                                       \
  if (depth == (D) + 10 * (V)) {                                                            \
    if (epi == DFD_EPI_BIAS) return launch256e<DFD_EPI_BIAS, D, V>(a, st);                  \
    if (epi == DFD_EPI_BIAS_QUICKGELU) return launch256e<DFD_EPI_BIAS_QUICKGELU, D, V>(a, st); \
    return 1;                                                                               \
  }
  E_CASE(4, 0) E_CASE(5, 0) E_CASE(6, 0) E_CASE(6, 4) E_CASE(6, 8) E_CASE(6, 16)
#undef E_CASE
  return 1;
}
