// LAB ONLY (not part of libdfdclip_hip.so; built by tools/lab/build_lab.sh, timed by tools/lab/gemm_lab).  Result on
// MI355X (profiles/r02_gemm_four_wave_variant.txt): bit-identical to the product kernels, 5-10 % SLOWER than the
// eight-wave persistent kernel on all four ViT-B/16 shapes (c_fc 984-1024 vs 1086-1114 TFLOP/s): with one wave per
// SIMD every LDS / barrier wait and the whole epilogue idle the matrix pipe, which the second wave of the product
// kernel fills; the saved LDS reads do not pay for that.
//
// Four-wave form of the persistent bf16 GEMM (dfd-clip_amd/csrc/gemm256p.hip): the same 256x256 tile, 2 x 64 KB LDS ring, XOR swizzle,
// flat K-step sequence across tiles and stores-left-in-flight scheme, but ONE wave per SIMD (256 threads) owning a
// 128x128 quarter of the tile instead of two waves per SIMD owning 128x64 each.
//   * LDS fragment reads per 64-deep step drop from 192 KB to 128 KB (an A fragment is re-read by 2 column waves
//     instead of 4): with the 64 KB of LDS-DMA fills the step moves 192 KB through LDS instead of 256 KB;
//   * a wave has the SIMD's whole register file: 256 accumulator registers + two 64-register fragment sets, so the
//     fragments of a k-half are read a full half step (64 MFMAs) ahead of their use;
//   * half as many waves meet at the step's barrier.
// A step = two halves of 64 MFMAs (k-half 0 and 1): half h multiplies fragment set h; between its groups of 8 MFMAs
// it reads the other set (k-half 1 of this step, then k-half 0 of the next one, from the other slot); the second
// half also requests step kt+2 (A and W: 16 one-KB LDS-DMA pieces per wave, two per gap) into the slot the step's
// barrier has just freed, so every request has between half a step and a whole one to land.
// Epilogue staging: 8 KB per wave beside the ring (32 rows x 256 B per pass), stores of 4 rows x 256 B.
// Epilogues: BIAS, BIAS_QUICKGELU (reference clip/model.py:186, :197, :208-212).
#include "../../dfd-clip_amd/csrc/gemm256p_common.hpp"

namespace {

constexpr int QSTAGE = 8192;

template <int EPI, int RB>
__global__ __launch_bounds__(256) void gemm256q_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int TMU = 32 * RB;    // rows a tile uses
  constexpr int WROWS = 16 * RB;  // rows per wave
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + 4 * QSTAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = tiles_m * tiles_n;

  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)(uint32_t)(a.M * a.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)(uint32_t)((int64_t)a.N * a.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(uint32_t)(a.M * a.ldc * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias ? a.bias : reinterpret_cast<const float*>(a.W)), 0,
                                                                        a.bias ? a.N * 4 : 0, 0x00020000);

  // ---- LDS-DMA staging: wave w fills rows [64w, 64w+64) of A and of W in 8-row pieces (1 KiB each) ----------
  uint32_t vA[8], vW[8];
  const uint32_t lda2 = (uint32_t)(a.lda * 2), ldw2 = (uint32_t)(a.ldw * 2);
  const uint32_t a_last = (uint32_t)(a.M - 1) * lda2;  // rows beyond M re-read the last one
  auto set_a = [&](const Tile& t) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t pr = (uint32_t)(l >> 3), pp = (uint32_t)(l & 7);
    const uint32_t ch0 = (pp ^ (pr >> 1)) << 4;  // chunk swizzle pos ^ ((row >> 1) & 7): rows 8p + prow -> (4 (p & 1)) ^ (prow >> 1)
    const uint32_t row0 = ((uint32_t)t.m0 + (uint32_t)(wave * 64)) * lda2 + pr * lda2;
#pragma unroll
    for (int p = 0; p < 8; ++p) vA[p] = min(row0 + (uint32_t)(p * 8) * lda2, a_last) + (ch0 ^ ((p & 1) << 6));
  };
  auto set_w = [&](const Tile& t) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t pr = (uint32_t)(l >> 3), pp = (uint32_t)(l & 7);
    const uint32_t ch0 = (pp ^ (pr >> 1)) << 4;
    const uint32_t row0 = ((uint32_t)t.n0 + (uint32_t)(wave * 64)) * ldw2 + pr * ldw2;
#pragma unroll
    for (int p = 0; p < 8; ++p) vW[p] = row0 + (uint32_t)(p * 8) * ldw2 + (ch0 ^ ((p & 1) << 6));
  };
  auto issue_a1 = [&](int kt, int slot, int p, uint32_t kill) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(smem + slot * SLOT + (wave * 64 + p * 8) * ROWB), 16, vA[p] | kill, kt * ROWB, 0, 0);
  };
  auto issue_w1 = [&](int kt, int slot, int p, uint32_t kill) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srdW, (lds_ptr_t)(smem + slot * SLOT + A_BYTES + (wave * 64 + p * 8) * ROWB), 16, vW[p] | kill, kt * ROWB, 0, 0);
  };
  auto issue_a = [&](int kt, int slot) {
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_a1(kt, slot, p, 0u);
  };
  auto issue_w = [&](int kt, int slot) {
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_w1(kt, slot, p, 0u);
  };

  // ---- fragment reads: lane (fr, fq) reads row fr of a 16-row block, chunk 4*ks + fq ----------------
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  int offA[2], offW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offA[ks] = (wr * WROWS + fr) * ROWB + (((4 * ks + fq) ^ sw) << 4);  // WROWS is a multiple of 16: the swizzle term is unchanged
    offW[ks] = A_BYTES + (wc * 128 + fr) * ROWB + (((4 * ks + fq) ^ sw) << 4);
  }
  auto read_w1 = [&](bf16x8 (&w)[8], int slot, int ks, int j) {
    w[j] = *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + offW[ks] + j * 16 * ROWB);
  };
  auto read_a1 = [&](bf16x8 (&f)[RB], int slot, int ks, int i) {
    f[i] = *reinterpret_cast<const bf16x8*>(smem + slot * SLOT + offA[ks] + i * 16 * ROWB);
  };

  f32x4 acc[RB][8];
  // one half step: 8*RB MFMAs on (w, f) with the half's LDS reads / LDS-DMA requests (`side`) woven between them.
  // One wave per SIMD: nothing else fills the matrix pipe while this wave issues something else, so the other
  // instructions go one at a time into the shadow of single MFMAs (16 cycles of pipe time for 4 of issue); the
  // order is imposed with sched_group_barrier: PAT 0 = {2 MFMA, 1 LDS read} x 16, PAT 1 = {MFMA, LDS read, MFMA,
  // LDS read, MFMA, LDS-DMA, MFMA, LDS-DMA} x 8, PAT 2 = {2 MFMA, 1 LDS-DMA} x 16; then the remaining MFMAs.
  auto half = [&](const bf16x8 (&w)[8], const bf16x8 (&f)[RB], auto pat_c, auto&& side) {
    constexpr int PAT = decltype(pat_c)::value;
    __builtin_amdgcn_sched_barrier(0);
    side();
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (PAT == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      } else if constexpr (PAT == 1) {
        if (k < 8) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * RB - 32, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = a.K / TK;  // >= 2
  int idx = pos;            // < ntiles: the launcher keeps G <= ntiles
  Tile cur = decode_tile(idx, tiles_m, tiles_n, TMU);
  set_a(cur);
  set_w(cur);
  int par = 0;  // ring slot of the current tile's step 0
  issue_a(0, 0);
  issue_w(0, 0);
  issue_a(1, 1);
  issue_w(1, 1);
  wait_vm<16>();  // step 0 landed, step 1 may be in flight
  __builtin_amdgcn_s_barrier();
  int s_prev = 0;  // stores of the previous epilogue still in flight when this tile's loop starts

  bf16x8 w0[8], w1[8], f0[RB], f1[RB];
  unsigned char* const ep = smem + RING + wave * QSTAGE;

  for (;;) {
    const int nidx = idx + G;
    const bool has_next = nidx < ntiles;
    const Tile nxt = has_next ? decode_tile(nidx, tiles_m, tiles_n, TMU) : cur;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) read_w1(w0, par, 0, j);
#pragma unroll
    for (int i = 0; i < RB; ++i) read_a1(f0, par, 0, i);
    f32x4 b4[8];  // bias of this wave's 128 columns

    auto load_col_vectors = [&] {
      int lb = lane;
      asm volatile("" : "+v"(lb));
      const uint32_t coff = (uint32_t)((cur.n0 + wc * 128 + (lb >> 4) * 4) * 4);
      const uint32_t boff = a.bias ? coff : 0xffffffffu;  // no bias: out of range reads 0
#pragma unroll
      for (int j = 0; j < 8; ++j) b4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdB, boff, j * 64, 0));
    };
    auto pin_col_vectors = [&] {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(b4[j]));
    };
    auto step_wait = [&](int kt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (kt == 0) {
        if (s_prev == 0) wait_vm<0>();
        else wait_vm<4 * RB>();
      } else {
        wait_vm<0>();
      }
    };

    auto frag_reads = [&](bf16x8 (&w)[8], bf16x8 (&f)[RB], int slot, int ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        read_w1(w, slot, ks, j);
        if (j < RB) read_a1(f, slot, ks, j);
      }
    };
    auto kstep = [&](int kt, auto last_c) {
      constexpr bool last = decltype(last_c)::value;
      const int slot = (par + kt) & 1;
      if (has_next && kt == nk - 2) {  // the requests below move on to the next tile
        set_a(nxt);
        set_w(nxt);
      }
      // no next tile: the (unconditional) requests of the last two steps are pushed out of range and fetch nothing
      const uint32_t kill = (!has_next && kt + 2 >= nk) ? 0xffffffffu : 0u;
      // half 0 | read k-half 1 of this step
      half(w0, f0, std::integral_constant<int, 0>{}, [&] { frag_reads(w1, f1, slot, 1); });
      if constexpr (last) load_col_vectors();
      step_wait(kt);  // my reads of this slot are done and step kt+1 has landed (requested one step ago)
      if constexpr (last) pin_col_vectors();
      __builtin_amdgcn_s_barrier();
      // half 1 (registers only) | read k-half 0 of step kt+1 from the other slot | request step kt+2 into this one
      const int k2 = kt + 2 < nk ? kt + 2 : kt + 2 - nk;
      // (program order = the order asked of the scheduler: an LDS-DMA write and an LDS read do not pass each other)
      half(w1, f1, std::integral_constant<int, last ? 2 : 1>{}, [&] {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          if constexpr (!last) {
            read_w1(w0, slot ^ 1, 0, p);
            if (p < RB) read_a1(f0, slot ^ 1, 0, p);
            else read_w1(w0, slot ^ 1, 0, p);  // 224-row tiles: keep two reads per group (a repeat)
          }
          issue_a1(k2, slot, p, kill);
          issue_w1(k2, slot, p, kill);
        }
      });
    };
    for (int kt = 0; kt < nk - 1; ++kt) kstep(kt, std::false_type{});
    kstep(nk - 1, std::true_type{});

    // ---- epilogue ------------------------------------------------------------------------------------------------
    int le = lane;
    asm volatile("" : "+v"(le));
    const int er = le & 15, eq = le >> 4;     // accumulator fragment: row er of a 16-row block, columns 4*eq ..
    const int drow = le >> 4, dc = le & 15;   // drain: row drow of a 4-row group, 16-byte chunk dc of its 256 bytes
    const int nb = cur.n0 + wc * 128;
    const int64_t mrow0 = (int64_t)cur.m0 + wr * WROWS + drow;  // first row this lane stores
    const int rows_left = (int)min((int64_t)0x7fffffff, a.M - mrow0);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] += b4[j];
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
    // 4 passes of 32 rows parked as bf16 (8 KB: 256-byte rows, 16-byte chunk c of row r at position c ^ (r & 15));
    // 8 wave-stores of 4 rows x 256 B per pass
    unsigned char* const park = ep + er * 256 + (eq & 1) * 8;  // + ii*4096 + (((2j + (eq >> 1)) ^ er) << 4)
    const int pc = eq >> 1;
    const uint32_t cbase = (uint32_t)((mrow0 * a.ldc + nb + dc * 8) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * q + ii;
        if (i >= RB) continue;  // 224-row tiles: the last pass holds 16 rows
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 v = activate(acc[i][j]);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(park + ii * 4096 + (((2 * j + pc) ^ er) << 4)) = o;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int rloc = q * 32 + rr * 4;  // row of the store relative to this lane's first row
        if (rloc >= WROWS) continue;
        const int row = rr * 4 + drow;
        const v4i d = *reinterpret_cast<const v4i*>(ep + row * 256 + ((dc ^ (row & 15)) << 4));
        const uint32_t off = rloc < rows_left ? cbase + (uint32_t)rloc * (uint32_t)(a.ldc * 2) : 0xffffffffu;  // out of range: dropped
        store_out(d, srdC, off, a.stream_out);
      }
    }
    if (!has_next) break;
    s_prev = 4 * RB;
    par = (par + nk) & 1;
    idx = nidx;
    cur = nxt;
  }
}

template <int EPI>
int launch256q(const GemmArgs& a, hipStream_t st) {
  const int tiles_n = a.N / TN;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      dfd_set_error("dfd_gemm(persistent, 4 waves): cannot query the device");
      return DFD_ERR_LAUNCH;
    }
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
  if (use224) hipLaunchKernelGGL((gemm256q_kernel<EPI, 7>), dim3(grid), dim3(256), 0, st, a, tiles_m, tiles_n);
  else hipLaunchKernelGGL((gemm256q_kernel<EPI, 8>), dim3(grid), dim3(256), 0, st, a, tiles_m, tiles_n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_gemm(persistent, 4 waves): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}

}  // namespace

// 1 = not served.  The caller (gemm256p.hip) has checked the shape.
int dfd_gemm256q_launch(const GemmArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case DFD_EPI_BIAS:
      return launch256q<DFD_EPI_BIAS>(a, st);
    case DFD_EPI_BIAS_QUICKGELU:
      return launch256q<DFD_EPI_BIAS_QUICKGELU>(a, st);
    default:
      return 1;
  }
}
