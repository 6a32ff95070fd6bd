// Persistent bf16 MFMA self-attention for frames of 32*8 + 1 = 257 tokens (ViT-L/14 at 224 px: 256 patches + CLS;
// reference clip/model.py:188-195), head_dim 64.  Same arithmetic, same bits as the per-item kernel of attention_mfma.hip.
//
// Nine 32-query blocks with one live row in the ninth do not fit the chip: a wave of the 288-key product holds
// 144 score registers, so a SIMD takes two waves and a CU eight — a ninth compute wave (or a loader wave, as the
// 197-token kernel has) would halve the register budget of all of them.  So:
//   * eight waves, one 32-query block each (queries 0..255), all of them full;
//   * query 256, the extra row, is spread over the workgroup so that no wave carries a second round:
//       - scores: wave w multiplies key block w (wave 0: block 8 as well) with the operands of the first product SWAPPED
//         (S[q][key] instead of Sᵀ[key][q]; every row of the A operand is the one query): four MFMAs, and each lane
//         ends up with the score of ITS key, which goes to LDS;
//       - softmax: wave 0 — 9 exponentials per lane instead of 144, the sums run across lanes in the order the
//         register form uses; P (bf16) and 1/l go to LDS;
//       - second product: waves 1 and 2 take one 32-channel half each, riding along their own second product (one more
//         MFMA per step over the V fragments already in registers; all 32 P columns equal, column 0 is stored).
//     The hand-overs are two counters in LDS (ds_add by the producer behind its writes — a wave's LDS operations
//     execute in order; the consumer polls; it practically never has to, the producers are a phase ahead);
//   * no loader wave: wave w issues pieces w, w+8, .. of the NEXT item's K and V images by LDS-DMA right behind the
//     barrier at the top of an item, a whole item ahead of their use (wave 7 also the extra row's Q, 128 bytes);
//     Q fragments go from global memory straight into registers (inline asm, issued behind the first product,
//     consumed at the top of the next item);
//   * ONE workgroup barrier per item (top: K(n), V(n) of every wave have landed, everybody is done with item n-1);
//     the output tile leaves through a 2 KB staging area per wave in two 16-row passes as whole 128-byte lines.
//   LDS: [K0][V0][K1][V1] images of 257 128-byte rows (131.6 KB), 8 x 2 KB staging, 2.5 KB for the extra row.
// vmcnt is counted by hand: memory operations retire in order, so "my DMAs and Q loads have landed" is "at most the
// stores issued after them are still in flight" (4 per wave, 5 for the two waves that also store half of the extra
// row; all of them in range — a fully out-of-range store would retire early and break the count).
// Stores go through attn_store_line() (no register in the scalar-offset field: attention_common.hpp says why).
#include "attention_common.hpp"

namespace {

constexpr int NB = 9;              // 32-key blocks
constexpr int NW = NB - 1;         // compute waves = full 32-query blocks
constexpr int XTOKENS = 32 * NW + 1;
constexpr int IMG = XTOKENS * 128;  // bytes of a K or V image
constexpr int NP = (XTOKENS * 8 + 63) >> 6;  // 1 KB pieces per image (the last one holds one row)
constexpr int STG = 2048;          // staging bytes per wave: 16 query rows x 128 B
// the extra row's corner of LDS, behind the staging areas
constexpr int X_SC = 0;            // 288 f32 scores
constexpr int X_P = 1152;          // 288 bf16 probabilities
constexpr int X_INV = 1728;        // 1 / row sum
constexpr int X_CNT_S = 1792;      // score hand-overs so far (8 per item)
constexpr int X_CNT_P = 1796;      // softmax hand-overs so far (1 per item)
constexpr int X_Q = 2048;          // its Q row, one 128-byte image per item parity
constexpr int X_BYTES = 2560;
constexpr int XROW_LDS = 4 * IMG + NW * STG + X_BYTES;
// Which waves carry the extra row: its softmax on wave 0, the halves of its second product on waves 1 and 2.
constexpr int XW_SOFTMAX = 0, XW_PV0 = 1;

__device__ __forceinline__ uint32_t lds_addr(const unsigned char* p) { return (uint32_t)(uintptr_t)(lds_ptr_t)(const_cast<unsigned char*>(p)); }
// LDS stores in the middle of an item are inline asm: in front of an LDS store it knows about the compiler puts a vmcnt(0)
// while LDS-DMA is in flight (it cannot tell that the two never touch the same bytes) — a wait for the next item's images.
__device__ __forceinline__ void lds_write_b32(const unsigned char* p, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b16(const unsigned char* p, bf16_t v) {
  asm volatile("ds_write_b16 %0, %1" ::"v"(lds_addr(p)), "v"((uint32_t)__builtin_bit_cast(unsigned short, v)) : "memory");
}
// until the counter at `p` (LDS) has reached `target`
__device__ __forceinline__ void lds_wait_count(const unsigned char* p, int target) {
  int v;
  int polls = 0;  // bounded: a count that never comes (it always does) ends in a wrong extra row, not in a hung GPU
  do {
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  } while (__builtin_amdgcn_readfirstlane(v) < target && ++polls < (1 << 20));
}
__device__ __forceinline__ void lds_count(const unsigned char* p, int lane) {  // behind the caller's LDS writes (in order)
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(lds_addr(p)), "v"(1) : "memory");
}

__global__ __launch_bounds__(512) void attn_mfma_xrow_kernel(const bf16_t* __restrict__ qkv, uint32_t ldq /* bytes */,
                                                             bf16_t* __restrict__ out, uint32_t ldo /* bytes */, int heads,
                                                             int n_frames, uint32_t qkv_bytes, uint32_t out_bytes,
                                                             float scale_log2e) {
  constexpr int tokens = XTOKENS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));  // opaque: per-item addresses are rebuilt from it instead of living in registers
  const int r = lane & 31, h = lane >> 5;
  const int ksw = (r >> 1) & 7;
  const uint32_t Db = (uint32_t)heads * HD * 2;
  const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
  const uint64_t qaddr = reinterpret_cast<uint64_t>(qkv);
  const v4i_t srdQw = {(int)(uint32_t)qaddr, (int)((uint32_t)(qaddr >> 32) & 0xffffu), (int)qkv_bytes, 0x00020000};  // for the asm loads
  unsigned char* const stg = smem + 4 * IMG + wave * STG;
  unsigned char* const xs = smem + 4 * IMG + NW * STG;

  // a finite start state for what the images do not cover (MFMA reads of key rows 257.. fall into whatever follows),
  // and zero counters
  for (int i = tid * 16; i < XROW_LDS; i += 512 * 16) *reinterpret_cast<v4i_t*>(smem + i) = v4i_t{0, 0, 0, 0};
  __syncthreads();

  // item n of this workgroup: the workgroups of one XCD (blockIdx % 8) walk the heads of the same few frames together
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  auto item_of = [&](int n, int& fr_, int& hd_) -> bool {
    const int li = n * per_xcd + slot;
    const int fl = li / heads;
    hd_ = li - fl * heads;
    fr_ = fl * 8 + xcd;
    return fr_ < n_frames;
  };
  auto sbase_of = [&](int f, int hh) { return (uint32_t)f * tokens * ldq + hh * (HD * 2); };
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };  // no release fence: stores stay in flight
  auto issue_kv = [&](int f, int hh, int buf) {  // my eighth of the pieces of both images; the last wave: + the extra row's Q
    const uint32_t sb = sbase_of(f, hh);
    int l = lane;
    asm volatile("" : "+v"(l));  // (nothing derived from the lane id for this stays live across the item)
    attn_stage_k(srdQ, smem + buf * 2 * IMG, sb, ldq, Db, l, NP, tokens, wave, NW);
    attn_stage_v(srdQ, smem + buf * 2 * IMG + IMG, sb, ldq, Db, l, NP, tokens, wave, NW);
    if (wave == NW - 1 && l < 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdQ, (lds_ptr_t)(xs + X_Q + buf * 128), 16, (uint32_t)(tokens - 1) * ldq + l * 16, sb, 0, 0);
  };
  // Q fragment of query `row`: 16-byte chunks 2s + h, s = 0..3.  Inline asm: the compiler neither sees these loads nor
  // waits for them (a load it knows about would cost a vmcnt(0), i.e. a wait for the DMA of the next item as well).
  auto load_q = [&](v4i_t (&q)[4], uint32_t row, uint32_t sb) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const uint32_t off = sb + row * ldq + (uint32_t)(l >> 5) * 16;
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %4, %5, 0 offen\n\t"
        "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:32\n\t"
        "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:64\n\t"
        "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:96"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
        : "v"(off), "s"(srdQw)
        : "memory");
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  int frame, head, nframe = 0, nhead = 0;
  bool have = item_of(0, frame, head);
  v4i_t qn[4];
  if (have) {
    issue_kv(frame, head, 0);
    load_q(qn, (uint32_t)(wave * 32 + r), sbase_of(frame, head));
  }
  for (int n = 0; have; ++n) {
    const int cur = n & 1;
    const unsigned char* Ks = smem + cur * 2 * IMG;
    const unsigned char* Vs = Ks + IMG;
    // my pieces of K(n), V(n) and my Q fragment are in; the stores of item n-1 may still be on their way
    if (n == 0) attn_wait_vm<0>();
    else if (wave == XW_PV0 || wave == XW_PV0 + 1) attn_wait_vm<5>();
    else attn_wait_vm<4>();
    asm volatile("" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]));
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(bf16x8, qn[s]);
    barrier();
    const bool have_next = item_of(n + 1, nframe, nhead);
    if (have_next) issue_kv(nframe, nhead, cur ^ 1);

    // ---- Sᵀ[key][q] for all NB key blocks; K fragments in two groups, each read one half-slice ahead ---------
    f32x16 S[NB];
    {
      constexpr int GA = (NB + 1) / 2, GB = NB / 2;
      bf16x8 kfa[GA], kfb[GB];
      auto read_a = [&](int s) {
#pragma unroll
        for (int kb = 0; kb < GA; ++kb)
          kfa[kb] = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * 128 + (((2 * s + h) ^ ksw) << 4));
      };
      auto read_b = [&](int s) {
#pragma unroll
        for (int kb = 0; kb < GB; ++kb)
          kfb[kb] = *reinterpret_cast<const bf16x8*>(Ks + ((GA + kb) * 32 + r) * 128 + (((2 * s + h) ^ ksw) << 4));
      };
      auto mma_a = [&](int s) {
#pragma unroll
        for (int kb = 0; kb < GA; ++kb) S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[kb], qf[s], s == 0 ? zero16 : S[kb], 0, 0, 0);
      };
      auto mma_b = [&](int s) {
#pragma unroll
        for (int kb = 0; kb < GB; ++kb)
          S[GA + kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfb[kb], qf[s], s == 0 ? zero16 : S[GA + kb], 0, 0, 0);
      };
      read_a(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        read_b(s);
        mma_a(s);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 4) read_a(s + 1);
        mma_b(s);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the next item's Q fragment: issued here, where its registers are free
    if (have_next) load_q(qn, (uint32_t)(wave * 32 + r), sbase_of(nframe, nhead));

    // ---- the extra row's scores of key block `wave`: S[q][key] = Q·Kᵀ with every A row = query 256, so element 0 of a
    // lane is the score of key 32 wave + r (the softmax wave: block 8 too; its keys past 256 are masked by the softmax)
    {
      const unsigned char* qx = xs + X_Q + cur * 128;
      int lx_ = lane;
      asm volatile("" : "+v"(lx_));
      const int r = lx_ & 31, h = lx_ >> 5, ksw = (r >> 1) & 7;
      auto xscore = [&](int kb) {
        f32x16 T;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(qx + (2 * s + h) * 16);
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * 128 + (((2 * s + h) ^ ksw) << 4));
          T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf, s == 0 ? zero16 : T, 0, 0, 0);
        }
        if (h == 0) lds_write_b32(xs + X_SC + (kb * 32 + r) * 4, T[0]);
      };
      xscore(wave);
      if (wave == XW_SOFTMAX) xscore(NB - 1);
      lds_count(xs + X_CNT_S, lane);
    }
    // ---- its softmax (one wave): lane r stands for element e = (r & 3) + 4 (r >> 3), half (r >> 2) & 1 of the
    // register form: per-lane sums over the blocks first, then the same tree over e (lane bits 0, 1, 3, 4) and last the
    // two halves (bit 2)
    if (wave == XW_SOFTMAX) {
      lds_wait_count(xs + X_CNT_S, NW * (n + 1));
      float sx[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) sx[kb] = *reinterpret_cast<const float*>(xs + X_SC + (kb * 32 + r) * 4);
      if (r >= tokens - (NB - 1) * 32) sx[NB - 1] = -INFINITY;
      float mx = sx[0];
#pragma unroll
      for (int kb = 1; kb < NB; ++kb) mx = fmaxf(mx, sx[kb]);
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float nmc = -mx * scale_log2e;
      float lx = 0.f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        sx[kb] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[kb], scale_log2e, nmc));
        lx += sx[kb];
        if (h == 0) lds_write_b16(xs + X_P + (kb * 32 + r) * 2, (bf16_t)sx[kb]);
      }
      lx += __shfl_xor(lx, 1, 64);
      lx += __shfl_xor(lx, 2, 64);
      lx += __shfl_xor(lx, 8, 64);
      lx += __shfl_xor(lx, 16, 64);
      lx += __shfl_xor(lx, 4, 64);
      if (lane == 0) lds_write_b32(xs + X_INV, 1.0f / lx);
      lds_count(xs + X_CNT_P, lane);
    }

    const float l = attn_softmax<NB, true>(S, tokens, h, scale_log2e);
    f32x16 O[2];
    // The extra row's second product rides along on two waves, one 32-channel half each: the usual product over the V
    // fragments of the main one; its P operand (keys base + 4h + {0..3}, base + 8 + 4h + {0..3}, from LDS) is the same
    // for every column, so all 32 columns come out equal and column 0 is stored.
    const bool xpv = wave == XW_PV0 || wave == XW_PV0 + 1;
    f32x16 OX = zero16;
    {
      int lv = lane;
      asm volatile("" : "+v"(lv));
      if (xpv) lds_wait_count(xs + X_CNT_P, n + 1);  // (the softmax wave finished it while this wave did its own)
      const unsigned char* px = xs + X_P + 8 * (lv >> 5);
      attn_pv<NB, true>(S, Vs, lv, O, [&](const bf16x8 (&vf)[2], int step) {
        if (xpv) {
          const short4v plo = *reinterpret_cast<const short4v*>(px + step * 32);
          const short4v phi = *reinterpret_cast<const short4v*>(px + step * 32 + 16);
          short8 pp;
#pragma unroll
          for (int e = 0; e < 4; ++e) { pp[e] = plo[e]; pp[4 + e] = phi[e]; }
          OX = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wave == XW_PV0 ? vf[0] : vf[1], __builtin_bit_cast(bf16x8, pp), OX, 0, 0, 0);
        }
      });
    }

    // ---- output: [q][64] bf16 through my staging area, 16 query rows at a time, then whole 128-byte lines ------
    const uint32_t obase = (uint32_t)frame * tokens * ldo + head * (HD * 2);
    {
      const float inv = 1.0f / l;
      int lo = lane;
      asm volatile("" : "+v"(lo));
      const int r = lo & 31, h = lo >> 5, ksw = (r >> 1) & 7, lane = lo;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if ((r >> 4) == pass) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(O[dt][4 * g + e] * inv);
              *reinterpret_cast<bf16x4*>(stg + (r & 15) * 128 + (((4 * dt + g) ^ ksw) << 4) + 8 * h) = o;
            }
        }
        // lanes exchange data through LDS here: to the compiler a lane that wrote nothing re-reads what it read in the
        // previous pass (it folded the read-back into the masked region), so the memory state is made opaque
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rl = 8 * i + (lane >> 3), pos = lane & 7;
          const v4i_t d = *reinterpret_cast<const v4i_t*>(stg + rl * 128 + ((pos ^ ((rl >> 1) & 7)) << 4));
          const uint32_t qq = (uint32_t)(wave * 32 + 16 * pass + rl);
          attn_store_line(d, srdO, obase + qq * ldo + pos * 16);
        }
      }
    }

    // ---- the extra row's output, channels 32 dt .. 32 dt + 31
    if (xpv) {
      const int dt = wave - XW_PV0;
      const float inv = *reinterpret_cast<const float*>(xs + X_INV);
      int lo = lane;
      asm volatile("" : "+v"(lo));
      if ((lo & 31) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(OX[4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(stg + g * 16 + (lo >> 5) * 8) = o;
        }
      }
      asm volatile("" ::: "memory");
      if (lo < 4) {
        const v4i_t d = *reinterpret_cast<const v4i_t*>(stg + lo * 16);
        attn_store_line(d, srdO, obase + (uint32_t)(tokens - 1) * ldo + dt * 64 + lo * 16);
      }
    }
    have = have_next;
    frame = nframe;
    head = nhead;
  }
}

}  // namespace

// 1 = not served (another token count, too few items to fill the chip, or 32-bit offsets do not reach)
int dfd_attention_mfma_xrow_try(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, int n_frames, int tokens, int heads,
                                float scale, hipStream_t st) {
  if (tokens != XTOKENS || n_frames * heads < 512) return 1;
  const int64_t qkv_bytes = ((int64_t)n_frames * tokens - 1) * ld_qkv * 2 + (int64_t)3 * heads * HD * 2;
  const int64_t out_bytes = ((int64_t)n_frames * tokens - 1) * ld_out * 2 + (int64_t)heads * HD * 2;
  if (qkv_bytes > (int64_t)0xfffffff0 || out_bytes > (int64_t)0xfffffff0 || (ld_qkv % 8) != 0 || (ld_out % 8) != 0) return 1;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_xrow_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int grid = ncu & ~7;  // a whole number of workgroups per XCD
  if (grid < 8) return 1;
  hipLaunchKernelGGL(attn_mfma_xrow_kernel, dim3(grid), dim3(512), (size_t)XROW_LDS, st, static_cast<const bf16_t*>(qkv),
                     (uint32_t)(ld_qkv * 2), static_cast<bf16_t*>(out), (uint32_t)(ld_out * 2), heads, n_frames,
                     (uint32_t)qkv_bytes, (uint32_t)out_bytes, scale * 1.4426950408889634f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dfd_set_error("dfd_attention_fwd(mfma, 257 tokens): launch failed: %s", hipGetErrorString(e));
    return DFD_ERR_LAUNCH;
  }
  return DFD_OK;
}
