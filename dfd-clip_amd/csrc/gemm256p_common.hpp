// Pieces shared by the persistent GEMM kernels (gemm256p.hip: bf16; gemm256p_f8.hip: fp8): tile / ring geometry, the
// tile order, counted vmcnt waits and the output store.  A K step is 128 BYTES of every operand row (64 bf16 or 128
// fp8 elements), so the LDS ring, the LDS-DMA pieces and the XOR swizzle are the same for both element types.
#pragma once
#include <type_traits>

#include "gemm_args.hpp"

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef v4i_t v4i;
typedef int v8i __attribute__((ext_vector_type(8)));

namespace {

// Output stores: non-temporal (aux bit 1) when the caller marks the output as streaming: it then goes past L2
// instead of evicting the operand panels the XCD's other workgroups are reading (c_fc 0.46 -> 0.41 ms).
__device__ __forceinline__ void store_out(v4i_t d, __amdgpu_buffer_rsrc_t srd, uint32_t off, int stream_out) {
  if (stream_out) {
    __builtin_amdgcn_raw_buffer_store_b128(d, srd, off, 0, 2);
  } else {
    __builtin_amdgcn_raw_buffer_store_b128(d, srd, off, 0, 0);
  }
}

constexpr int TM = 256, TN = 256, TK = 64;  // TM: rows of A staged per step; a tile USES 32*RB of them (RB = 8 or 7)
constexpr int ROWB = TK * 2;            // 128 B per LDS row = one cache line
constexpr int A_BYTES = TM * ROWB;      // 32 KB
constexpr int SLOT = (TM + TN) * ROWB;  // 64 KB
constexpr int RING = 2 * SLOT;          // 128 KB
constexpr int STAGE = 4096;             // per wave: 32 rows x 128 B
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Tile {
  int m0, n0;
};

// tile order: column GROUPS of at most 6 tiles, inside a group row panel major / column minor (gemm256.hip)
__device__ __forceinline__ Tile decode_tile(int idx, int tiles_m, int tiles_n, int tile_rows) {
  const int ngroups = (tiles_n + 5) / 6;  // (group widths 2 .. 12 measure the same within 1 %; single columns lose 8-25 %)
  const int gcols = (tiles_n + ngroups - 1) / ngroups;
  int grp = idx / (tiles_m * gcols);
  grp = grp < ngroups - 1 ? grp : ngroups - 1;
  const int rem = idx - grp * tiles_m * gcols;
  const int cols_here = min(gcols, tiles_n - grp * gcols);
  const int tm = rem / cols_here, tn = grp * gcols + (rem - tm * cols_here);
  return Tile{tm * tile_rows, tn * TN};
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
