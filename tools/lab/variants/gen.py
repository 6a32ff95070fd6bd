#!/usr/bin/env python3
"""Generates scheduling variants of the persistent GEMM's K step as patched copies of dfd-clip_amd/csrc/gemm256p.hip
(v1.hip .. v3.hip, not tracked) for tools/lab/gemm_variants.  Variants of the second batch of
profiles/r02_gemm_kstep_scheduling_variants.txt: third batch: v1 = the W requests of step kt+1 split over the gaps of P0 and P1, v2 = W of step kt+2 requested with A in P3
(nothing in P0), v3 = the product a second time (noise floor).  Earlier batches: the results file.  build.sh compiles and links them."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "..", "..", "dfd-clip_amd", "csrc", "gemm256p.hip")).read()
src = src.replace('#include "gemm256p_common.hpp"', '#include "../../../dfd-clip_amd/csrc/gemm256p_common.hpp"')


def sub(text, old, new):
    assert old in text, old[:60]
    return text.replace(old, new)


# v1: the four W pieces of step kt+1 split over the gaps of P0 and P1 (two each)
v1 = sub(src, """      phase(wA, lo, 0, [&] {
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
      });""", """      auto issue_w2 = [&](int kk, int sl, int p0) {
        unsigned char* d = smem + sl * SLOT + A_BYTES + wave * 32 * ROWB;
#pragma unroll
        for (int p = p0; p < p0 + 2; ++p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdW, (lds_ptr_t)(d + p * 8 * ROWB), 16, vW[p], kk * ROWB, 0, 0);
      };
      phase(wA, lo, 0, [&] {
        read_a(hi, slot, 0, 1);
        if (kt >= 1) {
          if (!last) issue_w2(kt + 1, slot ^ 1, 0);
          else if (has_next) issue_w2(0, slot ^ 1, 0);
        }
      });
      phase(wA, hi, 1, [&] {
        read_w(wB, slot, 1);
        read_a(lo, slot, 1, 0);
        if (kt >= 1) {
          if (!last) issue_w2(kt + 1, slot ^ 1, 2);
          else if (has_next) issue_w2(0, slot ^ 1, 2);
        }
      });""")
# v2: W of step kt+2 requested with A of step kt+2 in P3 (a step and a quarter ahead), nothing in P0
v2 = sub(src, """        if (kt == nk - 2) set_a(nxt);
        if (kt == nk - 1) set_w(nxt);
      }
      // P0: k-half 0, rows 0-3 | prefetch rows 4-7 | request W of step kt+1 (a tile's step 1 is requested before its loop)
      phase(wA, lo, 0, [&] {
        read_a(hi, slot, 0, 1);
        if (kt >= 1) {
          if (!last) issue_w(kt + 1, slot ^ 1);
          else if (has_next) issue_w(0, slot ^ 1);
        }
      });""", """        if (kt == nk - 2) { set_a(nxt); set_w(nxt); }
      }
      phase(wA, lo, 0, [&] { read_a(hi, slot, 0, 1); });""")
v2 = sub(v2, """        if (kt + 2 < nk) issue_a(kt + 2, slot);
        else if (has_next) issue_a(kt + 2 - nk, slot);
      });
    };
    if constexpr (F8) {""", """        if (kt + 2 < nk) { issue_a(kt + 2, slot); issue_w(kt + 2, slot); }
        else if (has_next) { issue_a(kt + 2 - nk, slot); issue_w(kt + 2 - nk, slot); }
      });
    };
    if constexpr (F8) {""")
v2 = sub(v2, """      kstep(nk - 1, std::true_type{});
    }
    if (has_next) issue_w(1, (par + nk - 1) & 1);""", """      kstep(nk - 1, std::true_type{});
    }
    if (F8 && has_next) issue_w(1, (par + nk - 1) & 1);""")
a2 = src.index("  auto phase = [&](const bf16x8 (&w)[4], const bf16x8 (&f)[4], int half, auto&& mid) {")
v3 = src  # the product itself a second time (noise floor)
for name, text in (("v1", v1), ("v2", v2), ("v3", v3)):
    open(os.path.join(HERE, name + ".hip"), "w").write(text)
print("wrote v1.hip v2.hip v3.hip")
