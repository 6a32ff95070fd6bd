#!/usr/bin/env python3
"""Generates scheduling variants of the persistent GEMM's K step as patched copies of dfd-clip_amd/csrc/gemm256p.hip
(v1.hip .. v3.hip, not tracked) for tools/lab/gemm_variants.  Variants of the second batch of
profiles/r02_gemm_kstep_scheduling_variants.txt: v1 = LDS-DMA requests ahead of the LDS reads inside a gap,
v2 = no sched_barrier inside a phase, v3 = 8 + 8 MFMA groups around the gap.  build.sh compiles and links them."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "..", "..", "dfd-clip_amd", "csrc", "gemm256p.hip")).read()
src = src.replace('#include "gemm256p_common.hpp"', '#include "../../../dfd-clip_amd/csrc/gemm256p_common.hpp"')


def sub(text, old, new):
    assert old in text, old[:60]
    return text.replace(old, new)


v1 = sub(src, """      phase(wA, lo, 0, [&] {
        read_a(hi, slot, 0, 1);
        if (kt >= 1) {
          if (!last) issue_w(kt + 1, slot ^ 1);
          else if (has_next) issue_w(0, slot ^ 1);
        }
      });""", """      phase(wA, lo, 0, [&] {
        if (kt >= 1) {
          if (!last) issue_w(kt + 1, slot ^ 1);
          else if (has_next) issue_w(0, slot ^ 1);
        }
        read_a(hi, slot, 0, 1);
      });""")
v1 = sub(v1, """        if constexpr (!last) {
          read_w(wA, slot ^ 1, 0);
          read_a(lo, slot ^ 1, 0, 0);
        }
        if (kt + 2 < nk) issue_a(kt + 2, slot);
        else if (has_next) issue_a(kt + 2 - nk, slot);
      });""", """        if (kt + 2 < nk) issue_a(kt + 2, slot);
        else if (has_next) issue_a(kt + 2 - nk, slot);
        if constexpr (!last) {
          read_w(wA, slot ^ 1, 0);
          read_a(lo, slot ^ 1, 0, 0);
        }
      });""")
a = src.index("  auto phase = [&](const bf16x8 (&w)[4], const bf16x8 (&f)[4], int half, auto&& mid) {")
b = src.index("  // fp8 form: one operand = both 16-byte chunks of the lane's row")
v2 = src[:a] + src[a:b].replace("    __builtin_amdgcn_sched_barrier(0);\n", "") + src[b:]
v3 = src[:a] + """  auto phase = [&](const bf16x8 (&w)[4], const bf16x8 (&f)[4], int half, auto&& mid) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[i], acc[4 * half + i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mid();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2; i < (half == 0 ? 4 : HB); ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * half + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], f[i], acc[4 * half + i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

""" + src[b:]
for name, text in (("v1", v1), ("v2", v2), ("v3", v3)):
    open(os.path.join(HERE, name + ".hip"), "w").write(text)
print("wrote v1.hip v2.hip v3.hip")
