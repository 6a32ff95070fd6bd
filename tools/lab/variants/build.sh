#!/bin/bash
# builds tools/lab/gemm_variants from gen.py's patched copies of the persistent GEMM + the product kernel
set -e
cd "$(dirname "$0")"
python3 gen.py
F="--offload-arch=gfx950 -O3 -std=c++17 -I../../../include"
mkdir -p ../build
for v in 1 2 3; do
  hipcc $F -Ddfd_gemm256p_try=dfd_gemm256p_try_v$v -Ddfd_gemm256p_f8_try=dfd_gemm256p_f8_try_v$v -c v$v.hip -o ../build/v$v.o &
done
hipcc $F -c ../../../dfd-clip_amd/csrc/gemm256p.hip -o ../build/k256p.o &
wait
hipcc $F -w -c main.hip -o ../build/vmain.o
hipcc --offload-arch=gfx950 ../build/vmain.o ../build/k256p.o ../build/v1.o ../build/v2.o ../build/v3.o -o ../gemm_variants
