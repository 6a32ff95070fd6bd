#!/bin/bash
# builds tools/lab/gemm_lab: the two tuned GEMM kernels of the product, linked directly (no Python), timed on the four
# ViT-B/16 encoder shapes.  (The macro-ablated variants of round 1 / early round 2 are gone from the product sources;
# their results are kept in profiles/r02_gemm_ablation_*.txt.)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17"
mkdir -p build
rm -f build/*.o
hipcc $F -c ../../dfd-clip_amd/csrc/gemm256.hip -o build/k256.o &
hipcc $F -c ../../dfd-clip_amd/csrc/gemm256p.hip -o build/k256p.o &
hipcc $F -I../../include -c gemm256q_lab.hip -o build/k256q.o &
hipcc $F -I../../include -save-temps=obj -c gemm256e_lab.hip -o build/k256e.o &
wait
hipcc $F -Wno-unused-result -Wno-unused-value -c gemm_lab.hip -o build/main.o
hipcc --offload-arch=gfx950 build/main.o build/k256.o build/k256p.o build/k256q.o build/k256e.o -o gemm_lab
