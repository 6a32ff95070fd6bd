#!/bin/bash
# builds tools/lab/gemm_lab from ablated variants of dfd-clip_amd/csrc/gemm256.hip
set -e
cd "$(dirname "$0")"
SRC=../../dfd-clip_amd/csrc/gemm256.hip
F="--offload-arch=gfx950 -O3 -std=c++17"
mkdir -p build
v() { hipcc $F -DDFD_GEMM256_TRY=$1 ${@:2} -c $SRC -o build/$1.o & }
v lab_full
v lab_same -DLAB_SAME_TILE=1
v lab_same_noepi -DLAB_SAME_TILE=1 -DLAB_NO_EPILOGUE=1
v lab_noepi -DLAB_NO_EPILOGUE=1
v lab_noglds -DLAB_NO_GLDS=1
v lab_nostore -DLAB_NO_STORE=1
v lab_nogelu -DLAB_NO_GELU=1
v lab_nostore_nogelu -DLAB_NO_STORE=1 -DLAB_NO_GELU=1
v lab_same_nostore -DLAB_SAME_TILE=1 -DLAB_NO_STORE=1
v lab_nods -DLAB_NO_DSREAD=1
v lab_nobar -DLAB_NO_BARRIER=1
v lab_noglds_nods -DLAB_NO_GLDS=1 -DLAB_NO_DSREAD=1
v lab_mfma_only -DLAB_NO_GLDS=1 -DLAB_NO_DSREAD=1 -DLAB_NO_BARRIER=1 -DLAB_NO_EPILOGUE=1
wait
hipcc $F -Wno-unused-result -c gemm_lab.hip -o build/main.o
hipcc --offload-arch=gfx950 build/main.o build/lab_*.o -o gemm_lab
