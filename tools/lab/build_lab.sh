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
v lab_nt -DLAB_NT_STORE=1
v lab_nogelu -DLAB_NO_GELU=1
v lab_nostore_nogelu -DLAB_NO_STORE=1 -DLAB_NO_GELU=1
v lab_same_nostore -DLAB_SAME_TILE=1 -DLAB_NO_STORE=1
v lab_nods -DLAB_NO_DSREAD=1
v lab_nobar -DLAB_NO_BARRIER=1
v lab_noglds_nods -DLAB_NO_GLDS=1 -DLAB_NO_DSREAD=1
v lab_mfma_only -DLAB_NO_GLDS=1 -DLAB_NO_DSREAD=1 -DLAB_NO_BARRIER=1 -DLAB_NO_EPILOGUE=1
P=../../dfd-clip_amd/csrc/gemm256p.hip
hipcc $F -c $P -o build/lab_p.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_nostore -DLABP_NO_STORE=1 -c $P -o build/lab_p_nostore.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_a1 -DLABP_STORE_AUX=1 -c $P -o build/lab_p_a1.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_a3 -DLABP_STORE_AUX=3 -c $P -o build/lab_p_a3.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_a16 -DLABP_STORE_AUX=16 -c $P -o build/lab_p_a16.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_a18 -DLABP_STORE_AUX=18 -c $P -o build/lab_p_a18.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_nt -DLABP_STORE_AUX=2 -c $P -o build/lab_p_nt.o &
hipcc $F -DDFD_GEMM256P_TRY=labp_noepi -DLABP_NO_EPI=1 -c $P -o build/lab_p_noepi.o &
wait
hipcc $F -Wno-unused-result -Wno-unused-value -c gemm_lab.hip -o build/main.o
hipcc --offload-arch=gfx950 build/main.o build/lab_*.o -o gemm_lab
