#!/bin/bash
# private build of the attention kernels with in-kernel cycle stamps (tools/lab/attn_stamps_persist.py)
set -e
C="$(cd "$(dirname "$0")/../../dfd-clip_amd/csrc" && pwd)"
mkdir -p "$(dirname "$0")/build"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DATTN_STAMPS=1 -I"$C/../../include" "$C/attention.hip" "$(dirname "$0")/attention_mfma_stamps.hip" "$C/capi.hip" \
  -o "$(dirname "$0")/build/libattn_stamps.so"
