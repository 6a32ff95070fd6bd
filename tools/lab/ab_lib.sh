#!/bin/bash
# A/B of two builds of the library on one box: tools/lab/ab_lib.sh <libA.so> <libB.so> <rounds> -- <command ...>
# (alternates the two files into place; the last one copied stays, so end with the build you want to keep)
A=$1; B=$2; R=$3; shift 4
L=dfd-clip_amd/libdfdclip_hip.so
for i in $(seq 1 "$R"); do
  cp "$A" $L && echo "== A ($A) round $i" && "$@" || exit 1
  cp "$B" $L && echo "== B ($B) round $i" && "$@" || exit 1
done
