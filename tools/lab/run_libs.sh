#!/bin/bash
# run one command under several builds of the library: tools/lab/run_libs.sh "<lib1> <lib2> .." -- <command ...>
# (each file is copied into place in turn; the product build is restored at the end)
LIBS=$1; shift 2
L=dfd-clip_amd/libdfdclip_hip.so
cp $L /tmp/lib_keep.so
for f in $LIBS; do cp "$f" $L && echo "== $f" && "$@" || break; done
cp /tmp/lib_keep.so $L
