#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for v in $1; do cp .ab/lib$v.so $L; echo "== $v"; timeout 120 python tools/trace_one.py --S 4096 2>&1 | tail -1; done | tee gpurun_out/r4_trace.txt
cp /tmp/keep.so $L
