#!/bin/bash
# same-box A/B of the step variants (tools/ab_variant.sh builds): variants installed alternately, three rounds: tools/ab_alternate.sh "v1 v2" "H:HQ:S ..."
export TMPDIR=/tmp
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
V="${1:-base oe1 oe2 pro key1 dma xcd1 xcd2}"
SH="${2:-8:32:4096 8:32:2560}"
for r in 1 2 3; do for v in $V; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 150 python tools/ab_step.py ${POLICY:-heavy_hitter} $SH 2>/dev/null || echo "FAILED/timeout"; done; done | tee gpurun_out/r4_ab1.txt
cp /tmp/keep.so $L
