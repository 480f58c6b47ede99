#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2; do for pol in heavy_hitter recent_global l2 random; do for v in $1; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py $pol 8:32:4096 8:32:2560 2>/dev/null || echo "FAILED"; done; done; done
cp /tmp/keep.so $L
