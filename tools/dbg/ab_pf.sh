#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
V="${1:-pf1 pf2}"
for r in 1 2; do for v in $V; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter 8:32:8192 8:32:18432 8:32:32768 4:16:18432 8:32:5000 2>/dev/null || echo "FAILED/timeout"; done; done
LAST=$(echo $V | awk '{print $NF}')
cp .ab/lib$LAST.so $L; echo "== $LAST tests"; timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_hybrid.py tests/test_gpu_extremes.py -q -m gpu 2>&1 | tail -n 5
cp /tmp/keep.so $L
