#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
V="${1:-base xlA xlB xlC xlD}"
for r in 1 2; do for v in $V; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 120 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 2>/dev/null || echo "FAILED/timeout"; done; done
for v in $V; do cp .ab/lib$v.so $L; echo "== $v tests"; timeout 600 python -m pytest tests/test_gpu_fused_step.py -q -m gpu -x -k "heavy or oracle_pipeline or recent" 2>&1 | tail -n 3; done
cp /tmp/keep.so $L
