#!/bin/bash
# r6 GPU call 20: what the reference-layout ring column costs the hybrid C4 step today (the store left out: results wrong, timing valid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2 3; do for v in base noring; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 300 python tools/ab_step.py hybrid 8:32:18432 8:32:4096 2>/dev/null || echo FAILED; done; done > gpurun_out/r6_c20_hybrid_noring.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c20_hybrid_noring.txt
