#!/bin/bash
# r6 GPU call 14: branch-free small loads (CC_V_FLATLOADS=1, flat1) against the r5 form (flat0): alternating step times + parity subset
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2 3; do for v in flat0 flat1; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 1:8:3488 4:16:4096 2>/dev/null || echo FAILED; done; done > gpurun_out/r6_c14_flat_ab.txt 2>&1
for v in flat0 flat1; do cp .ab/lib$v.so $L; for pol in recent_global l2; do echo -n "$v "; timeout 200 python tools/ab_step.py $pol 8:32:4096 1:8:3488 2>/dev/null || echo FAILED; done; done >> gpurun_out/r6_c14_flat_ab.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c14_flat_ab.txt | cut -c1-200
( timeout 1500 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_recovery.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r6_c14_tests.log 2>&1
cat gpurun_out/r6_c14_tests.log
