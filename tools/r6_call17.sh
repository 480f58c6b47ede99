#!/bin/bash
# r6 GPU call 17: merge factors kept in LDS for the partial-O publish (wf1) against recomputing them (wf0), alternating; parity subset
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2 3 4; do for v in wf0 wf1; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 1:8:3488 4:16:4096 2>/dev/null || echo FAILED; done; done > gpurun_out/r6_c17_wfact_ab.txt 2>&1
for v in wf0 wf1; do cp .ab/lib$v.so $L; for pol in recent_global l2; do echo -n "$v "; timeout 200 python tools/ab_step.py $pol 8:32:4096 1:8:3488 2>/dev/null || echo FAILED; done; done >> gpurun_out/r6_c17_wfact_ab.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c17_wfact_ab.txt | cut -c1-200
( timeout 1500 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_recovery.py tests/test_gpu_parity.py tests/test_gpu_quant_fused.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r6_c17_tests.log 2>&1
cat gpurun_out/r6_c17_tests.log
