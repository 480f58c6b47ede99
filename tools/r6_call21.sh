#!/bin/bash
# r6 GPU call 21: the l2 record checked against the state — product build and the CC_V_L2CARRY=1 build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
( echo "== product"; timeout 600 python -m pytest tests/test_gpu_fused_step.py -m gpu -x -q -k "l2" 2>&1 | tail -4
cp .ab/libl2c.so $L; echo "== CC_V_L2CARRY=1"; timeout 600 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_recovery.py -m gpu -x -q -k "l2 or strict_subset" 2>&1 | tail -4 ) > gpurun_out/r6_c21_l2_record.log 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c21_l2_record.log
