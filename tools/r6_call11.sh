#!/bin/bash
# r6 GPU call 11: the l2 step's carried norm record (CC_V_L2CARRY=1) — parity of every l2 test + the step's time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fused_step.py -m gpu -x -q -k "l2 or share_the_workspace" 2>&1 | tail -15 ) > gpurun_out/r6_c11_l2_tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_recovery.py -m gpu -x -q -k "l2 or strict_subset" 2>&1 | tail -8 ) >> gpurun_out/r6_c11_l2_tests.log 2>&1
( timeout 600 python tools/bench_policies.py 2>/dev/null | grep -i "l2\|heavy" ) > gpurun_out/r6_c11_policies.txt 2>&1
tail -5 gpurun_out/r6_c11_l2_tests.log; cat gpurun_out/r6_c11_policies.txt | cut -c1-300
