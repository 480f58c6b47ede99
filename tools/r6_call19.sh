#!/bin/bash
# r6 GPU call 19: runtime knobs against the launch boundary (the gap between dependent kernel nodes of a hipGraph): step time per setting
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo -n "$1 :: "; env $1 timeout 200 python tools/ab_step.py heavy_hitter 8:32:4096 1:8:3488 2>/dev/null | cut -c40-200 || echo FAILED; }
( for r in 1 2; do
run "CC_NOP=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "HIP_FORCE_DEV_KERNARG=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_INTERRUPT=0"
run "AMD_DIRECT_DISPATCH=0"
run "DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_GRAPH_NO_FENCE=1"
run "HSA_DISABLE_CACHE=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0"
run "ROC_AQL_QUEUE_SIZE=16384"
done ) > gpurun_out/r6_c19_runtime_knobs.txt 2>&1
cat gpurun_out/r6_c19_runtime_knobs.txt
