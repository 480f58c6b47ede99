#!/bin/bash
# r6 GPU call 23: few-head shapes, 8-wave (wide) against 4-wave workgroups
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( for r in 1 2 3; do for w in 1 0; do echo -n "wide=$w "; CC_STEP_WIDE=$w timeout 200 python tools/ab_step.py heavy_hitter 1:8:3488 1:4:4096 2:8:4096 4:16:4096 8:32:4096 2>/dev/null || echo FAILED; done; done ) > gpurun_out/r6_c23_wide_fewheads.txt 2>&1
cat gpurun_out/r6_c23_wide_fewheads.txt | cut -c1-220
