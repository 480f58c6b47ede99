#!/bin/bash
# same-box A/B of runtime knobs on the heavy-hitter layer step (graph replay)
export TMPDIR=/tmp
for r in 1 2; do
for knob in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "GPU_MAX_HW_QUEUES=1" "HSA_NO_SCRATCH_RECLAIM=1"; do
  echo -n "$knob  "
  env $knob timeout 200 python tools/ab_step.py heavy_hitter 8:32:4096 1:8:3488 2>/dev/null
done
done
