#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/ref2; mkdir -p $O
timeout 900 python tools/bench_policies.py > $O/policies_layer_step.jsonl 2>/dev/null
for w in 1 0; do echo -n "wide=$w "; CC_STEP_WIDE=$w timeout 300 python tools/ab_step.py heavy_hitter 2>/dev/null; done > $O/step_geometry.jsonl
for pol in recent_global l2 random; do for w in 1 0; do echo -n "wide=$w "; CC_STEP_WIDE=$w timeout 300 python tools/ab_step.py $pol 8:32:4096 8:32:2560 2>/dev/null; done; done >> $O/step_geometry.jsonl
timeout 900 python tools/run_configs.py > $O/configs_end_to_end.jsonl 2>/dev/null
cat $O/step_geometry.jsonl
