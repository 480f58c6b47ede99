#!/bin/bash
# round-4 measured artefacts (run on the GPU box through gpurun; copy gpurun_out/ref/* to profiles/r04_* afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/ref_refresh.log 2>&1
O=gpurun_out/ref
timeout 900 python tools/bench_policies.py > $O/policies_layer_step.jsonl 2>$O/policies.err
timeout 900 python tools/run_configs.py > $O/configs_end_to_end.jsonl 2>$O/configs.err
timeout 300 python tools/bench_prefill.py > $O/bench_prefill.jsonl 2>/dev/null
timeout 300 python tools/sweep_step.py > $O/sweep_step.jsonl 2>/dev/null
timeout 200 python tools/trace_one.py --S 4096 > $O/single_launch_trace_S4096.json 2>/dev/null
timeout 200 python tools/trace_one.py --S 18432 > $O/single_launch_trace_S18432.json 2>/dev/null
timeout 200 python tools/trace_one.py --S 4096 --policy l2 > $O/single_launch_trace_S4096_l2.json 2>/dev/null
timeout 200 python tools/trace_one.py --H 1 --HQ 4 --S 4096 > $O/single_launch_trace_S4096_H1.json 2>/dev/null
ls -la $O
cat $O/bench.json | head -c 1500
