#!/bin/bash
# round-5 measured artefacts (run on the GPU box through gpurun; copy gpurun_out/ref/* to profiles/r05_* afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/ref_refresh.log 2>&1
O=gpurun_out/ref
# the decode loop on the QKV form of the step (one launch for norm + wqkv + RoPE + step), same box
CC_FUSE_QKV=1 timeout 600 python bench.py --no_cpu_baseline --no_live_pmc > $O/bench_fuse_qkv.json 2> $O/bench_fuse_qkv.err
# the overlap probe: fused vs twin vs its parts, and where a workgroup of the fused launch spends its time (trace build)
timeout 300 python tools/trace_qkv.py > $O/overlap_probe_S4096.json 2>/dev/null
timeout 300 python tools/trace_qkv.py --S 2560 > $O/overlap_probe_S2560.json 2>/dev/null
[ -f .ab/libqkvtrace.so ] && CC_LIB=.ab/libqkvtrace.so timeout 300 python tools/trace_qkv.py > $O/overlap_trace_S4096.json 2>/dev/null
[ -x tools/probes/stream_occ_probe ] && timeout 120 tools/probes/stream_occ_probe > $O/stream_occ_probe.txt 2>&1
timeout 900 python tools/bench_policies.py > $O/policies_layer_step.jsonl 2>$O/policies.err
timeout 900 python tools/run_configs.py > $O/configs_end_to_end.jsonl 2>$O/configs.err
timeout 300 python tools/bench_prefill.py > $O/bench_prefill.jsonl 2>/dev/null
timeout 300 python tools/sweep_step.py > $O/sweep_step.jsonl 2>/dev/null
timeout 200 python tools/trace_one.py --S 4096 > $O/single_launch_trace_S4096.json 2>/dev/null
ls -la $O
cat $O/bench.json | head -c 2500
