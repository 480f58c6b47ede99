#!/bin/bash
# Time library variants alternately on ONE box (boxes differ by +-3 %, more than most single changes to the step):
#   gpurun -- 'tools/ab_run.sh "base exp" "C3:heavy_hitter C3:recent_global" 2'
# variants: names of .ab/lib<name>.so (tools/ab_variant.sh; `cp cold_compress_amd/csrc/libcoldcompress_hip.so .ab/libbase.so` for
# the current build); cases: CC_POLICIES_ONLY values of tools/bench_policies.py.  Prints fused-step us (and the uint8 step's).
# Leaves the LAST variant installed as the package's library: rebuild (python -c "import __graft_entry__ as g; g.build()") afterwards.
L=cold_compress_amd/csrc/libcoldcompress_hip.so
for r in $(seq 1 "$3"); do for v in $1; do cp ".ab/lib$v.so" $L; for c in $2; do echo -n "$v $c "; CC_POLICIES_ONLY=$c timeout 100 python tools/bench_policies.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['fused_step_us'], d.get('fused_quant8_step_us'))"; done; done; done
