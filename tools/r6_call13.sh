#!/bin/bash
# r6 GPU call 13: REAL-step traces (positions advance) of the heavy-hitter and the l2 step (r5 exchange), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
cp .ab/libl2x.so $L
( echo "== hh"; timeout 200 python tools/trace_one.py --advance 2>&1 | tail -1; echo "== l2x"; timeout 200 python tools/trace_one.py --policy l2 --advance 2>&1 | tail -1 ) > gpurun_out/r6_c13_real_traces.txt 2>&1
cp /tmp/keep.so $L
python3 - <<'PY'
import json
for line in open('gpurun_out/r6_c13_real_traces.txt'):
    if line.startswith('=='): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception: print(line[:300]); continue
    print(d['us_per_launch_events'], d['launch_span_us'], d['gap_to_next_launch_us'])
    for k,v in d['since_own_start_us_min_mean_max'].items(): print('  ',k, v)
    print('  tail', d['head_end_minus_last_merge_barrier_us'], 'mb', d['head_last_merge_barrier_us'])
PY
