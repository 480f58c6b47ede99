#!/bin/bash
# generic same-box A/B: tools/dbg/ab_gen.sh "v1 v2 ..." "policy shapes..." [test-selector]
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
V="$1"; ARGS="${2:-heavy_hitter 8:32:4096 8:32:2560 1:8:3488}"
for r in 1 2 3; do for v in $V; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py $ARGS 2>/dev/null || echo "FAILED/timeout"; done; done
if [ -n "$3" ]; then LAST=$(echo $V | awk '{print $NF}'); cp .ab/lib$LAST.so $L; echo "== $LAST tests"; timeout 900 python -m pytest $3 -q -m gpu 2>&1 | tail -n 4; fi
cp /tmp/keep.so $L
