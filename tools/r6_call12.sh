#!/bin/bash
# r6 GPU call 12: l2 step, carried record (l2c) against the r5 exchange (l2x): traces + alternating timings on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2 3; do for v in ${VARIANTS:-l2x l2c}; do cp .ab/lib$v.so $L; echo "== $v"; timeout 200 python tools/bench_policies.py 2>/dev/null | grep '"l2"' | cut -c1-140; done; done > gpurun_out/r6_c12_l2_ab.txt 2>&1
for v in ${VARIANTS:-l2x l2c}; do cp .ab/lib$v.so $L; echo "== $v"; timeout 200 python tools/trace_one.py --policy l2 2>&1 | tail -2; timeout 200 python tools/trace_one.py --policy l2 --H 1 --HQ 8 --S 3488 2>&1 | tail -1; done > gpurun_out/r6_c12_l2_trace.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c12_l2_ab.txt; cat gpurun_out/r6_c12_l2_trace.txt | cut -c1-1800
