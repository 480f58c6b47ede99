#!/bin/bash
# r6 GPU call 18: prefill kernels compiled for 3 waves per SIMD (168 VGPRs, spills) against the product's 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2; do for v in pf22 pf33 pf23 pf32; do cp .ab/lib$v.so $L; echo "== $v"; timeout 200 python tools/bench_prefill.py --L 8192 2>/dev/null | cut -c1-150; done; done > gpurun_out/r6_c18_prefill_occ.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c18_prefill_occ.txt
