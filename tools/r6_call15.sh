#!/bin/bash
# r6 GPU call 15: the V rows' request DELAYED (s_sleep n x 64 cycles in front of it) against the product order, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
for r in 1 2 3; do for v in ${VARIANTS:-flat0 vd8 vd16 vd32}; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 1:8:3488 4:16:4096 2>/dev/null || echo FAILED; done; done > gpurun_out/r6_c15_vdelay_ab.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c15_vdelay_ab.txt | cut -c1-200
