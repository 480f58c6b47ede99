#!/bin/bash
# r6 GPU call 1: the GPU suite on the preload build, then the same-box A/B of the prologue / few-head variants, then traces.
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r6_c1_gputest.log 2>&1
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L .ab/libcur.so
SH="8:32:4096 8:32:2560 4:16:4096 2:8:4096 1:8:3488 1:4:4096"
for r in 1 2 3; do for v in base cur pv1 px pxv; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter $SH 2>/dev/null || echo "FAILED/timeout"; done; done > gpurun_out/r6_c1_ab.txt 2>&1
cp .ab/libcur.so $L
for v in base cur; do cp .ab/lib$v.so $L; echo "== $v"; timeout 120 python tools/trace_one.py --S 4096 2>&1 | tail -2; timeout 120 python tools/trace_one.py --S 3488 --H 1 --HQ 4 2>&1 | tail -2; done > gpurun_out/r6_c1_trace.txt 2>&1
cp .ab/libcur.so $L
( timeout 600 python bench.py --steps 64 --warmup 8 2>&1 | tail -3 ) > gpurun_out/r6_c1_bench.log 2>&1
