#!/bin/bash
# r6 GPU call 3: early-issued argument loads (EARLYARGS) vs late-only vs base; traces; recovery / e2e / quant tests on the new build
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L .ab/libcur.so
for v in base cur; do cp .ab/lib$v.so $L; echo "== $v"; timeout 120 python tools/trace_one.py --S 4096 2>/dev/null | tail -1; timeout 120 python tools/trace_one.py --S 4096 --H 1 --HQ 4 2>/dev/null | tail -1; done > gpurun_out/r6_c3_trace.txt 2>&1
SH="8:32:4096 8:32:2560 4:16:4096 2:8:4096 1:8:3488"
for r in 1 2 3; do for v in base cur noearly; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter $SH 2>/dev/null || echo "FAILED/timeout"; done; done > gpurun_out/r6_c3_ab.txt 2>&1
cp .ab/libcur.so $L
for pol in l2 recent_global; do for v in base cur; do cp .ab/lib$v.so $L; echo -n "$pol $v "; timeout 200 python tools/ab_step.py $pol 8:32:4096 1:8:3488 2>/dev/null || echo "FAILED/timeout"; done; done >> gpurun_out/r6_c3_ab.txt 2>&1
cp .ab/libcur.so $L
( timeout 1200 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_e2e.py tests/test_gpu_quant.py tests/test_gpu_fused_step.py tests/test_gpu_stress_small_grid.py tests/test_gpu_qkv_step.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r6_c3_tests.log 2>&1
