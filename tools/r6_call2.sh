#!/bin/bash
# r6 GPU call 2: is kernarg preload honoured?  traces with the K-request stamp; base / preload+late / preload without late
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
( tools/probes/preload_probe tools/probes/preload_kernel.co ) > gpurun_out/r6_c2_preload_probe.txt 2>&1
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L .ab/libcur.so
for v in base cur nolate; do cp .ab/lib$v.so $L; echo "== $v"; timeout 120 python tools/trace_one.py --S 4096 2>/dev/null | tail -1; timeout 120 python tools/trace_one.py --S 3488 --H 1 --HQ 8 2>/dev/null | tail -1; done > gpurun_out/r6_c2_trace.txt 2>&1
SH="8:32:4096 8:32:2560 4:16:4096 1:8:3488"
for r in 1 2 3; do for v in base cur nolate; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter $SH 2>/dev/null || echo "FAILED/timeout"; done; done > gpurun_out/r6_c2_ab.txt 2>&1
cp .ab/libcur.so $L
( timeout 900 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_e2e.py tests/test_gpu_quant.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r6_c2_tests.log 2>&1
