#!/bin/bash
# r6 GPU call 4: the first-byte probe; what the recoverable hand-off costs (norc); the whole GPU suite on the default build
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
( timeout 120 tools/probes/first_byte_probe ) > gpurun_out/r6_c4_first_byte_probe.txt 2>&1
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L .ab/libcur.so
SH="8:32:4096 8:32:2560 1:8:3488"
for r in 1 2 3; do for v in cur norc; do cp .ab/lib$v.so $L; echo -n "$v "; timeout 200 python tools/ab_step.py heavy_hitter $SH 2>/dev/null || echo "FAILED/timeout"; done; done > gpurun_out/r6_c4_ab.txt 2>&1
cp .ab/libcur.so $L
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -60 ) > gpurun_out/r6_c4_gputest.log 2>&1
