#!/bin/bash
# run a test selection against a library variant: tools/dbg/r4_test.sh VARIANT "pytest args"
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
cp .ab/lib$1.so $L
timeout ${3:-900} python -m pytest $2 -q -m gpu -x 2>&1 | tail -n 15 | tee gpurun_out/r4_test_$1.txt
cp /tmp/keep.so $L
