#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
SEL="tests/test_gpu_fullsize.py -q -m gpu -k l2_prefill_to_decode -x"
echo "== product, l2 handoff off"; python tools/dbg/r4_dbg_run.py 0 $SEL 2>&1 | tail -n 3
cp $L /tmp/keep.so
for v in p0 m0 d0; do cp .ab/lib$v.so $L; echo "== $v"; python tools/dbg/r4_dbg_run.py 1 $SEL 2>&1 | tail -n 3; done
cp /tmp/keep.so $L
