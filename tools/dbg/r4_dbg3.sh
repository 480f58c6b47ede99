#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
SEL="tests/test_gpu_fullsize.py -q -m gpu -k l2_prefill_to_decode -x"
cp $L /tmp/keep.so
for v in keep d0 keep d0 keep m0 p0 keep; do [ $v = keep ] && cp /tmp/keep.so $L || cp .ab/lib$v.so $L; echo -n "== $v: "; python tools/dbg/r4_dbg_run.py 1 $SEL 2>&1 | tail -n 1; done
cp /tmp/keep.so $L
