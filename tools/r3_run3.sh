#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
O=gpurun_out/r3c
mkdir -p $O
V="${1:-eml2 eml3}"
LAST=$(echo $V | awk '{print $NF}')
cp .ab/lib$LAST.so $L
timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_hh_query_fixture.py tests/test_gpu_quant_fused.py tests/test_gpu_fullsize.py -q -m gpu --maxfail=12 > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
for r in 1 2; do for v in $V; do cp .ab/lib$v.so $L; for w in 0 1; do echo -n "$v "; CC_STEP_WIDE=$w timeout 300 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 2:8:4096 1:8:3488 2>/dev/null; done; done; done > $O/ab.log
cat $O/ab.log
cp .ab/lib$LAST.so $L
timeout 200 python tools/trace_one.py --wide 1 > $O/trace_wide.json 2>$O/trace_wide.err
timeout 200 python tools/trace_one.py --wide 0 > $O/trace_narrow.json 2>$O/trace_narrow.err
echo done
