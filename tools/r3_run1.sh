#!/bin/bash
# round-3 GPU run 1: parity of the early-(m, l) / wide forms, same-box A/B, traces
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
O=gpurun_out/r3a
mkdir -p $O
cp .ab/libeml.so $L
timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_e2e.py -q -m gpu --maxfail=12 > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
for r in 1 2; do for v in noeml eml; do cp .ab/lib$v.so $L; for w in 0 1; do echo -n "$v "; CC_STEP_WIDE=$w timeout 300 python tools/ab_step.py 2>/dev/null; done; done; done > $O/ab.log
cat $O/ab.log
cp .ab/libeml.so $L
timeout 200 python tools/trace_one.py --wide 1 > $O/trace_wide.json 2>$O/trace_wide.err
timeout 200 python tools/trace_one.py --wide 0 > $O/trace_narrow.json 2>$O/trace_narrow.err
cp .ab/libnoeml.so $L
timeout 200 python tools/trace_one.py --wide 0 > $O/trace_noeml_narrow.json 2>$O/trace_noeml_narrow.err
cp .ab/libeml.so $L
echo done
