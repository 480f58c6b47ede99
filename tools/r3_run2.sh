#!/bin/bash
export TMPDIR=/tmp
L=cold_compress_amd/csrc/libcoldcompress_hip.so
O=gpurun_out/r3b
mkdir -p $O
cp .ab/libeml2.so $L
timeout 900 python -m pytest tests -q -m gpu --maxfail=12 > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -8 $O/pytest.log
for r in 1 2; do for v in eml eml2; do cp .ab/lib$v.so $L; for w in 0 1; do echo -n "$v "; CC_STEP_WIDE=$w timeout 300 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 2:8:4096 1:8:3488 2>/dev/null; done; done; done > $O/ab.log
cat $O/ab.log
cp .ab/libeml2.so $L
timeout 200 python tools/trace_one.py --wide 1 > $O/trace_wide.json 2>$O/trace_wide.err
timeout 200 python tools/trace_one.py --wide 0 > $O/trace_narrow.json 2>$O/trace_narrow.err
for pol in recent_global l2 random; do echo -n "$pol "; CC_STEP_WIDE=1 timeout 300 python tools/ab_step.py $pol 8:32:4096 8:32:2560 2>/dev/null; echo -n "$pol "; CC_STEP_WIDE=0 timeout 300 python tools/ab_step.py $pol 8:32:4096 2>/dev/null; done > $O/ab_pol.log
cat $O/ab_pol.log
echo done
