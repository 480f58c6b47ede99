#!/bin/bash
# r6 GPU call 10: the side-sum pass as fenced slices — parity + timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( echo "== K-stationary (CC_PREFILL_KSTAT=1), fenced slices"; CC_PREFILL_KSTAT=1 timeout 300 python tools/bench_prefill.py 2>/dev/null; echo "== two-pass"; timeout 300 python tools/bench_prefill.py --L 8192 2>/dev/null | head -1 ) > gpurun_out/r6_c10_bench_prefill.txt 2>&1
rm -rf /tmp/pf_prof_1
CC_PREFILL_KSTAT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_prof_1 -- python tools/bench_prefill.py --L 8192 --iters 10 > /tmp/pf_1.log 2>&1
head -5 $(ls /tmp/pf_prof_1/*/*kernel_stats.csv | head -1) | cut -c1-200 >> gpurun_out/r6_c10_bench_prefill.txt
( CC_PREFILL_KSTAT=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_hybrid.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r6_c10_tests.log 2>&1
