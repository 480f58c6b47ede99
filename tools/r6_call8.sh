#!/bin/bash
# r6 GPU call 8: the K-stationary side-sum pass with DMA-staged Q tiles — parity, A/B, per-kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_hybrid.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r6_c8_tests.log 2>&1
( for k in 0 1; do echo "CC_PREFILL_KSTAT=$k"; CC_PREFILL_KSTAT=$k timeout 300 python tools/bench_prefill.py 2>/dev/null; done ) > gpurun_out/r6_c8_bench_prefill.txt 2>&1
rm -rf /tmp/pf_prof_1
CC_PREFILL_KSTAT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_prof_1 -- python tools/bench_prefill.py --L 8192 --iters 10 > /tmp/pf_1.log 2>&1
head -6 $(ls /tmp/pf_prof_1/*/*kernel_stats.csv | head -1) | cut -c1-200 >> gpurun_out/r6_c8_bench_prefill.txt
