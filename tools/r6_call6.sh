#!/bin/bash
# r6 GPU call 6: the K-stationary side-sum pass — parity tests that reach the prefill path, then the A/B of the two forms
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_hybrid.py tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_baseline_sizes.py tests/test_gpu_recovery.py tests/test_gpu_fused_step.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r6_c6_tests.log 2>&1
( for k in 0 1; do echo "CC_PREFILL_KSTAT=$k"; CC_PREFILL_KSTAT=$k timeout 300 python tools/bench_prefill.py 2>/dev/null; CC_PREFILL_KSTAT=$k timeout 300 python tools/bench_prefill.py --bands --L 16384 2>/dev/null; done ) > gpurun_out/r6_c6_bench_prefill.txt 2>&1
