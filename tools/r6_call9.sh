#!/bin/bash
# r6 GPU call 9: the woven side-sum pass — A/B of the weave density, parity of the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L .ab/libpf18.so
( for v in pf18 pf0 pf10; do cp .ab/lib$v.so $L; echo "== $v (CC_PREFILL_KSTAT=1)"; timeout 300 python tools/bench_prefill.py --L 8192 2>/dev/null | head -1; done; cp .ab/libpf18.so $L; echo "== two-pass (CC_PREFILL_KSTAT=0)"; CC_PREFILL_KSTAT=0 timeout 300 python tools/bench_prefill.py --L 8192 2>/dev/null | head -1 ) > gpurun_out/r6_c9_bench_prefill.txt 2>&1
cp .ab/libpf18.so $L
rm -rf /tmp/pf_prof_1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_prof_1 -- python tools/bench_prefill.py --L 8192 --iters 10 > /tmp/pf_1.log 2>&1
head -5 $(ls /tmp/pf_prof_1/*/*kernel_stats.csv | head -1) | cut -c1-200 >> gpurun_out/r6_c9_bench_prefill.txt
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_hybrid.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r6_c9_tests.log 2>&1
