cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pf
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_hybrid.py tests/test_gpu_e2e.py tests/test_gpu_properties.py tests/test_gpu_extremes.py -x -q -m gpu 2>&1 | tail -5
python tools/bench_prefill.py --bands > gpurun_out/pf/bench_prefill.jsonl 2>&1; cat gpurun_out/pf/bench_prefill.jsonl
rm -rf /tmp/pfp; rocprofv3 --kernel-trace --stats -d /tmp/pfp -- python tools/bench_prefill.py --L 8192 --iters 5 > /dev/null 2>&1
python tools/prof_db.py $(ls /tmp/pfp/*/*.db | head -1) prefill_ vt_perm > gpurun_out/pf/prefill_kernel_stats.txt 2>&1; cat gpurun_out/pf/prefill_kernel_stats.txt
