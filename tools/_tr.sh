python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
python tools/bench_policies.py 2>&1 | grep "C4" | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/hp; CC_POLICIES_ONLY=C4:hybrid rocprofv3 --kernel-trace --stats -d /tmp/hp -- python tools/bench_policies.py > /dev/null 2>&1
python tools/prof_db.py $(ls /tmp/hp/*/*.db | head -1) decode_attn hybrid 2>&1 | cut -c1-200
