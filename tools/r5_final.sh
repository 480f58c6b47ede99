#!/bin/bash
# round-5 closing run on the GPU box (through gpurun): the whole GPU suite, the fixture-driven GPU tests AGAIN on reference-made
# vectors from other seeds (.ab/golden_s1: `python oracle/gen_golden.py --seed_offset 1000 --out .ab/golden_s1` in the build
# container), smoke, and the bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?" > $O/rc.txt
if [ -d .ab/golden_s1 ]; then
  CC_GOLDEN_DIR=$PWD/.ab/golden_s1 timeout 400 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_hybrid.py \
    tests/test_gpu_quant.py tests/test_gpu_quant_fused.py tests/test_hh_ring.py tests/test_hh_query_fixture.py tests/test_window_sums.py \
    tests/test_gpu_fused_step.py > $O/gputest_fresh_seeds.log 2>&1; echo "fresh-seed rc=$?" >> $O/rc.txt
fi
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/gputest.log; tail -3 $O/gputest_fresh_seeds.log; head -c 600 $O/bench.json
