#!/bin/bash
# r6 GPU call 22: the GPU suite, smoke and the bench line on the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final2
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?" > $O/rc.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -2 $O/gputest.log; head -c 300 $O/bench.json; echo; head -c 300 $O/bench_20steps.json
