#!/bin/bash
# r3: full GPU suite + policy step times after the in-kernel random draws
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -6 $O/pytest.log
for p in random heavy_hitter recent_global l2; do timeout 300 python tools/ab_step.py $p 8:32:4096 2>/dev/null; done > $O/ab.log
cat $O/ab.log
timeout 300 python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log
echo done
