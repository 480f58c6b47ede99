#!/bin/bash
# r6 GPU call 25: the N > 1 control flow of bench.py once more on ONE GPU (dry run: gloo, collectives staged through the host; not a measurement)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CC_BENCH_DRYRUN_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --settle 8 --no_cpu_baseline --no_long_window > gpurun_out/r6_c25_dry2.json 2> gpurun_out/r6_c25_dry2.err
echo "rc=$?"; tail -5 gpurun_out/r6_c25_dry2.err | cut -c1-300; head -c 900 gpurun_out/r6_c25_dry2.json
