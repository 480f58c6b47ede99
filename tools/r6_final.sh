#!/bin/bash
# round-6 closing run on the GPU box (through gpurun): the whole GPU suite, smoke, the bench line with its rocprofv3 / PMC companions,
# the per-policy / per-config / prefill / sweep / trace artefacts (copy gpurun_out/final/* to profiles/r06_* afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?" > $O/rc.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20.err; echo "bench20 rc=$?" >> $O/rc.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 32 --warmup 4 --no_cpu_baseline --graph --no_long_window > $O/prof_bench.json 2>/dev/null
T=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python tools/summarize_prof.py $T > $O/decode_kernel_summary.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 4 --warmup 2 --settle 0 --no_cpu_baseline --no_long_window > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 4 --warmup 2 --settle 0 --no_cpu_baseline --no_long_window > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq -- python bench.py --steps 4 --warmup 2 --settle 0 --no_cpu_baseline --graph --no_long_window > /dev/null 2>&1
F=$(ls $O/fetch/*/*counter_collection.csv | head -1); W=$(ls $O/write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W > $O/pmc_traffic.json
python tools/pmc_sq.py $(ls $O/sq/*/*counter_collection.csv | head -1) > $O/pmc_sq_counters.json
rm -rf $O/prof $O/fetch $O/write $O/sq
timeout 900 python tools/bench_policies.py > $O/policies_layer_step.jsonl 2>$O/policies.err
timeout 900 python tools/run_configs.py > $O/configs_end_to_end.jsonl 2>$O/configs.err
timeout 300 python tools/bench_prefill.py > $O/bench_prefill.jsonl 2>/dev/null
timeout 300 python tools/sweep_step.py > $O/sweep_step.jsonl 2>/dev/null
timeout 200 python tools/trace_one.py --S 4096 > $O/single_launch_trace_S4096.json 2>/dev/null
timeout 200 python tools/trace_one.py --S 4096 --H 1 --HQ 4 > $O/single_launch_trace_S4096_H1.json 2>/dev/null
cat $O/rc.txt; tail -3 $O/gputest.log; head -c 700 $O/bench.json
