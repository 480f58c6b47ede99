# Regenerates the measured artefacts under profiles/ for the current round (run on the GPU box through gpurun; copy
# gpurun_out/ref/* to profiles/rNN_* afterwards).  PMC counters are collected in their own passes, never with a trace domain
# other than --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ref
python bench.py > gpurun_out/ref/bench.json 2> gpurun_out/ref/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ref/prof -- python bench.py --steps 32 --warmup 4 --no_cpu_baseline --graph > gpurun_out/ref/prof_bench.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/ref/fetch -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/ref/write -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/ref/sq -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline --graph > /dev/null 2>&1
T=$(ls gpurun_out/ref/prof/*/*kernel_trace.csv | head -1)
python tools/summarize_prof.py $T > gpurun_out/ref/summary.txt
cp $(ls gpurun_out/ref/prof/*/*kernel_stats.csv | head -1) gpurun_out/ref/kernel_stats.csv
F=$(ls gpurun_out/ref/fetch/*/*counter_collection.csv | head -1); W=$(ls gpurun_out/ref/write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W > gpurun_out/ref/pmc_traffic.json
python tools/pmc_sq.py $(ls gpurun_out/ref/sq/*/*counter_collection.csv | head -1) > gpurun_out/ref/pmc_sq_counters.json
rm -rf gpurun_out/ref/prof gpurun_out/ref/fetch gpurun_out/ref/write gpurun_out/ref/sq
cat gpurun_out/ref/bench.json
