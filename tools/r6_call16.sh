#!/bin/bash
# r6 GPU call 16: the PMC passes of the closing run again (they left empty files), stderr kept
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
rm -rf $O/fetch $O/write $O/sq
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 4 --warmup 2 --settle 0 --no_cpu_baseline --no_long_window > $O/pmc_fetch.out 2> $O/pmc_fetch.err; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 4 --warmup 2 --settle 0 --no_cpu_baseline --no_long_window > $O/pmc_write.out 2> $O/pmc_write.err; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq -- python bench.py --steps 4 --warmup 2 --settle 0 --no_cpu_baseline --graph --no_long_window > $O/pmc_sq.out 2> $O/pmc_sq.err; echo "sq rc=$?"
ls $O/fetch/* $O/write/* $O/sq/* 2>&1 | head; tail -3 $O/pmc_fetch.err
F=$(ls $O/fetch/*/*counter_collection.csv | head -1); W=$(ls $O/write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W > $O/pmc_traffic.json
python tools/pmc_sq.py $(ls $O/sq/*/*counter_collection.csv | head -1) > $O/pmc_sq_counters.json
rm -rf $O/fetch $O/write $O/sq
head -c 600 $O/pmc_traffic.json; head -c 400 $O/pmc_sq_counters.json
