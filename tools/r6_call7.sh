#!/bin/bash
# r6 GPU call 7: per-kernel durations and SQ counters of the two prefill forms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 0 1; do
  rm -rf /tmp/pf_prof_$k /tmp/pf_sq_$k
  CC_PREFILL_KSTAT=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_prof_$k -- python tools/bench_prefill.py --L 8192 --iters 10 > /tmp/pf_$k.log 2>&1
  echo "== CC_PREFILL_KSTAT=$k"
  head -12 $(ls /tmp/pf_prof_$k/*/*kernel_stats.csv | head -1) | cut -c1-230
  CC_PREFILL_KSTAT=$k rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pf_sq_$k -- python tools/bench_prefill.py --L 8192 --iters 3 > /dev/null 2>&1
  python tools/pmc_sq.py $(ls /tmp/pf_sq_$k/*/*counter_collection.csv | head -1) | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items():
    if 'prefill' in k or 'vt_perm' in k: print(k[:70], {a:b for a,b in v.items() if a in ('launches','wait_any_frac','active_inst_frac','mfma_busy_per_wave_cycle','SQ_LDS_BANK_CONFLICT','SQ_WAVE_CYCLES','SQ_INSTS_VALU')})
"
done > gpurun_out/r6_c7_prefill_kernels.txt 2>&1
