#!/usr/bin/env python3
"""Per-launch SQ counters of the hot kernels from ONE rocprofv3 --pmc pass (8 SQ slots):
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d out/sq -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline --graph
    python tools/pmc_sq.py out/sq/*/*counter_collection.csv > profiles/rNN_pmc_sq_counters.json
Median per launch; wave / wait / active counters are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import json
import sys


def main(path):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "(anonymous namespace)::" in k and "at::native" not in k:
            by[k.split("(anonymous namespace)::")[1].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "median per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles "
                   "(MI355X_MICROARCH.md)", "kernels": {}}
    for k, cs in sorted(by.items()):
        row = {"launches": max(len(v) for v in cs.values())}
        for c, v in sorted(cs.items()):
            v = sorted(v)
            row[c] = round(v[len(v) // 2])
        wc = row.get("SQ_WAVE_CYCLES", 0)
        if wc:
            row["wait_any_frac"] = round(row.get("SQ_WAIT_ANY", 0) / wc, 3)
            row["active_inst_frac"] = round(row.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in row:
                row["mfma_busy_per_wave_cycle"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 3)
        out["kernels"][k] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
