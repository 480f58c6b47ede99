#!/usr/bin/env python3
"""Per-launch HBM traffic of the hot-path kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE runs: together they exceed the 4 TCC counter slots of gfx950), corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is used as reported.  Counter values are KiB.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o f -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o w -- python bench.py --steps 4 --warmup 2 --no_cpu_baseline
    python tools/pmc_traffic.py out/fetch/f_counter_collection.csv out/write/w_counter_collection.csv > profiles/rNN_pmc_traffic.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] == counter and "(anonymous namespace)::" in k and "at::native" not in k:
            by[k.split("(anonymous namespace)::")[1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sorted(v) for k, v in by.items()}


def main(fetch_csv, write_csv):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {"unit": "bytes per launch (median over launches)", "fetch_correction": "x2 (gfx950 wide-read tally, MI355X_MICROARCH.md §HBM)",
           "kernels": {}}
    for k in sorted(f):
        fv, wv = f[k], w.get(k, [0.0])
        fetch = 2.0 * fv[len(fv) // 2] * 1024.0
        write = wv[len(wv) // 2] * 1024.0
        out["kernels"][k] = {"launches": len(fv), "fetch_bytes": round(fetch), "write_bytes": round(write),
                             "traffic_bytes": round(fetch + write), "fetch_size_raw_kib": fv[len(fv) // 2]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
