#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel duration stats for the hot-path kernels and a
steady-state per-token breakdown of the decode loop.  Usage: summarize_prof.py <kernel_trace.csv>"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])  # noqa: E731
    print(f"# {path}: {len(rows)} kernel dispatches")
    print("## hot-path kernels (ns): name  calls  mean  median  min  max  grid(threads)  workgroup  vgpr  lds")
    by = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "(anonymous namespace)::" in n and "at::native" not in n:
            by[n.split("(anonymous namespace)::")[1].split("(")[0]].append(r)
    for n, rs in sorted(by.items(), key=lambda kv: -sum(dur(r) for r in kv[1])):
        d = sorted(dur(r) for r in rs)
        r0 = rs[len(rs) // 2]
        print(f"{n:60s} {len(d):6d} {sum(d)/len(d):12.0f} {d[len(d)//2]:10d} {d[0]:10d} {d[-1]:12d}  "
              f"{r0.get('Grid_Size_X','?')}x{r0.get('Grid_Size_Y','?')}x{r0.get('Grid_Size_Z','?')} "
              f"{r0.get('Workgroup_Size_X','?')} {r0.get('VGPR_Count', r0.get('Arch_VGPR_Count','?'))} {r0.get('LDS_Block_Size','?')}")
    upd = [i for i, r in enumerate(rows) if "decode_update_kernel" in r["Kernel_Name"]]
    if len(upd) >= 32 * 12:
        a, b = upd[32 * 10], upd[32 * 11]
        tok = rows[a:b]
        wall = (int(rows[b]["Start_Timestamp"]) - int(tok[0]["Start_Timestamp"])) / 1e3
        agg = collections.defaultdict(lambda: [0, 0])
        for r in tok:
            k = r["Kernel_Name"][:100]
            agg[k][0] += dur(r)
            agg[k][1] += 1
        tot = sum(v[0] for v in agg.values()) / 1e3
        print(f"## one steady-state decode token (under the profiler): {len(tok)} kernels, wall {wall:.1f} us, "
              f"sum of kernel durations {tot:.1f} us")
        for k, v in sorted(agg.items(), key=lambda x: -x[1][0])[:18]:
            print(f"{v[0]/1e3:9.1f} us  n={v[1]:4d}  avg={v[0]/v[1]/1e3:7.2f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
