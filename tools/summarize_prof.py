#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel duration stats for the hot-path / glue kernels, a steady-state
per-token breakdown of the decode loop, and one layer in launch order.  Usage: summarize_prof.py <kernel_trace.csv>"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])  # noqa: E731
    out = [f"# {path}: {len(rows)} kernel dispatches",
           "## hot-path + glue kernels (ns): name  calls  mean  median  min  max"]
    by = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "(anonymous namespace)::" in n and "at::native" not in n:
            by[n.split("(anonymous namespace)::")[1].split("(")[0]].append(dur(r))
    for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        d = sorted(d)
        out.append(f"{n:64s} {len(d):6d} {sum(d) / len(d):11.0f} {d[len(d) // 2]:9d} {d[0]:9d} {d[-1]:11d}")
    # decode tokens are delimited by the streaming pass of layer 0: 32 of them per token
    idx = [i for i, r in enumerate(rows) if "decode_attn_split" in r["Kernel_Name"]]
    n_layer = 32
    if len(idx) >= n_layer * 22:
        a, b = idx[n_layer * 20], idx[n_layer * 21]
        tok = rows[a:b]
        wall = (int(rows[b]["Start_Timestamp"]) - int(tok[0]["Start_Timestamp"])) / 1e3
        out.append(f"## one steady-state decode token (under the profiler): {len(tok)} kernels, wall {wall:.1f} us")
        agg = collections.defaultdict(lambda: [0, 0])
        for r in tok:
            k = r["Kernel_Name"][:110]
            agg[k][0] += dur(r)
            agg[k][1] += 1
        for k, v in sorted(agg.items(), key=lambda x: -x[1][0])[:16]:
            out.append(f"{v[0] / 1e3:9.1f} us  n={v[1]:4d}  avg={v[0] / v[1] / 1e3:7.2f} us  {k}")
        out.append("## one layer, in launch order")
        i0, i1 = idx[n_layer * 20 + 5], idx[n_layer * 20 + 6]
        for r in rows[i0 - 2:i1 - 2]:
            out.append(f"{dur(r) / 1e3:7.2f} us  {r['Kernel_Name'][:110]}")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
