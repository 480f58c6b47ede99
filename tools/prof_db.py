#!/usr/bin/env python3
"""Per-kernel duration stats from a rocprofv3 rocpd SQLite database (the default output of ROCm 7.2's
`rocprofv3 --kernel-trace --stats`).  Usage: prof_db.py <results.db> [name-substring ...]"""
import sqlite3
import sys


def main(path, pats):
    c = sqlite3.connect(path).cursor()
    where = " or ".join("name like ?" for _ in pats) or "1"
    rows = c.execute(f"select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                     f"max(vgpr_count), max(lds_size), max(grid_x), max(grid_y), max(grid_z), max(workgroup_x) from kernels where {where} "
                     f"group by name order by 6 desc limit 40", [f"%{p}%" for p in pats]).fetchall()
    print("# calls  mean_ns  min_ns  max_ns  total_ms  vgpr  lds  grid(threads)  wg  name")
    for r in rows:
        print(f"{r[1]:6d} {r[2]:10.0f} {r[3]:8d} {r[4]:9d} {r[5]/1e6:9.2f} {r[6]:5d} {r[7]:6d} {r[8]}x{r[9]}x{r[10]} {r[11]:4d}  {r[0][:120]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
