#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in a `hipcc -Rpass-analysis=kernel-resource-usage` log (stderr of the compile).

    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o x.o 2> x.log ; python tools/kernel_resources.py x.log [substring]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        if want in d:
            d = re.sub(r"^void \(anonymous namespace\)::", "", d)
            sc, lds, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
            print(f"{d[:110]:110s} vgpr {g('VGPRs'):3d} agpr {g('AGPRs'):3d} sgpr {g('SGPRs'):3d} scratch {sc} lds {lds} occ {occ}")


if __name__ == "__main__":
    main()
