"""Small-grid stress of the fuzz families (tools/fuzz_step.py: small_grid_stress): 8 kv heads, one or two splits per head.
    python tools/dbg/stress_small_grid.py <step|ring|hybrid|quant> N [strategy]
(the driver-run form: tests/test_gpu_stress_small_grid.py)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fuzz_step as F  # noqa: E402

fam, n = sys.argv[1], int(sys.argv[2])
ran, bad = F.small_grid_stress(fam, n, sys.argv[3] if len(sys.argv) > 3 else None)
for b in bad:
    print("MISMATCH", b, flush=True)
print(f"small-grid stress {fam}: {ran} ran, {len(bad)} mismatches")
