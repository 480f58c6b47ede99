"""Small-grid stress of the fuzz families: the fuzz functions of tools/fuzz_step.py with the head count forced to 8 and the cache
length forced small (one or two splits per head: 8-16 workgroups — where late-waking XCDs expose cross-workgroup ordering holes)."""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fuzz_step as F


class Forced:
    """choice() call number -> forced value (None: S, drawn small); everything else random."""
    def __init__(self, seed, plan):
        self.r, self.plan, self.n = random.Random(seed), plan, 0

    def choice(self, seq):
        i, self.n = self.n, self.n + 1
        if i in self.plan:
            v = self.plan[i]
            return self.r.randint(70, 190) if v is None else v
        return self.r.choice(seq)

    def __getattr__(self, name):
        return getattr(self.r, name)


FAMILIES = {
    # family: (function, {index of the choice() call: forced value})
    "step": (F.one, {3: 8, 7: None}),            # strategy, dtype, D, H, R, [two nested choices], S
    "ring": (F.one_ring, {3: 8, 5: None}),       # strategy, dtype, D, H, R, S
    "hybrid": (F.one_hybrid_step, {1: 8, 4: None}),  # dtype, H, R, [nested], S
    "quant": (F.one_quant, {2: 8, 5: None}),     # strategy, dtype, H, R, [nested], S
}
fam, n = sys.argv[1], int(sys.argv[2])
fn, plan = FAMILIES[fam]
bad = ran = 0
for i in range(n):
    r = fn(Forced(7000 + i, plan), i)
    if r is None:
        continue
    ran += 1
    if r:
        bad += 1
        print("MISMATCH", r, flush=True)
print(f"small-grid stress {fam}: {ran} ran, {bad} mismatches")
