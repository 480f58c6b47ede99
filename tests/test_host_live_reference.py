"""The host-side budget arithmetic against the reference LIVE, on random inputs, where /root/reference exists (the
build container; skipped elsewhere): normalize_cache_length, apply_pyramid_pattern (PyramidKV budgets, increasing and decreasing),
apply_pattern (tile / repeat / pyramid / funnel) and find_multiple — same value, or the same exception type on both sides.  The reference runs
in a child process (its top-level module names stay out of this one); tests/golden/f8_budgets.json pins the fixed table."""
import json
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import contextlib, io, json, sys
sys.path.insert(0, sys.argv[1])
import gen_golden
A, C, G, M, P = gen_golden._import_reference()
out = []
for fn, args in json.load(sys.stdin):
    f = {"normalize": G.normalize_cache_length, "pyramid": lambda a, b, c, d: G.apply_pyramid_pattern(a, b, c, decreasing=d),
         "pattern": lambda p, n, s, m: G.apply_pattern(p, n, s, m), "find_multiple": M.find_multiple}[fn]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out.append(f(*args))
    except Exception as e:
        out.append(type(e).__name__)
print(json.dumps(out))
"""


def _ours(fn, args):
    import contextlib
    import io

    from cold_compress_amd.harness import generation as G
    from cold_compress_amd.harness.model import find_multiple

    f = {"normalize": G.normalize_cache_length, "pyramid": lambda a, b, c, d: G.apply_pyramid_pattern(a, b, c, decreasing=d),
         "pattern": lambda p, n, s, m: G.apply_pattern(p, n, s, m), "find_multiple": find_multiple}[fn]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            return f(*args)
    except Exception as e:
        return type(e).__name__


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
def test_budget_arithmetic_matches_reference_on_random_inputs():
    rng = random.Random(20260928)
    cases = []
    for _ in range(400):
        mx = rng.choice([52, 77, 1000, 4096, 8192, 10240, 18432, 34816, rng.randint(16, 40000)])
        v = rng.choice([round(rng.uniform(0.01, 1.0), rng.choice([1, 2, 3])), 1, rng.randint(2, 2 * mx), float(rng.randint(2, mx))])
        cases.append(["normalize", [v, mx]])
    for _ in range(300):
        mx = rng.choice([4096, 10240, 18432, 34816])
        n = rng.choice([2, 4, 8, 16, 32, 80, rng.randint(2, 64)])
        cases.append(["pyramid", [rng.choice([64, 256, 300, 512, 1024, 2560, 3488, rng.randint(32, mx // 2)]), mx, n, rng.random() < 0.7]])
    for _ in range(200):
        pat = [rng.choice([128, 512, 1024, "a", 0.5]) for _ in range(rng.choice([1, 1, 2, 4]))]
        n = rng.choice([4, 8, 16, 32, 6])
        cases.append(["pattern", [pat, n, rng.choice(["tile", "repeat", "pyramid", "funnel"]), rng.choice([4096, 18432])]])
    for _ in range(100):
        cases.append(["find_multiple", [rng.randint(0, 5000), rng.choice([1, 8, 16, 256])]])
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "oracle")], input=json.dumps(cases), capture_output=True, text=True,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(ref) == len(cases)
    bad = [(c, want, _ours(*c)) for c, want in zip(cases, ref) if _ours(*c) != want]
    assert not bad, f"{len(bad)} of {len(cases)} differ, first: {bad[:3]}"
    assert sum(1 for x in ref if not isinstance(x, str)) > 0.5 * len(cases)  # (mostly values, not mostly refusals)
