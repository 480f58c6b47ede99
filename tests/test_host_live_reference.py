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


CHILD_SETUP = r"""
import argparse, contextlib, io, json, sys
sys.path.insert(0, sys.argv[1])
import gen_golden
import torch
A, C, G, M, P = gen_golden._import_reference()
out = []
for n_layer, max_seq, kw in json.load(sys.stdin):
    cfg = dict(gen_golden.TINY); cfg["n_layer"] = n_layer; cfg["block_size"] = max(256, max_seq)
    model = M.Transformer(M.ModelArgs(**cfg)).to(torch.float32).eval()
    parser = argparse.ArgumentParser(); C.add_cache_arguments(parser); G.add_generation_arguments(parser)
    ck = vars(parser.parse_args([])); ck.update(kw)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            r = G.setup_caches(model, gen_golden.FakeTok(), "cpu", max_seq, dict(ck))
        layers = [[type(l.attention.kv_cache).__name__, int(l.attention.kv_cache.max_cache_length), list(l.attention.kv_cache.pos.shape),
                   int(getattr(l.attention.kv_cache, "recent_window", -1)), int(getattr(l.attention.kv_cache, "global_tokens", -1)),
                   type(l.attention.prompt_compressor).__name__] for l in model.layers]
        out.append({"max_cache_length": list(r["max_cache_length"]), "recent_window": list(r["recent_window"]), "cache_strategy": list(r["cache_strategy"]),
                    "prompt_compression_strategy": list(r["prompt_compression_strategy"]), "layers": layers})
    except Exception as e:
        out.append(type(e).__name__)
print(json.dumps(out))
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
def test_setup_caches_matches_reference_on_random_configs(monkeypatch, oracle):
    """setup_caches (ref: generation_utils.py:324-388, model.py:191-233) on random configurations — strategies and their patterns, cache
    lengths as fractions or counts with tile / repeat / pyramid / funnel, recent windows as fractions or counts, layer counts — against
    the reference's: the normalised keyword lists it returns, and per layer the cache class, its length, buffer shape, window, sinks
    and the prompt compressor's class; or the same exception type.  (The caches are built on CPU tensors over the oracle's twins.)"""
    import argparse
    import contextlib
    import io
    import sys as _sys

    _sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    from cpu_twin import cpu_twin

    import cold_compress_amd.cache as cache
    from cold_compress_amd.harness import ModelArgs, Transformer, setup_caches

    rng = random.Random(777)
    pairs = [("recent_global", "recent_global"), ("heavy_hitter", "heavy_hitter"), ("l2", "l2"), ("full", "full"), ("random", "random"),
             ("keep_it_odd", "keep_it_odd"), ("recent_global", "l2"), ("l2", "recent_global")]
    cases = []
    for _ in range(80):
        n_layer = rng.choice([2, 4, 8])
        max_seq = rng.choice([96, 200, 256, 1000])
        k = rng.choice([1, 1, 2])
        strat = [rng.choice(pairs) for _ in range(k)]
        lengths = [rng.choice([0.25, 0.5, 0.1, 1.0, 16, 32, 64, 24, 40]) for _ in range(rng.choice([1, 1, 2]))]
        pat = rng.choice(["tile", "repeat", "tile", "repeat", "pyramid", "funnel"])
        kw = dict(cache_strategy=[a for a, _ in strat], prompt_compression_strategy=[b for _, b in strat], max_cache_length=lengths,
                  cache_length_pattern=pat, cache_strategy_pattern=rng.choice(["tile", "repeat"]), global_tokens=rng.choice([1, 4, 4, 8, 30]),
                  recent_window=rng.choice([10, 4, 0.1, 0.5, 1, 100]))
        if "full" in kw["cache_strategy"]:
            kw["max_cache_length"] = [1.0]
        cases.append([n_layer, max_seq, kw])
    r = subprocess.run([sys.executable, "-c", CHILD_SETUP, os.path.join(ROOT, "oracle")], input=json.dumps(cases), capture_output=True, text=True,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = json.loads(r.stdout.strip().splitlines()[-1])
    n_ok = 0
    with cpu_twin(monkeypatch, oracle):
        for (n_layer, max_seq, kw), want in zip(cases, ref):
            cfg = dict(block_size=max(256, max_seq), vocab_size=128, n_layer=n_layer, n_head=4, n_local_heads=2, dim=64, intermediate_size=128)
            model = Transformer(ModelArgs(**cfg)).to(torch.float32).eval()
            ap = argparse.ArgumentParser()
            cache.add_cache_arguments(ap)
            ck = vars(ap.parse_args([]))
            ck.update(kw)

            class Tok:
                def special_ids(self):
                    return [[1], [2, 3]]

                def punctuation_ids(self):
                    return [5, 6, 7]

            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    got_kw = setup_caches(model, Tok(), "cpu", max_seq, dict(ck))
                layers = [[type(l.attention.kv_cache).__name__, int(l.attention.kv_cache.max_cache_length), list(l.attention.kv_cache.pos.shape),
                           int(getattr(l.attention.kv_cache, "recent_window", -1)), int(getattr(l.attention.kv_cache, "global_tokens", -1)),
                           type(l.attention.prompt_compressor).__name__] for l in model.layers]
                got = {"max_cache_length": list(got_kw["max_cache_length"]), "recent_window": list(got_kw["recent_window"]),
                       "cache_strategy": list(got_kw["cache_strategy"]), "prompt_compression_strategy": list(got_kw["prompt_compression_strategy"]),
                       "layers": layers}
            except Exception as e:
                got = type(e).__name__
            assert got == want, f"n_layer {n_layer}, max_seq {max_seq}, {kw}:\nours      {got}\nreference {want}"
            n_ok += isinstance(want, dict)
    assert n_ok >= 30, f"only {n_ok} of {len(cases)} configurations were valid in the reference: the draw is too hostile"
