"""The oracle against the reference on vectors NOBODY has looked at: where /root/reference exists (the build container; never the
GPU box), oracle/gen_golden.py makes every fixture family again from shifted seeds (`--seed_offset`, written to a temporary
directory, never into tests/golden) and every CPU oracle test runs on them (`CC_GOLDEN_DIR`, tests/helpers.py).  The committed
fixtures pin the oracle on fixed inputs; this pins it on fresh ones each time the suite runs here — a differential fuzz of the
restatement against the thing it restates ("outputs of the reference itself run here") — and of the product's Python layer, which runs
on the same fresh vectors over the oracle's twins (tests/cpu_twin.py).  CPU-only; skipped without the reference."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORACLE_TESTS = ["test_oracle_golden.py", "test_oracle_hybrid.py", "test_oracle_quant.py", "test_hh_ring.py", "test_window_sums.py",
                "test_hh_query_fixture.py", "test_hybrid_profile_ref.py"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
@pytest.mark.parametrize("offset,jitter", [(1009, False), (77003, True)])
def test_oracle_matches_reference_on_fresh_seeds(tmp_path, offset, jitter):
    """jitter: the cache replays and the hybrid cases also move their lengths, windows, thresholds and step counts (--jitter_shapes)."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--out", str(tmp_path), "--seed_offset", str(offset)]
                         + (["--jitter_shapes"] if jitter else []), capture_output=True, text=True, env=env, timeout=900)
    assert gen.returncode == 0, gen.stdout[-2000:] + gen.stderr[-2000:]
    committed = sorted(f for f in os.listdir(os.path.join(HERE, "golden")) if f.endswith((".npz", ".json")))
    assert sorted(os.listdir(tmp_path)) == committed, "the generator no longer writes the fixture set that is committed"
    env["CC_GOLDEN_DIR"] = str(tmp_path)
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"]
                         + [os.path.join(HERE, t) for t in ORACLE_TESTS], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert run.returncode == 0, f"seed offset {offset}:\n" + run.stdout[-4000:] + run.stderr[-2000:]
    assert " passed" in run.stdout and "failed" not in run.stdout
    # ... and the PRODUCT's host side on the same fresh vectors: the harness end to end and generate()'s branches over the oracle's twins
    # (tests/test_host_e2e_cpu.py; its child run of the fixture-driven `-m gpu` files stays with the committed vectors — on fresh ones it is
    # tools/fuzz_fresh_seeds.sh's business: the offsets here are fixed, so this is a regression test, and the CPU suite should stay short)
    sel = ["-k", "not fixture_driven"]
    twin = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider", os.path.join(HERE, "test_host_e2e_cpu.py")] + sel,
                          capture_output=True, text=True, env=env, cwd=ROOT, timeout=1800)
    assert twin.returncode == 0, f"seed offset {offset} (CPU twin):\n" + twin.stdout[-4000:] + twin.stderr[-2000:]
