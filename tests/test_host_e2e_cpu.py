"""The product's HOST logic end to end in the CPU suite: the tiny-Llama harness, per-layer cache construction, prompt compaction,
the generation loop and every cache class's Python side run on CPU tensors with the C-ABI calls served by the oracle's `_cpu` twins
(tests/cpu_twin.py — test-only wiring; the product has no CPU path), against the SAME reference-made F1 fixtures and the SAME
assertions as the GPU test (tests/test_gpu_e2e.py::check_e2e): generated tokens identical, per-step per-layer eviction slots exact,
fp32 logits within 1e-3, final pos / mask / counts / K / history.  What the GPU run adds is the HIP kernels; what this run pins
without a GPU is everything above the C ABI plus the oracle's fused-step twins driven by the real caller (ref: generation_utils.py
399-531, model.py:191-233, 363-432)."""
import os
import subprocess
import sys

import pytest
import torch

import test_gpu_e2e as E
from cpu_twin import TWIN_FILES, cpu_twin

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def twin(monkeypatch, oracle):
    monkeypatch.setattr(E, "DEV", "cpu")
    with cpu_twin(monkeypatch, oracle) as fns:
        yield fns


@pytest.mark.parametrize("name", ["f1_e2e_recent_global.npz", "f1_e2e_full.npz", "f1_e2e_heavy_hitter.npz",
                                  "f1_e2e_heavy_hitter_short.npz", "f1_e2e_l2.npz", "f1_e2e_hh_pyramid.npz"])
def test_e2e_host_logic_matches_reference_on_cpu(twin, name):
    E.check_e2e(name)


def test_c1_ring_known_answer_on_cpu(twin):
    """BASELINE config C1 (the reference's CPU-runnable case): recent_global, S = 16, g = 4 -> the slot at decode step t is 4 + (t mod 12)."""
    f, model, seq, log, _ = E._run("f1_e2e_recent_global.npz")
    for li in range(2):
        idx = [int(x) for x in torch.stack(log[li]).view(-1)]
        assert idx == [4 + (t % 12) for t in range(len(idx))]


def test_fixture_driven_gpu_tests_pass_on_the_cpu_twin():
    """Every fixture-driven `-m gpu` test file AGAIN, in a child run, with the product's Python layer on CPU tensors over the oracle's
    twins (CC_TEST_CPU_TWIN=1, tests/conftest.py): the cache classes' replays of the reference's traces (f2 / f3 / f4 / f9 / f10), the
    prompt compressors (f5), attention (f7), the hybrid cache (f6), the end-to-end runs (f1, f9) — all the assertions the GPU run makes,
    minus the tests that are about the device itself (cpu_twin.DEVICE_ONLY_TESTS).  Pins the host side of every class in the CPU suite."""
    env = dict(os.environ, CC_TEST_CPU_TWIN="1", CC_TEST_DEVICE="cpu", PYTHONDONTWRITEBYTECODE="1")
    run = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + [os.path.join(HERE, f) for f in TWIN_FILES],
                         capture_output=True, text=True, env=env, cwd=os.path.dirname(HERE), timeout=1500)
    assert run.returncode == 0, run.stdout[-5000:] + run.stderr[-2000:]
    tail = run.stdout.strip().splitlines()[-1]
    n_passed = int(tail.split(" passed")[0].split()[-1])
    assert n_passed >= 90 and "failed" not in tail, tail
