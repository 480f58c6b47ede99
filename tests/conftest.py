import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle_lib

    oracle_lib.build()
    oracle_lib.fns()
    return oracle_lib


@pytest.fixture(scope="session", autouse=True)
def _cpu_twin_session(request):
    """CC_TEST_CPU_TWIN=1 (set by tests/test_host_e2e_cpu.py for its child run only): every test of the session runs the product's
    Python layer on CPU tensors over the oracle's twins (tests/cpu_twin.py)."""
    if os.environ.get("CC_TEST_CPU_TWIN") != "1":
        yield
        return
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cpu_twin import cpu_twin

    mp = pytest.MonkeyPatch()
    with cpu_twin(mp, request.getfixturevalue("oracle")):
        yield
    mp.undo()


# Tests that need two or more GPUs have never executed on hardware (every lease so far had one GPU: they skip there).  On a box
# that has the GPUs they run LAST, so that a first-contact failure under `-x` cannot hide the rest of the suite's results.
_FIRST_CONTACT = ("test_tp2_rccl_matches_unsharded", "test_oneshot_allreduce_over_xgmi")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("CC_TEST_CPU_TWIN") == "1":  # the CPU-twin child run: tests about the device itself are not its business
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from cpu_twin import DEVICE_ONLY_TESTS

        drop = [it for it in items if any(rx.search(it.nodeid) for rx in DEVICE_ONLY_TESTS)]
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = [it for it in items if it not in drop]
    last = [it for it in items if any(n in it.nodeid for n in _FIRST_CONTACT)]
    if last:
        rest = [it for it in items if it not in last]
        items[:] = rest + last


# ---- the near-tie audit: tests that ACCEPT an eviction differing from the reference / the oracle when it is a
#      rounding-level tie in the reference's own scores report how often they did — printed with the run's summary, also under -q.
_AUDIT = []


@pytest.fixture
def audit(request):
    def add(text):
        _AUDIT.append(f"{request.node.nodeid}: {text}")

    return add


def pytest_terminal_summary(terminalreporter):
    if _AUDIT:
        terminalreporter.section("near-tie audit (accepted eviction differences: count / evictions compared)")
        for line in _AUDIT:
            terminalreporter.write_line(line)
