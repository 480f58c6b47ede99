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


# ---- the acceptance audit (near-tie evictions; VERDICT r5 #3: EVERY rule that accepts a difference, one table).
#      A test that lets something other than equality pass — an eviction inside a rounding-level tie of the reference's own scores, a
#      kept token swapped on a top-k boundary tie, an l2 eviction between two equal norms, an 8-bit code on a rounding boundary, a
#      probability behind a score on a bf16 rounding midpoint — reports how often it did (count), out of how many comparisons
#      (compared), against which limit.  Printed with the run's summary (also under -q): the per-test lines, then one row per rule.
_AUDIT = []
RULES = ("near-tie eviction", "near-tie candidate re-seat", "top-k boundary swap", "l2 norm tie", "q8 boundary code", "boundary probability")


@pytest.fixture
def audit(request):
    def add(text, rule=None, count=None, compared=None, limit=None):
        """text: the per-test line (kept from r3-r5); rule / count / compared / limit: its row of the table — one call per rule."""
        assert rule is None or rule in RULES, rule
        _AUDIT.append(dict(node=request.node.nodeid, text=text, rule=rule, count=count, compared=compared, limit=limit))

    return add


def pytest_terminal_summary(terminalreporter):
    if not _AUDIT:
        return
    terminalreporter.section("near-tie audit (accepted eviction differences: count / evictions compared)")
    for a in _AUDIT:
        terminalreporter.write_line(f"{a['node']}: {a['text']}")
    rows = {}
    for a in _AUDIT:
        if a["rule"] is None:
            continue
        r = rows.setdefault(a["rule"], dict(tests=0, count=0, compared=0, limits=[]))
        r["tests"] += 1
        r["count"] += int(a["count"] or 0)
        r["compared"] += int(a["compared"] or 0)
        if a["limit"] is not None and str(a["limit"]) not in r["limits"]:
            r["limits"].append(str(a["limit"]))
    terminalreporter.section("acceptance rules: every rule that lets a difference pass (accepted / compared, against its limit)")
    terminalreporter.write_line(f"{'rule':<28} {'tests':>5} {'accepted':>9} {'compared':>12}  limit")
    for rule in RULES:
        r = rows.get(rule)
        if r is None:
            terminalreporter.write_line(f"{rule:<28} {0:>5} {'-':>9} {'-':>12}  (no test of this selection uses the rule)")
        else:
            terminalreporter.write_line(f"{rule:<28} {r['tests']:>5} {r['count']:>9} {r['compared']:>12}  {'; '.join(r['limits'])}")
