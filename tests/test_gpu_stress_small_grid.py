"""The adversarial schedule of the in-launch hand-off, in the driver-run suite: the differential fuzz of the fused
decode step (tools/fuzz_step.py) with 8 kv heads and 70 .. 190 slots — one or two workgroups per kv head, 8-16 per launch, the chip
idle between launches.  XCDs that wake late then start a step's workgroups after other kv heads have FINISHED theirs, which is how
the shared key row of the head-constant policies was found in round 3 (3.6 % of the steps; `pytest -m gpu` had been green for two
rounds).  Every step kind gets its own run of fresh caches, each case = the fused step (ONE launch) against update_kv -> attention ->
update_state over six steps, every buffer bit for bit."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

# (family, strategy, cases): >= 1000 fresh fused-step caches per step kind (the head-constant kind shares its kernel: 1200 over its
# three policies)
KINDS = [("step", "heavy_hitter", 1000), ("step", "l2", 1600),  # (l2: a third of the draws is fp32, which has no fused step) ("step", "recent_global", 500), ("step", "full", 200),
         ("step", "random", 500), ("hybrid", None, 1000), ("quant", None, 1000)]


@pytest.mark.parametrize("family,strategy,n", KINDS, ids=[f"{f}-{s or 'all'}" for f, s, _ in KINDS])
def test_small_grid_stress(family, strategy, n):
    import fuzz_step as F

    from cold_compress_amd.attention_utils import reset_single_launch_status, single_launch_status

    ran, bad = F.small_grid_stress(family, n, strategy, extra={2: 128} if strategy == "l2" else None)  # (l2's fused step: head_dim 128)
    st = single_launch_status()
    if st:
        reset_single_launch_status()
    assert st == 0, "a hand-off timed out during the stress"
    assert ran >= (1000 if n >= 1000 else n * 0.6), f"only {ran} of {n} cases ran"
    assert not bad, f"{len(bad)} of {ran} cases differ; first: {bad[0]}"
