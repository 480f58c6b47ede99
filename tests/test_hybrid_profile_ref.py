"""Pin tests/hybrid_profile_ref.py — the numpy restatement the FULL-SIZE hybrid profiling test stands on
(tests/test_gpu_fullsize.py::test_hybrid_profiling_full_size) — to the reference itself: every f6_* fixture holds the
[1, H, L, L] attention the reference's KVCacheHybrid.profile_and_update was fed (cache.py:1138-1272) and what it decided:
`cache_strategies`, `cache_cts`, and the kept positions per head (`pos` after the prefill; the slot ORDER is the
reference's non-stable argsort, SURVEY §8 a14 — sets are compared).  VERDICT r2 "next" item 3."""
import json

import numpy as np
import pytest

import hybrid_profile_ref as hp
from helpers import load_golden, to_np

FIXTURES = ["f6_hybrid_f32.npz", "f6_hybrid_bf16.npz", "f6_hybrid_mixed_f32.npz", "f6_fastgen_f32.npz", "f6_hybrid_long_bf16.npz"]
SPECIAL = [[1], [2, 3]]  # oracle/gen_golden.py::hybrid_case
PUNC = [5, 6, 7]
G = 4


def special_mask(ids):  # ref: cache.py:1021-1034
    m = np.zeros(len(ids), bool)
    for seq in SPECIAL:
        n = len(seq)
        for i in range(len(ids) - n + 1):
            if list(ids[i:i + n]) == seq:
                m[i:i + n] = True
    return m


def as_f32(t, dtype_name):
    a = to_np(t)
    if dtype_name == "bfloat16":
        return (a.astype(np.uint32) << 16).view(np.float32)
    return a.astype(np.float32)


@pytest.mark.parametrize("name", FIXTURES)
def test_profile_restatement_matches_reference_capture(name):
    f = load_golden(name)
    dtype_name = str(f["dtype"])
    H, L, S = f["H"], f["L"], f["S"]
    strategies = json.loads(f["strategies_json"])
    ids = f["ids"][0].numpy()
    A = as_f32(f["attn0"][0], dtype_name)
    uses_special = any("special" in s["strategy"] for s in strategies)
    uses_punc = any("punc" in s["strategy"] for s in strategies)
    ref = hp.profile(A, strategies, G, float(f["min_recovery_frac"]), S, dtype_name,
                     special_mask=special_mask(ids) if uses_special else None,
                     punc_mask=np.isin(ids, PUNC) if uses_punc else None, chunk=16)
    want = f["cache_strategies"].numpy().astype(np.int64)
    cts = f["cts_after_prefill"].numpy().astype(np.int64)
    pos = f["pos_after_prefill"][0].numpy()
    for h in range(H):
        # the policy: equal, unless a score sits within one rounding of the threshold (the reference compares in the model dtype)
        if int(ref["strategies"][h]) != int(want[h]):
            ulp = 2.0 ** -8 if dtype_name == "bfloat16" else 2.0 ** -22
            near = np.abs(ref["scores"][:, h] - ref["threshold"]) <= 2 * ulp * max(1.0, ref["threshold"])
            assert near.any(), f"{name} head {h}: policy {int(ref['strategies'][h])} vs the reference's {int(want[h])}; scores {ref['scores'][:, h]}"
            continue
        keep = ref["mask_optimal"][h]
        assert int(keep.sum()) == int(cts[h]), f"{name} head {h}: {int(keep.sum())} kept vs the reference's {int(cts[h])}"
        kept_ref = np.zeros(L, bool)
        kept_ref[pos[h, :cts[h]]] = True
        assert (pos[h, cts[h]:] == -1).all()
        diff = keep ^ kept_ref
        if diff.any():
            # only members of the heavy-hitter top-k at its boundary VALUE may differ (torch.topk's tie order is its own;
            # the restatement takes the lowest index first) — SURVEY §8(c)(2)
            pol = strategies[int(want[h])]["strategy"]
            assert "heavy_hitter" in pol, f"{name} head {h}: kept sets differ without a top-k"
            cum = ref["cum_attn"][h]
            assert len(set(cum[diff].tolist())) == 1, f"{name} head {h}: differing members are not one tie class: {cum[diff]}"
