"""Pin the oracle's hybrid (FastGen) decode update and history ring against traces captured from the reference's
KVCacheHybrid (tests/golden/f6_*.npz).  The state after the reference's prefill profiling is loaded verbatim (its
slot order comes from a non-stable argsort and is implementation-defined), then every decode step is
replayed: fill indices, per-head counts, masks, positions and the history ring must match bit-for-bit."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from helpers import DT_CODE, DT_FROM_NAME, load_golden, to_np

_HF = {"heavy_hitter": 1, "window": 2, "punc": 4, "special": 8}
FIXTURES = ["f6_hybrid_f32.npz", "f6_hybrid_bf16.npz", "f6_hybrid_mixed_f32.npz", "f6_fastgen_f32.npz", "f6_hybrid_long_bf16.npz"]


def policy_table(strategies, S):
    rows = []
    for s in strategies:
        n = s["strategy"]
        flags = 16 if n == "full" else sum(v for k, v in _HF.items() if k in n)
        rows.append([flags, round(s.get("recent_window", 0) * S), round(s.get("heavy_hitter_frac", 0) * S)])
    return np.array(rows, np.int32)


@pytest.mark.parametrize("name", FIXTURES)
def test_hybrid_decode_replay_bit_exact(oracle, name):
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    code = DT_CODE[dtype]
    H, L, S, D, W = f["H"], f["L"], f["S"], f["D"], 400
    strategies = json.loads(f["strategies_json"])
    tab = policy_table(strategies, S)
    k = to_np(f["k_after_prefill"][0])
    es = k.dtype
    v = np.zeros_like(k)  # V is never read by the policy; only its writes are checked through K's twin path
    pos = f["pos_after_prefill"][0].numpy().copy()
    mask = f["mask_after_prefill"][0, :, 0].numpy().astype(np.uint8)
    cts = f["cts_after_prefill"].numpy().astype(np.int32).copy()
    num = to_np(f["num_after_prefill"][0])
    denom = f["denom_after_prefill"][0].numpy().copy()
    counter = np.array([1], np.int64)  # the prefill seeded ring slot 0
    strat = f["cache_strategies"].numpy().astype(np.int64).copy()
    special = f["special_mask_after_prefill"][0].numpy().astype(np.uint8) if "special_mask_after_prefill" in f else None
    punc = f["punc_mask_after_prefill"][0].numpy().astype(np.uint8) if "punc_mask_after_prefill" in f else None
    n_special = f["num_special"].numpy().astype(np.int32).reshape(1).copy() if special is not None else None
    n_punc = f["num_punc"].numpy().astype(np.int32).reshape(1).copy() if punc is not None else None
    punc_ids = {5, 6, 7}
    o = oracle
    ai = 0
    for t in range(f["steps"]):
        view = o.view(k, v, pos, mask, cts, code)
        p = np.array([L + t], np.int32)
        kn, vn = to_np(f["k_new"][t].reshape(H, D)), to_np(f["v_new"][t].reshape(H, D))
        is_punc = np.array([int(int(f["tok"][t][0]) in punc_ids)], np.uint8) if punc is not None else None
        fill = np.zeros(H, np.int64)
        o.call("cc_hybrid_decode_update", C.byref(view), o.ptr(kn), o.ptr(vn), o.ptr(p), o.ptr(strat), o.ptr(tab), len(tab),
               o.ptr(num), o.ptr(denom), W, o.ptr(special), o.ptr(punc), o.ptr(is_punc), None, None, 0, o.ptr(n_special), o.ptr(n_punc), 4,
               0, o.ptr(fill), None, None, None)  # 0: the reference's history reset on eviction is an effective no-op
        assert np.array_equal(fill, f["fill"][t].numpy()), f"step {t}: {fill} vs {f['fill'][t].tolist()}"
        assert np.array_equal(cts, f["cts_steps"][t].numpy()), f"step {t}"
        if f["requires_hh"]:
            a = to_np(f["attn"][ai][0, :, 0])
            ai += 1
            o.call("cc_hh_ring_update", o.ptr(num), o.ptr(denom), o.ptr(counter), o.ptr(a), H, S, S, W, code, None, None, None)
    assert np.array_equal(pos, f["final_pos"][0].numpy())
    assert np.array_equal(mask.astype(bool), f["final_mask"][0, :, 0].numpy())
    assert np.array_equal(k, to_np(f["final_k"][0]))
    assert np.array_equal(num, to_np(f["final_num"][0]))
    assert np.array_equal(denom, f["final_denom"][0].numpy())
    if punc is not None:
        assert np.array_equal(punc.astype(bool), f["final_punc_mask"][0].numpy())
        assert int(n_punc[0]) == int(f["final_num_punc"][0])


def test_bandsum_matches_window_mask_definition(oracle):
    """band sums == the reference's create_window_attention_mask applied to the attention (cache.py:142-149)."""
    gen = torch.Generator().manual_seed(0)
    H, L, w = 3, 40, 7
    attn = torch.softmax(torch.randn(H, L, L, generator=gen).masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), -1e9), -1)
    win = torch.zeros(L, L, dtype=torch.bool)
    for i in range(L):
        win[i, max(0, i + 1 - w): i + 1] = True
    ref = (attn * win).sum(dim=1)
    out = np.zeros((H, L), np.float32)
    a = attn.numpy().astype(np.float32).copy()
    oracle.call("cc_attn_bandsum", oracle.ptr(a), H, L, L, 0, w, oracle.ptr(out), None)
    assert np.allclose(out, ref.numpy(), atol=1e-6)
