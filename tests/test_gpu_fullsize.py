"""Parity at the prompt lengths the BASELINE configurations actually use (8192 and 16384 tokens), on a subset of heads so the
C oracle — OpenMP over the host's cores, still the same arithmetic — finishes in seconds:

  * the matrix-core prefill attention with its side planes (column sums, SnapKV observation window, FastGen band sums at the
    config's window width 0.1 * L) against the oracle (attention_utils.py:36-54, cache.py:1093, 1155);
  * KVCacheHybrid's prefill profiling at L = 16384 / S = 18432 against a row-by-row restatement of the reference's
    definition over the materialised [H, L, L] attention (cache.py:1066-1187; tests/hybrid_profile_ref.py);
  * a heavy-hitter prefill -> SnapKV compaction -> decode replay that NEVER overwrites the device's numeric state with the
    oracle's: every eviction is either identical or justified as a rounding-level near-tie in the oracle's own scores, and
    the drift of the float64 history is bounded.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch

from helpers import from_np, hh_own_state_steps, to_np

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16_ULP = 2.0 ** -8


@pytest.fixture()
def oracle_mt(oracle):
    """The oracle on 16 host threads for the full-size cases (identical results, see oracle/cc_oracle.c; measured on the
    256-thread GPU box, tools/oracle_scale.py: 16 threads 1.4 s, 64 threads 2.2 s, 256 threads 11.4 s at L = 8192 — the
    row loop is bound by the shared K / V image, not by cores)."""
    oracle.set_threads(min(16, os.cpu_count() or 1))
    yield oracle
    oracle.set_threads(1)


@pytest.mark.parametrize("L", [8192, 16384])
def test_prefill_bands_full_size(oracle_mt, L):
    from cold_compress_amd.attention_utils import prefill_attention

    o = oracle_mt
    HQ, H, D, dtype = 4, 1, 128, torch.bfloat16
    band = max(1, int(0.1 * L))  # hybrid.yaml: recent_window 0.1 (cache.py:1093)
    gen = torch.Generator().manual_seed(L)
    q = (1.5 * torch.randn(1, HQ, L, D, generator=gen)).to(dtype)
    k = (1.5 * torch.randn(1, H, L, D, generator=gen)).to(dtype)
    v = torch.randn(1, H, L, D, generator=gen).to(dtype)
    y, summ = prefill_attention(q.to(DEV), k.to(DEV), v.to(DEV), return_attn=True, bands=[band])
    torch.cuda.synchronize()
    yo, cs, ob = np.zeros((HQ, L, D), np.uint16), np.zeros((H, L), np.float32), np.zeros((H, L), np.float32)
    bo = np.zeros((1, H, L), np.float32)
    barr = (C.c_int32 * 1)(band)
    o.call("cc_prefill_attn_bands", o.ptr(to_np(q[0])), o.ptr(to_np(k[0])), o.ptr(to_np(v[0])), HQ, H, L, D, 1, 1.0 / math.sqrt(D),
           o.ptr(yo), o.ptr(cs), o.ptr(ob), 16, barr, 1, o.ptr(bo), None, 0, None)
    yref = from_np(yo, dtype).float()
    # y: 1e-3 (north star) + two roundings of the output dtype
    assert (y.cpu().float()[0] - yref).abs().max() <= 1e-3 + 2 * BF16_ULP * yref.abs().max()
    # sums of up to L probabilities, each rounded to bf16 on both sides (the device's exp / reciprocal may land on the
    # neighbouring bf16 value): one bf16 ulp of the sum, plus the 5e-2 the small-shape tests allow
    for name, mine, ref in (("colsum", summ.colsum.cpu(), torch.from_numpy(cs)), ("band", summ.bands[band].cpu(), torch.from_numpy(bo[0]))):
        err = (mine - ref).abs()
        assert bool((err <= 5e-2 + BF16_ULP * ref.abs()).all()), f"{name}: max err {float(err.max())}"
    assert (summ.obs_mean.cpu() - torch.from_numpy(ob)).abs().max() < 4e-3
    # size-independent properties: every query row's probabilities sum to ~1, so the column sums add up to ~L, and the
    # band sums are the part of them within `band` queries of each key
    assert abs(float(summ.colsum.sum()) / L - 1.0) < 2e-2
    assert bool((summ.bands[band] <= summ.colsum + 1e-3).all())


@pytest.mark.gpu
@pytest.mark.parametrize("L,H", [(8192, 1), (16384, 1), (1000, 2), (97, 1), (4133, 8)])
def test_prefill_single_pass_full_size(oracle_mt, L, H):
    """return_attn=False: the ONE-pass causal attention (online softmax; the reference's fused fast path,
    attention_utils.py:27-35) against the oracle's attention at the BASELINE prompt lengths and at ragged ones, and against
    the two-pass product path on the same inputs (a different rounding of the probabilities, the same contract)."""
    from cold_compress_amd.attention_utils import prefill_attention

    o = oracle_mt
    R, D, dtype = 4, 128, torch.bfloat16
    HQ = H * R
    gen = torch.Generator().manual_seed(L + H)
    q = (1.5 * torch.randn(1, HQ, L, D, generator=gen)).to(dtype)
    k = (1.5 * torch.randn(1, H, L, D, generator=gen)).to(dtype)
    v = torch.randn(1, H, L, D, generator=gen).to(dtype)
    y, summ = prefill_attention(q.to(DEV), k.to(DEV), v.to(DEV), return_attn=False)
    assert summ is None
    y2, _ = prefill_attention(q.to(DEV), k.to(DEV), v.to(DEV), return_attn=True)
    torch.cuda.synchronize()
    yo, cs, ob = np.zeros((HQ, L, D), np.uint16), np.zeros((H, L), np.float32), np.zeros((H, L), np.float32)
    o.call("cc_prefill_attn", o.ptr(to_np(q[0])), o.ptr(to_np(k[0])), o.ptr(to_np(v[0])), HQ, H, L, D, 1, 1.0 / math.sqrt(D),
           o.ptr(yo), o.ptr(cs), o.ptr(ob), 16, None, 0, None)
    yref = from_np(yo, dtype).float()
    tol = 1e-3 + 2 * BF16_ULP * float(yref.abs().max())  # the north star's 1e-3 + two roundings of the output dtype
    assert not bool(torch.isnan(y).any())
    assert float((y.cpu().float()[0] - yref).abs().max()) <= tol
    assert float((y.cpu().float() - y2.cpu().float()).abs().max()) <= tol


HYBRID = [{"strategy": "window", "recent_window": 0.1},
          {"strategy": "window_heavy_hitter", "heavy_hitter_frac": 0.25, "recent_window": 0.1},
          {"strategy": "window_heavy_hitter", "heavy_hitter_frac": 0.5, "recent_window": 0.1},
          {"strategy": "full"}]


@pytest.mark.parametrize("H,R", [(3, 2), (8, 4)])  # (r6, VERDICT r5 #3: the Llama-3-8B head geometry — 8 kv heads, 4 query heads each — as well)
def test_hybrid_profiling_full_size(oracle_mt, H, R):
    """C4: 16384-token prompt, cache 18432, hybrid.yaml's four policies; three kinds of heads (tests/hybrid_inputs.py)."""
    import cold_compress_amd.cache as cache
    import hybrid_profile_ref as hp
    from cold_compress_amd.attention_utils import prefill_attention
    from hybrid_inputs import make_inputs

    o = oracle_mt
    L, S, D, g, frac, dtype = 16384, 18432, 128, 4, 0.97, torch.bfloat16
    HQ = H * R
    q, k, v = make_inputs(L, H, R, D, 0, dtype)
    cls, rk = cache.get_cache_constructor("hybrid")
    kw = dict(max_cache_length=S, max_seq_length=S, cache_bits=None, global_tokens=g, token_ids={"special": [], "punctuation": []},
              min_recovery_frac=frac, hybrid_strategies=HYBRID)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{x: kw[x] for x in rk})
    pos0 = torch.arange(L, device=DEV)
    ids = torch.zeros(1, L, dtype=torch.int64, device=DEV)
    kd, vd = k.unsqueeze(0).to(DEV), v.unsqueeze(0).to(DEV)
    y, summ = prefill_attention(q.unsqueeze(0).to(DEV), kd, vd, return_attn=True, bands=kv.attn_bands(L))
    kv.update_kv(pos0, kd, vd, True, input_ids=ids)
    kv.update_state(pos0, kd, vd, True, summ, input_ids=ids)
    torch.cuda.synchronize()
    # ---- the reference's definition, row by row over the materialised attention
    A = np.zeros((H, L, L), np.float32)
    yo = np.zeros((HQ, L, D), np.uint16)
    o.prefill_attn_matrix(to_np(q), to_np(k), to_np(v), HQ, H, L, D, 1, 1.0 / math.sqrt(D), yo, A)
    ref = hp.profile(A, HYBRID, g, frac, S, "bfloat16")
    del A
    mine = kv.cache_strategies.cpu().tolist()
    cts = kv.cache_cts.cpu().tolist()
    pos = kv.pos.cpu()[0]
    assert len(set(ref["strategies"].tolist())) >= 2, "the synthetic heads were meant to pick different policies"
    checked = 0
    for h in range(H):
        near = [p for p in range(len(HYBRID)) if abs(float(ref["scores"][p, h]) - ref["threshold"]) <= 2 * BF16_ULP]
        if mine[h] != int(ref["strategies"][h]):
            # only a score within rounding of the threshold may tip a head to the neighbouring policy (cache.py:633-635 of ours)
            assert near, f"head {h}: policy {mine[h]} vs {int(ref['strategies'][h])}, scores {ref['scores'][:, h]}"
            continue
        checked += 1
        keep_ref = ref["mask_optimal"][h]
        assert cts[h] == int(keep_ref.sum()), f"head {h}: {cts[h]} kept vs {int(keep_ref.sum())}"
        kept = pos[h, :cts[h]]
        assert bool((kept[1:] > kept[:-1]).all()) and bool((pos[h, cts[h]:] == -1).all())
        mine_set = np.zeros(L, bool)
        mine_set[kept.numpy()] = True
        diff = mine_set ^ keep_ref
        if diff.any():
            # only heavy-hitter members at the k-th column mean may differ: ties (and values one bf16 step apart — the two
            # sides round their fp32 column sums independently) are interchangeable
            cols, win = ref["filling"][int(ref["strategies"][h])]
            assert "heavy_hitter" in HYBRID[int(ref["strategies"][h])]["strategy"], f"head {h}: kept sets differ without a top-k"
            cum = ref["cum_attn"][h]
            static = hp.static_columns(HYBRID[int(ref["strategies"][h])]["strategy"], L, g, np.zeros(L, bool), np.zeros(L, bool))
            hh_ref = keep_ref & ~static
            hh_ref[max(0, L - win):] = False
            vk = cum[hh_ref].min()  # the k-th largest column mean
            lo, hi = vk * (1 - 2 * BF16_ULP), vk * (1 + 2 * BF16_ULP)
            assert bool(((cum[diff] >= lo) & (cum[diff] <= hi)).all()), f"head {h}: a non-tie member differs"
            assert diff.sum() <= 0.02 * keep_ref.sum() + 8, f"head {h}: {int(diff.sum())} members differ"
    assert checked >= (2 if H == 3 else 6)
    # the attention output of the same pass, loosely: these heads have logits of magnitude ~40, where one bf16 step of a
    # score (the reference rounds q.k to the model dtype, attention_utils.py:37) is 0.25 — a last-bit difference of the
    # fp32 dot product moves a probability by a quarter.  The tight bound on y is test_prefill_bands_full_size's.
    yr = from_np(yo, dtype).float()
    # (0.05 max|y| in r2; stated here as what it is: NOT the attention contract — these heads were built with logits of magnitude
    #  ~40 to pick different policies — and bounded by one bf16 step of such a score moving a probability by a quarter)
    assert (y.cpu().float()[0] - yr).abs().max() <= 0.05 * yr.abs().max()


@pytest.mark.parametrize("S,H", [(4096, 2), (2560, 8), (4096, 8)])  # C3's length on a head subset; C2 (max_cache_length 0.25 -> 2560) on ALL 8 kv heads (r5); r6: C3 on all 8
def test_heavy_hitter_prefill_to_decode_without_state_sync(oracle_mt, audit, S, H):
    """8192-token prompt -> SnapKV compaction to S -> history from the column means -> 48 decode steps of the fused
    step, device and oracle each continuing from THEIR OWN numeric state (nothing is copied across after the prompt's
    keep set has been checked)."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import prefill_attention
    from cold_compress_amd.prompt_compression import get_prompt_compressor_constructor

    o = oracle_mt
    L, R, D, g, w, dtype, steps = 8192, 4, 128, 4, 10, torch.bfloat16, 48
    HQ, code = H * R, 1
    gen = torch.Generator().manual_seed(77)
    q = (1.5 * torch.randn(1, HQ, L, D, generator=gen)).to(dtype)
    k = (1.5 * torch.randn(1, H, L, D, generator=gen)).to(dtype)
    v = torch.randn(1, H, L, D, generator=gen).to(dtype)
    cls, rk = cache.get_cache_constructor("heavy_hitter")
    kw = dict(max_cache_length=S, global_tokens=g, max_seq_length=L + 2048, cache_bits=None, recent_window=w, history_window_size=1,
              attn_thresholding=False)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{x: kw[x] for x in rk})
    comp = get_prompt_compressor_constructor("heavy_hitter")(head_specific=True, **{x: kw[x] for x in rk})
    pos0 = torch.arange(L, device=DEV)
    kd, vd = k.to(DEV), v.to(DEV)
    y, summ = prefill_attention(q.to(DEV), kd, vd, return_attn=True)
    keep, kc, vc, state = comp(pos0, kd, vd, attn=summ)
    kv.update_kv(keep, kc, vc, True)
    kv.update_state(keep, kc, vc, True, state)
    torch.cuda.synchronize()
    # ---- oracle: the same pipeline from the same inputs
    yo, cs, ob = np.zeros((HQ, L, D), np.uint16), np.zeros((H, L), np.float32), np.zeros((H, L), np.float32)
    o.call("cc_prefill_attn", o.ptr(to_np(q[0])), o.ptr(to_np(k[0])), o.ptr(to_np(v[0])), HQ, H, L, D, code, 1.0 / math.sqrt(D),
           o.ptr(yo), o.ptr(cs), o.ptr(ob), 16, None, 0, None)
    obs_dt = to_np(torch.from_numpy(ob).to(dtype))
    prio = np.zeros((H, L), np.uint16)
    o.call("cc_snapkv_priority", o.ptr(obs_dt), H, L, code, 16, g, o.ptr(prio), None)
    keep_o = np.zeros((H, S), np.int64)
    o.call("cc_topk_keep", o.ptr(prio), 1, H, L, S, o.ptr(keep_o), None, 0, None)
    keep_d = keep.cpu().numpy().reshape(H, S)
    pf = from_np(prio, dtype).float().numpy()
    for h in range(H):  # tie contract of the keep set: members one bf16 step around the S-th priority may differ
        a, b = set(keep_d[h].tolist()), set(keep_o[h].tolist())
        if a != b:
            kth = np.sort(pf[h])[-S]
            d = np.array(sorted(a ^ b))
            assert bool((np.abs(pf[h][d] - kth) <= 2 * BF16_ULP * abs(kth) + 1e-30).all()), f"head {h}: non-tie keep members differ"
            assert len(d) <= 0.02 * S
    # the oracle continues with the DEVICE's keep set (same cache contents) but its OWN column means
    keep_use = np.ascontiguousarray(keep_d)
    ko, vo = np.zeros((H, S, D), np.uint16), np.zeros((H, S, D), np.uint16)
    o.call("cc_gather_rows", o.ptr(to_np(k[0])), o.ptr(keep_use), H, H, L, S, D, code, o.ptr(ko), None)
    o.call("cc_gather_rows", o.ptr(to_np(v[0])), o.ptr(keep_use), H, H, L, S, D, code, o.ptr(vo), None)
    assert np.array_equal(ko, to_np(kv.k_cache.cpu()[0]))
    mean = np.zeros((H, L), np.uint16)
    o.call("cc_colsum_to_mean", o.ptr(cs), None, H, L, code, o.ptr(mean), None)
    st0 = np.zeros((H, S), np.uint16)
    o.call("cc_gather_vec", o.ptr(mean), o.ptr(keep_use), H, L, S, code, o.ptr(st0), None)
    st = dict(k=ko, v=vo, pos=keep_use.astype(np.int32).copy(), mask=np.ones((H, S), np.uint8), cts=np.array([S], np.int32),
              num=np.zeros((H, S), np.float64), denom=np.zeros((H, S), np.int32), ctr=np.zeros(1, np.int64))
    o.call("cc_hh_update", o.ptr(st["num"]), o.ptr(st["denom"]), o.ptr(st["ctr"]), o.ptr(st0), H, S, S, code, None)
    num_d = kv.attn_history_num.cpu()[0, :, :, 0].numpy()
    assert np.allclose(num_d, st["num"], rtol=2 * BF16_ULP, atol=1e-6), "prefill history beyond one rounding of the column means"
    assert np.array_equal(kv.attn_history_denom.cpu()[0].numpy(), st["denom"])
    # ---- decode: both sides on their own state; a differing eviction must be a near-tie in the ORACLE's scores, and is then
    #      followed (the oracle is made to evict the device's slot) so that the caches stay comparable (tests/helpers.py)
    justified, total = hh_own_state_steps(o, kv, st, gen, L, steps, HQ, g, w, dtype)
    audit(f"n_just = {justified} of {total} evictions (limit 5 %)", rule="near-tie eviction", count=justified, compared=total, limit="5 % of the evictions, each within 2 bf16 roundings of the minimum")
    assert justified <= 0.05 * total, f"{justified} of {total} evictions were near-tie divergences"
    assert kv.step_status(HQ) == 0


def test_l2_prefill_to_decode_without_state_sync(oracle_mt):
    """KVCacheL2 at C3's size: 8192-token prompt -> L2 prompt compaction to 4096 -> key norms from the filled cache -> 40 decode
    steps of the fused step (ONE launch, wide geometry), device and oracle each on THEIR OWN state (the device never receives the
    oracle's norms; VERDICT r2: the golden l2 replays continue from the reference's norms).  Norms are canonical-order on both
    sides, so the evictions must agree EXACTLY; y within the attention contract."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.prompt_compression import get_prompt_compressor_constructor

    o = oracle_mt
    L, S, H, R, D, g, w, dtype, steps = 8192, 4096, 8, 4, 128, 4, 10, torch.bfloat16, 40
    HQ, code = H * R, 1
    gen = torch.Generator().manual_seed(99)
    k = (torch.randn(1, H, L, D, generator=gen) * (0.5 + torch.rand(1, H, L, 1, generator=gen))).to(dtype)  # a spread of norms
    v = torch.randn(1, H, L, D, generator=gen).to(dtype)
    cls, rk = cache.get_cache_constructor("l2")
    kw = dict(max_cache_length=S, global_tokens=g, max_seq_length=L + 2048, cache_bits=None, recent_window=w)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{x: kw[x] for x in rk})
    comp = get_prompt_compressor_constructor("l2")(head_specific=True, **{x: kw[x] for x in rk})
    pos0 = torch.arange(L, device=DEV)
    kd, vd = k.to(DEV), v.to(DEV)
    keep, kc, vc, _ = comp(pos0, kd, vd, attn=None)
    kv.update_kv(keep, kc, vc, True)
    kv.update_state(keep, kc, vc, True, None)
    torch.cuda.synchronize()
    # ---- oracle: its own norms of the prompt's keys -> the same priority rule -> keep set (ties at the boundary: lowest index
    #      first on both sides; norms are canonical-order on both sides, so the sets must be equal)
    kn_all = np.zeros((H, L), np.uint16)
    o.call("cc_row_l2_norm", o.ptr(to_np(k[0])), H, L, D, code, 1, o.ptr(kn_all), None)
    prio = from_np(kn_all, dtype).float()
    ip = torch.arange(L)
    prio[:, (ip < g) | (ip >= L - w)] = float("inf")  # ref: prompt_compression.py:28-43, 201-209
    prio_dt = to_np(prio.to(dtype))
    keep_o = np.zeros((H, S), np.int64)
    o.call("cc_topk_keep", o.ptr(prio_dt), 1, H, L, S, o.ptr(keep_o), None, 0, None)
    assert np.array_equal(keep.cpu().numpy().reshape(H, S), keep_o), "l2 keep sets differ"
    ko, vo = np.zeros((H, S, D), np.uint16), np.zeros((H, S, D), np.uint16)
    o.call("cc_gather_rows", o.ptr(to_np(k[0])), o.ptr(keep_o), H, H, L, S, D, code, o.ptr(ko), None)
    o.call("cc_gather_rows", o.ptr(to_np(v[0])), o.ptr(keep_o), H, H, L, S, D, code, o.ptr(vo), None)
    st = dict(k=ko, v=vo, pos=keep_o.astype(np.int32).copy(), mask=np.ones((H, S), np.uint8), cts=np.array([S], np.int32),
              kn=np.zeros((H, S), np.uint16))
    o.call("cc_row_l2_norm", o.ptr(st["k"]), H, S, D, code, 0, o.ptr(st["kn"]), None)  # ref: cache.py:611-612
    assert np.array_equal(to_np(kv.key_norm.cpu()[0]), st["kn"]), "prefill key norms (canonical order on both sides)"
    for t in range(steps):
        p = L + t
        pt = torch.tensor([p], dtype=torch.int32)
        k1 = (torch.randn(1, H, 1, D, generator=gen) * (0.5 + torch.rand(1, H, 1, 1, generator=gen))).to(dtype)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        q1 = torch.randn(1, HQ, 1, D, generator=gen).to(dtype)
        yd = kv.decode_step(q1.to(DEV), k1.to(DEV), v1.to(DEV), pt.to(DEV))
        torch.cuda.synchronize()
        view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], code)
        idx = np.zeros(H, np.int64)
        o.call("cc_decode_update_l2", C.byref(view), o.ptr(to_np(k1.reshape(H, D))), o.ptr(to_np(v1.reshape(H, D))), o.ptr(pt.numpy().copy()),
               o.ptr(st["kn"]), g, w, o.ptr(idx), None, 0, None)
        assert np.array_equal(kv.pos.cpu()[0].numpy(), st["pos"]), f"step {t}: positions (evicted slot {idx.tolist()})"
        yo1 = np.zeros((HQ, D), np.uint16)
        o.call("cc_decode_attn_gqa", o.ptr(to_np(q1.reshape(HQ, D))), o.ptr(st["k"]), o.ptr(st["v"]), o.ptr(st["mask"]), HQ, H, S, D, code,
               1.0 / math.sqrt(D), o.ptr(yo1), None, None, None, None, None, None, 0, None)
        yr = from_np(yo1, dtype).float()
        assert (yd.cpu().float()[0, :, 0] - yr).abs().max() <= 1e-3 + 2 * BF16_ULP * yr.abs().max(), f"step {t}: y"
    assert np.array_equal(to_np(kv.key_norm.cpu()[0]), st["kn"])
    assert np.array_equal(to_np(kv.k_cache.cpu()[0]), st["k"]) and np.array_equal(to_np(kv.v_cache.cpu()[0]), st["v"])
    from cold_compress_amd.attention_utils import single_launch_status

    assert single_launch_status(kv.pos.device) == 0


def test_hybrid_prefill_to_decode_without_state_sync(oracle_mt, audit):
    """KVCacheHybrid end to end on its OWN numeric state (the f6 replays continue from the reference's ring):
    3000-token prompt -> matrix-core prefill with band sums -> per-head profiling -> ring seeded from the device's column means
    -> 48 fused decode steps.  The oracle runs the same pipeline from the same inputs: profiling by the row-by-row restatement of
    the reference's definition (tests/hybrid_profile_ref.py, pinned to the f6 captures by tests/test_hybrid_profile_ref.py), ring
    seeded from ITS column means, then cc_decode_step_hybrid on its own state.  Nothing numeric crosses over after the prompt: the
    oracle only adopts the device's kept sets / slot order (ties of the prefill top-k are checked at full size elsewhere).  Every
    decode decision — append, evict by position, evict by windowed attention — must be identical, or a rounding-level near-tie
    in the ORACLE's scores (then the oracle is re-seated on the device's state and the stretch restarts)."""
    import cold_compress_amd.cache as cache
    import hybrid_profile_ref as hp
    from cold_compress_amd.attention_utils import prefill_attention
    from hybrid_inputs import make_inputs
    from test_oracle_hybrid import policy_table

    o = oracle_mt
    L, S, H, R, D, g, frac, dtype, steps, W = 3000, 3072, 6, 4, 128, 4, 0.97, torch.bfloat16, 48, 400
    HQ, code = H * R, 1
    q, k, v = make_inputs(L, H, R, D, 3, dtype)
    cls, rk = cache.get_cache_constructor("hybrid")
    kw = dict(max_cache_length=S, max_seq_length=S, cache_bits=None, global_tokens=g, token_ids={"special": [], "punctuation": []},
              min_recovery_frac=frac, hybrid_strategies=HYBRID)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{x: kw[x] for x in rk})
    got = {}
    stable = kv._partition_order
    kv._partition_order = lambda m: got.setdefault("order", stable(m))
    pos0 = torch.arange(L, device=DEV)
    ids = torch.zeros(1, L, dtype=torch.int64, device=DEV)
    kd, vd = k.unsqueeze(0).to(DEV), v.unsqueeze(0).to(DEV)
    _, summ = prefill_attention(q.unsqueeze(0).to(DEV), kd, vd, return_attn=True, bands=kv.attn_bands(L))
    kv.update_kv(pos0, kd, vd, True, input_ids=ids)
    kv.update_state(pos0, kd, vd, True, summ, input_ids=ids)
    torch.cuda.synchronize()
    assert kv.requires_heavy_hitter and kv.supports_fused_step()
    # ---- the oracle's prefill: the reference's definition over the materialised attention
    A = np.zeros((H, L, L), np.float32)
    yo = np.zeros((HQ, L, D), np.uint16)
    o.prefill_attn_matrix(to_np(q), to_np(k), to_np(v), HQ, H, L, D, code, 1.0 / math.sqrt(D), yo, A)
    ref = hp.profile(A, HYBRID, g, frac, S, "bfloat16")
    del A
    strat_d = kv.cache_strategies.cpu().numpy().astype(np.int64)
    assert len(set(strat_d.tolist())) >= 3, f"the synthetic heads were meant to pick three kinds of policies: {strat_d}"
    for h in range(H):
        if strat_d[h] != int(ref["strategies"][h]):  # only a score within rounding of the threshold may tip a head
            assert any(abs(float(ref["scores"][p, h]) - ref["threshold"]) <= 2 * BF16_ULP for p in range(len(HYBRID))), f"head {h}"
        else:
            assert int(kv.cache_cts[h]) == int(ref["mask_optimal"][h].sum()), f"head {h}: kept count"
    order = got["order"].cpu().numpy()  # [H, L]: slot -> prompt token (kept tokens first)
    seed_o = np.take_along_axis(ref["cum_attn"], order, axis=1)  # the ORACLE's column means in the device's slot order
    st = dict(k=to_np(kv.k_cache.cpu()[0]), v=to_np(kv.v_cache.cpu()[0]), pos=kv.pos.cpu()[0].numpy().copy(),
              mask=kv.mask.cpu()[0, :, 0].numpy().astype(np.uint8), cts=kv.cache_cts.cpu().numpy().copy(),
              num=np.zeros((H, S, W), np.uint16), denom=np.zeros((H, S), np.int32), ctr=np.zeros(1, np.int64),
              wsum=np.zeros(H * S, np.float32), acc=np.zeros(o.fns()["cc_hh_ring_acc_words"](H, S, W, code), np.uint64))
    o.call("cc_hh_ring_update", o.ptr(st["num"]), o.ptr(st["denom"]), o.ptr(st["ctr"]), o.ptr(to_np(torch.from_numpy(seed_o).to(dtype))),
           H, S, L, W, code, None, None, None)  # cache.py:1267-1272
    o.call("cc_hh_ring_window_sums", o.ptr(st["num"]), H, S, W, code, o.ptr(st["wsum"]), o.ptr(st["acc"]), None)
    ring_d = kv.attn_history_num.cpu()[0].float().numpy()
    ring_o = from_np(st["num"], dtype).float().numpy()
    assert np.array_equal(kv.attn_history_denom.cpu()[0].numpy(), st["denom"]) and int(kv.attn_counter) == 1
    seed_gap = np.abs(ring_d - ring_o)
    # cum_attn = dtype(dtype(column sum) / (L - pos)), cache.py:1155: a last-bit difference of the fp32 sum may move the rounded sum
    # by one bf16 step and the quotient by another — two steps, 2^-6 relative at the coarse end of a binade (measured on these
    # heads: 0.03 % of the seeds differ at all, the largest by 1.2 %, a column whose fp32 sum differs by 1.2 %: one flipped
    # rounding of a score of magnitude ~40, attention_utils.py:37)
    assert bool((seed_gap <= 2.0 ** -6 * np.abs(ring_o) + 1e-30).all()), "ring seed beyond two roundings of the column means"
    assert (seed_gap > 0).mean() < 0.002
    # ---- decode, each side on its own ring
    tab = policy_table(HYBRID, S)
    key = np.zeros(8 * H * ((S + 127) // 128), np.uint64)
    gen = torch.Generator().manual_seed(21)
    hh_heads = [h for h in range(H) if tab[strat_d[h], 0] & 1]
    win_heads = [h for h in range(H) if (tab[strat_d[h], 0] & 3) == 2]
    full_heads = [h for h in range(H) if tab[strat_d[h], 0] & 16]
    assert hh_heads and win_heads and full_heads
    reseated = 0
    evict_hh = evict_win = 0
    for t in range(steps):
        p = torch.tensor([L + t], dtype=torch.int32)
        k1 = (1.5 * torch.randn(1, H, 1, D, generator=gen)).to(dtype)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        q1 = (1.5 * torch.randn(1, HQ, 1, D, generator=gen)).to(dtype)
        pos_b, cts_b = st["pos"].copy(), st["cts"].copy()
        dn = np.minimum(np.maximum(st["denom"], 1), W).astype(np.float32)
        score_b = st["wsum"].reshape(H, S) / dn  # the oracle's heavy-hitter scores for this position (cache.py:844-894)
        yd = kv.decode_step(q1.to(DEV), k1.to(DEV), v1.to(DEV), p.to(DEV))
        torch.cuda.synchronize()
        view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], code)
        yo1 = np.zeros((HQ, D), np.uint16)
        o.call("cc_decode_step_hybrid", C.byref(view), o.ptr(to_np(q1.reshape(HQ, D))), o.ptr(to_np(k1.reshape(H, D))),
               o.ptr(to_np(v1.reshape(H, D))), o.ptr(p.numpy().copy()), o.ptr(strat_d.copy()), o.ptr(tab), len(tab), o.ptr(st["num"]),
               o.ptr(st["denom"]), o.ptr(st["ctr"]), W, o.ptr(st["acc"]), o.ptr(st["wsum"]), None, None, None, None, 0, None, None,
               o.ptr(key), g, HQ, 1.0 / math.sqrt(D), o.ptr(yo1), None, None, 0, None)
        pos_d, cts_d = kv.pos.cpu()[0].numpy(), kv.cache_cts.cpu().numpy()
        assert np.array_equal(cts_d, st["cts"]), f"step {t}: counts"
        for h in full_heads:
            assert cts_d[h] == cts_b[h] + 1
        evict_hh += sum(int(cts_d[h] == cts_b[h]) for h in hh_heads)
        evict_win += sum(int(cts_d[h] == cts_b[h]) for h in win_heads)
        if not np.array_equal(pos_d, st["pos"]):
            for h in range(H):
                if np.array_equal(pos_d[h], st["pos"][h]):
                    continue
                assert h in hh_heads, f"step {t} head {h}: a position-ordered decision differs"
                i_d = int(np.nonzero(pos_d[h] != pos_b[h])[0][0])
                i_o = int(np.nonzero(st["pos"][h] != pos_b[h])[0][0])
                gap = float(score_b[h, i_d] - score_b[h, i_o])
                assert 0 <= gap <= 2 * BF16_ULP * float(score_b[h, i_o]) + 1e-30, f"step {t} head {h}: evicted {i_d} vs {i_o} (score gap {gap})"
            reseated += 1  # an equally good choice: the oracle continues from the device's state (the only copy across)
            st.update(k=to_np(kv.k_cache.cpu()[0]), v=to_np(kv.v_cache.cpu()[0]), pos=pos_d.copy(),
                      mask=kv.mask.cpu()[0, :, 0].numpy().astype(np.uint8), num=to_np(kv.attn_history_num.cpu()[0]),
                      denom=kv.attn_history_denom.cpu()[0].numpy().copy())
            o.call("cc_hh_ring_window_sums", o.ptr(st["num"]), H, S, W, code, o.ptr(st["wsum"]), o.ptr(st["acc"]), None)
            continue
        yr = from_np(yo1, dtype).float()
        assert (yd.cpu().float()[0, :, 0] - yr).abs().max() <= 1e-3 + 2 * BF16_ULP * yr.abs().max(), f"step {t}: y"
    audit(f"n_just = {reseated} near-tie candidate re-seats in {steps} steps x {H} heads (limit 2)", rule="near-tie candidate re-seat", count=reseated, compared=steps * H, limit="2 per run")
    assert reseated <= 2, f"{reseated} near-tie divergences in {steps} steps"
    assert evict_hh >= len(hh_heads) * (steps - 2) and evict_win >= len(win_heads) * (steps - 2)  # the budgets were full: real evictions
    assert np.array_equal(kv.attn_history_denom.cpu()[0].numpy(), st["denom"]) and int(kv.attn_counter) == int(st["ctr"][0])
    assert np.array_equal(kv.mask.cpu()[0, :, 0].numpy().astype(np.uint8), st["mask"])
    assert np.array_equal(to_np(kv.k_cache.cpu()[0]), st["k"])
    ring_d = kv.attn_history_num.cpu()[0].float().numpy()
    ring_o = from_np(st["num"], dtype).float().numpy()
    assert bool((np.abs(ring_d - ring_o) <= 2.0 ** -6 * np.abs(ring_o) + 1e-6).all())
    from cold_compress_amd.attention_utils import single_launch_status

    assert single_launch_status(kv.pos.device) == 0
