"""Pin the CPU oracle (oracle/cc_oracle.c) against golden vectors captured from the reference itself
(oracle/gen_golden.py imports /root/reference; see tests/golden/).  CPU-only.

Bit-exact: eviction indices, positions, masks, counters, float64/int32 heavy-hitter history, K/V cache
contents, top-k keep sets (no boundary tie) and gathered rows.  Tolerance (1e-3, stated per test):
attention outputs/probabilities.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch

from helpers import DT_CODE, DT_FROM_NAME, GOLDEN, from_np, load_golden, to_np


def _i32(x):
    return np.array([x], dtype=np.int32)


class OracleCache:
    """numpy-side state mirroring the reference nn.Module buffers, driven through the oracle's C ABI."""

    def __init__(self, oracle, H, S, D, dtype, head_specific, strategy):
        self.o, self.H, self.S, self.D, self.dtype = oracle, H, S, D, dtype
        self.code = DT_CODE[dtype]
        es = np.float32 if dtype == torch.float32 else np.uint16
        self.k = np.zeros((H, S, D), es)
        self.v = np.zeros((H, S, D), es)
        self.pos = np.full((H if head_specific else 1, S), -1, np.int32)
        self.mask = np.zeros((H, S), np.uint8)
        self.cts = np.zeros((1,), np.int32)
        self.num = np.zeros((H, S), np.float64)
        self.denom = np.zeros((H, S), np.int32)
        self.counter = np.zeros((1,), np.int64)
        self.key_norm = np.zeros((H, S), es)
        self.strategy = strategy

    def view(self):
        return self.o.view(self.k, self.v, self.pos, self.mask, self.cts, self.code)

    def prefill(self, k0, v0, pos0):
        T = k0.shape[-2]
        k0n, v0n = to_np(k0[0]), to_np(v0[0])
        p = pos0.numpy().astype(np.int64).reshape(1, T).copy()
        self.o.call("cc_prefill_fill", C.byref(self.view()), self.o.ptr(k0n), self.o.ptr(v0n), self.o.ptr(p), 1, T, None)

    def decode(self, p, k1, v1, g, w, rand_u=None, scores=None, score_code=None):
        kn, vn = to_np(k1.reshape(self.H, self.D)), to_np(v1.reshape(self.H, self.D))
        idx = np.zeros((self.pos.shape[0],), np.int64)
        pp = _i32(p)
        o, vw = self.o, self.view()
        if self.strategy == "heavy_hitter":
            o.call("cc_decode_update_heavy_hitter", C.byref(vw), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(self.num),
                   o.ptr(self.denom), g, w, o.ptr(idx), None)
        elif self.strategy == "l2":
            o.call("cc_decode_update_l2", C.byref(vw), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(self.key_norm), g, w,
                   o.ptr(idx), None, 0, None)
        elif self.strategy == "random":
            r = rand_u.numpy().astype(np.float32).copy()
            o.call("cc_decode_update_random", C.byref(vw), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(r), g, w, o.ptr(idx), None)
        elif self.strategy == "full":
            o.call("cc_decode_update_full", C.byref(vw), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(idx), None)
        elif self.strategy == "recent_global":
            o.call("cc_decode_update_recent_global", C.byref(vw), o.ptr(kn), o.ptr(vn), o.ptr(pp), g, o.ptr(idx), None)
        elif self.strategy == "scores":
            o.call("cc_decode_update_scores", C.byref(vw), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(scores), score_code, g,
                   o.ptr(idx), None)
        else:
            raise AssertionError(self.strategy)
        return idx


def _check_final(c, f, dtype):
    assert np.array_equal(c.pos, f["final_pos"][0].numpy())
    assert np.array_equal(c.mask.astype(bool), f["final_mask"][0, :, 0].numpy())
    assert np.array_equal(c.cts, f["final_cts"].numpy())
    assert np.array_equal(c.k, to_np(f["final_k"][0]))
    assert np.array_equal(c.v, to_np(f["final_v"][0]))


def _hh_prefill_state(oracle, c, attn0, T):
    """ref: cache.py:700-723 — 4-D prefill attention -> column mean -> history."""
    a = to_np(attn0[0])  # [H,T,T]
    colsum = np.zeros((c.H, T), np.float32)
    oracle.call("cc_attn_colsum", oracle.ptr(a), c.H, T, T, c.code, oracle.ptr(colsum), None)
    mean = np.zeros((c.H, T), np.float32 if c.dtype == torch.float32 else np.uint16)
    oracle.call("cc_colsum_to_mean", oracle.ptr(colsum), None, c.H, T, c.code, oracle.ptr(mean), None)
    oracle.call("cc_hh_update", oracle.ptr(c.num), oracle.ptr(c.denom), oracle.ptr(c.counter), oracle.ptr(mean), c.H, c.S, T,
                c.code, None)


@pytest.mark.parametrize("name", ["f2_hh_f32.npz", "f2_hh_bf16.npz", "f2_hh_h1_bf16.npz", "f2_hh_long_bf16.npz"])
def test_heavy_hitter_replay_bit_exact(oracle, name):
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    c = OracleCache(oracle, H, S, D, dtype, True, "heavy_hitter")
    c.prefill(f["k0"], f["v0"], torch.arange(T))
    _hh_prefill_state(oracle, c, f["attn0"], T)
    # The column sum's fp32 summation order inside torch.sum is unspecified (SURVEY §7 hazard (ii)): the
    # prefill history is tolerance-class (1 ulp of the model dtype); decode then continues from the
    # reference's state so that everything after is compared bit-exactly on identical state.
    ref_num = f["num_after_prefill"][0, :, :, 0].numpy()
    assert np.allclose(c.num, ref_num, rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)
    assert np.array_equal(c.denom, f["denom_after_prefill"][0].numpy())
    c.num = ref_num.copy()
    for t in range(f["steps"]):
        idx = c.decode(T + t, f["k_new"][t], f["v_new"][t], g, w)
        assert np.array_equal(idx, f["idx"][t].numpy()), f"step {t}"
        assert np.array_equal(c.cts, f["cache_cts_steps"][t].numpy())
        a = to_np(f["attn"][t][0, :, 0])
        oracle.call("cc_hh_update", oracle.ptr(c.num), oracle.ptr(c.denom), oracle.ptr(c.counter), oracle.ptr(a), H, S, S,
                    c.code, None)
    assert np.array_equal(c.num, f["final_num"][0, :, :, 0].numpy())
    assert np.array_equal(c.denom, f["final_denom"][0].numpy())
    assert c.counter[0] == int(f["final_counter"][0])
    _check_final(c, f, dtype)


@pytest.mark.parametrize("name", ["f3_l2_bf16.npz", "f3_l2_f32.npz", "f3_l2_h1_bf16.npz", "f3_l2_long_bf16.npz"])
def test_l2_replay_bit_exact(oracle, name):
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    c = OracleCache(oracle, H, S, D, dtype, True, "l2")
    c.prefill(f["k0"], f["v0"], torch.arange(T))
    # ref: KVCacheL2.update_state cache.py:611-612
    oracle.call("cc_row_l2_norm", oracle.ptr(c.k), H, S, D, c.code, 0, oracle.ptr(c.key_norm), None)
    ref_kn = f["keynorm_after_prefill"][0]
    got = from_np(c.key_norm, dtype)
    # fp32 summation order of the reference's vectorised norm is not specified: allow 1 ulp of the dtype
    assert torch.allclose(got.float(), ref_kn.float(), rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)
    c.key_norm = to_np(ref_kn)  # continue from the reference's norms so indices are compared on equal state
    for t in range(f["steps"]):
        idx = c.decode(T + t, f["k_new"][t], f["v_new"][t], g, w)
        assert np.array_equal(idx, f["idx"][t].numpy()), f"step {t}"
    _check_final(c, f, dtype)
    got = from_np(c.key_norm, dtype).float()
    assert torch.allclose(got, f["final_keynorm"][0].float(), rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)


def test_random_replay_bit_exact(oracle):
    f = load_golden("f4_random.npz")
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    c = OracleCache(oracle, H, S, D, dtype, False, "random")
    c.prefill(f["k0"], f["v0"], torch.arange(T))
    for t in range(f["steps"]):
        idx = c.decode(T + t, f["k_new"][t], f["v_new"][t], g, w, rand_u=f["rand_u"][t])
        assert np.array_equal(idx, f["idx"][t].numpy().reshape(-1)), f"step {t}"
    _check_final(c, f, dtype)


@pytest.mark.parametrize("strategy", ["full", "recent_global", "keep_it_odd"])
def test_head_constant_replay_bit_exact(oracle, strategy):
    z = load_golden("f4_headconst.npz")
    f = {k[len(strategy) + 1:]: v for k, v in z.items() if k.startswith(strategy + ".")}
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    c = OracleCache(oracle, H, S, D, dtype, False, "scores" if strategy == "keep_it_odd" else strategy)
    if strategy == "full":
        g = 0
    c.prefill(f["k0"], f["v0"], torch.arange(T))
    for t in range(f["steps"]):
        kw = {}
        if strategy == "keep_it_odd":
            # ref: KVCacheKeepItOdd._token_importances cache.py:1437-1441 (bf16 scores), via the generic path
            p = T + t
            pos = torch.from_numpy(c.pos[0])
            sc = torch.zeros(S, dtype=torch.bfloat16)
            sc[pos % 2 == 1] = 1.0
            sc[pos >= p - w] = float("inf")
            kw = dict(scores=to_np(sc), score_code=1)
        idx = c.decode(T + t, f["k_new"][t], f["v_new"][t], g, w, **kw)
        assert np.array_equal(idx, f["idx"][t].numpy().reshape(-1)), f"step {t}"
    _check_final(c, f, dtype)


def test_recent_global_ring_known_answer(oracle):
    """with S=16, g=4 the slot at decode step t is 4 + (t mod 12) once the cache is full."""
    f = load_golden("f1_e2e_recent_global.npz")
    idx = f["evict_idx_L0"].numpy().reshape(-1)
    # prompt 40 > S=16: the cache is full after prefill compaction; decode step t overwrites the oldest window slot
    assert list(idx) == [4 + (t % 12) for t in range(len(idx))]
    c = OracleCache(oracle, 2, 16, 16, torch.float32, False, "recent_global")
    keep = list(range(4)) + list(range(40 - 12, 40))
    c.pos[0, :] = np.array(keep, np.int32)
    c.mask[:] = 1
    k = torch.zeros(1, 2, 1, 16)
    got = [int(c.decode(40 + t, k, k, 4, 0)[0]) for t in range(len(idx))]
    assert got == list(idx)
    assert np.array_equal(c.pos, f["final_pos_L0"][0].numpy())


# ----------------------------------------------------------------------------------- compaction (F5)


def _prio_code(t):
    return {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.int64: 3}[t.dtype]


def _tie_class_ok(prio_row, keep, ref_keep, K):
    """SURVEY §8(c) contract (2): every strictly-better element present, size K, remainder from the tie class."""
    v = prio_row.double()
    kth = v.sort(descending=True).values[K - 1]
    better = set(torch.nonzero(v > kth).view(-1).tolist())
    tie = set(torch.nonzero(v == kth).view(-1).tolist())
    for ks in (set(keep.tolist()), set(ref_keep.tolist())):
        assert len(ks) == K and better <= ks and ks <= (better | tie)


def test_topk_keep_and_gather(oracle):
    f = load_golden("f5_compress.npz")
    cases = [c for c in np.load(os.path.join(GOLDEN, "f5_compress.npz"))["cases"]]
    assert len(cases) >= 11
    for name in cases:
        prio = f[name + ".priority"]
        ref_keep = f[name + ".keep"]
        K = ref_keep.shape[-1]
        p2 = prio.reshape(-1, prio.shape[-1])
        Hs, L = p2.shape
        keep = np.zeros((Hs, K), np.int64)
        pn = to_np(p2)
        oracle.call("cc_topk_keep", oracle.ptr(pn), _prio_code(prio), Hs, L, K, oracle.ptr(keep), None, 0, None)
        rk = ref_keep.reshape(Hs, K)
        tie = bool(np.load(os.path.join(GOLDEN, "f5_compress.npz"))[name + ".tie"])
        if not tie:
            assert np.array_equal(keep, rk.numpy()), name
        for h in range(Hs):
            assert np.all(np.diff(keep[h]) > 0)
            _tie_class_ok(p2[h], torch.from_numpy(keep[h]), rk[h], K)
        # gather with the REFERENCE's keep so K/V are compared on equal indices
        k_in, k_out = f[name + ".k_in"][0], f[name + ".k_out"][0]
        H, _, D = k_in.shape
        dst = np.zeros_like(to_np(k_out))
        rkn = rk.numpy().astype(np.int64).copy()
        oracle.call("cc_gather_rows", oracle.ptr(to_np(k_in)), oracle.ptr(rkn), Hs, H, L, K, D, DT_CODE[k_in.dtype],
                    oracle.ptr(dst), None)
        assert np.array_equal(dst, to_np(k_out)), name


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_l2_and_snapkv_priorities(oracle, tag):
    f = load_golden("f5_compress.npz")
    # L2: priority = -||k|| with recent/global -> +inf (prompt_compression.py:201-209)
    k = f[f"l2_{tag}.k_in"][0]
    H, L, D = k.shape
    code = DT_CODE[k.dtype]
    out = np.zeros((H, L), np.float32 if code == 0 else np.uint16)
    oracle.call("cc_row_l2_norm", oracle.ptr(to_np(k)), H, L, D, code, 1, oracle.ptr(out), None)
    got = from_np(out, k.dtype).float()
    ref = f[f"l2_{tag}.priority"][0].float()
    finite = torch.isfinite(ref)
    assert torch.allclose(got[finite], ref[finite], rtol=2 ** -7 if code else 1e-6, atol=0)
    assert bool((~finite[:, :4]).all()) and bool((~finite[:, -10:]).all())
    # SnapKV: mean of last 16 rows -> avgpool5 -> forced ones (prompt_compression.py:170-187)
    attn = f[f"heavy_hitter_{tag}.attn"]
    obs = attn[:, :, -16:, :].mean(dim=2)[0]
    out = np.zeros((H, L), np.float32 if code == 0 else np.uint16)
    oracle.call("cc_snapkv_priority", oracle.ptr(to_np(obs)), H, L, code, 16, 4, oracle.ptr(out), None)
    got = from_np(out, k.dtype).float()
    ref = f[f"heavy_hitter_{tag}.priority"][0].float()
    assert torch.allclose(got, ref, rtol=2 ** -7 if code else 1e-6, atol=1e-7), (got - ref).abs().max()
    # SnapKV state: column mean gathered at keep (prompt_compression.py:189-194)
    colsum = np.zeros((H, L), np.float32)
    oracle.call("cc_attn_colsum", oracle.ptr(to_np(attn[0])), H, L, L, code, oracle.ptr(colsum), None)
    mean = np.zeros((H, L), np.float32 if code == 0 else np.uint16)
    oracle.call("cc_colsum_to_mean", oracle.ptr(colsum), None, H, L, code, oracle.ptr(mean), None)
    keep = f[f"heavy_hitter_{tag}.keep"].numpy().astype(np.int64).copy()
    K = keep.shape[-1]
    st = np.zeros((H, K), mean.dtype)
    oracle.call("cc_gather_vec", oracle.ptr(mean), oracle.ptr(keep), H, L, K, code, oracle.ptr(st), None)
    got = from_np(st, k.dtype).float()
    ref = f[f"heavy_hitter_{tag}.state"][0].float()
    assert torch.allclose(got, ref, rtol=2 ** -6 if code else 1e-5, atol=1e-6)


# ----------------------------------------------------------------------------------- attention (F7)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("case", ["dec", "dec8b"])
def test_decode_attention_within_1e3(oracle, tag, case):
    f = load_golden(f"f7_attn_{tag}.npz")
    q, k, v, mask = f[case + ".q"], f[case + ".k"], f[case + ".v"], f[case + ".mask"]
    HQ, D = q.shape[1], q.shape[3]
    H, S = k.shape[1], k.shape[2]
    code = DT_CODE[q.dtype]
    es = np.float32 if code == 0 else np.uint16
    y = np.zeros((HQ, D), es)
    attn = np.zeros((H, S), es)
    probs = np.zeros((HQ, S), es)
    oracle.call("cc_decode_attn_gqa", oracle.ptr(to_np(q[0, :, 0])), oracle.ptr(to_np(k[0])), oracle.ptr(to_np(v[0])),
                oracle.ptr(to_np(mask[0, :, 0])), HQ, H, S, D, code, 1.0 / np.sqrt(D), oracle.ptr(y), oracle.ptr(attn),
                oracle.ptr(probs), None, None, None, None, 0, None)
    tol = 1e-3  # north-star tolerance; bf16 outputs compared after upcasting (SURVEY §8(c) contract (3))
    if code:
        tol = 8e-3  # one bf16 ulp at |y|~1 is 7.8e-3: the reference's own two paths differ by that much
    assert (from_np(y, q.dtype).float() - f[case + ".y"][0, :, 0].float()).abs().max() < tol
    assert (from_np(probs, q.dtype).float() - f[case + ".probs"][0, :, 0].float()).abs().max() < 1e-3
    assert (from_np(attn, q.dtype).float() - f[case + ".attn_gm"][0, :, 0].float()).abs().max() < 1e-3


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_prefill_attention_within_1e3(oracle, tag):
    f = load_golden(f"f7_attn_{tag}.npz")
    q, k, v = f["pre.q"], f["pre.k"], f["pre.v"]
    HQ, L, D = q.shape[1:]
    H = k.shape[1]
    code = DT_CODE[q.dtype]
    es = np.float32 if code == 0 else np.uint16
    y = np.zeros((HQ, L, D), es)
    colsum = np.zeros((H, L), np.float32)
    obs = np.zeros((H, L), np.float32)
    oracle.call("cc_prefill_attn", oracle.ptr(to_np(q[0])), oracle.ptr(to_np(k[0])), oracle.ptr(to_np(v[0])), HQ, H, L, D,
                code, 1.0 / np.sqrt(D), oracle.ptr(y), oracle.ptr(colsum), oracle.ptr(obs), 16, None, 0, None)
    tol = 8e-3 if code else 1e-3
    assert (from_np(y, q.dtype).float() - f["pre.y"][0].float()).abs().max() < tol
    assert (torch.from_numpy(colsum) - f["pre.colsum"][0].float()).abs().max() < (6e-2 if code else 1e-3)
    assert (torch.from_numpy(obs) - f["pre.obs_mean"][0].float()).abs().max() < (4e-3 if code else 1e-3)


def test_budget_fixture_is_readable():
    with open(os.path.join(GOLDEN, "f8_budgets.json")) as fh:
        rows = json.load(fh)
    assert rows["normalize"][0] == [0.25, 10240, 2560]
    assert rows["normalize"][1] == [0.1, 34816, 3488]


@pytest.mark.parametrize("name", ["f10_analysis_hh_f32.npz", "f10_analysis_hh_bf16.npz"])
def test_analysis_loss_matches_reference_capture(oracle, name):
    """cc_analysis_loss_cpu (KVCacheAnalysis' decode-time gather + attention-loss record, cache.py:1391-1404) on the reference's
    own trace: the attention row over the full cache and the shadow cache's positions of every step -> the recorded loss
    (to one rounding of the dtype: torch.sum's fp32 order is its own) and the counter."""
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    code = DT_CODE[dtype]
    H, S, SF, steps = f["H"], f["S"], f["S_full"], f["steps"]
    losses = to_np(torch.full((SF,), -1.0).to(dtype))
    ctr = np.zeros(1, np.int32)
    for t in range(steps):
        attn = to_np(f["attn"][t].reshape(-1, SF)[:H])
        pos = f["comp_pos_steps"][t].numpy().astype(np.int32).reshape(H, S).copy()
        sub = np.zeros((H, S), attn.dtype)
        oracle.call("cc_analysis_loss", oracle.ptr(attn), oracle.ptr(pos), H, SF, S, code, oracle.ptr(sub), oracle.ptr(losses), oracle.ptr(ctr),
                    SF, None)
        got = float(from_np(losses, dtype)[t])
        want = float(f["loss_steps"][t])
        ulp = 2 ** -7 if dtype != torch.float32 else 1e-6
        assert abs(got - want) <= ulp * max(1.0, abs(want)), f"step {t}: {got} vs {want}"
        filled = pos != -1
        a32 = from_np(attn, dtype).float().numpy()
        s32 = from_np(sub, dtype).float().numpy()
        for h in range(H):
            assert np.array_equal(s32[h][filled[h]], a32[h][pos[h][filled[h]]])
            assert np.array_equal(s32[h][~filled[h]], np.full((~filled[h]).sum(), a32[h][SF - 1]))
    assert int(ctr[0]) == steps == int(f["loss_ctr"])


def test_in_kernel_generator_restatement():
    """KVCacheRandom's in-kernel generator (cc_rng_uniform, include/coldcompress.h): the oracle's C restatement against a numpy
    restatement of the published formula, plus the properties the policy needs — values in [0, 1) on the 2^-24 grid, different
    positions / seeds give different vectors, near-uniform mean."""
    from oracle import oracle_lib

    def mix(x):
        x = np.uint64(x)
        with np.errstate(over="ignore"):
            x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd); x ^= x >> np.uint64(33)
            x *= np.uint64(0xc4ceb9fe1a85ec53); x ^= x >> np.uint64(33)
        return x

    def ref(seed, pos, S):
        with np.errstate(over="ignore"):
            x = np.uint64(seed) + np.uint64(pos) * np.uint64(0x9E3779B97F4A7C15) + np.arange(S, dtype=np.uint64)
        x = mix(mix(x))
        return ((x >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)

    for seed, pos, S in [(0, 0, 64), (1234567890123456789, 4095, 4096), (2 ** 62 - 1, 2 ** 31 - 2, 257)]:
        u = oracle_lib.rng_vector(seed, pos, S)
        assert np.array_equal(u, ref(seed, pos, S))
        assert u.min() >= 0.0 and u.max() < 1.0
        assert np.array_equal(u * 2 ** 24, np.round(u * 2 ** 24))
    u = oracle_lib.rng_vector(7, 100, 4096)
    assert abs(float(u.mean()) - 0.5) < 0.02 and len(np.unique(u)) > 4000
    assert not np.array_equal(u, oracle_lib.rng_vector(7, 101, 4096))
    assert not np.array_equal(u, oracle_lib.rng_vector(8, 100, 4096))


def _pipeline_replay(oracle, c, f, T, g, w, strategy, entry):
    """The oracle's FUSED-step twins (what the device's single- and two-launch steps are checked against) driven by a reference
    capture: seed the key row, then one step per token — the slot each step fills must be the reference's eviction index."""
    o = oracle
    H, S, D = c.H, c.S, c.D
    nk = int(o.fns()["cc_hh_next_key_slots"](S))
    key = np.full((H, nk), ~np.uint64(0), np.uint64)
    commit = np.full((H, 68), -1, np.int32)  # (include/coldcompress.h: insert word, its position, one word per workgroup, the hybrid step's two)
    steps = f["steps"]
    rand = [f["rand_u"][t].numpy().astype(np.float32).copy() for t in range(steps)] if strategy == "random" else None
    p0 = _i32(T)
    if strategy == "random":
        o.call("cc_random_next_key_init", C.byref(c.view()), o.ptr(p0), o.ptr(rand[0]), g, w, o.ptr(key), None)
    else:
        o.call("cc_rg_next_key_init", C.byref(c.view()), o.ptr(p0), g, o.ptr(key), None)
    HQ = H
    ws = np.zeros(int(o.fns()["cc_decode_attn_workspace_bytes"](HQ, H, S, D, c.code)), np.uint8)
    q = np.zeros((HQ, D), c.k.dtype)
    y = np.zeros((HQ, D), c.k.dtype)
    for t in range(steps):
        before = c.pos.copy()
        kn, vn = to_np(f["k_new"][t].reshape(H, D)), to_np(f["v_new"][t].reshape(H, D))
        nxt = rand[t + 1] if strategy == "random" and t + 1 < steps else (np.zeros(S, np.float32) if strategy == "random" else None)
        pp = _i32(T + t)
        if entry == "rc":
            o.call("cc_decode_step_head_constant_rc", C.byref(c.view()), 3 if strategy == "random" else 2, o.ptr(q), o.ptr(kn), o.ptr(vn),
                   o.ptr(pp), o.ptr(nxt) if nxt is not None else None, 0, o.ptr(key), o.ptr(commit), g, w, HQ, 0.25, o.ptr(y), o.ptr(ws),
                   ws.size, None)
            assert bool((commit[:, 1:3] == T + t).all()) and bool((commit[:, 3:] == -1).all())
        elif strategy == "random":
            o.call("cc_decode_step_random", C.byref(c.view()), o.ptr(q), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(nxt), o.ptr(key), g, w, HQ,
                   0.25, o.ptr(y), o.ptr(ws), ws.size, None)
        else:
            o.call("cc_decode_step_recent_global", C.byref(c.view()), o.ptr(q), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(key), g, HQ, 0.25,
                   o.ptr(y), o.ptr(ws), ws.size, None)
        changed = np.nonzero(c.pos.reshape(-1) != before.reshape(-1))[0]
        assert np.array_equal(changed, f["idx"][t].numpy().reshape(-1)), f"step {t}: filled slot"


@pytest.mark.parametrize("entry", ["plain", "rc"])
@pytest.mark.parametrize("strategy", ["random", "recent_global", "full"])
def test_fused_pipeline_twins_replay_the_reference(oracle, strategy, entry):
    """f4 captures (the reference's own draws, evictions and final buffers) through the oracle's fused-step twins — the plain ones
    and the recoverable head-constant entry (cc_decode_step_head_constant_rc): same slots step by step, same final cache."""
    if strategy == "random":
        f = load_golden("f4_random.npz")
    else:
        z = load_golden("f4_headconst.npz")
        f = {k[len(strategy) + 1:]: v for k, v in z.items() if k.startswith(strategy + ".")}
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    if strategy == "full":
        g = 0
    c = OracleCache(oracle, H, S, D, dtype, False, strategy)
    c.prefill(f["k0"], f["v0"], torch.arange(T))
    _pipeline_replay(oracle, c, f, T, g, w, strategy, entry)
    _check_final(c, f, dtype)


def test_in_kernel_draw_step_twin_equals_the_vector_step(oracle):
    """cc_decode_step_random_rng_cpu / cc_random_next_key_init_rng_cpu == the vector forms fed oracle_lib.rng_vector(seed, position):
    the twin the device's in-kernel draws are checked against is the reference-pinned random step with a restated generator."""
    from oracle import oracle_lib

    f = load_golden("f4_random.npz")
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    o, seed = oracle, 0x1234567812345
    a = OracleCache(oracle, H, S, D, dtype, False, "random")
    b = OracleCache(oracle, H, S, D, dtype, False, "random")
    for c in (a, b):
        c.prefill(f["k0"], f["v0"], torch.arange(T))
    nk = int(o.fns()["cc_hh_next_key_slots"](S))
    ka, kb = np.zeros((H, nk), np.uint64), np.zeros((H, nk), np.uint64)
    o.call("cc_random_next_key_init_rng", C.byref(a.view()), o.ptr(_i32(T)), seed, g, w, o.ptr(ka), None)
    o.call("cc_random_next_key_init", C.byref(b.view()), o.ptr(_i32(T)), o.ptr(oracle_lib.rng_vector(seed, T, S)), g, w, o.ptr(kb), None)
    assert np.array_equal(ka[0], kb[0])
    ws = np.zeros(int(o.fns()["cc_decode_attn_workspace_bytes"](H, H, S, D, a.code)), np.uint8)
    q = to_np(torch.randn(H, D, generator=torch.Generator().manual_seed(1)).to(dtype))
    ya, yb = np.zeros_like(q), np.zeros_like(q)
    for t in range(min(40, f["steps"])):
        kn, vn = to_np(f["k_new"][t].reshape(H, D)), to_np(f["v_new"][t].reshape(H, D))
        pp = _i32(T + t)
        o.call("cc_decode_step_random_rng", C.byref(a.view()), o.ptr(q), o.ptr(kn), o.ptr(vn), o.ptr(pp), seed, o.ptr(ka), g, w, H, 0.25,
               o.ptr(ya), o.ptr(ws), ws.size, None)
        o.call("cc_decode_step_random", C.byref(b.view()), o.ptr(q), o.ptr(kn), o.ptr(vn), o.ptr(pp), o.ptr(oracle_lib.rng_vector(seed, T + t + 1, S)),
               o.ptr(kb), g, w, H, 0.25, o.ptr(yb), o.ptr(ws), ws.size, None)
        assert np.array_equal(a.pos, b.pos) and np.array_equal(ya, yb) and np.array_equal(ka[0], kb[0]), f"step {t}"
    assert np.array_equal(a.k, b.k) and np.array_equal(a.v, b.v) and np.array_equal(a.mask, b.mask)


@pytest.mark.parametrize("name", ["f3_l2_h1_bf16.npz", "f3_l2_long_bf16.npz"])  # (the fused l2 step serves 16-bit caches with head_dim 128)
def test_l2_fused_pipeline_twin_replays_the_reference(oracle, name):
    """The oracle's fused l2 step (cc_l2_next_key_init + cc_decode_step_l2: what the device's l2 layer step is checked against)
    through the reference's f3 captures, continuing from the reference's prefill norms: the slot every step fills is the
    reference's eviction index per head, the final cache and norms equal the reference's."""
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    o = oracle
    c = OracleCache(oracle, H, S, D, dtype, True, "l2")
    c.prefill(f["k0"], f["v0"], torch.arange(T))
    c.key_norm = to_np(f["keynorm_after_prefill"][0])
    nk = int(o.fns()["cc_hh_next_key_slots"](S))
    key = np.zeros((H, nk), np.uint64)
    o.call("cc_l2_next_key_init", C.byref(c.view()), o.ptr(_i32(T)), o.ptr(c.key_norm), g, w, o.ptr(key), None)
    ws = np.zeros(int(o.fns()["cc_decode_attn_workspace_bytes"](H, H, S, D, c.code)), np.uint8)
    q = to_np(torch.randn(H, D, generator=torch.Generator().manual_seed(2)).to(dtype))
    y = np.zeros_like(q)
    for t in range(f["steps"]):
        before = c.pos.copy()
        kn, vn = to_np(f["k_new"][t].reshape(H, D)), to_np(f["v_new"][t].reshape(H, D))
        o.call("cc_decode_step_l2", C.byref(c.view()), o.ptr(q), o.ptr(kn), o.ptr(vn), o.ptr(_i32(T + t)), o.ptr(c.key_norm), o.ptr(key), g, w,
               H, 1.0 / math.sqrt(D), o.ptr(y), o.ptr(ws), ws.size, None)
        filled = np.array([int(np.nonzero(c.pos[h] != before[h])[0][0]) for h in range(H)])
        assert np.array_equal(filled, f["idx"][t].numpy().reshape(-1)), f"step {t}: filled slots"
    _check_final(c, f, dtype)
    got = from_np(c.key_norm, dtype).float()
    assert torch.allclose(got, f["final_keynorm"][0].float(), rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)
