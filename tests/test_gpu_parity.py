"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, driven through the reference-style
Python classes (-> ctypes -> C ABI -> kernels), against (a) golden vectors captured from the reference and
(b) the CPU oracle on seeded inputs.

Bit-exact: eviction indices, pos/mask/cache_cts, K/V contents, float64/int32 history given identical
inputs, keep sets without boundary ties, gathered rows, row norms vs the oracle.
Tolerance 1e-3 (fp32) / one bf16 ulp (bf16 outputs): attention outputs and probabilities.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from helpers import DT_CODE, DT_FROM_NAME, from_np, load_golden, to_np

pytestmark = pytest.mark.gpu

DEV = __import__("helpers").TEST_DEVICE  # "cuda"; "cpu" only under tests/cpu_twin.py


@pytest.fixture(scope="module")
def cc():
    import cold_compress_amd.cache as cache
    from cold_compress_amd import _abi

    fns = _abi.lib()  # raises loudly if the HIP extension is missing
    assert fns["cc_abi_version"]() == 1
    return cache


def _kw(S, g, w):
    return dict(max_cache_length=S, global_tokens=g, max_seq_length=4 * S, cache_bits=None, recent_window=w,
                history_window_size=1, attn_thresholding=False)


def _make(cc, strategy, dtype, H, S, D, g, w):
    cls, rk = cc.get_cache_constructor(strategy)
    kw = _kw(S, g, w)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{k: kw[k] for k in rk})
    return kv


def _idx(kv):
    torch.cuda.synchronize()
    return kv._idx_buf().cpu().numpy().copy()


def test_library_and_device(cc):
    from cold_compress_amd import _abi

    n_cu, wave, lds = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(64)
    _abi.call("cc_device_info", C.byref(n_cu), C.byref(wave), C.byref(lds), name, 64)
    assert wave.value == 64 and n_cu.value >= 200 and b"gfx950" in name.value


def _final_equal(kv, f):
    torch.cuda.synchronize()
    assert torch.equal(kv.pos.cpu(), f["final_pos"])
    assert torch.equal(kv.mask.cpu(), f["final_mask"])
    assert torch.equal(kv.cache_cts.cpu(), f["final_cts"])
    assert torch.equal(kv.k_cache.cpu().view(torch.int16) if kv.k_cache.dtype != torch.float32 else kv.k_cache.cpu(),
                       f["final_k"].view(torch.int16) if kv.k_cache.dtype != torch.float32 else f["final_k"])
    assert torch.equal(kv.v_cache.cpu().float(), f["final_v"].float())


@pytest.mark.parametrize("name", ["f2_hh_f32.npz", "f2_hh_bf16.npz", "f2_hh_h1_bf16.npz", "f2_hh_long_bf16.npz"])
def test_heavy_hitter_replay_vs_reference(cc, name):
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    kv = _make(cc, "heavy_hitter", dtype, H, S, D, g, w)
    pos0 = torch.arange(T, device=DEV)
    kv.update_kv(pos0, f["k0"].to(DEV), f["v0"].to(DEV), True)
    kv.update_state(pos0, f["k0"].to(DEV), f["v0"].to(DEV), True, f["attn0"].to(DEV))
    ref_num = f["num_after_prefill"]
    assert torch.allclose(kv.attn_history_num.cpu(), ref_num, rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)
    assert torch.equal(kv.attn_history_denom.cpu(), f["denom_after_prefill"])
    kv.attn_history_num.copy_(ref_num.to(DEV))  # continue from identical state (column-sum order is unspecified)
    for t in range(f["steps"]):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1, v1 = f["k_new"][t].to(DEV), f["v_new"][t].to(DEV)
        kv.update_kv(p, k1, v1, False)
        assert np.array_equal(_idx(kv), f["idx"][t].numpy()), f"step {t}"
        assert torch.equal(kv.cache_cts.cpu(), f["cache_cts_steps"][t])
        kv.update_state(p, k1, v1, False, f["attn"][t].to(DEV))
    assert torch.equal(kv.attn_history_num.cpu(), f["final_num"])
    assert torch.equal(kv.attn_history_denom.cpu(), f["final_denom"])
    assert torch.equal(kv.attn_counter.cpu(), f["final_counter"])
    _final_equal(kv, f)


@pytest.mark.parametrize("name", ["f3_l2_bf16.npz", "f3_l2_f32.npz", "f3_l2_h1_bf16.npz", "f3_l2_long_bf16.npz"])
def test_l2_replay_vs_reference(cc, name):
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    kv = _make(cc, "l2", dtype, H, S, D, g, w)
    pos0 = torch.arange(T, device=DEV)
    kv.update_kv(pos0, f["k0"].to(DEV), f["v0"].to(DEV), True)
    kv.update_state(pos0, f["k0"].to(DEV), f["v0"].to(DEV), True, None)
    tol = dict(rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)
    assert torch.allclose(kv.key_norm.cpu().float(), f["keynorm_after_prefill"].float(), **tol)
    kv.key_norm.copy_(f["keynorm_after_prefill"].to(DEV))
    for t in range(f["steps"]):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        kv.update_kv(p, f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), False)
        assert np.array_equal(_idx(kv), f["idx"][t].numpy()), f"step {t}"
    _final_equal(kv, f)
    assert torch.allclose(kv.key_norm.cpu().float(), f["final_keynorm"].float(), **tol)


@pytest.mark.parametrize("name", ["f3_l2_h1_bf16.npz", "f3_l2_long_bf16.npz"])
@pytest.mark.parametrize("single", [False, True])
def test_l2_replay_vs_reference_through_the_fused_step(cc, name, single):
    """The reference's own l2 traces at head_dim 128 (120 / 300 steps) replayed through decode_step — two launches, and the
    single launch in which every workgroup gathers every workgroup's norm maximum: the slot every step writes is the slot the
    reference evicted, and the final positions, masks, counts, K, V and norms are the reference's.  (The queries are ours: the
    l2 policy does not look at attention.)"""
    from cold_compress_amd import _abi

    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    assert D == 128
    _abi.lib()["cc_decode_step_set_single_launch"](1 if single else 0)
    try:
        kv = _make(cc, "l2", dtype, H, S, D, g, w)
        pos0 = torch.arange(T, device=DEV)
        kv.update_kv(pos0, f["k0"].to(DEV), f["v0"].to(DEV), True)
        kv.update_state(pos0, f["k0"].to(DEV), f["v0"].to(DEV), True, None)
        kv.key_norm.copy_(f["keynorm_after_prefill"].to(DEV))  # vector_norm's summation order is unspecified: start from the reference's norms
        gen = torch.Generator().manual_seed(2)
        for t in range(f["steps"]):
            p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
            q = torch.randn(1, 4 * H, 1, D, generator=gen).to(dtype).to(DEV)
            kv.decode_step(q, f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), p)
            if t % 10 == 0 or t == f["steps"] - 1:
                pos = kv.pos.cpu()[0]
                for h in range(H):
                    assert int(pos[h, int(f["idx"][t][h])]) == T + t, f"step {t} head {h}: not the slot the reference evicted"
                assert torch.equal(kv.cache_cts.cpu(), f["cache_cts_steps"][t])
        _final_equal(kv, f)
        tol = dict(rtol=2 ** -7, atol=0)
        assert torch.allclose(kv.key_norm.cpu().float(), f["final_keynorm"].float(), **tol)
        from cold_compress_amd.attention_utils import single_launch_status

        assert single_launch_status() == 0
    finally:
        _abi.lib()["cc_decode_step_set_single_launch"](1)


def test_random_replay_vs_reference(cc):
    f = load_golden("f4_random.npz")
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    kv = _make(cc, "random", dtype, H, S, D, g, w)
    kv.update_kv(torch.arange(T, device=DEV), f["k0"].to(DEV), f["v0"].to(DEV), True)
    for t in range(f["steps"]):
        kv._rand = lambda t=t: f["rand_u"][t].to(DEV)
        kv.update_kv(torch.tensor([T + t], dtype=torch.int32, device=DEV), f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), False)
        assert np.array_equal(_idx(kv), f["idx"][t].numpy().reshape(-1)), f"step {t}"
    _final_equal(kv, f)


@pytest.mark.parametrize("strategy", ["full", "recent_global", "keep_it_odd"])
def test_head_constant_replay_vs_reference(cc, strategy):
    z = load_golden("f4_headconst.npz")
    f = {k[len(strategy) + 1:]: v for k, v in z.items() if k.startswith(strategy + ".")}
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    kv = _make(cc, strategy, dtype, H, S, D, g, w)
    kv.update_kv(torch.arange(T, device=DEV), f["k0"].to(DEV), f["v0"].to(DEV), True)
    for t in range(f["steps"]):
        kv.update_kv(torch.tensor([T + t], dtype=torch.int32, device=DEV), f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), False)
        assert np.array_equal(_idx(kv), f["idx"][t].numpy().reshape(-1)), f"step {t}"
    _final_equal(kv, f)


# ------------------------------------------------------------------------------------ compaction


def _tie_class_ok(prio_row, keep, K):
    v = prio_row.double()
    kth = v.sort(descending=True).values[K - 1]
    better = set(torch.nonzero(v > kth).view(-1).tolist())
    tie = set(torch.nonzero(v == kth).view(-1).tolist())
    ks = set(keep.tolist())
    assert len(ks) == K and better <= ks and ks <= (better | tie)


def test_compressors_vs_reference(cc):
    import cold_compress_amd.prompt_compression as P

    z = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "f5_compress.npz"))
    f = load_golden("f5_compress.npz")
    for name in [str(c) for c in z["cases"]]:
        prio, ref_keep = f[name + ".priority"], f[name + ".keep"]
        K = ref_keep.shape[-1]
        keep = P.topk_keep(prio.to(DEV), K).cpu()
        tie = bool(z[name + ".tie"])
        if not tie:
            assert torch.equal(keep.view(ref_keep.shape), ref_keep), name
        p2 = prio.reshape(-1, prio.shape[-1])
        for h in range(p2.shape[0]):
            assert bool((keep[h][1:] > keep[h][:-1]).all())
            _tie_class_ok(p2[h], keep[h], K)
        got = P.gather_rows(f[name + ".k_in"].to(DEV), ref_keep.to(DEV)).cpu()
        assert torch.equal(got.float(), f[name + ".k_out"].float()), name
    # full compressor objects (priority computed by the product too)
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        kw = dict(max_cache_length=int(f[f"l2_{tag}.keep"].shape[-1]), global_tokens=4, recent_window=10)  # (40 in the committed fixture)
        k, v = f[f"l2_{tag}.k_in"].to(DEV), f[f"l2_{tag}.v_in"].to(DEV)
        L = k.shape[2]
        pos = torch.arange(L, device=DEV)
        comp = P.PromptCompressorL2(head_specific=True, **kw)
        pr = comp._token_importances(pos, k, v).cpu()
        ref = f[f"l2_{tag}.priority"]
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(pr), fin)
        assert torch.allclose(pr[fin].float(), ref[fin].float(), rtol=2 ** -7 if tag == "bf16" else 1e-6, atol=0)
        rg = P.PromptCompressorRecentGlobal(head_specific=False, **kw)
        keep, k2, v2, st = rg(pos, f[f"recent_global_{tag}.k_in"].to(DEV), f[f"recent_global_{tag}.v_in"].to(DEV))
        assert torch.equal(keep.cpu(), f[f"recent_global_{tag}.keep"]) and st is None
        assert torch.equal(k2.cpu().float(), f[f"recent_global_{tag}.k_out"].float())
        hh = P.PromptCompressorHeavyHitter(head_specific=True, **kw)
        attn = f[f"heavy_hitter_{tag}.attn"].to(DEV)
        pr = hh._token_importances(pos, k, v, attn=attn).cpu()
        assert torch.allclose(pr.float(), f[f"heavy_hitter_{tag}.priority"].float(), rtol=2 ** -7 if tag == "bf16" else 1e-6, atol=1e-7)
        st = hh._update_state(f[f"heavy_hitter_{tag}.keep"].to(DEV), pos, attn=attn).cpu()
        assert torch.allclose(st.float(), f[f"heavy_hitter_{tag}.state"].float(), rtol=2 ** -6 if tag == "bf16" else 1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------ attention


@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("case", ["dec", "dec8b"])
def test_decode_attention_vs_reference(cc, tag, case):
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    f = load_golden(f"f7_attn_{tag}.npz")
    q, k, v, mask = (f[case + "." + n].to(DEV) for n in ("q", "k", "v", "mask"))
    y, probs = sdpa(q, k, v, attn_mask=mask, return_attn=True)
    y2, gm = sdpa(q, k, v, attn_mask=mask, return_attn=True, group_mean=True)
    y3, none = sdpa(q, k, v, attn_mask=mask, return_attn=False)
    assert none is None and torch.equal(y, y2) and torch.equal(y, y3)
    tol = 8e-3 if tag == "bf16" else 1e-3  # bf16: one ulp at |y|~1; fp32: the north-star 1e-3
    assert (y.cpu().float() - f[case + ".y"].float()).abs().max() < tol
    assert (probs.cpu().float() - f[case + ".probs"].float()).abs().max() < 1e-3
    assert (gm.cpu().float() - f[case + ".attn_gm"].float()).abs().max() < 1e-3
    # pre-repeated K/V and mask (what the reference's model.py passes) must give the same answer
    R = q.shape[1] // k.shape[1]
    y4, p4 = sdpa(q, k.repeat_interleave(R, 1), v.repeat_interleave(R, 1), attn_mask=mask.repeat_interleave(R, 1), return_attn=True)
    assert (y4.float() - y.float()).abs().max() < tol and (p4.float() - probs.float()).abs().max() < 1e-3


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_prefill_attention_vs_reference(cc, tag):
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    f = load_golden(f"f7_attn_{tag}.npz")
    q, k, v = (f["pre." + n].to(DEV) for n in ("q", "k", "v"))
    L = q.shape[2]
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool, device=DEV)).view(1, 1, L, L)
    y, summ = sdpa(q, k, v, attn_mask=causal, return_attn=True)
    tol = 8e-3 if tag == "bf16" else 1e-3
    assert (y.cpu().float() - f["pre.y"].float()).abs().max() < tol
    assert (summ.colsum.cpu() - f["pre.colsum"][0].float()).abs().max() < (6e-2 if tag == "bf16" else 1e-3)
    assert (summ.obs_mean.cpu() - f["pre.obs_mean"][0].float()).abs().max() < (4e-3 if tag == "bf16" else 1e-3)


# ------------------------------------------------------------------------------------ HIP vs oracle, seeded


def _oracle_state(oracle, kv, strategy):
    """numpy mirror of a device cache for the oracle."""
    H, S, D = kv.n_heads, kv.max_cache_length, kv.head_dim
    st = {"k": to_np(kv.k_cache.cpu()[0]), "v": to_np(kv.v_cache.cpu()[0]), "pos": kv.pos.cpu()[0].numpy().copy(),
          "mask": kv.mask.cpu()[0, :, 0].numpy().astype(np.uint8), "cts": kv.cache_cts.cpu().numpy().copy()}
    if strategy == "heavy_hitter":
        st["num"] = kv.attn_history_num.cpu()[0, :, :, 0].numpy().copy()
        st["denom"] = kv.attn_history_denom.cpu()[0].numpy().copy()
    if strategy == "l2":
        st["kn"] = to_np(kv.key_norm.cpu()[0])
    return st


@pytest.mark.parametrize("strategy,dtype,H,S,D", [
    ("heavy_hitter", torch.bfloat16, 8, 4096, 128), ("heavy_hitter", torch.float32, 2, 333, 16),
    ("l2", torch.bfloat16, 8, 4096, 128), ("l2", torch.float16, 4, 1000, 64), ("random", torch.bfloat16, 8, 4096, 128),
    ("recent_global", torch.bfloat16, 8, 4096, 128), ("full", torch.float32, 2, 512, 32),
    ("heavy_hitter", torch.bfloat16, 1, 3488, 128)])
def test_decode_update_bit_exact_vs_oracle(cc, oracle, strategy, dtype, H, S, D):
    """Seeded state at BASELINE sizes; 12 decode steps on both sides; every index and all state bit-exact."""
    _replay_vs_oracle(cc, oracle, strategy, dtype, H, S, D, S - 5, 4, 10)  # a few empty slots: the -1 path runs first


def test_decode_update_fuzz_vs_oracle(cc, oracle):
    """The same replay over 60 seeded random configurations: policy, dtype, head count, cache length 6 .. 3000,
    head_dim 16 .. 128, fill level 0 .. full, sinks and recent window."""
    import random

    rng = random.Random(2024)
    for i in range(60):
        strategy = rng.choice(["heavy_hitter", "l2", "random", "recent_global", "full"])
        dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        H, D = rng.choice([1, 2, 3, 8]), rng.choice([16, 32, 64, 128])
        S = rng.choice([rng.randint(6, 40), rng.randint(41, 300), rng.randint(301, 3000)])
        T = rng.choice([0, S, rng.randint(0, S)])
        g, w = rng.randint(0, min(4, S // 3)), rng.randint(1, max(1, min(10, S // 3)))
        _replay_vs_oracle(cc, oracle, strategy, dtype, H, S, D, T, g, w, steps=8, seed=i, tag=f"case {i}")


def _replay_vs_oracle(cc, oracle, strategy, dtype, H, S, D, T, g, w, steps=12, seed=None, tag=""):
    gen = torch.Generator().manual_seed(1234 + S if seed is None else 99_000 + seed)
    kv = _make(cc, strategy, dtype, H, S, D, g, w)
    hp = H if kv.head_specific else 1
    pos = torch.stack([torch.randperm(T + 50, generator=gen)[:T] for _ in range(hp)]).to(torch.int32)
    kv.pos[0, :, :T] = pos.to(DEV)
    kv.mask[0, :, 0, :T] = True
    kv.cache_cts.fill_(T)
    kv.k_cache.copy_(torch.randn(1, H, S, D, generator=gen).to(dtype))
    kv.v_cache.copy_(torch.randn(1, H, S, D, generator=gen).to(dtype))
    if strategy == "heavy_hitter" and T > 0:
        kv.attn_history_num[0, :, :T, 0] = torch.rand(H, T, generator=gen, dtype=torch.float64).to(DEV) * 3
        kv.attn_history_denom[0, :, :T] = torch.randint(1, 9, (H, T), generator=gen, dtype=torch.int32).to(DEV)
        # engineered ties: identical averages in several slots -> lowest index must win
        kv.attn_history_num[0, :, 100:110, 0] = 0.0
    if strategy == "l2":
        kv.update_state(None, None, None, True, None)
    st = _oracle_state(oracle, kv, strategy)
    code = DT_CODE[dtype]
    p0 = T + 60
    for t in range(steps):
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        p = torch.tensor([p0 + t], dtype=torch.int32)
        r = torch.rand(S, generator=gen)
        if strategy == "random":
            kv._rand = lambda r=r: r.to(DEV)
        kv.update_kv(p.to(DEV), k1.to(DEV), v1.to(DEV), False)
        idx = np.zeros((hp,), np.int64)
        view = oracle.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], code)
        kn, vn, pn = to_np(k1.reshape(H, D)), to_np(v1.reshape(H, D)), p.numpy().copy()
        o = oracle
        if strategy == "heavy_hitter":
            o.call("cc_decode_update_heavy_hitter", C.byref(view), o.ptr(kn), o.ptr(vn), o.ptr(pn), o.ptr(st["num"]), o.ptr(st["denom"]), g, w, o.ptr(idx), None)
        elif strategy == "l2":
            o.call("cc_decode_update_l2", C.byref(view), o.ptr(kn), o.ptr(vn), o.ptr(pn), o.ptr(st["kn"]), g, w, o.ptr(idx), None, 0, None)
        elif strategy == "random":
            rn = r.numpy().copy()
            o.call("cc_decode_update_random", C.byref(view), o.ptr(kn), o.ptr(vn), o.ptr(pn), o.ptr(rn), g, w, o.ptr(idx), None)
        elif strategy == "recent_global":
            o.call("cc_decode_update_recent_global", C.byref(view), o.ptr(kn), o.ptr(vn), o.ptr(pn), g, o.ptr(idx), None)
        else:
            o.call("cc_decode_update_full", C.byref(view), o.ptr(kn), o.ptr(vn), o.ptr(pn), o.ptr(idx), None)
        assert np.array_equal(_idx(kv), idx), f"{tag} step {t}: {strategy} {dtype} H={H} S={S} D={D} T={T} g={g} w={w}"
        if strategy == "heavy_hitter":  # evolve the history identically on both sides
            a = torch.softmax(torch.randn(H, S, generator=gen) * 3, -1).to(dtype)
            kv.update_state(p.to(DEV), k1.to(DEV), v1.to(DEV), False, a.view(1, H, 1, S).to(DEV))
            an = to_np(a)
            o.call("cc_hh_update", o.ptr(st["num"]), o.ptr(st["denom"]), None, o.ptr(an), H, S, S, code, None)
    got = _oracle_state(oracle, kv, strategy)
    for key in st:
        if not np.array_equal(got[key], st[key]):
            where = np.argwhere(np.asarray(got[key]) != np.asarray(st[key]))[:4].tolist()
            vals = [(np.asarray(got[key])[tuple(ix)].item(), np.asarray(st[key])[tuple(ix)].item()) for ix in where]
            raise AssertionError(f"{tag} {key}: {strategy} {dtype} H={H} S={S} D={D} T={T} g={g} w={w}: first differences at {where}: {vals}")


@pytest.mark.parametrize("dtype,HQ,H,S,D", [(torch.bfloat16, 32, 8, 4096, 128), (torch.bfloat16, 32, 8, 2560, 128),
                                            (torch.float32, 4, 2, 77, 16), (torch.float16, 8, 8, 300, 64),
                                            (torch.bfloat16, 8, 1, 3488, 128), (torch.bfloat16, 28, 4, 513, 128),
                                            (torch.bfloat16, 16, 8, 200, 128), (torch.float16, 32, 8, 1000, 128),
                                            (torch.bfloat16, 4, 4, 40, 128), (torch.float16, 6, 3, 8200, 128),
                                            (torch.bfloat16, 12, 3, 1001, 128), (torch.bfloat16, 2, 1, 13, 128)])
def test_decode_attention_vs_oracle(cc, oracle, dtype, HQ, H, S, D):
    _decode_attention_vs_oracle(oracle, dtype, HQ, H, S, D, 7 + S, 0.1)


def test_decode_attention_fuzz_vs_oracle(cc, oracle):
    """40 seeded random shapes: 1 .. 3000 slots, head_dim 16 .. 128, 1 .. 32 query heads per kv head, masks from almost
    empty to full (at least one live slot per head, as the cache guarantees)."""
    import random

    rng = random.Random(77)
    for i in range(40):
        dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        H, R, D = rng.choice([1, 2, 3, 8]), rng.choice([1, 2, 4, 8, 32]), rng.choice([16, 32, 64, 128, 128])
        S = rng.choice([rng.randint(1, 40), rng.randint(41, 300), rng.randint(301, 3000)])
        _decode_attention_vs_oracle(oracle, dtype, H * R, H, S, D, 5000 + i, rng.choice([0.0, 0.1, 0.5, 0.97]), f"case {i}")


def _decode_attention_vs_oracle(oracle, dtype, HQ, H, S, D, seed, p_masked, tag=""):
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype)
    k = torch.randn(1, H, S, D, generator=gen).to(dtype)
    v = torch.randn(1, H, S, D, generator=gen).to(dtype)
    mask = torch.rand(1, H, 1, S, generator=gen) >= p_masked
    mask[..., -1] = True
    y, gm = sdpa(q.to(DEV), k.to(DEV), v.to(DEV), attn_mask=mask.to(DEV), return_attn=True, group_mean=True)
    _, probs = sdpa(q.to(DEV), k.to(DEV), v.to(DEV), attn_mask=mask.to(DEV), return_attn=True)
    code = DT_CODE[dtype]
    es = np.float32 if code == 0 else np.uint16
    yo, ao, po = np.zeros((HQ, D), es), np.zeros((H, S), es), np.zeros((HQ, S), es)
    o = oracle
    o.call("cc_decode_attn_gqa", o.ptr(to_np(q[0, :, 0])), o.ptr(to_np(k[0])), o.ptr(to_np(v[0])), o.ptr(to_np(mask[0, :, 0])),
           HQ, H, S, D, code, 1.0 / math.sqrt(D), o.ptr(yo), o.ptr(ao), o.ptr(po), None, None, None, None, 0, None)
    ulp = {torch.float32: 1e-5, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[dtype]
    yref, yg = from_np(yo, dtype).float(), y.cpu().float()[0, :, 0]
    what = f"{tag} {dtype} HQ={HQ} H={H} S={S} D={D} p_masked={p_masked}"
    assert (yg - yref).abs().max() <= 1e-3 + 2 * ulp * yref.abs().max(), what
    assert (probs.cpu().float()[0, :, 0] - from_np(po, dtype).float()).abs().max() < 1e-3 + 2 * ulp, what
    assert (gm.cpu().float()[0, :, 0] - from_np(ao, dtype).float()).abs().max() < 1e-3 + 2 * ulp, what
    # masked slots carry exactly zero probability and rows sum to ~1
    pm = probs.cpu().float()[0, :, 0].view(H, HQ // H, S)
    dead = pm[~mask[0, :, 0].unsqueeze(1).expand_as(pm)]
    assert dead.numel() == 0 or float(dead.abs().max()) == 0.0, what
    assert (pm.sum(-1) - 1).abs().max() < (2e-2 if code else 1e-4), what


@pytest.mark.parametrize("dtype,HQ,H,L,D", [(torch.float32, 4, 2, 70, 16), (torch.bfloat16, 8, 2, 130, 64),
                                            (torch.bfloat16, 32, 8, 96, 128), (torch.float16, 6, 3, 33, 32)])
def test_prefill_attention_vs_oracle(cc, oracle, dtype, HQ, H, L, D):
    _prefill_attention_vs_oracle(oracle, dtype, HQ, H, L, D, 11 + L)


def test_prefill_attention_fuzz_vs_oracle(cc, oracle):
    """24 seeded random shapes: prompts of 1 .. 400 tokens (ragged against the 32-token tiles), head_dim 16 .. 128,
    1 .. 8 query heads per kv head; the matrix-core path (16-bit, head_dim 128, 4 heads per group) and the VALU path."""
    import random

    rng = random.Random(78)
    for i in range(24):
        dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        H, R, D = rng.choice([1, 2, 3]), rng.choice([1, 2, 4, 4, 8]), rng.choice([16, 64, 128, 128])
        L = rng.choice([rng.randint(1, 40), rng.randint(41, 130), rng.randint(131, 400)])
        _prefill_attention_vs_oracle(oracle, dtype, H * R, H, L, D, 6000 + i, f"case {i}")


def _prefill_attention_vs_oracle(oracle, dtype, HQ, H, L, D, seed, tag=""):
    from cold_compress_amd.attention_utils import prefill_attention

    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(1, HQ, L, D, generator=gen).to(dtype)
    k = torch.randn(1, H, L, D, generator=gen).to(dtype)
    v = torch.randn(1, H, L, D, generator=gen).to(dtype)
    y, summ = prefill_attention(q.to(DEV), k.to(DEV), v.to(DEV), return_attn=True)
    code = DT_CODE[dtype]
    es = np.float32 if code == 0 else np.uint16
    yo, cs, ob = np.zeros((HQ, L, D), es), np.zeros((H, L), np.float32), np.zeros((H, L), np.float32)
    o = oracle
    o.call("cc_prefill_attn", o.ptr(to_np(q[0])), o.ptr(to_np(k[0])), o.ptr(to_np(v[0])), HQ, H, L, D, code, 1.0 / math.sqrt(D),
           o.ptr(yo), o.ptr(cs), o.ptr(ob), 16, None, 0, None)
    ulp = {torch.float32: 1e-5, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[dtype]
    yref = from_np(yo, dtype).float()
    what = f"{tag} {dtype} HQ={HQ} H={H} L={L} D={D}"
    assert (y.cpu().float()[0] - yref).abs().max() <= 1e-3 + 2 * ulp * yref.abs().max(), what
    assert (summ.colsum.cpu() - torch.from_numpy(cs)).abs().max() < (5e-2 if code else 1e-3), what
    assert (summ.obs_mean.cpu() - torch.from_numpy(ob)).abs().max() < (4e-3 if code else 1e-3), what


def test_fused_history_equals_separate_update(cc):
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    H, HQ, S, D = 8, 32, 4096, 128
    gen = torch.Generator().manual_seed(5)
    a = _make(cc, "heavy_hitter", torch.bfloat16, H, S, D, 4, 10)
    b = _make(cc, "heavy_hitter", torch.bfloat16, H, S, D, 4, 10)
    k0 = torch.randn(1, H, S - 3, D, generator=gen).to(torch.bfloat16).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(S - 3, device=DEV), k0, k0, True)
    for t in range(6):
        p = torch.tensor([S + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(torch.bfloat16).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(torch.bfloat16).to(DEV)
        ka, va, ma = a.update_kv(p, k1, k1, False)
        ya, attn = sdpa(q, ka, va, attn_mask=ma, return_attn=True, group_mean=True)
        a.update_state(p, k1, k1, False, attn)
        kb, vb, mb = b.update_kv(p, k1, k1, False)
        yb, none = sdpa(q, kb, vb, attn_mask=mb, return_attn=False, group_mean=True, history=b.fused_history())
        assert none is None and torch.equal(ya, yb)
        assert np.array_equal(_idx(a), _idx(b))
    assert torch.equal(a.attn_history_num, b.attn_history_num)
    assert torch.equal(a.attn_history_denom, b.attn_history_denom)
    assert torch.equal(a.attn_counter, b.attn_counter)


def test_cpu_tensors_are_refused(cc):
    from cold_compress_amd._abi import ColdCompressError

    cls, rk = cc.get_cache_constructor("recent_global")
    kw = _kw(16, 4, 3)
    kv = cls(1, 2, 16, torch.float32, **{k: kw[k] for k in rk})  # buffers on CPU
    with pytest.raises(ColdCompressError):
        kv.update_kv(torch.arange(4), torch.zeros(1, 2, 4, 16), torch.zeros(1, 2, 4, 16), True)


def test_topk_decode_attention_matches_reference():
    """attention_utils.py:24-26, 45-50 (attn_top_k < 1, L == 1, no mask): output and the top-k probabilities (descending
    order, as torch.topk returns them) within 1e-5 in fp32; with a mask the reference asserts, and so do we."""
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    f = load_golden("f7_attn_topk_f32.npz")
    assert f["mask_asserts"] == 1
    q, k, v = f["q"].to(DEV), f["k"].to(DEV), f["v"].to(DEV)
    for pct in (25, 50):
        y, p = sdpa(q, k, v, attn_mask=None, return_attn=True, attn_top_k=pct / 100.0)
        torch.cuda.synchronize()
        assert p.shape == f[f"p_{pct}"].shape
        assert (y.cpu() - f[f"y_{pct}"]).abs().max() < 1e-5
        assert (p.cpu() - f[f"p_{pct}"]).abs().max() < 1e-5
    with pytest.raises(AssertionError, match="Top-k attention not supported with masks"):
        sdpa(q, k, v, attn_mask=torch.ones(1, 4, 1, 64, dtype=torch.bool, device=DEV), return_attn=True, attn_top_k=0.5)


@pytest.mark.parametrize("name", ["f10_analysis_hh_f32.npz", "f10_analysis_hh_bf16.npz"])
def test_debug_analysis_cache_vs_reference(name):
    """`debug_heavy_hitter` (KVCacheAnalysis, cache.py:1291-1420) replayed on the reference's own trace — captured with the
    one keyword its constructor forgets injected (oracle/gen_golden.py: analysis_case): prompt longer than the shadow cache
    (SnapKV compaction inside update_state), then decode steps with the attention over the FULL cache given; the shadow
    cache's positions after every step exactly, its history and the recorded attention losses to the dtype's rounding."""
    import json

    import cold_compress_amd.cache as cache

    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, D, S, SF, L, steps, g, w = f["H"], f["D"], f["S"], f["S_full"], f["L"], f["steps"], f["g"], f["w"]
    ctor, rk = cache.get_cache_constructor("debug_heavy_hitter")
    kw = dict(max_cache_length=S, global_tokens=g, max_seq_length=SF, cache_bits=None, recent_window=w, history_window_size=1,
              attn_thresholding=False, prompt_compression_strategy="heavy_hitter")
    with torch.device(DEV):
        kv = ctor(1, H, D, dtype, **{k: kw[k] for k in rk})
    pos0 = torch.arange(L, device=DEV)
    k0, v0 = f["k0"].to(DEV), f["v0"].to(DEV)
    kv.update_kv(pos0, k0, v0, True)
    kv.update_state(pos0, k0, v0, True, f["attn0"].to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(kv.pos.cpu(), f["full_pos_after_prefill"])
    tie_ok = dtype != torch.float32  # 16-bit SnapKV priorities can tie at the boundary: then only the set size is pinned
    same_keep = torch.equal(kv.compressed.pos.cpu(), f["comp_pos_after_prefill"])
    assert same_keep or tie_ok
    if not same_keep:
        # boundary tie in the 16-bit prompt-compaction priorities: the two keep sets are equally good.  The replay
        # FOLLOWS the reference's (the shadow cache is rebuilt from the prompt rows at the reference's positions) instead of
        # skipping, so that the 16-bit fixture always pins the decode-time bookkeeping too.
        comp, rp = kv.compressed, f["comp_pos_after_prefill"].to(DEV)
        a, b = set(kv.compressed.pos.cpu().flatten().tolist()), set(rp.cpu().flatten().tolist())
        assert len(a ^ b) <= 2 * H * 2, "more than a boundary tie separates the keep sets"
        idx = rp[0].to(torch.int64)  # [H, S] prompt positions
        comp.pos.copy_(rp)
        comp.k_cache.copy_(torch.gather(k0[0], 1, idx.unsqueeze(-1).expand(-1, -1, D)).unsqueeze(0))
        comp.v_cache.copy_(torch.gather(v0[0], 1, idx.unsqueeze(-1).expand(-1, -1, D)).unsqueeze(0))
        comp._next_valid = False
    tol = dict(rtol=2 ** -7, atol=1e-6) if dtype != torch.float32 else dict(rtol=1e-5, atol=1e-7)
    if same_keep:
        assert torch.allclose(kv.compressed.attn_history_num.cpu().float(), f["comp_num_after_prefill"].float(), **tol)
    kv.compressed.attn_history_num.copy_(f["comp_num_after_prefill"].to(DEV))  # column-sum order is unspecified: continue on equal state
    for t in range(steps):
        p = torch.tensor([L + t], dtype=torch.int32, device=DEV)
        k1, v1 = f["k_new"][t].to(DEV), f["v_new"][t].to(DEV)
        kv.update_kv(p, k1, v1, False)
        kv.update_state(p, k1, v1, False, f["attn"][t].to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(kv.compressed.pos.cpu(), f["comp_pos_steps"][t]), f"step {t}: shadow cache positions"
        ulp = 2 ** -7 if dtype != torch.float32 else 1e-6
        assert abs(float(kv.attention_losses[t]) - float(f["loss_steps"][t])) <= ulp * max(1.0, abs(float(f["loss_steps"][t]))), f"step {t}: loss"
    assert int(kv.attention_loss_ctr) == int(f["loss_ctr"]) == steps
    assert torch.equal(kv.pos.cpu(), f["final_full_pos"])
    assert torch.equal(kv.compressed.attn_history_denom.cpu(), f["final_comp_denom"])
    assert torch.allclose(kv.compressed.attn_history_num.cpu().float(), f["final_comp_num"].float(), **tol)
    assert (kv.compressed.k_cache.cpu().float() - f["final_comp_k"].float()).abs().max() == 0
    st, ref = kv.compute_statistics(torch.tensor(L + steps)), json.loads(f["stats_json"])
    assert abs(st["attention_loss"] - ref["attention_loss"]) <= (2 ** -7 if dtype != torch.float32 else 1e-6)
    assert abs(st["compression_ratio"] - ref["compression_ratio"]) < 1e-6
