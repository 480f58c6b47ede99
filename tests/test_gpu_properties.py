"""Size-independent properties of the hot path at the BASELINE configuration (Llama-3-8B geometry, heavy_hitter,
S = 4096, bf16) — checked with plain torch on the device, independently of the oracle:
  * the evicted slot is the arg-min of the reference's score with the lowest index on ties (cache.py:725-749);
  * an insert touches exactly one row per head: that row holds the new token bit for bit, every other byte of K, V,
    pos is unchanged (cache.py:460-490);
  * attention probabilities: masked slots are exactly zero, every group-averaged row sums to 1 (bf16 tolerance);
  * attention is invariant under a permutation of the cache slots (K, V, mask permuted together);
  * masked-out padding does not matter: the same cache embedded in a longer, masked buffer gives the same output;
  * prompt compaction keeps indices sorted, keeps every global / recent token, and gathers rows bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, HQ, S, D, G, W = 8, 32, 4096, 128, 4, 10


def _hh(seed=0, fill=S):
    import cold_compress_amd.cache as cache

    g = torch.Generator().manual_seed(seed)
    with torch.device(DEV):
        kv = cache.KVCacheHeavyHitter(1, H, D, torch.bfloat16, max_cache_length=S, max_seq_length=4 * S, cache_bits=None,
                                      global_tokens=G, history_window_size=1, recent_window=W, attn_thresholding=False)
    kv.update_kv(torch.arange(fill, device=DEV), torch.randn(1, H, fill, D, generator=g).to(torch.bfloat16).to(DEV),
                 torch.randn(1, H, fill, D, generator=g).to(torch.bfloat16).to(DEV), True)
    kv.attn_history_num[0, :, :fill, 0] = torch.rand(H, fill, generator=g, dtype=torch.float64).to(DEV)
    kv.attn_history_denom[0, :, :fill] = torch.randint(1, 6, (H, fill), generator=g, dtype=torch.int32).to(DEV)
    kv.attn_history_num[0, :, 100:140, 0] = 0.0  # a tie class: the lowest index must win
    kv.pos[0] = torch.stack([torch.randperm(fill + 500, generator=g)[:fill] for _ in range(H)]).int().to(DEV) if fill == S else kv.pos[0]
    return kv, g


def _ref_scores(kv, p):
    """cache.py:727-749 in plain torch."""
    num = kv.attn_history_num[0, :, :, 0].float()
    den = kv.attn_history_denom[0].clamp(min=1).float()
    sc = num / den
    pos = kv.pos[0]
    sc = sc.masked_fill((pos < G) | (pos >= p - W), 1.0)
    return sc.masked_fill(pos == -1, 0.0)


@pytest.mark.parametrize("fused", [False, True])
def test_eviction_is_argmin_and_insert_touches_one_row(fused):
    kv, g = _hh()
    for step in range(3):
        p = 3 * S + step
        want = _ref_scores(kv, p).argmin(dim=-1)  # first minimal element, like the reference
        k0, v0, pos0 = kv.k_cache.clone(), kv.v_cache.clone(), kv.pos.clone()
        k1 = torch.randn(1, H, 1, D, generator=g).to(torch.bfloat16).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=g).to(torch.bfloat16).to(DEV)
        pt = torch.tensor([p], dtype=torch.int32, device=DEV)
        if fused:
            q = torch.randn(1, HQ, 1, D, generator=g).to(torch.bfloat16).to(DEV)
            kv.decode_step(q, k1, v1, pt)
        else:
            kv.update_kv(pt, k1, v1, False)
        torch.cuda.synchronize()
        changed = (kv.pos[0] != pos0[0]).nonzero()
        assert changed.shape[0] == H and torch.equal(changed[:, 0], torch.arange(H, device=DEV))
        assert torch.equal(changed[:, 1], want), f"step {step}: not the arg-min slot"
        rows = torch.arange(H, device=DEV)
        assert torch.equal(kv.k_cache[0, rows, want], k1[0, :, 0]) and torch.equal(kv.v_cache[0, rows, want], v1[0, :, 0])
        assert bool((kv.pos[0, rows, want] == p).all())
        keep = torch.ones(H, S, dtype=torch.bool, device=DEV)
        keep[rows, want] = False
        assert torch.equal(kv.k_cache[0][keep], k0[0][keep]) and torch.equal(kv.v_cache[0][keep], v0[0][keep])
        assert torch.equal(kv.pos[0][keep], pos0[0][keep])
        if not fused:  # the three-call path leaves the history to update_state; give it something to move on
            kv.attn_history_num[0, rows, want, 0] = 0.5
            kv.attn_history_denom[0, rows, want] = 1


def test_probabilities_masked_zero_rows_sum_to_one_and_permutation_invariance():
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    g = torch.Generator().manual_seed(4)
    q = torch.randn(1, HQ, 1, D, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(1, H, S, D, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(1, H, S, D, generator=g).to(torch.bfloat16).to(DEV)
    m = (torch.rand(1, H, 1, S, generator=g) > 0.3).to(DEV)
    m[..., 0] = True
    y, a = sdpa(q, k, v, attn_mask=m, return_attn=True, group_mean=True)
    assert bool((a[~m] == 0).all()), "masked slots must get exactly zero probability"
    assert (a.float().sum(-1) - 1).abs().max() < 2e-2  # 4096 bf16 roundings of ~1/2900
    perm = torch.stack([torch.randperm(S, generator=g) for _ in range(H)]).to(DEV)
    idx = perm.view(1, H, S, 1).expand(1, H, S, D)
    y2, a2 = sdpa(q, k.gather(2, idx), v.gather(2, idx), attn_mask=m.gather(3, perm.view(1, H, 1, S)), return_attn=True, group_mean=True)
    tol = 1e-3 + 2 * 2.0 ** -8 * float(y.float().abs().max())  # the attention contract: 1e-3 + two roundings of the output dtype
    assert (y.float() - y2.float()).abs().max() <= tol, "attention must not depend on the slot order"
    assert (a.gather(3, perm.view(1, H, 1, S)).float() - a2.float()).abs().max() < 1e-5 + 2 ** -9 * float(a.max())
    # the same cache embedded in a longer buffer whose tail is masked out
    pad = 2 * S
    kp = torch.cat([k, torch.randn(1, H, S, D, generator=g).to(torch.bfloat16).to(DEV)], 2)
    vp = torch.cat([v, torch.randn(1, H, S, D, generator=g).to(torch.bfloat16).to(DEV)], 2)
    mp = torch.cat([m, torch.zeros(1, H, 1, S, dtype=torch.bool, device=DEV)], 3)
    y3, a3 = sdpa(q, kp, vp, attn_mask=mp, return_attn=True, group_mean=True)
    assert a3.shape[-1] == pad and bool((a3[..., S:] == 0).all())
    assert (y.float() - y3.float()).abs().max() <= tol


def test_prompt_compaction_properties():
    from cold_compress_amd.prompt_compression import get_prompt_compressor_constructor

    L = 8192
    g = torch.Generator().manual_seed(9)
    k = torch.randn(1, H, L, D, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(1, H, L, D, generator=g).to(torch.bfloat16).to(DEV)
    pos = torch.arange(L, device=DEV)
    comp = get_prompt_compressor_constructor("l2")(head_specific=True, max_cache_length=S, global_tokens=G, recent_window=W)
    keep, k2, v2, _ = comp(pos, k, v)
    keep = keep.view(-1, S)
    assert bool((keep[:, 1:] > keep[:, :-1]).all()), "kept indices must be strictly ascending"
    must = torch.cat([torch.arange(G), torch.arange(L - W, L)]).to(DEV)
    assert all(bool(torch.isin(must, row).all()) for row in keep), "global and recent tokens are always kept"
    idx = keep.view(1, H, S, 1).expand(1, H, S, D)
    assert torch.equal(k2, k.gather(2, idx)) and torch.equal(v2, v.gather(2, idx))
    # every dropped key has a norm >= every kept, non-protected key's norm (smallest norms are kept: prompt_compression.py:201-209)
    norms = torch.linalg.vector_norm(k[0].float(), dim=-1)  # [H, L]
    kept = torch.zeros(H, L, dtype=torch.bool, device=DEV).scatter_(1, keep, True)
    prot = torch.zeros(L, dtype=torch.bool, device=DEV)
    prot[must] = True
    for h in range(H):
        worst_kept = norms[h][kept[h] & ~prot].max()
        best_dropped = norms[h][~kept[h]].min()
        assert best_dropped >= worst_kept * (1 - 2 ** -7), "a dropped key is clearly smaller than a kept one"


def test_topk_keep_fuzz_vs_oracle(oracle):
    """Prompt compaction's keep set (cc_topk_keep) over 60 seeded random cases — priority dtype (f32 / bf16 / f16 / int64),
    1 .. 9 rows, 1 .. 5000 keys, K from 1 to L, heavy ties (few distinct values), +-inf and NaN entries — device ==
    oracle exactly: the tie rule (lowest index first) and NaN-above-+inf are part of the contract."""
    import random

    from cold_compress_amd.prompt_compression import topk_keep
    from helpers import to_np

    rng = random.Random(17)
    codes = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.int64: 3}
    for i in range(60):
        dt = rng.choice(list(codes))
        Hs = rng.choice([1, 2, 8, 9])
        L = rng.choice([rng.randint(1, 70), rng.randint(71, 1100), rng.randint(1101, 5000)])
        K = rng.choice([1, L, rng.randint(1, L), max(1, L // 2)])
        gen = torch.Generator().manual_seed(4000 + i)
        if dt == torch.int64:
            pr = torch.randint(-5, rng.choice([3, 1000, 2 ** 40]), (Hs, L), generator=gen, dtype=torch.int64)
        else:
            pr = torch.randn(Hs, L, generator=gen)
            kind = rng.randrange(4)
            if kind == 1:
                pr = (pr * 2).round() / 2  # few distinct values: long tie classes at the K-th priority
            elif kind == 2:
                pr[:, :: max(1, L // 7)] = float("inf")
                pr[:, 1:: max(2, L // 5)] = float("-inf")
            elif kind == 3:
                pr[:, :: max(1, L // 4)] = float("nan")
            pr = pr.to(dt)
        keep = topk_keep(pr.to(DEV), K).cpu().numpy()
        ko = np.zeros((Hs, K), np.int64)
        arr = pr.numpy().copy() if dt == torch.int64 else to_np(pr)
        oracle.call("cc_topk_keep", oracle.ptr(arr), codes[dt], Hs, L, K, oracle.ptr(ko), None, 0, None)
        assert np.array_equal(keep, ko), f"case {i}: {dt} Hs={Hs} L={L} K={K}"
