#!/usr/bin/env python3
"""Differential fuzz: random (policy, dtype, heads, cache length, head_dim, fill level, sinks, window, cache_bits) —
the fused decode step (ONE launch where the shape allows it, two otherwise; all cases share one workspace and one set of epoch
words) against update_kv -> attention -> update_state, every buffer bit for bit, y within one rounding of the model dtype; the
history rings against their three-call form; the fused quantised cache against the 16-bit step on its dequantised values.
    python tools/fuzz_step.py [--n 300] [--seed 0]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import cold_compress_amd.cache as cache  # noqa: E402
from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa  # noqa: E402

DEV = "cuda"


def one(rng, idx):
    strategy = rng.choice(["heavy_hitter", "recent_global", "full", "random", "l2"])
    dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    D = rng.choice([16, 32, 64, 128, 128, 128])
    H = rng.choice([1, 2, 3, 8])
    R = rng.choice([1, 2, 4, 8])
    S = rng.choice([rng.randint(6, 40), rng.randint(41, 300), rng.randint(301, 3000), rng.choice([4096, 5000, 9000]),
                    rng.choice([8192, 12000, 18432]) if H == 8 else rng.randint(3001, 4096)])
    T = rng.choice([0, S, S, rng.randint(0, S)])
    g = rng.randint(0, min(4, S // 3))
    w = rng.randint(1, max(1, min(10, S // 3)))
    bits = rng.choice([None, None, None, 8, 4]) if strategy not in ("l2",) else None
    W = rng.choice([1, 1, 2, 5, 33]) if strategy == "heavy_hitter" else 1
    cfg = dict(strategy=strategy, dtype=str(dtype), H=H, R=R, S=S, D=D, T=T, g=g, w=w, bits=bits)
    cls, rk = cache.get_cache_constructor(strategy)
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, history_window_size=W, attn_thresholding=False,
              max_seq_length=4 * S + 64, cache_bits=bits)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    try:
        a, b = mk(), mk()
    except NotImplementedError:
        return None
    if not b.supports_fused_step():
        return None
    gen = torch.Generator().manual_seed(idx)
    steps = 6
    if strategy == "random":
        draws = [torch.rand(S, generator=gen).to(DEV) for _ in range(steps + 1)]
        ia, ib = iter(draws), iter(draws)
        a._rand = lambda: next(ia)
        b._rand = lambda: next(ib)
    if T > 0:
        k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
        v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
        for kv in (a, b):
            kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
            if strategy == "l2":
                kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None)
    for t in range(steps):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, H * R, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        fuse = strategy == "heavy_hitter" and W == 1
        ya, at = sdpa(q, ka, va, attn_mask=ma, return_attn=a.return_attn() and not fuse, group_mean=True,
                      history=a.fused_history() if fuse else None)
        if fuse:
            a._state_fused = True
        a.update_state(p, k1, v1, False, at)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        if not torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6):  # the single launch folds y in another fixed order
            return f"{cfg} step {t}: y differs (max {float((ya.float() - yb.float()).abs().max())})"
        a.dequantize_cache(), b.dequantize_cache()
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit") and not torch.equal(ta, tb):
                return f"{cfg} step {t}: buffer {na} differs"
    return ""


HYB = [{"strategy": "special"}, {"strategy": "special_punc"}, {"strategy": "special_punc_heavy_hitter", "heavy_hitter_frac": 0.3},
       {"strategy": "special_punc_window", "recent_window": 0.3}, {"strategy": "full"}]


def one_ring(rng, idx):
    """history ring (heavy_hitter with W > 1, hybrid with W = 400): the update folded into the combine pass against
    attention -> update_state, tracked window sums included."""
    strategy = rng.choice(["heavy_hitter", "hybrid"])
    dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    D = rng.choice([16, 64, 128, 128])
    H = rng.choice([1, 2, 5, 8])
    R = rng.choice([1, 4, 8])
    S = rng.choice([rng.randint(8, 60), rng.randint(61, 700), rng.randint(701, 2500)])
    T = rng.choice([S, S, rng.randint(1, S)])
    W = rng.choice([2, 3, 8, 33]) if strategy == "heavy_hitter" else 400
    g = rng.randint(0, min(4, S // 3))
    w = rng.randint(1, max(1, min(10, S // 3)))
    cfg = dict(strategy=strategy, dtype=str(dtype), H=H, R=R, S=S, D=D, T=T, g=g, w=w, W=W)
    cls, rk = cache.get_cache_constructor(strategy)
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, history_window_size=W, attn_thresholding=False,
              max_seq_length=4 * S + 64, cache_bits=None, token_ids={"special": [[1], [2, 3]], "punctuation": [5, 6, 7]},
              min_recovery_frac=0.9, hybrid_strategies=HYB)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    gen = torch.Generator().manual_seed(10_000 + idx)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True, input_ids=torch.zeros(T, dtype=torch.int64, device=DEV))
        if strategy == "hybrid":
            kv.cache_strategies = (torch.arange(H, device=DEV) % len(HYB)).to(torch.int64).contiguous()
            kv.requires_heavy_hitter = True
            kv.cache_cts.fill_(T)
            kv.mask[..., :T] = True
            kv.pos[0, :, :T] = torch.arange(T, device=DEV, dtype=kv.pos.dtype)
    for t in range(2 * min(W, 6) + 2):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        ids = torch.tensor([5 if t % 3 == 1 else 9], dtype=torch.int64, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, H * R, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False, input_ids=ids)
        ya, attn = sdpa(q, ka, va, attn_mask=ma, return_attn=True, group_mean=True)
        a.update_state(p, k1, v1, False, attn, input_ids=ids)
        kb, vb, mb = b.update_kv(p, k1, v1, False, input_ids=ids)
        yb, _ = sdpa(q, kb, vb, attn_mask=mb, group_mean=True, history=b.fused_history())
        b._state_fused = True
        b.update_state(p, k1, v1, False, None, input_ids=ids)
        torch.cuda.synchronize()
        if not torch.equal(ya, yb):
            return f"{cfg} step {t}: y differs"
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if not torch.equal(ta, tb):
                return f"{cfg} step {t}: buffer {na} differs"
    return ""


HYB_YAML = [{"strategy": "window", "recent_window": 0.1},
            {"strategy": "window_heavy_hitter", "heavy_hitter_frac": 0.25, "recent_window": 0.1},
            {"strategy": "window_heavy_hitter", "heavy_hitter_frac": 0.5, "recent_window": 0.1}, {"strategy": "full"}]


def one_hybrid_step(rng, idx):
    """KVCacheHybrid.decode_step (ONE launch where the shape allows it, two otherwise) against update_kv -> attention (ring update
    fused) -> update_state: every buffer bit for bit, y within one rounding; random per-head fill levels, policies, protection
    masks, punctuation tokens (3 ids or 100)."""
    dtype = rng.choice([torch.bfloat16, torch.float16])
    D = 128
    H = rng.choice([1, 2, 5, 8])
    R = rng.choice([2, 4, 4, 8])
    S = rng.choice([rng.randint(8, 200), rng.randint(201, 3000), rng.randint(3001, 4096), rng.choice([4100, 6000, 9000]),
                    18432 if H * R <= 32 else rng.randint(300, 2000)])
    strategies = rng.choice([HYB, HYB_YAML])
    long_punc = rng.random() < 0.3
    tids = {"special": [[1], [2, 3]], "punctuation": ([200 + i for i in range(80)] + [6] + [400 + i for i in range(19)]) if long_punc else [5, 6, 7]}
    g = rng.randint(0, min(4, S // 3))
    cfg = dict(strategy="hybrid_step", dtype=str(dtype), H=H, R=R, S=S, g=g, yaml=strategies is HYB_YAML, long_punc=long_punc)
    cls, rk = cache.get_cache_constructor("hybrid")
    kw = dict(max_cache_length=S, max_seq_length=S, cache_bits=None, global_tokens=g, token_ids=tids, min_recovery_frac=0.9,
              hybrid_strategies=strategies)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    gen = torch.Generator().manual_seed(20_000 + idx)
    T = rng.choice([S, max(1, S - 3), rng.randint(1, S)])
    fill = torch.tensor([rng.choice([T, max(1, T // 2), rng.randint(1, T)]) for _ in range(H)], dtype=torch.int32)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    ring0 = (torch.rand(H, S, 1, generator=gen) * torch.rand(1, 1, a.history_window_size, generator=gen) * 1e-2).to(dtype)
    den0 = torch.randint(1, 500, (H, S), generator=gen, dtype=torch.int32)
    sp0 = torch.rand(H, S, generator=gen) < 0.02
    pu0 = torch.rand(H, S, generator=gen) < 0.02
    off = rng.randint(0, len(strategies) - 1)
    npu0 = rng.randint(0, 5)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True, input_ids=torch.zeros(T, dtype=torch.int64, device=DEV))
        kv.cache_strategies = ((torch.arange(H, device=DEV) + off) % len(strategies)).to(torch.int64).contiguous()
        kv.requires_heavy_hitter = any("heavy_hitter" in s_["strategy"] for s_ in strategies)
        kv.cache_cts.copy_(fill.to(DEV))
        live = torch.arange(S, device=DEV).view(1, S) < fill.to(DEV).view(H, 1)
        kv.mask[0, :, 0, :] = live
        kv.pos[0] = torch.where(live, torch.arange(S, device=DEV, dtype=kv.pos.dtype).view(1, S).expand(H, S), torch.full_like(kv.pos[0], -1))
        kv.attn_history_num.copy_(ring0.to(DEV).unsqueeze(0))
        kv.attn_history_denom.copy_(den0.to(DEV).unsqueeze(0))
        if hasattr(kv, "special_mask"):
            kv.special_mask[0] = sp0.to(DEV) & live
            kv.num_special.fill_(int(sp0[0, : int(fill[0])].sum()))
        if hasattr(kv, "punc_mask"):
            kv.punc_mask[0] = pu0.to(DEV) & live
            kv.num_punc.fill_(npu0)
    if not b.supports_fused_step():
        return None
    for t in range(8):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        ids = torch.tensor([[6 if rng.random() < 0.3 else 11]], dtype=torch.int64, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, H * R, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False, input_ids=ids)
        hist = a.fused_history()
        ya, attn = sdpa(q, ka, va, attn_mask=ma, return_attn=a.return_attn() and hist is None, group_mean=True, history=hist)
        if hist is not None:
            a._state_fused = True
        a.update_state(p, k1, v1, False, attn, input_ids=ids)
        yb = b.decode_step(q, k1, v1, p, input_ids=ids)
        torch.cuda.synchronize()
        if not torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6):
            return f"{cfg} step {t}: y differs"
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit") and not torch.equal(ta, tb):
                return f"{cfg} step {t}: buffer {na} differs"
    return ""


def one_quant(rng, idx):
    """the fused quantised cache against the same policy's 16-bit cache holding its dequantised values (bit for bit)."""
    import ctypes as C

    from cold_compress_amd import _abi

    strategy = rng.choice(["heavy_hitter", "recent_global", "full", "random"])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    H = rng.choice([1, 2, 3, 8])
    R = rng.choice([4, 8])
    S = rng.choice([rng.randint(64, 300), rng.randint(301, 3000), rng.choice([4096, 2560, 3488])])
    T = rng.choice([S, S, rng.randint(1, S)])
    g = rng.randint(0, min(4, S // 3))
    w = rng.randint(1, max(1, min(10, S // 3)))
    D = 128
    cfg = dict(kind="fused_quant", strategy=strategy, dtype=str(dtype), H=H, R=R, S=S, T=T, g=g, w=w)
    cls, rk = cache.get_cache_constructor(strategy)
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, history_window_size=1, attn_thresholding=False,
              max_seq_length=4 * S + 64, cache_bits=None)

    def mk(fused):
        lk = {k: kw[k] for k in rk}
        if fused:
            lk.update(cache_bits=8, cache_quant_mode="fused")
        with torch.device(DEV):
            return cls(1, H, D, dtype, **lk)

    def round_trip(x):
        n = x.shape[0]
        kq, vq = torch.empty((n, D), dtype=torch.uint8, device=DEV), torch.empty((n, D), dtype=torch.uint8, device=DEV)
        par = torch.empty((n, 4), dtype=torch.float32, device=DEV)
        ko, vo = torch.empty_like(x), torch.empty_like(x)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        code = {torch.bfloat16: 1, torch.float16: 2}[x.dtype]
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        _abi.call("cc_kv_quant_rows", p(x), p(x), 1, n, D, code, 8, p(kq), p(vq), p(par), st)
        _abi.call("cc_kv_dequant_rows", p(kq), p(vq), p(par), 1, n, D, code, 8, p(ko), p(vo), st)
        return ko

    a, b = mk(False), mk(True)
    gen = torch.Generator().manual_seed(20_000 + idx)
    steps = 6
    if strategy == "random":
        draws = [torch.rand(S, generator=gen).to(DEV) for _ in range(steps + 2)]
        ia, ib = iter(list(draws)), iter(list(draws))
        a._rand = lambda: next(ia)
        b._rand = lambda: next(ib)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = (2.0 * torch.randn(1, H, T, D, generator=gen)).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
    kd, vd = b.dequantized_kv()
    a.k_cache.copy_(kd)
    a.v_cache.copy_(vd)
    for t in range(steps):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, H * R, 1, D, generator=gen).to(dtype).to(DEV)
        ya = a.decode_step(q, round_trip(k1.reshape(H, D)).view(1, H, 1, D), round_trip(v1.reshape(H, D)).view(1, H, 1, D), p)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        if not torch.equal(ya, yb):
            return f"{cfg} step {t}: y differs (max {float((ya.float() - yb.float()).abs().max())})"
        kd, vd = b.dequantized_kv()
        if not (torch.equal(kd, a.k_cache) and torch.equal(vd, a.v_cache)):
            return f"{cfg} step {t}: cache contents differ"
        for name in ("pos", "mask", "cache_cts", "attn_history_num", "attn_history_denom"):
            if hasattr(a, name) and not torch.equal(getattr(a, name), getattr(b, name)):
                return f"{cfg} step {t}: buffer {name} differs"
    return ""


class Forced:
    """random.Random whose choice() call number i returns plan[i] where given (None: a small cache length, drawn)."""

    def __init__(self, seed, plan):
        self.r, self.plan, self.n = random.Random(seed), plan, 0

    def choice(self, seq):
        i, self.n = self.n, self.n + 1
        if i in self.plan:
            v = self.plan[i]
            return self.r.randint(70, 190) if v is None else v
        return self.r.choice(seq)

    def __getattr__(self, name):
        return getattr(self.r, name)


# family: (function, {index of the choice() call: forced value}, index of the strategy's choice() call or None)
SMALL_GRID = {
    "step": (one, {3: 8, 7: None}, 0),            # strategy, dtype, D, H, R, [two nested choices], S
    "ring": (one_ring, {3: 8, 5: None}, 0),       # strategy, dtype, D, H, R, S
    "hybrid": (one_hybrid_step, {1: 8, 4: None}, None),  # dtype, H, R, [nested], S
    "quant": (one_quant, {2: 8, 5: None}, 0),     # strategy, dtype, H, R, [nested], S
}


def small_grid_stress(family, n, strategy=None, seed0=7000, extra=None):
    """The fuzz families with 8 kv heads and 70 .. 190 slots — one or two splits per kv head: 8-16 workgroups, where late-waking XCDs
    expose ordering holes between workgroups (how the shared key row of round 3 was found).  -> (cases run, list of mismatches)."""
    fn, plan, si = SMALL_GRID[family]
    plan = dict(plan)
    if strategy is not None:
        plan[si] = strategy
    plan.update(extra or {})
    ran, bad = 0, []
    for i in range(n):
        r = fn(Forced(seed0 + i, plan), i)
        if r is None:
            continue
        ran += 1
        if r:
            bad.append(r)
    return ran, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", choices=["hybrid_step"], default=None, help="run one family of cases only")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    ran = bad = 0
    for i in range(a.n):
        try:
            r = one_hybrid_step(rng, i) if a.only == "hybrid_step" else (one_hybrid_step(rng, i) if i % 8 == 4 else one_ring(rng, i)) if i % 4 == 0 else (one_quant(rng, i) if i % 4 == 1 else one(rng, i))
        except Exception as e:  # a crash is a finding too
            r = f"case {i}: {type(e).__name__}: {e}"
        if r is None:
            continue
        ran += 1
        if r:
            bad += 1
            print("MISMATCH", r, flush=True)
    from cold_compress_amd.attention_utils import single_launch_status

    st = single_launch_status()
    print(f"fuzz: {ran} cases ran, {bad} mismatches, single-launch hand-off timeouts: {st}")
    sys.exit(1 if bad or st else 0)


if __name__ == "__main__":
    main()
