"""Extreme shapes of the two-launch decode step against update_kv -> attention -> update_state, every buffer bit for
bit: caches of 1 .. 129 slots (empty, partly filled, full), 131072 slots, 32 query heads per kv head, fp32 with a
small head_dim, and KVCacheFull constructed with a `global_tokens` kwarg (ignored by its arg-min, cache.py:502)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run(strategy, dtype, H, HQ, S, D, T, steps=6, W=1):
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    cls, rk = cache.get_cache_constructor(strategy)
    kw = dict(max_cache_length=S, global_tokens=min(2, S // 2), recent_window=min(3, max(S // 4, 1)), history_window_size=W,
              attn_thresholding=False, max_seq_length=4 * S + 64, cache_bits=None)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    from cold_compress_amd import _abi

    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    one = (strategy in ("recent_global", "full", "random") or (strategy == "heavy_hitter" and W == 1)) and \
        _abi.lib()["cc_decode_step_single_launch"](HQ, H, S, D, code) == 1
    gen = torch.Generator().manual_seed(3)
    if strategy == "random":  # the same uniform draws for both caches (the two-launch step draws one step ahead)
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
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        fuse = strategy == "heavy_hitter" and W == 1
        ya, at = sdpa(q, ka, va, attn_mask=ma, return_attn=a.return_attn() and not fuse, group_mean=True,
                      history=a.fused_history() if fuse else None)
        if fuse:
            a._state_fused = True
        a.update_state(p, k1, v1, False, at)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        if one:  # the single-launch step folds y's partial sums in its own fixed order: one rounding apart at most
            assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), (strategy, S, t)
        else:
            assert torch.equal(ya, yb), (strategy, S, t)
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), (strategy, S, t, na)


@pytest.mark.parametrize("strategy", ["heavy_hitter", "recent_global", "l2", "full", "random"])
@pytest.mark.parametrize("S,T", [(1, 0), (2, 1), (5, 5), (17, 0), (63, 63), (129, 100)])
def test_tiny_caches(strategy, S, T):
    if strategy != "full" and S < 5:
        pytest.skip("sinks + recent window need a few slots")
    run(strategy, torch.bfloat16, 2, 8, S, 128, T)


@pytest.mark.parametrize("strategy", ["heavy_hitter", "l2"])
def test_cache_of_131072_slots(strategy):
    run(strategy, torch.bfloat16, 8, 32, 131072, 128, 131072, steps=3)


def test_32_query_heads_per_kv_head():
    run("heavy_hitter", torch.float16, 1, 32, 300, 128, 290)


def test_fp32_small_head_dim_single_query_head():
    run("heavy_hitter", torch.float32, 3, 3, 50, 32, 45)


@pytest.mark.parametrize("dtype,H,HQ,S,D,T,W,steps", [(torch.bfloat16, 8, 32, 1024, 128, 1024, 8, 20), (torch.float32, 2, 4, 77, 16, 60, 3, 30),
                                                      (torch.float16, 3, 12, 300, 128, 300, 33, 12), (torch.bfloat16, 1, 8, 40, 128, 0, 2, 50)])
def test_finite_history_window_two_launch_step(dtype, H, HQ, S, D, T, W, steps):
    """KVCacheHeavyHitter with history_window_size W > 1: cc_decode_step_heavy_hitter_ring (eviction scored from the
    tracked window sums in the combine pass; the refilled slot's ring row / shadow / accumulator restart from zero
    there) against update_kv -> attention -> update_state; ring, window sums, accumulators, everything bit for bit,
    across ring wrap-arounds."""
    run("heavy_hitter", dtype, H, HQ, S, D, T, steps=steps, W=W)
