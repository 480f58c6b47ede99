"""Long differential run at the benchmark's shapes: the fused step (as the product runs it) against update_kv -> attention ->
update_state for `steps` decode steps per policy, every buffer bit for bit at every step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import cold_compress_amd.cache as cache
from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa, single_launch_status
DEV = "cuda"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
shapes = [(8, 32, 4096), (8, 32, 2560), (1, 8, 3488), (4, 16, 4096)]
tot_bad = 0
for H, HQ, S in shapes:
    for strategy in ["recent_global", "full", "random", "l2", "heavy_hitter"]:
        dtype, D, g, w = torch.bfloat16, 128, 4, 10
        cls, rk = cache.get_cache_constructor(strategy)
        kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, history_window_size=1, attn_thresholding=False, max_seq_length=8 * S, cache_bits=None)
        with torch.device(DEV):
            a, b = cls(1, H, D, dtype, **{k: kw[k] for k in rk}), cls(1, H, D, dtype, **{k: kw[k] for k in rk})
        gen = torch.Generator().manual_seed(H * 1000 + S)
        T = S - 50  # the cache fills up after 50 steps, then every step evicts
        if strategy == "random":
            draws = [torch.rand(S, generator=gen).to(DEV) for _ in range(steps + 2)]
            ia, ib = iter(draws), iter(draws)
            a._rand = lambda: next(ia); b._rand = lambda: next(ib)
        k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV); v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
        for kv in (a, b):
            kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
            if strategy == "l2":
                kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None)
        bad = None
        for t in range(steps):
            p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
            k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV); v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
            q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
            ka, va, ma = a.update_kv(p, k1, v1, False)
            fuse = strategy == "heavy_hitter"
            ya, at = sdpa(q, ka, va, attn_mask=ma, return_attn=a.return_attn() and not fuse, group_mean=True, history=a.fused_history() if fuse else None)
            if fuse:
                a._state_fused = True
            a.update_state(p, k1, v1, False, at)
            yb = b.decode_step(q, k1, v1, p)
            if t % 25 == 24 or t == steps - 1:
                torch.cuda.synchronize()
                diffb = [na for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()) if na not in ("next_key", "step_commit") and not torch.equal(ta, tb)]
                if diffb or not torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6):
                    bad = (t, diffb, float((ya.float() - yb.float()).abs().max()))
                    break
        tot_bad += bad is not None
        print(f"H={H} HQ={HQ} S={S} {strategy:14s} {'OK' if bad is None else 'MISMATCH ' + str(bad)}", flush=True)
print("hand-off status", single_launch_status(), "mismatching runs", tot_bad)
