"""Stress one fuzz configuration many times (different data each time); on a y mismatch also report which buffers differ."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import cold_compress_amd.cache as cache
from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa
DEV = "cuda"
strategy, dtype, H, R, S, D, T, g, w = sys.argv[1], getattr(torch, sys.argv[2]), *[int(x) for x in sys.argv[3:10]]
N = int(sys.argv[10]) if len(sys.argv) > 10 else 500
cls, rk = cache.get_cache_constructor(strategy)
kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, history_window_size=1, attn_thresholding=False, max_seq_length=4 * S + 64, cache_bits=None)
bad = 0
for it in range(N):
    with torch.device(DEV):
        a, b = cls(1, H, D, dtype, **{k: kw[k] for k in rk}), cls(1, H, D, dtype, **{k: kw[k] for k in rk})
    gen = torch.Generator().manual_seed(1000 + it)
    if strategy == "random":
        draws = [torch.rand(S, generator=gen).to(DEV) for _ in range(8)]
        ia, ib = iter(draws), iter(draws)
        a._rand = lambda: next(ia); b._rand = lambda: next(ib)
    if T > 0:
        k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV); v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
        for kv in (a, b):
            kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
            if strategy == "l2":
                kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None)
    for t in range(6):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV); v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, H * R, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        fuse = strategy == "heavy_hitter"
        ya, at = sdpa(q, ka, va, attn_mask=ma, return_attn=a.return_attn() and not fuse, group_mean=True, history=a.fused_history() if fuse else None)
        if fuse:
            a._state_fused = True
        a.update_state(p, k1, v1, False, at)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        if not torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6):
            d = (ya.float() - yb.float()).abs()[0, :, 0]
            heads = (d.max(dim=1).values > 1e-3).nonzero().reshape(-1).tolist()
            diffb = [na for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()) if na not in ("next_key", "step_commit") and not torch.equal(ta, tb)]
            print(f"it {it} step {t}: y max diff {float(d.max()):.4g} in query heads {heads}; differing buffers {diffb}", flush=True)
            if "pos" in diffb:
                print("   pos a", a.pos.cpu()[0].reshape(-1)[:0].tolist(), (a.pos != b.pos).nonzero().tolist()[:6])
            bad += 1
            break
print(f"{strategy} {dtype} H={H} R={R} S={S}: {bad} bad of {N}")
