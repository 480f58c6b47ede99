"""The layer step with the layer's QKV projection folded in (cc_decode_step_qkv_rc, r5) against its two-launch twin —
cc_gemv_fused (RMSNorm prologue, RoPE epilogue) followed by the single-launch decode_step — on twin caches: the projection,
h = x + delta, the attention output and EVERY cache buffer must be bit-identical, step after step (the fused launch runs
cc_gemv_fused's arithmetic in its order and the same step code on the same plan).  The twin itself is pinned to the oracle
and to the three-call sequence by tests/test_gpu_fused_step.py; the last test here closes the loop against the oracle directly.

ref: model.py:375-387, 389-427, 452-457; cache.py:690-765."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from helpers import to_np

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(kind, H, S, D, dtype, g=4, w=10):
    import cold_compress_amd.cache as cache

    common = dict(max_cache_length=S, max_seq_length=4 * S, cache_bits=None)
    with torch.device(DEV):
        if kind == "heavy_hitter":
            return cache.KVCacheHeavyHitter(1, H, D, dtype, global_tokens=g, history_window_size=1, recent_window=w, attn_thresholding=False,
                                            **common)
        if kind == "recent_global":
            return cache.KVCacheRecentGlobal(1, H, D, dtype, global_tokens=g, **common)
        if kind == "full":
            return cache.KVCacheFull(1, H, D, dtype, **common)
        if kind == "random":
            return cache.KVCacheRandom(1, H, D, dtype, global_tokens=g, recent_window=w, **common)
    raise ValueError(kind)


def _seed(kv, gen, T):
    H, D = kv.n_heads, kv.head_dim
    dt = kv.k_cache.dtype
    kv.update_kv(torch.arange(T, device=DEV), torch.randn(1, H, T, D, generator=gen).to(dt).to(DEV),
                 torch.randn(1, H, T, D, generator=gen).to(dt).to(DEV), True)
    if hasattr(kv, "attn_history_num"):
        kv.attn_history_num[0, :, :T, 0] = torch.rand(H, T, generator=gen, dtype=torch.float64).to(DEV)
        kv.attn_history_denom[0, :, :T] = torch.randint(1, 5, (H, T), generator=gen, dtype=torch.int32).to(DEV)
        kv.attn_history_num[0, :, 50:60, 0] = 0.0  # engineered ties


def _layer(gen, HQ, H, D, K, dtype, bias):
    N = (HQ + 2 * H) * D
    w = (0.02 * torch.randn(N, K, generator=gen)).to(dtype).to(DEV)
    nw = (1.0 + 0.1 * torch.randn(K, generator=gen)).to(dtype).to(DEV)
    b = (0.1 * torch.randn(N, generator=gen)).to(dtype).to(DEV) if bias else None
    return w, nw, b


def _freqs(p, D, dtype):
    f = 1.0 / (500000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = float(p) * f
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(dtype).to(DEV).contiguous()  # [D / 2, 2]


def _state_equal(a, b, t):
    for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
        assert na == nb
        assert torch.equal(ta, tb), f"step {t}: buffer {na} differs"


SHAPES = [
    # kind, dtype, H, HQ, S, D, K, T (prefilled), steps, bias
    ("heavy_hitter", torch.bfloat16, 8, 32, 4096, 128, 4096, 4090, 24, False),   # the headline shape (C3): 6 units per workgroup
    ("heavy_hitter", torch.bfloat16, 8, 32, 2560, 128, 4096, 2560, 16, False),   # C2: 20 splits -> 9 / 10 units, two rounds
    ("recent_global", torch.bfloat16, 8, 32, 4096, 128, 4096, 4096, 12, False),
    ("full", torch.bfloat16, 8, 32, 4096, 128, 4096, 4000, 12, False),
    ("random", torch.bfloat16, 8, 32, 4096, 128, 4096, 4096, 12, False),
    ("heavy_hitter", torch.float16, 8, 32, 4096, 128, 4096, 4090, 10, True),     # fp16, with a projection bias (Qwen2)
    ("heavy_hitter", torch.bfloat16, 8, 64, 4096, 128, 4096, 4000, 10, False),    # 8 query heads per kv head
    ("heavy_hitter", torch.bfloat16, 8, 32, 3000, 128, 2048, 2990, 10, False),    # ragged last split, a shorter model dim (one segment per wave)
    ("heavy_hitter", torch.bfloat16, 4, 16, 4096, 128, 2048, 4096, 10, False),    # 4 kv heads (a TP = 2 rank): memory hand-off, 4-wave workgroups
    ("recent_global", torch.float16, 8, 64, 2560, 128, 3584, 2560, 10, True),
]


@pytest.mark.parametrize("kind,dtype,H,HQ,S,D,K,T,steps,bias", SHAPES)
def test_qkv_step_equals_gemv_then_step(kind, dtype, H, HQ, S, D, K, T, steps, bias):
    from cold_compress_amd.harness import glue

    a, b = _mk(kind, H, S, D, dtype), _mk(kind, H, S, D, dtype)
    if not b.qkv_step_available(HQ, K):
        pytest.skip("shape not eligible for the QKV form of the step on this device")
    if kind == "random":
        a._rng_seed = b._rng_seed = 0x1234567  # (the same in-kernel draws on both)
    for kv in (a, b):
        _seed(kv, torch.Generator().manual_seed(31), T)
    gen = torch.Generator().manual_seed(7)
    w, nw, bv = _layer(gen, HQ, H, D, K, dtype, bias)
    for t in range(steps):
        pos = T + 5 + t
        p = torch.tensor([pos], dtype=torch.int32, device=DEV)
        x = torch.randn(1, 1, K, generator=gen).to(dtype).to(DEV)
        delta = torch.randn(1, 1, K, generator=gen).to(dtype).to(DEV) if t % 3 != 2 else None
        fr = _freqs(pos, D, dtype)
        ha = torch.zeros_like(x)
        qkv = glue.gemv_fused(w, x, delta=delta, norm_weight=nw, eps=1e-5, h_out=ha, bias=bv, freqs=fr, rope_rows=(HQ + H) * D, head_dim=D)
        q = qkv[: HQ * D].view(1, HQ, 1, D)
        k1 = qkv[HQ * D: (HQ + H) * D].view(1, H, 1, D)
        v1 = qkv[(HQ + H) * D:].view(1, H, 1, D)
        ya = a.decode_step(q, k1, v1, p)
        hb = torch.zeros_like(x)
        qkv_b = torch.zeros_like(qkv)
        yb = b.decode_step_qkv(w, bv, x, delta, nw, 1e-5, hb, fr, p, HQ, qkv_out=qkv_b)
        torch.cuda.synchronize()
        assert b.step_status(HQ) == 0 if hasattr(b, "step_status") else True
        assert torch.equal(qkv, qkv_b), f"step {t}: projection differs in {(qkv != qkv_b).sum().item()} of {qkv.numel()} values"
        assert torch.equal(ha, hb), f"step {t}: h = x + delta"
        assert torch.equal(ya, yb), f"step {t}: attention output (max diff {(ya.float() - yb.float()).abs().max().item()})"
        _state_equal(a, b, t)


def test_qkv_step_replays_in_a_hipgraph():
    """The fused launch captured once and replayed: *input_pos, x and the RoPE row are read from device memory, so replays advance —
    against eager steps of a twin cache."""
    from cold_compress_amd.harness import glue  # noqa: F401

    kind, dtype, H, HQ, S, D, K, T = "heavy_hitter", torch.bfloat16, 8, 32, 4096, 128, 4096, 4096
    a, b = _mk(kind, H, S, D, dtype), _mk(kind, H, S, D, dtype)
    if not b.qkv_step_available(HQ, K):
        pytest.skip("shape not eligible for the QKV form of the step on this device")
    for kv in (a, b):
        _seed(kv, torch.Generator().manual_seed(3), T)
    gen = torch.Generator().manual_seed(11)
    w, nw, _ = _layer(gen, HQ, H, D, K, dtype, False)
    xs = torch.randn(1, 1, K, generator=gen).to(dtype).to(DEV)
    x = torch.empty_like(xs)
    p = torch.tensor([T + 1], dtype=torch.int32, device=DEV)
    fr = _freqs(T + 1, D, dtype)
    h = torch.zeros_like(x)
    x.copy_(xs)
    b.prepare_decode(p)
    graph = torch.cuda.CUDAGraph()
    yb_buf = None
    # warm-up outside capture (workspace allocation, the XCD probe) on a third cache whose state is discarded
    twin = _mk(kind, H, S, D, dtype)
    _seed(twin, torch.Generator().manual_seed(3), T)
    twin.decode_step_qkv(w, None, x, None, nw, 1e-5, h, fr, p, HQ)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        yb_buf = b.decode_step_qkv(w, None, x, None, nw, 1e-5, h, fr, p, HQ)
    for t in range(8):
        pos = T + 1 + t
        p.fill_(pos)
        fr.copy_(_freqs(pos, D, dtype))
        x.copy_((xs.float() * (1.0 + 0.25 * t)).to(dtype))
        graph.replay()
        ya = a.decode_step_qkv(w, None, x, None, nw, 1e-5, torch.zeros_like(x), fr, p, HQ)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb_buf), f"replay {t}"
        _state_equal(a, b, t)


def test_qkv_step_vs_oracle_pipeline(oracle):
    """Closing the loop without the twin: the fused launch's projection against a float64 host evaluation of
    RMSNorm -> Linear -> RoPE (tolerance: one rounding of the model dtype around an fp32-accumulated sum), and its step
    against the oracle's pipeline twin (oracle/cc_oracle.c: cc_decode_step_heavy_hitter_cpu) fed the DEVICE's q / k / v:
    eviction slots, positions, denominators, K / V bit-exact; y within 1e-3 + two bf16 roundings."""
    o = oracle
    kind, dtype, H, HQ, S, D, K, T = "heavy_hitter", torch.bfloat16, 8, 32, 4096, 128, 4096, 4090
    g, w_ = 4, 10
    b = _mk(kind, H, S, D, dtype, g, w_)
    if not b.qkv_step_available(HQ, K):
        pytest.skip("shape not eligible for the QKV form of the step on this device")
    _seed(b, torch.Generator().manual_seed(13), T)
    gen = torch.Generator().manual_seed(19)
    w, nw, _ = _layer(gen, HQ, H, D, K, dtype, False)
    st = dict(k=to_np(b.k_cache.cpu()[0]), v=to_np(b.v_cache.cpu()[0]), pos=b.pos.cpu()[0].numpy().copy(),
              mask=b.mask.cpu()[0, :, 0].numpy().astype(np.uint8), cts=b.cache_cts.cpu().numpy().copy(),
              num=b.attn_history_num.cpu()[0, :, :, 0].numpy().copy(), denom=b.attn_history_denom.cpu()[0].numpy().copy(),
              ctr=np.zeros(1, np.int64))
    p0 = T + 2
    key = np.zeros((H, (S + 127) // 128), np.uint64)
    view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], 1)
    o.call("cc_hh_next_key_init", C.byref(view), o.ptr(np.array([p0], np.int32)), o.ptr(st["num"]), o.ptr(st["denom"]), g, w_, o.ptr(key), None)
    for t in range(6):
        pos = p0 + t
        p = torch.tensor([pos], dtype=torch.int32, device=DEV)
        x = torch.randn(1, 1, K, generator=gen).to(dtype).to(DEV)
        delta = torch.randn(1, 1, K, generator=gen).to(dtype).to(DEV)
        fr = _freqs(pos, D, dtype)
        qkv = torch.zeros((HQ + 2 * H) * D, dtype=dtype, device=DEV)
        y = b.decode_step_qkv(w, None, x, delta, nw, 1e-5, None, fr, p, HQ, qkv_out=qkv)
        torch.cuda.synchronize()
        # ---- the projection, in float64 on the host
        hsum = (x.float() + delta.float()).to(dtype).double().cpu().view(-1)
        rs = 1.0 / torch.sqrt((hsum * hsum).mean() + 1e-5)
        n = ((hsum * rs).to(dtype).double() * nw.double().cpu()).to(dtype).double()
        lin = (w.double().cpu() @ n).to(dtype).double()
        fc = fr.double().cpu()
        rows = lin[: (HQ + H) * D].view(-1, D // 2, 2)
        rot = torch.stack([rows[..., 0] * fc[:, 0] - rows[..., 1] * fc[:, 1], rows[..., 1] * fc[:, 0] + rows[..., 0] * fc[:, 1]], dim=-1)
        want = torch.cat([rot.reshape(-1), lin[(HQ + H) * D:]]).float()
        got = qkv.float().cpu()
        tol = 2.0 ** -7 * want.abs().clamp_min(2.0 ** -6) + 2e-3  # one bf16 rounding of the Linear, one of the rotation, fp32 sums
        assert bool(((got - want).abs() <= tol).all()), f"step {t}: projection off by {(got - want).abs().max().item()}"
        # ---- the step, by the oracle on the device's q / k / v
        qn = to_np(qkv[: HQ * D].view(HQ, D).cpu())
        kn = to_np(qkv[HQ * D: (HQ + H) * D].view(H, D).cpu())
        vn = to_np(qkv[(HQ + H) * D:].view(H, D).cpu())
        view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], 1)
        yo = np.zeros((HQ, D), np.uint16)
        o.call("cc_decode_step_heavy_hitter", C.byref(view), o.ptr(qn), o.ptr(kn), o.ptr(vn), o.ptr(np.array([pos], np.int32)),
               o.ptr(st["num"]), o.ptr(st["denom"]), o.ptr(st["ctr"]), o.ptr(key), g, w_, HQ, 1.0 / math.sqrt(D), o.ptr(yo), None, None, 0, None)
        assert np.array_equal(b.pos.cpu()[0].numpy(), st["pos"]), f"step {t}: eviction slot"
        assert np.array_equal(b.attn_history_denom.cpu()[0].numpy(), st["denom"]), f"step {t}: denominators"
        yr = torch.from_numpy(yo.view(np.int16).copy()).view(dtype).float()
        err = (y.cpu().float()[0, :, 0] - yr).abs().max().item()
        assert err < 1e-3 + 2.0 * 2.0 ** -8 * float(yr.abs().max()), f"step {t}: y differs from the oracle by {err}"
    assert np.allclose(b.attn_history_num.cpu()[0, :, :, 0].numpy(), st["num"], rtol=2 * 2.0 ** -8, atol=6 * 2.0 ** -16)
    assert np.array_equal(to_np(b.k_cache.cpu()[0]), st["k"]) and np.array_equal(to_np(b.v_cache.cpu()[0]), st["v"])


@pytest.mark.parametrize("strategy", ["heavy_hitter", "recent_global"])
def test_decode_loop_with_the_qkv_form_equals_the_default_loop(strategy):
    """The harness decode loop with Attention.fuse_qkv_step on (ONE launch for norm + wqkv + RoPE + the step) against the default
    loop on a twin model: 12 greedy tokens after a 4300-token prompt compacted to 4096 slots — tokens, probabilities and every
    cache buffer bit-identical (Llama-3-8B's attention shape, two layers, a small FFN and vocabulary)."""
    import argparse

    import cold_compress_amd.cache as cache
    from cold_compress_amd.harness import ModelArgs, Transformer, decode_one_token, generate, prefill, setup_caches

    cfg = dict(block_size=8192, vocab_size=512, n_layer=2, n_head=32, n_local_heads=8, dim=4096, intermediate_size=1024, rope_base=500000)
    torch.manual_seed(5)
    ref = Transformer(ModelArgs(**cfg)).to(torch.bfloat16).eval()
    for p_ in ref.parameters():
        p_.data.normal_(0.0, 0.02)
    models = []
    for fuse in (False, True):
        m = Transformer(ModelArgs(**cfg)).to(torch.bfloat16).eval()
        m.load_state_dict(ref.state_dict())
        m = m.to(DEV)
        ap = argparse.ArgumentParser()
        cache.add_cache_arguments(ap)
        kw = vars(ap.parse_args([]))
        kw.update(cache_strategy=[strategy], prompt_compression_strategy=[strategy], max_cache_length=[4096], global_tokens=4, recent_window=10)
        setup_caches(m, None, DEV, 4300 + 16, dict(kw))
        for l in m.layers:
            l.attention.fuse_qkv_step = fuse
        models.append(m)
    assert models[1].layers[0].attention.kv_cache.qkv_step_available(32, 4096) or pytest.skip("QKV form not eligible on this device")
    prompt = torch.randint(0, 512, (4300,), generator=torch.Generator().manual_seed(2), dtype=torch.int32).to(DEV)
    outs = []
    for m in models:
        seq, probs, _ = generate(m, prompt, prefill, decode_one_token, max_new_tokens=12)
        torch.cuda.synchronize()
        outs.append((seq, probs))
    assert torch.equal(outs[0][0], outs[1][0]), "generated tokens"
    for t, (pa, pb) in enumerate(zip(outs[0][1], outs[1][1])):
        assert torch.equal(pa, pb), f"token {t}: probabilities"
    for la, lb in zip(models[0].layers, models[1].layers):
        _state_equal(la.attention.kv_cache, lb.attention.kv_cache, "end")
