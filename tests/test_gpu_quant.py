"""Quantised KV cache (--cache_bits {8,4,2}) on the MI355X: cc_kv_requant / cc_kv_dequant against the reference's
quantization_utils known answers, the quantised heavy-hitter / recent-global replays (the K/V attention sees at
every step and the final int8 / packed images are bit-exact), and one end-to-end run with cache_bits=8."""
import numpy as np
import pytest
import torch

from helpers import DT_FROM_NAME, load_golden
from test_gpu_e2e import _build, _log_evictions  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = __import__("helpers").TEST_DEVICE  # "cuda"; "cpu" only under tests/cpu_twin.py
TAGS = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def _bits(t):
    t = t.contiguous().cpu()
    return t.view(torch.int16) if t.dtype in (torch.bfloat16, torch.float16) else t


@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("nb", [8, 4, 2])
def test_requant_known_answers(tag, nb):
    import ctypes as C

    from cold_compress_amd import _abi

    f = load_golden("f9_quant_known_answers.npz")
    dt = TAGS[tag]
    x = f[f"x_{tag}"][0].to(DEV).contiguous()
    H, S, D = x.shape
    q = torch.zeros((H, S, D) if nb == 8 else (H * S * D * nb // 8,), dtype=torch.uint8, device=DEV)
    sc, zp = torch.zeros(S, dtype=dt, device=DEV), torch.zeros(S, dtype=dt, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dt]
    _abi.call("cc_kv_requant", p(x), p(q), p(sc), p(zp), H, S, D, code, nb, st)
    torch.cuda.synchronize()
    assert torch.equal(_bits(sc), _bits(f[f"scales_{tag}_{nb}"]))
    assert torch.equal(_bits(zp), _bits(f[f"zeros_{tag}_{nb}"]))
    assert torch.equal(q.cpu().view(-1), f[f"q_{tag}_{nb}"].view(-1))
    assert torch.equal(_bits(x), _bits(f[f"y_{tag}_{nb}"][0]))
    out = torch.empty_like(x)
    _abi.call("cc_kv_dequant", p(q), p(sc), p(zp), p(out), H, S, D, code, nb, st)
    torch.cuda.synchronize()
    assert torch.equal(_bits(out), _bits(x))


@pytest.mark.parametrize("name", [f"f9_quant_hh_{t}_{nb}.npz" for t in ("f32", "bf16") for nb in (8, 4, 2)]
                         + ["f9_quant_recent_global_f32_8.npz", "f9_quant_hh_long_bf16_8.npz", "f9_quant_hh_long_bf16_4.npz"])
def test_quantised_cache_replay_bit_exact(name):
    import cold_compress_amd.cache as cache

    f = load_golden(name)
    dt = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w, nb = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"], f["cache_bits"]
    cls, rk = cache.get_cache_constructor(f["strategy"])
    kw = dict(max_cache_length=S, global_tokens=g, max_seq_length=4 * S, cache_bits=nb, recent_window=w, history_window_size=1,
              attn_thresholding=False)
    with torch.device(DEV):
        kv = cls(1, H, D, dt, **{k: kw[k] for k in rk})
    kv.update_kv(torch.arange(T, device=DEV), f["k0"].to(DEV), f["v0"].to(DEV), True)
    kv.update_state(torch.arange(T, device=DEV), f["k0"].to(DEV), f["v0"].to(DEV), True, f["attn0"].to(DEV) if "attn0" in f else None)
    for t in range(f["steps"]):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k, v, _ = kv.update_kv(p, f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), False)
        assert torch.equal(_bits(k), _bits(f["k_ret"][t])), f"step {t}: K seen by attention"
        assert torch.equal(_bits(v), _bits(f["v_ret"][t])), f"step {t}: V seen by attention"
        kv.update_state(p, f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), False, f["attn"][t].to(DEV) if "attn" in f else None)
    kv.quantize_cache()
    torch.cuda.synchronize()
    assert torch.equal(kv.k_cache_q.cpu().view(torch.uint8).view(-1), f["final_k"].view(torch.uint8).view(-1))
    assert torch.equal(kv.v_cache_q.cpu().view(torch.uint8).view(-1), f["final_v"].view(torch.uint8).view(-1))
    for a, b in ((kv.k_scales, "k_scales"), (kv.k_zero_points, "k_zero_points"), (kv.v_scales, "v_scales"),
                 (kv.v_zero_points, "v_zero_points")):
        assert torch.equal(_bits(a), _bits(f[b])), b
    assert torch.equal(kv.pos.cpu(), f["final_pos"])


@pytest.mark.parametrize("graphed", [False, True])
def test_e2e_cache_bits_8(graphed, audit):
    """Tiny-Llama end-to-end with cache_bits=8 (fp32): tokens identical, logits within the north-star 1e-3, final scales / zero
    points to 1e-4, the final quantised images equal up to AT MOST 8 codes per image one rounding step off (a K / V value that sits on
    a rounding boundary behind this build's own fp32 GEMMs: see below; 0 on the committed fixture — the quantiser itself is pinned
    bit for bit by the known-answer and replay tests).  The fused two-launch decode step is in the loop."""
    from cold_compress_amd.harness import GraphedDecoder, decode_one_token, generate, prefill

    f = load_golden("f9_e2e_heavy_hitter_q8.npz")
    model, ck = _build(f, f["n_layer"])
    logits = []
    orig = model.forward

    def fwd(*a, **k):
        out = orig(*a, **k)
        logits.append(out[0, -1].detach().float().clone())
        return out

    if not graphed:
        model.forward = fwd
    dec = GraphedDecoder(model) if graphed else decode_one_token
    seq, _, _ = generate(model, f["prompt"].to(DEV), prefill, dec, max_new_tokens=f["new_tokens"])
    torch.cuda.synchronize()
    assert torch.equal(seq.cpu(), f["seq"])
    if not graphed:
        assert (torch.stack(logits).cpu() - f["logits"]).abs().max() < 1e-3
    off = 0
    for li, layer in enumerate(model.layers):
        kv = layer.attention.kv_cache
        kv.quantize_cache()
        # K/V rows come out of this build's own GEMMs / RoPE (fp32, equal to the reference's to ~1e-6): the fp32 grids agree to
        # rounding, and so do the 8-bit images — a code is round(x / scale), and a value that sits ON a rounding boundary flips its
        # code on that 1e-6 (found by running this test on reference-made fixtures from other seeds, r5: one code of 4096 off by
        # one at seed offset 5003, three at offset 11 with jittered shapes, none at 0 / 1000; tools/dbg/q8_fresh_seed_diag.py).
        # Accepted: single-step differences in at most 8 codes of an image (counted in the audit; a slot's scale and zero point
        # move by the same 1e-6, which shifts every code of the slot by a hair); the quantiser itself is pinned bit for bit by the
        # known-answer and replay tests above (same inputs on both sides).
        for nm, mine in (("K", kv.k_cache_q), ("V", kv.v_cache_q)):
            d = (mine.cpu().view(torch.uint8).to(torch.int16) - f[f"final_{nm.lower()}_L{li}"].view(torch.uint8).to(torch.int16)).abs()
            off += int((d > 0).sum())
            # (a small ABSOLUTE count — at most 8 codes of an image, observed 0 .. 3 — not a fraction of it: 0.5 % of an
            #  image was hundreds of codes, room enough to hide a quantiser regression end to end)
            assert int(d.max()) <= 1 and int((d > 0).sum()) <= 8, f"layer {li} {nm} image: {int((d > 0).sum())} codes differ, by up to {int(d.max())}"
        assert torch.allclose(kv.k_scales.cpu(), f[f"final_k_scales_L{li}"], rtol=1e-4, atol=1e-7)
        assert torch.allclose(kv.k_zero_points.cpu(), f[f"final_k_zero_points_L{li}"], rtol=1e-4, atol=1e-6)
        assert torch.equal(kv.pos.cpu(), f[f"final_pos_L{li}"])
        assert torch.equal(kv.attn_history_denom.cpu(), f[f"final_denom_L{li}"])
    audit(f"8-bit codes off by one rounding step = {off} (limit 8 per image)", rule="q8 boundary code", count=off, compared=sum(2 * l.attention.kv_cache.k_cache_q.numel() for l in model.layers), limit="8 per image")
    stats = model.get_cache_stats(f["prompt_len"], f["new_tokens"])
    assert abs(stats["compression_ratio_avg"] - f["compression_ratio_avg"]) < 1e-6


def test_requant_fuzz_vs_oracle(oracle):
    """40 seeded random shapes (heads, slots, head_dim, dtype, bits) and value patterns (normal, constant rows, zeros,
    huge spread, a slot of subnormals): device round trip == oracle, bit for bit — working cache, image, scales, zeros.
    Both device kernels are hit: one thread per 16-byte vector (H * D / vec <= 1024) and the element-wise fallback."""
    import ctypes as C
    import random

    from cold_compress_amd import _abi
    from helpers import DT_CODE, to_np

    rng = random.Random(5)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for i in range(40):
        dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        nb = rng.choice([8, 4, 2])
        H, S = rng.choice([1, 2, 3, 8, 32]), rng.randint(1, 200)
        D = rng.choice([8, 16, 64, 128, 256])
        gen = torch.Generator().manual_seed(900 + i)
        x = torch.randn(H, S, D, generator=gen)
        kind = rng.randrange(5)
        if kind == 1:
            x[:, ::3] = 0.75  # constant slots: zero range -> the 1e-6 floor
        elif kind == 2:
            x[:, 1::2] = 0.0
        elif kind == 3:
            x = x * torch.logspace(-6, 4, S).view(1, S, 1)
        elif kind == 4 and dt != torch.float16:
            x[:, 0] = 1e-39
        x = x.to(dt)
        code = DT_CODE[dt]
        work_o = to_np(x).copy()
        q_o = np.zeros(H * S * D * nb // 8, np.uint8)
        es = np.float32 if code == 0 else np.uint16
        sc_o, zp_o = np.zeros(S, es), np.zeros(S, es)
        oracle.call("cc_kv_requant", oracle.ptr(work_o), oracle.ptr(q_o), oracle.ptr(sc_o), oracle.ptr(zp_o), H, S, D, code, nb, None)
        work = x.to(DEV).contiguous()
        q = torch.zeros(H * S * D * nb // 8, dtype=torch.uint8, device=DEV)
        sc, zp = torch.zeros(S, dtype=dt, device=DEV), torch.zeros(S, dtype=dt, device=DEV)
        _abi.call("cc_kv_requant", p(work), p(q), p(sc), p(zp), H, S, D, code, nb, None)
        torch.cuda.synchronize()
        what = f"case {i}: {dt} H={H} S={S} D={D} bits={nb} kind={kind}"
        assert np.array_equal(to_np(work.cpu()), work_o), what + " (working cache)"
        assert np.array_equal(q.cpu().numpy(), q_o), what + " (image)"
        assert np.array_equal(to_np(sc.cpu()), sc_o) and np.array_equal(to_np(zp.cpu()), zp_o), what + " (scale / zero point)"


@pytest.mark.parametrize("nb", [8, 4])
def test_batched_round_trip_equals_per_cache(nb):
    """cc_kv_requant_batch (every layer's round trip as one launch behind the last layer: cache.flush_quantized, what the
    harness model runs) against the per-cache launches at the start of each cache's next update: five caches of three
    lengths, two policies, 40 decode tokens — working caches, images, scales, zero points and the stable-slot bookkeeping
    bit for bit after every token."""
    import cold_compress_amd.cache as cache

    H, D, T = 8, 128, 24
    specs = [("heavy_hitter", 64), ("recent_global", 96), ("heavy_hitter", 200), ("heavy_hitter", 64), ("recent_global", 130)]

    def build():
        out = []
        for strat, S in specs:
            cls, rk = cache.get_cache_constructor(strat)
            kw = dict(max_cache_length=S, global_tokens=2, max_seq_length=1024, cache_bits=nb, recent_window=8, history_window_size=1,
                      attn_thresholding=False)
            with torch.device(DEV):
                out.append(cls(1, H, D, torch.bfloat16, **{k: kw[k] for k in rk}))
        return out

    A, B = build(), build()
    gen = torch.Generator().manual_seed(77)
    rows = lambda n: (torch.randn(1, H, n, D, generator=gen) * 2).to(torch.bfloat16).to(DEV)  # noqa: E731

    def attn_for(kv, n):
        a = torch.rand(1, H, n, kv.max_cache_length if n == 1 else n, generator=gen)
        return (a / a.sum(-1, keepdim=True)).to(torch.bfloat16).to(DEV)

    names = ("k_cache", "v_cache", "k_cache_q", "v_cache_q", "k_scales", "v_scales", "k_zero_points", "v_zero_points", "pos",
             "_quant_stable", "_quant_pos_seen")
    pre = torch.arange(T, device=DEV)
    for i in range(len(specs)):
        k0, v0 = rows(T), rows(T)
        a0 = attn_for(A[i], T)
        for kv in (A[i], B[i]):
            kv.update_kv(pre, k0, v0, True)
            kv.update_state(pre, k0, v0, True, a0 if kv.return_attn() else None)
    cache.flush_quantized(B)
    for t in range(40):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        for i in range(len(specs)):
            k1, v1 = rows(1), rows(1)
            a1 = attn_for(A[i], 1)
            seen = []
            for kv in (A[i], B[i]):
                kc, vc, _ = kv.update_kv(p, k1, v1, False)
                seen.append((kc.clone(), vc.clone()))
                kv.update_state(p, k1, v1, False, a1 if kv.return_attn() else None)
            assert torch.equal(_bits(seen[0][0]), _bits(seen[1][0])) and torch.equal(_bits(seen[0][1]), _bits(seen[1][1])), \
                f"token {t}, cache {i}: K/V attention sees"
        cache.flush_quantized(B)
        assert not any(kv._quant_pending for kv in B)
        for kv in A:
            kv.quantize_cache()
        for i in range(len(specs)):
            for n in names:
                a, b = getattr(A[i], n), getattr(B[i], n)
                assert torch.equal(_bits(a) if a.is_floating_point() else a.cpu(), _bits(b) if b.is_floating_point() else b.cpu()), \
                    f"token {t}, cache {i} ({specs[i]}): {n}"
    # steady state: most slots are skipped
    assert all(int(kv._quant_stable.sum()) > kv.max_cache_length for kv in B)
