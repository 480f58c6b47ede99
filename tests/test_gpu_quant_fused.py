"""The opt-in FUSED quantised cache (cache_quant_mode="fused"; include/coldcompress.h): uint8 images on a per-(head, slot)
grid, dequantised inside the decode kernels.  It is the build's own numerical contract, so it is pinned three ways:
  * cc_kv_quant_rows / cc_kv_dequant_rows against the oracle's twins, bit for bit;
  * the fused-quant decode step against the SAME policy's 16-bit step fed with the dequantised values (every buffer and
    the attention output bit for bit: after the dequantisation in registers the arithmetic is the 16-bit step's);
  * cc_decode_step_quant against the oracle's twin (cc_decode_step_quant_cpu) through the C ABI.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from helpers import DT_CODE, from_np, to_np

pytestmark = pytest.mark.gpu
DEV = __import__("helpers").TEST_DEVICE  # "cuda"; "cpu" only under tests/cpu_twin.py


def _abi():
    from cold_compress_amd import _abi

    return _abi


def _p(t):
    return C.c_void_p(t.data_ptr())


def _round_trip_rows(x):
    """[N, D] model-dtype rows -> the values the fused cache holds for them (device quantise + dequantise)."""
    abi = _abi()
    N, D = x.shape
    kq, vq = torch.empty((N, D), dtype=torch.uint8, device=DEV), torch.empty((N, D), dtype=torch.uint8, device=DEV)
    par = torch.empty((N, 4), dtype=torch.float32, device=DEV)
    ko, vo = torch.empty_like(x), torch.empty_like(x)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    abi.call("cc_kv_quant_rows", _p(x), _p(x), 1, N, D, DT_CODE[x.dtype], 8, _p(kq), _p(vq), _p(par), st)
    abi.call("cc_kv_dequant_rows", _p(kq), _p(vq), _p(par), 1, N, D, DT_CODE[x.dtype], 8, _p(ko), _p(vo), st)
    return ko


@pytest.mark.parametrize("dtype,H,S,D", [(torch.bfloat16, 8, 513, 128), (torch.float16, 3, 100, 128), (torch.float32, 2, 65, 64),
                                         (torch.bfloat16, 1, 7, 200)])
def test_quant_rows_equal_oracle(oracle, dtype, H, S, D):
    o, abi = oracle, _abi()
    gen = torch.Generator().manual_seed(H * 1000 + S)
    k = (torch.randn(H, S, D, generator=gen) * torch.rand(H, S, 1, generator=gen) * 4).to(dtype)
    v = torch.randn(H, S, D, generator=gen).to(dtype)
    k[0, 0] = 0  # a constant row: scale clamps at 1e-6 / 255
    v[0, 1] = 3.0
    k[H - 1, S - 1, 0] = -0.0
    kd, vd = k.to(DEV), v.to(DEV)
    kq, vq = torch.empty((H, S, D), dtype=torch.uint8, device=DEV), torch.empty((H, S, D), dtype=torch.uint8, device=DEV)
    par = torch.empty((H, S, 4), dtype=torch.float32, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    abi.call("cc_kv_quant_rows", _p(kd), _p(vd), H, S, D, DT_CODE[dtype], 8, _p(kq), _p(vq), _p(par), st)
    ko, vo = torch.empty_like(kd), torch.empty_like(vd)
    abi.call("cc_kv_dequant_rows", _p(kq), _p(vq), _p(par), H, S, D, DT_CODE[dtype], 8, _p(ko), _p(vo), st)
    torch.cuda.synchronize()
    kn, vn = to_np(k), to_np(v)
    kq_o, vq_o = np.zeros((H, S, D), np.uint8), np.zeros((H, S, D), np.uint8)
    par_o = np.zeros((H, S, 4), np.float32)
    o.call("cc_kv_quant_rows", o.ptr(kn), o.ptr(vn), H, S, D, DT_CODE[dtype], 8, o.ptr(kq_o), o.ptr(vq_o), o.ptr(par_o), None)
    ko_o, vo_o = np.zeros_like(kn), np.zeros_like(vn)
    o.call("cc_kv_dequant_rows", o.ptr(kq_o), o.ptr(vq_o), o.ptr(par_o), H, S, D, DT_CODE[dtype], 8, o.ptr(ko_o), o.ptr(vo_o), None)
    assert np.array_equal(kq.cpu().numpy(), kq_o) and np.array_equal(vq.cpu().numpy(), vq_o)
    assert np.array_equal(par.cpu().numpy(), par_o)  # value equality: -0.0 == 0.0
    assert torch.equal(ko.cpu(), from_np(ko_o, dtype)) and torch.equal(vo.cpu(), from_np(vo_o, dtype))
    # the grid spans the row: dequantised values within half a step of the originals (+ one rounding of the dtype)
    step = par.cpu()[..., 0:1]
    assert ((ko.cpu().float() - k.float()).abs() <= 0.51 * step + k.float().abs() * 2.0 ** -7 + 1e-6).all()


def _mk(strategy, H, S, D, dtype, fused, g=4, w=10):
    import cold_compress_amd.cache as cache

    cls, rk = cache.get_cache_constructor(strategy)
    kw = dict(max_cache_length=S, max_seq_length=4 * S, cache_bits=8 if fused else None, global_tokens=g, recent_window=w,
              history_window_size=1, attn_thresholding=False)
    lk = {k: kw[k] for k in rk}
    if fused:
        lk["cache_quant_mode"] = "fused"
    with torch.device(DEV):
        return cls(1, H, D, dtype, **lk)


@pytest.mark.parametrize("strategy", ["heavy_hitter", "recent_global", "full", "random"])
@pytest.mark.parametrize("dtype,H,HQ,S,T", [(torch.bfloat16, 8, 32, 4096, 4090), (torch.float16, 2, 16, 300, 290),
                                            (torch.bfloat16, 1, 8, 3488, 3488), (torch.bfloat16, 3, 12, 1001, 700)])
@pytest.mark.parametrize("single", [False, True])
def test_fused_quant_step_equals_16bit_step_on_dequantised_values(strategy, dtype, H, HQ, S, T, single):
    """`b` = the fused quantised cache, `a` = the same policy's 16-bit cache holding b's DEQUANTISED values and fed with the
    round trip of every new token: every buffer, the history and the attention output must agree bit for bit, step after
    step, in the two-launch form and (where the shape allows) as one launch."""
    abi = _abi()
    D = 128
    abi.lib()["cc_decode_step_set_single_launch"](1 if single else 0)
    try:
        a, b = _mk(strategy, H, S, D, dtype, False), _mk(strategy, H, S, D, dtype, True)
        for kv in (a, b):
            if hasattr(kv, "single_launch"):
                kv.single_launch = single
        gen = torch.Generator().manual_seed(31)
        k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
        v0 = (2.0 * torch.randn(1, H, T, D, generator=gen)).to(dtype).to(DEV)
        for kv in (a, b):
            kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
            if strategy == "heavy_hitter":
                g2 = torch.Generator().manual_seed(32)
                kv.attn_history_num[0, :, :T, 0] = torch.rand(H, T, generator=g2, dtype=torch.float64).to(DEV)
                kv.attn_history_denom[0, :, :T] = torch.randint(1, 5, (H, T), generator=g2, dtype=torch.int32).to(DEV)
        kd, vd = b.dequantized_kv()
        a.k_cache.copy_(kd)
        a.v_cache.copy_(vd)
        assert b.memory_usage() < 0.56 * a.memory_usage() + 1e-4 or strategy == "heavy_hitter"
        if strategy == "random":
            draws = [torch.rand(S, generator=gen).to(DEV) for _ in range(20)]
            for kv in (a, b):
                it = iter(list(draws))
                kv._rand = lambda it=it: next(it)
        for t in range(12):
            p = torch.tensor([T + 5 + t], dtype=torch.int32, device=DEV)
            k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
            v1 = (2.0 * torch.randn(1, H, 1, D, generator=gen)).to(dtype).to(DEV)
            q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
            kh = _round_trip_rows(k1.reshape(H, D)).view(1, H, 1, D)
            vh = _round_trip_rows(v1.reshape(H, D)).view(1, H, 1, D)
            ya = a.decode_step(q, kh, vh, p)
            yb = b.decode_step(q, k1, v1, p)
            torch.cuda.synchronize()
            if not torch.equal(ya, yb):
                d = (ya.float() - yb.float()).abs()[0, :, 0]
                kd, vd = b.dequantized_kv()
                raise AssertionError(f"step {t}: attention output: max |dy| {float(d.max()):.3e} on query heads "
                                     f"{d.amax(dim=1).nonzero().flatten().tolist()}; K rows differing "
                                     f"{(kd != a.k_cache).any(-1).nonzero().tolist()[:6]}; V rows {(vd != a.v_cache).any(-1).nonzero().tolist()[:6]}")
            kd, vd = b.dequantized_kv()
            assert torch.equal(kd, a.k_cache) and torch.equal(vd, a.v_cache), f"step {t}: cache contents"
            for name in ("pos", "mask", "cache_cts", "attn_history_num", "attn_history_denom", "attn_counter"):
                if hasattr(a, name):
                    assert torch.equal(getattr(a, name), getattr(b, name)), f"step {t}: {name}"
        if strategy == "heavy_hitter":
            assert b.step_status(HQ) == 0
            if single and S == 4096:
                assert b.single_launch_active(HQ)
    finally:
        abi.lib()["cc_decode_step_set_single_launch"](1)


def test_fused_quant_step_vs_oracle(oracle):
    """cc_decode_step_quant against the oracle's twin through the C ABI: images, row parameters, positions, counts and
    denominators bit for bit; the attention output within the 16-bit tolerance of the oracle's dot-product order."""
    o = oracle
    H, HQ, S, D, g, w, T = 2, 8, 384, 128, 4, 10, 380
    dtype = torch.bfloat16
    kv = _mk("heavy_hitter", H, S, D, dtype, True, g, w)
    gen = torch.Generator().manual_seed(41)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype)
    kv.update_kv(torch.arange(T, device=DEV), k0.to(DEV), v0.to(DEV), True)
    kv.attn_history_num[0, :, :T, 0] = torch.rand(H, T, generator=gen, dtype=torch.float64).to(DEV)
    kv.attn_history_denom[0, :, :T] = torch.randint(1, 5, (H, T), generator=gen, dtype=torch.int32).to(DEV)
    torch.cuda.synchronize()
    st = dict(kq=kv.k_cache_q.cpu()[0].numpy().copy(), vq=kv.v_cache_q.cpu()[0].numpy().copy(), par=kv.kv_qparams.cpu()[0].numpy().copy(),
              pos=kv.pos.cpu()[0].numpy().copy(), mask=kv.mask.cpu()[0, :, 0].numpy().astype(np.uint8), cts=kv.cache_cts.cpu().numpy().copy(),
              num=kv.attn_history_num.cpu()[0, :, :, 0].numpy().copy(), denom=kv.attn_history_denom.cpu()[0].numpy().copy(),
              ctr=np.zeros(1, np.int64))
    # the oracle's prefill images equal the device's
    kq_o, vq_o, par_o = np.zeros_like(st["kq"]), np.zeros_like(st["vq"]), np.zeros_like(st["par"])
    kfull, vfull = np.zeros((H, S, D), np.uint16), np.zeros((H, S, D), np.uint16)
    kfull[:, :T], vfull[:, :T] = to_np(k0[0]), to_np(v0[0])
    o.call("cc_kv_quant_rows", o.ptr(kfull), o.ptr(vfull), H, S, D, 1, 8, o.ptr(kq_o), o.ptr(vq_o), o.ptr(par_o), None)
    assert np.array_equal(kq_o, st["kq"]) and np.array_equal(vq_o, st["vq"]) and np.array_equal(par_o, st["par"])
    p0 = T + 9
    nk = (S + 127) // 128
    key = np.zeros((H, nk), np.uint64)  # the oracle keeps one partial key per 128 slots
    view = o.view(st["kq"], st["vq"], st["pos"], st["mask"], st["cts"], 1)
    o.call("cc_hh_next_key_init", C.byref(view), o.ptr(np.array([p0], np.int32)), o.ptr(st["num"]), o.ptr(st["denom"]), g, w, o.ptr(key), None)
    for t in range(8):
        p = torch.tensor([p0 + t], dtype=torch.int32)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype)
        y = kv.decode_step(q.to(DEV), k1.to(DEV), v1.to(DEV), p.to(DEV))
        torch.cuda.synchronize()
        view = o.view(st["kq"], st["vq"], st["pos"], st["mask"], st["cts"], 1)
        yo = np.zeros((HQ, D), np.uint16)
        o.call("cc_decode_step_quant", C.byref(view), o.ptr(st["par"]), 8, 1, o.ptr(to_np(q.reshape(HQ, D))), o.ptr(to_np(k1.reshape(H, D))),
               o.ptr(to_np(v1.reshape(H, D))), o.ptr(p.numpy().copy()), o.ptr(st["num"]), o.ptr(st["denom"]), o.ptr(st["ctr"]), None,
               o.ptr(key), g, w, HQ, 1.0 / math.sqrt(D), o.ptr(yo), None, None, 0, None, 3)
        assert np.array_equal(kv.pos.cpu()[0].numpy(), st["pos"]), f"step {t}: slots"
        assert np.array_equal(kv.k_cache_q.cpu()[0].numpy(), st["kq"]) and np.array_equal(kv.v_cache_q.cpu()[0].numpy(), st["vq"]), f"step {t}"
        assert np.array_equal(kv.kv_qparams.cpu()[0].numpy(), st["par"]), f"step {t}: row parameters"
        yr = from_np(yo, dtype).float()
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11  # the attention contract: 1e-3 + two roundings of the output dtype
        assert (y.cpu().float()[0, :, 0] - yr).abs().max() <= 1e-3 + 2 * ulp * yr.abs().max()
    assert np.array_equal(kv.attn_history_denom.cpu()[0].numpy(), st["denom"])
    assert np.array_equal(kv.cache_cts.cpu().numpy(), st["cts"])
    assert np.allclose(kv.attn_history_num.cpu()[0, :, :, 0].numpy(), st["num"], rtol=0, atol=2e-2)


def test_fused_quant_mode_is_opt_in_and_loud():
    from cold_compress_amd import _abi as abi

    with pytest.raises(abi.ColdCompressError):
        _mk("l2", 2, 64, 128, torch.bfloat16, True)
    import cold_compress_amd.cache as cache

    with torch.device(DEV):
        with pytest.raises(abi.ColdCompressError):  # 4 bits: not in this mode
            cache.KVCacheRecentGlobal(1, 2, 128, torch.bfloat16, max_cache_length=64, max_seq_length=128, cache_bits=4, global_tokens=4,
                                      cache_quant_mode="fused")
        kv = cache.KVCacheRecentGlobal(1, 2, 128, torch.bfloat16, max_cache_length=64, max_seq_length=128, cache_bits=8, global_tokens=4,
                                       cache_quant_mode="fused")
    with pytest.raises(abi.ColdCompressError):  # the three-call decode path does not exist in this mode
        kv.update_kv(torch.tensor([3], device=DEV), torch.zeros(1, 2, 1, 128, dtype=torch.bfloat16, device=DEV),
                     torch.zeros(1, 2, 1, 128, dtype=torch.bfloat16, device=DEV), False)
    stats = kv.compute_statistics(torch.tensor(10))
    assert "working_cache_gb" not in stats and stats["cache_memory_gb"] > 0


def test_fused_quant_end_to_end_in_the_harness():
    """The whole loop (prefill through the HIP path, row quantisation of the compacted prompt, hipGraph decode over the uint8
    images) on a small Llama-shaped model: next-token distributions stay close to the unquantised run's (8-bit rows), the
    cache statistics report half the bytes, and the single-launch hand-off never times out."""
    from cold_compress_amd.harness import GraphedDecoder, ModelArgs, Transformer, decode_one_token, prefill, setup_caches

    dev = torch.device(DEV)
    cfg = dict(block_size=1024, vocab_size=512, n_layer=2, n_head=8, n_local_heads=2, dim=1024, intermediate_size=2048)
    torch.manual_seed(5)
    model = Transformer(ModelArgs(**cfg)).to(torch.bfloat16).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for n, p in model.named_parameters():
            p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.05, generator=g)
    model = model.to(dev)
    prompt = torch.randint(0, cfg["vocab_size"], (300,), generator=torch.Generator().manual_seed(3), dtype=torch.int32).to(dev)
    runs = {}
    for name, extra in (("bf16", {}), ("fused", {"cache_bits": 8, "cache_quant_mode": "fused"})):
        kw = dict(max_cache_length=[128.0], cache_bits=None, cache_length_pattern="tile", cache_strategy=["heavy_hitter"],
                  cache_strategy_pattern="tile", feed_long_prompts=False, prompt_compression_strategy=["heavy_hitter"], global_tokens=4,
                  recent_window=10, history_window_size=1, attn_thresholding=False, min_recovery_frac=0.9)
        kw.update(extra)
        setup_caches(model, None, dev, 400, kw)
        with torch.no_grad():
            tok, probs = prefill(model, prompt.view(1, -1), torch.arange(300, device=dev))
            pos = torch.tensor([300], dtype=torch.int32, device=dev)
            plist, toks = [probs.float().clone()], [int(tok)]
            cur = tok.view(1, 1).to(torch.int32)
            step = GraphedDecoder(model) if name == "fused" else decode_one_token
            for i in range(16):
                nt, pr = step(model, cur, pos)
                plist.append(pr.float().clone())
                toks.append(int(nt))
                # teacher-force the unquantised run's tokens: both runs see the same inputs
                cur = (nt if name == "bf16" else torch.tensor(runs["bf16"][0][len(toks) - 1], device=dev)).view(1, 1).to(torch.int32)
                pos += 1
        torch.cuda.synchronize()
        kv = model.layers[0].attention.kv_cache
        runs[name] = (toks, plist, kv.memory_usage(), kv)
    assert torch.equal(runs["bf16"][1][0], runs["fused"][1][0])  # prefill attends to the prompt's own k / v: identical
    worst = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(runs["bf16"][1], runs["fused"][1]))
    agree = sum(int(a == b) for a, b in zip(runs["bf16"][0], runs["fused"][0]))
    assert worst < 0.2 and agree >= len(runs["bf16"][0]) - 2, (worst, agree)
    assert runs["fused"][2] < 0.6 * runs["bf16"][2]
    kv = runs["fused"][3]
    assert kv.fused_quant and kv.k_cache.numel() == 0 and kv.step_status(cfg["n_head"]) == 0
