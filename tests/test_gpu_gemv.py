"""cc_gemv_fused (decode-time dense layers with the glue fused) against a plain PyTorch fp32 composition of the same
op chain and against the oracle's twin: plain, RMSNorm(x + delta) prologue, SwiGLU pair, RoPE epilogue, bias; bf16 /
fp16 / fp32; the Llama-3-8B decode shapes and ragged ones.  Tolerance: 2 ulp of the output dtype relative to the
largest output (the summation order of a dot product is the kernel's own), stated per case."""

import numpy as np
import pytest
import torch

from helpers import DT_CODE, to_np

pytestmark = pytest.mark.gpu
DEV = "cuda"
ULP = {torch.float32: 2e-6, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}


def _rnd(t, dt):
    return t.to(dt).float()


def _ref(W, x, dt, W3=None, delta=None, nw=None, eps=1e-5, bias=None, freqs=None, rope_rows=0, hd=0):
    """fp32 torch composition with a rounding to dt after every tensor op (model.py:317-327, 375-387, 442-457, 507-519)."""
    xf = x.float()
    h = None
    if nw is not None:
        h = _rnd(xf + delta.float(), dt) if delta is not None else xf
        n = _rnd(h * torch.rsqrt((h * h).mean() + eps), dt)
        xin = _rnd(n * nw.float(), dt)
    else:
        xin = xf
    t = W.float() @ xin
    if bias is not None:
        t = t + bias.float()
    t = _rnd(t, dt)
    if W3 is not None:
        t3 = _rnd(W3.float() @ xin, dt)
        t = _rnd(_rnd(torch.nn.functional.silu(t), dt) * t3, dt)
    if freqs is not None:
        f = freqs.float().view(-1, 2)
        rr = t[:rope_rows].view(-1, hd // 2, 2)
        c, s = f[:, 0].view(1, -1), f[:, 1].view(1, -1)
        out = torch.stack([rr[..., 0] * c - rr[..., 1] * s, rr[..., 1] * c + rr[..., 0] * s], -1).reshape(-1)
        t = torch.cat([_rnd(out, dt), t[rope_rows:]])
    return t, h


CASES = [
    ("wo", 4096, 4096, {}), ("wqkv_rope_norm", 6144, 4096, dict(norm=True, delta=True, rope=(5120, 128))),
    ("w13_norm", 14336, 4096, dict(norm=True, delta=True, swiglu=True)), ("w2", 4096, 14336, {}),
    ("ragged", 1030, 1000, dict(norm=True)), ("ragged_pair", 77, 264, dict(swiglu=True, norm=True, delta=True)),
    ("tiny_rope_bias", 96, 64, dict(norm=True, rope=(64, 16), bias=True)), ("tp8_wo", 4096, 512, {}),
    # r5: BASELINE C5 — the Llama-3-70B shape (dim 8192, ffn 28672, 64 / 8 heads) unsharded and as ONE rank of TP = 8 sees it
    ("70b_wqkv", 10240, 8192, dict(norm=True, delta=True, rope=(9216, 128))), ("70b_wo", 8192, 8192, {}),
    ("70b_w13", 28672, 8192, dict(norm=True, delta=True, swiglu=True)), ("70b_w2", 8192, 28672, {}),
    ("c5_rank_wqkv", 1280, 8192, dict(norm=True, delta=True, rope=(1152, 128))), ("c5_rank_wo", 8192, 1024, {}),
    ("c5_rank_w13", 3584, 8192, dict(norm=True, delta=True, swiglu=True)), ("c5_rank_w2", 8192, 3584, {}),
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("name,N,K,opt", CASES)
def test_gemv_fused_matches_fp32_reference(oracle, dt, name, N, K, opt):
    if dt == torch.float32 and K * 4 > 64 * 1024:
        pytest.skip("input vector beyond the LDS staging buffer for fp32 (the harness falls back to the library GEMV)")
    _gemv_case(oracle, dt, name, N, K, opt, N * 7 + K)


def test_gemv_fused_fuzz(oracle):
    """40 seeded random shapes and option sets: 1 .. 3000 rows (odd counts included), K a multiple of the 16-byte vector
    up to the 64 KiB input limit, any valid combination of norm / delta / SwiGLU pair / RoPE rows / bias."""
    import random

    rng = random.Random(23)
    for i in range(40):
        dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        vec = 4 if dt == torch.float32 else 8
        K = vec * rng.choice([1, 2, 7, 33, 64, 129, 512, rng.randint(1, 1000)])
        if K * (4 if dt == torch.float32 else 2) > 64 * 1024:
            K = vec * 512
        N = rng.choice([1, 2, 3, rng.randint(4, 300), rng.randint(301, 3000)])
        opt = {}
        if rng.random() < 0.6:
            opt["norm"] = True
            opt["delta"] = rng.random() < 0.6
        mode = rng.randrange(3)
        if mode == 1:
            opt["swiglu"] = True
        elif mode == 2 and N >= 2:
            hd = rng.choice([2, 4, 16, 64])
            rows = (N // hd) * hd if rng.random() < 0.5 else ((N // hd) // 2) * hd
            if rows > 0:
                opt["rope"] = (rows, hd)
            opt["bias"] = rng.random() < 0.5
        elif mode == 0:
            opt["bias"] = rng.random() < 0.3
        _gemv_case(oracle, dt, f"case {i}: {dt} N={N} K={K} {opt}", N, K, opt, 8000 + i)


def _gemv_case(oracle, dt, name, N, K, opt, seed):
    from cold_compress_amd.harness import glue

    g = torch.Generator().manual_seed(seed)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dt)
    W3 = (torch.randn(N, K, generator=g) * 0.05).to(dt) if opt.get("swiglu") else None
    x = torch.randn(K, generator=g).to(dt)
    delta = (torch.randn(K, generator=g) * 0.5).to(dt) if opt.get("delta") else None
    nw = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dt) if opt.get("norm") else None
    bias = (torch.randn(N, generator=g) * 0.1).to(dt) if opt.get("bias") else None
    freqs, rope_rows, hd = None, 0, 0
    if opt.get("rope"):
        rope_rows, hd = opt["rope"]
        ang = torch.rand(hd // 2, generator=g) * 6.28
        freqs = torch.stack([torch.cos(ang), torch.sin(ang)], -1).to(dt)
    ref, h_ref = _ref(W, x, dt, W3, delta, nw, 1e-5, bias, freqs, rope_rows, hd)
    d = lambda t: t.to(DEV) if t is not None else None  # noqa: E731
    h_out = torch.empty(K, dtype=dt, device=DEV) if nw is not None else None
    y = glue.gemv_fused(d(W), d(x), w3=d(W3), delta=d(delta), norm_weight=d(nw), eps=1e-5, h_out=h_out, bias=d(bias), freqs=d(freqs),
                        rope_rows=rope_rows, head_dim=hd)
    torch.cuda.synchronize()
    tol = 2 * ULP[dt] * max(1.0, float(ref.abs().max()))
    assert (y.cpu().float() - ref).abs().max() <= tol, name
    if h_out is not None:
        assert torch.equal(h_out.cpu().float(), h_ref), "updated residual stream"
    # oracle twin (double-precision accumulation): same tolerance
    es = np.float32 if dt == torch.float32 else np.uint16
    yo = np.zeros(N, es)
    ho = np.zeros(K, es) if nw is not None else None
    o = oracle
    pp = lambda t: o.ptr(to_np(t)) if t is not None else None  # noqa: E731
    o.call("cc_gemv_fused", pp(W), pp(W3), pp(x), pp(delta), pp(nw), 1e-5, o.ptr(ho) if ho is not None else None, pp(bias), pp(freqs),
           rope_rows, hd, o.ptr(yo), N, K, DT_CODE[dt], None)
    yo_t = torch.from_numpy(yo.view(np.int16).copy()).view(dt).float() if dt != torch.float32 else torch.from_numpy(yo)
    assert (y.cpu().float() - yo_t).abs().max() <= tol, name
