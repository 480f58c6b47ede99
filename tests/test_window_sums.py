"""Window sums of the attention-history ring (history_window_size W > 1; hybrid W = 400): dtype(sum_W num) is defined
as the EXACT sum rounded once to the model dtype (include/coldcompress.h).  CPU: the oracle's restatement against
exact rational arithmetic, including wide exponent spreads, rounding ties and non-finite entries.  GPU: the stateless
pre-pass and the incrementally TRACKED state against the oracle and against each other, bit for bit, through
overwrites, evictions and pathological values."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest
import torch

from helpers import DT_CODE, to_np

FMT = {torch.float32: (24, -149), torch.bfloat16: (8, -133), torch.float16: (11, -24)}  # mantissa bits, log2(quantum_min)


def exact_round(x, dtype):
    """Fraction -> nearest-even value of dtype, as a python float (no overflow cases here)."""
    if x == 0:
        return 0.0
    sgn = -1 if x < 0 else 1
    x = abs(x)
    p, qmin = FMT[dtype]
    e = x.numerator.bit_length() - x.denominator.bit_length()  # 2^(e-1) <= x < 2^(e+1)
    if Fraction(2) ** e > x:
        e -= 1
    q = Fraction(2) ** max(e - (p - 1), qmin)
    n = x / q
    r = n.numerator // n.denominator
    frac = n - r
    if frac > Fraction(1, 2) or (frac == Fraction(1, 2) and (r & 1)):
        r += 1
    return sgn * float(r * q)


def rows_for(dtype, W, gen):
    """A list of W-entry rows (python floats, exactly representable in dtype) exercising the definition."""
    def cast(vals):
        return torch.as_tensor(vals, dtype=torch.float64).to(dtype)

    rows = [cast(torch.rand(W, generator=gen, dtype=torch.float64) * 0.01) for _ in range(6)]
    rows.append(cast(torch.rand(W, generator=gen, dtype=torch.float64) * 2.0 - 1.0))  # signs
    rows.append(cast([0.0] * W))
    z = [0.0] * W
    z[0], z[1] = 1.0, 2.0 ** -8  # exact tie at bf16 precision (1 + 2^-8): even -> 1.0
    rows.append(cast(z))
    z2 = list(z)
    z2[2] = 2.0 ** -120 if dtype != torch.float16 else 2.0 ** -24  # sticky bit far below: the tie breaks upward
    rows.append(cast(z2))
    z3 = [0.0] * W
    z3[0], z3[1], z3[2] = 1.0, 2.0 ** -126 if dtype != torch.float16 else 2.0 ** -14, 3.0  # wide spread
    rows.append(cast(z3))
    z4 = [2.0 ** -133 if dtype != torch.float16 else 2.0 ** -24] * W  # subnormals only
    rows.append(cast(z4))
    z5 = [0.0] * W
    z5[0], z5[1] = 0.5, -0.5  # cancellation to exactly zero
    z5[2] = 2.0 ** -100 if dtype != torch.float16 else 2.0 ** -20
    rows.append(cast(z5))
    return torch.stack(rows)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_oracle_window_sums_are_exact(oracle, dtype):
    o, code, W = oracle, DT_CODE[dtype], 37
    gen = torch.Generator().manual_seed(5)
    ring = rows_for(dtype, W, gen)  # [n, W]
    n = ring.shape[0]
    num = to_np(ring.reshape(1, n, W))
    wsum = np.zeros(n, np.float32)
    acc = np.zeros(o.fns()["cc_hh_ring_acc_words"](1, n, W, code), np.uint64)
    o.call("cc_hh_ring_window_sums", o.ptr(num), 1, n, W, code, o.ptr(wsum), o.ptr(acc), None)
    for i in range(n):
        exact = sum((Fraction(float(v)) for v in ring[i].double().tolist()), Fraction(0))
        assert float(wsum[i]) == exact_round(exact, dtype), f"row {i}: {wsum[i]} vs exact {float(exact)}"
        # the accumulator words are the exact sum in units of 2^-149, two's complement over 192 bits
        val = int(acc[4 * i]) | (int(acc[4 * i + 1]) << 64) | (int(acc[4 * i + 2]) << 128)
        if val >> 191:
            val -= 1 << 192
        assert Fraction(val, 2 ** 149) == exact and acc[4 * i + 3] == 0
    # entries that do not fit (|v| >= 4, inf, nan): counted, the row sums to NaN
    bad = torch.zeros(3, W, dtype=torch.float64)
    bad[0, 3], bad[1, 5], bad[2, 7] = 4.0, float("inf"), float("nan")
    num = to_np(bad.to(dtype).reshape(1, 3, W))
    wsum, acc = np.zeros(3, np.float32), np.zeros(o.fns()["cc_hh_ring_acc_words"](1, 3, W, code), np.uint64)
    o.call("cc_hh_ring_window_sums", o.ptr(num), 1, 3, W, code, o.ptr(wsum), o.ptr(acc), None)
    assert np.isnan(wsum).all() and (acc[3::4][:3] == 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,H,S,W", [(torch.bfloat16, 3, 200, 400), (torch.float32, 2, 65, 7), (torch.float16, 1, 130, 33)])
def test_gpu_window_sums_equal_oracle(oracle, dtype, H, S, W):
    from cold_compress_amd import _abi

    code = DT_CODE[dtype]
    gen = torch.Generator().manual_seed(9)
    ring = (torch.rand(H, S, W, generator=gen, dtype=torch.float64) ** 8).to(dtype)  # wide dynamic range
    special = rows_for(dtype, W, gen)
    ring[0, :special.shape[0]] = special
    ring[-1, -1, 0] = float("inf")
    ring[-1, -2, W - 1] = 5.0
    num = to_np(ring)
    words = int(_abi.lib()["cc_hh_ring_acc_words"](H, S, W, code))
    assert words == oracle.fns()["cc_hh_ring_acc_words"](H, S, W, code)
    ws_o, acc_o = np.zeros(H * S, np.float32), np.zeros(words, np.uint64)
    oracle.call("cc_hh_ring_window_sums", oracle.ptr(num), H, S, W, code, oracle.ptr(ws_o), oracle.ptr(acc_o), None)
    d = ring.cuda()
    ws = torch.full((H * S,), -1.0, device="cuda")
    acc = torch.full((words,), -1, dtype=torch.int64, device="cuda")
    if (H * S * W * ring.element_size()) % 8:
        acc[H * S * 4 + 2 + (H * S * W * ring.element_size()) // 8] = 0  # tail bytes of the last shadow word: not state
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _abi.call("cc_hh_ring_window_sums", p(d), H, S, W, code, p(ws), p(acc), None)
    torch.cuda.synchronize()
    assert np.array_equal(ws.cpu().numpy().view(np.uint32), ws_o.view(np.uint32))  # NaN rows included, bit for bit
    assert np.array_equal(acc.cpu().numpy().view(np.uint64), acc_o)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_tracked_window_sums_stay_exact(dtype):
    """KVCacheHeavyHitter with a 5-entry ring: after every update_state (one ring column overwritten, tracked
    incrementally) and every eviction (row zeroed) the tracked sums / accumulators equal a rebuild from the ring."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd import _abi

    H, S, D, W, T = 2, 96, 16, 5, 90
    cls, rk = cache.get_cache_constructor("heavy_hitter")
    kw = dict(max_cache_length=S, global_tokens=2, recent_window=3, history_window_size=W, attn_thresholding=False,
              max_seq_length=4 * S, cache_bits=None)
    with torch.device("cuda"):
        kv = cls(1, H, D, dtype, **{k: kw[k] for k in rk})
    gen = torch.Generator().manual_seed(2)
    kv.update_kv(torch.arange(T, device="cuda"), torch.randn(1, H, T, D, generator=gen).to(dtype).cuda(),
                 torch.randn(1, H, T, D, generator=gen).to(dtype).cuda(), True)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    code = DT_CODE[dtype]

    def check(tag):
        ws = torch.empty(H * S, device="cuda")
        acc = torch.zeros_like(kv.attn_window_acc)
        _abi.call("cc_hh_ring_window_sums", p(kv.attn_history_num), H, S, W, code, p(ws), p(acc), None)
        torch.cuda.synchronize()
        assert torch.equal(ws.view(torch.int32), kv.attn_window_sum.reshape(-1).view(torch.int32)), tag
        assert torch.equal(acc, kv.attn_window_acc), tag  # accumulators, ticket back at zero, shadow == ring transposed

    tiny = 2.0 ** -120
    for t in range(4 * W):
        pos = torch.tensor([T + t], dtype=torch.int32, device="cuda")
        kv.update_kv(pos, torch.randn(1, H, 1, D, generator=gen).to(dtype).cuda(), torch.randn(1, H, 1, D, generator=gen).to(dtype).cuda(), False)
        check(f"after eviction {t}")
        a = torch.rand(1, H, 1, S, generator=gen) ** 6
        a[0, 0, 0, 10] = 1.0 if t == 2 else tiny  # a large entry that later leaves a window of tiny ones
        a[0, 1, 0, 20] = float("nan") if t == 3 else 0.25  # a NaN enters, then leaves W steps later
        a[0, 1, 0, 21] = 6.0 if t == 4 else 0.125
        kv.update_state(pos, None, None, False, a.to(dtype).cuda())
        check(f"after update {t}")
        assert int(kv.attn_counter) == t + 1
    ws = kv.attn_window_sum.cpu()
    assert torch.isfinite(ws).all()  # the NaN / out-of-range entries have left the window again
