"""The fused heavy-hitter decode step (cc_decode_step_heavy_hitter: insert folded into the K/V streaming pass,
history update + next-position arg-min folded into the combine pass) must be bit-identical to the three-call
sequence update_kv -> attention -> update_state, step after step, at the BASELINE shape — and to the oracle's
pipeline twin on a small shape."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from helpers import to_np

pytestmark = pytest.mark.gpu
DEV = __import__("helpers").TEST_DEVICE  # "cuda"; "cpu" only under tests/cpu_twin.py


@pytest.fixture()
def single_launch_switch():
    """Process-wide single-launch switch of the fused decode steps (include/coldcompress.h); restored to its default."""
    from cold_compress_amd import _abi

    fn = _abi.lib()["cc_decode_step_set_single_launch"]
    yield lambda on: fn(1 if on else 0)
    fn(1)


def _y_check(ya, yb, single, t):
    """Two-launch fused step: y bit-identical to the three-call attention; single launch: its partial sums are folded in a
    different fixed order, one rounding of the model dtype apart at most."""
    if single:
        assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), f"step {t}: attention output"
    else:
        assert torch.equal(ya, yb), f"step {t}: attention output"


def _mk(H, S, D, dtype, g=4, w=10):
    import cold_compress_amd.cache as cache

    with torch.device(DEV):
        return cache.KVCacheHeavyHitter(1, H, D, dtype, max_cache_length=S, max_seq_length=4 * S, cache_bits=None, global_tokens=g,
                                        history_window_size=1, recent_window=w, attn_thresholding=False)


def _seed(kv, gen, T):
    H, S, D = kv.n_heads, kv.max_cache_length, kv.head_dim
    dt = kv.k_cache.dtype
    kv.update_kv(torch.arange(T, device=DEV), torch.randn(1, H, T, D, generator=gen).to(dt).to(DEV),
                 torch.randn(1, H, T, D, generator=gen).to(dt).to(DEV), True)
    kv.attn_history_num[0, :, :T, 0] = torch.rand(H, T, generator=gen, dtype=torch.float64).to(DEV)
    kv.attn_history_denom[0, :, :T] = torch.randint(1, 5, (H, T), generator=gen, dtype=torch.int32).to(DEV)
    kv.attn_history_num[0, :, 50:60, 0] = 0.0  # engineered ties


@pytest.mark.parametrize("dtype,H,HQ,S,D,T", [(torch.bfloat16, 8, 32, 4096, 128, 4090), (torch.float32, 2, 4, 333, 16, 300),
                                              (torch.bfloat16, 1, 8, 3488, 128, 3488), (torch.float16, 4, 8, 1000, 64, 1000),
                                              (torch.bfloat16, 3, 12, 1001, 128, 990), (torch.float16, 2, 2, 67, 128, 60),
                                              (torch.bfloat16, 5, 10, 8200, 128, 8200),
                                              # caches beyond 64 x 64 slots per head: several tiles per wave in the single launch
                                              (torch.bfloat16, 8, 32, 8192, 128, 8190), (torch.float16, 4, 16, 5000, 128, 4990),
                                              (torch.bfloat16, 2, 16, 18432, 128, 18400), (torch.bfloat16, 1, 8, 32768, 128, 32768),
                                              # the 8-wave geometry in its other instantiations: fp16, and 8 query heads per kv head
                                              (torch.float16, 8, 32, 4096, 128, 4090), (torch.bfloat16, 8, 64, 4096, 128, 4000),
                                              (torch.float16, 8, 64, 2560, 128, 2560)])
@pytest.mark.parametrize("single", [False, True])
def test_fused_step_equals_three_calls(dtype, H, HQ, S, D, T, single):
    """`single`: decode_step may run as ONE launch where the shape allows it (include/coldcompress.h); every buffer must
    still equal the three-call sequence bit for bit — only y, whose partial sums are folded in a different fixed order
    there, is held to one rounding of the model dtype."""
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    a, b = _mk(H, S, D, dtype), _mk(H, S, D, dtype)
    b.single_launch = single
    one = b.single_launch_active(HQ)
    if single and not one:
        pytest.skip("shape not eligible for the single-launch step: covered by single=False")
    for kv in (a, b):
        _seed(kv, torch.Generator().manual_seed(17), T)
    gen = torch.Generator().manual_seed(5)
    for t in range(14):
        p = torch.tensor([T + 7 + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        ya, attn = sdpa(q, ka, va, attn_mask=ma, return_attn=True, group_mean=True)
        a.update_state(p, k1, v1, False, attn)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        if one:
            assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), f"step {t}: attention output"
        else:
            assert torch.equal(ya, yb), f"step {t}: attention output"
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), f"step {t}: {na}"
    assert b._next_valid and not a._next_valid
    assert b.step_status(HQ) == 0


@pytest.mark.parametrize("dtype,H,HQ,S,D,T,steps", [(torch.bfloat16, 8, 32, 4096, 128, 4000, 400), (torch.bfloat16, 8, 32, 2560, 128, 2560, 150),
                                                    (torch.float16, 3, 6, 1001, 128, 900, 150), (torch.bfloat16, 2, 2, 130, 128, 100, 60),
                                                    (torch.bfloat16, 16, 64, 2048, 128, 2048, 100), (torch.bfloat16, 8, 32, 18432, 128, 18432, 60),
                                                    (torch.bfloat16, 8, 32, 40000, 128, 39990, 30)])  # (r4: 10 tiles per wave: the 16-tile instantiation)
def test_single_launch_equals_two_launch_long(dtype, H, HQ, S, D, T, steps):
    """The single-launch layer step against the two-launch step over hundreds of steps on twin caches, interleaved with an
    unrelated bandwidth-heavy kernel so the workgroups of a launch do not arrive evenly: history (float64), denominators,
    positions, masks, counts, K/V and counters stay bit-identical at every checkpoint (a stale or torn hand-off would show
    up as a different probability, hence a different history), y within one rounding, and the timeout word stays 0."""
    a, b = _mk(H, S, D, dtype), _mk(H, S, D, dtype)
    a.single_launch, b.single_launch = False, True
    if not b.single_launch_active(HQ):
        pytest.skip("shape not eligible for the single-launch step on this device")
    for kv in (a, b):
        _seed(kv, torch.Generator().manual_seed(23), T)
    gen = torch.Generator().manual_seed(9)
    noise = torch.randn(64 << 20, device=DEV)
    side = torch.cuda.Stream()
    for t in range(steps):
        p = torch.tensor([T + 3 + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = (3.0 * torch.randn(1, HQ, 1, D, generator=gen)).to(dtype).to(DEV)
        ya = a.decode_step(q, k1, v1, p)
        if t % 3 == 0:  # concurrent traffic on another stream: uneven arrival of the step's workgroups
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        yb = b.decode_step(q, k1, v1, p)
        if t % 25 == 24 or t == steps - 1:
            torch.cuda.synchronize()
            assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), f"step {t}: y"
            for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
                if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                    assert torch.equal(ta, tb), f"step {t}: {na}"
    torch.cuda.synchronize()
    assert b.step_status(HQ) == 0


def _raw_hh_step(kv, q, k1, v1, p, HQ, phases, attn_out):
    """cc_decode_step_heavy_hitter_phases straight through the C ABI (the class never asks for attn_out)."""
    import math
    from cold_compress_amd import _abi
    from cold_compress_amd.attention_utils import _workspace
    from cold_compress_amd.cache import _DT, _ptr, _stream

    D = kv.head_dim
    y = torch.empty((1, HQ, 1, D), dtype=q.dtype, device=q.device)
    nbytes = _abi.lib()["cc_decode_attn_workspace_bytes"](HQ, kv.n_heads, kv.max_cache_length, D, _DT[kv.k_cache.dtype])
    ws = _workspace(nbytes, q.device)
    k, v = kv._new_rows(k1, v1)
    rc = _abi.lib()["cc_decode_step_heavy_hitter_phases"](
        kv._view(), _ptr(q.reshape(HQ, D).contiguous()), _ptr(k), _ptr(v), _ptr(p), _ptr(kv.attn_history_num), _ptr(kv.attn_history_denom),
        _ptr(kv.attn_counter), _ptr(kv.next_key), int(kv.global_tokens), int(kv.recent_window), HQ, 1.0 / math.sqrt(D), _ptr(y),
        _ptr(attn_out), _ptr(ws), ws.numel(), _stream(), phases)
    return rc, y


@pytest.mark.parametrize("dtype,full_exists", [(torch.bfloat16, True), (torch.float16, False)])
def test_attn_out_is_served_by_the_full_single_launch_kernel_or_by_two_launches(dtype, full_exists):
    """The kernels the single launch normally runs have no attn_out store (include/coldcompress.h): a call that passes attn_out
    gets a full-featured instantiation (bf16, four query heads per kv head) or the two-launch step; demanding ONE launch where
    there is no such instantiation is CC_ERR_UNSUPPORTED.  attn_out and every buffer equal the two-launch step's bit for bit."""
    from cold_compress_amd import _abi

    H, HQ, S, D, T = 8, 32, 4096, 128, 4090
    a, b = _mk(H, S, D, dtype), _mk(H, S, D, dtype)
    for kv in (a, b):
        _seed(kv, torch.Generator().manual_seed(31), T)
    if not a.single_launch_active(HQ):
        pytest.skip("shape not eligible for the single-launch step on this device")
    gen = torch.Generator().manual_seed(6)
    oa = torch.zeros(H, S, dtype=dtype, device=DEV)
    ob = torch.zeros(H, S, dtype=dtype, device=DEV)
    for t in range(6):
        p = torch.tensor([T + 1 + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        for kv in (a, b):
            if not kv._next_valid:
                kv.prepare_decode(p)
        rc, ya = _raw_hh_step(a, q, k1, v1, p, HQ, 3 | _abi.CC_PHASE_TWO_LAUNCH, oa)
        assert rc == 0
        rc1, yb = _raw_hh_step(b, q, k1, v1, p, HQ, 3 | _abi.CC_PHASE_ONE_LAUNCH, ob)
        if full_exists:
            assert rc1 == 0
        else:
            assert rc1 != 0  # no full-featured f16 instantiation: the demand cannot be met ...
            rc2, yb = _raw_hh_step(b, q, k1, v1, p, HQ, 3, ob)  # ... the library's own choice (two launches) serves the call
            assert rc2 == 0
        torch.cuda.synchronize()
        assert torch.equal(oa, ob) and float(oa.float().abs().sum()) > 0, f"step {t}: attn_out"
        assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), f"step {t}: y"
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), f"step {t}: {na}"
    assert b.step_status(HQ) == 0


def test_single_launch_in_hipgraph():
    """Replayed from a hipGraph (the way the harness decodes), with several layers sharing one workspace: the epoch words
    advance on the device, so replays need no reset node; results equal the two-launch twins bit for bit."""
    H, HQ, S, D, dtype, L = 8, 32, 4096, 128, torch.bfloat16, 4
    A, B = [_mk(H, S, D, dtype) for _ in range(L)], [_mk(H, S, D, dtype) for _ in range(L)]
    if not B[0].single_launch_active(HQ):
        pytest.skip("shape not eligible for the single-launch step on this device")
    for l in range(L):
        A[l].single_launch = False
        for kv in (A[l], B[l]):
            _seed(kv, torch.Generator().manual_seed(31 + l), S - 5)
    gen = torch.Generator().manual_seed(2)
    q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
    k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
    v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
    pos = torch.tensor([S + 11], dtype=torch.int32, device=DEV)

    def token(caches):
        for kv in caches:
            kv.decode_step(q, k1, v1, pos)

    for caches in (A, B):  # eager warm-up (allocates the workspace, seeds the pipelines), then capture
        token(caches)
    pos += 1
    graphs = []
    for caches in (A, B):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            pass
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        snap = [{k: v.clone() for k, v in c._buffers.items()} for c in caches]
        with torch.cuda.graph(g):
            token(caches)
        for c, sn in zip(caches, snap):  # capture does not execute: nothing to restore, but keep the twins aligned explicitly
            for k, v in sn.items():
                c._buffers[k].copy_(v)
        graphs.append(g)
    for t in range(40):
        for g in graphs:
            g.replay()
        pos += 1
    torch.cuda.synchronize()
    for l in range(L):
        for (na, ta), (nb, tb) in zip(A[l].named_buffers(), B[l].named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), f"layer {l}: {na}"
    assert B[0].step_status(HQ) == 0


@pytest.mark.parametrize("H,HQ,S", [(4, 16, 512), (8, 32, 4096), (1, 8, 3488)])
def test_fused_step_vs_oracle_pipeline(oracle, H, HQ, S):
    """decode_step (ONE launch at these shapes — the headline one on the wide geometry) against the oracle's fused step, each on
    its own state: identical slots / counts / denominators / K, y within the contract's 1e-3 + two roundings of the output."""
    D, g, w = 128, 4, 10
    kv = _mk(H, S, D, torch.bfloat16, g, w)
    assert kv.single_launch_active(HQ)
    _seed(kv, torch.Generator().manual_seed(3), S - 3)
    st = dict(k=to_np(kv.k_cache.cpu()[0]), v=to_np(kv.v_cache.cpu()[0]), pos=kv.pos.cpu()[0].numpy().copy(),
              mask=kv.mask.cpu()[0, :, 0].numpy().astype(np.uint8), cts=kv.cache_cts.cpu().numpy().copy(),
              num=kv.attn_history_num.cpu()[0, :, :, 0].numpy().copy(), denom=kv.attn_history_denom.cpu()[0].numpy().copy(),
              ctr=np.zeros(1, np.int64), key=np.zeros((H, (S + 127) // 128), np.uint64))
    gen = torch.Generator().manual_seed(8)
    o = oracle
    p0 = S + 20
    view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], 1)
    o.call("cc_hh_next_key_init", C.byref(view), o.ptr(np.array([p0], np.int32)), o.ptr(st["num"]), o.ptr(st["denom"]), g, w,
           o.ptr(st["key"]), None)
    for t in range(6):
        p = torch.tensor([p0 + t], dtype=torch.int32)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(torch.bfloat16)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(torch.bfloat16)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(torch.bfloat16)
        y = kv.decode_step(q.to(DEV), k1.to(DEV), v1.to(DEV), p.to(DEV))
        torch.cuda.synchronize()
        view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], 1)
        yo = np.zeros((HQ, D), np.uint16)
        o.call("cc_decode_step_heavy_hitter", C.byref(view), o.ptr(to_np(q.reshape(HQ, D))), o.ptr(to_np(k1.reshape(H, D))),
               o.ptr(to_np(v1.reshape(H, D))), o.ptr(p.numpy().copy()), o.ptr(st["num"]), o.ptr(st["denom"]), o.ptr(st["ctr"]),
               o.ptr(st["key"]), g, w, HQ, 1.0 / math.sqrt(D), o.ptr(yo), None, None, 0, None)
        # slots chosen so far are identical (pos is written by the insert); outputs within bf16 tolerance
        assert np.array_equal(kv.pos.cpu()[0].numpy(), st["pos"]), f"step {t}"
        assert np.array_equal(kv.cache_cts.cpu().numpy(), st["cts"])
        yr = torch.from_numpy(yo.view(np.int16).copy()).view(torch.bfloat16).float()
        assert float((y.cpu().float()[0, :, 0] - yr).abs().max()) <= 1e-3 + 2 * 2.0 ** -8 * float(yr.abs().max()), f"step {t}: y"
    assert np.array_equal(kv.attn_history_denom.cpu()[0].numpy(), st["denom"])
    assert np.allclose(kv.attn_history_num.cpu()[0, :, :, 0].numpy(), st["num"], rtol=2 * 2.0 ** -8, atol=6 * 2.0 ** -16)
    assert kv.step_status(HQ) == 0
    assert np.array_equal(to_np(kv.k_cache.cpu()[0]), st["k"])


@pytest.mark.parametrize("strategy,dtype,H,HQ,S,D,T", [("recent_global", torch.bfloat16, 8, 32, 4096, 128, 4090), ("recent_global", torch.float32, 2, 4, 77, 16, 77),
                                                       ("full", torch.bfloat16, 4, 16, 600, 128, 500), ("recent_global", torch.float16, 1, 8, 3488, 128, 3488),
                                                       ("recent_global", torch.bfloat16, 8, 32, 8192, 128, 8100), ("full", torch.bfloat16, 2, 8, 20000, 128, 19990)])
@pytest.mark.parametrize("single", [False, True])
def test_ring_fused_step_equals_three_calls(strategy, dtype, H, HQ, S, D, T, single, single_launch_switch):
    """Head-constant ring policies (recent_global, full): the fused step (cc_decode_step_recent_global; `single`: as ONE
    launch where the shape allows it) against update_kv -> attention, every buffer bit for bit, appends (empty slots) and
    evictions (ring) both covered; y exact for the two-launch form, within one rounding for the single launch."""
    import cold_compress_amd.cache as cache

    single_launch_switch(single)
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    cls, rk = cache.get_cache_constructor(strategy)
    kw = dict(max_cache_length=S, global_tokens=4, max_seq_length=4 * S, cache_bits=None)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    gen = torch.Generator().manual_seed(11)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
    for t in range(12):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        ya, _ = sdpa(q, ka, va, attn_mask=ma)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        _y_check(ya, yb, single, t)
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), f"step {t}: {na}"


def test_head_constant_key_rows_on_a_small_grid():
    """Regression (late r3, found by tools/fuzz_step.py): the single-launch step of a head-constant policy on a 16-workgroup grid
    (8 kv heads x 2 splits), where whole XCDs wake up late.  With ONE key row shared by all heads and rewritten by kv head 0's
    waves, 3.6 % of such steps put two heads' rows into the NEXT position's slot (their workgroups read the row after it had been
    rewritten).  Every kv head keeps its own copy now; 250 fresh caches x 6 steps, K / V compared exactly."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    dtype, H, R, S, D, g = torch.float16, 8, 2, 101, 128, 1
    cls, rk = cache.get_cache_constructor("recent_global")
    kw = dict(max_cache_length=S, global_tokens=g, max_seq_length=4 * S + 64, cache_bits=None)
    for it in range(250):
        with torch.device(DEV):
            a, b = cls(1, H, D, dtype, **{k: kw[k] for k in rk}), cls(1, H, D, dtype, **{k: kw[k] for k in rk})
        assert tuple(b.next_key.shape)[0] == H
        gen = torch.Generator().manual_seed(4000 + it)
        k0 = torch.randn(1, H, S, D, generator=gen).to(dtype).to(DEV)
        v0 = torch.randn(1, H, S, D, generator=gen).to(dtype).to(DEV)
        for kv in (a, b):
            kv.update_kv(torch.arange(S, device=DEV), k0, v0, True)
        for t in range(6):
            p = torch.tensor([S + t], dtype=torch.int32, device=DEV)
            k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
            v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
            q = torch.randn(1, H * R, 1, D, generator=gen).to(dtype).to(DEV)
            ka, va, ma = a.update_kv(p, k1, v1, False)
            sdpa(q, ka, va, attn_mask=ma)
            b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        assert torch.equal(a.pos, b.pos), f"cache {it}: positions"
        assert torch.equal(a.k_cache, b.k_cache) and torch.equal(a.v_cache, b.v_cache), f"cache {it}: a kv head's row went to another slot"
        rows = b.next_key.cpu().numpy().view(np.uint64).min(axis=1)
        assert (rows == rows[0]).all(), f"cache {it}: the kv heads' key rows differ"


@pytest.mark.parametrize("single", [False, True])
@pytest.mark.parametrize("dtype,H,HQ,S,D,T,g,w", [(torch.bfloat16, 8, 32, 4096, 128, 4090, 4, 10), (torch.float32, 2, 4, 77, 16, 70, 2, 3),
                                                  (torch.float16, 4, 16, 600, 128, 600, 0, 1)])
def test_random_fused_step_equals_three_calls(dtype, H, HQ, S, D, T, g, w, single, single_launch_switch):
    """KVCacheRandom: the fused step (cc_decode_step_random, the draw for p + 1 scored in step p; `single`: as ONE launch
    where the shape allows it) against update_kv -> attention on the same sequence of uniform draws; every buffer bit for bit."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    single_launch_switch(single)

    cls, rk = cache.get_cache_constructor("random")
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, max_seq_length=4 * S, cache_bits=None)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    gen = torch.Generator().manual_seed(12)
    steps = 12
    draws = [torch.rand(S, generator=gen).to(DEV) for _ in range(steps + 1)]
    ia, ib = iter(draws), iter(draws)
    a._rand = lambda: next(ia)
    b._rand = lambda: next(ib)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
    for t in range(steps):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        ya, _ = sdpa(q, ka, va, attn_mask=ma)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        _y_check(ya, yb, single, t)
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), f"step {t}: {na}"


@pytest.mark.parametrize("single", [False, True])
@pytest.mark.parametrize("dtype,H,HQ,S,D,T,g,w", [(torch.bfloat16, 8, 32, 4096, 128, 4090, 4, 10), (torch.float32, 2, 4, 77, 16, 70, 2, 3)])
def test_random_in_kernel_draws(dtype, H, HQ, S, D, T, g, w, single, single_launch_switch):
    """KVCacheRandom without an injected vector: the fused step draws IN the kernels (cc_decode_step_random_rng).  Against the
    three-call path fed the oracle's restatement of the generator (oracle_lib.rng_vector, position by position): every buffer
    bit for bit, so the generator, its (seed, position, slot) indexing and the single / two-launch forms all agree."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa
    from oracle import oracle_lib

    single_launch_switch(single)
    cls, rk = cache.get_cache_constructor("random")
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, max_seq_length=4 * S, cache_bits=None)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    assert b._in_kernel_rng()
    gen = torch.Generator().manual_seed(13)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
    torch.manual_seed(99)
    b.prepare_decode(torch.tensor([T], dtype=torch.int32, device=DEV))
    seed = b._rng_seed
    assert 0 < seed < 2 ** 62
    steps = 12
    draws = iter([torch.from_numpy(oracle_lib.rng_vector(seed, T + t, S)).to(DEV) for t in range(steps)])
    a._rand = lambda: next(draws)
    assert not a._in_kernel_rng()
    evicted = 0
    for t in range(steps):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        before = a.pos.clone()
        ka, va, ma = a.update_kv(p, k1, v1, False)
        evicted += int((before != -1).logical_and(a.pos != before).any())
        ya, _ = sdpa(q, ka, va, attn_mask=ma)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        _y_check(ya, yb, single, t)
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):
                assert torch.equal(ta, tb), f"step {t}: {na}"
    assert evicted >= steps - (S - T) - 1  # the cache filled up: the draws decided real evictions


def test_random_fused_replay_vs_reference():
    """The reference's own random-policy trace (tests/golden/f4_random.npz: its draws, its evicted slots, its final
    buffers) replayed through the two-launch step."""
    import cold_compress_amd.cache as cache
    from helpers import DT_FROM_NAME, load_golden

    f = load_golden("f4_random.npz")
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    cls, rk = cache.get_cache_constructor("random")
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, max_seq_length=4 * S, cache_bits=None)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{k: kw[k] for k in rk})
    steps = f["steps"]
    draws = iter([f["rand_u"][t].to(DEV) for t in range(steps)] + [torch.zeros(S, device=DEV)])
    kv._rand = lambda: next(draws)
    kv.update_kv(torch.arange(T, device=DEV), f["k0"].to(DEV), f["v0"].to(DEV), True)
    gen = torch.Generator().manual_seed(3)
    for t in range(steps):
        before = kv.pos.clone()
        q = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        kv.decode_step(q, f["k_new"][t].to(DEV), f["v_new"][t].to(DEV), torch.tensor([T + t], dtype=torch.int32, device=DEV))
        changed = (kv.pos != before).reshape(-1).nonzero().reshape(-1).cpu().numpy()
        assert np.array_equal(changed, f["idx"][t].numpy().reshape(-1)), f"step {t}: evicted slot"
    torch.cuda.synchronize()
    assert torch.equal(kv.pos.cpu(), f["final_pos"])
    assert torch.equal(kv.mask.cpu(), f["final_mask"])
    assert torch.equal(kv.cache_cts.cpu(), f["final_cts"])
    assert torch.equal(kv.k_cache.cpu().view(torch.int16), f["final_k"].view(torch.int16))
    assert torch.equal(kv.v_cache.cpu().view(torch.int16), f["final_v"].view(torch.int16))


@pytest.mark.parametrize("strategy,dtype,H,HQ,S,D,T,W", [("heavy_hitter", torch.bfloat16, 8, 32, 1024, 128, 1000, 8),
                                                         ("heavy_hitter", torch.float32, 2, 4, 77, 16, 70, 3),
                                                         ("hybrid", torch.bfloat16, 5, 20, 640, 128, 600, 400)])
def test_ring_history_folded_into_combine(strategy, dtype, H, HQ, S, D, T, W):
    """history_window_size > 1 (and the hybrid cache's W = 400 ring): attention with the ring update folded into its
    combine pass (cc_decode_attn_gqa_ring) against attention -> update_state; every buffer, the tracked window sums
    included, bit for bit over appends and evictions."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    cls, rk = cache.get_cache_constructor(strategy)
    hyb = [{"strategy": "special"}, {"strategy": "special_punc"}, {"strategy": "special_punc_heavy_hitter", "heavy_hitter_frac": 0.3},
           {"strategy": "special_punc_window", "recent_window": 0.3}, {"strategy": "full"}]
    kw = dict(max_cache_length=S, global_tokens=4, recent_window=10, history_window_size=W, attn_thresholding=False,
              max_seq_length=4 * S, cache_bits=None, token_ids={"special": [[1], [2, 3]], "punctuation": [5, 6, 7]},
              min_recovery_frac=0.9, hybrid_strategies=hyb)

    def mk():
        with torch.device(DEV):
            kv = cls(1, H, D, dtype, **{k: kw[k] for k in rk})
        return kv

    a, b = mk(), mk()
    gen = torch.Generator().manual_seed(21)
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True, input_ids=torch.zeros(T, dtype=torch.int64, device=DEV))
        if strategy == "hybrid":  # a decode-ready state without the profiling pass: head h runs policy h % 5
            kv.cache_strategies = (torch.arange(H, device=DEV) % len(hyb)).to(torch.int64).contiguous()
            kv.requires_heavy_hitter = True
            kv.cache_cts.fill_(T)
            kv.mask[..., :T] = True
            kv.pos[0, :, :T] = torch.arange(T, device=DEV, dtype=kv.pos.dtype)
    for t in range(2 * min(W, 12) + 3):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        ids = torch.tensor([5 if t % 4 == 1 else 9], dtype=torch.int64, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False, input_ids=ids)
        ya, attn = sdpa(q, ka, va, attn_mask=ma, return_attn=True, group_mean=True)
        a.update_state(p, k1, v1, False, attn, input_ids=ids)
        kb, vb, mb = b.update_kv(p, k1, v1, False, input_ids=ids)
        hist = b.fused_history()
        assert hist is not None and len(hist) == 6
        yb, _ = sdpa(q, kb, vb, attn_mask=mb, group_mean=True, history=hist)
        b._state_fused = True
        b.update_state(p, k1, v1, False, None, input_ids=ids)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), f"step {t}: attention output"
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            assert torch.equal(ta, tb), f"step {t}: {na}"


@pytest.mark.parametrize("dtype,H,HQ,S,D,T,g,w", [(torch.bfloat16, 8, 32, 4096, 128, 4090, 4, 10), (torch.float16, 1, 8, 600, 128, 600, 2, 3),
                                                  (torch.bfloat16, 4, 16, 333, 128, 300, 0, 1), (torch.bfloat16, 8, 32, 18432, 128, 18432, 4, 10)])
@pytest.mark.parametrize("single", [False, True])
def test_l2_fused_step_equals_three_calls(dtype, H, HQ, S, D, T, g, w, single, single_launch_switch):
    """KVCacheL2: the fused step (cc_decode_step_l2: global norm maximum folded across the step boundary by the two launches,
    or — `single` — handed over inside ONE launch, every workgroup gathering every workgroup's norm maximum) against
    update_kv -> attention; every buffer (key_norm included) bit for bit.  The evicted slot is usually the one holding
    the global maximum, so the exclusion of its old norm from the running maximum is exercised on every step."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    single_launch_switch(single)

    cls, rk = cache.get_cache_constructor("l2")
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, max_seq_length=4 * S, cache_bits=None)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    assert b.supports_fused_step()
    gen = torch.Generator().manual_seed(13)
    k0 = (torch.randn(1, H, T, D, generator=gen) * torch.rand(1, H, T, 1, generator=gen)).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
        kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None)
    for t in range(12):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        scale_k = 3.0 if t == 5 else 1.0  # one very large key: it becomes the global maximum inside the recent window
        k1 = (torch.randn(1, H, 1, D, generator=gen) * scale_k).to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        ya, _ = sdpa(q, ka, va, attn_mask=ma)
        yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        _y_check(ya, yb, single, t)
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta, tb), f"step {t}: {na}"


def _check_l2_record(kv, p):
    """The l2 cache's carried norm record of position p (key row tail, entry [live + (p & 1)], bits 0-15; cc_common.h): the head's
    largest norm over the slots it keeps at p + 1 — all but the arg-min of the key row's live entries — as a model-dtype pattern
    (unsigned order; NaN on top)."""
    nk = kv.next_key.cpu().numpy().view(np.uint64)
    kn = kv.key_norm.cpu()[0].view(torch.int16).numpy().view(np.uint16).astype(np.uint32)  # [H, S] patterns
    live = nk.shape[1] - 8
    for h in range(nk.shape[0]):
        kmin = int(nk[h, :live].min())
        keep = np.ones(kn.shape[1], bool)
        if kmin != 0xFFFFFFFFFFFFFFFF:
            keep[(kmin & 0xFFFFFFFF) >> 1] = False
        want = int(kn[h][keep].max()) if keep.any() else 0
        got = int(nk[h, live + (p & 1)]) & 0xFFFF
        assert got == want, f"position {p}, kv head {h}: record {got:#06x}, state {want:#06x}"


@pytest.mark.parametrize("H,HQ,S,T", [(8, 32, 4096, 4093), (2, 8, 600, 600), (8, 32, 1024, 1000)])
def test_l2_carried_norm_record_across_step_forms(H, HQ, S, T, single_launch_switch):
    """r6: the single-launch l2 step takes its head's norm maximum from the RECORD the previous step left in the key
    row's tail (the head's two largest norms and a holder) instead of reducing the head's norms inside the launch.  Every writer of that
    record is exercised against the three-call path (update_kv -> attention), every buffer bit for bit: the pipeline's seed, the
    single-launch step, the two-launch step (a third small launch), in every alternation; with the cases the record exists for —
    the holder of the maximum evicted (most steps), the holder PROTECTED by the recent window (a huge key: the maximum stays, another
    slot goes), two slots holding the same maximum (T1 == T2: the same key inserted twice), the maximum in one head only, a NaN key
    (torch.max propagates it: every score NaN), and a re-seed in the middle (update_kv invalidates the pipeline)."""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.attention_utils import scaled_dot_product_attention as sdpa

    D, dtype, g, w = 128, torch.bfloat16, 4, 6
    cls, rk = cache.get_cache_constructor("l2")
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, max_seq_length=4 * S, cache_bits=None)

    def mk():
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    a, b = mk(), mk()
    gen = torch.Generator().manual_seed(29)
    k0 = (torch.randn(1, H, T, D, generator=gen) * torch.rand(1, H, T, 1, generator=gen)).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    for kv in (a, b):
        kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
        kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None)
    forms = [True, True, False, True, False, False, True, True, True, False, True, True]  # single launch?
    # (the record exists in CC_V_L2CARRY builds only: tools/r6_call21.sh runs this test on one)
    from cold_compress_amd import _abi

    l2_carry = bool(_abi.lib()["cc_decode_step_l2_carry"]())
    b.prepare_decode(torch.tensor([T], dtype=torch.int32, device=DEV))  # seeds the pipeline: the record of position T - 1
    torch.cuda.synchronize()
    if l2_carry:
        _check_l2_record(b, T - 1)
    n_steps = 40
    k_dup = None
    for t in range(n_steps):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen)
        if t == 4:
            k1 = k1 * 6.0           # the global maximum, protected by the recent window for the next w steps
        if t == 9:
            k1[:, 1:] *= 0.05       # ... a maximum in kv head 0 only
            k1[:, 0] *= 5.0
        if t == 14:
            k_dup = k1.clone() * 4.0
        if t in (14, 15):
            k1 = k_dup.clone()      # two slots of every head hold the same (largest) norm
        if t == 30:
            k1[0, H - 1, 0, 3] = float("nan")
        k1 = k1.to(dtype).to(DEV)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
        ka, va, ma = a.update_kv(p, k1, v1, False)
        ya, _ = sdpa(q, ka, va, attn_mask=ma)
        if t == 22:  # the three-call path on b too: its pipeline is invalid afterwards and must be seeded again (record included)
            kb, vb, mb = b.update_kv(p, k1, v1, False)
            yb, _ = sdpa(q, kb, vb, attn_mask=mb)
            single = False
        else:
            single = forms[t % len(forms)]
            single_launch_switch(single)
            yb = b.decode_step(q, k1, v1, p)
        torch.cuda.synchronize()
        if t < 30:  # (behind the NaN key the outputs of the head that holds it are NaN on both sides)
            _y_check(ya, yb, single, t)
        if t != 22 and l2_carry:  # the step left the head's record for position T + t: check it against the state
            _check_l2_record(b, T + t)
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                assert torch.equal(ta.view(torch.int16) if ta.dtype == dtype else ta, tb.view(torch.int16) if tb.dtype == dtype else tb), f"step {t} ({'one' if single else 'two'} launch): {na}"
    from cold_compress_amd.attention_utils import single_launch_status

    assert single_launch_status() == 0


def test_l2_fused_step_vs_oracle():
    """The same step through the C ABI against the oracle's twin: slots, norms, K/V and the attention output."""
    import ctypes as C
    from cold_compress_amd import _abi
    from oracle import oracle_lib as o

    o.build(); o.fns()
    H, HQ, S, D, T, g, w = 2, 8, 200, 128, 190, 2, 4
    dtype, code = torch.bfloat16, 1
    gen = torch.Generator().manual_seed(14)
    import cold_compress_amd.cache as cache
    cls, rk = cache.get_cache_constructor("l2")
    kw = dict(max_cache_length=S, global_tokens=g, recent_window=w, max_seq_length=4 * S, cache_bits=None)
    with torch.device(DEV):
        kv = cls(1, H, D, dtype, **{k: kw[k] for k in rk})
    k0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
    kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
    kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None)
    torch.cuda.synchronize()
    st = dict(k=to_np(kv.k_cache.cpu()[0]), v=to_np(kv.v_cache.cpu()[0]), pos=kv.pos.cpu()[0].numpy().astype(np.int32).copy(),
              mask=kv.mask.cpu().reshape(H, S).numpy().astype(np.uint8).copy(), cts=kv.cache_cts.cpu().numpy().astype(np.int32).copy(),
              kn=to_np(kv.key_norm.cpu()[0]))
    nk = int(_abi.lib()["cc_hh_next_key_slots"](S))
    nkey = np.zeros((H, nk), np.uint64)
    view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], code)
    p0 = np.array([T], np.int32)
    o.call("cc_l2_next_key_init", C.byref(view), o.ptr(p0), o.ptr(st["kn"]), g, w, o.ptr(nkey), None)
    nbytes = _abi.lib()["cc_decode_attn_workspace_bytes"](HQ, H, S, D, code)
    ws = np.zeros(nbytes, np.uint8)
    for t in range(16):
        p = torch.tensor([T + t], dtype=torch.int32, device=DEV)
        k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype)
        y = kv.decode_step(q.to(DEV), k1.to(DEV), v1.to(DEV), p)
        yo = np.zeros((HQ, D), np.uint16)
        view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], code)
        o.call("cc_decode_step_l2", C.byref(view), o.ptr(to_np(q.reshape(HQ, D))), o.ptr(to_np(k1.reshape(H, D))),
               o.ptr(to_np(v1.reshape(H, D))), o.ptr(np.array([T + t], np.int32)), o.ptr(st["kn"]), o.ptr(nkey), g, w, HQ,
               1.0 / math.sqrt(D), o.ptr(yo), o.ptr(ws), ws.size, None)
        torch.cuda.synchronize()
        assert np.array_equal(kv.pos.cpu()[0].numpy(), st["pos"]), f"step {t}: slots"
        assert np.array_equal(to_np(kv.key_norm.cpu()[0]), st["kn"]), f"step {t}: norms"
        from helpers import from_np
        yr = from_np(yo, dtype).float()  # the attention contract: 1e-3 + two roundings of the output dtype
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        assert (y.cpu().float().reshape(HQ, D) - yr).abs().max() <= 1e-3 + 2 * ulp * yr.abs().max(), f"step {t}: y"
    assert np.array_equal(to_np(kv.k_cache.cpu()[0]), st["k"]) and np.array_equal(kv.cache_cts.cpu().numpy(), st["cts"])


def test_single_launch_caches_of_different_head_counts_share_the_workspace(single_launch_switch):
    """Caches with different numbers of kv heads and different lengths take turns on ONE workspace (one set of epoch words):
    launches with fewer heads advance only the first heads' epochs, so tags of different heads drift apart.  The granule regions
    are per kv head at fixed strides — a location only ever sees its own head's growing tags — and the l2 step compares every
    gathered norm granule with ITS head's tag: nothing stale may match, no hand-off may time out, and every single-launch step
    stays bit-identical to its two-launch twin."""
    import cold_compress_amd.cache as cache

    D, dtype = 128, torch.bfloat16

    def mk(strategy, H, S):
        cls, rk = cache.get_cache_constructor(strategy)
        kw = dict(max_cache_length=S, global_tokens=4, max_seq_length=4 * S, cache_bits=None, recent_window=10, history_window_size=1,
                  attn_thresholding=False)
        with torch.device(DEV):
            return cls(1, H, D, dtype, **{k: kw[k] for k in rk})

    shapes = [("heavy_hitter", 1, 8, 3488), ("heavy_hitter", 8, 32, 4096), ("l2", 8, 32, 1024), ("l2", 2, 8, 600), ("heavy_hitter", 4, 16, 333),
              ("l2", 8, 32, 4096), ("recent_global", 1, 4, 2048)]
    gen = torch.Generator().manual_seed(77)
    pairs = []
    for strategy, H, HQ, S in shapes:
        a, b = mk(strategy, H, S), mk(strategy, H, S)
        T = S - 5
        k0 = (torch.randn(1, H, T, D, generator=gen) * torch.rand(1, H, T, 1, generator=gen)).to(dtype).to(DEV)
        v0 = torch.randn(1, H, T, D, generator=gen).to(dtype).to(DEV)
        for kv in (a, b):
            kv.update_kv(torch.arange(T, device=DEV), k0, v0, True)
            kv.update_state(torch.arange(T, device=DEV), k0, v0, True, None) if strategy == "l2" else None
        pairs.append((strategy, H, HQ, S, T, a, b))
    for t in range(9):  # round-robin over the shapes: uneven epoch advance (the one-head caches step three times as often)
        for strategy, H, HQ, S, T, a, b in pairs:
            for rep in range(3 if H == 1 else 1):
                p = torch.tensor([T + 3 * t + rep], dtype=torch.int32, device=DEV)
                k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
                v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
                q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
                single_launch_switch(False)
                if hasattr(a, "single_launch"):
                    a.single_launch = False
                ya = a.decode_step(q, k1, v1, p)
                single_launch_switch(True)
                yb = b.decode_step(q, k1, v1, p)
                torch.cuda.synchronize()
                assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), (strategy, H, S, t)
                for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
                    if na not in ("next_key", "step_commit"):  # (pipeline bookkeeping of the fused step, not reference state)
                        assert torch.equal(ta, tb), (strategy, H, S, t, na)
    from cold_compress_amd.attention_utils import single_launch_status

    assert single_launch_status() == 0


def test_per_device_single_launch_knob():
    """the knob for shared devices is a property of ONE device (cc_decode_step_device_single_launch, boundary header;
    attention_utils.set_device_single_launch) — co-residency is a per-device fact.  Off: the availability queries of this device answer
    0 and a cache's steps take the two-launch forms, leaving state bit-identical to the single-launch twin's; on again: the single
    launch is back.  The process-wide A/B switch of the debug header is untouched by it."""
    from cold_compress_amd import _abi
    from cold_compress_amd.attention_utils import set_device_single_launch, single_launch_status
    from cold_compress_amd.cache import step_is_recoverable

    fns = _abi.lib()
    dev = torch.device(DEV, torch.cuda.current_device())
    H, HQ, S, D, dtype = 8, 32, 1024, 128, torch.bfloat16
    a, b = _mk(H, S, D, dtype), _mk(H, S, D, dtype)
    _seed(a, torch.Generator().manual_seed(5), S - 3)
    _seed(b, torch.Generator().manual_seed(5), S - 3)
    gen = torch.Generator().manual_seed(6)
    assert fns["cc_decode_step_single_launch"](HQ, H, S, D, 1) == 1 and step_is_recoverable(a, HQ)
    try:
        assert set_device_single_launch(dev, False) is True  # (the previous setting)
        assert fns["cc_decode_step_single_launch_enabled"]() == 0 and not step_is_recoverable(a, HQ)
        for t in range(4):
            p = torch.tensor([S + 10 + t], dtype=torch.int32, device=DEV)
            k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
            q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
            set_device_single_launch(dev, False)
            ya = a.decode_step(q, k1, k1, p)   # two launches: the device is "shared"
            set_device_single_launch(dev, True)
            yb = b.decode_step(q, k1, k1, p)   # one launch
            torch.cuda.synchronize()
            assert torch.allclose(ya.float(), yb.float(), rtol=2.0 ** -7, atol=1e-6), t
            for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
                if na not in ("next_key", "step_commit"):
                    assert torch.equal(ta, tb), (t, na)
            assert (a.step_commit[:, 2:66] == -1).all(), "the two-launch form writes no commit words"
    finally:
        set_device_single_launch(dev, True)
    assert fns["cc_decode_step_single_launch_enabled"]() == 1 and single_launch_status() == 0


def test_hand_off_timeout_is_reported_loudly():
    """A single-launch step whose workgroups were not all resident gives up after a bounded spin and leaves a word in the decode
    workspace; the harness's generate() (and bench.py) check it and raise — simulated here by setting the word."""
    from cold_compress_amd import _abi
    from cold_compress_amd.attention_utils import _WS, check_single_launch_status, single_launch_status

    kv = _mk(2, 256, 128, torch.bfloat16)
    _seed(kv, torch.Generator().manual_seed(1), 250)
    q = torch.randn(1, 8, 1, 128).to(torch.bfloat16).to(DEV)
    k1 = torch.randn(1, 2, 1, 128).to(torch.bfloat16).to(DEV)
    kv.decode_step(q, k1, k1, torch.tensor([300], dtype=torch.int32, device=DEV))
    assert single_launch_status() == 0
    check_single_launch_status()
    ws = _WS[(str(kv.pos.device), "decode")]
    off = int(_abi.lib()["cc_decode_step_status_offset"]())
    try:
        ws[off:off + 4].view(torch.int32).fill_(1)
        assert kv.step_status() == 1
        with pytest.raises(_abi.ColdCompressError):
            check_single_launch_status(kv.pos.device)
    finally:
        ws[off:off + 4].view(torch.int32).fill_(0)
    assert single_launch_status() == 0


def test_fused_step_differential_fuzz():
    """tools/fuzz_step.py on a fixed seed: ~110 random (policy, dtype, heads, length, fill level, sinks, window, cache_bits)
    cases, every one on the same workspace — the fused step (single launch where the shape allows) vs the three-call path, the
    history rings, and the fused quantised cache vs the 16-bit step on its dequantised values."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_step.py"), "--n", "120", "--seed", "3"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches, single-launch hand-off timeouts: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
