"""Shared test helpers: golden-fixture loading and dtype plumbing."""
import os

import numpy as np
import torch

# CC_GOLDEN_DIR: the same fixture families made by the reference from OTHER seeds (oracle/gen_golden.py --seed_offset N --out DIR; only
# tests/test_oracle_fresh_seeds.py sets it, in the build container where /root/reference exists)
GOLDEN = os.environ.get("CC_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# CC_TEST_DEVICE=cpu (with CC_TEST_CPU_TWIN=1, tests/conftest.py): the fixture-driven `-m gpu` tests run the product's Python layer on CPU
# tensors over the oracle's twins (tests/cpu_twin.py) — tests/test_host_e2e_cpu.py launches that run inside the CPU suite
TEST_DEVICE = os.environ.get("CC_TEST_DEVICE", "cuda")

DT_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
DT_FROM_NAME = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def load_golden(name):
    """npz -> dict of torch tensors (16-bit floats restored from their uint16 bit patterns)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for k in z.files:
        if k.endswith("__dtype"):
            continue
        a = z[k]
        if k + "__dtype" in z.files:
            dt = torch.bfloat16 if str(z[k + "__dtype"]) == "bf16" else torch.float16
            out[k] = torch.from_numpy(a.view(np.int16).copy()).view(dt)
        elif a.dtype.kind in "US":
            out[k] = str(a)
        elif a.ndim == 0 and a.dtype.kind in "iub":
            out[k] = int(a)
        elif a.ndim == 0 and a.dtype.kind == "f":
            out[k] = float(a)
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


def to_np(t):
    """torch tensor -> numpy array the C ABI can read (16-bit floats as uint16)."""
    t = t.contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype == torch.bool:
        return t.numpy().astype(np.uint8)
    return t.numpy().copy()


def from_np(a, dtype):
    if dtype in (torch.bfloat16, torch.float16):
        return torch.from_numpy(a.view(np.int16).copy()).view(dtype)
    return torch.from_numpy(a.copy())


def hh_own_state_steps(o, kv, st, gen, p0, steps, HQ, g, w, dtype, scale_q=1.5):
    """`steps` decode steps of the fused heavy-hitter layer step (kv.decode_step on the DEVICE) beside the oracle's
    update -> attention -> history sequence, each side on ITS OWN numeric state (nothing is copied across): a differing eviction
    must be a rounding-level near-tie in the ORACLE's scores (cache.py:727-749) and is then followed, so that the caches stay
    comparable; y within 1e-3 + two roundings of the output dtype; positions identical after every step.
    st: dict(k, v, pos, mask, cts, num, denom, ctr) of host arrays (the oracle's state).  -> (justified, total)."""
    import ctypes as C
    import math

    H, S, D = kv.n_heads, kv.max_cache_length, kv.head_dim
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    code = DT_CODE[dtype]
    justified = total = 0
    for t in range(steps):
        p = p0 + t
        pt = torch.tensor([p], dtype=torch.int32)
        k1 = (scale_q * torch.randn(1, H, 1, D, generator=gen)).to(dtype)
        v1 = torch.randn(1, H, 1, D, generator=gen).to(dtype)
        q1 = (scale_q * torch.randn(1, HQ, 1, D, generator=gen)).to(dtype)
        pos_before = kv.pos.cpu()[0].numpy().copy()
        yd = kv.decode_step(q1.to(TEST_DEVICE), k1.to(TEST_DEVICE), v1.to(TEST_DEVICE), pt.to(TEST_DEVICE))
        torch.cuda.synchronize()
        pos_after = kv.pos.cpu()[0].numpy()
        idx_d = np.array([int(np.nonzero(pos_after[h] != pos_before[h])[0][0]) for h in range(H)])
        dn = np.maximum(st["denom"], 1).astype(np.float32)
        sc = (st["num"].astype(np.float32) / dn).astype(np.float32)
        sc[(st["pos"] < g) | (st["pos"] >= p - w)] = 1.0
        sc[st["pos"] == -1] = 0.0
        idx_o = sc.argmin(axis=1)
        for h in range(H):
            total += 1
            if idx_d[h] != idx_o[h]:
                gap = float(sc[h, idx_d[h]] - sc[h, idx_o[h]])
                assert 0 <= gap <= 2 * ulp * float(sc[h, idx_o[h]]) + 1e-12, f"step {t} head {h}: evicted {idx_d[h]} (score gap {gap})"
                justified += 1
                st["num"][h, idx_d[h]] = -1.0  # make the oracle follow the device's (equally good) choice
        view = o.view(st["k"], st["v"], st["pos"], st["mask"], st["cts"], code)
        idx = np.zeros(H, np.int64)
        o.call("cc_decode_update_heavy_hitter", C.byref(view), o.ptr(to_np(k1.reshape(H, D))), o.ptr(to_np(v1.reshape(H, D))),
               o.ptr(pt.numpy().copy()), o.ptr(st["num"]), o.ptr(st["denom"]), g, w, o.ptr(idx), None)
        assert np.array_equal(idx, idx_d), f"step {t}"
        yo1 = np.zeros((HQ, D), np.uint16)
        o.call("cc_decode_attn_gqa", o.ptr(to_np(q1.reshape(HQ, D))), o.ptr(st["k"]), o.ptr(st["v"]), o.ptr(st["mask"]), HQ, H, S, D, code,
               1.0 / math.sqrt(D), o.ptr(yo1), None, None, o.ptr(st["num"]), o.ptr(st["denom"]), o.ptr(st["ctr"]), None, 0, None)
        yr = from_np(yo1, dtype).float()
        assert (yd.cpu().float()[0, :, 0] - yr).abs().max() <= 1e-3 + 2 * ulp * yr.abs().max(), f"step {t}: y"
        assert np.array_equal(pos_after, st["pos"]), f"step {t}: positions"
    num_d = kv.attn_history_num.cpu()[0, :, :, 0].numpy()
    assert np.array_equal(kv.attn_history_denom.cpu()[0].numpy(), st["denom"])
    # drift of the float64 history after `steps` unsynchronised steps: each step adds one dtype-rounded probability per slot
    assert np.allclose(num_d, st["num"], rtol=2 * ulp, atol=steps * 2.0 ** -16), float(np.abs(num_d - st["num"]).max())
    assert np.array_equal(to_np(kv.k_cache.cpu()[0]), st["k"]) and np.array_equal(to_np(kv.v_cache.cpu()[0]), st["v"])
    assert np.array_equal(kv.cache_cts.cpu().numpy().reshape(-1)[: len(st["cts"])], st["cts"])
    return justified, total
