"""KVCacheHeavyHitter with a finite history window (history_window_size = 8 and 33; ScissorHands' m), pinned against traces
captured from the reference (tests/golden/f2_hh_w8_*.npz): oracle on CPU, HIP path on the GPU.  The prefill column
mean is tolerance-class (unspecified fp32 summation order), after which the replay continues on the reference's
state and every eviction index, count and the ring contents are compared bit-exactly."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import DT_CODE, DT_FROM_NAME, load_golden, to_np

NAMES = ["f2_hh_w8_f32.npz", "f2_hh_w8_bf16.npz", "f2_hh_w33_long_bf16.npz"]
WINDOW = {"f2_hh_w33_long_bf16.npz": 33}  # history_window_size of each fixture (default 8)


@pytest.mark.parametrize("name", NAMES)
def test_ring_replay_oracle(oracle, name):
    W = WINDOW.get(name, 8)
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    code = DT_CODE[dtype]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    o = oracle
    es = np.float32 if code == 0 else np.uint16
    k, v = np.zeros((H, S, D), es), np.zeros((H, S, D), es)
    pos, mask, cts = np.full((H, S), -1, np.int32), np.zeros((H, S), np.uint8), np.zeros(1, np.int32)
    view = o.view(k, v, pos, mask, cts, code)
    p0 = np.arange(T, dtype=np.int64).reshape(1, T).copy()
    o.call("cc_prefill_fill", C.byref(view), o.ptr(to_np(f["k0"][0])), o.ptr(to_np(f["v0"][0])), o.ptr(p0), 1, T, None)
    num = to_np(f["num_after_prefill"][0])  # continue from the reference's prefill state
    denom = f["denom_after_prefill"][0].numpy().copy()
    counter = np.array([1], np.int64)
    # prefill column mean -> ring slot 0 (tolerance class)
    colsum = np.zeros((H, T), np.float32)
    o.call("cc_attn_colsum", o.ptr(to_np(f["attn0"][0])), H, T, T, code, o.ptr(colsum), None)
    mean = np.zeros((H, T), es)
    o.call("cc_colsum_to_mean", o.ptr(colsum), None, H, T, code, o.ptr(mean), None)
    num2, den2, ctr2 = np.zeros((H, S, W), es), np.zeros((H, S), np.int32), np.zeros(1, np.int64)
    o.call("cc_hh_ring_update", o.ptr(num2), o.ptr(den2), o.ptr(ctr2), o.ptr(mean), H, S, T, W, code, None, None, None)
    from helpers import from_np
    assert torch.allclose(from_np(num2, dtype).float(), from_np(num, dtype).float(), rtol=2 ** -7 if code else 1e-6, atol=0)
    assert np.array_equal(den2, denom) and ctr2[0] == 1
    for t in range(f["steps"]):
        view = o.view(k, v, pos, mask, cts, code)
        idx = np.zeros(H, np.int64)
        pp = np.array([T + t], np.int32)
        o.call("cc_decode_update_heavy_hitter_ring", C.byref(view), o.ptr(to_np(f["k_new"][t].reshape(H, D))),
               o.ptr(to_np(f["v_new"][t].reshape(H, D))), o.ptr(pp), o.ptr(num), o.ptr(denom), W, g, w, o.ptr(idx), None, None, None)
        assert np.array_equal(idx, f["idx"][t].numpy()), f"step {t}"
        assert np.array_equal(cts, f["cache_cts_steps"][t].numpy())
        o.call("cc_hh_ring_update", o.ptr(num), o.ptr(denom), o.ptr(counter), o.ptr(to_np(f["attn"][t][0, :, 0])), H, S, S, W, code, None, None, None)
    assert np.array_equal(num, to_np(f["final_num"][0])) and np.array_equal(denom, f["final_denom"][0].numpy())
    assert np.array_equal(pos, f["final_pos"][0].numpy()) and np.array_equal(k, to_np(f["final_k"][0]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_ring_replay_gpu(name):
    import cold_compress_amd.cache as cache

    W = WINDOW.get(name, 8)
    f = load_golden(name)
    dtype = DT_FROM_NAME[f["dtype"]]
    H, S, D, T, g, w = f["H"], f["S"], f["D"], f["T_prefill"], f["g"], f["w"]
    dev = __import__("helpers").TEST_DEVICE
    with torch.device(dev):
        kv = cache.KVCacheHeavyHitter(1, H, D, dtype, max_cache_length=S, max_seq_length=4 * S, cache_bits=None, global_tokens=g,
                                      history_window_size=W, recent_window=w, attn_thresholding=False)
    pos0 = torch.arange(T, device=dev)
    kv.update_kv(pos0, f["k0"].to(dev), f["v0"].to(dev), True)
    kv.update_state(pos0, f["k0"].to(dev), f["v0"].to(dev), True, f["attn0"].to(dev))
    assert torch.allclose(kv.attn_history_num.cpu().float(), f["num_after_prefill"].float(),
                          rtol=2 ** -7 if dtype != torch.float32 else 1e-6, atol=0)
    assert torch.equal(kv.attn_history_denom.cpu(), f["denom_after_prefill"])
    kv.attn_history_num.copy_(f["num_after_prefill"].to(dev))
    for t in range(f["steps"]):
        p = torch.tensor([T + t], dtype=torch.int32, device=dev)
        k1, v1 = f["k_new"][t].to(dev), f["v_new"][t].to(dev)
        kv.update_kv(p, k1, v1, False)
        torch.cuda.synchronize()
        assert kv._idx_buf().cpu().tolist() == f["idx"][t].tolist(), f"step {t}"
        kv.update_state(p, k1, v1, False, f["attn"][t].to(dev))
    assert torch.equal(kv.attn_history_num.cpu().float(), f["final_num"].float())
    assert torch.equal(kv.attn_history_denom.cpu(), f["final_denom"]) and torch.equal(kv.pos.cpu(), f["final_pos"])
    assert torch.equal(kv.attn_counter.cpu(), f["final_counter"])
