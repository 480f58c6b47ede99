#!/usr/bin/env python3
"""Micro-benchmark of the decode hot-path kernels vs cache length (evidence for DESIGN.md §roofline).

For each S: rotate over enough distinct K/V buffers to exceed the 256 MB Infinity Cache, launch the split
kernel (phase 1), split+combine (phase 3) and the heavy-hitter evict+insert from a hipGraph, and report the
per-launch time and algorithmic GB/s.  Prints one JSON object per line.

    python tools/sweep_attn.py [--S 1024 2560 4096 ...]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cold_compress_amd import _abi  # noqa: E402
from cold_compress_amd.cache import get_cache_constructor  # noqa: E402


def timed_graph(fn, n_items, iters=10):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n_items):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n_items)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, nargs="+", default=[1024, 2560, 4096, 8192, 18432, 32768, 65536])
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--HQ", type=int, default=32)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--ablate", action="store_true", help="time the split kernel with parts switched off")
    ap.add_argument("--fused", action="store_true", help="time the fused two-launch heavy-hitter step and its parts")
    a = ap.parse_args()
    dev = "cuda"
    fns = _abi.lib()
    _abi.probe_device()  # (r5: loading the library no longer probes the dispatch order: without this the tool measures the memory hand-off)
    H, HQ, D = a.H, a.HQ, a.D
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for S in a.S:
        kv_bytes = 2 * H * S * D * 2
        n_buf = max(4, min(64, (600 << 20) // kv_bytes + 1))
        cls, rk = get_cache_constructor("heavy_hitter")
        kw = dict(max_cache_length=S, global_tokens=4, max_seq_length=4 * S, cache_bits=None, recent_window=10,
                  history_window_size=1, attn_thresholding=False)
        caches = []
        for _ in range(n_buf):
            with torch.device(dev):
                kv = cls(1, H, D, torch.bfloat16, **{k: kw[k] for k in rk})
            kv.k_cache.normal_()
            kv.v_cache.normal_()
            kv.pos[0] = torch.stack([torch.randperm(S + 64, device=dev)[:S] for _ in range(H)]).int()
            kv.mask.fill_(True)
            kv.cache_cts.fill_(S)
            kv.attn_history_num.uniform_()
            kv.attn_history_denom.fill_(3)
            caches.append(kv)
        nbytes = fns["cc_decode_attn_workspace_bytes"](HQ, H, S, D, 1)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)  # zero: the single-launch step's epoch words live here
        q = torch.randn(HQ, D, device=dev).to(torch.bfloat16)
        y = torch.empty(HQ, D, device=dev, dtype=torch.bfloat16)
        k1 = torch.randn(H, D, device=dev).to(torch.bfloat16)
        pos = torch.tensor([S + 100], dtype=torch.int32, device=dev)

        def attn(i, phases, hist):
            kv = caches[i % n_buf]
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            hn = (p(kv.attn_history_num), p(kv.attn_history_denom), p(kv.attn_counter)) if hist else (None, None, None)
            rc = fns["cc_decode_attn_gqa_phases"](p(q), p(kv.k_cache), p(kv.v_cache), p(kv.mask), HQ, H, S, D, 1,
                                                   1.0 / math.sqrt(D), p(y), None, None, *hn, p(ws), nbytes, st, phases)
            assert rc == 0, rc

        def evict(i):
            kv = caches[i % n_buf]
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = fns["cc_decode_update_heavy_hitter"](kv._view(), p(k1), p(k1), p(pos), p(kv.attn_history_num),
                                                       p(kv.attn_history_denom), 4, 10, p(kv._idx_buf()), st)
            assert rc == 0, rc

        def step(i):
            evict(i)
            attn(i, 3, True)

        n = n_buf
        if a.fused:
            for kv in caches:
                kv.prepare_decode(pos)

            def fstep(i, phases):
                kv = caches[i % n_buf]
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = fns["cc_decode_step_heavy_hitter_phases"](
                    kv._view(), p(q), p(k1), p(k1), p(pos), p(kv.attn_history_num), p(kv.attn_history_denom),
                    p(kv.attn_counter), p(kv.next_key), 4, 10, HQ, 1.0 / math.sqrt(D), p(y), None, p(ws), nbytes, st, phases)
                assert rc == 0, rc

            res = {"S": S}
            one = fns["cc_decode_step_single_launch"](HQ, H, S, D, 1) == 1
            res["single_launch_supported"] = bool(one)
            if one:  # the single-launch step (0x20000 demands it; abl bit 2 << 8: the streaming part only, no hand-off / finish)
                for name, ph in (("step_one", 3 | 0x20000), ("step_one_noepi", 3 | 0x20000 | (2 << 8))):
                    t, tmin = timed_graph(lambda i, ph=ph: fstep(i, ph), n)
                    res[name + "_us"] = round(t, 2)
                    res[name + "_min_us"] = round(tmin, 2)
            for name, ph in (("step", 3 | 0x10000), ("split", 1), ("split_nobranch", 1 | (64 << 8)), ("split_nokey", 1 | (128 << 8)), ("combine", 2), ("combine_nokey", 2 | (8 << 8)),
                             ("combine_noy", 2 | (16 << 8)), ("combine_noslot", 2 | (32 << 8)),
                             ("combine_empty", 2 | (56 << 8))):
                t, tmin = timed_graph(lambda i, ph=ph: fstep(i, ph), n)
                res[name + "_us"] = round(t, 2)
            t, _ = timed_graph(lambda i: attn(i, 2, True), n)
            res["combine_unfused_hist_us"] = round(t, 2)
            print(json.dumps(res), flush=True)
            continue
        if a.ablate:
            res = {"S": S}
            for name, bits in (("full", 0), ("no_store", 1), ("no_epilogue", 2), ("no_mask", 4), ("no_store_epi", 3), ("no_partial_store", 16), ("no_stores_at_all", 17),
                               ("none", 7)):
                t, _ = timed_graph(lambda i, b=bits: attn(i, 1 | (b << 8), False), n)
                res[name + "_us"] = round(t, 2)
            t, _ = timed_graph(lambda i: attn(i, 2, True), n)
            res["combine_only_fused_us"] = round(t, 2)
            t, _ = timed_graph(lambda i: attn(i, 2, False), n)
            res["combine_only_plain_us"] = round(t, 2)
            print(json.dumps(res), flush=True)
            continue
        t_split, t_split_min = timed_graph(lambda i: attn(i, 1, False), n)
        t_both, _ = timed_graph(lambda i: attn(i, 3, True), n)
        t_ev, _ = timed_graph(evict, n)
        t_step, _ = timed_graph(step, n)
        alg_split = kv_bytes + H * S + HQ * D * 2
        step_bytes = kv_bytes + H * S * 29
        print(json.dumps({"S": S, "n_buf": n_buf, "split_us": round(t_split, 2), "split_min_us": round(t_split_min, 2),
                          "split_GBps": round(alg_split / t_split / 1e3, 1), "attn_both_us": round(t_both, 2),
                          "evict_us": round(t_ev, 2), "layer_step_us": round(t_step, 2),
                          "layer_step_GBps": round(step_bytes / t_step / 1e3, 1),
                          "layer_step_frac_of_8TBps": round(step_bytes / t_step / 1e3 / 8000, 4)}), flush=True)
        del caches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
