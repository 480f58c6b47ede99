#!/usr/bin/env python3
"""Where the single-launch layer step spends its time: per-workgroup s_memtime stamps (cc_decode_step_trace) of one
launch inside a back-to-back graph of launches over rotating caches.  Prints one JSON object per configuration.

    python tools/trace_one.py [--S 4096] [--pollall]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cold_compress_amd import _abi  # noqa: E402
from cold_compress_amd.cache import get_cache_constructor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, nargs="+", default=[4096])
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--HQ", type=int, default=32)
    ap.add_argument("--abl", type=int, nargs="+", default=[0], help="measurement bits of the phases word (0 = the product path)")
    ap.add_argument("--quant", action="store_true", help="the fused quantised cache (cache_bits=8, cache_quant_mode='fused')")
    ap.add_argument("--hybrid", action="store_true", help="KVCacheHybrid (the decode-ready state of tools/bench_policies.py)")
    ap.add_argument("--policy", default=None, choices=["l2", "recent_global", "random", "full"],
                    help="another policy's step through its class (the state of tools/bench_policies.py)")
    ap.add_argument("--wide", type=int, default=1, help="cc_decode_step_set_wide")
    ap.add_argument("--advance", action="store_true",
                    help="the position moves on behind every replay of the graph (r6): without it the recoverable steps find their position "
                         "committed after the first replay and only REPLAY it — no insert, no state stores — and the traced launch is lighter "
                         "than a real step")
    a = ap.parse_args()
    _abi.lib()["cc_decode_step_set_wide"](a.wide)
    _abi.probe_device()  # (r5: loading the library no longer probes the dispatch order: without this the tool measures the memory hand-off)
    dev, D, H, HQ = "cuda", 128, a.H, a.HQ
    fns = _abi.lib()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for S in a.S:
        n_buf = max(4, min(64, (600 << 20) // (2 * H * S * D * 2) + 1))
        cls, rk = get_cache_constructor("heavy_hitter")
        kw = dict(max_cache_length=S, global_tokens=4, max_seq_length=4 * S, cache_bits=None, recent_window=10,
                  history_window_size=1, attn_thresholding=False)
        caches = []
        via_class = a.hybrid or a.policy is not None
        if via_class:
            from bench_policies import make

            caches = [make("hybrid" if a.hybrid else a.policy, H, S, D) for _ in range(n_buf)]
        for _ in range(0 if via_class else n_buf):
            lk = {k: kw[k] for k in rk}
            if a.quant:
                lk.update(cache_bits=8, cache_quant_mode="fused")
            with torch.device(dev):
                kv = cls(1, H, D, torch.bfloat16, **lk)
            if a.quant:
                kv.k_cache_q.copy_((torch.randn(kv.cache_shape, device=dev) * 40 + 128).clamp_(0, 255).to(torch.uint8))
                kv.v_cache_q.copy_((torch.randn(kv.cache_shape, device=dev) * 40 + 128).clamp_(0, 255).to(torch.uint8))
                kv.kv_qparams[..., 0::2] = 6.4 / 255
                kv.kv_qparams[..., 1::2] = -3.2
            else:
                kv.k_cache.normal_()
                kv.v_cache.normal_()
            kv.pos[0] = torch.stack([torch.randperm(S + 64, device=dev)[:S] for _ in range(H)]).int()
            kv.mask.fill_(True)
            kv.cache_cts.fill_(S)
            kv.attn_history_num.uniform_()
            kv.attn_history_denom.fill_(3)
            caches.append(kv)
        nbytes = fns["cc_decode_attn_workspace_bytes"](HQ, H, S, D, 1)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        q = torch.randn(HQ, D, device=dev).to(torch.bfloat16)
        y = torch.empty(HQ, D, device=dev, dtype=torch.bfloat16)
        k1 = torch.randn(H, D, device=dev).to(torch.bfloat16)
        pos = torch.tensor([S + 100], dtype=torch.int32, device=dev)
        n_wg = 64 * H
        trace = torch.zeros((n_wg, 16), dtype=torch.int64, device=dev)
        for kv in caches:
            kv.prepare_decode(pos)
        q4, k4, ids = q.view(1, HQ, 1, D), k1.view(1, H, 1, D), torch.tensor([[11]], device=dev)

        def fstep(i, phases):
            kv = caches[i % n_buf]
            if a.hybrid:  # the class's own step (its workspace, the library's choice of launch form)
                kv.decode_step(q4, k4, k4, pos, input_ids=ids)
                return
            if via_class:
                kv.decode_step(q4, k4, k4, pos)
                return
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if a.quant:
                rc = fns["cc_decode_step_quant"](
                    kv._view(), p(kv.kv_qparams), 8, 1, p(q), p(k1), p(k1), p(pos), p(kv.attn_history_num), p(kv.attn_history_denom),
                    p(kv.attn_counter), None, p(kv.next_key), 4, 10, HQ, 1.0 / math.sqrt(D), p(y), None, p(ws), nbytes, st, phases)
                assert rc == 0, rc
                return
            rc = fns["cc_decode_step_heavy_hitter_phases"](
                kv._view(), p(q), p(k1), p(k1), p(pos), p(kv.attn_history_num), p(kv.attn_history_denom),
                p(kv.attn_counter), p(kv.next_key), 4, 10, HQ, 1.0 / math.sqrt(D), p(y), None, p(ws), nbytes, st, phases)
            assert rc == 0, rc

        for abl in a.abl:
            ph = 3 | 0x20000 | (abl << 8)
            fns["cc_decode_step_trace"](p(trace))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fstep(0, ph)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(n_buf):
                    fstep(i, ph)
                if a.advance:
                    pos.add_(1)
            fns["cc_decode_step_trace"](None)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            us_per = e0.elapsed_time(e1) * 1e3 / n_buf
            t = trace.cpu().numpy().astype(np.int64)
            used = t[:, 0] != 0
            t = t[used]
            # s_memtime (columns 0-5, 11-15) is a per-XCD clock: only differences INSIDE a workgroup mean anything (r3's summary
            # subtracted a global minimum across XCDs from some columns: garbage means).  s_memrealtime (columns 6-8,
            # 100 MHz) is one clock for the device: it places the workgroups' starts and ends against each other.
            out = {"S": S, "quant": bool(a.quant), "hybrid": bool(a.hybrid), "policy": a.policy or ("hybrid" if a.hybrid else "heavy_hitter"),
                   "abl": abl, "us_per_launch_events": round(us_per, 2), "workgroups": int(used.sum())}
            base = t[:, 6].min()
            r0, r1, r2 = (t[:, 6] - base) / 100.0, (t[:, 7] - base) / 100.0, (t[:, 8] - base) / 100.0  # us on the device clock
            # ticks of s_memtime per microsecond, from each workgroup's own (start, end) pair on both clocks
            tpu = float(np.median((t[:, 5] - t[:, 0]) / np.maximum(r2 - r0, 1e-3)))
            out["memtime_ticks_per_us"] = round(tpu, 1)
            hw, xcc = t[:, 9], t[:, 10] & 15
            cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)  # cu, sh, se, xcc
            out["distinct_cus"] = int(len(np.unique(cu)))
            nsplit = int(used.sum()) // H
            xh = xcc.reshape(H, nsplit)
            out["xcds_per_kv_head"] = [int(len(np.unique(xh[h_]))) for h_ in range(H)]

            def mmm(v):
                return [round(float(np.min(v)), 2), round(float(np.mean(v)), 2), round(float(np.max(v)), 2)]

            out["start_us_min_mean_max"] = mmm(r0)
            out["end_us_min_mean_max"] = mmm(r2)
            out["launch_span_us"] = round(float(r2.max() - r0.min()), 2)
            out["gap_to_next_launch_us"] = round(us_per - float(r2.max() - r0.min()), 2)
            # phases of a workgroup (wave 0), microseconds since ITS start
            cols = [("k_requested", 3), ("k_arrived", 11), ("scores_ready", 12), ("pv_issued", 13), ("merge_barrier_passed", 1), ("o_published", 2),
                    ("ml_gathered", 4), ("ML_visible", 14), ("o_gathered", 15), ("end", 5)]
            ph = {}
            for nm, col in cols:
                if (t[:, col] != 0).all():
                    ph[nm] = mmm((t[:, col] - t[:, 0]) / tpu)
            out["since_own_start_us_min_mean_max"] = ph
            # per kv head: the LAST workgroup's merge barrier on the device clock, and what follows it until the head's last end
            hr1, hr2 = r1.reshape(H, nsplit), r2.reshape(H, nsplit)
            out["head_last_merge_barrier_us"] = [round(float(x), 2) for x in hr1.max(axis=1)]
            out["head_end_minus_last_merge_barrier_us"] = [round(float(x), 2) for x in (hr2.max(axis=1) - hr1.max(axis=1))]
            out["head_merge_barrier_spread_us"] = round(float((hr1.max(axis=1) - hr1.min(axis=1)).mean()), 2)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
