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
    a = ap.parse_args()
    _abi.lib()["cc_decode_step_set_wide"](a.wide)
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
            t0 = t[:, 0].min()
            rel = (t[:, :6] - t0).astype(np.float64)
            span = rel[:, 5].max()
            names = ["start", "stream_done", "published", "sentinel_seen", "gathered", "end"]
            out = {"S": S, "quant": bool(a.quant), "hybrid": bool(a.hybrid), "policy": a.policy or ("hybrid" if a.hybrid else "heavy_hitter"), "abl": abl, "us_per_launch_events": round(us_per, 2), "workgroups": int(used.sum()),
                   "ticks_total": float(span)}
            base = t[:, 6].min()
            r0, r1, r2 = t[:, 6] - base, t[:, 7] - base, t[:, 8] - base
            hw, xcc = t[:, 9], t[:, 10] & 15
            cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)  # cu, sh, se, xcc
            out["distinct_cus"] = int(len(np.unique(cu)))
            out["stream_done_hist_500ns"] = np.bincount((r1 // 50).astype(np.int64)).tolist()
            out["stream_done_by_xcc_mean"] = [round(float(r1[xcc == x].mean()), 1) for x in range(8)]
            out["stream_done_by_xcc_max"] = [int(r1[xcc == x].max()) for x in range(8)]
            # first / second workgroup on the same CU (by start time)
            first, second = [], []
            for cid in np.unique(cu):
                idx = np.where(cu == cid)[0]
                idx = idx[np.argsort(r0[idx])]
                if len(idx) >= 2:
                    first.append(r1[idx[0]]); second.append(r1[idx[1]])
            if first:
                out["cu_first_wg_stream_done_mean_max"] = [round(float(np.mean(first)), 1), int(np.max(first))]
                out["cu_second_wg_stream_done_mean_max"] = [round(float(np.mean(second)), 1), int(np.max(second))]
            late = np.argsort(-r1)[:12]
            out["latest"] = [[int(r1[i]), int(i // (len(r1) // H)), int(i % (len(r1) // H)), int(xcc[i]), int(cu[i] & 255)] for i in late]
            out["realtime_10ns"] = {"start": [int(r0.min()), float(r0.mean()), int(r0.max())],
                                    "stream_done": [int(r1.min()), float(r1.mean()), int(r1.max())],
                                    "end": [int(r2.min()), float(r2.mean()), int(r2.max())]}
            # per-head: when did the LAST workgroup of the head publish, and how long after that did the head's workgroups finish
            nsplit = int(used.sum()) // H
            ph_ = rel.reshape(H, nsplit, 6)
            out["stream_mean"] = round(float((ph_[:, :, 1] - ph_[:, :, 0]).mean()), 1)
            out["publish_mean"] = round(float((ph_[:, :, 2] - ph_[:, :, 1]).mean()), 1)
            out["sentinel_wait_mean"] = round(float((ph_[:, :, 3] - ph_[:, :, 2]).mean()), 1)
            hr1 = r1.reshape(H, nsplit)
            out["head_stream_done_spread_10ns"] = round(float((hr1.max(axis=1) - hr1.min(axis=1)).mean()), 1)
            out["end_minus_head_last_stream_done_10ns"] = round(float((r2.reshape(H, nsplit).max(axis=1) - hr1.max(axis=1)).mean()), 1)
            out["gather_rtt_mean"] = round(float((ph_[:, :, 4] - ph_[:, :, 3]).mean()), 1)
            out["finish_mean"] = round(float((ph_[:, :, 5] - ph_[:, :, 4]).mean()), 1)
            if t.shape[1] > 13 and (t[:, 11] != 0).all():  # wave 0 of every workgroup, ticks since its start
                for nm, col in (("k_arrived", 11), ("scores_ready", 12), ("pv_issued", 13)):
                    d_ = (t[:, col] - t[:, 0]).astype(np.float64)
                    out[nm + "_min_mean_max"] = [float(d_.min()), round(float(d_.mean()), 1), float(d_.max())]
                if t.shape[1] > 15 and (t[:, 14] != 0).all():
                    out["finish_ML_weights_mean"] = round(float((t[:, 14] - t[:, 4]).mean()), 1)
                    out["finish_slots_y_keys_mean"] = round(float((t[:, 15] - t[:, 14]).mean()), 1)
                    # r3, early (m, l): [4] = (m, l) gathered, [14] = final (M, L) visible to every wave, [15] = partial-O gather
                    # complete (the per-slot pass ran in its shadow), [5] = end -> what is left behind the last O granule
                    out["tail_behind_o_gather_mean_max"] = [round(float((t[:, 5] - t[:, 15]).mean()), 1), float((t[:, 5] - t[:, 15]).max())]
                    out["o_gather_done_min_mean_max"] = [float((t[:, 15] - t0).min()), round(float((t[:, 15] - t0).mean()), 1), float((t[:, 15] - t0).max())]
                    out["ml_gather_done_min_mean_max"] = [float((t[:, 4] - t0).min()), round(float((t[:, 4] - t0).mean()), 1), float((t[:, 4] - t0).max())]
                    out["end_min_mean_max"] = [float((t[:, 5] - t0).min()), round(float((t[:, 5] - t0).mean()), 1), float((t[:, 5] - t0).max())]
                d_ = (t[:, 1] - t[:, 0]).astype(np.float64)
                out["stream_done_min_mean_max"] = [float(d_.min()), round(float(d_.mean()), 1), float(d_.max())]
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
