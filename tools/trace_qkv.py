#!/usr/bin/env python3
"""The layer step with the QKV projection folded in (cc_decode_step_qkv_rc): what it takes against its two-launch twin, and —
with a CC_QKV_TRACE build of cc_attn_decode_qkv.hip (tools/ab_variant_qkv.sh) loaded through CC_LIB — where a workgroup spends it.

    python tools/trace_qkv.py [--S 4096] [--H 8] [--HQ 32] [--K 4096] [--layers 32]

Timing: HIP events around hipGraph replays of `layers` back-to-back steps, each on its own cache and its own weight matrix
(> 512 MiB of distinct bytes: neither the Infinity Cache nor an L2 serves a re-read), positions advancing between replays.
Prints one JSON object: fused_us, twin_us (cc_gemv_fused + single-launch step), gemv_us, step_us, and — trace builds — the mean
time of every stamp since the launch's first workgroup started (device clock, us)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cold_compress_amd import _abi  # noqa: E402

if os.environ.get("CC_LIB"):  # a variant build (tools/ab_variant_qkv.sh)
    _abi.LIB_PATH = os.path.abspath(os.environ["CC_LIB"])
from cold_compress_amd.cache import get_cache_constructor  # noqa: E402
from cold_compress_amd.harness import glue  # noqa: E402

STAMPS = ["entry", "weights_issued", "compute_start", "x_normed", "dots_done", "published", "gathered", "q_in_lds", "scores", "pv_issued",
          "merge_barrier", "end"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=4096)
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--HQ", type=int, default=32)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--policy", default="heavy_hitter")
    a = ap.parse_args()
    dev, D, H, HQ, S, K, NL = "cuda", 128, a.H, a.HQ, a.S, a.K, a.layers
    dt = torch.bfloat16
    cls, rk = get_cache_constructor(a.policy)
    kw = dict(max_cache_length=S, global_tokens=4, max_seq_length=4 * S, cache_bits=None, recent_window=10, history_window_size=1,
              attn_thresholding=False)
    gen = torch.Generator().manual_seed(1)
    caches, ws = [], []
    for _ in range(NL):
        with torch.device(dev):
            kv = cls(1, H, D, dt, **{k: kw[k] for k in rk})
        kv.update_kv(torch.arange(S, device=dev), torch.randn(1, H, S, D, device=dev).to(dt), torch.randn(1, H, S, D, device=dev).to(dt), True)
        if hasattr(kv, "attn_history_num"):
            kv.attn_history_num[0, :, :, 0] = torch.rand(H, S, device=dev, dtype=torch.float64)
            kv.attn_history_denom[0] = 1
        caches.append(kv)
        ws.append((0.02 * torch.randn((HQ + 2 * H) * D, K, device=dev)).to(dt))
    nw = torch.ones(K, device=dev, dtype=dt)
    x = torch.randn(1, 1, K, device=dev).to(dt)
    p = torch.tensor([S + 8], dtype=torch.int32, device=dev)
    fr = torch.stack([torch.ones(D // 2), torch.zeros(D // 2)], dim=-1).to(dt).to(dev).contiguous()
    h = torch.empty_like(x)
    if not caches[0].qkv_step_available(HQ, K):
        print(json.dumps({"error": "shape not eligible for the QKV form"}))
        return
    for kv in caches:
        kv.prepare_decode(p)

    def fused():
        for kv, w in zip(caches, ws):
            kv.decode_step_qkv(w, None, x, None, nw, 1e-5, h, fr, p, HQ)

    def twin():
        for kv, w in zip(caches, ws):
            qkv = glue.gemv_fused(w, x, norm_weight=nw, eps=1e-5, h_out=h, freqs=fr, rope_rows=(HQ + H) * D, head_dim=D)
            kv.decode_step(qkv[: HQ * D].view(1, HQ, 1, D), qkv[HQ * D: (HQ + H) * D].view(1, H, 1, D), qkv[(HQ + H) * D:].view(1, H, 1, D), p)

    def gemv_only():
        for w in ws:
            glue.gemv_fused(w, x, norm_weight=nw, eps=1e-5, h_out=h, freqs=fr, rope_rows=(HQ + H) * D, head_dim=D)

    qfix = torch.randn(1, HQ, 1, D, device=dev).to(dt)
    kfix = torch.randn(1, H, 1, D, device=dev).to(dt)

    def step_only():
        for kv in caches:
            kv.decode_step(qfix, kfix, kfix, p)

    def timed(fn):
        fn()  # warm (workspace, probes)
        p.add_(1)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        ts = []
        for _ in range(a.iters):
            p.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / NL)
        ts.sort()
        return ts[len(ts) // 2], g

    out = {"S": S, "H": H, "HQ": HQ, "K": K, "layers": NL, "policy": a.policy}
    out["fused_us"], gf = timed(fused)
    out["twin_us"], _ = timed(twin)
    out["gemv_us"], _ = timed(gemv_only)
    out["step_us"], _ = timed(step_only)
    from cold_compress_amd.attention_utils import single_launch_status

    out["status_word"] = int(single_launch_status(torch.device(dev)))
    bytes_fused = (HQ + 2 * H) * D * K * 2 + 2 * H * S * D * 2 + 29 * H * S
    out["fused_bytes"] = bytes_fused
    out["fused_tb_s"] = round(bytes_fused / out["fused_us"] / 1e6, 3)
    # ---- stamps (trace builds only: the hook exists in every build, the stamps are compiled in with -DCC_QKV_TRACE=1)
    _abi.lib()
    lib = _abi._LIB
    if hasattr(lib, "cc_debug_qkv_trace"):
        n_wg = 1024
        buf = torch.zeros(n_wg * 16, dtype=torch.int64, device=dev)
        fn = lib.cc_debug_qkv_trace
        fn.argtypes = [C.c_void_p]
        fn.restype = None
        fn(C.c_void_p(buf.data_ptr()))
        acc = []
        for _ in range(8):
            buf.zero_()
            p.add_(1)
            fused()  # eager: the LAST layer's launch leaves its stamps
            torch.cuda.synchronize()
            t = buf.cpu().numpy().reshape(n_wg, 16)
            t = t[t[:, 0] != 0]
            if len(t) == 0:
                break
            t0 = t[:, 0].min()
            acc.append(((t[:, :12] - t0) / 100.0))  # 100 MHz -> us
        fn(None)
        if acc:
            m = np.concatenate(acc, 0)
            out["trace_workgroups"] = int(m.shape[0] / len(acc))
            out["trace_us_mean"] = {n: round(float(m[:, i].mean()), 2) for i, n in enumerate(STAMPS)}
            out["trace_us_max"] = {n: round(float(m[:, i].max()), 2) for i, n in enumerate(STAMPS)}
            out["trace_us_min"] = {n: round(float(m[:, i].min()), 2) for i, n in enumerate(STAMPS)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
