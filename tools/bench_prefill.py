#!/usr/bin/env python3
"""Time the prefill attention pass (cc_prefill_attn_bands: V^T permutation, row statistics, P.V + side planes, plane fold) at
the BASELINE prompt lengths, Llama-3-8B head geometry, one layer per call.  One JSON line per configuration.

    python tools/bench_prefill.py [--L 8192 16384] [--bands]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cold_compress_amd.attention_utils import prefill_attention  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, nargs="+", default=[8192, 16384])
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--HQ", type=int, default=32)
    ap.add_argument("--bands", action="store_true", help="also the FastGen band plane at 0.1 * L (hybrid profiling)")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev, D = "cuda", 128
    for L in a.L:
        gen = torch.Generator(device=dev).manual_seed(L)
        q = torch.randn(1, a.HQ, L, D, device=dev, generator=gen).to(torch.bfloat16)
        k = torch.randn(1, a.H, L, D, device=dev, generator=gen).to(torch.bfloat16)
        v = torch.randn(1, a.H, L, D, device=dev, generator=gen).to(torch.bfloat16)
        bands = [max(1, int(0.1 * L))] if a.bands else []
        for ret in (True, False):
            for _ in range(2):
                prefill_attention(q, k, v, return_attn=ret, bands=bands)
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                prefill_attention(q, k, v, return_attn=ret, bands=bands)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            # causal flops = 2 HQ L^2 D for QK^T + P.V (the useful work); the two-pass path (return_attn: column / band /
            # window sums need the probabilities) recomputes QK^T in its second pass: x1.5 EXECUTED.  The single pass (return_attn
            # False, no side planes) executes the useful flops only — r3's file applied the x1.5 to it too and overstated it.
            useful = 2 * a.HQ * L * L * D
            two_pass = ret or bool(bands)
            executed = useful * (1.5 if two_pass else 1.0)
            t = ts[len(ts) // 2] * 1e-3
            print(json.dumps({"L": L, "H": a.H, "HQ": a.HQ, "return_attn": ret, "bands": bands, "passes": 2 if two_pass else 1,
                              "ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3),
                              "useful_causal_TFLOPs": round(useful / t / 1e12, 1), "executed_TFLOPs": round(executed / t / 1e12, 1),
                              "frac_of_2.5PF_dense_bf16_useful": round(useful / t / 2.5e15, 3)}), flush=True)


if __name__ == "__main__":
    main()
