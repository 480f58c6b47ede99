#!/usr/bin/env python3
"""cc_gemv_fused vs the library GEMV (torch F.linear -> hipBLASLt) at the decode shapes of Llama-3-8B, bf16.
Rotates over enough distinct weight copies to exceed the 256 MB Infinity Cache; hipGraph replays, HIP events.
Also checks the fused variants against an fp32 torch composition of the same op chain."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cold_compress_amd import _abi  # noqa: E402

dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def timed(fn, n, iters=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
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
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    ts.sort()
    return ts[len(ts) // 2]


def gemv(W, x, y, W3=None, delta=None, nw=None, h_out=None, freqs=None, rope_rows=0, hd=0, eps=1e-5):
    N, K = W.shape
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _abi.call("cc_gemv_fused", p(W), p(W3), p(x), p(delta), p(nw), eps, p(h_out), None, p(freqs), rope_rows, hd, p(y), N, K, 1, st)


def main():
    torch.manual_seed(0)
    shapes = {"wqkv": (6144, 4096), "wo": (4096, 4096), "w1": (14336, 4096), "w2": (4096, 14336), "lm_head": (128256, 4096)}
    only = sys.argv[1:]
    for name, (N, K) in shapes.items():
        if only and name not in only:
            continue
        nbytes = N * K * 2
        ncopy = max(2, (600 << 20) // nbytes + 1) if nbytes < (600 << 20) else 1
        if os.environ.get("CC_GEMV_HOT"):  # same weights every launch: Infinity-Cache-resident timing
            ncopy = 1
        Ws = [torch.randn(N, K, device=dev).mul_(0.02).to(torch.bfloat16) for _ in range(ncopy)]
        x = torch.randn(K, device=dev).to(torch.bfloat16)
        y = torch.empty(N, device=dev, dtype=torch.bfloat16)
        xr = x.view(1, 1, K)
        reps = max(ncopy, 8)
        t_lib = timed(lambda i: F.linear(xr, Ws[i % ncopy]), reps)
        t_own = timed(lambda i: gemv(Ws[i % ncopy], x, y), reps)
        gemv(Ws[0], x, y)
        ref = (Ws[0].float() @ x.float())
        err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
        res = {"shape": name, "N": N, "K": K, "MB": round(nbytes / 1e6, 1), "hipblaslt_us": round(t_lib, 2),
               "cc_gemv_us": round(t_own, 2), "hipblaslt_GBps": round(nbytes / t_lib / 1e3), "cc_gemv_GBps": round(nbytes / t_own / 1e3),
               "rel_err_vs_fp32": err}
        if name == "wqkv":  # as the decode loop launches it: norm(x + delta) prologue, RoPE epilogue on q and k
            delta = torch.randn(K, device=dev).to(torch.bfloat16)
            nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
            h = torch.empty(K, device=dev, dtype=torch.bfloat16)
            fr = torch.rand(64, 2, device=dev).to(torch.bfloat16)
            res["norm_only_us"] = round(timed(lambda i: gemv(Ws[i % ncopy], x, y, delta=delta, nw=nw, h_out=h), ncopy), 2)
            res["rope_only_us"] = round(timed(lambda i: gemv(Ws[i % ncopy], x, y, freqs=fr, rope_rows=5120, hd=128), ncopy), 2)
            res["norm_rope_us"] = round(timed(lambda i: gemv(Ws[i % ncopy], x, y, delta=delta, nw=nw, h_out=h, freqs=fr, rope_rows=5120,
                                                             hd=128), ncopy), 2)
        if name == "w1":  # SwiGLU pair in one pass
            W3s = [torch.randn(N, K, device=dev).mul_(0.02).to(torch.bfloat16) for _ in range(ncopy)]
            try:
                t_pair = timed(lambda i: gemv(Ws[i % ncopy], x, y, W3=W3s[i % ncopy]), ncopy)
                res["swiglu_pair_us"] = round(t_pair, 2)
                res["swiglu_pair_GBps"] = round(2 * nbytes / t_pair / 1e3)
            except Exception as e:  # a tuning override without a SwiGLU instantiation
                res["swiglu_pair_us"] = None
            del W3s
        print(json.dumps(res), flush=True)
        del Ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
