"""Caller glue (residual + RMSNorm, QKV split + RoPE + head layout, SwiGLU gate) as three fused HIP launches.

These are NOT part of the cache/attention hot path — they are the model-side code around it
(ref: model.py:317-327, 375-387, 442-443, 452-457, 507-519), which the reference leaves to ~45 eager elementwise
launches per layer or to torch.compile.  On device tensors they call the C ABI (`cc_add_rmsnorm`, `cc_qkv_rope`,
`cc_silu_mul`, `cc_gemv_fused`, `cc_softmax_argmax`); CPU tensors raise — there is no host path in the package (the
CPU model-wiring test brings its own eager twins: tests/host_glue.py).
"""
import ctypes as C

import torch

from .. import _abi

_DT = {torch.float32: _abi.CC_DT_F32, torch.bfloat16: _abi.CC_DT_BF16, torch.float16: _abi.CC_DT_F16}

def _host(t, what):
    raise _abi.ColdCompressError(f"{what} is on {t.device}: the HIP path needs ROCm device tensors (no CPU fallback).")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def add_rmsnorm(x, weight, eps, delta=None):
    """-> (h, normed) with h = x + delta (or x itself when delta is None)."""
    if not x.is_cuda:
        _host(x, "add_rmsnorm input")
    dim = x.shape[-1]
    xc = x.contiguous()
    T = xc.numel() // dim
    out = torch.empty_like(xc)
    h = xc
    dc = None
    if delta is not None:
        dc = delta.contiguous()
        h = torch.empty_like(xc)
    _abi.call("cc_add_rmsnorm", _p(xc), _p(dc), _p(weight), T, dim, float(eps), _DT[x.dtype], _p(h) if delta is not None else None,
              _p(out), _stream())
    return h, out


def qkv_rope(qkv, freqs_cis, n_head, n_local_heads, head_dim):
    """qkv [1, T, (HQ+2H)*D], freqs_cis [T, D/2, 2] -> q [1,HQ,T,D], k [1,H,T,D], v [1,H,T,D] (rotated, head-major)."""
    bsz, T, _ = qkv.shape
    HQ, H, D = n_head, n_local_heads, head_dim
    if not qkv.is_cuda:
        _host(qkv, "qkv_rope input")
    qc = qkv.contiguous()
    fc = freqs_cis.contiguous()
    q = torch.empty((1, HQ, T, D), dtype=qkv.dtype, device=qkv.device)
    k = torch.empty((1, H, T, D), dtype=qkv.dtype, device=qkv.device)
    v = torch.empty((1, H, T, D), dtype=qkv.dtype, device=qkv.device)
    _abi.call("cc_qkv_rope", _p(qc), _p(fc), T, HQ, H, D, _DT[qkv.dtype], _p(q), _p(k), _p(v), _stream())
    return q, k, v


def silu_mul(a, b):
    if not a.is_cuda:
        _host(a, "silu_mul input")
    ac, bc = a.contiguous(), b.contiguous()
    out = torch.empty_like(ac)
    _abi.call("cc_silu_mul", _p(ac), _p(bc), ac.numel(), _DT[a.dtype], _p(out), _stream())
    return out


def gemv_supported(*weights):
    """Single-token dense layers on the device path: the input vector must fit the kernel's LDS staging buffer."""
    return all(w.is_cuda and w.is_contiguous() and w.dtype in _DT and w.shape[1] * w.element_size() <= 64 * 1024
               and w.shape[1] % (16 // w.element_size()) == 0 for w in weights)


def gemv_fused(weight, x, w3=None, delta=None, norm_weight=None, eps=1e-5, h_out=None, bias=None, freqs=None, rope_rows=0,
               head_dim=0):
    """One decode-time dense layer with its glue fused (cc_gemv_fused): optional RMSNorm(x + delta) prologue (h_out
    receives x + delta), optional SwiGLU pairing with `w3`, optional RoPE epilogue on the first `rope_rows` rows.
    x: [K] (any shape with K elements); returns [N] in the model dtype."""
    N, K = weight.shape
    xc = x.contiguous()
    y = torch.empty((N,), dtype=weight.dtype, device=weight.device)
    _abi.call("cc_gemv_fused", _p(weight), _p(w3), _p(xc), _p(delta.contiguous() if delta is not None else None), _p(norm_weight),
              float(eps), _p(h_out), _p(bias), _p(freqs.contiguous() if freqs is not None else None), int(rope_rows), int(head_dim),
              _p(y), N, K, _DT[weight.dtype], _stream())
    return y


def softmax_argmax(logits):
    """probs = softmax(logits) in the model dtype and the greedy token (first index of the largest rounded probability),
    one launch (cc_softmax_argmax).  logits: [V]."""
    lc = logits.contiguous()
    probs = torch.empty_like(lc)
    idx = torch.empty((1,), dtype=torch.int32, device=lc.device)
    ws = _SM_WS.get(lc.device)
    if ws is None:
        ws = _SM_WS[lc.device] = torch.empty(int(_abi.lib()["cc_softmax_argmax_workspace_bytes"]()), dtype=torch.uint8, device=lc.device)
    _abi.call("cc_softmax_argmax", _p(lc), lc.numel(), _DT[lc.dtype], _p(probs), _p(idx), _p(ws), ws.numel(), _stream())
    return probs, idx


_SM_WS = {}
