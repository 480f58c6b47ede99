"""`scaled_dot_product_attention` behind the reference's signature (ref: attention_utils.py:8-54), executed
by the HIP kernels in csrc/cc_attn_decode.hip (decode over the pruned cache) and csrc/cc_attn_prefill.hip
(causal prefill).

Differences from the reference that a caller can rely on:
  * GQA-aware: `key`/`value` may carry H kv heads with H | HQ; nothing needs `repeat_interleave`
    (model.py:399-400 materialises 4x K/V per step).  Pre-repeated inputs (H == HQ) work too.
  * `group_mean=True` returns the probabilities already averaged over each query group, [1, H, 1, S]
    (what model.py:413-418 computes next), instead of [1, HQ, 1, S].
  * Prefill with `return_attn=True` returns an `AttnSummary` (column sums + observation-window mean) instead
    of the [1, HQ, L, L] tensor; every policy in cache.py / prompt_compression.py consumes exactly that.
  * `history=(num, denom, counter)` folds the heavy-hitter history update into the decode combine pass;
    `history=(ring, denom, counter, W, acc, wsum)` does the same for the `history_window_size > 1` ring.
Unsupported (raised loudly): dropout_p != 0, non-causal prefill masks; attn_top_k < 1 asserts with a mask, like the reference.
"""
import ctypes as C
import math

import torch

from . import _abi
from ._abi import ColdCompressError
from .prompt_compression import AttnSummary

_DT = {torch.float32: _abi.CC_DT_F32, torch.bfloat16: _abi.CC_DT_BF16, torch.float16: _abi.CC_DT_F16}
_WS = {}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_RETIRED = []  # superseded scratch buffers stay alive: a captured hipGraph may still hold their addresses


def _workspace(nbytes, device, kind="decode"):
    """Scratch for the attention kernels, sized by the ABI's query: one zero-initialised buffer per (device, kind).
    Decode and prefill never share one: the decode buffer carries state across launches (the single-launch step's epoch
    words, include/coldcompress.h) and its address is baked into captured hipGraphs, so it is only ever replaced by a
    larger one — and the old one is kept alive, never handed back to the allocator."""
    key = (_norm_device(device), kind)  # ("cuda" and "cuda:<current>" are one device)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise ColdCompressError("attention scratch must be allocated before hipGraph capture (run one eager step first): "
                                    "a captured allocation would re-zero the single-launch step's epoch words on every replay")
        if ws is not None:
            _RETIRED.append((kind, ws))
        ws = torch.zeros(max(nbytes, 1), dtype=torch.uint8, device=device)
        _WS[key] = ws
        if kind == "decode" and torch.device(device).type == "cuda":  # (per device, cached: the L2-resident hand-off's eligibility)
            with torch.cuda.device(torch.device(device)):
                _abi.probe_device()
    return ws


def single_launch_status(device=None):
    """0, or 1 if a single-launch layer step ever failed to complete its in-launch hand-off (a launch whose workgroups were not
    all resident: its results are invalid) on `device` (default: every device) — read from the decode workspaces, the retired
    ones included (captured hipGraphs may still run on them).  Synchronises."""
    off = int(_abi.lib()["cc_decode_step_status_offset"]())
    bad = 0
    for ws in _decode_workspaces(device):
        if off + 4 <= ws.numel():
            bad |= int(ws[off:off + 4].view(torch.int32).item())
    return bad


def _norm_device(device):
    """torch.device('cuda') and 'cuda' name the CURRENT device: workspaces are keyed by the indexed form."""
    if device is None:
        return None
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def _decode_workspaces(device=None):
    want = _norm_device(device)
    return [ws for (dev, kind), ws in list(_WS.items()) + [((str(w.device), k), w) for k, w in _RETIRED]
            if kind == "decode" and (want is None or _norm_device(dev) == want)]


def reset_single_launch_status(device=None):
    """Clear the (sticky) hand-off timeout word of `device`'s decode workspaces (default: every device) — after the failure has
    been reported (check_single_launch_status) or the single-launch form has been switched off, so that later generations on the
    device are judged on their own.  Synchronises."""
    off = int(_abi.lib()["cc_decode_step_status_offset"]())
    if torch.cuda.is_available():
        torch.cuda.synchronize()  # the failed launch (and whatever returned at once behind it) has drained
    for ws in _decode_workspaces(device):
        if off + 4 <= ws.numel():
            ws[off:off + 4].zero_()
            # ... and every kv head's epoch word moves on: no granule the failed attempt left behind — published by a workgroup
            # that gave up, or by one that only started after that — can carry the tag of a later launch (the tags only grow)
            ws[0:128].view(torch.int32).add_(4)
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def set_device_single_launch(device, enabled):
    """The knob for a SHARED GPU (include/coldcompress.h, cc_decode_step_device_single_launch): enabled = False makes every decode step
    launched on `device` from now on take the two-launch forms — nothing in them waits for other workgroups, so a device this process
    shares with other processes or kernels cannot time them out; True restores the single-launch forms.  Per device: other devices of
    the process keep theirs.  Drop captured decode graphs (GraphedDecoder.graph = None) after flipping it: a captured step keeps the
    form it was captured with.  -> the previous setting."""
    dev = torch.device(device)
    with torch.cuda.device(dev):
        return bool(_abi.lib()["cc_decode_step_device_single_launch"](1 if enabled else 0))


def raise_single_launch_failure(device=None):
    """Clear the (sticky) status word — reported once; the next generation starts clean — and raise."""
    reset_single_launch_status(device)
    raise ColdCompressError(
        "a single-launch layer step did not complete its in-launch hand-off (its workgroups were not all resident, e.g. the "
        "GPU was shared with another kernel): the tokens produced since are invalid.  On a shared device switch the single-launch "
        "forms off for THAT device: cold_compress_amd.attention_utils.set_device_single_launch(device, False) "
        "(cc_decode_step_device_single_launch(0) in include/coldcompress.h).")


def check_single_launch_status(device=None):
    """Raise loudly if a single-launch step timed out (see single_launch_status)."""
    if single_launch_status(device):
        raise_single_launch_failure(device)


def _need_device(t, message):
    if not t.is_cuda:
        raise ColdCompressError(message)


def decode_attention(query, key, value, attn_mask=None, scale=None, return_attn=False, group_mean=False,
                     history=None):
    """query [1, HQ, 1, D]; key/value [1, H, S, D]; attn_mask bool [1, H or HQ, 1, S] or None."""
    _need_device(query, "decode attention needs ROCm device tensors (no CPU fallback)")
    _, HQ, _, D = query.shape
    _, H, S, _ = key.shape
    if HQ % H:
        raise ColdCompressError(f"query heads {HQ} not a multiple of kv heads {H}")
    dt = query.dtype
    if key.dtype != dt or value.dtype != dt:
        raise ColdCompressError("q/k/v dtypes differ")
    R = HQ // H
    q = query.reshape(HQ, D).contiguous()
    k = key if key.is_contiguous() else key.contiguous()
    v = value if value.is_contiguous() else value.contiguous()
    m = None
    if attn_mask is not None:
        m = attn_mask
        if m.shape[1] == HQ and R > 1:
            m = m[:, ::R]  # a caller that repeat_interleave'd the mask: every R-th row is the kv head's mask
        m = m.reshape(H, S).contiguous()
        if m.dtype != torch.bool:
            raise ColdCompressError("attn_mask must be bool")
    y = torch.empty((1, HQ, 1, D), dtype=dt, device=query.device)
    attn = probs = None
    if return_attn or history is not None:
        if group_mean:
            attn = torch.empty((1, H, 1, S), dtype=dt, device=query.device)
        else:
            probs = torch.empty((1, HQ, 1, S), dtype=dt, device=query.device)
    nbytes = _abi.lib()["cc_decode_attn_workspace_bytes"](HQ, H, S, D, _DT[dt])
    ws = _workspace(nbytes, query.device)
    sc = 1.0 / math.sqrt(D) if scale is None else scale
    hn = hd = hc = None
    if history is not None and len(history) == 6:  # W > 1 ring + tracked window sums folded into the combine pass
        ring, hd, hc, W, acc, wsum = history
        _abi.call("cc_decode_attn_gqa_ring", _ptr(q), _ptr(k), _ptr(v), _ptr(m), HQ, H, S, D, _DT[dt], sc, _ptr(y),
                  _ptr(attn) if return_attn else None, _ptr(ring), _ptr(hd), _ptr(hc), W, _ptr(acc), _ptr(wsum), _ptr(ws), ws.numel(), _stream())
        return y, attn if return_attn else None
    if history is not None:
        hn, hd, hc = history
    _abi.call("cc_decode_attn_gqa", _ptr(q), _ptr(k), _ptr(v), _ptr(m), HQ, H, S, D, _DT[dt], sc, _ptr(y), _ptr(attn),
              _ptr(probs), _ptr(hn), _ptr(hd), _ptr(hc), _ptr(ws), ws.numel(), _stream())
    return y, (attn if group_mean else probs) if return_attn else None


def prefill_attention(query, key, value, scale=None, return_attn=False, obs_len=16, bands=()):
    """Causal attention, query [1, HQ, L, D], key/value [1, H, L, D]; side outputs as an AttnSummary.
    `bands`: window widths (<= 4) whose band sums the hybrid cache's profiling score needs."""
    _need_device(query, "prefill attention needs ROCm device tensors (no CPU fallback)")
    _, HQ, L, D = query.shape
    H = key.shape[1]
    dt = query.dtype
    q, k, v = query.contiguous(), key.contiguous(), value.contiguous()
    y = torch.empty((1, HQ, L, D), dtype=dt, device=query.device)
    colsum = obs = None
    ol = min(obs_len, L)
    if return_attn:
        colsum = torch.empty((H, L), dtype=torch.float32, device=query.device)
        obs = torch.empty((H, L), dtype=torch.float32, device=query.device)
    nbytes = _abi.lib()["cc_prefill_attn_workspace_bytes"](HQ, H, L, D, _DT[dt])
    ws = _workspace(nbytes, query.device, "prefill")
    sc = 1.0 / math.sqrt(D) if scale is None else scale
    bands = [int(b) for b in bands] if return_attn else []
    if len(bands) > 4:
        raise ColdCompressError("at most 4 distinct window widths are supported per prefill")
    band_out = torch.empty((len(bands), H, L), dtype=torch.float32, device=query.device) if bands else None
    barr = (C.c_int32 * len(bands))(*bands) if bands else None
    _abi.call("cc_prefill_attn_bands", _ptr(q), _ptr(k), _ptr(v), HQ, H, L, D, _DT[dt], sc, _ptr(y), _ptr(colsum), _ptr(obs),
              ol, barr, len(bands), _ptr(band_out), _ptr(ws), ws.numel(), _stream())
    if not return_attn:
        return y, None
    return y, AttnSummary(colsum, obs, ol, dt, {b: band_out[i] for i, b in enumerate(bands)})


_CAUSAL_SEEN = {}  # id(tensor) -> (weak reference to it, its version, verdict of the full comparison)


def _is_causal_mask(attn_mask, L):
    """Is `attn_mask` the lower-triangular causal mask?  The caller model hands the SAME mask tensor to every layer of a forward
    pass (model.py:402-411: one slice of its precomputed causal_mask): the L x L comparison (268 MB of traffic at L = 16k) is made
    once per tensor OBJECT (weakly referenced, so a recycled address can never inherit a verdict; its version counter catches
    in-place edits) — after an O(L) look at the first and last rows has not already ruled it out."""
    import weakref

    if attn_mask is None or attn_mask.shape[-2:] != (L, L) or attn_mask.dtype != torch.bool:
        return False
    hit = _CAUSAL_SEEN.get(id(attn_mask))
    if hit is not None and hit[0]() is attn_mask and hit[1] == attn_mask._version:
        return hit[2]
    m = attn_mask.reshape(-1, L, L)
    ok = bool(m[:, -1, :].all()) and bool(m[:, 0, 0].all()) and (L == 1 or not bool(m[:, 0, 1:].any()))
    if ok:
        tril = torch.ones(L, L, dtype=torch.bool, device=attn_mask.device).tril_()
        ok = bool((m == tril).all())
    key = id(attn_mask)
    _CAUSAL_SEEN[key] = (weakref.ref(attn_mask, lambda _r, key=key: _CAUSAL_SEEN.pop(key, None)), attn_mask._version, ok)
    return ok


def _topk_decode_attention(query, key, value, attn_mask, scale, top_k):
    """ref: attention_utils.py:24-26, 40-50 — decode attention over the top-k logits only.  The reference asserts when
    a mask is present ("Top-k attention not supported with masks."), and its model always passes the cache mask at
    decode time (model.py:389-396), so this is reachable only by a direct, mask-less call with keys already
    expanded to the query heads — reproduced with the same behaviour: same assertion; otherwise softmax over the
    k largest logits (a membership mask fed to the decode kernel), probabilities returned in top-k order."""
    assert attn_mask is None, "Top-k attention not supported with masks."
    from .prompt_compression import topk_keep

    _, HQ, _, D = query.shape
    H, S = key.shape[1], key.shape[2]
    if H != HQ:
        raise ColdCompressError("top-k decode attention selects per query head: expand K/V to the query heads (reference call shape)")
    _, probs = decode_attention(query, key, value, None, scale, True, False, None)  # softmax is monotone: same top-k set
    p2 = probs.reshape(HQ, S)
    keep = topk_keep(p2, top_k)  # ascending indices, ties lowest-index-first
    member = torch.zeros((1, HQ, 1, S), dtype=torch.bool, device=query.device)
    member.view(HQ, S).scatter_(1, keep, True)
    y, p_sel = decode_attention(query, key, value, member, scale, True, False, None)
    vals = p_sel.reshape(HQ, S).gather(1, keep)
    order = torch.sort(vals.float(), dim=1, descending=True, stable=True).indices  # torch.topk returns descending values
    return y, vals.gather(1, order).view(1, HQ, 1, top_k)


def scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0, scale=None, return_attn=False,
                                 attn_top_k=1.0, group_mean=False, history=None, is_causal=None, bands=()):
    """ref: attention_utils.py:8-54 (same positional/keyword surface; extra keywords documented above)."""
    if dropout_p != 0.0:
        raise ColdCompressError("dropout is not part of the inference path")
    L, S = query.size(-2), key.size(-2)
    if L == 1:
        top_k = int(attn_top_k * S)
        if top_k != S:
            return _topk_decode_attention(query, key, value, attn_mask, scale, top_k)
        return decode_attention(query, key, value, attn_mask, scale, return_attn, group_mean, history)
    if is_causal is None:
        is_causal = _is_causal_mask(attn_mask, L)
    if not is_causal or L != S:
        raise ColdCompressError("prefill attention supports the causal [L, L] mask of generation_utils.py:153-158 only")
    return prefill_attention(query, key, value, scale, return_attn, bands=bands)
