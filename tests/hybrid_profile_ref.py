"""Test-side restatement (numpy, row-chunked) of the reference's hybrid profiling — KVCacheHybrid.build_masks and
profile_attn_heads, cache.py:1066-1187, with create_window_attention_mask, cache.py:142-149 — evaluated the way the
reference DEFINES it: per query row over the materialised [H, L, L] group-averaged attention, NOT through the column-sum /
band-sum identities the product uses.  TEST INFRASTRUCTURE: the full-size GPU test compares the product's chosen policies,
counts and kept position sets against this.

Rounding points of the reference in a 16-bit model dtype are reproduced: cum_attn = dtype(dtype(sum over queries) /
(L - pos)); compressed score = dtype(mean over queries of dtype(sum over kept keys)); the comparison with
min_recovery_frac happens in the model dtype (a Python scalar does not promote a tensor).  Top-k ties are broken
lowest-index-first (the product's documented rule; torch's is implementation-defined).
"""
import math

import numpy as np


def _rnd(x, dtype_name):
    """float32 array -> rounded to the model dtype (round-to-nearest-even), still float32."""
    x = np.asarray(x, np.float32)
    if dtype_name == "float32":
        return x
    if dtype_name == "bfloat16":
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    return x.astype(np.float16).astype(np.float32)


def static_columns(name, L, g, special_mask, punc_mask):
    col = np.arange(L) < g  # cache.py:1076 — every policy keeps the global tokens
    if "special" in name:
        col = col | special_mask
    if "punc" in name:
        col = col | punc_mask
    return col


def column_sets(strategies, cum_attn, L, g, special_mask, punc_mask, total_len):
    """Per policy: (static column set incl. heavy hitters [H, L] bool, window width or 0) — the column part of
    build_masks for `total_len` (cache.py:1066-1136)."""
    H = cum_attn.shape[0]
    out = []
    for s in strategies:
        name = s["strategy"]
        col = static_columns(name, L, g, special_mask, punc_mask)
        win = 0
        last_row = col.copy()
        if "window" in name:
            win = max(1, int(s["recent_window"] * total_len))
            last_row[max(0, L - win):] = True  # row L - 1 of create_window_attention_mask
        cols = np.broadcast_to(col, (H, L)).copy()
        if "heavy_hitter" in name:
            avail = np.where(~last_row)[0]  # cache.py:1104: the LAST query row decides what is still available
            num_hh = math.ceil(min(s["heavy_hitter_frac"] * total_len, len(avail)))
            for h in range(H):
                order = np.argsort(-cum_attn[h, avail], kind="stable")  # largest first, ties lowest index first
                cols[h, avail[order[:num_hh]]] = True
        if name == "full":
            cols[:] = True
        out.append((cols, win))
    return out


def profile(attn, strategies, g, min_recovery_frac, S, dtype_name, special_mask=None, punc_mask=None, chunk=512):
    """attn: float32 [H, L, L] (values already rounded to the model dtype, zero above the diagonal).
    -> dict(cum_attn [H, L], scores [n_pol, H], strategies [H], mask_optimal [H, L], hh_threshold info)."""
    H, L, _ = attn.shape
    z = np.zeros(L, bool)
    sm = special_mask if special_mask is not None else z
    pm = punc_mask if punc_mask is not None else z
    # cache.py:1155: attn.squeeze(0).sum(dim=1) / (seq_len - input_pos)
    colsum = np.zeros((H, L), np.float64)
    for i0 in range(0, L, chunk):
        colsum += attn[:, i0:i0 + chunk, :].sum(axis=1, dtype=np.float64)
    cum = _rnd(_rnd(colsum.astype(np.float32), dtype_name) / (L - np.arange(L, dtype=np.float32)), dtype_name)
    scoring = column_sets(strategies, cum, L, g, sm, pm, L)
    scores = np.zeros((len(strategies), H), np.float32)
    for p, (cols, win) in enumerate(scoring):
        rowsum_total = np.zeros(H, np.float64)
        for i0 in range(0, L, chunk):
            i1 = min(L, i0 + chunk)
            rows = np.arange(i0, i1)[:, None]
            keys = np.arange(L)[None, :]
            wmask = (keys <= rows) & (keys > rows - win) if win else np.zeros((i1 - i0, L), bool)
            for h in range(H):
                m = wmask | cols[h][None, :]
                rs = (attn[h, i0:i1, :] * m).sum(axis=1, dtype=np.float64).astype(np.float32)
                rowsum_total[h] += _rnd(rs, dtype_name).sum(dtype=np.float64)  # .sum(dim=-1) -> dtype, then the mean's fp32 sum
        scores[p] = _rnd((rowsum_total / L).astype(np.float32), dtype_name)
    thr = _rnd(np.float32(min_recovery_frac), dtype_name)
    ok = scores >= thr
    strat = np.where(ok.any(axis=0), ok.argmax(axis=0), 0)  # first policy that reaches the threshold; all-false -> 0
    filling = column_sets(strategies, cum, L, g, sm, pm, S)
    mask_optimal = np.zeros((H, L), bool)
    for h in range(H):
        cols, win = filling[strat[h]]
        row = cols[h].copy()
        if win:
            row[max(0, L - win):] = True
        mask_optimal[h] = row
    return dict(cum_attn=cum, scores=scores, strategies=strat, mask_optimal=mask_optimal, threshold=float(thr), filling=filling)
