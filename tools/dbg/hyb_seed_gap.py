"""Diagnostic: per-head discrepancy between the device's and the oracle's hybrid ring seeds (column means) on the synthetic heads."""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hybrid_inputs import make_inputs
import hybrid_profile_ref as hp
from oracle import oracle_lib as o
from helpers import to_np
import cold_compress_amd.cache as cache
from cold_compress_amd.attention_utils import prefill_attention
o.set_threads(16)
HYBRID = [{"strategy": "window", "recent_window": 0.1},
          {"strategy": "window_heavy_hitter", "heavy_hitter_frac": 0.25, "recent_window": 0.1},
          {"strategy": "window_heavy_hitter", "heavy_hitter_frac": 0.5, "recent_window": 0.1},
          {"strategy": "full"}]
L, S, H, R, D, g, frac = 3000, 3072, 6, 4, 128, 4, 0.97
dtype = torch.bfloat16
q, k, v = make_inputs(L, H, R, D, 3, dtype)
cls, rk = cache.get_cache_constructor("hybrid")
kw = dict(max_cache_length=S, max_seq_length=S, cache_bits=None, global_tokens=g, token_ids={"special": [], "punctuation": []},
          min_recovery_frac=frac, hybrid_strategies=HYBRID)
with torch.device("cuda"):
    kv = cls(1, H, D, dtype, **{x: kw[x] for x in rk})
pos0 = torch.arange(L, device="cuda")
kd, vd = k.unsqueeze(0).cuda(), v.unsqueeze(0).cuda()
_, summ = prefill_attention(q.unsqueeze(0).cuda(), kd, vd, return_attn=True, bands=kv.attn_bands(L))
mean_d = summ.column_mean(pos0)[0].float().cpu().numpy()
cs_d = summ.colsum.cpu().numpy()
A = np.zeros((H, L, L), np.float32); yo = np.zeros((H * R, L, D), np.uint16)
o.prefill_attn_matrix(to_np(q), to_np(k), to_np(v), H * R, H, L, D, 1, 1 / math.sqrt(D), yo, A)
cs_o = A.sum(axis=1, dtype=np.float64)
ref = hp.profile(A, HYBRID, g, frac, S, "bfloat16")
mean_o = ref["cum_attn"]
for h in range(H):
    rel = np.abs(mean_d[h] - mean_o[h]) / np.maximum(np.abs(mean_o[h]), 1e-30)
    relcs = np.abs(cs_d[h] - cs_o[h]) / np.maximum(np.abs(cs_o[h]), 1e-30)
    print(h, "mean: max rel %.4g  frac>0 %.4f  frac>2^-7 %.4f  frac>2^-6 %.4f | colsum f32: max rel %.3g  median %.3g" % (
        rel.max(), (rel > 0).mean(), (rel > 2 ** -7).mean(), (rel > 2 ** -6).mean(), relcs.max(), np.median(relcs)), "argmax", int(rel.argmax()))
