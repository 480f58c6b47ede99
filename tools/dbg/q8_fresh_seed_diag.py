"""What differs when tests/test_gpu_quant.py::test_e2e_cache_bits_8 runs on a reference-made fixture from another seed
(CC_GOLDEN_DIR=... python tools/dbg/q8_fresh_seed_diag.py): tokens, logits, 8-bit images (how many codes, by how much), scales, pos,
denominators — one JSON line.  The 8-bit image of a K/V row is round(x / scale): a row value that sits on a rounding boundary flips
its code on a 1e-6 difference between this build's fp32 GEMM / RoPE and the reference's."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden  # noqa: E402
from test_gpu_e2e import _build  # noqa: E402

from cold_compress_amd.harness import decode_one_token, generate, prefill  # noqa: E402

f = load_golden("f9_e2e_heavy_hitter_q8.npz")
model, ck = _build(f, f["n_layer"])
logits = []
orig = model.forward


def fwd(*a, **k):
    out = orig(*a, **k)
    logits.append(out[0, -1].detach().float().clone())
    return out


model.forward = fwd
seq, _, _ = generate(model, f["prompt"].to("cuda"), prefill, decode_one_token, max_new_tokens=f["new_tokens"])
torch.cuda.synchronize()
out = {"tokens_equal": bool(torch.equal(seq.cpu(), f["seq"]))}
if not out["tokens_equal"]:
    d = (seq.cpu() != f["seq"]).nonzero().flatten().tolist()
    out["first_token_diff_at"] = d[0]
    out["prompt_len"] = int(f["prompt_len"])
lg = torch.stack(logits).cpu()
n = min(lg.shape[0], f["logits"].shape[0])
dl = (lg[:n] - f["logits"][:n]).abs().amax(dim=-1)
out["logits_maxdiff_per_step"] = [round(float(x), 6) for x in dl]
top2 = f["logits"].float().topk(2, dim=-1).values
out["ref_top2_margin_per_step"] = [round(float(a - b), 6) for a, b in top2]
for li, layer in enumerate(model.layers):
    kv = layer.attention.kv_cache
    kv.quantize_cache()
    for nm, mine in (("k", kv.k_cache_q), ("v", kv.v_cache_q)):
        a = mine.cpu().view(torch.uint8).to(torch.int16)
        b = f[f"final_{nm}_L{li}"].view(torch.uint8).to(torch.int16)
        dd = (a - b).abs()
        out[f"L{li}_{nm}_codes_differ"] = int((dd > 0).sum())
        out[f"L{li}_{nm}_max_code_diff"] = int(dd.max())
        if int((dd > 0).sum()):
            idx = (dd > 0).nonzero()[:4].tolist()
            out[f"L{li}_{nm}_where"] = idx
    out[f"L{li}_k_scales_maxrel"] = float(((kv.k_scales.cpu() - f[f"final_k_scales_L{li}"]).abs() / f[f"final_k_scales_L{li}"].abs().clamp_min(1e-12)).max())
    out[f"L{li}_pos_equal"] = bool(torch.equal(kv.pos.cpu(), f[f"final_pos_L{li}"]))
    out[f"L{li}_denom_equal"] = bool(torch.equal(kv.attn_history_denom.cpu(), f[f"final_denom_L{li}"]))
print(json.dumps(out))
