"""debug: the l2 test's own flow (prefill -> compaction -> norms -> first decode steps), status and header words printed"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

DEV = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "late"
if mode == "early":
    from cold_compress_amd import _abi
    _abi.lib()
x = torch.zeros(4, device=DEV); torch.cuda.synchronize()
import cold_compress_amd.cache as cache  # noqa: E402
from cold_compress_amd import _abi  # noqa: E402
from cold_compress_amd.attention_utils import _decode_workspaces, single_launch_status  # noqa: E402
from cold_compress_amd.prompt_compression import get_prompt_compressor_constructor  # noqa: E402

L, S, H, R, D, g, w, dtype = 8192, 4096, 8, 4, 128, 4, 10, torch.bfloat16
HQ = H * R
gen = torch.Generator().manual_seed(99)
k = (torch.randn(1, H, L, D, generator=gen) * (0.5 + torch.rand(1, H, L, 1, generator=gen))).to(dtype)
v = torch.randn(1, H, L, D, generator=gen).to(dtype)
cls, rk = cache.get_cache_constructor("l2")
kw = dict(max_cache_length=S, global_tokens=g, max_seq_length=L + 2048, cache_bits=None, recent_window=w)
with torch.device(DEV):
    kv = cls(1, H, D, dtype, **{x: kw[x] for x in rk})
print("mode", mode, "l2h", _abi.lib()["cc_decode_step_l2_handoff"](), flush=True)
comp = get_prompt_compressor_constructor("l2")(head_specific=True, **{x: kw[x] for x in rk})
keep, kc, vc, _ = comp(torch.arange(L, device=DEV), k.to(DEV), v.to(DEV), attn=None)
kv.update_kv(keep, kc, vc, True)
kv.update_state(keep, kc, vc, True, None)
torch.cuda.synchronize()
for t in range(4):
    pt = torch.tensor([L + t], dtype=torch.int32)
    k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype); q1 = torch.randn(1, HQ, 1, D, generator=gen).to(dtype)
    t0 = time.time()
    y = kv.decode_step(q1.to(DEV), k1.to(DEV), k1.to(DEV), pt.to(DEV))
    torch.cuda.synchronize()
    ws = _decode_workspaces()[0]
    hdr = ws[:4096].view(torch.int32).cpu()
    print(f"step {t}: dt {time.time() - t0:.3f}s status {single_launch_status()} epochs {hdr[:8].tolist()} fail {hdr[64:72].tolist()}",
          "y zero heads:", [int((y[0, 4 * h:4 * h + 4].float().abs().sum() == 0)) for h in range(8)], flush=True)
