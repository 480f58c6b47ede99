"""debug: the l2 policy's single-launch step with the L2-resident hand-off, fresh caches, header words dumped on a failure"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch  # noqa: E402
from bench_policies import make  # noqa: E402
from cold_compress_amd import _abi  # noqa: E402
from cold_compress_amd.attention_utils import _decode_workspaces, reset_single_launch_status, single_launch_status  # noqa: E402

fns = _abi.lib()
print("probe", fns["cc_decode_step_probe_xcd"](), "l2h", fns["cc_decode_step_l2_handoff"](), flush=True)
pols = sys.argv[1:] or ["l2", "heavy_hitter", "l2"]
for rep, pol in enumerate(pols * 3):
    kv = make(pol, 8, 4096, 128)
    q = torch.randn(1, 32, 1, 128, device="cuda").to(torch.bfloat16)
    k1 = torch.randn(1, 8, 1, 128, device="cuda").to(torch.bfloat16)
    pos = torch.tensor([8192], dtype=torch.int32, device="cuda")
    for t in range(6):
        t0 = time.time()
        y = kv.decode_step(q, k1, k1, pos); pos += 1
        torch.cuda.synchronize()
        dt = time.time() - t0
        st = single_launch_status()
        if st or dt > 0.5:
            ws = _decode_workspaces()[0]
            hdr = ws[:4096].view(torch.int32).cpu()
            print(f"FAIL rep {rep} {pol} step {t}: status {st} dt {dt:.2f}s epochs {hdr[:8].tolist()} fail {hdr[64:72].tolist()} ticket {int(hdr[1022])}",
                  "y zero heads:", [int((y[0, 4 * h:4 * h + 4].float().abs().sum() == 0)) for h in range(8)], flush=True)
            reset_single_launch_status()
            break
    else:
        print(f"ok rep {rep} {pol}", flush=True)
