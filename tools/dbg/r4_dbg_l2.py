import torch, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench_policies import make
from cold_compress_amd import _abi
from cold_compress_amd.attention_utils import single_launch_status, reset_single_launch_status
fns = _abi.lib()
print("probe", fns["cc_decode_step_probe_xcd"](), "l2h", fns["cc_decode_step_l2_handoff"]())
for pol in ("heavy_hitter", "recent_global", "l2", "random"):
    for l2h in (1, 0):
        fns["cc_decode_step_set_l2_handoff"](l2h)
        kv = make(pol, 8, 4096, 128)
        q = torch.randn(1, 32, 1, 128, device="cuda").to(torch.bfloat16)
        k1 = torch.randn(1, 8, 1, 128, device="cuda").to(torch.bfloat16)
        pos = torch.tensor([4196], dtype=torch.int32, device="cuda")
        ys = []
        for t in range(3):
            y = kv.decode_step(q, k1, k1, pos); pos += 1
            torch.cuda.synchronize()
            ys.append(float(y.float().abs().sum()))
        st = single_launch_status()
        print(pol, "l2_handoff", l2h, "status", st, "ysum", ys, flush=True)
        if st: reset_single_launch_status()
