#!/bin/bash
# r6 GPU call 24: ONE 16-wave workgroup per 256 rows (CC_V_NW16 build, cc_decode_step_set_wide(2)) against the 8-wave plan; y compared
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=cold_compress_amd/csrc/libcoldcompress_hip.so
cp $L /tmp/keep.so
cp .ab/libnw16.so $L
( timeout 300 python - <<'PY'
import sys, os, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
from bench_policies import make
from cold_compress_amd import _abi
fns = _abi.lib(); _abi.probe_device()
H, HQ, S, D = 8, 32, 4096, 128
torch.manual_seed(3)
outs = {}
for wide in (1, 2):
    fns["cc_decode_step_set_wide"](wide)
    torch.manual_seed(3)
    kv = make("heavy_hitter", H, S, D)
    g = torch.Generator(device="cuda").manual_seed(5)
    ys = []
    for t in range(6):
        q = torch.randn(1, HQ, 1, D, device="cuda", generator=g).to(torch.bfloat16)
        k1 = torch.randn(1, H, 1, D, device="cuda", generator=g).to(torch.bfloat16)
        pos = torch.tensor([S + 100 + t], dtype=torch.int32, device="cuda")
        ys.append(kv.decode_step(q, k1, k1, pos).float().clone())
    torch.cuda.synchronize()
    outs[wide] = (ys, kv.pos.clone(), kv.attn_history_num.clone(), kv.step_status(HQ))
a, b = outs[1], outs[2]
print("status", a[3], b[3], "max |dy|", max(float((x - y).abs().max()) for x, y in zip(a[0], b[0])), "pos equal", bool(torch.equal(a[1], b[1])),
      "history max diff", float((a[2] - b[2]).abs().max()))
PY
) > gpurun_out/r6_c24_nw16.txt 2>&1
for r in 1 2 3; do for w in 1 2; do echo -n "wide=$w "; CC_STEP_WIDE=$w timeout 200 python tools/ab_step.py heavy_hitter 8:32:4096 8:32:2560 8:32:8192 2>/dev/null || echo FAILED; done; done >> gpurun_out/r6_c24_nw16.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r6_c24_nw16.txt | cut -c1-200
