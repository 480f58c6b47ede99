import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch, numpy as np
from cold_compress_amd import _abi
from cold_compress_amd.attention_utils import _WS
import test_gpu_quant_fused as T
DEV="cuda"
wide=int(os.environ.get("CC_STEP_WIDE","1"))
_abi.lib()["cc_decode_step_set_wide"](wide)
_abi.lib()["cc_decode_step_set_single_launch"](0)
strategy, dtype, H, HQ, S, Tn, D = "recent_global", torch.bfloat16, 8, 32, 4096, 4090, 128
a, b = T._mk(strategy, H, S, D, dtype, False), T._mk(strategy, H, S, D, dtype, True)
gen = torch.Generator().manual_seed(31)
k0 = torch.randn(1, H, Tn, D, generator=gen).to(dtype).to(DEV)
v0 = (2.0 * torch.randn(1, H, Tn, D, generator=gen)).to(dtype).to(DEV)
for kv in (a, b):
    kv.update_kv(torch.arange(Tn, device=DEV), k0, v0, True)
kd, vd = b.dequantized_kv()
a.k_cache.copy_(kd); a.v_cache.copy_(vd)
one_bytes = 4096 + 32*(64*8*16+64*16) + 32*8*64*64*16
for t in range(4):
    p = torch.tensor([Tn + 5 + t], dtype=torch.int32, device=DEV)
    k1 = torch.randn(1, H, 1, D, generator=gen).to(dtype).to(DEV)
    v1 = (2.0 * torch.randn(1, H, 1, D, generator=gen)).to(dtype).to(DEV)
    q = torch.randn(1, HQ, 1, D, generator=gen).to(dtype).to(DEV)
    kh = T._round_trip_rows(k1.reshape(H, D)).view(1, H, 1, D)
    vh = T._round_trip_rows(v1.reshape(H, D)).view(1, H, 1, D)
    ya = a.decode_step(q, kh, vh, p); torch.cuda.synchronize()
    ws = [w for (d, kind), w in _WS.items() if kind == "decode"][0]
    wa = ws[one_bytes:one_bytes + (4 << 20)].clone()
    yb = b.decode_step(q, k1, v1, p); torch.cuda.synchronize()
    wb = ws[one_bytes:one_bytes + (4 << 20)].clone()
    ns = 32 if wide else 64
    sc_bytes = HQ * S * 2
    off_ml = (sc_bytes + 255) // 256 * 256
    ml_bytes = HQ * ns * 2 * 4
    off_o = off_ml + (ml_bytes + 255) // 256 * 256
    sa, sb = wa[:sc_bytes].view(torch.bfloat16).view(HQ, S).float(), wb[:sc_bytes].view(torch.bfloat16).view(HQ, S).float()
    mla, mlb = wa[off_ml:off_ml + ml_bytes].view(torch.float32).view(HQ, ns, 2), wb[off_ml:off_ml + ml_bytes].view(torch.float32).view(HQ, ns, 2)
    oa, ob = wa[off_o:off_o + HQ*ns*D*4].view(torch.float32).view(HQ, ns, D), wb[off_o:off_o + HQ*ns*D*4].view(torch.float32).view(HQ, ns, D)
    dsc = (sa != sb) & ~(sa.isnan() & sb.isnan())
    print(t, "y equal", torch.equal(ya, yb), "scores differ", int(dsc.sum()), dsc.nonzero()[:5].tolist(),
          "ml differ", int((mla != mlb).sum()), (mla != mlb).nonzero()[:6].tolist(), "o differ", int((oa != ob).sum()), (oa != ob).any(-1).nonzero()[:6].tolist())
    if int((mla != mlb).sum()):
        i = (mla != mlb).nonzero()[0].tolist()
        print("   ml a", mla[i[0], i[1]].tolist(), "b", mlb[i[0], i[1]].tolist())
