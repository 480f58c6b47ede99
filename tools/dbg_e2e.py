"""Debug aid: per-layer/per-step eviction index comparison of an F1 fixture run on the GPU."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import test_gpu_e2e as T

name = sys.argv[1]
f, model, seq, log, logits = T._run(name)
print("seq equal:", torch.equal(seq, f["seq"]))
got = torch.stack(logits).cpu()
print("logit max err per step:", [(round(float(x), 6)) for x in (got - f["logits"]).abs().amax(dim=1)])
for li, layer in enumerate(model.layers):
    ref = f[f"evict_idx_L{li}"].long()
    mine = torch.stack(log[li]).cpu().view(ref.shape[0], -1)
    ref = ref.view(ref.shape[0], -1)
    bad = (mine != ref).any(dim=1).nonzero().view(-1).tolist()
    print(f"layer {li}: first mismatching steps {bad[:5]}")
    if bad:
        t = bad[0]
        print("  mine", mine[t].tolist(), "ref", ref[t].tolist())
    kv = layer.attention.kv_cache
    if hasattr(kv, "key_norm"):
        print("  keynorm max abs diff", float((kv.key_norm.cpu() - f[f"final_keynorm_L{li}"]).abs().max()))
    print("  pos equal", torch.equal(kv.pos.cpu(), f[f"final_pos_L{li}"]))
