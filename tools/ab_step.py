#!/usr/bin/env python3
"""Fused layer-step time of one policy at a list of shapes, for same-box A/B runs of library variants / geometry switches.

    CC_STEP_WIDE=0|1 python tools/ab_step.py [heavy_hitter] [H:HQ:S ...]

Prints one JSON line: {"lib": ..., "wide": ..., "policy": ..., "us": {"8:32:4096": 10.1, ...}} (median of 15 graph replays over
rotating caches, as tools/bench_policies.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench_policies import make, timed  # noqa: E402
from cold_compress_amd import _abi  # noqa: E402

DEFAULT = ["8:32:4096", "8:32:2560", "4:16:4096", "2:8:4096", "1:4:4096", "1:8:3488"]


def main():
    args = sys.argv[1:]
    policy = args.pop(0) if args and ":" not in args[0] else "heavy_hitter"
    shapes = args or DEFAULT
    wide = int(os.environ.get("CC_STEP_WIDE", "1"))
    _abi.lib()["cc_decode_step_set_wide"](wide)
    out = {}
    D = 128
    for sh in shapes:
        H, HQ, S = (int(x) for x in sh.split(":"))
        n_buf = max(4, min(32, (600 << 20) // (2 * H * S * D * 2) + 1))
        n_buf = int(os.environ.get("CC_AB_NBUF", n_buf))  # (a small rotation keeps the caches in the 256 MB Infinity Cache: how the step would run on MALL-resident K/V)
        caches = [make(policy, H, S, D) for _ in range(n_buf)]
        q = torch.randn(1, HQ, 1, D, device="cuda").to(torch.bfloat16)
        k1 = torch.randn(1, H, 1, D, device="cuda").to(torch.bfloat16)
        pos = torch.tensor([S + 100], dtype=torch.int32, device="cuda")
        for kv in caches:
            kv.prepare_decode(pos)
        extra = {"input_ids": torch.tensor([[11]], device="cuda")} if policy == "hybrid" else {}
        for kv in caches:
            kv.decode_step(q, k1, k1, pos, **extra)
        n_nodes = max(n_buf, int(os.environ.get("CC_AB_NODES", n_buf)))  # graph nodes per replay (cycling over the n_buf caches)
        def step(i):
            caches[i % n_buf].decode_step(q, k1, k1, pos, **extra)
            # positions ADVANCE, one per replay (a tiny add kernel per n_nodes steps): at a constant position the recoverable
            # heavy-hitter step finds step_commit[h] == *input_pos from the second replay on and REPLAYS — attention only, no
            # insert, no history / key stores — which is not the step (r3: every heavy-hitter A/B between the recoverable
            # hand-off and this fix timed the replay path, ~0.4 us short of the real step)
            if i == n_nodes - 1 and os.environ.get("CC_AB_CONST_POS") != "1":
                pos.add_(1)

        out[sh] = round(timed(step, n_nodes, iters=15), 2)
        del caches
        torch.cuda.empty_cache()
    from cold_compress_amd.attention_utils import single_launch_status

    st = single_launch_status(torch.device("cuda", torch.cuda.current_device()))
    rec = {"wide": wide, "policy": policy, "us": out}
    if st:  # a hand-off timed out: every later launch was a no-op and the times above mean nothing
        rec["HANDOFF_TIMEOUT_STATUS"] = int(st)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
