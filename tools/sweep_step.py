#!/usr/bin/env python3
"""The single-launch (where the shape allows: else two-launch) heavy-hitter layer step across cache lengths — the honest statement
of "what fraction of the HBM roofline from which S up" (VERDICT r3 item 1).

    python tools/sweep_step.py [S ...]  ->  one JSON line per S: B_step (2 H S D 2 + 29 H S), us per step (median of
    15 hipGraph replays over rotating caches > 512 MiB, positions advancing), GB/s, fraction of 8 TB/s, launches per step.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench_policies import make, timed  # noqa: E402
from cold_compress_amd import _abi  # noqa: E402

H, HQ, D = 8, 32, 128


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1024, 2560, 4096, 8192, 18432, 32768, 65536]
    fns = _abi.lib()
    for S in sizes:
        n_buf = max(4, min(36, (600 << 20) // (2 * H * S * D * 2) + 1))
        caches = [make("heavy_hitter", H, S, D) for _ in range(n_buf)]
        q = torch.randn(1, HQ, 1, D, device="cuda").to(torch.bfloat16)
        k1 = torch.randn(1, H, 1, D, device="cuda").to(torch.bfloat16)
        pos = torch.tensor([S + 100], dtype=torch.int32, device="cuda")
        for kv in caches:
            kv.prepare_decode(pos)
            kv.decode_step(q, k1, k1, pos)

        def step(i):
            caches[i % n_buf].decode_step(q, k1, k1, pos)
            if i == n_buf - 1:
                pos.add_(1)

        us = timed(step, n_buf, iters=15)
        b_step = 2 * H * S * D * 2 + 29 * H * S
        one = int(fns["cc_decode_step_single_launch"](HQ, H, S, D, 1))
        print(json.dumps({"policy": "heavy_hitter", "H": H, "HQ": HQ, "S": S, "D": D, "dtype": "bf16", "B_step_bytes": b_step,
                          "us_per_step": round(us, 2), "GBps": round(b_step / us / 1e3, 1), "frac_of_8TBps": round(b_step / us / 1e3 / 8000.0, 4),
                          "launches_per_step": 1 if one else 2, "l2_resident_handoff": int(fns["cc_decode_step_l2_handoff"]()) if one else 0,
                          "rotating_caches": n_buf}), flush=True)
        del caches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
