#!/usr/bin/env python3
"""Few-head ranks (tp.py:151-154: H = 1-2 per rank): what a launch that STREAMS the rank's cache costs at finer geometries — the
step's own (4 waves x 64 rows: 64 workgroups at S = 4096) against 128 x 2 waves and 256 x 1 wave (one 16-row tile per workgroup)
— next to the step itself.  The streaming-only launches bound what a 256-workgroup step could gain.

    python tools/geometry_h1.py  ->  JSON lines (profiles/r04_step_geometry_H1.jsonl)
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench_policies import make, timed  # noqa: E402
from cold_compress_amd import _abi  # noqa: E402


def main():
    fns = _abi.lib()
    D = 128
    for H, HQ, S in ((1, 4, 4096), (1, 8, 3488), (2, 8, 4096)):
        n_buf = 36
        caches = [make("heavy_hitter", H, S, D) for _ in range(n_buf)]
        scratch = torch.zeros(64, dtype=torch.int32, device="cuda")
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

        us_step = timed(step, n_buf, iters=15)
        rows = []
        for waves, rpw in ((8, 128), (4, 64), (2, 32), (1, 16)):
            def floor(i, waves=waves, rpw=rpw):
                rc = fns["cc_decode_step_stream_floor_geom"](caches[i % n_buf]._view(), waves, rpw, C.c_void_p(scratch.data_ptr()),
                                                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            us = timed(floor, n_buf, iters=15)
            rows.append({"waves_per_workgroup": waves, "rows_per_workgroup": rpw, "workgroups": H * ((S + rpw - 1) // rpw), "stream_only_launch_us": round(us, 2)})
        base = next(r for r in rows if r["waves_per_workgroup"] == 4)["stream_only_launch_us"]
        best = min(r["stream_only_launch_us"] for r in rows)
        print(json.dumps({"H": H, "HQ": HQ, "S": S, "kv_bytes": 2 * H * S * D * 2, "single_launch_step_us": round(us_step, 2),
                          "stream_only_launches": rows,
                          "most_a_finer_geometry_can_take_off_the_streaming_part_us": round(base - best, 2),
                          "note": "stream-only = the step's K/V loads and nothing else (cc_decode_step_stream_floor_geom), hipGraph of 36 launches over "
                                  "rotating caches, launch boundary included; the step's hand-off tail GROWS with the workgroup count (every gatherer "
                                  "folds n_split pairs and granules), so the streaming difference is an upper bound of the gain"}), flush=True)
        del caches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
