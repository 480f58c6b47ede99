#!/usr/bin/env python3
"""The one-shot all-reduce (cc_allreduce_*, include/coldcompress.h) against the rank-ordered fp32 sum computed on the host:
hundreds of messages of 1 .. 16 KiB in three dtypes, eagerly and replayed from a hipGraph.

  --one_gpu : every rank drives cuda:0 (what a 1-GPU box can run: the IPC mapping, the remote stores, the flags, the epochs
              and the two alternating slot sets are exercised for real; the stores just do not cross xGMI)
  default   : one GPU per rank
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 tools/allreduce_check.py --one_gpu"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cold_compress_amd import tp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one_gpu", action="store_true")
    ap.add_argument("--iters", type=int, default=150)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0 if a.one_gpu else rank)
    dev = torch.device("cuda", 0 if a.one_gpu else rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ar = tp.OneShotAllReduce(max_bytes=32 * 1024)
    ok = True

    def expected(x):  # rank-ordered fp32 sum of everybody's vector, rounded once
        parts = [torch.empty_like(x, dtype=torch.float32, device="cpu") for _ in range(world)]
        dist.all_gather(parts, x.float().cpu())
        acc = parts[0].clone()
        for r in range(1, world):
            acc += parts[r]
        return acc.to(x.dtype)

    g = torch.Generator().manual_seed(1000 + rank)
    for it in range(a.iters):
        dtype = (torch.bfloat16, torch.float32, torch.float16)[it % 3]
        n = (8, 4096, 8192, 1000, 3, 16384 // (4 if dtype == torch.float32 else 2))[it % 6]
        x = torch.randn(n, generator=g).to(dtype).to(dev)
        want = expected(x)
        ar.all_reduce(x)
        torch.cuda.synchronize()
        if not torch.equal(x.cpu(), want):
            ok = False
            print(f"rank {rank}: mismatch at iteration {it} ({dtype}, n={n}): max |d| = {(x.cpu().float() - want.float()).abs().max()}", flush=True)
            break
    # ---- hipGraph: four all-reduces per replay on static buffers, twenty replays
    bufs = [torch.zeros(4096, dtype=torch.bfloat16, device=dev) for _ in range(4)]
    srcs = [torch.randn(4096, generator=g).to(torch.bfloat16).to(dev) for _ in range(4)]
    wants = [expected(s_) for s_ in srcs]
    for b, s_ in zip(bufs, srcs):
        b.copy_(s_)
        ar.all_reduce(b)  # eager warm-up on the side stream's state
    torch.cuda.synchronize()
    dist.barrier()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for b, s_ in zip(bufs, srcs):
            b.copy_(s_)
            ar.all_reduce(b)
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    for b, w_ in zip(bufs, wants):
        ok = ok and torch.equal(b.cpu(), w_)
    # ---- timing (what profiles/r04_scale_projection.json charges per all-reduce): sixteen 8 KiB all-reduces per replay, nothing else
    tg = torch.cuda.CUDAGraph()
    tb = torch.zeros(4096, dtype=torch.bfloat16, device=dev)
    ar.all_reduce(tb)
    torch.cuda.synchronize()
    dist.barrier()
    with torch.cuda.graph(tg):
        for _ in range(16):
            ar.all_reduce(tb)
    tg.replay()
    torch.cuda.synchronize()
    dist.barrier()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tg.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 16)
    ts.sort()
    if rank == 0:
        import json
        print("ONESHOT_TIMING " + json.dumps({"world": world, "one_gpu": bool(a.one_gpu), "bytes": 8192, "us_per_allreduce_median": round(ts[5], 2),
                                              "us_min": round(ts[0], 2), "note": "hipGraph replay of 16 back-to-back cc_allreduce_sum of 4096 bf16; "
                                              "ranks sharing ONE GPU when one_gpu (no xGMI hop: launch + flag round trip through memory only)"}), flush=True)
    ok = ok and ar.status() == 0
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"world {world} one_gpu {a.one_gpu}: ONESHOT ALLREDUCE CHECK {'OK' if int(flag) else 'FAIL'}", flush=True)
    dist.barrier()
    ar.close()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) else 1)


if __name__ == "__main__":
    main()
