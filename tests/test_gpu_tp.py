"""The tensor-parallel product path on real kernels (ref: tp.py:41-56, 124-176): cache + attention under TP = 2 against the
unsharded model on the same weights — tokens equal, layer-0 evictions identical (tools/tp2_check.py is the worker).

  * with two GPUs: one rank per GPU, RCCL all-reduces over xGMI, once with eager launches and once with the sharded decode
    step (collectives included) replayed from a hipGraph;
  * on a single GPU: both ranks on cuda:0 with the collectives staged over the host (gloo) — no RCCL, but every kernel of
    the sharded decode path runs for real.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, world=2, timeout=600, script="tp2_check.py", marker=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", script)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    assert (marker or f"TP{world} CHECK OK") in r.stdout, r.stdout[-3000:]
    return r.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per GPU")
@pytest.mark.parametrize("graph", [False, True])
def test_tp2_rccl_matches_unsharded(graph):
    out = _launch(["--backend", "nccl"] + (["--graph"] if graph else []))
    assert "backend nccl world 2" in out


@pytest.mark.parametrize("world", [8] + ([2, 4] if os.environ.get("CC_LONG_TESTS") == "1" else []))
def test_tp_one_gpu_staged_collectives_matches_unsharded(world):
    """The PRODUCT cache + attention kernels under TP = 8 (and 2 / 4 with CC_LONG_TESTS=1: 30 s of process start-up each; world 2 runs
    by default in test_tp2_one_gpu_oneshot_allreduce_in_hipgraph, all three ran green on the final r5 build) of the 8B
    shape — 4 / 1 (/ 2) kv heads per rank, the shapes the ranks of the 2 / 8 (/ 4)-GPU points run (H = 1: the few-head form of the
    single-launch step) — every rank on cuda:0, collectives staged over the host (r5: world 4 and 8 ran green; RCCL over xGMI itself
    still needs a multi-GPU box)."""
    out = _launch(["--backend", "gloo"], world=world, timeout=1200)
    assert f"backend gloo world {world}" in out


def test_tp8_c5_rank_shape_one_gpu():
    """BASELINE C5's per-rank shapes end to end (one layer of the Llama-3-70B shape, a 4096-entry vocabulary, under TP = 8: one kv head and 8 query heads per
    rank, model dim 8192, cache 3488 after a 3600-token prompt) — product kernels on every rank, all on cuda:0, collectives staged over
    the host: layer-0 evictions identical to the unsharded model's on every rank, tokens equal up to counted arg-max near-ties."""
    out = _launch(["--backend", "gloo", "--model", "Llama-3-70B-shape", "--layers", "1", "--vocab", "4096", "--cache", "3488", "--prompt", "3600"], world=8, timeout=1500)
    assert "backend gloo world 8" in out


def test_tp2_one_gpu_oneshot_allreduce_in_hipgraph():
    """The sharded decode step with its two all-reduces per layer on the one-shot transport (self-test against the staged
    collectives first), replayed from a hipGraph: tokens equal the unsharded model's, layer-0 evictions identical."""
    out = _launch(["--backend", "gloo", "--oneshot", "--graph"])
    assert "oneshot True" in out and "graph True" in out


def test_oneshot_allreduce_two_ranks_on_one_gpu():
    """cc_allreduce_* with two ranks sharing cuda:0: IPC-mapped peer buffers, remote stores, flags, epochs, the two alternating
    slot sets and hipGraph replays run for real, against the rank-ordered fp32 sum computed on the host (bit-exact)."""
    _launch(["--one_gpu"], script="allreduce_check.py", marker="ONESHOT ALLREDUCE CHECK OK")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the stores cross xGMI")
def test_oneshot_allreduce_over_xgmi():
    _launch([], script="allreduce_check.py", marker="ONESHOT ALLREDUCE CHECK OK")


@pytest.mark.parametrize("oneshot", [False, True])
def test_bench_gpus8_control_flow_dry_run(oneshot):
    """`bench.py --gpus 8` end to end on ONE GPU (CC_BENCH_DRYRUN_ONE_GPU=1: every rank on cuda:0, collectives staged through the
    host; NOT a measurement): self-launch of 8 ranks, the KV-head split down to H = 1 per rank (the shape every rank of the 8-GPU
    point runs), max-over-ranks timing, per-rank report, one JSON line from rank 0.  Two layers and a short prompt keep it cheap;
    the control flow is the full-size run's."""
    import json

    # oneshot (r4): the decode all-reduces on cc_allreduce_sum — IPC-mapped peer buffers of eight processes, verified against the
    # staged collectives at start-up, all ranks or none — INSIDE the captured decode graph (what the 8-GPU point will run once the
    # transport has proved itself over xGMI)
    env = dict(os.environ, CC_BENCH_DRYRUN_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CC_ONESHOT_ALLREDUCE="1" if oneshot else "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--n_layer", "2",
           "--prompt_len", "1024", "--cache_len", "512", "--no_cpu_baseline", "--no_live_pmc", "--roofline_iters", "2"] + (["--graph"] if oneshot else [])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]  # rank 0 alone reports
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 4 and out["warmup"] == 1 and out["value"] > 0
    cfg = out["config"]
    assert cfg["parallelism"] == "tp8" and cfg["rccl_ranks"] == 8
    assert cfg["per_rank"] is not None and len(cfg["per_rank"]) == 8
    if oneshot:
        assert cfg["decode_mode"] == "hipgraph" and cfg["decode_allreduce"].startswith("one-shot")
        assert all(r_["oneshot_selftest"] == "passed" and r_["oneshot_status"] == 0 for r_ in cfg["per_rank"])
