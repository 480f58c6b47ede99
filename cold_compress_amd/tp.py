"""Tensor parallelism over the GPUs of one node: weight slicing + RCCL all-reduce over xGMI.

Same surface as ref: tp.py (`maybe_init_dist() -> Optional[int]`, `apply_tp(model)`), re-expressed for
one-process-per-GPU `torch.distributed` where backend "nccl" IS RCCL on ROCm (gloo on CPU for tests).
What shards: wqkv column-wise per q/k/v block, wo/w2 row-wise, w1/w3 column-wise; KV heads
and ALL per-head cache state shard with them (eviction needs no exchange); embeddings and the LM head are
replicas.  Two sum all-reduces per layer (after attention, after the FFN), issued in place on the ROCm
stream.  At decode the messages are 2*dim bytes (8-16 KiB): latency-bound on xGMI, one RCCL call each.

Unlike the reference, a rank may own a single KV head (Llama-3 70B at TP=8): the reference's
KVCacheHeavyHitter crashes there (cache.py:751/:480); ours does not.
KVCacheL2's score uses the max key norm over the heads of THIS rank, like the reference under TP.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


def _get_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def _get_world_size() -> int:
    return int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))


def maybe_init_dist() -> Optional[int]:
    """ref: tp.py:41-56.  Returns the rank, or None when there is nothing to parallelise."""
    rank, world = _get_rank(), _get_world_size()
    if world < 2:
        return None
    # the host driver only supports dmabuf IPC: RCCL / IPC handles between the ranks need this (exported by the image; set here as
    # well in case the launcher's environment lost it) — BEFORE the first torch.cuda call of this function, which may bring up
    # HIP / HSA (behind torch.cuda.is_available() the setdefault could be too late)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available():
        torch.cuda.set_device(rank)
        backend = "nccl"  # RCCL
    else:
        backend = "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", rank)),
                                world_size=int(os.environ.get("WORLD_SIZE", world)))
    # decode-size all-reduces over the one-shot xGMI transport: OPT-IN (CC_ONESHOT_ALLREDUCE=1) until it has run
    # on a real multi-GPU node (so far only two ranks sharing one GPU have exercised it); it still has to verify itself against
    # RCCL on this node before it takes over.  RCCL carries every all-reduce otherwise.
    if torch.cuda.is_available() and os.environ.get("CC_ONESHOT_ALLREDUCE", "0") == "1":
        enable_oneshot_allreduce()
    return rank


def _world_rank():
    if dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return _get_world_size(), _get_rank()


def _shard(x: torch.Tensor, dim: int, world: int, rank: int) -> torch.Tensor:
    assert x.size(dim) % world == 0, f"cannot shard size {x.size(dim)} over {world} ranks"
    return torch.tensor_split(x, world, dim=dim)[rank]


def _apply_tp_linear(linear: nn.Linear, style: str, weight_splits: List[int] = ()) -> None:
    """ref: tp.py:59-121 (bf16/fp32 linears only; weight-only int8/int4 are out of scope)."""
    world, rank = _world_rank()
    dim, attr = {"colwise": (0, "out_features"), "rowwise": (1, "in_features")}[style]
    assert getattr(linear, attr) % world == 0

    def split(t, d):
        if weight_splits:
            return torch.cat([_shard(p, d, world, rank) for p in t.split(list(weight_splits), dim=d)], dim=d)
        return _shard(t, d, world, rank)

    linear.weight = nn.Parameter(split(linear.weight, dim).contiguous(), requires_grad=False)
    if linear.bias is not None and style == "colwise":
        linear.bias = nn.Parameter(split(linear.bias, 0).contiguous(), requires_grad=False)
    setattr(linear, attr, getattr(linear, attr) // world)


class OneShotAllReduce:
    """The hand-written one-shot xGMI all-reduce of include/coldcompress.h (`cc_allreduce_*`) for the decode-size
    messages: every rank stores its vector into every peer's IPC-mapped buffer, flags, waits, and sums in rank order — one
    launch, capturable, bitwise-identical on all ranks.  Handles travel through torch.distributed (any backend).  Larger
    messages (prefill) stay on RCCL."""

    def __init__(self, max_bytes=64 * 1024, group=None):
        import ctypes as C

        from . import _abi

        self._abi, self._C = _abi, C
        self._comm = None
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.max_bytes = int(max_bytes)
        on_dev = dist.get_backend(group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_dev else torch.device("cpu")
        # Everything that can fail LOCALLY (a missing library, a HIP error) is caught and carried into the collectives below:
        # every rank always runs all_gather, agree, agree — whatever happened to it
        fns, nb, err = None, 64, ""  # (64 = sizeof(hipIpcMemHandle_t); only used to keep the all_gather's shape when the library is missing)
        try:
            fns = _abi.lib()
            nb = int(fns["cc_allreduce_handle_bytes"]())
        except Exception as e:
            err = f"{type(e).__name__}: {e}"

        def agree(ok):
            """MIN over the ranks of a local verdict: every rank runs the SAME sequence of collectives whatever fails locally
            (a rank that raised before a collective its peers entered would leave them waiting for ever)."""
            f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
            return int(f.item()) == 1

        comm, mine = C.c_void_p(), (C.c_uint8 * nb)()
        try:
            if err:
                raise RuntimeError(err)
            _abi.check(fns["cc_allreduce_create"](self.rank, self.world, self.max_bytes, C.byref(comm)), "cc_allreduce_create")
            self._comm = comm
            _abi.check(fns["cc_allreduce_export"](comm, mine), "cc_allreduce_export")
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        h = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
        allh = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(allh, h, group=group)
        if not agree(not err):
            self.close()
            raise RuntimeError(f"one-shot all-reduce: buffer creation / export failed on some rank ({err or 'another rank'})")
        try:
            flat = torch.cat([t.cpu() for t in allh]).contiguous()
            buf = (C.c_uint8 * flat.numel()).from_buffer_copy(flat.numpy().tobytes())
            _abi.check(fns["cc_allreduce_connect"](comm, buf), "cc_allreduce_connect")
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        # (also the barrier: every rank has mapped every buffer before the first store into one)
        if not agree(not err):
            self.close()
            raise RuntimeError(f"one-shot all-reduce: peer buffers could not be mapped on some rank ({err or 'another rank'})")

    def fits(self, t):
        return t.is_cuda and t.is_contiguous() and t.numel() * t.element_size() <= self.max_bytes and t.data_ptr() % 16 == 0 \
            and t.dtype in (torch.float32, torch.bfloat16, torch.float16)

    def all_reduce(self, t):
        """In place, on the current stream."""
        code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[t.dtype]
        st = self._C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._abi.check(self._abi.lib()["cc_allreduce_sum"](self._comm, self._C.c_void_p(t.data_ptr()), t.numel(), code, st),
                        "cc_allreduce_sum")
        return t

    def status(self):
        return int(self._abi.lib()["cc_allreduce_status"](self._comm))

    def close(self):
        if getattr(self, "_comm", None) is not None:
            self._abi.lib()["cc_allreduce_destroy"](self._comm)
            self._comm = None


_ONESHOT = None


def enable_oneshot_allreduce(max_bytes=64 * 1024, verify=True):
    """Route the two per-layer all-reduces of apply_tp through `OneShotAllReduce` when the message fits (decode), RCCL
    otherwise.  With `verify` (default) the transport first proves itself on this node: three all-reduces of a random vector
    (both slot sets) must match RCCL's result and leave the status word clear, ON EVERY RANK (the verdicts are combined with
    a MIN all-reduce over RCCL, so either all ranks switch or none does); any failure — IPC handles that cannot be opened,
    a timed-out flag wait, a wrong sum — leaves RCCL in place and returns None."""
    global _ONESHOT
    if _ONESHOT is not None:
        return _ONESHOT
    dev = torch.device("cuda", torch.cuda.current_device())
    ok, comm, why = 1, None, ""
    try:
        comm = OneShotAllReduce(max_bytes)
    except Exception as e:  # creation / handle exchange failed on this rank
        ok, why = 0, f"{type(e).__name__}: {e}"
    if verify:
        # every rank takes part in the same collectives whatever its own verdict: no rank may wait alone
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            g = torch.Generator(device=dev).manual_seed(4321 + dist.get_rank())
            for it in range(3):  # (no early exit: a rank that left the loop would leave its peers alone in the next RCCL call)
                x = torch.randn(4096, device=dev, generator=g).to(torch.bfloat16)
                ref = x.clone()
                dist.all_reduce(ref, op=dist.ReduceOp.SUM)
                if not ok:
                    continue
                got = x.clone()
                try:
                    comm.all_reduce(got)
                    torch.cuda.synchronize()
                    good = comm.status() == 0 and bool(torch.allclose(got.float(), ref.float(), rtol=2.0 ** -6, atol=2.0 ** -6))
                except Exception as e:
                    good, why = False, f"{type(e).__name__}: {e}"
                if not good:
                    ok, why = 0, why or f"self-test {it}: status {comm.status()} or sums differ from RCCL's"
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        if why:
            print(f"[cold_compress_amd.tp] one-shot all-reduce not enabled on rank {dist.get_rank()}: {why}", flush=True)
        if comm is not None:
            comm.close()  # (the IPC mappings and the uncached buffer are not left behind)
        return None
    _ONESHOT = comm
    return _ONESHOT


def oneshot_allreduce_status() -> int:
    """0, or non-zero if a one-shot all-reduce on ANY rank ever timed out waiting for a peer (its output was poisoned with NaN
    and the sums of that step are invalid on that rank).  A collective (MAX over the ranks) when torch.distributed is up:
    call it from every rank, after the decode loop.  0 when the transport is not in use."""
    if _ONESHOT is None:
        return 0
    st = 1 if _ONESHOT.status() != 0 else 0
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([st], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        st = int(t.item())
    return st


def check_oneshot_allreduce_status():
    """Raise loudly (on every rank) if a one-shot all-reduce timed out anywhere since the transport was enabled."""
    if oneshot_allreduce_status():
        raise RuntimeError("a one-shot xGMI all-reduce timed out waiting for a peer on some rank: the tokens produced since are "
                           "invalid (its output was poisoned with NaN).  Unset CC_ONESHOT_ALLREDUCE to keep RCCL for every all-reduce.")


def _all_reduce_hook(_module, _input, output):
    if _ONESHOT is not None and _ONESHOT.fits(output):
        _ONESHOT.all_reduce(output)
    else:
        dist.all_reduce(output, op=dist.ReduceOp.SUM)
    return output


def _apply_tp_ffn(mlp) -> None:
    """ref: tp.py:124-138."""
    _apply_tp_linear(mlp.w1, "colwise")
    _apply_tp_linear(mlp.w3, "colwise")
    _apply_tp_linear(mlp.w2, "rowwise")
    mlp.register_forward_hook(_all_reduce_hook)


def _apply_tp_attn(attn) -> None:
    """ref: tp.py:141-160."""
    world, _ = _world_rank()
    kv_size = attn.n_local_heads * attn.head_dim
    _apply_tp_linear(attn.wqkv, "colwise", [attn.dim, kv_size, kv_size])
    _apply_tp_linear(attn.wo, "rowwise")
    assert attn.n_local_heads % world == 0, "more ranks than KV heads: replicas only (SURVEY §8(e))"
    attn.n_head //= world
    attn.dim //= world
    attn.head_dim = attn.dim // attn.n_head
    attn.n_local_heads //= world
    attn.register_forward_hook(_all_reduce_hook)


def apply_tp(model) -> None:
    """ref: tp.py:163-176.  Call BEFORE setup_caches so caches are built with n_local_heads/world heads."""
    world, _ = _world_rank()
    cfg = model.config
    cfg.n_head //= world
    cfg.dim //= world
    cfg.n_local_heads //= world
    for block in model.layers:
        _apply_tp_ffn(block.feed_forward)
        _apply_tp_attn(block.attention)
