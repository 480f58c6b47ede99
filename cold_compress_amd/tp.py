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
    if torch.cuda.is_available():
        torch.cuda.set_device(rank)
        backend = "nccl"  # RCCL
    else:
        backend = "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", rank)),
                                world_size=int(os.environ.get("WORLD_SIZE", world)))
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


def _all_reduce_hook(_module, _input, output):
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
