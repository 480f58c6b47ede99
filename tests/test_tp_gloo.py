"""Tensor-parallel path on CPU with the gloo backend, world_size = 2 (ref: tp.py:59-176).

What is checked without a GPU: `apply_tp` slices wqkv per q/k/v block, wo/w2 row-wise, w1/w3 column-wise, shrinks
the head counts, and the two sum all-reduces per layer reconstruct exactly the single-process block output.
The attention CORE (cache + HIP attention) needs the device, so this test swaps in a test-local, per-head
independent double for it (plain causal softmax attention in torch, defined below) — the property under test is
the sharding/all-reduce wiring, which is independent of what the per-head core computes.  Per-head cache
state needs no exchange (every buffer is indexed by kv head), which the GPU parity tests cover per head.
"""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # spawned workers import tests/host_glue.py

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FullCacheDouble:
    """Test double with the cache surface model.py needs: keeps everything, never evicts."""
    head_specific = False
    max_cache_length = 1 << 20

    def return_attn(self):
        return False

    def update_kv(self, *a, **k):
        return None

    def update_state(self, *a, **k):
        return None


def _attention_double(q, k, v, attn_mask=None, return_attn=False, is_causal=None, **kw):
    """Per-head independent causal attention, GQA by head index (test-local reference, NOT the product path)."""
    HQ, H = q.shape[1], k.shape[1]
    R = HQ // H
    L = q.shape[2]
    kk, vv = k.repeat_interleave(R, 1), v.repeat_interleave(R, 1)
    w = (q @ kk.transpose(-1, -2)) / (q.shape[-1] ** 0.5)
    w = w.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    return torch.softmax(w, -1) @ vv, None


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
        import cold_compress_amd.harness.model as hm
        from cold_compress_amd import tp
        from cold_compress_amd.harness import ModelArgs, Transformer

        assert tp.maybe_init_dist() == rank and dist.get_backend() == "gloo"
        hm.scaled_dot_product_attention = _attention_double
        import host_glue

        host_glue.install(hm.glue)  # CPU tensors: model wiring only, with test-local glue and attention doubles
        torch.manual_seed(0)
        cfg = dict(block_size=64, vocab_size=64, n_layer=2, n_head=8, n_local_heads=4, dim=64, intermediate_size=96)
        full = Transformer(ModelArgs(**cfg)).eval()
        sharded = Transformer(ModelArgs(**cfg)).eval()
        sharded.load_state_dict(full.state_dict())
        tp.apply_tp(sharded)
        for m in (full, sharded):
            m.freqs_cis = hm.precompute_freqs_cis(64, 8, 10000, torch.float32)
            for layer in m.layers:
                layer.attention.kv_cache = _FullCacheDouble()
        a = sharded.layers[0].attention
        assert (a.n_head, a.n_local_heads, a.dim, a.head_dim) == (4, 2, 32, 8)
        assert a.wqkv.weight.shape == (32 + 16 + 16, 64) and a.wo.weight.shape == (64, 32)
        ff = sharded.layers[0].feed_forward
        assert ff.w1.weight.shape == (48, 64) and ff.w2.weight.shape == (64, 48)
        assert sharded.config.n_local_heads == 2  # caches will be built with H / world heads (tp.py:163-168)
        idx = torch.arange(12).view(1, 12) % 64
        pos = torch.arange(12)
        with torch.no_grad():
            y_full = full(idx, pos, is_prefill=True)
            y_tp = sharded(idx, pos, is_prefill=True)
        err = (y_full - y_tp).abs().max().item()
        # the q heads this rank owns are heads [rank*4, rank*4+4) of the full model, kv heads [rank*2, rank*2+2)
        wq_full = full.layers[0].attention.wqkv.weight[:64]
        assert torch.equal(a.wqkv.weight[:32], wq_full[rank * 32:(rank + 1) * 32])
        wk_full = full.layers[0].attention.wqkv.weight[64:96]
        assert torch.equal(a.wqkv.weight[32:48], wk_full[rank * 16:(rank + 1) * 16])
        q.put((rank, err))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise


def test_tp2_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert isinstance(err, float), f"rank {rank}: {err}"
        assert err < 1e-4, f"rank {rank}: TP output differs from the single-process model by {err}"


def test_tp_noop_without_torchrun(monkeypatch):
    from cold_compress_amd import tp

    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert tp.maybe_init_dist() is None
