"""Tensor-parallel path on CPU with the gloo backend, world_size 2, 4 and 8 with Llama-3-8B's head counts — 32 query / 8 kv heads:
4 / 2 / 1 kv heads per rank, the last being the `n_local_heads == 1` case the reference cannot run (ref: tp.py:59-176).

What is checked without a GPU: `apply_tp` slices wqkv per q/k/v block, wo/w2 row-wise, w1/w3 column-wise, shrinks
the head counts, and the two sum all-reduces per layer reconstruct exactly the single-process block output.
The attention CORE (cache + HIP attention) needs the device, so this test swaps in a test-local, per-head
independent double for it (plain causal softmax attention in torch, defined below) — the property under test is
the sharding/all-reduce wiring, which is independent of what the per-head core computes.  Per-head cache
state needs no exchange (every buffer is indexed by kv head), which the GPU parity tests cover per head.

Late r5, second test: the same sharding with the REAL product caches, compressors and step code — every C-ABI call served by the
oracle's twin on CPU tensors (tests/cpu_twin.py, test-only) — at world 2 and 8 (ONE kv head per rank): a compacted prompt and decode
steps with evictions give the unsharded run's tokens, and every rank's heads hold exactly the unsharded run's positions for them.
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
        # Llama-3-8B's head counts at a small width: 32 query heads, 8 kv heads (head_dim 8), so that world 2 / 4 / 8 leave 4 / 2 / 1 kv
        # heads per rank
        cfg = dict(block_size=64, vocab_size=64, n_layer=2, n_head=32, n_local_heads=8, dim=256, intermediate_size=384)
        full = Transformer(ModelArgs(**cfg)).eval()
        sharded = Transformer(ModelArgs(**cfg)).eval()
        sharded.load_state_dict(full.state_dict())
        tp.apply_tp(sharded)
        for m in (full, sharded):
            m.freqs_cis = hm.precompute_freqs_cis(64, 8, 10000, torch.float32)
            for layer in m.layers:
                layer.attention.kv_cache = _FullCacheDouble()
        a = sharded.layers[0].attention
        hq, hk = 32 // world, 8 // world
        assert (a.n_head, a.n_local_heads, a.dim, a.head_dim) == (hq, hk, hq * 8, 8)
        assert a.wqkv.weight.shape == ((hq + 2 * hk) * 8, 256) and a.wo.weight.shape == (256, hq * 8)
        ff = sharded.layers[0].feed_forward
        assert ff.w1.weight.shape == (384 // world, 256) and ff.w2.weight.shape == (256, 384 // world)
        assert sharded.config.n_local_heads == hk  # caches will be built with H / world heads (tp.py:163-168); 1 at world 8
        idx = torch.arange(12).view(1, 12) % 64
        pos = torch.arange(12)
        with torch.no_grad():
            y_full = full(idx, pos, is_prefill=True)
            y_tp = sharded(idx, pos, is_prefill=True)
        err = (y_full - y_tp).abs().max().item()
        # the q heads this rank owns are heads [rank*hq, (rank+1)*hq) of the full model, kv heads [rank*hk, (rank+1)*hk)
        wq_full = full.layers[0].attention.wqkv.weight[:256]
        assert torch.equal(a.wqkv.weight[: hq * 8], wq_full[rank * hq * 8:(rank + 1) * hq * 8])
        wk_full = full.layers[0].attention.wqkv.weight[256:320]
        assert torch.equal(a.wqkv.weight[hq * 8: (hq + hk) * 8], wk_full[rank * hk * 8:(rank + 1) * hk * 8])
        wv_full = full.layers[0].attention.wqkv.weight[320:384]
        assert torch.equal(a.wqkv.weight[(hq + hk) * 8:], wv_full[rank * hk * 8:(rank + 1) * hk * 8])
        # a real collective ran: the world the JSON line of bench.py reports comes from here (dist.get_world_size() after an all-reduce)
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t.item()) == world == dist.get_world_size()
        q.put((rank, err))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise


def _worker_real(rank, world, port, q):
    """The sharded model with the REAL product caches, compressors and attention entry points (every C-ABI call served by the oracle's
    twin on CPU tensors: tests/cpu_twin.py) against the unsharded model on the same weights: a compacted prompt, then decode steps with
    evictions — tokens equal, every rank's kv heads hold exactly the positions the unsharded run keeps for those heads."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
        import copy

        import pytest as _pytest

        from cold_compress_amd import tp
        from cold_compress_amd.harness import ModelArgs, Transformer, decode_one_token, prefill, setup_caches
        from cpu_twin import cpu_twin
        from oracle import oracle_lib

        torch.set_num_threads(1)
        oracle_lib.fns()
        mp_ = _pytest.MonkeyPatch()
        with cpu_twin(mp_, oracle_lib):
            assert tp.maybe_init_dist() == rank and dist.get_backend() == "gloo"
            torch.manual_seed(11)
            cfg = dict(block_size=128, vocab_size=64, n_layer=2, n_head=32, n_local_heads=8, dim=512, intermediate_size=256)
            full = Transformer(ModelArgs(**cfg)).eval()
            sharded = copy.deepcopy(full)
            tp.apply_tp(sharded)
            kw = dict(max_cache_length=[32.0], cache_bits=None, cache_length_pattern="tile", cache_strategy=["heavy_hitter"],
                      cache_strategy_pattern="tile", feed_long_prompts=False, prompt_compression_strategy=["heavy_hitter"], global_tokens=4,
                      recent_window=6, history_window_size=1, attn_thresholding=False, min_recovery_frac=0.9)
            L, steps = 60, 14
            prompt = torch.randint(0, 64, (L,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
            outs = []
            for model in (full, sharded):
                setup_caches(model, None, "cpu", L + 40, dict(kw))
                with torch.no_grad():
                    tok, _ = prefill(model, prompt.view(1, -1), torch.arange(L))
                    pos = torch.tensor([L], dtype=torch.int32)
                    toks, cur = [int(tok)], tok.view(1, 1).to(torch.int32)
                    for i in range(steps):
                        nt, _ = decode_one_token(model, cur, pos)
                        toks.append(int(nt))
                        # teacher-force the unsharded run's tokens: both runs see the same inputs
                        cur = (nt if model is full else torch.tensor(outs[0][0][len(toks) - 1])).view(1, 1).to(torch.int32)
                        pos += 1
                outs.append((toks, [l.attention.kv_cache.pos.clone() for l in model.layers], [int(l.attention.kv_cache.n_heads) for l in model.layers]))
            hk = 8 // world
            assert outs[1][2] == [hk, hk] and outs[0][2] == [8, 8]
            differ = [int((pa[:, rank * hk:(rank + 1) * hk] != pb).sum()) for pa, pb in zip(outs[0][1], outs[1][1])]
            same = sum(int(a == b) for a, b in zip(outs[0][0], outs[1][0]))
            evicted = int((outs[0][1][0] >= L).sum())  # (decode-time inserts present in the unsharded cache: evictions happened)
            q.put((rank, {"differ": differ, "same_tokens": same, "n_tokens": len(outs[0][0]), "inserted": evicted}))
            dist.barrier()
            dist.destroy_process_group()
        mp_.undo()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, repr(e) + traceback.format_exc()[-1500:]))
        raise


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 8])
def test_tp_gloo_real_cache_over_oracle_twins(world):
    """(world 8: ONE kv head per rank — the `n_local_heads == 1` path — through the real KVCacheHeavyHitter / compressor / step code)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, r in res:
        assert isinstance(r, dict), f"rank {rank}: {r}"
        assert r["inserted"] > 0, "no decode-time insert survived: the run did not evict"
        assert r["differ"] == [0, 0], f"rank {rank}: cache positions differ from the unsharded run's, per layer: {r['differ']}"
        assert r["same_tokens"] == r["n_tokens"], f"rank {rank}: {r['same_tokens']} of {r['n_tokens']} tokens equal"


@pytest.mark.parametrize("world", [2, 4, 8])
def test_tp_gloo_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert isinstance(err, float), f"rank {rank}: {err}"
        assert err < 1e-4, f"rank {rank}: TP output differs from the single-process model by {err}"


def _worker_capture_refused(rank, world, port, q, refusing_rank):
    """One rank's capture is refused (fault injection); every rank must end up on eager launches — TOGETHER: the verdicts are combined
    in a collective that every rank enters whatever happened locally, and none hangs in it."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cold_compress_amd.harness import negotiate_graphed_decoder

        captured, logs = [], []

        class Dec:  # stands in for GraphedDecoder: "capturing" is its first step
            pass

        def first_step(d):
            if rank == refusing_rank:
                raise RuntimeError("hipErrorStreamCaptureUnsupported: operation not permitted when stream is capturing (injected)")
            captured.append(d)

        dec = negotiate_graphed_decoder(Dec, first_step, torch.device("cpu"), log=logs.append)
        # a second negotiation in which nobody refuses: all ranks keep their decoder
        dec2 = negotiate_graphed_decoder(Dec, lambda d: None, torch.device("cpu"), log=logs.append)
        # ... and the ranks are still in step with each other afterwards (a collective behind the negotiation completes)
        t = torch.ones(1)
        dist.all_reduce(t)
        q.put((rank, dict(dec_is_none=dec is None, dec2_kept=dec2 is not None, captured_locally=len(captured), logs=logs, ranks=int(t.item()))))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world,refusing_rank", [(2, 1), (4, 0)])
def test_capture_refused_on_one_rank_sends_every_rank_to_eager(world, refusing_rank):
    """'hipGraph capture of the RCCL all-reduce refused -> every rank falls back together' (bench.py had the logic since
    r2, nothing exercised it).  harness.negotiate_graphed_decoder under gloo: the refusing rank raises inside its capture; EVERY rank
    must come back with no decoder (the ranks whose own capture succeeded drop it), nobody hangs, and a negotiation without a
    refusal keeps the decoders.  ref: tp.py:134-160 (the all-reduces a captured step would contain)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_capture_refused, args=(r, world, port, q, refusing_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, r in res:
        assert isinstance(r, dict), f"rank {rank}: {r}"
        assert r["dec_is_none"], f"rank {rank} kept a captured decoder although rank {refusing_rank}'s capture was refused"
        assert r["dec2_kept"], f"rank {rank}: a negotiation nobody refused must keep the decoder"
        assert r["ranks"] == world
        assert r["captured_locally"] == (0 if rank == refusing_rank else 1)
        assert any("refused" in m or "failed" in m for m in r["logs"]), f"rank {rank} fell back silently: {r['logs']}"


def test_tp_noop_without_torchrun(monkeypatch):
    from cold_compress_amd import tp

    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert tp.maybe_init_dist() is None
