"""End-to-end GPU tests: the build's tiny-Llama harness + HIP cache/attention path against fixture F1
(captured from the reference's generation_utils.generate on CPU, fp32): generated tokens identical, per-step
per-layer eviction indices bit-exact, fp32 logits within 1e-3 (the north-star tolerance), final cache state.
Also: the hipGraph-captured decode step must reproduce the eager decode exactly.
"""
import argparse
import json

import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = __import__("helpers").TEST_DEVICE  # "cuda"; "cpu" only under tests/cpu_twin.py


def _build(f, n_layer):
    import cold_compress_amd.cache as cache
    from cold_compress_amd.harness import ModelArgs, Transformer, setup_caches

    cfg = dict(block_size=256, vocab_size=128, n_layer=n_layer, n_head=4, n_local_heads=2, dim=64, intermediate_size=128)
    model = Transformer(ModelArgs(**cfg)).to(torch.float32).eval()
    sd = {k[3:]: v for k, v in f.items() if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    ap = argparse.ArgumentParser()
    cache.add_cache_arguments(ap)
    kw = vars(ap.parse_args([]))
    kw.update(json.loads(f["cache_args_json"]))
    ck = setup_caches(model, None, DEV, f["prompt_len"] + f["new_tokens"], dict(kw))
    return model, ck


def _log_evictions(model):
    log = {i: [] for i in range(len(model.layers))}
    log["norms"] = {i: [] for i in range(len(model.layers))}
    for i, layer in enumerate(model.layers):
        kv = layer.attention.kv_cache
        orig = kv._run_select

        def wrapped(input_pos, k, v, _orig=orig, _kv=kv, _i=i):
            if hasattr(_kv, "key_norm"):
                log["norms"][_i].append(_kv.key_norm.clone())
            _orig(input_pos, k, v)
            log[_i].append(_kv._idx_buf().clone())

        kv._run_select = wrapped
        if hasattr(kv, "decode_step"):  # fused two-launch step: the slot is the arg-min key left for this position
            orig_step = kv.decode_step

            def step(query, k_val, v_val, input_pos, scale=None, _orig=orig_step, _kv=kv, _i=i):
                if not _kv._next_valid:
                    _kv.prepare_decode(input_pos)
                keys = _kv.next_key.cpu().numpy().view("uint64").min(axis=1)  # partial minima per chunk -> arg-min key
                if _kv.pos.shape[1] == 1:  # head-constant policy: every kv head keeps its own copy of the (identical) key row
                    if _kv.k_cache.is_cuda:  # (the oracle's twins — tests/cpu_twin.py — keep ONE row: only the minimum is contractual)
                        assert (keys == keys[0]).all(), "the kv heads' copies of a head-constant key row differ"
                    keys = keys.min(keepdims=True)
                log[_i].append(torch.from_numpy(((keys & 0xffffffff) >> 1).astype("int64")).to(DEV))
                return _orig(query, k_val, v_val, input_pos, scale)

            kv.decode_step = step
    return log


def _l2_tie_divergence(f, log, n_layer):
    """KVCacheL2 near-tie contract: keys of a REPEATED token differ only by a RoPE rotation, so
    their norms are equal up to the last ulp of torch.linalg.vector_norm's unspecified fp32 summation order
    (layer 0 of a looping greedy decode hits this).  A mismatching index is accepted only if the two slots'
    norms agree to 4 ulp; returns the first (layer, step) where that happened, else None."""
    first = None
    for li in range(n_layer):
        ref = f[f"evict_idx_L{li}"].long()
        mine = torch.stack(log[li]).cpu().view(ref.shape[0], -1)
        ref = ref.view(ref.shape[0], -1)
        for t in range(ref.shape[0]):
            if torch.equal(mine[t], ref[t]):
                continue
            kn = log["norms"][li][t].cpu()[0].double()
            for h in range(ref.shape[1]):
                a, b = kn[h, mine[t, h]], kn[h, ref[t, h]]
                assert abs(a - b) <= 4 * 1.2e-7 * max(abs(a), abs(b)), f"layer {li} step {t}: not a norm tie"
            if first is None or t < first[1]:
                first = (li, t)
            break
    return first


def _run(name, graphed=False):
    from cold_compress_amd.harness import GraphedDecoder, decode_one_token, generate, prefill

    f = load_golden(name)
    model, ck = _build(f, f["n_layer"])
    assert list(ck["max_cache_length"]) == f["max_cache_length"].tolist()
    assert list(ck["recent_window"]) == f["recent_window"].tolist()
    log = None if graphed else _log_evictions(model)
    logits = []
    if not graphed:
        orig = model.forward

        def fwd(*a, **k):
            out = orig(*a, **k)
            logits.append(out[0, -1].detach().float().clone())
            return out

        model.forward = fwd
    dec = GraphedDecoder(model) if graphed else decode_one_token
    seq, probs, stats = generate(model, f["prompt"].to(DEV), prefill, dec, max_new_tokens=f["new_tokens"])
    torch.cuda.synchronize()
    return f, model, seq.cpu(), log, logits


@pytest.mark.parametrize("name", ["f1_e2e_recent_global.npz", "f1_e2e_full.npz", "f1_e2e_heavy_hitter.npz",
                                  "f1_e2e_heavy_hitter_short.npz", "f1_e2e_l2.npz", "f1_e2e_hh_pyramid.npz"])
def test_e2e_matches_reference(name, audit):
    check_e2e(name, audit)


def check_e2e(name, audit=None):
    """(also run by tests/test_host_e2e_cpu.py: the same harness on CPU tensors over the oracle's twins)"""
    f, model, seq, log, logits = _run(name)
    got = torch.stack(logits).cpu()
    assert got.shape == f["logits"].shape
    if "l2" in name:
        div = _l2_tie_divergence(f, log, f["n_layer"])
        if audit is not None:
            n_ev = sum(int(f[f"evict_idx_L{li}"].numel()) for li in range(f["n_layer"]))
            audit(f"l2 runs that left the reference behind a verified norm tie = {0 if div is None else 1}", rule="l2 norm tie",
                  count=0 if div is None else 1, compared=n_ev, limit="the two slots' norms within 4 ulp; tokens then follow the reference's own top-2 margin")
        if div is not None:  # identical up to the verified norm tie; afterwards the cache contents legitimately differ
            t, n = div[1] + 1, int(f["prompt_len"])  # logits row 0 is the prefill (token index n); decode step t is row t + 1
            assert torch.equal(seq[: n + t], f["seq"][: n + t]), "generated tokens differ from the reference before the norm tie"
            assert (got[:t] - f["logits"][:t]).abs().max() < 1e-3
            # behind the tie: the logits stay close while the tokens agree, and a token may differ only where the reference's own two
            # best logits are closer than twice the deviation there (r5, a fresh-seed run: margin 0.0013 at a deviation of 0.003) —
            # from then on it is another generation
            for r in range(t, got.shape[0]):
                dev = float((got[r] - f["logits"][r]).abs().max())
                assert dev < 2e-2, f"logits row {r}: {dev}"
                if int(seq[n + r]) != int(f["seq"][n + r]):
                    top2 = f["logits"][r].float().topk(2).values
                    assert float(top2[0] - top2[1]) <= 2 * dev, f"token {n + r} differs where the reference's margin is {float(top2[0] - top2[1])}"
                    break
            return
    assert torch.equal(seq, f["seq"]), "generated tokens differ from the reference"
    assert (got - f["logits"]).abs().max() < 1e-3, "fp32 logits differ by more than the north-star 1e-3"
    for li, layer in enumerate(model.layers):
        kv = layer.attention.kv_cache
        ref_idx = f[f"evict_idx_L{li}"]
        mine = torch.stack(log[li]).cpu() if log[li] else torch.zeros(0, 1)
        assert mine.shape[0] == ref_idx.shape[0]
        assert torch.equal(mine.view(ref_idx.shape[0], -1), ref_idx.view(ref_idx.shape[0], -1).long()), f"layer {li} eviction indices"
        assert torch.equal(kv.pos.cpu(), f[f"final_pos_L{li}"])
        assert torch.equal(kv.mask.cpu(), f[f"final_mask_L{li}"])
        assert torch.equal(kv.cache_cts.cpu(), f[f"final_cts_L{li}"])
        assert (kv.k_cache.cpu() - f[f"final_k_L{li}"]).abs().max() < 1e-4
        if f"final_denom_L{li}" in f:
            assert torch.equal(kv.attn_history_denom.cpu(), f[f"final_denom_L{li}"])
            assert torch.allclose(kv.attn_history_num.cpu(), f[f"final_num_L{li}"], rtol=1e-4, atol=1e-6)
    stats = model.get_cache_stats(f["prompt_len"], f["new_tokens"])
    assert abs(stats["compression_ratio_avg"] - f["compression_ratio_avg"]) < 1e-6


def test_c1_ring_known_answer():
    """BASELINE config C1: recent_global, S=16, g=4 -> slot at decode step t is 4 + (t mod 12)."""
    f, model, seq, log, _ = _run("f1_e2e_recent_global.npz")
    for li in range(2):
        idx = [int(x) for x in torch.stack(log[li]).cpu().view(-1)]
        assert idx == [4 + (t % 12) for t in range(len(idx))]


@pytest.mark.parametrize("name", ["f1_e2e_heavy_hitter.npz", "f1_e2e_l2.npz", "f1_e2e_recent_global.npz"])
def test_hipgraph_decode_equals_eager(name):
    """The hipGraph-captured decode step must reproduce eager launches exactly (tokens and all cache state)."""
    f, model_g, seq_g, _, _ = _run(name, graphed=True)
    _, model_e, seq_e, _, _ = _run(name, graphed=False)
    assert torch.equal(seq_g, seq_e) and torch.equal(seq_g, f["seq"])
    for lg, le in zip(model_g.layers, model_e.layers):
        a, b = lg.attention.kv_cache, le.attention.kv_cache
        for (na, ta), (nb, tb) in zip(a.named_buffers(), b.named_buffers()):
            assert na == nb and torch.equal(ta, tb), na


@pytest.mark.parametrize("name", ["f1_e2e_heavy_hitter.npz", "f1_e2e_l2.npz", "f1_e2e_recent_global.npz"])
def test_one_graphed_decoder_across_two_generations(name):
    """ONE GraphedDecoder reused across reset() + prefill.  prepare_decode sits outside the captured step, so the
    replay of the second generation's first token must find every cache's fused pipeline (`next_key`, `step_commit`) re-seeded at ITS
    position — r5 re-seeded KVCacheRandom only; the others evicted at the previous generation's candidate.  Both generations (same
    prompt, caches reset in between) must reproduce the reference's sequence and leave the eager run's cache state."""
    from cold_compress_amd.harness import GraphedDecoder, decode_one_token, generate, prefill

    f = load_golden(name)
    model, _ = _build(f, f["n_layer"])
    dec = GraphedDecoder(model)
    seqs = []
    for gen_no in range(2):
        model.reset_caches()
        seq, _, _ = generate(model, f["prompt"].to(DEV), prefill, dec, max_new_tokens=f["new_tokens"])
        torch.cuda.synchronize()
        seqs.append(seq.cpu())
        assert dec.graph is not None
    _, model_e, seq_e, _, _ = _run(name, graphed=False)
    assert torch.equal(seqs[0], seq_e), "first generation on the graphed decoder"
    assert torch.equal(seqs[1], seq_e), "second generation on the SAME graphed decoder"
    for lg, le in zip(model.layers, model_e.layers):
        for (na, ta), (nb, tb) in zip(lg.attention.kv_cache.named_buffers(), le.attention.kv_cache.named_buffers()):
            assert na == nb and torch.equal(ta, tb), na


@pytest.mark.parametrize("strategy,extra", [("heavy_hitter", {"history_window_size": 8}), ("l2", {}), ("hybrid", {})])
def test_harness_two_launch_step_equals_three_call_path(strategy, extra):
    """Through the caller model (bf16, head_dim 128, hipGraph decode): the fused decode step / fused history of every
    policy added after the goldens were captured — finite history window, l2, hybrid — generates the same tokens and
    leaves the same cache state as the plain update_kv -> attention -> update_state sequence.  (random is covered with
    injected draws in test_gpu_fused_step.py: its RNG stream differs between eager launches and a captured graph.)"""
    import cold_compress_amd.cache as cache
    from cold_compress_amd.harness import GraphedDecoder, ModelArgs, Transformer, decode_one_token, prefill, setup_caches

    hyb = [{"strategy": "special"}, {"strategy": "special_punc"}, {"strategy": "special_punc_heavy_hitter", "heavy_hitter_frac": 0.3},
           {"strategy": "special_punc_window", "recent_window": 0.3}, {"strategy": "full"}]

    class Tok:
        def special_ids(self):
            return [[1], [2, 3]]

        def punctuation_ids(self):
            return [5, 6, 7]

    def build(fused):
        torch.manual_seed(5)
        cfg = dict(block_size=512, vocab_size=256, n_layer=2, n_head=8, n_local_heads=2, dim=1024, intermediate_size=512)
        model = Transformer(ModelArgs(**cfg)).to(torch.bfloat16).eval().to(DEV)
        ap = argparse.ArgumentParser()
        cache.add_cache_arguments(ap)
        kw = vars(ap.parse_args([]))
        kw.update(dict(cache_strategy=[strategy], prompt_compression_strategy=["full" if strategy == "hybrid" else strategy],
                       max_cache_length=[1.0 if strategy == "hybrid" else 96.0], global_tokens=4, recent_window=10,
                       hybrid_strategies=hyb, min_recovery_frac=0.9), **extra)
        setup_caches(model, Tok(), DEV, 300, dict(kw))
        for layer in model.layers:
            layer.attention.fuse_decode_step = fused
            layer.attention.fuse_state_update = fused
        return model

    outs = []
    for fused in (False, True):
        model = build(fused)
        gen = torch.Generator().manual_seed(9)
        prompt = torch.randint(8, 256, (200,), generator=gen, dtype=torch.int32).to(DEV)
        prompt[::17] = 6  # some punctuation for the hybrid policies
        with torch.no_grad():
            tok, _ = prefill(model, prompt.view(1, -1), torch.arange(200, device=DEV))
            if strategy == "hybrid":  # random weights profile every head as "full": force the reference's policy mix
                for layer in model.layers:
                    kv = layer.attention.kv_cache
                    kv.cache_strategies = (torch.arange(kv.n_heads, device=DEV) % len(hyb)).to(torch.int64).contiguous()
                    kv.requires_heavy_hitter = kv.requires_punc = kv.requires_special = True
            dec = GraphedDecoder(model) if fused else decode_one_token
            pos = torch.tensor([200], dtype=torch.int32, device=DEV)
            cur = tok.view(1, 1).to(torch.int32)
            toks = []
            for _ in range(24):
                cur = dec(model, cur, pos)[0].view(1, 1)
                toks.append(int(cur))
                pos += 1
        state = {}
        for li, layer in enumerate(model.layers):
            kv = layer.attention.kv_cache
            for n, b in kv.named_buffers():
                if n not in ("next_key", "step_commit"):
                    state[f"{li}.{n}"] = b.clone()
        outs.append((toks, state))
    assert outs[0][0] == outs[1][0], "generated tokens"
    for n in outs[0][1]:
        assert torch.equal(outs[0][1][n], outs[1][1][n]), n
