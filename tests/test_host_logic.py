"""CPU tests: host logic (budgets, registry, kwargs plumbing, statistics), the C-ABI surface of the built
library (every symbol include/coldcompress.h declares is exported; no compute calls without a GPU), and the
no-fallback rule (CPU tensors are refused loudly)."""
import argparse
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def f8():
    with open(os.path.join(GOLDEN, "f8_budgets.json")) as fh:
        return json.load(fh)


def test_budget_arithmetic_matches_reference(f8):
    from cold_compress_amd.harness import apply_pattern, apply_pyramid_pattern, find_multiple, normalize_cache_length

    for frac, mx, want in f8["normalize"]:
        assert normalize_cache_length(frac, mx) == want
    for pat, n, strat, want in f8["pattern"]:
        assert apply_pattern(pat, n, strat) == want
    for length, mx, n, dec, want in f8["pyramid"]:
        assert apply_pyramid_pattern(length, mx, n, decreasing=dec) == want
        assert apply_pattern([length], n, "pyramid" if dec else "funnel", max_seq_length=mx) == want
    for n, k, want in f8["find_multiple"]:
        assert find_multiple(n, k) == want


def test_baseline_config_lengths():
    """C2 -> 2560, C5 -> 3488, C4(ii) pyramid starts 2036, 1916 and ends in 256s."""
    from cold_compress_amd.harness import apply_pyramid_pattern, normalize_cache_length

    assert normalize_cache_length(0.25, 8192 + 2048) == 2560
    assert normalize_cache_length(0.1, 32768 + 2048) == 3488
    lens = apply_pyramid_pattern(1024, 18432, 32)
    assert lens[:2] == [2036, 1916] and lens[-4:] == [256] * 4 and len(lens) == 32


def test_registry_and_relevant_kwargs(f8):
    import cold_compress_amd.cache as cache

    for strat in ["full", "random", "recent_global", "heavy_hitter", "l2", "keep_it_odd"]:
        cls, rk = cache.get_cache_constructor(strat)
        assert rk == f8["relevant_kwargs"][strat], strat
        assert callable(cls)
    with pytest.raises(ValueError):
        cache.get_cache_constructor("nope")
    ap = argparse.ArgumentParser()
    cache.add_cache_arguments(ap)
    assert vars(ap.parse_args([])) == f8["arg_defaults"]
    cache.add_extension_arguments(ap)  # our own flags live apart from the reference's set
    assert set(vars(ap.parse_args([]))) - set(f8["arg_defaults"]) == {"cache_quant_mode"}
    ns = ap.parse_args(["--cache_strategy", "heavy_hitter", "--prompt_compression_strategy", "heavy_hitter",
                        "--max_cache_length", "0.25"])
    cache.cache_compatibility(ns)
    ns.prompt_compression_strategy = ["recent_global"]
    with pytest.raises(AssertionError):
        cache.cache_compatibility(ns)


def test_recent_window_lists(f8):
    from cold_compress_amd.harness import ModelArgs, Transformer, setup_caches

    for rw, lens, want in f8["recent_window"]:
        model = Transformer(ModelArgs(block_size=64, vocab_size=16, n_layer=len(lens), n_head=2, dim=16, intermediate_size=32))
        ck = dict(max_cache_length=[float(x) for x in lens], cache_length_pattern="tile", cache_strategy=["l2"],
                  prompt_compression_strategy=["l2"], cache_strategy_pattern="tile", recent_window=rw, global_tokens=1,
                  cache_bits=None)
        out = setup_caches(model, None, "cpu", 1 << 20, ck)
        # lengths are rounded up to multiples of 8 first (generation_utils.py:276); windows follow those
        exp = [max(1, int(rw * n)) if rw <= 1 else max(1, min(rw, n)) for n in out["max_cache_length"]]
        assert out["recent_window"] == exp
        for layer, n in zip(model.layers, out["max_cache_length"]):
            kv = layer.attention.kv_cache
            assert kv.max_cache_length == n and kv.pos.shape == (1, 2, n) and kv.pos.dtype == torch.int32
            assert kv.key_norm.shape == (1, 2, n) and kv.mask.dtype == torch.bool and kv.cache_cts.dtype == torch.int32


def test_buffers_and_statistics_on_cpu():
    import cold_compress_amd.cache as cache

    kw = dict(max_cache_length=16, global_tokens=2, max_seq_length=64, cache_bits=None, history_window_size=1,
              recent_window=3, attn_thresholding=False)
    kv = cache.KVCacheHeavyHitter(1, 2, 8, torch.bfloat16, **kw)
    assert kv.attn_history_num.dtype == torch.float64 and kv.attn_history_num.shape == (1, 2, 16, 1)
    assert kv.attn_history_denom.dtype == torch.int32 and kv.attn_counter.dtype == torch.int64
    assert kv.return_attn() and kv.head_specific and bool((kv.pos == -1).all())
    kv.cache_cts.fill_(10)
    st = kv.compute_statistics(torch.tensor(41))
    assert abs(st["compression_ratio"] - (40 - 10) / 40) < 1e-6
    # the reference's figure (cache.py:247-257): the cache's own state tensors; our non-persistent pipeline state
    # (partial arg-min keys of the fused decode step) is not cache content
    nbytes = sum(b.numel() * b.element_size() for n, b in kv.named_buffers() if n not in kv._non_persistent_buffers_set)
    assert abs(st["cache_memory_gb"] - nbytes / 2 ** 30) < 1e-12
    ref_names = {"k_cache", "v_cache", "pos", "cache_cts", "mask", "attn_history_num", "attn_history_denom", "attn_counter"}
    assert {n for n, _ in kv.named_buffers() if n not in kv._non_persistent_buffers_set} == ref_names
    kv.reset()
    assert int(kv.cache_cts[0]) == 0
    with pytest.raises(NotImplementedError):  # the reference itself crashes on this path
        cache.KVCacheHeavyHitter(1, 2, 8, torch.bfloat16, **{**kw, "attn_thresholding": True})
    # quantised KV (cache.py:180-198): the image the reference holds + one (scale, zero point) per slot
    q4 = cache.KVCacheHeavyHitter(1, 2, 8, torch.bfloat16, **{**kw, "cache_bits": 4})
    assert q4.k_cache_q.dtype == torch.uint8 and q4.k_cache_q.shape == (2 * 16 * 8 // 2,) and q4.k_scales.shape == (16,)
    q8 = cache.KVCacheRecentGlobal(1, 2, 8, torch.float32, max_cache_length=16, global_tokens=2, max_seq_length=64, cache_bits=8)
    assert q8.k_cache_q.dtype == torch.int8 and q8.k_cache_q.shape == (1, 2, 16, 8) and q8.v_zero_points.dtype == torch.float32
    q8.cache_cts.fill_(10)
    assert abs(q8.compute_statistics(torch.tensor(41))["compression_ratio"] - (40 - 10 * 8 / 16) / 40) < 1e-6  # cache.py:279-281
    for cls, extra in ((cache.KVCacheL2, dict(recent_window=3)), ):
        with pytest.raises(NotImplementedError):  # reference crashes: norm of the int8 image
            cls(1, 2, 8, torch.bfloat16, max_cache_length=16, global_tokens=2, max_seq_length=64, cache_bits=8, **extra)
    # debug_<strategy> (cache.py:1460-1474): a constructor + the analysed strategy's kwargs + prompt_compression_strategy
    ctor, rk = cache.get_cache_constructor("debug_heavy_hitter")
    assert rk == cache.KVCacheHeavyHitter.relevant_kwargs + ["prompt_compression_strategy"]
    f10 = json.loads(str(np.load(os.path.join(GOLDEN, "f10_analysis_hh_f32.npz"))["relevant_kwargs_json"]))
    assert rk == f10  # the reference's list (captured with its missing cache_bits keyword injected)
    an = ctor(1, 2, 8, torch.bfloat16, **{**kw, "prompt_compression_strategy": "heavy_hitter"})
    assert isinstance(an, cache.KVCacheAnalysis) and an.max_cache_length == 64 and an.compressed.max_cache_length == 16
    assert an.attention_losses.shape == (64,) and bool((an.attention_losses == -1).all()) and an.return_attn() and an.head_specific
    with pytest.raises(ValueError):
        cache.get_cache_constructor("debug_nonsense")
    ring = cache.KVCacheHeavyHitter(1, 2, 8, torch.bfloat16, **{**kw, "history_window_size": 4})
    assert ring.attn_history_num.shape == (1, 2, 16, 4) and ring.attn_history_num.dtype == torch.bfloat16  # cache.py:661-667
    hist = ring.fused_history()  # W > 1: ring, denom, counter, W, tracked accumulators, window sums (no launch: state is current)
    assert len(hist) == 6 and hist[3] == 4 and hist[5].shape == (2, 16) and hist[4].dtype == torch.int64


def test_cpu_tensors_refused_no_fallback():
    import cold_compress_amd.cache as cache
    from cold_compress_amd._abi import ColdCompressError
    from cold_compress_amd.attention_utils import scaled_dot_product_attention

    kv = cache.KVCacheRecentGlobal(1, 2, 8, torch.float32, max_cache_length=16, global_tokens=2, max_seq_length=64, cache_bits=None)
    with pytest.raises(ColdCompressError):
        kv.update_kv(torch.arange(4), torch.zeros(1, 2, 4, 8), torch.zeros(1, 2, 4, 8), True)
    with pytest.raises(ColdCompressError):
        scaled_dot_product_attention(torch.zeros(1, 4, 1, 8), torch.zeros(1, 2, 16, 8), torch.zeros(1, 2, 16, 8))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under cold_compress_amd/ may reference it."""
    pkg = os.path.join(ROOT, "cold_compress_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+[\w.]*oracle|CDLL\([^)]*oracle|import_module\([^)]*oracle|liboracle", src, re.M), fn


def test_abi_exports_every_declared_symbol():
    from cold_compress_amd import _abi, _build

    so = _build.build()
    header = open(os.path.join(ROOT, "include", "coldcompress.h")).read()
    debug = open(os.path.join(ROOT, "include", "coldcompress_debug.h")).read()  # measurement / test hooks and the A/B switches (r5)
    decl = lambda text: set(re.findall(r"^(?:int|int32_t|size_t|void|const char\*)\s+(cc_[a-z0-9_]+)\s*\(", text, re.M))  # noqa: E731
    declared = decl(header) | decl(debug)
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    # the boundary header carries no hook and no process-wide setter
    assert not [n for n in decl(header) if "_set_" in n or "debug" in n or "trace" in n or n.endswith("_phases") or "stream_floor" in n]
    lib = C.CDLL(so)
    fns = _abi.bind(lib)  # raises AttributeError if a symbol is missing
    assert fns["cc_abi_version"]() == 1
    assert fns["cc_error_string"](-4) == b"workspace too small"
    # argument validation happens before any launch, so it is checkable without a GPU
    assert fns["cc_hh_update"](None, None, None, None, 1, 1, 1, 0, None) == -1
    assert fns["cc_decode_attn_workspace_bytes"](32, 8, 4096, 128, 1) > 32 * 4096 * 2
    assert fns["cc_topk_keep"](None, 0, 1, 1, 1, None, None, 0, None) == -1


def test_oracle_exports_cpu_twins(oracle):
    from cold_compress_amd import _abi

    fns = oracle.fns()
    assert set(fns) == set(_abi.SIGNATURES) - _abi.DEVICE_ONLY


def test_oracle_glue_twins_match_torch_cpu(oracle):
    """The oracle's twins of the caller-glue entry points (cc_gemv_fused, cc_softmax_argmax) against plain PyTorch on
    CPU in fp32 — they are the checkers of the GPU tests, so they get pinned here (tolerance: fp32 summation order)."""
    import numpy as np

    g = torch.Generator().manual_seed(3)
    N, K, hd = 96, 64, 16
    W, W3 = torch.randn(N, K, generator=g) * 0.1, torch.randn(N, K, generator=g) * 0.1
    x, delta, nw = torch.randn(K, generator=g), torch.randn(K, generator=g) * 0.3, 1 + 0.1 * torch.randn(K, generator=g)
    ang = torch.rand(hd // 2, generator=g) * 6.28
    freqs = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    h = x + delta
    n = h * torch.rsqrt((h * h).mean() + 1e-5) * nw
    o = oracle
    f = lambda t: o.ptr(t.numpy().copy()) if t is not None else None  # noqa: E731
    # norm prologue + RoPE epilogue on the first 64 rows
    y, hout = np.zeros(N, np.float32), np.zeros(K, np.float32)
    o.call("cc_gemv_fused", f(W), None, f(x), f(delta), f(nw), 1e-5, o.ptr(hout), None, f(freqs), 64, hd, o.ptr(y), N, K, 0, None)
    t = W @ n
    rr = t[:64].view(-1, hd // 2, 2)
    c, s = freqs[:, 0].view(1, -1), freqs[:, 1].view(1, -1)
    ref = torch.cat([torch.stack([rr[..., 0] * c - rr[..., 1] * s, rr[..., 1] * c + rr[..., 0] * s], -1).reshape(-1), t[64:]])
    assert np.allclose(y, ref.numpy(), rtol=1e-5, atol=1e-5) and np.allclose(hout, h.numpy(), rtol=0, atol=0)
    # SwiGLU pair
    o.call("cc_gemv_fused", f(W), f(W3), f(x), None, None, 1e-5, None, None, None, 0, 0, o.ptr(y), N, K, 0, None)
    assert np.allclose(y, (torch.nn.functional.silu(W @ x) * (W3 @ x)).numpy(), rtol=1e-5, atol=1e-5)
    # greedy tail: first maximal element on ties
    logits = torch.randn(1000, generator=g)
    logits[17] = logits[900] = 9.0
    probs, idx = np.zeros(1000, np.float32), np.zeros(1, np.int32)
    o.call("cc_softmax_argmax", f(logits), 1000, 0, o.ptr(probs), o.ptr(idx), None, 0, None)
    assert np.allclose(probs, torch.softmax(logits, -1).numpy(), rtol=1e-5, atol=1e-8) and int(idx[0]) == 17
    # NaN: torch.argmax returns the first NaN's index (r5: never an index outside the vocabulary)
    logits[400] = float("nan")
    o.call("cc_softmax_argmax", f(logits), 1000, 0, o.ptr(probs), o.ptr(idx), None, 0, None)
    assert int(idx[0]) == int(torch.argmax(torch.softmax(logits, -1))) == 0


def test_attn_summary_runs_reference_group_mean_unchanged():
    """model.py:413-418 of the reference, verbatim call sequence, on the prefill kernel's AttnSummary: both for a
    GQA-shaped kernel call (summary already per kv head) and for pre-repeated K/V (one row per query head)."""
    from cold_compress_amd._abi import ColdCompressError
    from cold_compress_amd.prompt_compression import AttnSummary

    bsz, n_local_heads, n_head, seqlen = 1, 2, 8, 12
    g = torch.Generator().manual_seed(0)
    for rows in (n_local_heads, n_head):
        colsum, obs = torch.rand(rows, seqlen, generator=g), torch.rand(rows, seqlen, generator=g)
        attn = AttnSummary(colsum, obs, 4, torch.bfloat16, {3: torch.rand(rows, seqlen, generator=g)})
        assert attn is not None and attn.ndim == 4
        attn2 = attn.view(bsz, n_local_heads, n_head // n_local_heads, seqlen, -1).mean(dim=2)  # the reference's two lines
        assert isinstance(attn2, AttnSummary) and attn2.colsum.shape == (n_local_heads, seqlen)
        if rows == n_local_heads:
            assert attn2 is attn
        else:
            R = n_head // n_local_heads
            assert torch.allclose(attn2.colsum, colsum.view(n_local_heads, R, seqlen).mean(1))
            assert torch.allclose(attn2.obs_mean, obs.view(n_local_heads, R, seqlen).mean(1))
            assert set(attn2.bands) == {3} and attn2.bands[3].shape == (n_local_heads, seqlen)
    with pytest.raises(ColdCompressError):
        attn.view(bsz, 3, 2, seqlen, -1)
    with pytest.raises(ColdCompressError):
        attn.view(bsz, n_local_heads, 4, seqlen, -1).mean(dim=1)


def test_recoverable_classification_and_key_row_shapes():
    """Which caches the in-band retry of a timed-out single-launch step may touch (harness._recover_token asks `recoverable()`),
    and the layout the fused pipelines rely on: one key row per kv head for EVERY policy (the head-constant ones used to share
    one row across heads — a race, DESIGN round-3 table), commit words per kv head."""
    import cold_compress_amd.cache as cache

    H, D, S = 4, 16, 64
    base = dict(max_cache_length=S, max_seq_length=4 * S, cache_bits=None, global_tokens=2, recent_window=3, history_window_size=1,
                attn_thresholding=False)

    def mk(strategy, **extra):
        cls, rk = cache.get_cache_constructor(strategy)
        kw = dict(base, **extra)
        return cls(1, H, D, torch.float32, **{k: kw[k] for k in rk})

    for strategy in ("heavy_hitter", "recent_global", "full", "random"):
        kv = mk(strategy)
        assert kv.recoverable(), strategy
        assert tuple(kv.next_key.shape)[0] == H and tuple(kv.step_commit.shape) == (H, 68) and int(kv.step_commit.max()) == -1
    assert not mk("heavy_hitter", history_window_size=8).recoverable()  # the ring step carries no commit words
    assert mk("l2").recoverable()  # (r4: per-workgroup commit words; the norm maximum is republished by every workgroup on a retry)
    rnd = mk("random")
    rnd._rand = lambda: torch.zeros(S)  # an injected vector would be drawn again by a retry
    assert not rnd.recoverable() and not rnd._in_kernel_rng()


def test_decode_loop_rewinds_to_a_late_detected_failed_token(monkeypatch):
    """harness.decode_n_tokens reads the single-launch status word through an asynchronous copy and may see a failure a few
    tokens late (no device synchronisation per token): it must rewind to the failed token — positions, token list,
    the token fed next — and end with exactly the fault-free sequence.  Host logic only: the watch and the retry are stand-ins."""
    import torch

    from cold_compress_amd.harness import generation as G

    calls = []
    real_watch = G._StatusWatch  # (run() patches the module attribute)

    live = {"state": None, "fail_at": None, "n": 0}

    def step(model, x, pos, next_token=None, attn_top_k=1.0, **kw):
        calls.append((int(x.view(-1)[0]), int(pos[0])))
        t = torch.tensor([(int(x.view(-1)[0]) * 7 + int(pos[0])) % 101], dtype=torch.int32)
        st = live["state"]
        # the fake device: the failing token's launch and every launch behind a set status word produce GARBAGE (they did nothing)
        k = int(pos[0]) - 40
        if st is not None and (st["set"] or (k == live["fail_at"] and not st["failed_once"])):
            t = torch.tensor([977], dtype=torch.int32)
        return (next_token if next_token is not None else t), torch.ones(1)

    def run(fail_at, lag, n=12, terminators=None, depth=64):
        """The fake device: token `fail_at` (first attempt only) sets the word; it becomes visible `lag` tokens later."""
        state = {"set": False, "failed_once": False}
        live["state"], live["fail_at"] = state, fail_at

        class Watch(real_watch):
            """The REAL ring / overflow / ordering logic over a stand-in device: a verdict is `done` once `lag` later tokens were posted."""

            def __init__(self, dev):
                super().__init__(dev, depth=depth)

            def _open(self, dev):
                self.values, self.started, self.n_posted = {}, {}, 0

            def _start(self, slot):
                i = self.n_posted  # (tokens are posted in order; a rewind restarts the count through clear())
                if i == fail_at and not state["failed_once"]:
                    state["set"], state["failed_once"] = True, True
                self.values[slot] = int(state["set"])  # (sticky: every later token sees it too)
                self.started[slot] = i
                self.n_posted += 1

            def post(self, token_index):
                self.n_posted = token_index
                super().post(token_index)

            def _is_done(self, slot):
                return self.n_posted - 1 - self.started[slot] >= lag

            def _wait_done(self, slot):
                pass

            def _status(self, slot):
                return self.values[slot]

        def recover(model, cur, pos, fn, nt, npb, forced, top_k, kw, max_retries=6):
            assert state["set"]
            state["set"] = False  # reset_single_launch_status
            return fn(model, cur, pos, next_token=forced, attn_top_k=top_k, **kw)

        monkeypatch.setattr(G, "_StatusWatch", Watch)
        monkeypatch.setattr(G, "_recover_token", recover)
        pos = torch.tensor([40], dtype=torch.int32)
        toks, _ = G.decode_n_tokens(None, torch.tensor([[3]], dtype=torch.int32), pos, step, n, recover=fail_at is not None or None,
                                    terminator_ids=terminators)
        return [int(t) for t in toks], int(pos[0])

    clean, end = run(None, 0)
    assert end == 52 and len(clean) == 12
    for fail_at in (0, 1, 5, 11):
        for lag in (0, 1, 3, 20):
            got, e = run(fail_at, lag)
            assert got == clean and e == end, (fail_at, lag, got, clean, e)
    # the host runs further ahead than the ring is deep (graph replay, no terminators): the verdict taken early to free a slot must
    # not be lost (it was — a failure seen only on token f + 1 committed token f's garbage)
    for depth in (1, 2, 4):
        for fail_at in (0, 3, 7, 11):
            for lag in (depth, depth + 1, 3 * depth, 40):
                got, e = run(fail_at, lag, depth=depth)
                assert got == clean and e == end, (depth, fail_at, lag, got, clean, e)
    # a terminator behind the failed token: the rewound run stops where the clean one does, without the position bump of the stop
    stop = clean[6]
    c2, e2 = run(None, 0, terminators=[stop])
    assert c2 == clean[:7] and e2 == 46
    for lag in (0, 2, 9):
        g2, e3 = run(4, lag, terminators=[stop])
        assert g2 == c2 and e3 == e2, (lag, g2, e3)
