"""Cache-budget arithmetic and the generate loop (the build's counterpart of the reference's
generation_utils.py:224-531), with the decode step optionally captured in a hipGraph instead of being traced
by torch.compile (generation_utils.py:578-594).

Budget helpers are exact-integer restatements checked against tests/golden/f8_budgets.json.
"""
import time

import torch

from .model import Transformer, find_multiple


# ------------------------------------------------------------------------------ budgets (ref: :224-321)
def normalize_cache_length(max_cache_length: float, max_seq_length: int, multiple_of: int = 8) -> int:
    """ref: generation_utils.py:260-276."""
    if 0 < max_cache_length <= 1:
        n = round(max_seq_length * max_cache_length)
    else:
        assert int(max_cache_length) == max_cache_length
        n = int(max_cache_length)
        if n > max_seq_length:
            print(f"FYI: max_cache_length ({n}) is greater than max_seq_length ({max_seq_length}). Setting to {max_seq_length}")
            n = max_seq_length
    return min(find_multiple(n, multiple_of), max_seq_length)


def apply_pyramid_pattern(max_cache_length, max_seq_length, model_n_layer, decreasing=True, min_cache_length=256):
    """ref: generation_utils.py:279-321 (PyramidKV, beta = 14)."""
    beta = 14
    floor = min(min_cache_length, max_cache_length)
    total = max_cache_length * model_n_layer
    lo, hi = total / (model_n_layer * beta), 2 * total / model_n_layer
    step = (hi - lo) / model_n_layer
    lens = [lo] + [lo + step * i for i in range(1, model_n_layer - 1)] + [hi]
    lens = [normalize_cache_length(int(x), max_seq_length) for x in lens]
    overflow = n_over = 0
    for i, x in enumerate(lens):
        if x < floor:
            overflow += floor - x
            lens[i] = floor
            n_over += 1
    if n_over < len(lens):
        dec = overflow // (len(lens) - n_over)
        lens = [max(floor, x - dec) if x > floor else x for x in lens]
    if decreasing:
        lens = lens[::-1]
        assert lens[-1] < lens[0], "Cache lengths should be decreasing."
    else:
        assert lens[0] < lens[-1], "Cache lengths should be increasing."
    return lens


def apply_pattern(pattern, out_size, extension_strategy="tile", max_seq_length=None):
    """ref: generation_utils.py:224-257."""
    assert extension_strategy in {"tile", "repeat", "pyramid", "funnel"}
    assert out_size % len(pattern) == 0, f"{len(pattern)} must be a divisible factor of the number of layers ({out_size})."
    factor = out_size // len(pattern)
    if extension_strategy in {"funnel", "pyramid"}:
        assert len(pattern) == 1, "Funnel and pyramid patterns must have a single element."
        return apply_pyramid_pattern(pattern[0], max_seq_length, out_size, decreasing=extension_strategy == "pyramid")
    if extension_strategy == "tile":
        return [x for x in pattern for _ in range(factor)]
    return list(pattern) * factor


def setup_caches(model: Transformer, tokenizer, device, max_seq_length: int, cache_kwargs: dict) -> dict:
    """ref: generation_utils.py:324-388."""
    ck = cache_kwargs
    ck["max_seq_length"] = max_seq_length
    ck["max_cache_length"] = [normalize_cache_length(x, max_seq_length) for x in ck["max_cache_length"]]
    ck["max_cache_length"] = apply_pattern(ck["max_cache_length"], model.config.n_layer, ck["cache_length_pattern"],
                                           max_seq_length)
    assert len(ck["cache_strategy"]) == len(ck["prompt_compression_strategy"]), \
        "You must specify a prompt_compression_strategy for each cache_strategy."
    ck["cache_strategy"] = apply_pattern(ck["cache_strategy"], model.config.n_layer, ck["cache_strategy_pattern"])
    ck["prompt_compression_strategy"] = apply_pattern(ck["prompt_compression_strategy"], model.config.n_layer,
                                                      ck["cache_strategy_pattern"])
    if not isinstance(ck["recent_window"], list):
        rw = ck["recent_window"]
        if rw <= 1:
            ck["recent_window"] = [max(1, int(rw * n)) for n in ck["max_cache_length"]]
        else:
            ck["recent_window"] = [max(1, min(rw, n)) for n in ck["max_cache_length"]]
    assert ck["global_tokens"] <= min(ck["max_cache_length"]), "Global tokens must be less than max_cache_length."
    if ck["cache_strategy"][0] == "hybrid":
        ck["token_ids"] = {"special": tokenizer.special_ids(), "punctuation": tokenizer.punctuation_ids()}
    with torch.device(device):
        model.setup_caches(max_batch_size=1, **ck)
    return ck


# ------------------------------------------------------------------------------ sampling / steps
def greedy(logits, next_token):
    """ref: generation_utils.py:136-142."""
    row = logits[0, -1]
    if row.is_cuda and next_token is None:
        from . import glue

        probs, idx_next = glue.softmax_argmax(row)  # one launch instead of softmax + arg-max reduce
        return idx_next, probs
    probs = torch.nn.functional.softmax(row, dim=-1)
    idx_next = torch.argmax(probs, keepdim=True).to(dtype=torch.int) if next_token is None else next_token
    return idx_next, probs


def prefill(model, x, input_pos, next_token=None, **_):
    """ref: generation_utils.py:145-160 — the causal mask is implicit in the HIP prefill kernel."""
    logits = model(x, input_pos, mask=None, is_prefill=True)
    return greedy(logits, next_token)


def decode_one_token(model, x, input_pos, next_token=None, attn_top_k=1.0, **_):
    """ref: generation_utils.py:163-178."""
    logits = model(x, input_pos, is_prefill=False, attn_top_k=attn_top_k)
    return greedy(logits, next_token)


class GraphedDecoder:
    """One decode step captured in a hipGraph (HIP-native replacement for the reference's
    `torch.compile(mode="reduce-overhead")`, generation_utils.py:581-587).  The token and position live in
    static device tensors; every HIP entry point reads `input_pos` from device memory, so replays advance."""

    def __init__(self, model, warmup=2):
        self.model = model
        dev = model.output.weight.device
        self.tok = torch.zeros((1, 1), dtype=torch.int32, device=dev)
        self.pos = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.graph = None
        self.out_tok = self.out_probs = None
        self.warmup = warmup

    def capture(self):
        """Capture must start from a state where one extra (discarded) step is harmless: run it on a snapshot."""
        caches = [l.attention.kv_cache for l in self.model.layers]
        for c in caches:  # seed the fused decode-step pipelines BEFORE the snapshot so the restored state is runnable
            if hasattr(c, "prepare_decode") and c.supports_fused_step() and not c._next_valid:
                c.prepare_decode(self.pos)
        snap = [{k: v.clone() for k, v in c._buffers.items()} for c in caches]
        flags = [(getattr(c, "_next_valid", None), getattr(c, "_quant_pending", False)) for c in caches]
        pos0 = self.pos.clone()
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(self.warmup):  # positions advance like a real decode (the pipeline assumes +1 steps)
                    decode_one_token(self.model, self.tok, self.pos)
                    self.pos += 1
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.out_tok, self.out_probs = decode_one_token(self.model, self.tok, self.pos)
            self.graph = graph
        finally:  # also when capture is refused (e.g. a collective that cannot be captured): the caller falls back to
            self.pos.copy_(pos0)  # eager launches and must find the state it handed in
            for c, sn, fl in zip(caches, snap, flags):
                for k, v in sn.items():
                    c._buffers[k].copy_(v)
                if fl[0] is not None:
                    c._next_valid = fl[0]
                c._quant_pending = fl[1]
                if hasattr(c, "_ring_version"):  # ring and tracked window sums were restored together: still in step
                    c._ring_version = c._ring_tag()

    def __call__(self, model, x, input_pos, next_token=None, **_):
        if self.graph is None:
            self.tok.copy_(x)
            self.pos.copy_(input_pos)
            self.capture()
        self.tok.copy_(x)
        self.pos.copy_(input_pos)
        self.graph.replay()
        if next_token is not None:
            return next_token, self.out_probs
        return self.out_tok, self.out_probs


def _recover_token(model, cur_token, input_pos, decode_fn, nt, npb, forced, attn_top_k, kw, max_retries=6):
    """In-band recovery from a single-launch hand-off that could not complete (a launch whose workgroups were not all resident: a
    co-tenant kernel on the device).  One 4-byte read of the decode workspace's status word per token; when it is set the failed
    step committed nothing of its kv head, every later launch of the token returned at once (include/coldcompress.h,
    cc_decode_step_heavy_hitter_rc) — so the word is cleared, the epoch words are advanced, and the SAME token runs again: heads
    whose step is committed replay it (attention only), the others step.  Caches whose step carries no commit words (`recoverable()`
    False: l2, hybrid, history windows, the fused uint8 mode, random with an injected vector) and a failure that persists raise."""
    from ..attention_utils import check_single_launch_status, reset_single_launch_status, single_launch_status

    dev = cur_token.device
    tries = 0
    while single_launch_status(dev):
        caches = [l.attention.kv_cache for l in model.layers]
        ok = all(callable(getattr(c, "recoverable", None)) and c.recoverable() for c in caches)
        if not ok or tries >= max_retries:
            check_single_launch_status(dev)  # raises (and clears the word)
        reset_single_launch_status(dev)
        tries += 1
        time.sleep(0.05 * tries)  # whatever shared the device gets a moment to leave
        nt, npb = decode_fn(model, cur_token, input_pos, next_token=forced, attn_top_k=attn_top_k, **kw)
    return nt, npb


def decode_n_tokens(model, cur_token, input_pos, decode_one_token, num_new_tokens, terminator_ids=None, attn_top_k=1.0,
                    prefix=None, **kw):
    """ref: generation_utils.py:181-217."""
    new_tokens, new_probs = [], []
    recover = bool(kw.pop("recover", True)) and cur_token.is_cuda
    for i in range(num_new_tokens):
        teacher_force = prefix is not None and i < len(prefix)
        nt = prefix[i].view(1) if teacher_force else None
        nt, npb = decode_one_token(model, cur_token, input_pos, next_token=nt, attn_top_k=attn_top_k, **kw)
        if recover:
            nt, npb = _recover_token(model, cur_token, input_pos, decode_one_token, nt, npb, prefix[i].view(1) if teacher_force else None,
                                     attn_top_k, kw)
        new_tokens.append(nt.clone())
        new_probs.append(npb.clone())
        if terminator_ids and nt in terminator_ids and not teacher_force:
            break
        input_pos += 1
        cur_token = nt.view(1, -1)
    return new_tokens, new_probs


@torch.no_grad()
def generate(model, prompt, prefill, decode_one_token, max_new_tokens, next_tokens=None, terminator_ids=None,
             feed_long_prompts=False, decode_first_token=False, attn_top_k=1.0, **kw):
    """ref: generation_utils.py:399-531 (prompt-splitting rules, teacher forcing, perf stats).  Unlike the
    reference, the prefill timer is closed after a device sync (SURVEY §5 note)."""
    prompt_length = prompt.size(0)
    device, dtype = prompt.device, prompt.dtype
    min_cache_length = model.min_cache_length()
    max_prompt_len = min_cache_length - 1
    prefix = None
    if (feed_long_prompts and prompt_length > max_prompt_len) or prompt_length == min_cache_length:
        prompt, prefix = prompt[:max_prompt_len], prompt[max_prompt_len:]
        max_new_tokens += len(prefix)
        prompt_length = max_prompt_len
    if decode_first_token:
        prompt, prefix = prompt[:-1], prompt[-1:]
        max_new_tokens += 1
        prompt_length -= 1
    if next_tokens is not None:
        max_new_tokens = len(next_tokens)
        next_token, prefix = next_tokens[0].view(1), next_tokens[1:]
    elif prefix is not None:
        next_token, prefix = prefix[0].view(1), prefix[1:]
    else:
        next_token = prefix = None
    seq = torch.full((prompt_length + max_new_tokens,), -1, dtype=dtype, device=device)
    seq[:prompt_length] = prompt
    input_pos = torch.arange(0, prompt_length, device=device)

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    sync()
    t0 = time.perf_counter()
    ret = prefill(model, prompt.view(1, -1), input_pos, next_token=next_token, **kw)
    sync()
    t1 = time.perf_counter()
    next_token, next_tok_probs = ret[0].clone(), ret[1].clone()
    seq[prompt_length] = next_token
    input_pos = torch.tensor([prompt_length], device=device, dtype=torch.int)
    toks, tok_probs = decode_n_tokens(model, next_token.view(1, -1), input_pos, decode_one_token, max_new_tokens - 1,
                                      terminator_ids=terminator_ids, prefix=prefix, attn_top_k=attn_top_k, **kw)
    sync()
    t2 = time.perf_counter()
    if device.type == "cuda":  # fail loudly: a single-launch step that timed out leaves a word in the decode workspace
        from ..attention_utils import check_single_launch_status

        check_single_launch_status(device)
        from ..tp import check_oneshot_allreduce_status

        check_oneshot_allreduce_status()  # (a collective under tensor parallelism: every rank passes here)
    decode_tokens = len(toks) + 1
    stats = {
        "prefill_tokens": prompt_length, "decode_tokens": decode_tokens,
        "prefill_toks_per_sec": prompt_length / (t1 - t0), "decode_toks_per_sec": decode_tokens / (t2 - t1),
        "total_toks_per_sec": decode_tokens / (t2 - t0), "total_seconds": t2 - t0, "prefill_seconds": t1 - t0,
        "decode_seconds": t2 - t1, "decode_seconds_frac_of_total": (t2 - t1) / (t2 - t0),
        "memory_used_gb": (torch.cuda.max_memory_reserved() / 1e9) if device.type == "cuda" else 0.0,
    }
    if toks:
        seq[prompt_length + 1: prompt_length + 1 + len(toks)] = torch.cat(toks)
    if -1 in seq:
        seq = seq[: torch.where(seq == -1)[0][0]]
    return seq, [next_tok_probs] + tok_probs, stats
