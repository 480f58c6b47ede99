"""KV-cache policies behind the reference's `cache.py` class surface, executed by hand-written HIP kernels.

Drop-in for the consumer call sites listed in SURVEY.md §8(b): `get_cache_constructor`, `add_cache_arguments`,
`cache_compatibility`, and `KVCache*` modules exposing `update_kv / update_state / return_attn / reset /
compute_statistics` plus the buffers `k_cache, v_cache, pos, mask, cache_cts` (same names, shapes and dtypes
as ref: cache.py:178-227).  All device work goes through the C ABI in include/coldcompress.h via
`_abi.call`; there is no CPU or eager-PyTorch fallback — tensors must live on the ROCm device.

Not yet covered (raised loudly): `cache_bits`, `history_window_size > 1`,
`attn_thresholding`, `hybrid`, `debug_*`.
"""
import argparse
import ctypes as C
import re

import torch
import torch.nn as nn

from . import _abi
from ._abi import ColdCompressError, KVView
from .prompt_compression import AttnSummary, get_prompt_compressor_constructor  # noqa: F401  (re-export like the reference)

_DT = {torch.float32: _abi.CC_DT_F32, torch.bfloat16: _abi.CC_DT_BF16, torch.float16: _abi.CC_DT_F16}

STRATEGIES = ["full", "random", "recent_global", "heavy_hitter", "l2", "hybrid", "keep_it_odd"]


def add_cache_arguments(parser: argparse.ArgumentParser):
    """Same flags, defaults and choices as ref: cache.py:13-118 (the CLI is the compatibility surface)."""
    g = parser.add_argument_group("cache_args")
    g.add_argument("--max_cache_length", type=float, default=[1.0], nargs="+",
                   help="Cache size per layer: a fraction of |prompt|+max_new_tokens if in (0,1], else an absolute size.")
    g.add_argument("--cache_bits", default=None, type=int, choices=[2, 4, 8], help="Quantize the cache.")
    g.add_argument("--cache_length_pattern", default="tile", choices=["tile", "repeat", "funnel", "pyramid"])
    g.add_argument("--cache_strategy", default=["full"], nargs="+",
                   choices=STRATEGIES + [f"debug_{s}" for s in STRATEGIES])
    g.add_argument("--cache_strategy_pattern", default="tile", choices=["tile", "repeat"])
    parser.add_argument("--feed_long_prompts", default=False, action="store_true")
    g.add_argument("--prompt_compression_strategy", default=["recent_global"], nargs="+")
    g.add_argument("--global_tokens", default=1, type=int)
    g.add_argument("--recent_window", default=10, type=float)
    g.add_argument("--history_window_size", default=1, type=int)
    g.add_argument("--attn_thresholding", default=False, action="store_true")
    parser.add_argument("--min_recovery_frac", default=0.9, type=float)


def cache_compatibility(args):
    """ref: cache.py:121-139."""
    for length, cache_strat, prompt_strat in zip(args.max_cache_length, args.cache_strategy,
                                                 args.prompt_compression_strategy):
        if cache_strat == "heavy_hitter":
            assert prompt_strat == "heavy_hitter", \
                "heavy_hitter cache needs --prompt_compression_strategy heavy_hitter (it consumes attention)."
        if cache_strat == "hybrid":
            assert not getattr(args, "compile", False), "hybrid is not supported with compile=True."
        if cache_strat in {"full", "hybrid"}:
            assert length == 1.0, f"{cache_strat} cache strategy only supports max_cache_length=1.0."
    print("The cache argument values you provided appear compatible with each other!")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_device(t, what):
    if not t.is_cuda:
        raise ColdCompressError(
            f"{what} is on {t.device}: the cold-compress HIP path needs ROCm device tensors (there is no CPU fallback).")


class _Scratch:
    """Per-cache device scratch kept OUT of the module's buffers so `cache_memory_gb` matches the reference."""

    def __init__(self):
        self.t = {}

    def get(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype, str(device))
        v = self.t.get(key)
        if v is None:
            v = torch.empty(shape, dtype=dtype, device=device)
            self.t[key] = v
        return v


class KVCache(nn.Module):
    # ref: cache.py:155-160
    relevant_kwargs = ["max_cache_length", "global_tokens", "max_seq_length", "cache_bits"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, head_specific=False,
                 variable_length=False, **kwargs):
        super().__init__()
        for key, value in kwargs.items():
            setattr(self, key, value)
        if max_batch_size != 1:
            raise ColdCompressError("batch size is fixed at 1 (ref: model.py:188-189)")
        if getattr(self, "cache_bits", None) is not None:
            raise NotImplementedError("cache_bits (quantised KV) is a SURVEY §8(f) follow-up and not built yet")
        if dtype not in _DT:
            raise ColdCompressError(f"unsupported cache dtype {dtype}")
        self.cache_bits = None
        self.n_bit = None
        self.n_heads = n_heads
        self.head_dim = head_dim
        self.head_specific = head_specific
        self.variable_length = variable_length
        self.cache_shape = (1, n_heads, self.max_cache_length, head_dim)
        S = self.max_cache_length
        self.register_buffer("k_cache", torch.zeros(self.cache_shape, dtype=dtype))
        self.register_buffer("v_cache", torch.zeros(self.cache_shape, dtype=dtype))
        self.register_buffer("pos", torch.full((1, n_heads if head_specific else 1, S), -1, dtype=torch.int32))
        self.register_buffer("cache_cts", torch.zeros((n_heads if variable_length else 1,), dtype=torch.int32))
        self.register_buffer("mask", torch.zeros((1, n_heads, 1, S), dtype=torch.bool))
        self._scratch = _Scratch()
        self._view_cache = None

    # ------------------------------------------------------------------ bookkeeping (ref: cache.py:229-281)
    def reset(self):
        self.k_cache.zero_()
        self.v_cache.zero_()
        self.mask.zero_()
        self.cache_cts.zero_()
        self.pos.fill_(-1)

    def return_attn(self):
        return False

    def memory_usage(self):
        tensors = [b for b in self._buffers.values() if torch.is_tensor(b)]
        for obj in vars(self).values():
            if torch.is_tensor(obj):
                tensors.append(obj)
        return sum(t.element_size() * t.numel() for t in tensors) / (1024 ** 3)

    def compression_ratio(self, seq_len):
        n = seq_len - 1  # the final token never reaches the cache
        assert torch.all(self.cache_cts <= self.max_cache_length)
        size = self.cache_cts.clone().float()
        if self.n_bit is not None:
            size *= self.n_bit / 16.0
        return ((n - size) / n).mean()

    def compute_statistics(self, seq_len):
        return {"compression_ratio": self.compression_ratio(seq_len).item(), "cache_memory_gb": self.memory_usage()}

    def return_kv_cache(self):
        return self.k_cache, self.v_cache, self.mask

    # ------------------------------------------------------------------ ABI plumbing
    def _view(self):
        key = (self.k_cache.data_ptr(), self.v_cache.data_ptr(), self.pos.data_ptr(), self.mask.data_ptr(),
               self.cache_cts.data_ptr())
        if self._view_cache is None or self._view_cache[0] != key:
            _need_device(self.k_cache, "k_cache")
            H, S, D = self.n_heads, self.max_cache_length, self.head_dim
            v = KVView(key[0], key[1], key[2], key[3], key[4], H, self.pos.shape[1], self.cache_cts.shape[0], S, D,
                       _DT[self.k_cache.dtype])
            self._view_cache = (key, v)
        return C.byref(self._view_cache[1])

    def _idx_buf(self):
        return self._scratch.get("idx", (self.pos.shape[1],), torch.int64, self.k_cache.device)

    @staticmethod
    def _pos32(input_pos):
        return input_pos if input_pos.dtype == torch.int32 else input_pos.to(torch.int32)

    def _new_rows(self, k_val, v_val):
        H, D = self.n_heads, self.head_dim
        if k_val.dtype != self.k_cache.dtype:
            raise ColdCompressError(f"k/v dtype {k_val.dtype} != cache dtype {self.k_cache.dtype}")
        _need_device(k_val, "k_val")
        return k_val.reshape(H, D).contiguous(), v_val.reshape(H, D).contiguous()

    # ------------------------------------------------------------------ update_kv (ref: cache.py:314-340)
    def update_kv(self, input_pos, k_val, v_val, is_prefill, **kwargs):
        if is_prefill:
            self._prefill_update(input_pos, k_val, v_val, **kwargs)
        else:
            self._decoding_update(input_pos, k_val, v_val, **kwargs)
        return self.return_kv_cache()

    def update_state(self, *args, **kwargs):
        pass

    def _prefill_update(self, input_pos, k_val, v_val, **kwargs):
        """ref: cache.py:381-401 — slots [0, T) <- the (possibly compacted) prompt; pos stores original positions."""
        _need_device(k_val, "k_val")
        H, D = self.n_heads, self.head_dim
        T = input_pos.shape[-1]
        assert k_val.shape[2] == T == v_val.shape[2]
        if T > self.max_cache_length:
            raise ColdCompressError(f"prefill of {T} tokens exceeds max_cache_length={self.max_cache_length}")
        k = k_val.reshape(H, T, D).contiguous() if k_val.is_contiguous() else k_val.contiguous().view(H, T, D)
        v = v_val.reshape(H, T, D).contiguous() if v_val.is_contiguous() else v_val.contiguous().view(H, T, D)
        p = input_pos.to(torch.int64).reshape(-1, T).contiguous()
        _abi.call("cc_prefill_fill", self._view(), _ptr(k), _ptr(v), _ptr(p), p.shape[0], T, _stream())

    def _decoding_update(self, input_pos, k_val, v_val, **kwargs):
        """ref: cache.py:348-364: generic path = `_token_importances` -> base rules -> arg-min -> insert."""
        k, v = self._new_rows(k_val, v_val)
        self._run_select(input_pos, k, v)

    def _eviction_idx(self, input_pos):
        """ref: cache.py:366-379 — selection only (no insert); returns int64 [Hp]."""
        self._run_select(input_pos, None, None)
        return self._idx_buf().clone()

    def _run_select(self, input_pos, k, v):
        scores = self._token_importances(input_pos)
        if scores.ndim == 1:
            scores = scores.unsqueeze(0)
        scores = scores.contiguous()
        _abi.call("cc_decode_update_scores", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)), _ptr(scores),
                  _DT[scores.dtype], int(self.global_tokens), _ptr(self._idx_buf()), _stream())

    def _token_importances(self, input_pos):
        raise NotImplementedError


class KVCacheHeadConstant(KVCache):
    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, head_specific=False, **kwargs)


class KVCacheHeadSpecific(KVCache):
    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, variable_length=False, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, head_specific=True, variable_length=variable_length,
                         **kwargs)


class KVCacheFull(KVCacheHeadConstant):
    """ref: cache.py:493-502."""

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        self.global_tokens = 0
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **kwargs)

    def _run_select(self, input_pos, k, v):
        _abi.call("cc_decode_update_full", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  _ptr(self._idx_buf()), _stream())


class KVCacheRandom(KVCacheHeadConstant):
    """ref: cache.py:505-524.  The uniform vector is drawn by torch on the device (RNG streams are
    backend-specific); `_rand` is the injection point tests use to replay the reference's draws."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "recent_window"]

    def _rand(self):
        return torch.rand(self.max_cache_length, device=self.k_cache.device)

    def _run_select(self, input_pos, k, v):
        r = self._rand().to(torch.float32).contiguous()
        _abi.call("cc_decode_update_random", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)), _ptr(r),
                  int(self.global_tokens), int(self.recent_window), _ptr(self._idx_buf()), _stream())


class KVCacheRecentGlobal(KVCacheHeadConstant):
    """ref: cache.py:527-556 (ring buffer behind the global sink tokens)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens"]

    def _run_select(self, input_pos, k, v):
        _abi.call("cc_decode_update_recent_global", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  int(self.global_tokens), _ptr(self._idx_buf()), _stream())


class KVCacheL2(KVCacheHeadSpecific):
    """ref: cache.py:559-612."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "recent_window"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **kwargs)
        self.register_buffer("key_norm", torch.zeros((1, n_heads, self.max_cache_length), dtype=dtype))

    def reset(self):
        super().reset()
        self.key_norm.zero_()

    def _run_select(self, input_pos, k, v):
        _abi.call("cc_decode_update_l2", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  _ptr(self.key_norm), int(self.global_tokens), int(self.recent_window), _ptr(self._idx_buf()), None, 0,
                  _stream())

    def update_state(self, input_pos, k_val, v_val, is_prefill, attn, **kwargs):
        if is_prefill:  # ref: cache.py:611-612 — norms of every slot of the freshly filled cache
            _abi.call("cc_row_l2_norm", _ptr(self.k_cache), self.n_heads, self.max_cache_length, self.head_dim,
                      _DT[self.k_cache.dtype], 0, _ptr(self.key_norm), _stream())


class KVCacheHeavyHitter(KVCacheHeadSpecific):
    """ref: cache.py:615-765 (ScissorHands / H2O style accumulated attention), history_window_size == 1."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "history_window_size",
                       "recent_window", "attn_thresholding"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, variable_length=False, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, variable_length, **kwargs)
        if self.attn_thresholding or self.history_window_size != 1:
            raise NotImplementedError(
                "attn_thresholding / history_window_size > 1 are SURVEY §8(f) follow-ups and not built yet")
        S = self.max_cache_length
        self.register_buffer("attn_history_num", torch.zeros((1, n_heads, S, 1), dtype=torch.float64))
        self.register_buffer("attn_history_denom", torch.zeros((1, n_heads, S), dtype=torch.int32))
        self.register_buffer("attn_counter", torch.zeros((1,), dtype=torch.int64))
        # set by the attention op when it already applied this step's history update in its combine pass
        self._state_fused = False

    def reset(self):
        super().reset()
        self.attn_history_num.zero_()
        self.attn_history_denom.zero_()
        self.attn_counter.zero_()

    def return_attn(self) -> bool:
        return True

    def _run_select(self, input_pos, k, v):
        _abi.call("cc_decode_update_heavy_hitter", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  _ptr(self.attn_history_num), _ptr(self.attn_history_denom), int(self.global_tokens),
                  int(self.recent_window), _ptr(self._idx_buf()), _stream())

    def fused_history(self):
        """Pointers the decode attention kernel needs to fold cache.py:690-723 into its combine pass."""
        return self.attn_history_num, self.attn_history_denom, self.attn_counter

    def _apply(self, attn_hs, T):
        _abi.call("cc_hh_update", _ptr(self.attn_history_num), _ptr(self.attn_history_denom), _ptr(self.attn_counter),
                  _ptr(attn_hs), self.n_heads, self.max_cache_length, T, _DT[self.k_cache.dtype], _stream())

    def update_state(self, input_pos, k_val, v_val, is_prefill, attn, **kwargs):
        """ref: cache.py:690-723."""
        if self._state_fused:
            self._state_fused = False
            return
        H, dt = self.n_heads, self.k_cache.dtype
        if isinstance(attn, AttnSummary):  # our prefill kernel's side output: column sums, never [L, L]
            attn = attn.column_mean(input_pos)
        elif is_prefill and attn.ndim == 4:  # a materialised [1, H, L, L] tensor from a reference-style caller
            _need_device(attn, "attn")
            L = attn.shape[-1]
            colsum = self._scratch.get("colsum", (H, L), torch.float32, attn.device)
            _abi.call("cc_attn_colsum", _ptr(attn.contiguous()), H, attn.shape[-2], L, _DT[attn.dtype], _ptr(colsum), _stream())
            mean = torch.empty((1, H, L), dtype=dt, device=attn.device)
            ip = input_pos.to(torch.int64).contiguous()
            _abi.call("cc_colsum_to_mean", _ptr(colsum), _ptr(ip), H, L, _DT[dt], _ptr(mean), _stream())
            attn = mean
        _need_device(attn, "attn")
        if attn.dtype != dt:
            raise ColdCompressError(f"attention dtype {attn.dtype} != cache dtype {dt}")
        a = attn.reshape(H, -1).contiguous()
        T = a.shape[1]
        if T > self.max_cache_length:
            raise ColdCompressError("attention longer than the cache")
        self._apply(a, T)


class KVCacheKeepItOdd(KVCacheHeadConstant):
    """ref: cache.py:1423-1441 (toy policy; exercises the generic caller-supplied-importances path)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "recent_window"]

    def _token_importances(self, input_pos):
        p = self.pos[:, 0]
        scores = torch.zeros_like(p, dtype=torch.bfloat16)
        scores[p % 2 == 1] = 1.0
        scores[p >= input_pos - self.recent_window] = float("inf")
        return scores


def get_cache_constructor(cache_strategy):
    """ref: cache.py:1444-1478 -> (constructor, relevant_kwargs)."""
    table = {
        "full": KVCacheFull,
        "l2": KVCacheL2,
        "random": KVCacheRandom,
        "recent_global": KVCacheRecentGlobal,
        "heavy_hitter": KVCacheHeavyHitter,
        "keep_it_odd": KVCacheKeepItOdd,
    }
    if cache_strategy in table:
        cls = table[cache_strategy]
        return cls, cls.relevant_kwargs
    if cache_strategy == "hybrid" or cache_strategy.startswith("debug"):
        name = re.sub(r"debug_+", "", cache_strategy).strip()
        if name in table or name == "hybrid":
            raise NotImplementedError(f"cache strategy '{cache_strategy}' is a SURVEY §8 follow-up and not built yet")
    raise ValueError(f"Invalid cache strategy: {cache_strategy}")
