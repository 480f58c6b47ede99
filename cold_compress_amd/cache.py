"""KV-cache policies behind the reference's `cache.py` class surface, executed by hand-written HIP kernels.

Drop-in for the consumer call sites listed in SURVEY.md §8(b): `get_cache_constructor`, `add_cache_arguments`,
`cache_compatibility`, and `KVCache*` modules exposing `update_kv / update_state / return_attn / reset /
compute_statistics` plus the buffers `k_cache, v_cache, pos, mask, cache_cts` (same names, shapes and dtypes
as ref: cache.py:178-227).  All device work goes through the C ABI in include/coldcompress.h via
`_abi.call`; there is no CPU or eager-PyTorch fallback — tensors must live on the ROCm device.

Raised loudly because the reference itself fails there (nothing to pin): `attn_thresholding`, `l2` or `hybrid` with
`cache_bits`.  Everything else of SURVEY §8 (a)/(f) is implemented, including `cache_bits`, the `history_window_size > 1`
ring, `hybrid` and the `debug_*` analysis wrapper.
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


def add_extension_arguments(parser: argparse.ArgumentParser):
    """Flags the reference does not have (kept out of add_cache_arguments so that its flag set stays the reference's).
    --cache_quant_mode fused: with --cache_bits 8, uint8 images on a per-(head, slot) grid, dequantised inside the decode
    kernels — a different numerical contract (include/coldcompress.h); "reference" = cache.py:283-338 bit for bit."""
    parser.add_argument("--cache_quant_mode", default="reference", choices=["reference", "fused"])


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


_FUSED_SCRATCH = {}  # (H, S, D, dtype, device) -> staging pair of the fused quantised cache's prefill


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
        if dtype not in _DT:
            raise ColdCompressError(f"unsupported cache dtype {dtype}")
        # ref: cache.py:180-183 — quantised KV: 8 / 4 / 2 bits, one (scale, zero point) per cache slot (axis 2)
        self.cache_bits = getattr(self, "cache_bits", None)
        if self.cache_bits not in (None, 8, 4, 2):
            raise ColdCompressError("Only 2-bit, 4-bit, and 8-bit quantization are supported (ref: quantization_utils.py:5)")
        self.quantize = self.cache_bits is not None
        self.n_bit = self.cache_bits
        self.quantization_axis = 2
        # opt-in fused quantised cache (our extension, include/coldcompress.h): the decode kernels stream uint8 images
        mode = getattr(self, "cache_quant_mode", None) or "reference"
        if mode not in ("reference", "fused"):
            raise ColdCompressError(f"cache_quant_mode={mode!r}: 'reference' or 'fused'")
        self.fused_quant = self.quantize and mode == "fused"
        if self.fused_quant:
            if self.n_bit != 8 or dtype not in (torch.bfloat16, torch.float16) or head_dim != 128:
                raise ColdCompressError("cache_quant_mode='fused' serves cache_bits=8, 16-bit models, head_dim 128")
            if not self._fused_quant_policy():
                raise ColdCompressError(f"cache_quant_mode='fused' is not available for {type(self).__name__}: it serves "
                                        "heavy_hitter (history_window_size 1), recent_global, full and random")
            self.quantize = False  # none of the reference mode's round-trip machinery runs
        self.n_heads = n_heads
        self.head_dim = head_dim
        self.head_specific = head_specific
        self.variable_length = variable_length
        self.cache_shape = (1, n_heads, self.max_cache_length, head_dim)
        S = self.max_cache_length
        if self.fused_quant:
            # the images ARE the cache; k_cache / v_cache stay as empty tensors that carry the model dtype
            self.register_buffer("k_cache", torch.zeros((1, n_heads, 0, head_dim), dtype=dtype))
            self.register_buffer("v_cache", torch.zeros((1, n_heads, 0, head_dim), dtype=dtype))
            self.register_buffer("k_cache_q", torch.zeros(self.cache_shape, dtype=torch.uint8))
            self.register_buffer("v_cache_q", torch.zeros(self.cache_shape, dtype=torch.uint8))
            self.register_buffer("kv_qparams", torch.zeros((1, n_heads, S, 4), dtype=torch.float32))  # k_scale, k_min, v_scale, v_min
        else:
            self.register_buffer("k_cache", torch.zeros(self.cache_shape, dtype=dtype))
            self.register_buffer("v_cache", torch.zeros(self.cache_shape, dtype=dtype))
        self.register_buffer("pos", torch.full((1, n_heads if head_specific else 1, S), -1, dtype=torch.int32))
        self.register_buffer("cache_cts", torch.zeros((n_heads if variable_length else 1,), dtype=torch.int32))
        self.register_buffer("mask", torch.zeros((1, n_heads, 1, S), dtype=torch.bool))
        if self.quantize:
            # k_cache / v_cache stay the model-dtype WORKING cache every kernel reads (what the reference's
            # dequantize_cache() would produce); the image the reference holds between updates lives beside it.
            if head_dim % (8 // self.n_bit):
                raise ColdCompressError("head_dim must be a multiple of the values packed per byte")
            qshape = self.cache_shape if self.n_bit == 8 else (n_heads * S * head_dim * self.n_bit // 8,)
            qdt = torch.int8 if self.n_bit == 8 else torch.uint8
            for name in ("k", "v"):
                self.register_buffer(f"{name}_cache_q", torch.zeros(qshape, dtype=qdt))
                self.register_buffer(f"{name}_scales", torch.zeros((S,), dtype=dtype))
                self.register_buffer(f"{name}_zero_points", torch.zeros((S,), dtype=dtype))
            # which slots the round trip no longer changes, and the positions they held when that was established
            self.register_buffer("_quant_stable", torch.zeros((2, S), dtype=torch.uint8), persistent=False)
            self.register_buffer("_quant_pos_seen", torch.zeros((2, n_heads if head_specific else 1, S), dtype=torch.int32),
                                 persistent=False)
            self._quant_tag = None
        # the constructor quantises the zero cache (cache.py:188-197): done by the first flush
        self._quant_pending = self.quantize
        self._scratch = _Scratch()
        self._view_cache = None

    # ------------------------------------------------------------------ bookkeeping (ref: cache.py:229-281)
    def reset(self):
        self.k_cache.zero_()
        self.v_cache.zero_()
        if self.fused_quant:
            self.k_cache_q.zero_()
            self.v_cache_q.zero_()
            self.kv_qparams.zero_()
        self.mask.zero_()
        self.cache_cts.zero_()
        self.pos.fill_(-1)
        self._quant_pending = self.quantize

    # ------------------------------------------------------------------ fused quantised cache (our extension)
    def _fused_quant_policy(self):
        """Policy code of cc_decode_step_quant this class runs under cache_quant_mode='fused' (0: not available)."""
        return 0

    def _fused_scratch(self):
        """Model-dtype [H, S, D] staging pair shared by every layer of the same shape (prefill fill -> row quantisation)."""
        key = (self.n_heads, self.max_cache_length, self.head_dim, self.k_cache.dtype, str(self.k_cache_q.device))
        pair = _FUSED_SCRATCH.get(key)
        if pair is None:
            shape = (self.n_heads, self.max_cache_length, self.head_dim)
            pair = _FUSED_SCRATCH[key] = (torch.empty(shape, dtype=key[3], device=self.k_cache_q.device),
                                          torch.empty(shape, dtype=key[3], device=self.k_cache_q.device))
        return pair

    def dequantized_kv(self):
        """Fused mode: the values the decode kernels see, [1, H, S, D] model dtype (fresh tensors; debugging / tests)."""
        H, S, D = self.n_heads, self.max_cache_length, self.head_dim
        k = torch.empty(self.cache_shape, dtype=self.k_cache.dtype, device=self.k_cache_q.device)
        v = torch.empty_like(k)
        _abi.call("cc_kv_dequant_rows", _ptr(self.k_cache_q), _ptr(self.v_cache_q), _ptr(self.kv_qparams), H, S, D,
                  _DT[self.k_cache.dtype], 8, _ptr(k), _ptr(v), _stream())
        return k, v

    def _quant_step(self, q, k, v, p32, HQ, scale, y, ws, num=None, denom=None, counter=None, rand=None, seed=0, g=0, w=0, phases=3):
        """The fused step over the uint8 images, recoverable form (cc_decode_step_quant_rc: commit words; random: `rand` = the
        injected vector, or None -> in-kernel draws from `seed`)."""
        _abi.call("cc_decode_step_quant_rc", self._view(), _ptr(self.kv_qparams), 8, self._fused_quant_policy(), _ptr(q), _ptr(k),
                  _ptr(v), _ptr(p32), _ptr(num), _ptr(denom), _ptr(counter), _ptr(rand), int(seed), _ptr(self.next_key),
                  _ptr(self.step_commit), int(g), int(w), HQ, scale, _ptr(y), _ptr(ws), ws.numel(), _stream(), phases)

    # ------------------------------------------------------------------ quantised KV (ref: cache.py:283-309, 323-338)
    def quantize_cache(self):
        """The reference quantises the whole cache at the end of every update and dequantises it at the start of the
        next; attention in between reads the UNQUANTISED tensors update_kv returned.  Here the working cache is
        replaced by its quantise -> dequantise round trip (one pass, cc_kv_requant) lazily — at the start of the
        next update, or when this method is called — and the quantised image / scales / zero points are emitted."""
        if self.quantize and self._quant_pending:
            H, S, D = self.n_heads, self.max_cache_length, self.head_dim
            _need_device(self.k_cache, "k/v cache")
            # slots the round trip no longer changes are skipped exactly (include/coldcompress.h); torch-side writes to
            # the working caches (prefill gathers, reset, a test poking values) move the tensors' version counters
            tag = (self.k_cache._version, self.v_cache._version, self.k_cache.data_ptr(), self.v_cache.data_ptr())
            if tag != self._quant_tag:
                self._quant_stable.zero_()
                self._quant_pos_seen.zero_()
                self._quant_tag = tag
            _abi.call("cc_kv_requant_pair", _ptr(self.k_cache), _ptr(self.k_cache_q), _ptr(self.k_scales), _ptr(self.k_zero_points),
                      _ptr(self.v_cache), _ptr(self.v_cache_q), _ptr(self.v_scales), _ptr(self.v_zero_points), H, S, D,
                      _DT[self.k_cache.dtype], int(self.n_bit), _ptr(self.pos), int(self.pos.shape[1]), _ptr(self._quant_stable),
                      _ptr(self._quant_pos_seen), _stream())
            self._quant_pending = False

    def dequantize_cache(self):
        """No-op: k_cache / v_cache always hold the dequantised values (see quantize_cache)."""

    def return_attn(self):
        return False

    def memory_usage(self):
        """ref: cache.py:247-257 — bytes of the cache's state tensors.  Our non-persistent pipeline state (partial arg-min
        keys, tracked window sums, quantisation marks) is not cache content and is left out; for a quantised cache the
        figure is the reference's (int8 / packed image + scales + zero points), not the model-dtype working copy the
        kernels read, which is reported separately as `working_cache_gb` by compute_statistics."""
        skip = set(self._non_persistent_buffers_set)
        if self.quantize:
            skip |= {"k_cache", "v_cache"}
        tensors = [b for n, b in self._buffers.items() if torch.is_tensor(b) and n not in skip]
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
        stats = {"compression_ratio": self.compression_ratio(seq_len).item(), "cache_memory_gb": self.memory_usage()}
        if self.quantize:
            stats["working_cache_gb"] = (self.k_cache.numel() + self.v_cache.numel()) * self.k_cache.element_size() / (1024 ** 3)
        return stats

    def return_kv_cache(self):
        if self.fused_quant:  # nobody on the decode path reads these: the kernels stream the images (decode_step)
            k, v = self.dequantized_kv()
            return k, v, self.mask
        return self.k_cache, self.v_cache, self.mask

    # ------------------------------------------------------------------ ABI plumbing
    def _view(self):
        kc, vc = (self.k_cache_q, self.v_cache_q) if self.fused_quant else (self.k_cache, self.v_cache)
        key = (kc.data_ptr(), vc.data_ptr(), self.pos.data_ptr(), self.mask.data_ptr(), self.cache_cts.data_ptr())
        if self._view_cache is None or self._view_cache[0] != key:
            _need_device(kc, "k_cache")
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
        self.quantize_cache()  # the previous update's quantise + this update's dequantise (no-op without cache_bits)
        if is_prefill:
            self._prefill_update(input_pos, k_val, v_val, **kwargs)
        else:
            self._decoding_update(input_pos, k_val, v_val, **kwargs)
        self._quant_pending = self.quantize
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
        if self.fused_quant:  # rows land in a model-dtype staging pair, then every row is quantised on its own grid, once
            ks, vs = self._fused_scratch()
            ks.zero_()
            vs.zero_()
            S = self.max_cache_length
            tmp = KVView(ks.data_ptr(), vs.data_ptr(), self.pos.data_ptr(), self.mask.data_ptr(), self.cache_cts.data_ptr(), H,
                         self.pos.shape[1], self.cache_cts.shape[0], S, D, _DT[self.k_cache.dtype])
            _abi.call("cc_prefill_fill", C.byref(tmp), _ptr(k), _ptr(v), _ptr(p), p.shape[0], T, _stream())
            _abi.call("cc_kv_quant_rows", _ptr(ks), _ptr(vs), H, S, D, _DT[self.k_cache.dtype], 8, _ptr(self.k_cache_q),
                      _ptr(self.v_cache_q), _ptr(self.kv_qparams), _stream())
            return
        _abi.call("cc_prefill_fill", self._view(), _ptr(k), _ptr(v), _ptr(p), p.shape[0], T, _stream())
        if self.quantize:  # a raw-pointer bulk write: the tensors' version counters did not move, the stable marks are void
            self._quant_tag = None

    def _decoding_update(self, input_pos, k_val, v_val, **kwargs):
        """ref: cache.py:348-364: generic path = `_token_importances` -> base rules -> arg-min -> insert."""
        if self.fused_quant:
            raise ColdCompressError("cache_quant_mode='fused': decode through decode_step() — the three-call path would have to "
                                    "materialise the dequantised cache every step, which is what this mode exists to avoid")
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


_REQUANT_TABLES = {}  # tuple of the caches' buffer addresses -> device table of cc_kv_requant_batch


def flush_quantized(caches):
    """The pending quantise -> dequantise round trips (KVCache.quantize_cache) of SEVERAL caches as one launch
    (cc_kv_requant_batch): what a model calls behind its last layer, so that a token costs one round-trip launch instead of
    one per layer.  The reference runs the round trip inside every layer's update (cache.py:323-338); its result is only read
    by that layer's NEXT update, so running all of them behind the last layer gives the same numbers.  Caches this form does
    not cover (not quantised in the reference's mode, nothing pending, an odd shape) take their own quantize_cache()."""
    todo = [c for c in caches if getattr(c, "quantize", False) and c._quant_pending]
    groups = {}
    for c in todo:
        vec = 16 // c.k_cache.element_size()
        if c.head_dim % vec or c.n_heads * (c.head_dim // vec) > 1024 or c.pos.shape[1] > 64 or not c.k_cache.is_cuda:
            c.quantize_cache()
            continue
        groups.setdefault((c.n_heads, c.head_dim, c.k_cache.dtype, int(c.n_bit), c.k_cache.device), []).append(c)
    for (H, D, dtype, n_bit, device), cs in groups.items():
        if len(cs) == 1:
            cs[0].quantize_cache()
            continue
        for c in cs:  # torch-side writes to the working caches void the stable marks (see quantize_cache)
            tag = (c.k_cache._version, c.v_cache._version, c.k_cache.data_ptr(), c.v_cache.data_ptr())
            if tag != c._quant_tag:
                c._quant_stable.zero_()
                c._quant_pos_seen.zero_()
                c._quant_tag = tag
        rows = [[t.data_ptr() for t in (c.k_cache, c.k_cache_q, c.k_scales, c.k_zero_points, c.v_cache, c.v_cache_q, c.v_scales,
                                        c.v_zero_points, c.pos, c._quant_stable, c._quant_pos_seen)]
                + [c.max_cache_length, int(c.pos.shape[1]), 0, 0, 0] for c in cs]
        key = tuple(x for r in rows for x in r)
        table = _REQUANT_TABLES.get(key)
        if table is None:
            if len(_REQUANT_TABLES) > 64:
                _REQUANT_TABLES.clear()
            table = _REQUANT_TABLES[key] = torch.tensor(rows, dtype=torch.int64).to(device)
        _abi.call("cc_kv_requant_batch", _ptr(table), len(cs), H, max(c.max_cache_length for c in cs), D, _DT[dtype], n_bit, _stream())
        for c in cs:
            c._quant_pending = False


def _new_step_commit(n_heads):
    """The recoverable hand-off's commit words (include/coldcompress.h, cc_decode_step_heavy_hitter_rc): per kv head the insert
    word, its position, and one committed position per workgroup of the head; -1 = nothing."""
    return torch.full((n_heads, int(_abi.lib()["cc_decode_step_commit_stride"]())), -1, dtype=torch.int32)


def step_committed(kv, position, HQ=None):
    """True when every workgroup word of `kv.step_commit` that the last single-launch step wrote holds `position` (words of splits
    the launch does not have stay -1) and at least one does."""
    w = kv.step_commit[:, 2:66]  # (words [66], [67]: the hybrid step's count and ring column)
    used = w != -1
    return bool(used.any()) and bool((w[used] == int(position)).all()) and bool(used.any(dim=1).all())


def step_is_recoverable(cache, HQ, attention=None):
    """`cache.recoverable()` AND the decode step of this cache, for `HQ` query heads on this device, actually takes the form that
    honours the status / commit words of the recoverable hand-off: the single-launch form.  The two-launch and three-call forms
    ignore both — they would step on garbage behind a failed launch and step AGAIN on the retry: the harness refuses
    in-band recovery unless every layer answers True here.  `attention`: the layer's Attention module, when the caller
    has one — a layer that does not route its decode through the fused step (`fuse_decode_step = False`: the three-call path)
    answers False whatever its cache could do."""
    if attention is not None and not getattr(attention, "fuse_decode_step", True):
        return False
    rec = getattr(cache, "recoverable", None)
    if not callable(rec) or not rec():
        return False
    if not cache.k_cache.is_cuda or getattr(cache, "single_launch", True) is False or not cache.supports_fused_step():
        return False
    lib = _abi.lib()
    if not lib["cc_decode_step_single_launch_enabled"]():
        return False
    args = (int(HQ), cache.n_heads, cache.max_cache_length, cache.head_dim, _DT[cache.k_cache.dtype])
    with torch.cuda.device(cache.k_cache.device):  # (residency and the dispatch-order verdict are facts of THAT device)
        if isinstance(cache, KVCacheHybrid):  # (isinstance: a subclass takes its parent's step)
            return bool(lib["cc_decode_step_hybrid_single_launch"](*args))
        if cache.fused_quant:
            return bool(lib["cc_decode_step_quant_single_launch"](*args, 8))
        if isinstance(cache, KVCacheL2):
            return bool(lib["cc_decode_step_l2_single_launch"](*args))
        return bool(lib["cc_decode_step_single_launch"](*args))


class KVCacheHeadConstant(KVCache):
    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, head_specific=False, **kwargs)


class KVCacheHeadSpecific(KVCache):
    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, variable_length=False, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, head_specific=True, variable_length=variable_length,
                         **kwargs)


def _qkv_step_available(cache, HQ, K):
    """The layer step can take the layer's QKV projection along (include/coldcompress.h, cc_decode_step_qkv_rc) for this cache,
    `HQ` query heads and model dim `K` on this device."""
    # NOT memoised: the answer hangs on state that moves under a running process — the single-launch switch, a demoted
    # L2 hand-off, the dispatch-order probe (not yet run at the first question) — and a stale "yes" makes cc_decode_step_qkv_rc
    # return CC_ERR_UNSUPPORTED in the middle of a decode or a recovery.  One ctypes call per layer and token in the eager loop.
    if not (cache.k_cache.is_cuda and cache.k_cache.dtype in (torch.bfloat16, torch.float16) and not cache.fused_quant and not cache.quantize):
        return False
    with torch.cuda.device(cache.k_cache.device):
        return bool(_abi.lib()["cc_decode_step_qkv_available"](int(HQ), cache.n_heads, cache.max_cache_length, cache.head_dim,
                                                               _DT[cache.k_cache.dtype], int(K)))


def _qkv_step(cache, policy, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale, qkv_out, num=None, denom=None,
              counter=None, seed=0, g=0, w=0):
    """One launch: RMSNorm(x + delta) -> wqkv (+ bias) -> RoPE -> update_kv + attention + update_state (ref: model.py:375-427)."""
    from .attention_utils import _workspace
    import math

    p32 = cache._pos32(input_pos)
    if not cache._next_valid:
        cache.prepare_decode(p32)
    D = cache.head_dim
    N, K = wqkv.shape
    assert N == (HQ + 2 * cache.n_heads) * D and x.numel() == K, "wqkv must be [(HQ + 2H) * D, dim] and x one token"
    xc = x.contiguous()
    dc = delta.contiguous() if delta is not None else None
    fc = freqs.contiguous() if freqs is not None else None
    y = torch.empty((1, HQ, 1, D), dtype=wqkv.dtype, device=wqkv.device)
    code = _DT[cache.k_cache.dtype]
    ws = _workspace(_abi.lib()["cc_decode_attn_workspace_bytes"](HQ, cache.n_heads, cache.max_cache_length, D, code), wqkv.device)
    _abi.call("cc_decode_step_qkv_rc", cache._view(), policy, _ptr(wqkv), _ptr(bias) if bias is not None else None, _ptr(xc),
              _ptr(dc) if dc is not None else None, _ptr(norm_w), float(eps), _ptr(h_out) if h_out is not None else None,
              _ptr(fc) if fc is not None else None, int(K), _ptr(qkv_out) if qkv_out is not None else None, _ptr(p32),
              _ptr(num) if num is not None else None, _ptr(denom) if denom is not None else None,
              _ptr(counter) if counter is not None else None, None, int(seed), _ptr(cache.next_key), _ptr(cache.step_commit), int(g), int(w),
              int(HQ), 1.0 / math.sqrt(D) if scale is None else float(scale), _ptr(y), _ptr(ws), ws.numel(), _stream())
    return y


class _RingFusedStep:
    """Two-launch decode step for the head-constant policies (recent_global, full, random): the slot for position
    p + 1 is scored in the combine pass of step p (arg-min of `pos` behind the sinks; for random, of the next uniform
    draw) and consumed by the K/V streaming pass of step p + 1 (cc_decode_step_recent_global / cc_decode_step_random).
    Same contract as KVCacheHeavyHitter.decode_step."""

    def _init_ring_pipeline(self, rows=None):
        """next_key: one row of partial arg-min keys per kv head — also for the head-constant policies (identical rows): each head
        reads and rewrites its own copy (include/coldcompress.h, cc_decode_step_recent_global)."""
        nk = int(_abi.lib()["cc_hh_next_key_slots"](self.max_cache_length))
        rows = self.n_heads if rows is None else rows
        self.register_buffer("next_key", torch.full((rows, nk), -1, dtype=torch.int64), persistent=False)
        self._next_valid = False
        # recoverable hand-off (include/coldcompress.h, cc_decode_step_head_constant_rc): the last position whose step is fully
        # committed, per kv head; -1 = none.  Used by the head-constant policies; l2 carries it unused.
        self.register_buffer("step_commit", _new_step_commit(self.n_heads), persistent=False)

    def recoverable(self):
        """True: a timed-out single-launch step of this cache can be retried in band (harness._recover_token): its step
        carries commit words and a retry scores the same keys.  recent_global / full / l2 / random with in-kernel draws, 16-bit
        caches and (r4) the fused uint8 mode."""
        return False

    def supports_fused_step(self):
        return True

    def reset(self):
        super().reset()
        self._next_valid = False
        self.step_commit.fill_(-1)

    def update_kv(self, input_pos, k_val, v_val, is_prefill, **kwargs):
        self._next_valid = False  # the three-call path mutates pos outside the pipeline
        if is_prefill:
            self.step_commit.fill_(-1)  # positions restart: no decode position is committed
        return super().update_kv(input_pos, k_val, v_val, is_prefill, **kwargs)

    def _ring_sinks(self):
        """Slots the ring's arg-min skips (cache.py:554 `pos[:, :, g:]`)."""
        return int(self.global_tokens)

    def _pipeline_init(self, p32):
        _abi.call("cc_rg_next_key_init", self._view(), _ptr(p32), self._ring_sinks(), _ptr(self.next_key), _stream())

    def _pipeline_step(self, q, k, v, p32, HQ, scale, y, ws):
        if self.fused_quant:
            return self._quant_step(q, k, v, p32, HQ, scale, y, ws, g=self._ring_sinks())
        _abi.call("cc_decode_step_head_constant_rc", self._view(), 2, _ptr(q), _ptr(k), _ptr(v), _ptr(p32), None, 0,
                  _ptr(self.next_key), _ptr(self.step_commit), self._ring_sinks(), 0, HQ, scale, _ptr(y), _ptr(ws), ws.numel(),
                  _stream())

    # ---- the step with the layer's QKV projection folded in (r5): recent_global / full, random with in-kernel draws
    _qkv_policy = 2

    def qkv_step_available(self, HQ, K):
        return type(self)._pipeline_step is _RingFusedStep._pipeline_step and _qkv_step_available(self, HQ, K)

    def decode_step_qkv(self, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale=None, qkv_out=None):
        return _qkv_step(self, self._qkv_policy, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale, qkv_out,
                         g=self._ring_sinks())

    def prepare_decode(self, input_pos):
        self._pipeline_init(self._pos32(input_pos))
        self._next_valid = True
        # a re-seeded pipeline starts from no committed position: a caller that rolled the cache back and runs a position AGAIN
        # gets a step, not the recoverable hand-off's replay of it (a retry after a failed hand-off does not come here)
        self.step_commit.fill_(-1)

    def decode_step(self, query, k_val, v_val, input_pos, scale=None):
        from .attention_utils import _workspace
        import math

        self.quantize_cache()
        k, v = self._new_rows(k_val, v_val)
        p32 = self._pos32(input_pos)
        if not self._next_valid:
            self.prepare_decode(p32)
        _, HQ, _, D = query.shape
        q = query.reshape(HQ, D).contiguous()
        y = torch.empty((1, HQ, 1, D), dtype=query.dtype, device=query.device)
        nbytes = _abi.lib()["cc_decode_attn_workspace_bytes"](HQ, self.n_heads, self.max_cache_length, D, _DT[self.k_cache.dtype])
        ws = _workspace(nbytes, query.device)
        self._pipeline_step(q, k, v, p32, HQ, 1.0 / math.sqrt(D) if scale is None else scale, y, ws)
        self._quant_pending = self.quantize
        return y


class KVCacheFull(_RingFusedStep, KVCacheHeadConstant):
    """ref: cache.py:493-502."""

    def recoverable(self):
        return True

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        self.global_tokens = 0
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **kwargs)
        self._init_ring_pipeline()

    def _fused_quant_policy(self):
        return 2

    def _ring_sinks(self):
        return 0  # cache.py:502 is a plain pos.argmin(): a `global_tokens` kwarg (it overrides the 0 above, as in the reference) is ignored

    def _run_select(self, input_pos, k, v):
        _abi.call("cc_decode_update_full", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  _ptr(self._idx_buf()), _stream())


class KVCacheRandom(_RingFusedStep, KVCacheHeadConstant):
    """ref: cache.py:505-524.  The reference draws torch.rand(S) per eviction — a backend-specific stream; `_rand` is the injection
    point tests use to replay the reference's draws (parity is defined given the vector).  In the fused pipeline the draw for
    position p + 1 is made during step p (one draw per step, same order as the reference), and when `_rand` is not overridden it
    is made IN the kernels (cc_decode_step_random_rng: a stateless hash of (seed, position, slot)) — no vector, no extra launch per
    step.  The seed is drawn from torch's CPU generator when the pipeline is first seeded after construction or reset(): every
    generation draws fresh eviction scores, like the reference's torch.rand per step (cache.py:521), and torch.manual_seed is honoured
    per generation.  The steps carry the seed by value: a caller that captured them in a hipGraph must capture again when
    `_graph_epoch` has moved (harness.GraphedDecoder does)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "recent_window"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **kwargs)
        self._init_ring_pipeline()
        self._rng_seed = 0
        self._graph_epoch = 0  # bumped whenever a value that captured steps carry BY VALUE (the seed) changes

    def reset(self):
        super().reset()
        self._rng_seed = 0  # the next generation draws its own seed (one seed for the object's lifetime made every
        # generation on the same model draw identical eviction scores at the same (position, slot))

    def _rand(self):
        return torch.rand(self.max_cache_length, device=self.k_cache.device)

    def _in_kernel_rng(self):
        return "_rand" not in self.__dict__ and type(self)._rand is KVCacheRandom._rand

    def recoverable(self):
        return self._in_kernel_rng()  # (an injected vector would be drawn again by the retry)

    def _pipeline_init(self, p32):
        if self._in_kernel_rng():
            # one seed per generation: drawn from torch's CPU generator when the pipeline is first seeded after construction / reset()
            # (follows torch.manual_seed, not torch.cuda.manual_seed; no device sync) and kept until the next reset().  The steps carry
            # it by value — a hipGraph captured with the previous seed would replay THAT one while this init kernel scored with the new
            # (ADVICE r3): `_graph_epoch` tells the holder of such a graph to capture again (harness.GraphedDecoder checks it).
            if not self._rng_seed:
                self._rng_seed = int(torch.randint(1, 2 ** 62, (1,)).item())
                self._graph_epoch += 1
            _abi.call("cc_random_next_key_init_rng", self._view(), _ptr(p32), self._rng_seed, int(self.global_tokens),
                      int(self.recent_window), _ptr(self.next_key), _stream())
            return
        r = self._rand().to(torch.float32).contiguous()
        _abi.call("cc_random_next_key_init", self._view(), _ptr(p32), _ptr(r), int(self.global_tokens), int(self.recent_window),
                  _ptr(self.next_key), _stream())

    def _fused_quant_policy(self):
        return 3

    def qkv_step_available(self, HQ, K):
        return self._in_kernel_rng() and _qkv_step_available(self, HQ, K)

    def decode_step_qkv(self, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale=None, qkv_out=None):
        if not self._next_valid:
            self.prepare_decode(self._pos32(input_pos))  # (draws the seed)
        return _qkv_step(self, 3, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale, qkv_out, seed=self._rng_seed,
                         g=self.global_tokens, w=self.recent_window)

    def _pipeline_step(self, q, k, v, p32, HQ, scale, y, ws):
        if self._in_kernel_rng() and self.fused_quant:  # (r4: in-kernel draws for the uint8 images too: no vector, no torch.rand launch)
            return self._quant_step(q, k, v, p32, HQ, scale, y, ws, rand=None, seed=self._rng_seed, g=self.global_tokens, w=self.recent_window)
        if self._in_kernel_rng():  # (the recoverable form: a retried step scores the same draws)
            _abi.call("cc_decode_step_head_constant_rc", self._view(), 3, _ptr(q), _ptr(k), _ptr(v), _ptr(p32), None,
                      self._rng_seed, _ptr(self.next_key), _ptr(self.step_commit), int(self.global_tokens),
                      int(self.recent_window), HQ, scale, _ptr(y), _ptr(ws), ws.numel(), _stream())
            return
        r = self._rand().to(torch.float32).contiguous()
        if self.fused_quant:
            return self._quant_step(q, k, v, p32, HQ, scale, y, ws, rand=r, g=self.global_tokens, w=self.recent_window)
        _abi.call("cc_decode_step_random", self._view(), _ptr(q), _ptr(k), _ptr(v), _ptr(p32), _ptr(r), _ptr(self.next_key),
                  int(self.global_tokens), int(self.recent_window), HQ, scale, _ptr(y), _ptr(ws), ws.numel(), _stream())

    def _run_select(self, input_pos, k, v):
        r = self._rand().to(torch.float32).contiguous()
        _abi.call("cc_decode_update_random", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)), _ptr(r),
                  int(self.global_tokens), int(self.recent_window), _ptr(self._idx_buf()), _stream())


class KVCacheRecentGlobal(_RingFusedStep, KVCacheHeadConstant):
    """ref: cache.py:527-556 (ring buffer behind the global sink tokens)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **kwargs)
        self._init_ring_pipeline()

    def recoverable(self):
        return True

    def _fused_quant_policy(self):
        return 2

    def _run_select(self, input_pos, k, v):
        _abi.call("cc_decode_update_recent_global", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  int(self.global_tokens), _ptr(self._idx_buf()), _stream())


class KVCacheL2(_RingFusedStep, KVCacheHeadSpecific):
    """ref: cache.py:559-612.  Two-launch decode step (cc_decode_step_l2) for 16-bit caches with head_dim 128: the
    global norm maximum of cache.py:602 is folded across the step boundary (per-wave maxima from the streaming pass +
    the freshly inserted norms, combined in the combine pass)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "recent_window"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **kwargs)
        if self.quantize:
            raise NotImplementedError(
                "l2 with cache_bits: the reference itself fails on this path (cache.py:611 takes the norm of the "
                "quantised int8 cache: 'linalg.vector_norm: Expected a floating point ... Got Char'), so there is "
                "no behaviour to reproduce or pin")
        self.register_buffer("key_norm", torch.zeros((1, n_heads, self.max_cache_length), dtype=dtype))
        self._init_ring_pipeline(rows=n_heads)

    def supports_fused_step(self):
        return self.k_cache.dtype in (torch.bfloat16, torch.float16) and self.head_dim == 128

    def qkv_step_available(self, HQ, K):
        return False  # (the l2 step keeps its own launch: its norm bookkeeping has no QKV instantiation)

    def _pipeline_init(self, p32):
        _abi.call("cc_l2_next_key_init", self._view(), _ptr(p32), _ptr(self.key_norm), int(self.global_tokens),
                  int(self.recent_window), _ptr(self.next_key), _stream())

    def _pipeline_step(self, q, k, v, p32, HQ, scale, y, ws):
        _abi.call("cc_decode_step_l2_rc", self._view(), _ptr(q), _ptr(k), _ptr(v), _ptr(p32), _ptr(self.key_norm), _ptr(self.next_key),
                  _ptr(self.step_commit), int(self.global_tokens), int(self.recent_window), HQ, scale, _ptr(y), _ptr(ws), ws.numel(),
                  _stream())

    def recoverable(self):
        return True

    def reset(self):
        super().reset()
        self.key_norm.zero_()

    def _run_select(self, input_pos, k, v):
        ws = self._scratch.get("l2_max", (256,), torch.uint8, self.k_cache.device)  # the norm maximum, taken before any insert
        _abi.call("cc_decode_update_l2", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  _ptr(self.key_norm), int(self.global_tokens), int(self.recent_window), _ptr(self._idx_buf()), _ptr(ws), ws.numel(),
                  _stream())

    def update_state(self, input_pos, k_val, v_val, is_prefill, attn, **kwargs):
        if is_prefill:  # ref: cache.py:611-612 — norms of every slot of the freshly filled cache
            _abi.call("cc_row_l2_norm", _ptr(self.k_cache), self.n_heads, self.max_cache_length, self.head_dim,
                      _DT[self.k_cache.dtype], 0, _ptr(self.key_norm), _stream())


class _TrackedWindowSums:
    """Exact window sums of the [H, S, W] attention-history ring, kept incrementally on the device (cc_hh_ring_update
    with tracked state; see include/coldcompress.h) instead of re-reading the whole ring on every decode step.  The
    kernels write through raw pointers, so torch's tensor version counter moves only when torch-side code touches the
    ring (reset, load_state_dict, a test poking values): that is the signal to rebuild the state from the ring."""

    def _init_window_state(self):
        H, S = self.n_heads, self.max_cache_length
        words = int(_abi.lib()["cc_hh_ring_acc_words"](H, S, int(self.history_window_size), _DT[self.attn_history_num.dtype]))
        self.register_buffer("attn_window_acc", torch.zeros(words, dtype=torch.int64), persistent=False)
        self.register_buffer("attn_window_sum", torch.zeros((H, S), dtype=torch.float32), persistent=False)
        self._ring_version = self._ring_tag()  # all-zero state of an all-zero ring

    def _ring_tag(self):
        r = self.attn_history_num
        return (r._version, r.data_ptr(), self.attn_window_acc.data_ptr())

    def _zero_window_state(self):
        self.attn_window_acc.zero_()
        self.attn_window_sum.zero_()
        self._ring_version = self._ring_tag()

    def _window_state(self):
        """-> (wsum, acc), current for the ring."""
        if self._ring_tag() != self._ring_version:
            _abi.call("cc_hh_ring_window_sums", _ptr(self.attn_history_num), self.n_heads, self.max_cache_length,
                      int(self.history_window_size), _DT[self.attn_history_num.dtype], _ptr(self.attn_window_sum),
                      _ptr(self.attn_window_acc), _stream())
            self._ring_version = self._ring_tag()
        return self.attn_window_sum, self.attn_window_acc


class KVCacheHeavyHitter(_TrackedWindowSums, KVCacheHeadSpecific):
    """ref: cache.py:615-765 (ScissorHands / H2O style accumulated attention), history_window_size == 1."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "history_window_size",
                       "recent_window", "attn_thresholding"]
    single_launch = True  # decode_step: one launch per layer step where possible (False: always the two-launch step)

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, variable_length=False, **kwargs):
        super().__init__(max_batch_size, n_heads, head_dim, dtype, variable_length, **kwargs)
        if self.attn_thresholding:
            raise NotImplementedError(
                "attn_thresholding: the reference itself fails on this path (cache.py:721 index_put of an Int source "
                "into its Bool history buffer raises), so there is no behaviour to reproduce or pin")
        S, W = self.max_cache_length, int(self.history_window_size)
        # ref: cache.py:661-667 — one float64 accumulator for the full history (W == 1), else a model-dtype ring
        self.register_buffer("attn_history_num", torch.zeros((1, n_heads, S, W), dtype=torch.float64 if W == 1 else dtype))
        self.register_buffer("attn_history_denom", torch.zeros((1, n_heads, S), dtype=torch.int32))
        self.register_buffer("attn_counter", torch.zeros((1,), dtype=torch.int64))
        # set by the attention op when it already applied this step's history update in its combine pass
        self._state_fused = False
        # fused decode-step pipeline (decode_step): partial arg-min keys for the NEXT position, [H, NK]
        nk = int(_abi.lib()["cc_hh_next_key_slots"](S))
        self.register_buffer("next_key", torch.full((n_heads, nk), -1, dtype=torch.int64), persistent=False)
        self._next_valid = False
        # recoverable hand-off of the single-launch step (include/coldcompress.h, cc_decode_step_heavy_hitter_rc): the last
        # position whose step is fully committed, per kv head; -1 = none
        self.register_buffer("step_commit", _new_step_commit(n_heads), persistent=False)
        if W > 1:
            self._init_window_state()

    def reset(self):
        super().reset()
        self.attn_history_num.zero_()
        self.attn_history_denom.zero_()
        self.attn_counter.zero_()
        self._next_valid = False
        self.step_commit.fill_(-1)
        if self.history_window_size > 1:
            self._zero_window_state()

    def return_attn(self) -> bool:
        return True

    def supports_fused_step(self):
        return True

    def recoverable(self):
        """A timed-out single-launch step can be retried in band (cc_decode_step_heavy_hitter_rc): W = 1, 16-bit cache."""
        return self.history_window_size == 1

    def _fused_quant_policy(self):
        return 1 if int(getattr(self, "history_window_size", 1)) == 1 else 0

    def update_kv(self, input_pos, k_val, v_val, is_prefill, **kwargs):
        self._next_valid = False  # the three-call path mutates pos / history outside the pipeline
        if is_prefill:
            self.step_commit.fill_(-1)  # positions restart: no decode position is committed
        return super().update_kv(input_pos, k_val, v_val, is_prefill, **kwargs)

    # ------------------------------------------------------------------ fused decode step (1 or 2 launches per layer)
    def prepare_decode(self, input_pos):
        """Seed the pipeline: arg-min keys for `input_pos` from the current state (one select-only launch)."""
        if self.history_window_size != 1:  # finite history window: scored from the tracked window sums
            wsum, _ = self._window_state()
            _abi.call("cc_hh_ring_next_key_init", self._view(), _ptr(self._pos32(input_pos)), _ptr(self.attn_history_denom),
                      int(self.history_window_size), _ptr(wsum), int(self.global_tokens), int(self.recent_window),
                      _ptr(self.next_key), _stream())
        else:
            _abi.call("cc_hh_next_key_init", self._view(), _ptr(self._pos32(input_pos)), _ptr(self.attn_history_num),
                      _ptr(self.attn_history_denom), int(self.global_tokens), int(self.recent_window), _ptr(self.next_key),
                      _stream())
        self._next_valid = True
        self.step_commit.fill_(-1)  # (a re-seeded pipeline starts from no committed position: see _RingFusedStep.prepare_decode)

    def decode_step(self, query, k_val, v_val, input_pos, scale=None):
        """update_kv + attention over the pruned cache + update_state for one decode token in ONE launch where the shape and the
        device allow it (include/coldcompress.h: single-launch layer step), two otherwise (cc_decode_step_heavy_hitter).  Bit-identical to the three-call sequence; positions must advance by one
        between calls (any update_kv / update_state / reset in between re-seeds automatically)."""
        from .attention_utils import _workspace
        import math

        self.quantize_cache()
        k, v = self._new_rows(k_val, v_val)
        p32 = self._pos32(input_pos)
        if not self._next_valid:
            self.prepare_decode(p32)
        _, HQ, _, D = query.shape
        q = query.reshape(HQ, D).contiguous()
        y = torch.empty((1, HQ, 1, D), dtype=query.dtype, device=query.device)
        code = _DT[self.k_cache.dtype]
        nbytes = _abi.lib()["cc_decode_attn_workspace_bytes"](HQ, self.n_heads, self.max_cache_length, D, code)
        ws = _workspace(nbytes, query.device)
        if self.history_window_size != 1:
            wsum, acc = self._window_state()
            _abi.call("cc_decode_step_heavy_hitter_ring", self._view(), _ptr(q), _ptr(k), _ptr(v), _ptr(p32), _ptr(self.attn_history_num),
                      _ptr(self.attn_history_denom), _ptr(self.attn_counter), int(self.history_window_size), _ptr(acc), _ptr(wsum),
                      _ptr(self.next_key), int(self.global_tokens), int(self.recent_window), HQ,
                      1.0 / math.sqrt(D) if scale is None else scale, _ptr(y), None, _ptr(ws), ws.numel(), _stream())
            self._quant_pending = self.quantize
            return y
        # one launch per layer step where the shape and the device allow it (include/coldcompress.h), else two
        phases = 3 if self.single_launch else 3 | _abi.CC_PHASE_TWO_LAUNCH
        if self.fused_quant:  # the same step over the uint8 images (cc_decode_step_quant)
            self._quant_step(q, k, v, p32, HQ, 1.0 / math.sqrt(D) if scale is None else scale, y, ws, num=self.attn_history_num,
                             denom=self.attn_history_denom, counter=self.attn_counter, g=self.global_tokens, w=self.recent_window,
                             phases=phases)
            return y
        _abi.call("cc_decode_step_heavy_hitter_rc", self._view(), _ptr(q), _ptr(k), _ptr(v), _ptr(p32), _ptr(self.attn_history_num),
                  _ptr(self.attn_history_denom), _ptr(self.attn_counter), _ptr(self.next_key), _ptr(self.step_commit),
                  int(self.global_tokens), int(self.recent_window), HQ, 1.0 / math.sqrt(D) if scale is None else scale, _ptr(y),
                  _ptr(ws), ws.numel(), _stream(), phases)
        self._quant_pending = self.quantize
        return y

    def qkv_step_available(self, HQ, K):
        return self.single_launch and self.history_window_size == 1 and _qkv_step_available(self, HQ, K)

    def decode_step_qkv(self, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale=None, qkv_out=None):
        """decode_step with the layer's QKV projection folded into the launch (cc_decode_step_qkv_rc): same cache state, same y —
        q / k / v are cc_gemv_fused's, bit for bit."""
        return _qkv_step(self, 1, wqkv, bias, x, delta, norm_w, eps, h_out, freqs, input_pos, HQ, scale, qkv_out,
                         num=self.attn_history_num, denom=self.attn_history_denom, counter=self.attn_counter, g=self.global_tokens,
                         w=self.recent_window)

    def single_launch_active(self, HQ):
        """True when decode_step runs as ONE launch for `HQ` query heads on this device."""
        if self.fused_quant:
            return bool(self.single_launch and _abi.lib()["cc_decode_step_quant_single_launch"](
                HQ, self.n_heads, self.max_cache_length, self.head_dim, _DT[self.k_cache.dtype], 8))
        return bool(self.single_launch and self.history_window_size == 1 and _abi.lib()["cc_decode_step_single_launch"](
            HQ, self.n_heads, self.max_cache_length, self.head_dim, _DT[self.k_cache.dtype]))

    def step_status(self, HQ=None):
        """0, or 1 if a single-launch step ever failed to complete its in-launch hand-off on this device (synchronises)."""
        from .attention_utils import single_launch_status

        return single_launch_status(self.pos.device)

    def _run_select(self, input_pos, k, v):
        if self.history_window_size == 1:
            _abi.call("cc_decode_update_heavy_hitter", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                      _ptr(self.attn_history_num), _ptr(self.attn_history_denom), int(self.global_tokens),
                      int(self.recent_window), _ptr(self._idx_buf()), _stream())
        else:
            wsum, acc = self._window_state()
            _abi.call("cc_decode_update_heavy_hitter_ring", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                      _ptr(self.attn_history_num), _ptr(self.attn_history_denom), int(self.history_window_size),
                      int(self.global_tokens), int(self.recent_window), _ptr(self._idx_buf()), _ptr(wsum), _ptr(acc), _stream())

    def fused_history(self):
        """What the decode attention kernel needs to fold cache.py:690-723 into its combine pass: (num, denom, counter)
        for the W == 1 accumulator, (ring, denom, counter, W, acc, wsum) for the W > 1 ring with tracked window sums."""
        if self.history_window_size != 1:
            wsum, acc = self._window_state()
            return self.attn_history_num, self.attn_history_denom, self.attn_counter, int(self.history_window_size), acc, wsum
        return self.attn_history_num, self.attn_history_denom, self.attn_counter

    def _apply(self, attn_hs, T):
        if self.history_window_size == 1:
            _abi.call("cc_hh_update", _ptr(self.attn_history_num), _ptr(self.attn_history_denom), _ptr(self.attn_counter),
                      _ptr(attn_hs), self.n_heads, self.max_cache_length, T, _DT[self.k_cache.dtype], _stream())
        else:
            wsum, acc = self._window_state()
            _abi.call("cc_hh_ring_update", _ptr(self.attn_history_num), _ptr(self.attn_history_denom), _ptr(self.attn_counter),
                      _ptr(attn_hs), self.n_heads, self.max_cache_length, T, int(self.history_window_size),
                      _DT[self.k_cache.dtype], _ptr(acc), _ptr(wsum), _stream())

    def update_state(self, input_pos, k_val, v_val, is_prefill, attn, **kwargs):
        """ref: cache.py:690-723."""
        self._next_valid = False
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


_HF = {"heavy_hitter": 1, "window": 2, "punc": 4, "special": 8}


class KVCacheHybrid(_TrackedWindowSums, KVCacheHeadSpecific):
    """FastGen-style per-head policies (ref: cache.py:768-1288): at prefill every head is profiled against the
    ordered list `hybrid_strategies` and takes the FIRST policy that recovers `min_recovery_frac` of its attention
    mass; at decode each head appends or evicts according to its own policy (one launch for all heads; the
    reference loops over heads in Python with a device sync each).

    Differences a caller can rely on:
      * the prefill partition (kept tokens first) is STABLE; the reference's non-stable argsort leaves the slot
        order within each class implementation-defined on CPU.  `_partition_order` is the hook.
      * the profiling score is evaluated per key from column sums and window band sums in fp32 instead of from
        [n_policies, H, L, L] boolean masks (8.6 GB at L = 16k); heads whose score sits within rounding of
        `min_recovery_frac` may therefore pick a neighbouring policy.
      * `reset_history_on_evict = False` reproduces the reference's EFFECTIVE behaviour: its reset of the evicted
        slot's history (cache.py:992-996) indexes with a tensor, so `.fill_(0)` lands on an advanced-indexing copy
        and the ring / denominator of the reused slot are never cleared (verified against the reference: the
        denominator of a re-used slot keeps counting).  Set it to True for the behaviour the code intends.
    """
    reset_history_on_evict = False
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "token_ids",
                       "min_recovery_frac", "hybrid_strategies"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, **kwargs):
        self.attn_thresholding = False
        self.history_window_size = 400  # ScissorHands default, fixed by the reference (cache.py:789-790)
        self.recent_window = None
        super().__init__(max_batch_size, n_heads, head_dim, dtype, variable_length=True, **kwargs)
        if self.quantize:
            raise NotImplementedError(
                "hybrid with cache_bits: the reference itself fails on this path (profile_and_update writes float rows "
                "into the quantised int8 cache, cache.py:1228-1260: 'Index put requires the source and destination "
                "dtypes match'), so there is no behaviour to reproduce or pin")
        S, W = self.max_cache_length, self.history_window_size
        self.register_buffer("attn_history_num", torch.zeros((1, n_heads, S, W), dtype=dtype))
        self.register_buffer("attn_history_denom", torch.zeros((1, n_heads, S), dtype=torch.int32))
        self.register_buffer("attn_counter", torch.zeros((1,), dtype=torch.int64))
        names = [s["strategy"] for s in self.hybrid_strategies]
        self.requires_special = any("special" in n for n in names)
        self.requires_punc = any("punc" in n for n in names)
        if self.requires_special:
            self.special_ids = [list(ids) for ids in kwargs["token_ids"]["special"]]
            self.register_buffer("special_mask", torch.zeros((1, n_heads, S), dtype=torch.bool))
            self.register_buffer("num_special", torch.zeros((1,), dtype=torch.int32))
        if self.requires_punc:
            self.register_buffer("punc_ids", torch.tensor(kwargs["token_ids"]["punctuation"], dtype=torch.int64))
            self.register_buffer("punc_mask", torch.zeros((1, n_heads, S), dtype=torch.bool))
            self.register_buffer("num_punc", torch.zeros((1,), dtype=torch.int32))
        self.requires_heavy_hitter = self._init_requires_heavy_hitter()
        self.cache_strategies = None
        self._table = None
        self._state_fused = False  # set by the attention op when its combine pass already recorded this step's attention
        self._init_window_state()
        # fused decode step (decode_step): every head's eviction candidate for the NEXT position, [H, NK]
        nk = int(_abi.lib()["cc_hh_next_key_slots"](S)) if _abi.built() else 0
        self.register_buffer("next_key", torch.full((n_heads, max(nk, 1)), -1, dtype=torch.int64), persistent=False)
        self.register_buffer("step_commit", _new_step_commit(n_heads), persistent=False)  # recoverable hand-off (cc_decode_step_hybrid_rc)
        self._next_valid = False

    # ------------------------------------------------------------------ small helpers
    def _init_requires_heavy_hitter(self):
        return any("heavy_hitter" in s["strategy"] for s in self.hybrid_strategies)

    def return_attn(self):
        return self.requires_heavy_hitter

    def attn_bands(self, seq_len):
        """Window widths (in queries) whose band sums the profiling score needs (ref: cache.py:1093)."""
        return sorted({max(1, int(s["recent_window"] * seq_len)) for s in self.hybrid_strategies if "window" in s["strategy"]})

    def _policy_table(self):
        if self._table is None or self._table.device != self.k_cache.device:
            S = self.max_cache_length
            rows = []
            for s in self.hybrid_strategies:
                name = s["strategy"]
                flags = 16 if name == "full" else sum(v for k, v in _HF.items() if k in name)
                rows.append([flags, round(s.get("recent_window", 0) * S), round(s.get("heavy_hitter_frac", 0) * S)])
            self._table = torch.tensor(rows, dtype=torch.int32, device=self.k_cache.device)
        return self._table

    def reset(self):
        super().reset()
        self.attn_history_num.zero_()
        self.attn_history_denom.zero_()
        self.attn_counter.zero_()
        self._zero_window_state()
        self.cache_strategies = None
        self._next_valid = False
        self.step_commit.fill_(-1)
        self.requires_heavy_hitter = self._init_requires_heavy_hitter()
        if hasattr(self, "special_mask"):
            self.special_mask.zero_()
            self.num_special.zero_()
        if hasattr(self, "punc_mask"):
            self.punc_mask.zero_()
            self.num_punc.zero_()

    def build_special_ids_mask(self, input_ids):
        """ref: cache.py:1021-1034 (exact sub-sequence match for multi-token special ids) — as shifted compares on the
        device: no `.tolist()` round trip per layer (it cost ~10 ms x 32 layers on a 16k prompt)."""
        ids = input_ids.reshape(-1)
        L = ids.numel()
        m = torch.zeros(L, dtype=torch.bool, device=ids.device)
        for sp in self.special_ids:
            n = len(sp)
            if n == 0 or n > L:
                continue
            hit = torch.ones(L - n + 1, dtype=torch.bool, device=ids.device)  # hit[i]: ids[i : i + n] == sp
            for j, tok in enumerate(sp):
                hit &= ids[j:L - n + 1 + j] == tok
            for j in range(n):
                m[j:L - n + 1 + j] |= hit
        return m

    def _partition_order(self, mask_optimal):
        """Kept tokens first, original order preserved inside each class (stable)."""
        return torch.argsort((~mask_optimal).to(torch.int8), dim=1, stable=True)

    # ------------------------------------------------------------------ fused decode step (1 or 2 launches per layer)
    def supports_fused_step(self):
        """cc_decode_step_hybrid: 16-bit caches with head_dim 128, profiled heads, and the reference's effective
        no-reset-on-evict behaviour (see the class docstring)."""
        return (self.cache_strategies is not None and self.k_cache.dtype in (torch.bfloat16, torch.float16) and self.head_dim == 128
                and not self.reset_history_on_evict and self.n_heads <= 48 and len(self.hybrid_strategies) <= 21)

    def _punc_operands(self, input_ids):
        if not hasattr(self, "punc_ids"):
            return None, None
        tok = input_ids.to(device=self.punc_ids.device, dtype=torch.int64).reshape(-1)[:1].contiguous()
        return tok, self.punc_ids

    def _check_next_key(self):
        """The key rows are strided by the C side's cc_hh_next_key_slots(S): a cache constructed before the library was built
        holds a placeholder row — resize it (the kernels would write past its end)."""
        nk = int(_abi.lib()["cc_hh_next_key_slots"](self.max_cache_length))
        if self.next_key.shape[1] != nk:
            self.next_key = torch.full((self.n_heads, nk), -1, dtype=torch.int64, device=self.next_key.device)
            self._next_valid = False

    def prepare_decode(self, input_pos):
        """Seed the pipeline: every head's eviction candidate for `input_pos` from the current state (one launch)."""
        self._check_next_key()
        tab = self._policy_table()
        wsum, _ = self._window_state()
        _abi.call("cc_hybrid_next_key_init", self._view(), _ptr(self._pos32(input_pos)), _ptr(self.cache_strategies), _ptr(tab),
                  tab.shape[0], _ptr(self.attn_history_denom), self.history_window_size, _ptr(wsum),
                  _ptr(getattr(self, "special_mask", None)), _ptr(getattr(self, "punc_mask", None)), int(self.global_tokens),
                  _ptr(self.next_key), _stream())
        self._next_valid = True
        self.step_commit.fill_(-1)  # (a re-seeded pipeline starts from no committed position: see _RingFusedStep.prepare_decode)
        # ... and from no head counted: a failed step that nobody retried (recover=False) leaves the single-launch step's ticket
        # word — heads that have committed the current step — short of H (include/coldcompress.h)
        from .attention_utils import _decode_workspaces

        off = int(_abi.lib()["cc_decode_step_status_offset"]()) - 4
        for ws in _decode_workspaces(self.pos.device):
            if off + 4 <= ws.numel():
                ws[off:off + 4].zero_()

    def recoverable(self):
        """A timed-out single-launch step can be retried in band (cc_decode_step_hybrid_rc, late r4): the per-head decision is
        recorded in the commit words, every workgroup commits or repeats its own part."""
        return self.supports_fused_step()

    def decode_step(self, query, k_val, v_val, input_pos, scale=None, input_ids=None):
        """update_kv + attention + update_state of one decode token (cc_decode_step_hybrid): the per-head decision and the
        insert ride the K/V streaming pass; the ring update, the next candidates, the counts and num_punc its tail (ONE launch:
        cc_decode_step_hybrid_single_launch) or the combine pass (two launches).  Every buffer bit-identical to the three-launch
        sequence (y within one rounding in the single-launch form); positions must advance by one between calls."""
        from .attention_utils import _workspace
        import math

        if hasattr(self, "punc_ids") and input_ids is None:
            # ref: cache.py:975 needs the token to classify it (torch.isin(input_ids, punc_ids)); the three-call path fails on
            # None too.  Without it the launch would skip the punctuation bookkeeping silently.
            raise ColdCompressError("KVCacheHybrid.decode_step needs input_ids=<the token being inserted> when a policy uses "
                                    "punctuation (cache.py:975)")
        self._check_next_key()
        k, v = self._new_rows(k_val, v_val)
        p32 = self._pos32(input_pos)
        if not self._next_valid:
            self.prepare_decode(p32)
        _, HQ, _, D = query.shape
        q = query.reshape(HQ, D).contiguous()
        y = torch.empty((1, HQ, 1, D), dtype=query.dtype, device=query.device)
        nbytes = _abi.lib()["cc_decode_attn_workspace_bytes"](HQ, self.n_heads, self.max_cache_length, D, _DT[self.k_cache.dtype])
        ws = _workspace(nbytes, query.device)
        tab = self._policy_table()
        tok, pids = self._punc_operands(input_ids) if input_ids is not None else (None, None)
        ring = denom = counter = acc = wsum = None
        if self.requires_heavy_hitter:
            wsum, acc = self._window_state()
            ring, denom, counter = self.attn_history_num, self.attn_history_denom, self.attn_counter
        _abi.call("cc_decode_step_hybrid_rc", self._view(), _ptr(q), _ptr(k), _ptr(v), _ptr(p32), _ptr(self.cache_strategies), _ptr(tab),
                  tab.shape[0], _ptr(ring), _ptr(denom), _ptr(counter), self.history_window_size, _ptr(acc), _ptr(wsum),
                  _ptr(getattr(self, "special_mask", None)), _ptr(getattr(self, "punc_mask", None)), _ptr(tok), _ptr(pids),
                  0 if pids is None else pids.numel(), _ptr(getattr(self, "num_special", None)), _ptr(getattr(self, "num_punc", None)),
                  _ptr(self.next_key), _ptr(self.step_commit), int(self.global_tokens), HQ,
                  1.0 / math.sqrt(D) if scale is None else scale, _ptr(y), None, _ptr(ws), ws.numel(), _stream())
        return y

    # ------------------------------------------------------------------ decode (ref: cache.py:965-1019)
    def _decoding_update(self, input_pos, k_val, v_val, **kwargs):
        if self.cache_strategies is None:
            raise ColdCompressError("hybrid cache used before prefill profiling (update_state with is_prefill=True)")
        self._next_valid = False  # the three-call path mutates pos / counts / history outside the pipeline
        k, v = self._new_rows(k_val, v_val)
        tok = pids = None
        if hasattr(self, "punc_ids"):  # ref: cache.py:975 torch.isin(input_ids, punc_ids) — evaluated inside the launch
            ids = kwargs.get("input_ids")
            tok = ids.to(device=self.punc_ids.device, dtype=torch.int64).reshape(-1)[:1].contiguous()
            pids = self.punc_ids
        tab = self._policy_table()
        wsum, acc = self._window_state()
        _abi.call("cc_hybrid_decode_update", self._view(), _ptr(k), _ptr(v), _ptr(self._pos32(input_pos)),
                  _ptr(self.cache_strategies), _ptr(tab), tab.shape[0], _ptr(self.attn_history_num),
                  _ptr(self.attn_history_denom), self.history_window_size, _ptr(getattr(self, "special_mask", None)),
                  _ptr(getattr(self, "punc_mask", None)), None, _ptr(tok), _ptr(pids), 0 if pids is None else pids.numel(),
                  _ptr(getattr(self, "num_special", None)),
                  _ptr(getattr(self, "num_punc", None)), int(self.global_tokens),
                  int(bool(self.requires_heavy_hitter and self.reset_history_on_evict)),
                  _ptr(self._idx_buf()), _ptr(wsum), _ptr(acc), _stream())

    def _ring_update(self, attn_ht, T):
        wsum, acc = self._window_state()
        _abi.call("cc_hh_ring_update", _ptr(self.attn_history_num), _ptr(self.attn_history_denom), _ptr(self.attn_counter),
                  _ptr(attn_ht), self.n_heads, self.max_cache_length, T, self.history_window_size,
                  _DT[self.k_cache.dtype], _ptr(acc), _ptr(wsum), _stream())

    def fused_history(self):
        """The ring update of cache.py:1283-1286 folded into the decode attention's combine pass (None: no head scores by
        accumulated attention, nothing to record)."""
        if not self.requires_heavy_hitter:
            return None
        wsum, acc = self._window_state()
        return self.attn_history_num, self.attn_history_denom, self.attn_counter, int(self.history_window_size), acc, wsum

    def update_state(self, input_pos, k_val, v_val, is_prefill, attn, **kwargs):
        """ref: cache.py:1274-1288."""
        if not is_prefill and self._state_fused:
            self._state_fused = False
            return
        self._next_valid = False
        if is_prefill:
            self.profile_and_update(input_pos, k_val, v_val, attn, **kwargs)
        elif self.requires_heavy_hitter:
            _need_device(attn, "attn")
            a = attn.reshape(self.n_heads, -1).contiguous()
            self._ring_update(a, a.shape[1])
        else:
            assert attn is None, "Attn should be None if no attention is required."

    # ------------------------------------------------------------------ prefill (ref: cache.py:1066-1272)
    def _column_stats(self, attn, input_pos, bands):
        """-> colsum f32 [H,L], column mean dtype [H,L], {band: band sums f32 [H,L]}."""
        H, dt = self.n_heads, self.k_cache.dtype
        if isinstance(attn, AttnSummary):
            missing = [b for b in bands if b not in attn.bands]
            if missing:
                raise ColdCompressError(f"prefill attention was run without band sums for window widths {missing}")
            return attn.colsum, attn.column_mean(input_pos)[0], {b: attn.bands[b] for b in bands}
        _need_device(attn, "attn")
        a = attn.contiguous()
        Lq, L = a.shape[-2], a.shape[-1]
        colsum = torch.empty((H, L), dtype=torch.float32, device=a.device)
        _abi.call("cc_attn_colsum", _ptr(a), H, Lq, L, _DT[a.dtype], _ptr(colsum), _stream())
        mean = torch.empty((H, L), dtype=dt, device=a.device)
        _abi.call("cc_colsum_to_mean", _ptr(colsum), _ptr(input_pos.to(torch.int64).contiguous()), H, L, _DT[dt], _ptr(mean), _stream())
        out = {}
        for b in bands:
            t = torch.empty((H, L), dtype=torch.float32, device=a.device)
            _abi.call("cc_attn_bandsum", _ptr(a), H, Lq, L, _DT[a.dtype], int(b), _ptr(t), _stream())
            out[b] = t
        return colsum, mean, out

    def _column_sets(self, cum_attn, special_mask, punc_mask, total_len, L):
        """Per policy: (static column set incl. heavy hitters [H,L] bool, window width or 0).  ref: build_masks
        cache.py:1066-1136 — only the last query row of each [L,L] mask is a column set; the window part of the
        other rows is accounted for by band sums."""
        from .prompt_compression import topk_keep

        H, dev = self.n_heads, cum_attn.device
        t = torch.arange(L, device=dev)
        out = []
        for s in self.hybrid_strategies:
            name = s["strategy"]
            col = (t < self.global_tokens)
            if "special" in name:
                col = col | special_mask
            if "punc" in name:
                col = col | punc_mask
            win = 0
            last_row = col
            if "window" in name:
                assert "recent_window" in s and s["recent_window"] <= 1, \
                    "Window strategy should have recent_window expressed as a fraction <= 1."
                win = max(1, int(s["recent_window"] * total_len))
                last_row = col | (t >= L - win)
            cols = col.unsqueeze(0).expand(H, L).clone()
            if "heavy_hitter" in name:
                avail = ~last_row
                n_avail = int(avail.sum())
                import math
                num_hh = math.ceil(min(s["heavy_hitter_frac"] * total_len, n_avail))
                if num_hh > 0:
                    prio = cum_attn.float().masked_fill(~avail.unsqueeze(0), float("-inf")).contiguous()
                    keep = topk_keep(prio, num_hh)  # [H, num_hh], ties lowest-index-first
                    cols.scatter_(1, keep, True)
            if name == "full":
                cols.fill_(True)
            out.append((cols, win))
        return out

    def profile_and_update(self, input_pos, k_val, v_val, attn, **kwargs):
        input_ids = kwargs["input_ids"].reshape(-1)
        dev = self.k_cache.device
        input_ids = input_ids.to(dev)
        L, H, S, D = input_ids.shape[-1], self.n_heads, self.max_cache_length, self.head_dim
        assert S >= L
        special_mask = punc_mask = None
        if self.requires_special:
            special_mask = self.build_special_ids_mask(input_ids)
            self.num_special.copy_(special_mask.sum().to(torch.int32).view(1))
        if self.requires_punc:
            punc_mask = torch.isin(input_ids, self.punc_ids)
            self.num_punc.copy_(punc_mask.sum().to(torch.int32).view(1))
        zeros = torch.zeros(L, dtype=torch.bool, device=dev)
        sm = special_mask if special_mask is not None else zeros
        pm = punc_mask if punc_mask is not None else zeros
        needs_attn = any("heavy_hitter" in s["strategy"] or "window" in s["strategy"] for s in self.hybrid_strategies)
        bands = self.attn_bands(L)
        if attn is None:
            raise ColdCompressError("hybrid profiling needs the prefill attention (return_attn() was True)")
        colsum, cum_attn, band = self._column_stats(attn, input_pos, bands)
        # ---- score every policy per head (ref: profile_attn_heads cache.py:1160-1173)
        scoring = self._column_sets(cum_attn, sm, pm, L, L)
        scores = []
        for cols, win in scoring:
            outside = band[win] if win else torch.zeros_like(colsum)
            scores.append(torch.where(cols, colsum, outside).sum(dim=1) / L)
        # the reference's scores are model-dtype tensors (cache.py:1165-1167: .sum(-1).mean(-1) of the probabilities) and a
        # Python scalar does not promote a tensor, so the comparison of cache.py:1171 happens in the model dtype:
        # bf16(0.9) = 0.8984 is the effective threshold of a bf16 model
        scores = torch.stack(scores).to(self.k_cache.dtype)  # [n_policies, H]
        self.compressed_scores = scores
        thr = torch.tensor(self.min_recovery_frac, dtype=scores.dtype, device=scores.device)
        self.cache_strategies = (scores >= thr).int().argmax(dim=0).to(torch.int64).contiguous()
        # ---- fill mask from the chosen policy, built for the FULL cache length (cache.py:1177-1185)
        filling = self._column_sets(cum_attn, sm, pm, S, L)
        t = torch.arange(L, device=dev)
        mask_all = torch.stack([cols | (t >= L - win if win else zeros).unsqueeze(0) for cols, win in filling])  # [n,H,L]
        mask_optimal = mask_all.gather(0, self.cache_strategies.view(1, H, 1).expand(1, H, L)).squeeze(0)
        chosen = [self.hybrid_strategies[i]["strategy"] for i in self.cache_strategies.tolist()]
        self.requires_heavy_hitter = any("heavy_hitter" in n for n in chosen)
        self.requires_punc = any("punc" in n for n in chosen)
        self.requires_special = any("special" in n for n in chosen)
        # ---- kept tokens first (cache.py:1228-1246)
        order = self._partition_order(mask_optimal).contiguous()
        from .prompt_compression import gather_rows

        k_ord, v_ord = gather_rows(k_val, order), gather_rows(v_val, order)
        pos_ord = input_pos.to(dev).to(torch.int64).unsqueeze(0).expand(H, -1).gather(1, order).contiguous()
        self.cache_cts.zero_()
        self.mask.zero_()
        _abi.call("cc_prefill_fill", self._view(), _ptr(k_ord), _ptr(v_ord), _ptr(pos_ord), H, L, _stream())
        cts = mask_optimal.sum(dim=1).to(torch.int32)
        self.cache_cts.copy_(cts)
        live = torch.arange(S, device=dev).view(1, S) < cts.view(H, 1)  # [H,S]
        self.pos[0].masked_fill_(~live, -1)
        self.k_cache[0].masked_fill_(~live.unsqueeze(-1), 0)
        self.v_cache[0].masked_fill_(~live.unsqueeze(-1), 0)
        self.mask[0, :, 0, :] = live & (torch.arange(S, device=dev).view(1, S) < L)
        if hasattr(self, "special_mask"):
            self.special_mask[0, :, :L] = sm.unsqueeze(0).expand(H, -1).gather(1, order)
        if hasattr(self, "punc_mask"):
            self.punc_mask[0, :, :L] = pm.unsqueeze(0).expand(H, -1).gather(1, order)
        if self.requires_heavy_hitter:  # cache.py:1267-1272 seeds the ring with the gathered column means
            a = cum_attn.gather(1, order).contiguous()
            self._ring_update(a, L)

    def compute_statistics(self, seq_len):
        """ref: cache.py:1043-1064."""
        stats = super().compute_statistics(seq_len)
        idxs = self.cache_strategies.tolist()
        names = [self.hybrid_strategies[i]["strategy"] for i in idxs]
        stats["avg_strategy_idx"] = sum(idxs) / len(idxs)
        for n in sorted({s["strategy"] for s in self.hybrid_strategies}):
            stats[n] = names.count(n) / len(names)
        return stats


class KVCacheKeepItOdd(KVCacheHeadConstant):
    """ref: cache.py:1423-1441 (toy policy; exercises the generic caller-supplied-importances path)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "global_tokens", "recent_window"]

    def _token_importances(self, input_pos):
        p = self.pos[:, 0]
        scores = torch.zeros_like(p, dtype=torch.bfloat16)
        scores[p % 2 == 1] = 1.0
        scores[p >= input_pos - self.recent_window] = float("inf")
        return scores


class KVCacheAnalysis(KVCacheFull):
    """`debug_<strategy>` (ref: cache.py:1291-1420): the model attends over a FULL cache while a shadow cache of the
    analysed strategy is kept beside it, fed with the attention restricted to the slots it still holds; the attention
    mass it lost — 1 - sum of the probabilities of its surviving tokens, averaged over heads — is recorded per decode step
    (`attention_losses`) and reported by compute_statistics (`attention_loss`, `attention_loss@500`, ...).

    At the reference commit this class cannot be constructed: its `full_kwargs` (cache.py:1319-1324) omit `cache_bits`,
    which KVCache.__init__ reads (cache.py:181 -> AttributeError).  What is implemented — and pinned by
    tests/golden/f10_analysis_* — is the INTENDED behaviour: the reference's own code with that one keyword supplied
    (cache_bits=None for the full cache; oracle/gen_golden.py injects exactly that and nothing else).

    As in the reference: the analysed cache must return attention (heavy_hitter, hybrid with a heavy-hitter policy) —
    for the others model.py hands update_state `attn=None` at decode time and the reference fails on `attn.shape`
    (raised here as a ColdCompressError); for a head-constant shadow cache the loss would be taken from kv head 0 only
    (gather with a [1, 1, S] index, cache.py:1393)."""
    relevant_kwargs = ["max_cache_length", "max_seq_length", "cache_bits", "history_window_size", "recent_window",
                       "attn_thresholding", "global_tokens", "prompt_compression_strategy"]

    def __init__(self, max_batch_size, n_heads, head_dim, dtype=torch.bfloat16, cache_strategy="heavy_hitter", **kwargs):
        full_kwargs = {"global_tokens": 0, "max_cache_length": kwargs["max_seq_length"],
                       "prompt_compression_strategy": kwargs["prompt_compression_strategy"],
                       "cache_bits": None}  # <- the keyword the reference forgot
        super().__init__(max_batch_size, n_heads, head_dim, dtype, **full_kwargs)
        self.compressed = get_cache_constructor(cache_strategy)[0](max_batch_size, n_heads, head_dim, dtype, **kwargs)
        self.register_buffer("attention_losses", torch.full((self.max_cache_length,), -1, dtype=dtype))
        self.register_buffer("attention_loss_ctr", torch.zeros((1,), dtype=torch.int32))
        self.prompt_compressor = get_prompt_compressor_constructor(self.prompt_compression_strategy)(
            head_specific=self.compressed.head_specific, **kwargs)
        self.head_specific = self.compressed.head_specific  # compatibility check of the prompt compressor (model.py:225)

    def supports_fused_step(self):
        return False  # the full cache must hand the attention row to update_state

    def return_attn(self):
        return self.compressed.return_attn()

    def update_kv(self, input_pos, k_val, v_val, is_prefill, **kwargs):
        k, v, mask = super().update_kv(input_pos, k_val, v_val, is_prefill, **kwargs)
        # a prompt longer than the shadow cache is compressed in update_state (it may need the attention)
        if input_pos.shape[-1] < self.compressed.max_cache_length:
            self.compressed.update_kv(input_pos, k_val, v_val, is_prefill, **kwargs)
        return k, v, mask

    def reset(self):
        super().reset()
        self.compressed.reset()
        self.attention_losses.fill_(-1)
        self.attention_loss_ctr.zero_()

    def update_state(self, input_pos, k_val, v_val, is_prefill, attn, **kwargs):
        if is_prefill and input_pos.shape[-1] > self.compressed.max_cache_length:
            input_pos, k_val, v_val, attn = self.prompt_compressor(input_pos, k_val, v_val, attn=attn)
            self.compressed.update_kv(input_pos, k_val, v_val, is_prefill)
            self.compressed.update_state(input_pos, k_val, v_val, is_prefill, attn)
        elif is_prefill:  # no loss at prefill: compressed and uncompressed attention are the same
            self.compressed.update_state(input_pos, k_val, v_val, is_prefill, attn)
        else:
            if attn is None:
                raise ColdCompressError(f"debug_{type(self.compressed).__name__}: the analysed cache returns no attention, so there "
                                        "is nothing to measure (the reference fails here on `attn.shape`, cache.py:1395)")
            _need_device(attn, "attn")
            # ONE launch (cc_analysis_loss): the shadow cache's view of the attention row (unfilled slots read the last, zero,
            # column; a head-constant shadow cache reads kv head 0, like the reference's [1, 1, S] gather index), the attention
            # mass it lost — 1 - the mass of the tokens it still holds, averaged over heads — recorded at attention_loss_ctr,
            # and the counter's increment: no device scalar travels to the host
            comp = self.compressed
            Hp, S, S_full = comp.pos.shape[1], comp.max_cache_length, attn.shape[-1]
            src = attn.reshape(-1, S_full)
            src = src if src.is_contiguous() else src.contiguous()
            sub = torch.empty((1, Hp, S), dtype=attn.dtype, device=attn.device)
            _abi.call("cc_analysis_loss", _ptr(src), _ptr(comp.pos), Hp, S_full, S, _DT[attn.dtype], _ptr(sub),
                      _ptr(self.attention_losses), _ptr(self.attention_loss_ctr), self.attention_losses.numel(), _stream())
            comp.update_state(input_pos, k_val, v_val, is_prefill, sub)

    def compute_statistics(self, seq_len):
        stats = super().compute_statistics(seq_len)
        losses = self.attention_losses[: int(self.attention_loss_ctr)]
        assert not torch.any(losses == -1)
        for k in range(500, len(losses), 500):
            stats[f"attention_loss@{k}"] = losses[:k].mean().item()
        stats["attention_loss"] = losses.mean().item()
        return stats


def get_cache_constructor(cache_strategy):
    """ref: cache.py:1444-1478 -> (constructor, relevant_kwargs)."""
    table = {
        "full": KVCacheFull,
        "l2": KVCacheL2,
        "random": KVCacheRandom,
        "recent_global": KVCacheRecentGlobal,
        "heavy_hitter": KVCacheHeavyHitter,
        "hybrid": KVCacheHybrid,
        "keep_it_odd": KVCacheKeepItOdd,
    }
    if cache_strategy in table:
        cls = table[cache_strategy]
        return cls, cls.relevant_kwargs
    if cache_strategy.startswith("debug"):  # ref: cache.py:1460-1474
        name = re.sub(r"debug_+", "", cache_strategy).strip()
        rk = get_cache_constructor(name)[1] + ["prompt_compression_strategy"]

        def ctor(max_batch_size, n_heads, head_dim, dtype, **kwargs):
            return KVCacheAnalysis(max_batch_size, n_heads, head_dim, dtype, cache_strategy=name, **kwargs)

        return ctor, rk
    raise ValueError(f"Invalid cache strategy: {cache_strategy}")
