"""Prefill-time prompt compaction behind the reference's `prompt_compression.py` surface.

`get_prompt_compressor_constructor(name)(head_specific=..., **layer_kwargs)` and
`compressor(input_pos, k_val, v_val, attn=...) -> (keep_idxs, k, v, state)` as in ref:
prompt_compression.py:28-43, 233-247.  Selection (radix-select top-K with ordered output), the K/V row
gathers, row norms and the SnapKV pooling run in HIP (include/coldcompress.h); index-vector construction
for the head-constant policies is plain tensor arithmetic on the device.

Top-k tie contract: ties at the K-th value are broken lowest-index-first here; torch's
CPU `topk` order at a tie is implementation-defined.
"""
import ctypes as C

import torch

from . import _abi
from ._abi import ColdCompressError

_DT = {torch.float32: _abi.CC_DT_F32, torch.bfloat16: _abi.CC_DT_BF16, torch.float16: _abi.CC_DT_F16}
_PRIO = {torch.float32: _abi.CC_PRIO_F32, torch.bfloat16: _abi.CC_PRIO_BF16, torch.float16: _abi.CC_PRIO_F16,
         torch.int64: _abi.CC_PRIO_I64}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_device(t, what):
    if not t.is_cuda:
        raise ColdCompressError(f"{what} is on {t.device}: the HIP path needs ROCm device tensors (no CPU fallback).")


class AttnSummary:
    """What the policies actually consume from prefill attention, instead of the reference's materialised
    [1, H, L, L] probability tensor (attention_utils.py:36-54): per kv-head column sums of the group-averaged
    probabilities and their mean over the last `obs_len` query rows, both float32 [H, L]."""

    def __init__(self, colsum, obs_mean, obs_len, dtype, bands=None):
        self.colsum, self.obs_mean, self.obs_len, self.dtype = colsum, obs_mean, obs_len, dtype
        self.bands = bands or {}  # {window width in queries: band sums float32 [H, L]} (hybrid profiling)
        self.ndim = 4  # quacks like the reference's 4-D attention for `attn.ndim == 4` checks
        self.shape = (1, colsum.shape[0], colsum.shape[1], colsum.shape[1])

    def column_mean(self, input_pos):
        """ref: cache.py:704 / prompt_compression.py:191: attn.sum(dim=2) / (seq_len - input_pos) -> [1, H, L] dtype."""
        H, L = self.colsum.shape
        out = torch.empty((1, H, L), dtype=self.dtype, device=self.colsum.device)
        ip = input_pos.to(torch.int64).contiguous()
        _abi.call("cc_colsum_to_mean", _ptr(self.colsum), _ptr(ip), H, L, _DT[self.dtype], _ptr(out), _stream())
        return out

    def observation_mean(self):
        """ref: prompt_compression.py:173 attn[:, :, -obs_len:, :].mean(dim=2) -> [1, H, L] dtype."""
        return self.obs_mean.to(self.dtype).unsqueeze(0)

    def view(self, *shape):
        """The reference's model.py:413-418 runs UNCHANGED on a summary: `attn.view(bsz, n_local_heads, R, seqlen, -1)
        .mean(dim=2)`.  When the kernel was handed GQA-shaped K/V the summary is already per kv head (R groups averaged
        inside the prefill pass) and the call returns it as is; when the caller repeat_interleave'd K/V first
        (model.py:399-400), the summary has one row per QUERY head and `.mean(dim=2)` averages each group of R rows —
        on the fp32 column sums rather than per bf16 probability (same value up to the rounding of the reference's
        per-element mean)."""
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        rows, L = self.colsum.shape
        if len(shape) != 5 or shape[0] != 1 or shape[3] != L or shape[4] not in (-1, L) or shape[1] <= 0 or shape[2] <= 0:
            raise ColdCompressError(f"AttnSummary.view{tuple(shape)}: only the group view (1, H, R, L, -1) of model.py:413-418 exists")
        H, R = shape[1], shape[2]
        if H * R == rows:
            return _GroupView(self, H, R)
        if H == rows:  # already averaged over the group inside the kernel
            return _GroupView(self, H, 1)
        raise ColdCompressError(f"AttnSummary.view: {rows} summary rows cannot be seen as {H} kv heads x {R} query heads")


class _GroupView:
    """`AttnSummary.view(1, H, R, L, -1)`: the only thing the reference does with it is `.mean(dim=2)`."""

    def __init__(self, summary, H, R):
        self.summary, self.H, self.R = summary, H, R

    def mean(self, dim):
        if dim != 2:
            raise ColdCompressError("the group view of an AttnSummary supports .mean(dim=2) only (model.py:416-418)")
        s, H, R = self.summary, self.H, self.R
        if R == 1:
            return s
        L = s.colsum.shape[1]

        def grp(t):
            return t.view(H, R, L).mean(dim=1)

        return AttnSummary(grp(s.colsum), grp(s.obs_mean), s.obs_len, s.dtype, {b: grp(t) for b, t in s.bands.items()})


def topk_keep(priority, K):
    """Ascending indices of the K largest entries per row (ref: prompt_compression.py:21-26)."""
    _need_device(priority, "priority")
    if priority.dtype not in _PRIO:
        raise ColdCompressError(f"unsupported priority dtype {priority.dtype}")
    L = priority.shape[-1]
    p2 = priority.reshape(-1, L).contiguous()
    keep = torch.empty((p2.shape[0], K), dtype=torch.int64, device=priority.device)
    _abi.call("cc_topk_keep", _ptr(p2), _PRIO[priority.dtype], p2.shape[0], L, K, _ptr(keep), None, 0, _stream())
    return keep


def gather_rows(x, keep):
    """x [1, H, L, D], keep [K] or [H, K] -> [1, H, K, D]."""
    _need_device(x, "k/v")
    _, H, L, D = x.shape
    xc = x.contiguous()
    k2 = keep.reshape(-1, keep.shape[-1]).contiguous()
    K = k2.shape[1]
    out = torch.empty((1, H, K, D), dtype=x.dtype, device=x.device)
    _abi.call("cc_gather_rows", _ptr(xc), _ptr(k2), k2.shape[0], H, L, K, D, _DT[x.dtype], _ptr(out), _stream())
    return out


class PromptCompressor:
    def __init__(self, head_specific, **kwargs) -> None:
        for key, value in kwargs.items():
            setattr(self, key, value)
        self.head_specific = head_specific
        assert self.is_compatible(), \
            f"Prompt compressor ({self.__class__.__name__}) is not compatible with the chosen cache strategy."

    def _recent_global_mask(self, input_pos):
        seq_len = input_pos.shape[-1]
        return torch.logical_or(input_pos < self.global_tokens, input_pos >= seq_len - self.recent_window)

    def _keep_idxs(self, priority):
        return topk_keep(priority, self.max_cache_length).squeeze(0)

    def __call__(self, input_pos, k_val, v_val, **kwargs):
        priority = self._token_importances(input_pos, k_val, v_val, **kwargs)
        keep_idxs = self._keep_idxs(priority)
        k_val, v_val = self._filter_kv(keep_idxs, k_val, v_val)
        return keep_idxs, k_val, v_val, self._update_state(keep_idxs, input_pos, **kwargs)

    def _update_state(self, keep_idxs, input_pos, **kwargs):
        return None

    def _filter_kv(self, keep_idxs, k_val, v_val):
        return gather_rows(k_val, keep_idxs), gather_rows(v_val, keep_idxs)

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        raise NotImplementedError

    def is_compatible(self) -> bool:
        raise NotImplementedError


class PromptCompressorHeadConstant(PromptCompressor):
    def is_compatible(self) -> bool:
        return True


class PromptCompressorHeadSpecific(PromptCompressor):
    def is_compatible(self) -> bool:
        return self.head_specific


class PromptCompressorFull(PromptCompressorHeadConstant):
    """ref: prompt_compression.py:91-106 (pass-through)."""

    def __call__(self, input_pos, k_val, v_val, **kwargs):
        return input_pos, k_val, v_val, None

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        raise Exception("This method should not be called!")


class PromptCompressorRandom(PromptCompressorHeadConstant):
    """ref: prompt_compression.py:109-125."""

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        seq_len = input_pos.shape[-1]
        save = self._recent_global_mask(input_pos)
        priority = input_pos.to(torch.int64).masked_fill(save, seq_len).masked_fill(~save, -seq_len)
        return priority + torch.randperm(seq_len, device=priority.device)


class PromptCompressorRecentGlobal(PromptCompressorHeadConstant):
    """ref: prompt_compression.py:128-145."""

    def __init__(self, head_specific, **kwargs) -> None:
        super().__init__(head_specific, **kwargs)
        assert self.max_cache_length - self.global_tokens > 0, (
            f"Number of global tokens ({self.global_tokens}) cannot exceed the max cache length ({self.max_cache_length})")

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        ip = input_pos.to(torch.int64)
        return ip.masked_fill(ip < self.global_tokens, ip.shape[-1])


class PromptCompressorHeavyHitter(PromptCompressorHeadSpecific):
    """SnapKV (ref: prompt_compression.py:148-194)."""

    def __init__(self, head_specific, **kwargs) -> None:
        super().__init__(head_specific, **kwargs)
        self.kernel_size = 5
        self.observation_len = 16

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        attn = kwargs["attn"]
        seq_len = input_pos.shape[-1]
        obs_len = min(self.observation_len, seq_len)
        if isinstance(attn, AttnSummary):
            assert attn.obs_len == obs_len, "prefill attention was run with a different observation window"
            obs = attn.observation_mean()
        else:  # a materialised [1, H, L, L] tensor from a reference-style caller
            obs = attn[:, :, -obs_len:, :].mean(dim=2)
        _need_device(obs, "attn")
        obs = obs.contiguous()
        _, H, L = obs.shape
        out = torch.empty_like(obs)
        _abi.call("cc_snapkv_priority", _ptr(obs), H, L, _DT[obs.dtype], obs_len, int(self.global_tokens), _ptr(out),
                  _stream())
        return out

    def _update_state(self, keep_idxs, input_pos, **kwargs):
        attn = kwargs["attn"]
        if isinstance(attn, AttnSummary):
            cum = attn.column_mean(input_pos)
        else:
            _need_device(attn, "attn")
            _, H, Lq, L = attn.shape
            colsum = torch.empty((H, L), dtype=torch.float32, device=attn.device)
            _abi.call("cc_attn_colsum", _ptr(attn.contiguous()), H, Lq, L, _DT[attn.dtype], _ptr(colsum), _stream())
            cum = torch.empty((1, H, L), dtype=attn.dtype, device=attn.device)
            ip = input_pos.to(torch.int64).contiguous()
            _abi.call("cc_colsum_to_mean", _ptr(colsum), _ptr(ip), H, L, _DT[attn.dtype], _ptr(cum), _stream())
        _, H, L = cum.shape
        K = self.max_cache_length
        keep = keep_idxs.reshape(H, K).contiguous()
        out = torch.empty((1, H, K), dtype=cum.dtype, device=cum.device)
        _abi.call("cc_gather_vec", _ptr(cum), _ptr(keep), H, L, K, _DT[cum.dtype], _ptr(out), _stream())
        return out


class PromptCompressorL2(PromptCompressorHeadSpecific):
    """ref: prompt_compression.py:197-209."""

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        _need_device(k_val, "k_val")
        _, H, L, D = k_val.shape
        prio = torch.empty((1, H, L), dtype=k_val.dtype, device=k_val.device)
        _abi.call("cc_row_l2_norm", _ptr(k_val.contiguous()), H, L, D, _DT[k_val.dtype], 1, _ptr(prio), _stream())
        save = self._recent_global_mask(input_pos).view(1, 1, -1)
        return prio.masked_fill(save, float("inf"))


class PromptCompressorKeepItOdd(PromptCompressorHeadConstant):
    """ref: prompt_compression.py:212-230."""

    def _token_importances(self, input_pos, k_val, v_val, **kwargs):
        seq_len = input_pos.shape[-1]
        ip = input_pos.to(torch.int64)
        priority = ip.masked_fill(self._recent_global_mask(ip), seq_len * 2)
        priority[ip % 2 == 0] -= seq_len
        return priority


def get_prompt_compressor_constructor(strategy):
    table = {
        "full": PromptCompressorFull,
        "recent_global": PromptCompressorRecentGlobal,
        "heavy_hitter": PromptCompressorHeavyHitter,
        "l2": PromptCompressorL2,
        "random": PromptCompressorRandom,
        "keep_it_odd": PromptCompressorKeepItOdd,
    }
    if strategy not in table:
        raise ValueError(f"Unknown prompt compression strategy: {strategy}")
    return table[strategy]
