"""Llama/Qwen2-style decoder used as the CALLER of the hot path (the build's counterpart of the reference's
model.py, which never ships).  Dense projections, norms and RoPE are ordinary PyTorch-ROCm ops (hipBLASLt
GEMMs — out of scope for hand kernels); every cache / attention step goes through the HIP path.

Conventions that fixture F1's logits depend on (ref: model.py):
  parameter names/layout (:179-184, :335-338, :438-440), pre-norm blocks (:317-327), RMSNorm in fp32
  (:452-457), interleaved-pair RoPE with the table stored in the model dtype (:460-519), decode inserts
  into the cache BEFORE attending (:391-411), probabilities averaged per query group (:413-418).
"""
import math
import os
from collections import defaultdict
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import glue
from ..attention_utils import scaled_dot_product_attention
from ..cache import (KVCacheFull, KVCacheHeavyHitter, KVCacheHybrid, KVCacheL2, KVCacheRandom, KVCacheRecentGlobal, flush_quantized,
                     get_cache_constructor)
from ..prompt_compression import get_prompt_compressor_constructor


def find_multiple(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class ModelArgs:
    block_size: int = 2048
    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: Optional[int] = None
    n_local_heads: int = -1
    head_dim: int = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5
    attention_bias: bool = False
    max_length: int = 4096
    rope_scaling: Optional[Dict[str, Any]] = None

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        if self.intermediate_size is None:
            self.intermediate_size = find_multiple(int(2 * (4 * self.dim) / 3), 256)
        self.head_dim = self.dim // self.n_head


_LLAMA31_SCALING = {"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                    "original_max_position_embeddings": 8192, "rope_type": "llama3"}

# Shapes the BASELINE configs name (ref: model.py:103-131, :90-92; the 70B/128k-vocab shape is SURVEY C5's).
CONFIGS = {
    "tiny": dict(block_size=256, vocab_size=128, n_layer=2, n_head=4, n_local_heads=2, dim=64, intermediate_size=128),
    "Meta-Llama-3-8B-Instruct": dict(block_size=8192, n_layer=32, n_head=32, n_local_heads=8, dim=4096,
                                     intermediate_size=14336, vocab_size=128256, rope_base=500000, max_length=8192),
    "Meta-Llama-3.1-8B-Instruct": dict(block_size=131072, n_layer=32, n_head=32, n_local_heads=8, dim=4096,
                                       intermediate_size=14336, vocab_size=128256, rope_base=500000, max_length=131072,
                                       rope_scaling=_LLAMA31_SCALING),
    "Llama-3-70B-shape": dict(block_size=131072, n_layer=80, n_head=64, n_local_heads=8, dim=8192,
                              intermediate_size=28672, vocab_size=128256, rope_base=500000, max_length=131072,
                              rope_scaling=_LLAMA31_SCALING),
    "Qwen2-7B-Instruct": dict(block_size=32768, n_layer=28, n_head=28, n_local_heads=4, dim=3584, intermediate_size=18944,
                              vocab_size=152064, rope_base=1000000, attention_bias=True, norm_eps=1e-6, max_length=32768),
}


def precompute_freqs_cis(seq_len, n_elem, base=10000, dtype=torch.bfloat16, rope_scaling=None) -> Tensor:
    """ref: model.py:460-504 — [seq_len, n_elem/2, 2] (cos, sin), built in fp32, STORED in the model dtype."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    if rope_scaling is not None:
        assert rope_scaling["rope_type"] == "llama3", "Only Llama 3.1 scaling is supported"
        orig = rope_scaling["original_max_position_embeddings"]
        lo_wl, hi_wl = orig / rope_scaling["low_freq_factor"], orig / rope_scaling["high_freq_factor"]
        out = []
        for f in freqs:
            wl = 2 * math.pi / f
            if wl < hi_wl:
                out.append(f)
            elif wl > lo_wl:
                out.append(f / rope_scaling["factor"])
            else:
                smooth = (orig / wl - rope_scaling["low_freq_factor"]) / (
                    rope_scaling["high_freq_factor"] - rope_scaling["low_freq_factor"])
                out.append((1 - smooth) * f / rope_scaling["factor"] + smooth * f)
        freqs = torch.tensor(out)
    ang = torch.outer(torch.arange(seq_len), freqs)
    cis = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([cis.real, cis.imag], dim=-1).to(dtype=dtype)




class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x: Tensor, delta: Optional[Tensor] = None):
        """-> (h, norm(h)) with h = x + delta (the pending residual of the previous sub-block), one fused launch."""
        return glue.add_rmsnorm(x, self.weight, self.eps, delta)


class FeedForward(nn.Module):
    def __init__(self, config: ModelArgs) -> None:
        super().__init__()
        self.w1 = nn.Linear(config.dim, config.intermediate_size, bias=False)
        self.w3 = nn.Linear(config.dim, config.intermediate_size, bias=False)
        self.w2 = nn.Linear(config.intermediate_size, config.dim, bias=False)

    def forward(self, x: Tensor, fused=None) -> Tensor:
        """`fused = (delta, norm, h_out)`: single-token decode on the device — RMSNorm(x + delta) is the prologue of
        the w1/w3 pass (h_out <- x + delta), SwiGLU its epilogue; w2 follows as a plain streamed GEMV."""
        if fused is not None:
            delta, norm, h_out = fused
            g = glue.gemv_fused(self.w1.weight, x, w3=self.w3.weight, delta=delta, norm_weight=norm.weight, eps=norm.eps, h_out=h_out)
            return glue.gemv_fused(self.w2.weight, g).view(1, 1, -1)
        return self.w2(glue.silu_mul(self.w1(x), self.w3(x)))


class Attention(nn.Module):
    def __init__(self, config: ModelArgs):
        super().__init__()
        assert config.dim % config.n_head == 0
        total = (config.n_head + 2 * config.n_local_heads) * config.head_dim
        self.wqkv = nn.Linear(config.dim, total, bias=config.attention_bias)
        self.wo = nn.Linear(config.dim, config.dim, bias=False)
        self.kv_cache = None
        self.prompt_compressor = None
        self.n_head, self.head_dim = config.n_head, config.head_dim
        self.n_local_heads, self.dim = config.n_local_heads, config.dim
        self.fuse_state_update = True  # fold cache.py:690-723 into the decode attention combine pass
        self.fuse_decode_step = True   # whole update_kv + attention + update_state in one launch (two where the shape does not allow one)
        # ... with RMSNorm + wqkv + RoPE folded into that launch where the shape allows (cc_decode_step_qkv_rc).  OPT-IN (CC_FUSE_QKV=1 or
        # this attribute): bit-identical results, but at the benchmark's shape 0.5-0.8 us per layer SLOWER than the two launches it
        # replaces (r5: profiles/r05_overlap_probe.md — 19.0 us against 18.3-18.6), so the pair stays the default
        self.fuse_qkv_step = os.environ.get("CC_FUSE_QKV", "0") == "1"

    def compress_prompt(self, input_pos, k_val, v_val, attn):
        if self.kv_cache.max_cache_length < input_pos.shape[0]:
            return self.prompt_compressor(input_pos, k_val, v_val, attn=attn)
        return input_pos, k_val, v_val, attn

    def forward(self, x, input_ids, freqs_cis, mask, is_prefill, input_pos=None, attn_top_k=1.0, fused=None):
        """The glue of ref: model.py:363-432, GQA-aware (no repeat_interleave).
        `fused = (delta, norm, h_out)`: single-token decode on the device — x is the un-normalised residual stream,
        RMSNorm(x + delta) is the prologue of the wqkv pass (h_out <- x + delta), RoPE its epilogue."""
        bsz, seqlen, _ = x.shape
        if fused is not None:
            delta, norm, h_out = fused
            HQ, H, D = self.n_head, self.n_local_heads, self.head_dim
            cache = self.kv_cache
            if (self.fuse_qkv_step and self.fuse_decode_step and attn_top_k == 1.0 and D == 128 and hasattr(cache, "qkv_step_available")
                    and cache.supports_fused_step() and cache.qkv_step_available(HQ, x.shape[-1])):
                # ONE launch for norm + wqkv + RoPE + update_kv + attention + update_state: the cache's K / V rows stream in the
                # shadow of the projection's weights (q / k / v bit-identical to the two launches below)
                y = cache.decode_step_qkv(self.wqkv.weight, self.wqkv.bias, x, delta, norm.weight, norm.eps, h_out, freqs_cis, input_pos, HQ)
                return glue.gemv_fused(self.wo.weight, y).view(1, 1, -1)
            qkv = glue.gemv_fused(self.wqkv.weight, x, delta=delta, norm_weight=norm.weight, eps=norm.eps, h_out=h_out,
                                  bias=self.wqkv.bias, freqs=freqs_cis, rope_rows=(HQ + H) * D, head_dim=D)
            q = qkv[: HQ * D].view(1, HQ, 1, D)
            k = qkv[HQ * D: (HQ + H) * D].view(1, H, 1, D)
            v = qkv[(HQ + H) * D:].view(1, H, 1, D)
        else:
            # split + RoPE(q, k) + head-major layout in one launch (ref: model.py:375-387)
            q, k, v = glue.qkv_rope(self.wqkv(x), freqs_cis, self.n_head, self.n_local_heads, self.head_dim)
        cache = self.kv_cache
        ck = {"input_ids": input_ids}
        if (not is_prefill and self.fuse_decode_step and type(cache) in (KVCacheHeavyHitter, KVCacheRecentGlobal, KVCacheFull, KVCacheRandom, KVCacheL2,
                                                                          KVCacheHybrid)
                and cache.supports_fused_step() and attn_top_k == 1.0):
            # one or two launches per layer: insert folded into the K/V streaming pass, history update + next eviction
            # scoring folded into the combine pass (bit-identical to the three-call sequence below)
            y = cache.decode_step(q, k, v, input_pos, input_ids=input_ids) if type(cache) is KVCacheHybrid else cache.decode_step(q, k, v, input_pos)
        elif not is_prefill:
            kc, vc, kv_mask = cache.update_kv(input_pos, k, v, False, **ck)  # insert first, then attend
            hist = cache.fused_history() if (self.fuse_state_update and type(cache) in (KVCacheHeavyHitter, KVCacheHybrid)
                                             and attn_top_k == 1.0) else None
            fuse = hist is not None
            y, attn = scaled_dot_product_attention(
                q, kc, vc, attn_mask=kv_mask, attn_top_k=attn_top_k, return_attn=cache.return_attn() and not fuse,
                group_mean=True, history=hist)
            if fuse:
                cache._state_fused = True
            cache.update_state(input_pos, k, v, False, attn, **ck)
        else:
            bands = cache.attn_bands(seqlen) if hasattr(cache, "attn_bands") else ()  # hybrid profiling side outputs
            y, attn = scaled_dot_product_attention(q, k, v, attn_mask=mask, return_attn=cache.return_attn(),
                                                   is_causal=True, bands=bands)
            input_pos, k, v, attn = self.compress_prompt(input_pos, k, v, attn)
            cache.update_kv(input_pos, k, v, True, **ck)
            cache.update_state(input_pos, k, v, True, attn, **ck)
        if fused is not None:
            return glue.gemv_fused(self.wo.weight, y).view(1, 1, -1)
        y = y.transpose(1, 2).contiguous().view(bsz, seqlen, self.dim)
        return self.wo(y)


class TransformerBlock(nn.Module):
    def __init__(self, config: ModelArgs) -> None:
        super().__init__()
        self.attention = Attention(config)
        self.feed_forward = FeedForward(config)
        self.ffn_norm = RMSNorm(config.dim, config.norm_eps)
        self.attention_norm = RMSNorm(config.dim, config.norm_eps)
        self.fuse_gemv = True  # single-token decode: hand-written streamed GEMVs with the glue fused (cc_gemv_fused)

    def forward(self, x, delta, input_ids, input_pos, is_prefill, freqs_cis, mask, attn_top_k=1.0):
        """Pre-norm block (ref: model.py:317-327) with the residual adds folded into the norms:
        takes (x, pending residual delta) and returns (h, f) with the block output being h + f."""
        att, ffn = self.attention, self.feed_forward
        if (not is_prefill and self.fuse_gemv and x.shape[1] == 1 and x.is_cuda
                and glue.gemv_supported(att.wqkv.weight, att.wo.weight, ffn.w1.weight, ffn.w3.weight, ffn.w2.weight)):
            # decode on the device: six launches per layer — norm + wqkv + RoPE | K/V streaming pass | combine | wo |
            # norm + w1/w3 + SwiGLU | w2 — the residual adds ride in the norm prologues (h1, h2 are their outputs)
            h1 = torch.empty_like(x)
            a = att(x, input_ids, freqs_cis, mask, False, input_pos, attn_top_k=attn_top_k, fused=(delta, self.attention_norm, h1))
            h2 = torch.empty_like(x)
            return h2, ffn(h1, fused=(a, self.ffn_norm, h2))
        x, n1 = self.attention_norm(x, delta)  # x <- x + delta
        a = att(n1, input_ids, freqs_cis, mask, is_prefill, input_pos, attn_top_k=attn_top_k)
        h, n2 = self.ffn_norm(x, a)  # h = x + attn
        return h, ffn(n2)


class Transformer(nn.Module):
    def __init__(self, config: ModelArgs) -> None:
        super().__init__()
        self.config = config
        self.tok_embeddings = nn.Embedding(config.vocab_size, config.dim)
        self.layers = nn.ModuleList(TransformerBlock(config) for _ in range(config.n_layer))
        self.norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.output = nn.Linear(config.dim, config.vocab_size, bias=False)
        self.freqs_cis: Optional[Tensor] = None
        self.max_batch_size = 1
        self.batch_quant_flush = True  # False: one round-trip launch per layer, at the start of its next update

    @classmethod
    def from_name(cls, name: str):
        return cls(ModelArgs(**CONFIGS[name]))

    def setup_caches(self, **kwargs):
        """ref: model.py:191-233 — one cache + one prompt compressor per layer, each given only its relevant kwargs."""
        cache_strategy = kwargs.pop("cache_strategy")
        head_dim = self.config.dim // self.config.n_head
        dtype = self.output.weight.dtype
        layerwise = {"max_cache_length", "recent_window", "prompt_compression_strategy"}
        for i, b in enumerate(self.layers):
            ctor, relevant = get_cache_constructor(cache_strategy=cache_strategy[i])
            lk = {k: kwargs[k][i] if k in layerwise else kwargs[k] for k in relevant}
            if kwargs.get("cache_quant_mode") and "cache_bits" in relevant:  # our extension: the fused quantised cache
                lk["cache_quant_mode"] = kwargs["cache_quant_mode"]
            b.attention.kv_cache = ctor(self.max_batch_size, self.config.n_local_heads, head_dim, dtype, **lk)
            b.attention.prompt_compressor = get_prompt_compressor_constructor(
                kwargs["prompt_compression_strategy"][i])(head_specific=b.attention.kv_cache.head_specific, **lk)
        self.freqs_cis = precompute_freqs_cis(self.config.block_size, head_dim, self.config.rope_base, dtype,
                                              self.config.rope_scaling)
        dev = self.output.weight.device
        self.freqs_cis = self.freqs_cis.to(dev)

    def reset_caches(self):
        for layer in self.layers:
            layer.attention.kv_cache.reset()

    def min_cache_length(self):
        return min(layer.attention.kv_cache.max_cache_length for layer in self.layers)

    def get_cache_stats(self, prompt_len, gen_len):
        """ref: model.py:245-263."""
        stats, avgs, mem = {}, defaultdict(list), 0
        for i, layer in enumerate(self.layers):
            st = layer.attention.kv_cache.compute_statistics(seq_len=torch.tensor(prompt_len + gen_len))
            mem += st.pop("cache_memory_gb")
            for k, v in st.items():
                stats[f"{k}_{i}"] = v
                avgs[k].append(v)
        for k, v in avgs.items():
            stats[f"{k}_avg"] = sum(v) / len(v)
        stats["cache_memory_gb"] = mem
        return stats

    def forward(self, idx, input_pos, is_prefill, mask=None, attn_top_k=1.0) -> Tensor:
        assert self.freqs_cis is not None, "Caches must be initialized first"
        freqs_cis = self.freqs_cis[input_pos]
        x, delta = self.tok_embeddings(idx), None
        for layer in self.layers:
            x, delta = layer(x, delta, idx, input_pos, is_prefill, freqs_cis, mask, attn_top_k=attn_top_k)
        if self.batch_quant_flush:  # --cache_bits (reference mode): every layer's round trip as ONE launch behind the last layer
            flush_quantized([layer.attention.kv_cache for layer in self.layers])
        if (not is_prefill and x.shape[1] == 1 and x.is_cuda and self.layers[0].fuse_gemv and self.output.bias is None
                and glue.gemv_supported(self.output.weight)):
            # final RMSNorm (with the last pending residual) fused into the streamed LM head
            return glue.gemv_fused(self.output.weight, x, delta=delta, norm_weight=self.norm.weight, eps=self.norm.eps).view(1, 1, -1)
        return self.output(self.norm(x, delta)[1])
