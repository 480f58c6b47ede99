"""Host-side eager twins of the three caller-glue formulas (residual + RMSNorm, QKV split + RoPE, SwiGLU gate), for TESTS
only: the gloo tensor-parallel test runs the model wiring on CPU tensors with them (and a test-local attention double),
and the GPU glue tests use them as the reference the HIP kernels are compared with.  The product package has no host
path: CPU tensors reaching cold_compress_amd.harness.glue raise.
"""
import torch
import torch.nn.functional as F


def rope(x, freqs_cis):
    """x [B, T, heads, D]; freqs_cis [T, D/2, 2] (cos, sin): rotate adjacent pairs in fp32, cast back."""
    pairs = x.float().unflatten(-1, (-1, 2))
    cos, sin = freqs_cis[..., 0].view(1, pairs.size(1), 1, -1), freqs_cis[..., 1].view(1, pairs.size(1), 1, -1)
    re = pairs[..., 0] * cos - pairs[..., 1] * sin
    im = pairs[..., 1] * cos + pairs[..., 0] * sin
    return torch.stack((re, im), dim=-1).flatten(-2).type_as(x)


def add_rmsnorm(x, weight, eps, delta=None):
    h = x if delta is None else x + delta
    hf = h.float()
    return h, (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps)).type_as(h) * weight


def qkv_rope(qkv, freqs_cis, n_head, n_local_heads, head_dim):
    B, T, _ = qkv.shape
    q, k, v = qkv.split([n_head * head_dim, n_local_heads * head_dim, n_local_heads * head_dim], dim=-1)
    q = rope(q.view(B, T, n_head, head_dim), freqs_cis).transpose(1, 2)
    k = rope(k.view(B, T, n_local_heads, head_dim), freqs_cis).transpose(1, 2)
    return q, k, v.view(B, T, n_local_heads, head_dim).transpose(1, 2)


def silu_mul(a, b):
    return F.silu(a) * b


def install(glue_module):
    """Swap the device entry points of cold_compress_amd.harness.glue for the host twins (CPU model-wiring tests)."""
    glue_module.add_rmsnorm, glue_module.qkv_rope, glue_module.silu_mul = add_rmsnorm, qkv_rope, silu_mul
