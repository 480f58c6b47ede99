"""GPU tests of the fused caller-glue kernels (cc_add_rmsnorm, cc_qkv_rope, cc_silu_mul) against (a) the eager
PyTorch formulas the reference's model.py uses, evaluated on the same device, and (b) the CPU oracle.
RoPE / split / layout and the residual add are bit-exact; RMSNorm and SiLU are within one ulp of the model dtype
(the fp32 reduction order of torch.mean and the exp implementation are not part of the reference's contract)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import DT_CODE, from_np, to_np

pytestmark = pytest.mark.gpu
DEV = "cuda"
ULP = {torch.float32: 2e-6, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}


def _eager_rmsnorm(x, w, eps, delta=None):
    h = x if delta is None else x + delta
    hf = h.float()
    return h, (hf * torch.rsqrt(torch.mean(hf * hf, dim=-1, keepdim=True) + eps)).type_as(h) * w


@pytest.mark.parametrize("dtype,T,dim", [(torch.bfloat16, 1, 4096), (torch.bfloat16, 37, 4096), (torch.float32, 5, 64),
                                         (torch.float16, 3, 1024), (torch.bfloat16, 2, 8192)])
def test_add_rmsnorm(oracle, dtype, T, dim):
    from cold_compress_amd.harness import glue

    gen = torch.Generator().manual_seed(dim + T)
    x = torch.randn(1, T, dim, generator=gen).to(dtype).to(DEV)
    d = torch.randn(1, T, dim, generator=gen).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(dim, generator=gen)).to(dtype).to(DEV)
    for delta in (None, d):
        h, n = glue.add_rmsnorm(x, w, 1e-5, delta)
        h_ref, n_ref = _eager_rmsnorm(x, w, 1e-5, delta)
        assert torch.equal(h, h_ref)
        assert torch.allclose(n.float(), n_ref.float(), rtol=ULP[dtype], atol=1e-6)
    code = DT_CODE[dtype]
    out = np.zeros((T, dim), np.float32 if code == 0 else np.uint16)
    hh = np.zeros_like(out)
    oracle.call("cc_add_rmsnorm", oracle.ptr(to_np(x.cpu()[0])), oracle.ptr(to_np(d.cpu()[0])), oracle.ptr(to_np(w.cpu())), T, dim,
                1e-5, code, oracle.ptr(hh), oracle.ptr(out), None)
    assert torch.equal(from_np(hh, dtype), h.cpu()[0])
    assert torch.allclose(from_np(out, dtype).float(), n.cpu()[0].float(), rtol=ULP[dtype], atol=1e-6)


@pytest.mark.parametrize("dtype,T,HQ,H,D", [(torch.bfloat16, 1, 32, 8, 128), (torch.bfloat16, 50, 32, 8, 128),
                                            (torch.float32, 7, 4, 2, 16), (torch.float16, 3, 6, 3, 64)])
def test_qkv_rope_bit_exact(oracle, dtype, T, HQ, H, D):
    from cold_compress_amd.harness import glue
    from cold_compress_amd.harness.model import precompute_freqs_cis

    gen = torch.Generator().manual_seed(T * D)
    qkv = torch.randn(1, T, (HQ + 2 * H) * D, generator=gen).to(dtype).to(DEV)
    table = precompute_freqs_cis(4096, D, 500000, dtype).to(DEV)
    pos = torch.randint(0, 4096, (T,), generator=gen).to(DEV)
    fc = table[pos]
    q, k, v = glue.qkv_rope(qkv, fc, HQ, H, D)
    qs, ks, vs = qkv.split([HQ * D, H * D, H * D], dim=-1)
    import host_glue

    q_ref = host_glue.rope(qs.view(1, T, HQ, D), fc).transpose(1, 2)
    k_ref = host_glue.rope(ks.view(1, T, H, D), fc).transpose(1, 2)
    v_ref = vs.view(1, T, H, D).transpose(1, 2)
    assert torch.equal(q, q_ref) and torch.equal(k, k_ref) and torch.equal(v, v_ref)
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    code = DT_CODE[dtype]
    es = np.float32 if code == 0 else np.uint16
    qo, ko, vo = np.zeros((HQ, T, D), es), np.zeros((H, T, D), es), np.zeros((H, T, D), es)
    oracle.call("cc_qkv_rope", oracle.ptr(to_np(qkv.cpu()[0])), oracle.ptr(to_np(fc.cpu())), T, HQ, H, D, code, oracle.ptr(qo),
                oracle.ptr(ko), oracle.ptr(vo), None)
    assert torch.equal(from_np(qo, dtype), q.cpu()[0]) and torch.equal(from_np(ko, dtype), k.cpu()[0])
    assert torch.equal(from_np(vo, dtype), v.cpu()[0])


@pytest.mark.parametrize("dtype,n", [(torch.bfloat16, 14336), (torch.float32, 1000), (torch.float16, 4099), (torch.bfloat16, 40 * 14336)])
def test_silu_mul(oracle, dtype, n):
    from cold_compress_amd.harness import glue

    gen = torch.Generator().manual_seed(n)
    a = (3 * torch.randn(n, generator=gen)).to(dtype).to(DEV)
    b = torch.randn(n, generator=gen).to(dtype).to(DEV)
    out = glue.silu_mul(a, b)
    ref = F.silu(a) * b
    assert torch.allclose(out.float(), ref.float(), rtol=2 * ULP[dtype], atol=1e-6)
    code = DT_CODE[dtype]
    o = np.zeros(n, np.float32 if code == 0 else np.uint16)
    oracle.call("cc_silu_mul", oracle.ptr(to_np(a.cpu())), oracle.ptr(to_np(b.cpu())), n, code, oracle.ptr(o), None)
    assert torch.allclose(from_np(o, dtype).float(), out.cpu().float(), rtol=2 * ULP[dtype], atol=1e-6)


@pytest.mark.parametrize("dt,V", [(torch.bfloat16, 128256), (torch.float32, 128), (torch.float16, 32000), (torch.float32, 50257)])
def test_softmax_argmax_matches_torch(dt, V):
    """cc_softmax_argmax vs torch.softmax / torch.argmax on the device (generation_utils.py:136-142): probabilities
    within one ulp of the dtype at the largest value, token = first index of the largest ROUNDED probability."""
    from cold_compress_amd.harness import glue

    g = torch.Generator().manual_seed(V)
    logits = (torch.randn(V, generator=g) * 3).to(dt).to(DEV)
    probs, idx = glue.softmax_argmax(logits)
    ref = torch.softmax(logits, dim=-1)
    torch.cuda.synchronize()
    ulp = {torch.float32: 1.2e-7, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[dt]
    assert (probs.float() - ref.float()).abs().max() <= 2 * ulp * float(ref.max()) + 1e-12
    assert int(idx) == int(torch.argmax(probs))  # exact given OUR probabilities: first maximal element
    assert float(ref[int(idx)]) >= float(ref.max()) * (1 - 4 * ulp)
    # engineered tie: two equal maxima -> the first index wins
    logits[7] = logits[V - 3] = logits.max() + 1
    probs, idx = glue.softmax_argmax(logits)
    assert int(idx) == 7


@pytest.mark.parametrize("dt,V", [(torch.bfloat16, 128256), (torch.float32, 300)])
def test_softmax_argmax_on_nan_logits_returns_a_valid_index(dt, V):
    """Garbage logits (behind a failed single-launch step the attention output is whatever its buffer held) must still sample a token
    INSIDE the vocabulary: torch.argmax treats NaN as the maximum and returns the first one's index — so does cc_softmax_argmax (r5:
    it used to return -1, and the next token's embedding lookup asserted on the device before the host had seen the status word)."""
    from cold_compress_amd.harness import glue

    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(V, generator=g) * 3).to(dt)
    logits[V // 2] = float("nan")  # every probability becomes NaN (the sum is NaN): the first index wins
    probs, idx = glue.softmax_argmax(logits.to(DEV))
    torch.cuda.synchronize()
    ref = torch.softmax(logits.float(), dim=-1).to(dt)
    assert 0 <= int(idx) < V and int(idx) == int(torch.argmax(ref)) == 0
    all_nan = torch.full((V,), float("nan"), dtype=dt, device=DEV)
    _, idx = glue.softmax_argmax(all_nan)
    assert int(idx) == 0
