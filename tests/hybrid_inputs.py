"""Synthetic prefill inputs with three kinds of kv heads, for the full-size hybrid profiling test: sharply local attention
(the window policy suffices), weakly local attention plus a set of heavy-hitter keys (window + heavy hitters), and
near-uniform attention (only `full` recovers the mass).  Head h is of kind h % 3.  Deterministic in (L, H, R, D, seed)."""
import math

import torch


def make_inputs(L, H, R, D=128, seed=0, dtype=torch.bfloat16):
    assert D >= 100
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L, dtype=torch.float64)
    nf = 48
    w = torch.rand(nf, generator=g, dtype=torch.float64) * (480.0 / L)  # incommensurate frequencies: no aliasing
    ang = t[:, None] * w[None, :]
    u = torch.cat([torch.cos(ang), torch.sin(ang)], 1).float() / math.sqrt(nf)  # [L, 96] unit vectors; u_i . u_s decays with |i - s|
    k = torch.zeros(H, L, D)
    q = torch.zeros(H * R, L, D)
    root_d = math.sqrt(D)
    for h in range(H):
        kind = h % 3
        if kind == 0:  # sharply local
            a = math.sqrt(40.0 * root_d)
            k[h, :, :96] = u * a
            for r in range(R):
                q[h * R + r, :, :96] = u * a * (1 + 0.1 * r)
        elif kind == 1:  # weakly local + heavy hitters
            a = math.sqrt(9.0 * root_d)
            k[h, :, :96] = u * a
            hh = torch.randperm(L, generator=g)[: L // 8]
            k[h, hh, 96] = 6.0
            for r in range(R):
                q[h * R + r, :, :96] = u * a
                q[h * R + r, :, 96] = 1.5 * root_d
        else:  # near-uniform
            k[h] = 0.3 * torch.randn(L, D, generator=g)
            for r in range(R):
                q[h * R + r] = 0.3 * torch.randn(L, D, generator=g)
    k = k + 0.02 * torch.randn(H, L, D, generator=g)
    v = torch.randn(H, L, D, generator=g)
    return q.to(dtype), k.to(dtype), v.to(dtype)
