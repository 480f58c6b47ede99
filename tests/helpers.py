"""Shared test helpers: golden-fixture loading and dtype plumbing."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DT_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
DT_FROM_NAME = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def load_golden(name):
    """npz -> dict of torch tensors (16-bit floats restored from their uint16 bit patterns)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for k in z.files:
        if k.endswith("__dtype"):
            continue
        a = z[k]
        if k + "__dtype" in z.files:
            dt = torch.bfloat16 if str(z[k + "__dtype"]) == "bf16" else torch.float16
            out[k] = torch.from_numpy(a.view(np.int16).copy()).view(dt)
        elif a.dtype.kind in "US":
            out[k] = str(a)
        elif a.ndim == 0 and a.dtype.kind in "iub":
            out[k] = int(a)
        elif a.ndim == 0 and a.dtype.kind == "f":
            out[k] = float(a)
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


def to_np(t):
    """torch tensor -> numpy array the C ABI can read (16-bit floats as uint16)."""
    t = t.contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype == torch.bool:
        return t.numpy().astype(np.uint8)
    return t.numpy().copy()


def from_np(a, dtype):
    if dtype in (torch.bfloat16, torch.float16):
        return torch.from_numpy(a.view(np.int16).copy()).view(dtype)
    return torch.from_numpy(a.copy())
