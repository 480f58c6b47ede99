"""Loader + numpy-level wrappers for the CPU oracle (oracle/cc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  It binds
`liboracle.so` with the same signature table as the device library (cold_compress_amd/_abi.py), symbol
suffix `_cpu`, and takes numpy arrays (host memory).  16-bit floats travel as uint16 bit patterns.
"""
import ctypes as C
import os
import subprocess


from cold_compress_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "liboracle.so")
_FNS = None
_SET_THREADS = None
_ATTN_MATRIX = None


def build(force=False):
    src = os.path.join(_HERE, "cc_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return SO


def fns():
    global _FNS
    if _FNS is None:
        if not os.path.exists(SO):
            build()
        lib = C.CDLL(SO)
        _FNS = _abi.bind(lib, suffix="_cpu")
        lib.cc_oracle_set_threads.restype = C.c_int
        lib.cc_oracle_set_threads.argtypes = [C.c_int]
        global _SET_THREADS
        _SET_THREADS = lib.cc_oracle_set_threads
        lib.cc_oracle_set_threads(1)  # the checker runs single-threaded; bench.py's CPU baseline raises it explicitly
        global _ATTN_MATRIX
        lib.cc_prefill_attn_matrix_cpu.restype = C.c_int
        lib.cc_prefill_attn_matrix_cpu.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_float, C.c_void_p, C.c_void_p]
        _ATTN_MATRIX = lib.cc_prefill_attn_matrix_cpu
    return _FNS


def rng_vector(seed, pos, S):
    """KVCacheRandom's in-kernel generator restated on the host (cc_rng_uniform_cpu): the float32 [S] draw for position `pos`."""
    import numpy as np
    fns()
    lib = C.CDLL(SO)
    lib.cc_rng_uniform_cpu.restype = C.c_float
    lib.cc_rng_uniform_cpu.argtypes = [C.c_uint64, C.c_int32, C.c_int32]
    return np.array([lib.cc_rng_uniform_cpu(int(seed), int(pos), s) for s in range(S)], dtype=np.float32)


def set_threads(n):
    """Thread count of the oracle's OpenMP loops (per-head loops of the decode hot path); returns the previous one."""
    fns()
    return _SET_THREADS(int(n))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def view(k, v, pos, mask, cts, dtype):
    """k,v: [H,S,D] arrays (uint16 for bf16/f16), pos [Hp,S] int32, mask [H,S] uint8/bool, cts [Hc] int32."""
    H, S, D = k.shape
    kv = _abi.KVView(ptr(k), ptr(v), ptr(pos), ptr(mask), ptr(cts), H, pos.shape[0], cts.shape[0], S, D, dtype)
    return kv


def call(name, *args):
    rc = fns()[name](*args)
    if rc != 0:
        raise RuntimeError(f"oracle {name} -> {rc}")


def prefill_attn_matrix(q, k, v, HQ, H, L, D, dtype, scale, y, attn_full):
    """Oracle-only: causal prefill attention with the materialised [H, L, L] group-averaged probabilities (float32)."""
    fns()
    rc = _ATTN_MATRIX(ptr(q), ptr(k), ptr(v), HQ, H, L, D, dtype, scale, ptr(y), ptr(attn_full))
    if rc != 0:
        raise RuntimeError(f"oracle cc_prefill_attn_matrix -> {rc}")
