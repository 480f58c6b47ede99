"""ctypes binding of the C ABI declared in include/coldcompress.h (+ the hooks of include/coldcompress_debug.h).

The product path has NO CPU fallback: `lib()` raises if `libcoldcompress_hip.so` has not been built
(`python -c "import __graft_entry__ as g; g.build()"`), and every wrapper raises `ColdCompressError` on a
non-zero return code.  The same signature table binds the CPU oracle (`oracle/oracle_lib.py`, `_cpu`
suffix) for tests — the oracle is never imported from here.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must load ITS bundled HIP runtime before our library resolves libamdhip64)

CC_OK = 0
CC_DT_F32, CC_DT_BF16, CC_DT_F16 = 0, 1, 2
CC_PRIO_F32, CC_PRIO_BF16, CC_PRIO_F16, CC_PRIO_I64 = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcoldcompress_hip.so")


class ColdCompressError(RuntimeError):
    pass


class KVView(C.Structure):
    """struct cc_kv_view (include/coldcompress.h)."""

    _fields_ = [
        ("k_cache", C.c_void_p),
        ("v_cache", C.c_void_p),
        ("pos", C.c_void_p),
        ("mask", C.c_void_p),
        ("cache_cts", C.c_void_p),
        ("H", C.c_int32),
        ("Hp", C.c_int32),
        ("Hc", C.c_int32),
        ("S", C.c_int32),
        ("D", C.c_int32),
        ("dtype", C.c_int32),
    ]


_vp, _i32, _f32, _sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
_view = C.POINTER(KVView)

# name -> (restype, argtypes); one row per declaration in include/coldcompress.h
SIGNATURES = {
    "cc_abi_version": (C.c_int, []),
    "cc_error_string": (C.c_char_p, [C.c_int]),
    "cc_device_info": (C.c_int, [_vp, _vp, _vp, C.c_char_p, C.c_int]),
    "cc_decode_update_full": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp]),
    "cc_decode_update_recent_global": (C.c_int, [_view, _vp, _vp, _vp, _i32, _vp, _vp]),
    "cc_decode_update_scores": (C.c_int, [_view, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cc_decode_update_random": (C.c_int, [_view, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cc_decode_update_l2_workspace_bytes": (_sz, [_i32, _i32]),
    "cc_decode_update_l2": (C.c_int, [_view, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "cc_decode_update_heavy_hitter": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cc_hh_update": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cc_decode_attn_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "cc_decode_attn_gqa": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _sz, _vp]),
    "cc_decode_attn_gqa_ring": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i32,
                                          _vp, _vp, _vp, _sz, _vp]),
    "cc_softmax_argmax_workspace_bytes": (_sz, []),
    "cc_softmax_argmax": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "cc_gemv_fused": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "cc_kv_requant": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cc_kv_requant_pair": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "cc_kv_requant_batch": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cc_kv_dequant": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cc_kv_quant_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "cc_kv_dequant_rows": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "cc_decode_step_quant": (C.c_int, [_view, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32,
                                       _vp, _vp, _vp, _sz, _vp, _i32]),
    "cc_decode_step_quant_rc": (C.c_int, [_view, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _i32, _i32, _i32,
                                          _f32, _vp, _vp, _sz, _vp, _i32]),
    "cc_decode_step_quant_single_launch": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "cc_decode_step_hybrid_single_launch": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "cc_decode_step_l2_single_launch": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "cc_decode_step_single_launch_enabled": (_i32, []),
    "cc_decode_step_device_single_launch": (_i32, [_i32]),
    "cc_hh_next_key_slots": (_i32, [_i32]),
    "cc_decode_step_single_launch": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "cc_decode_step_status_offset": (_i32, []),
    "cc_decode_step_wait_bound_us": (_i32, []),
    "cc_decode_step_trace": (None, [_vp]),
    "cc_decode_step_set_single_launch": (None, [_i32]),
    "cc_decode_step_set_wide": (None, [_i32]),
    "cc_decode_step_l2_carry": (_i32, []),
    "cc_decode_step_probe_xcd": (_i32, []),
    "cc_decode_step_set_l2_handoff": (None, [_i32]),
    "cc_decode_step_l2_handoff": (_i32, []),
    "cc_decode_step_demote_l2_handoff": (_i32, [_i32]),
    "cc_decode_step_heavy_hitter_rc": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz,
                                                 _vp, _i32]),
    "cc_decode_step_qkv_available": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "cc_decode_step_qkv_rc": (C.c_int, [_view, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                        C.c_uint64, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "cc_debug_occupy": (C.c_int, [_i32, _i32, _i32, _vp, _vp]),
    "cc_debug_qkv_trace": (None, [_vp]),
    "cc_decode_step_stream_floor": (C.c_int, [_view, _i32, _vp, _vp]),
    "cc_decode_step_stream_floor_geom": (C.c_int, [_view, _i32, _i32, _vp, _vp]),
    "cc_rg_next_key_init": (C.c_int, [_view, _vp, _i32, _vp, _vp]),
    "cc_decode_step_recent_global": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "cc_hh_ring_next_key_init": (C.c_int, [_view, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "cc_decode_step_heavy_hitter_ring": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32,
                                                   _f32, _vp, _vp, _vp, _sz, _vp]),
    "cc_hybrid_next_key_init": (C.c_int, [_view, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp]),
    "cc_decode_step_hybrid": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _i32, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "cc_decode_step_hybrid_rc": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _i32, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "cc_l2_next_key_init": (C.c_int, [_view, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cc_decode_step_l2": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "cc_decode_step_l2_rc": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "cc_decode_step_commit_stride": (_i32, []),
    "cc_random_next_key_init": (C.c_int, [_view, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cc_decode_step_random": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "cc_random_next_key_init_rng": (C.c_int, [_view, _vp, C.c_uint64, _i32, _i32, _vp, _vp]),
    "cc_decode_step_head_constant_rc": (C.c_int, [_view, _i32, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _i32, _i32, _i32, _f32,
                                                  _vp, _vp, _sz, _vp]),
    "cc_decode_step_random_rng": (C.c_int, [_view, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "cc_hh_next_key_init": (C.c_int, [_view, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cc_decode_step_heavy_hitter": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp,
                                              _vp, _sz, _vp]),
    "cc_decode_step_heavy_hitter_phases": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp,
                                                     _vp, _vp, _sz, _vp, _i32]),
    "cc_decode_attn_gqa_phases": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _vp, _sz, _vp, _i32]),
    "cc_prefill_fill": (C.c_int, [_view, _vp, _vp, _vp, _i32, _i32, _vp]),
    "cc_row_l2_norm": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cc_topk_keep_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "cc_topk_keep": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "cc_gather_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cc_gather_vec": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cc_analysis_loss": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "cc_snapkv_priority": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cc_prefill_attn_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "cc_prefill_attn": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _vp,
                                  _sz, _vp]),
    "cc_prefill_attn_bands": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _vp, _i32,
                                        _vp, _vp, _sz, _vp]),
    "cc_attn_colsum": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cc_colsum_to_mean": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "cc_hybrid_decode_update": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32,
                                          _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "cc_decode_update_heavy_hitter_ring": (C.c_int, [_view, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "cc_hh_ring_update": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "cc_hh_ring_acc_words": (_sz, [_i32, _i32, _i32, _i32]),
    "cc_hh_ring_window_sums": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "cc_attn_bandsum": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cc_add_rmsnorm": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp, _vp]),
    "cc_qkv_rope": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "cc_silu_mul": (C.c_int, [_vp, _vp, C.c_int64, _i32, _vp, _vp]),
    "cc_allreduce_handle_bytes": (_sz, []),
    "cc_allreduce_create": (C.c_int, [_i32, _i32, _sz, C.POINTER(_vp)]),
    "cc_allreduce_export": (C.c_int, [_vp, _vp]),
    "cc_allreduce_connect": (C.c_int, [_vp, _vp]),
    "cc_allreduce_sum": (C.c_int, [_vp, _vp, C.c_int64, _i32, _vp]),
    "cc_allreduce_status": (_i32, [_vp]),
    "cc_allreduce_destroy": (C.c_int, [_vp]),
}

# entry points that only the device library has (no `_cpu` twin)
DEVICE_ONLY = {"cc_error_string", "cc_device_info", "cc_decode_step_single_launch", "cc_decode_step_status_offset", "cc_decode_step_wait_bound_us",
               "cc_decode_step_trace", "cc_decode_step_set_single_launch", "cc_decode_step_set_wide", "cc_decode_step_l2_carry", "cc_decode_step_probe_xcd", "cc_decode_step_commit_stride", "cc_decode_step_l2_rc", "cc_decode_step_quant_rc", "cc_decode_step_set_l2_handoff", "cc_decode_step_l2_handoff", "cc_decode_step_demote_l2_handoff", "cc_decode_step_stream_floor", "cc_decode_step_stream_floor_geom", "cc_debug_occupy", "cc_decode_step_quant_single_launch",
               "cc_decode_step_hybrid_single_launch", "cc_decode_step_l2_single_launch", "cc_decode_step_single_launch_enabled", "cc_decode_step_device_single_launch",
               "cc_decode_step_qkv_available", "cc_debug_qkv_trace",
               "cc_kv_requant_batch",  # (its oracle is the per-cache twin of cc_kv_requant_pair)
               # inter-GPU transport: no CPU twin (the oracle of the all-reduce is torch.distributed's)
               "cc_allreduce_handle_bytes", "cc_allreduce_create", "cc_allreduce_export", "cc_allreduce_connect", "cc_allreduce_sum",
               "cc_allreduce_status", "cc_allreduce_destroy"}
CC_PHASE_TWO_LAUNCH, CC_PHASE_ONE_LAUNCH = 0x10000, 0x20000


def bind(cdll, suffix=""):
    """Attach restype/argtypes for every ABI symbol; raises AttributeError if one is missing."""
    fns = {}
    for name, (res, args) in SIGNATURES.items():
        if suffix and name in DEVICE_ONLY:
            continue
        fn = getattr(cdll, name + suffix)
        fn.restype = res
        fn.argtypes = args
        fns[name] = fn
    return fns


_LIB = None
_FNS = None


def built():
    """True when the HIP extension exists (host-side constructors size their pipeline buffers through it)."""
    return os.path.exists(LIB_PATH)


def lib():
    """The device library's bound functions; raises loudly when it has not been built."""
    global _LIB, _FNS
    if _FNS is None:
        if not os.path.exists(LIB_PATH):
            raise ColdCompressError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        _LIB = C.CDLL(LIB_PATH)
        _FNS = bind(_LIB)
        v = _FNS["cc_abi_version"]()
        if v != 1:
            raise ColdCompressError(f"ABI version mismatch: library {v}, python 1")
        # (no device probe here — ADVICE r4: loading the library must not allocate or launch on whatever device happens to be current,
        #  possibly before a TP rank has selected its own; the probe runs when a decode workspace is created: attention_utils._workspace)
    return _FNS


def probe_device():
    """Observe the current device's block -> XCD dispatch order once (cc_decode_step_probe_xcd: synchronous, outside capture): where
    it is verified, caches with a multiple of 8 kv heads run their single-tile step with the L2-resident hand-off
    (include/coldcompress.h).  No GPU: nothing happens.  Called when a decode workspace is created on a device (attention_utils._workspace)."""
    if _FNS is None:
        return 0
    try:
        import torch

        if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return 0
    except Exception:  # pragma: no cover
        return 0
    return int(_FNS["cc_decode_step_probe_xcd"]())


def check(code, what):
    if code != CC_OK:
        msg = lib()["cc_error_string"](code)
        raise ColdCompressError(f"{what} failed: {code} ({msg.decode() if msg else '?'})")


def call(name, *args):
    check(lib()[name](*args), name)
