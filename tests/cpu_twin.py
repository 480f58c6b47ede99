"""TEST-ONLY wiring: the product's Python layer (cold_compress_amd.cache / attention_utils / prompt_compression / harness) on CPU
tensors with every C-ABI call served by the oracle's `_cpu` twin — so that the host logic (budgets, dispatch, buffer bookkeeping,
prompt splitting, the generation loop, per-layer cache construction) runs end to end in the CPU suite, against the reference-made
fixtures.  The product itself has no CPU path: the refusals this module switches off (`_need_device`, the `is_cuda` checks) are
asserted by tests/test_host_logic.py::test_cpu_tensors_refused_no_fallback, and nothing under cold_compress_amd/ knows the oracle.
Entry points without a twin (queries of the device library: single-launch availability, probe, switches) answer "not available", which
puts every cache on its three-call path — the reference's own call sequence (update_kv -> attention -> update_state)."""
import contextlib
import re

import torch

# The fixture-driven `-m gpu` test files that the CPU-twin child run of tests/test_host_e2e_cpu.py executes, and the tests in them that
# are about the DEVICE (hipGraph capture, the single-launch form, raw HIP streams, the device library's own queries): dropped there.
TWIN_FILES = ["test_gpu_parity.py", "test_gpu_e2e.py", "test_gpu_hybrid.py", "test_gpu_quant.py", "test_hh_ring.py", "test_hh_query_fixture.py"]
DEVICE_ONLY_TESTS = [re.compile(x) for x in (
    r"test_library_and_device", r"test_cpu_tensors_are_refused", r"hipgraph", r"test_harness_two_launch_step_equals_three_call_path",
    r"test_requant_known_answers", r"test_batched_round_trip_equals_per_cache", r"test_e2e_cache_bits_8\[True\]",
    r"test_hybrid_two_launch_step_equals_three_launches\[.*-True\]", r"test_fused_step_on_reference_query_trace\[True\]",
    r"test_one_graphed_decoder_across_two_generations",
    r"test_hybrid_fused_step_vs_oracle_pipeline\[8-32-18432")]  # (the C4 size asserts the DEVICE's single-launch form — and is 16 oracle steps at S = 18432)


class _NoStream:
    cuda_stream = None


@contextlib.contextmanager
def cpu_twin(monkeypatch, oracle, single_launch=False):
    """single_launch: the availability queries answer YES, so that the Python layer takes the call sequence of the single-launch /
    recoverable forms (the `_rc` entry points with their commit words, the QKV form's query) — served by the same twins."""
    import host_glue

    import cold_compress_amd.attention_utils as au
    import cold_compress_amd.cache as cache
    import cold_compress_amd.harness.glue as glue
    import cold_compress_amd.harness.model as hm
    import cold_compress_amd.prompt_compression as pc
    from cold_compress_amd import _abi

    twins = dict(oracle.fns())
    real = None
    try:
        import ctypes as C

        real = _abi.bind(C.CDLL(_abi.LIB_PATH)) if _abi.built() else None
    except Exception:  # pragma: no cover
        real = None
    host_only = ("cc_error_string", "cc_decode_step_status_offset", "cc_decode_step_commit_stride")  # pure host functions of the device library
    answers_no = {"cc_decode_step_single_launch", "cc_decode_step_quant_single_launch", "cc_decode_step_hybrid_single_launch",
                  "cc_decode_step_l2_single_launch", "cc_decode_step_single_launch_enabled", "cc_decode_step_qkv_available",
                  "cc_decode_step_l2_handoff", "cc_decode_step_probe_xcd", "cc_decode_step_demote_l2_handoff",
                  # switches and hooks: nothing to switch
                  "cc_decode_step_set_single_launch", "cc_decode_step_set_wide", "cc_decode_step_set_l2_handoff", "cc_decode_step_trace",
                  "cc_debug_qkv_trace"}
    CC_ERR_UNSUPPORTED = -2

    def l2_rc(*a):  # = cc_decode_step_l2 with the commit words (argument 7) dropped: the twin has no hand-off to recover from
        return twins["cc_decode_step_l2"](*a[:7], *a[8:])

    def quant_rc(c, qparams, n_bit, policy, q, k, v, pos, num, denom, counter, rand_next, seed, next_key, commit, g, w, HQ, scale, y, ws,
                 ws_bytes, stream, phases):
        if policy == 3 and not rand_next:
            return CC_ERR_UNSUPPORTED  # (in-kernel draws: no plain-form twin)
        return twins["cc_decode_step_quant"](c, qparams, n_bit, policy, q, k, v, pos, num, denom, counter, rand_next, next_key, g, w, HQ,
                                             scale, y, None, ws, ws_bytes, stream, phases)

    for name in _abi.SIGNATURES:
        if name in twins:
            continue
        if real is not None and name in host_only:
            twins[name] = real[name]
        elif name in answers_no:
            yes = single_launch and name in ("cc_decode_step_single_launch", "cc_decode_step_quant_single_launch", "cc_decode_step_hybrid_single_launch",
                                             "cc_decode_step_l2_single_launch", "cc_decode_step_single_launch_enabled")
            twins[name] = (lambda *a, **k: 1) if yes else (lambda *a, **k: 0)
        else:  # compute entry points without a twin fail LOUDLY (check() raises), never silently do nothing
            twins[name] = (lambda *a, **k: CC_ERR_UNSUPPORTED)
    twins["cc_decode_step_l2_rc"], twins["cc_decode_step_quant_rc"] = l2_rc, quant_rc
    monkeypatch.setattr(_abi, "_FNS", twins)
    for mod in (cache, pc, au):
        monkeypatch.setattr(mod, "_need_device", lambda t, what: None)
    for mod in (cache, au, pc, glue):
        if hasattr(mod, "_stream"):
            monkeypatch.setattr(mod, "_stream", lambda: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    saved = (glue.add_rmsnorm, glue.qkv_rope, glue.silu_mul)
    host_glue.install(hm.glue)
    try:
        yield twins
    finally:
        glue.add_rmsnorm, glue.qkv_rope, glue.silu_mul = saved
