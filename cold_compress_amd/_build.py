"""Build recipe for the HIP extension: hipcc cross-compiles gfx950 without a GPU present.

    python -m cold_compress_amd._build            # or __graft_entry__.build()

The shared library is built IN-TREE (cold_compress_amd/csrc/libcoldcompress_hip.so) so that it travels
to the GPU box with the repo snapshot.  `-ffp-contract=off` keeps every multiply/add exactly as written
(fused multiply-adds are spelled fmaf() where wanted) so that device arithmetic matches the oracle's.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libcoldcompress_hip.so")
SOURCES = ["cc_api.hip", "cc_evict.hip", "cc_attn_decode.hip", "cc_attn_decode_qkv.hip", "cc_compact.hip", "cc_attn_prefill.hip", "cc_glue.hip",
           "cc_hybrid.hip", "cc_attn_prefill_mfma.hip", "cc_quant.hip", "cc_gemv.hip", "cc_allreduce.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# Per-file extras.  cc_attn_decode: keep the matrix-core accumulators in architectural VGPRs (gfx950 has one unified register
# file) — the streaming pass rescales its 32 accumulators with VALU multiplies whenever the running maximum moves, and with the
# accumulators parked in AGPRs that costs 68 v_accvgpr_read / _write per tile, a sixth of the loop's instructions.
# -amdgpu-kernarg-preload-count=14 (r6): the step kernels' leading scalar arguments — what their first requests need — arrive in SGPRs
# with the wave instead of through a round of kernel-argument loads (cc_attn_decode_kernels.h, CC_V_PRELOAD).
EXTRA_FLAGS = {"cc_attn_decode.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-kernarg-preload-count=14"],
               "cc_attn_decode_qkv.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-kernarg-preload-count=14"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    inc = os.path.join(HERE, "..", "include")
    deps = [src, os.path.join(CSRC, "cc_common.h"), os.path.join(CSRC, "cc_wacc.h"), os.path.join(CSRC, "cc_gemv_core.h"),
            os.path.join(inc, "coldcompress.h"), os.path.join(inc, "coldcompress_debug.h"), __file__]
    if "cc_attn_decode" in os.path.basename(src):  # the two translation units that instantiate the step kernels
        deps += [os.path.join(CSRC, "cc_attn_decode_kernels.h"), os.path.join(CSRC, "cc_attn_decode_qkv.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, src):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):  # (an object compiled by hand)
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
