#!/bin/bash
# policies table (compact): tools/dbg/r4_pol.sh [CC_POLICIES_ONLY value]
export TMPDIR=/tmp
for on in 1 0; do echo "== l2_handoff $on"; CC_L2H=$on python - <<'PY'
import os, sys, json, subprocess
sys.path.insert(0, os.getcwd())
from cold_compress_amd import _abi
_abi.lib()["cc_decode_step_set_l2_handoff"](int(os.environ["CC_L2H"]))
sys.argv = ["bench_policies.py"]
sys.path.insert(0, "tools")
import io, contextlib
import bench_policies
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench_policies.main()
for l in buf.getvalue().splitlines():
    try: d = json.loads(l)
    except Exception: continue
    print({k: d[k] for k in d if k in ("config", "cfg", "policy", "strategy", "fused_step_us", "fused_quant8_step_us", "S", "H")})
PY
done
