"""debug: run a pytest selection with the L2-resident hand-off on / off:  python tools/dbg/r4_dbg_run.py 0|1 <pytest args...>"""
import os
import sys

sys.path.insert(0, os.getcwd())
import pytest  # noqa: E402

from cold_compress_amd import _abi  # noqa: E402

_abi.lib()["cc_decode_step_set_l2_handoff"](int(sys.argv[1]))
print("l2_handoff:", _abi.lib()["cc_decode_step_l2_handoff"](), flush=True)
sys.exit(pytest.main(sys.argv[2:]))
