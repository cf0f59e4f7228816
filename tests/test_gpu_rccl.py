"""Multi-GPU readiness on a one-GPU box: the RCCL path of the data-parallel step with ONE rank, in a fresh process (the
process group must exist before the first GPU call).  The 8-GPU run itself belongs to the driver; the N>1 arithmetic is
covered by the world-2 gloo tests (tests/test_ddp_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_step_is_bit_identical_to_the_local_step():
    env = dict(os.environ, NCCL_DEBUG="INFO", HSA_ENABLE_IPC_MODE_LEGACY="0")
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        env["NCCL_DEBUG_FILE"] = os.path.join(log_dir, "rccl_one_rank_debug.log")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL1 ")]
    assert p.returncode == 0 and lines, (p.stdout[-1500:], p.stderr[-3000:])
    res = json.loads(lines[-1][6:])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["losses_equal"] and res["params_equal"], res
    assert res["hook_fired"] and res["async_work_launched_in_backward"], res
