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


def test_one_rank_rccl_step_is_bit_identical_to_the_local_step(tmp_path):
    env = dict(os.environ, NCCL_DEBUG="INFO", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["NCCL_DEBUG_FILE"] = str(tmp_path / "rccl_one_rank_debug.log")  # (tools/rccl_one_rank.py keeps a copy for profiles/)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL1 ")]
    assert p.returncode == 0 and lines, (p.stdout[-1500:], p.stderr[-3000:])
    res = json.loads(lines[-1][6:])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["losses_equal"] and res["params_equal"], res
    assert res["hook_fired"] and res["async_work_launched_in_backward"], res


def _child(case, tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    env.pop("MEDNET_BUCKETS", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py"), case], env=env, capture_output=True,
                       text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL1 ")]
    assert p.returncode == 0 and lines, (p.stdout[-1500:], p.stderr[-3000:])
    res = json.loads(lines[-1][6:])
    assert res["backend"] == "nccl" and res["world"] == 1
    return res


def test_fp16_loss_scaler_skips_and_recovers_under_the_rccl_exchange(tmp_path):
    """fp16 storage on N GPUs (BASELINE config 5): scaled gradients are all-reduced, the overflow check reads the REDUCED buffer,
    so all ranks skip or step together (train_seg.py:126 -> Trainer(gpus=N)).  One rank, real RCCL: a forced overflow leaves
    parameters and moments untouched and halves the scale; the steps after it equal the no-exchange run bit for bit."""
    res = _child("fp16_overflow", tmp_path)
    for arm in ("overflow_local", "overflow_allreduce"):
        o = res[arm]
        assert o["grads_nonfinite"] and o["params_untouched"] and o["moments_zero"] and o["scale_halved"], (arm, o)
        assert o["steps_taken"] == 0 and o["skipped_steps"] == 1, (arm, o)
    assert res["losses_equal"] and res["params_equal"] and res["scaler_equal"], res
    assert res["steps_taken_after"] == 3, res
    assert "all-reduce" in res["exchange"], res


def test_two_bucket_exchange_at_config5_buffer_size(tmp_path):
    """The two-bucket exchange overlapped with backward (opt-in, MEDNET_BUCKETS=1) at config 5's 565 MB gradient buffer, one
    rank over RCCL: bit-identical to the single all-reduce and to the no-exchange step; the early bucket (everything outside the
    two full-resolution encoders) is launched from inside backward."""
    res = _child("cfg5_buckets", tmp_path)
    assert res["losses_equal"] and res["params_equal"], res
    assert res["grad_buffer_MB"] > 500 and res["early_bucket_MB"] > 0.9 * res["grad_buffer_MB"], res
    assert res["async_work_launched_in_backward"], res
