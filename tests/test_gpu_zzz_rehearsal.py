"""Last in the GPU suite: TWO ranks of bench.py on the one GPU of the test box, over gloo (bench.py's MEDNET_REHEARSE_ONE_GPU
mode).  RCCL refuses two ranks on one device, so this is not the N = 2 run -- it is everything around it: the launcher line the
driver uses (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), both ranks' code paths
through the trainer, the gradient exchange after backward, the barriers, the MAX-over-ranks timing, ONE JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_of_the_bench_on_one_gpu_over_gloo():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MEDNET_REHEARSE_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "torch-mednet_amd"), os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--cpu-steps", "0",
           "--fp32-steps", "0", "--no-roofline", "--patch", "64", "--batch", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0), got {len(lines)}"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    assert rec["scaling"] == "weak" and rec["value"] > 0 and "rehearsal" in rec
    assert "all-reduce" in rec["config"]["gradient_exchange"]
    assert abs(rec["config"]["loss"]) < 10 and rec["config"]["loss"] == rec["config"]["loss"]  # finite
