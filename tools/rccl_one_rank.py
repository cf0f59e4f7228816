"""One-rank RCCL rehearsal of the data-parallel step (train_seg.py:126 -> Trainer(gpus=N)) in a FRESH process:
the process group is created before any other GPU call, then train.SegmentationStep runs (a) without a process group's
collective, (b) with the single flat all-reduce forced (world 1), (c) with the two-bucket exchange overlapped with backward
(MEDNET_BUCKETS=1).  All three must leave bit-identical losses and parameters after the same steps (an all-reduce over one
rank is the identity, and the bucket slices are disjoint).  Prints one JSON line.  Run by tests/test_gpu_rccl.py; with
NCCL_DEBUG=INFO the channel / algorithm lines go to NCCL_DEBUG_FILE (kept under profiles/ for the 8-GPU run)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torch-mednet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm; before any other GPU call

import mednet_hip  # noqa: E402
from mednet_hip import train as T  # noqa: E402
from mednet_hip.synth import keyed_init_, synthetic_batch  # noqa: E402
from mednet_hip.unet.model import ResidualUNet3D  # noqa: E402

CASE = sys.argv[1] if len(sys.argv) > 1 else "bf16_three_forms"


def fp16_overflow_case():
    """fp16 storage + device-side loss scaler under the RCCL exchange (BASELINE config 5's mode on N GPUs): the all-reduce sums the
    SCALED gradients, the overflow check runs on the reduced buffer, so every rank takes or skips the step together.  A forced
    overflow (scale 2^60: the head's data gradient leaves fp16's range) must leave parameters and Adam moments untouched, halve
    the scale and count one skipped step; after the scale is put back the steps must equal the no-exchange run bit for bit."""
    mednet_hip.set_precision("fp16")
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128])
    batches = [{k: v.to(dev) for k, v in synthetic_batch(2, 1, (32, 32, 32), 4, 0, seed=60 + i).items()} for i in range(3)]
    out = {}
    for mode in ("local", "allreduce"):
        net = keyed_init_(ResidualUNet3D(**ctor)).to(dev)
        step = T.SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3, world_size=1)
        step.force_allreduce = mode == "allreduce"
        p0 = step.flat.flat.clone()
        step.scaler.state[0] = float(2 ** 60)
        step(batches[0])
        torch.cuda.synchronize()
        snap = step.scaler.snapshot()
        skipped = {"params_untouched": bool(torch.equal(step.flat.flat, p0)), "moments_zero": bool((step.opt.m == 0).all() and (step.opt.v == 0).all()),
                   "scale_halved": snap[0] == float(2 ** 59), "steps_taken": snap[2], "skipped_steps": step.scaler.skipped_steps(),
                   "grads_nonfinite": not bool(torch.isfinite(step.flat.grad).all())}
        step.scaler.state[0] = 65536.0
        losses = [float(step(b)) for b in batches]
        torch.cuda.synchronize()
        out[mode] = (skipped, losses, step.flat.flat.clone(), step.scaler.snapshot(), step._exchange.describe())
        step.flat.release()
    res = {"case": "fp16_overflow", "backend": dist.get_backend(), "world": dist.get_world_size(),
           "overflow_local": out["local"][0], "overflow_allreduce": out["allreduce"][0],
           "losses_equal": out["local"][1] == out["allreduce"][1], "params_equal": bool(torch.equal(out["local"][2], out["allreduce"][2])),
           "scaler_equal": out["local"][3] == out["allreduce"][3], "steps_taken_after": out["allreduce"][3][2],
           "exchange": out["allreduce"][4]}
    print("RCCL1 " + json.dumps(res), flush=True)


def cfg5_buckets_case():
    """The exchange forms at config 5's gradient-buffer size (141 M parameters = 565 MB fp32, the size the two-bucket overlapped
    form is meant for): local / one all-reduce / two buckets overlapped with backward (MEDNET_BUCKETS=1), one rank, at a small
    spatial size.  All three bit-identical; the early bucket must have been launched from inside backward."""
    mednet_hip.set_precision("bf16")
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[64, 128, 256, 512, 1024])
    batches = [{k: v.to(dev) for k, v in synthetic_batch(2, 1, (32, 32, 16), 4, 0, seed=70 + i).items()} for i in range(2)]
    out, info = {}, {}
    for mode in ("local", "allreduce", "buckets"):
        os.environ["MEDNET_BUCKETS"] = "1" if mode == "buckets" else "0"
        net = keyed_init_(ResidualUNet3D(**ctor)).to(dev)
        step = T.SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3, world_size=1)
        step.force_allreduce = mode != "local"
        losses = [float(step(b)) for b in batches]
        if mode == "buckets":
            ex = step._exchange
            assert ex.enabled, "bucketed exchange not enabled"
            seen = []
            orig_fin = ex.finish
            ex.finish = lambda: (seen.append(ex.work is not None), orig_fin())[1]
            losses.append(float(step(batches[0])))
            info = {"grad_buffer_MB": round(step.flat.total * 4 / 1e6, 1), "early_bucket_MB": round((step.flat.total - ex.split) * 4 / 1e6, 1),
                    "async_work_launched_in_backward": bool(seen and seen[0]), "exchange": ex.describe()}
        else:
            losses.append(float(step(batches[0])))
        torch.cuda.synchronize()
        out[mode] = (losses, step.flat.flat.clone())
        step.flat.release()
        del net, step
        torch.cuda.empty_cache()
    res = {"case": "cfg5_buckets", "backend": dist.get_backend(), "world": dist.get_world_size(),
           "losses_equal": out["local"][0] == out["allreduce"][0] == out["buckets"][0],
           "params_equal": bool(torch.equal(out["local"][1], out["allreduce"][1]) and torch.equal(out["local"][1], out["buckets"][1]))}
    res.update(info)
    print("RCCL1 " + json.dumps(res), flush=True)


if CASE == "fp16_overflow":
    fp16_overflow_case()
    dist.destroy_process_group()
    sys.exit(0)
if CASE == "cfg5_buckets":
    cfg5_buckets_case()
    dist.destroy_process_group()
    sys.exit(0)

mednet_hip.set_precision("bf16")
ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128])
batches = [{k: v.to(dev) for k, v in synthetic_batch(2, 1, (32, 32, 32), 4, 0, seed=40 + i).items()} for i in range(3)]
out = {}
fired = {}
for mode in ("local", "allreduce", "buckets"):
    os.environ["MEDNET_BUCKETS"] = "1" if mode == "buckets" else "0"
    net = keyed_init_(ResidualUNet3D(**ctor)).to(dev)
    step = T.SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3, world_size=1)
    step.force_allreduce = mode != "local"
    losses = []
    for b in batches:
        losses.append(float(step(b)))
        if mode == "buckets":
            assert step._exchange.enabled, "bucketed exchange not enabled"
    if mode == "buckets":
        # the early bucket must have been launched from inside backward on the product model's forward path
        seen = []
        orig = step._exchange._on_grad
        step._exchange._on_grad = lambda g: (seen.append(1), orig(g))[1]
        step._exchange._work_seen = False
        orig_fin = step._exchange.finish

        def fin():
            step._exchange._work_seen = step._exchange.work is not None
            return orig_fin()
        step._exchange.finish = fin
        losses.append(float(step(batches[0])))
        fired = {"hook_fired": bool(seen), "async_work_launched_in_backward": bool(step._exchange._work_seen)}
    else:
        losses.append(float(step(batches[0])))
    torch.cuda.synchronize()
    out[mode] = (losses, step.flat.flat.clone())
    step.flat.release()
res = {"losses_equal": out["local"][0] == out["allreduce"][0] == out["buckets"][0],
       "params_equal": bool(torch.equal(out["local"][1], out["allreduce"][1]) and torch.equal(out["local"][1], out["buckets"][1])),
       "losses": out["local"][0], "backend": dist.get_backend(), "world": dist.get_world_size()}
res.update(fired)
print("RCCL1 " + json.dumps(res), flush=True)
dist.destroy_process_group()
