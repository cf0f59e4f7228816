#!/usr/bin/env python3
"""Headline benchmark: 128^3 patches/sec, training, 3D U-Net 32-base-ch (BASELINE.json), on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = SegmentationNet.training_step on one batch (segmentation.py:58-65): fwd + DiceLoss + bwd, one RCCL
all-reduce of the flat gradient buffer, Adam -- BASELINE config 2 (ResidualUNet3D f_maps=[32,64,128,256], 4 classes,
128^3 patches, batch 4 per GPU, bf16 storage / fp32 accumulate), weak scaling.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "torch-mednet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp16x2": 2500.0, "fp32": 157.3}  # dense peaks, MI355X_MICROARCH.md
F_MAPS = [32, 64, 128, 256]
FLOP_PER_PATCH = 3447.9e9  # 6 x forward conv/convT MACs of ResidualUNet3D cfg2 at 128^3


def host_cores() -> int:
    """Cores this process may actually use: affinity mask and cgroup CPU quota; a box that shows every core of a shared
    host but grants a share (the 1-GPU boxes grant 16) would otherwise be oversubscribed 16x by the oracle."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p))
    except Exception:
        pass
    cap = int(os.environ.get("MEDNET_CPU_BASELINE_CORES", "16"))  # CPU share of a 1-GPU box
    return min(n, cap)


def cpu_baseline(steps: int):
    """The oracle (plain torch.nn restatement of the reference) timed on this box's host cores: cfg2's model, N=1,
    128^3, fp32, fwd + DiceLoss + bwd + Adam -- a bounded sample (1 warm-up + `steps` timed steps)."""
    from oracle import ref_cpu as O
    cores = host_cores()
    torch.set_num_threads(cores)
    model = O.keyed_init_(O.ResidualUNet3D(1, 4, False, f_maps=F_MAPS))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = O.DiceLoss(weight=torch.tensor([0.05, 1.0, 1.0, 1.0]))
    batch = O.synthetic_batch(1, 1, (128, 128, 128), 4, 0, seed=1234)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = O.seg_training_step(model, crit, batch)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    return {"value": round(1.0 / med, 5), "unit": "patches/s", "cores": cores, "cpu_model": cpu_model(),
            "cores_visible": os.cpu_count(), "kind": "port",
            "sample": f"oracle ResidualUNet3D {F_MAPS} 4-class, one 128^3 patch (N=1), fp32, fwd+Dice+bwd+Adam, "
                      f"1 warm-up + {steps} timed steps (median {med:.2f} s/step)"}


def cfg1_plumbing(dev, steps: int = 20):
    """BASELINE config 1 -- the call sequence of examples/train_seg.py: ResidualUNet3D f_maps=[8] (1 level), 2 classes, 32^3
    patches, batch 2, fp32, SegmentationNet.training_step + Adam.  The reference runs it on PyTorch CPU: the oracle is timed
    on the host cores, the same step through the HIP path (fp32 storage mode) beside it.  It is plumbing, not a benchmark
    (1.5 GFLOP per batch: launch-bound on the GPU)."""
    import mednet_hip
    from oracle import ref_cpu as O
    from mednet_hip.train import SegmentationStep
    from mednet_hip.unet.model import ResidualUNet3D
    from mednet_hip.synth import keyed_init_, synthetic_batch
    w = [0.05, 1.0]
    cores = host_cores()
    torch.set_num_threads(cores)
    ora = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8]))
    opt = torch.optim.Adam(ora.parameters(), lr=1e-3)
    crit = O.DiceLoss(weight=torch.tensor(w))
    batch = O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234)
    times, first_cpu = [], None
    for _ in range(steps + 3):
        t0 = time.perf_counter()
        opt.zero_grad()
        lo = O.seg_training_step(ora, crit, batch)
        lo.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if first_cpu is None:
            first_cpu = float(lo.detach())
    cpu_s = sorted(times[3:])[len(times[3:]) // 2]
    with mednet_hip.precision("fp32"):
        model = keyed_init_(ResidualUNet3D(1, 2, False, f_maps=[8])).to(dev)
        step = SegmentationStep(model, loss_weight=w, lr=1e-3)
        b = {k: v.to(dev) for k, v in synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234).items()}
        first_hip = float(step(b))
        for _ in range(2):
            step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(b)
        torch.cuda.synchronize()
        hip_s = (time.perf_counter() - t0) / steps
        step.flat.release()
    del model, step
    return {"workload": "BASELINE config 1: ResidualUNet3D f_maps=[8] 2-class, 32^3 patches, batch 2, fp32, "
                        "training_step + Adam (examples/train_seg.py call sequence)",
            "cpu_oracle_patches_per_s": round(2 / cpu_s, 2), "cpu_cores": cores, "cpu_model": cpu_model(), "cpu_kind": "port",
            "hip_fp32_patches_per_s": round(2 / hip_s, 1), "first_step_loss_cpu": round(first_cpu, 6),
            "first_step_loss_hip": round(first_hip, 6), "steps": steps, "unit": "32^3 patches/s"}


DOMINANT_KEY = "mednet::conv32_mfma_kernel<4> conv3d 32->32@128^3 (forward launches with fused GroupNorm statistics, grid=65536)"
KERNEL_SOURCE = os.path.join(ROOT, "torch-mednet_amd", "csrc", "conv_mfma.hip")


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def source_sha256(path: str = KERNEL_SOURCE) -> str:
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def pmc_traffic(batch: int, patch: int):
    """-> (HBM bytes per launch of the dominant kernel or None, where that number comes from).

    PMC counters cannot be collected from inside the timed run, so `traffic` is the value of the newest committed rocprofv3
    PMC profile (profiles/rNN_pmc_hbm_traffic.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command,
    FETCH_SIZE doubled as the gfx950 guide prescribes) -- and ONLY if that profile was taken from the kernel source that is
    being timed now: the profile records the sha256 of csrc/conv_mfma.hip; on a mismatch the number is stale and null is
    reported instead."""
    import glob
    if batch != 4 or patch != 128:
        return None, "no PMC profile for this launch shape"
    now = source_sha256()
    stale = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            meta = d.get("_meta", {})
            name = os.path.relpath(path, ROOT)
            if meta.get("kernel_source_sha256") != now:
                stale.append(name)
                continue
            return d[DOMINANT_KEY]["hbm_bytes_per_launch"], (
                f"{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py`, git {meta.get('git_head', '?')}, "
                f"conv_mfma.hip sha256 {now[:12]})")
        except Exception:
            continue
    return None, ("stale: csrc/conv_mfma.hip (sha256 " + now[:12] + ") is newer than " +
                  (", ".join(stale) if stale else "every profile under profiles/") + "; re-run tools/gpu_profile.sh")


def wgrad_alone_ms(dev, batch: int, patch: int, precision: str) -> float:
    """Weight gradient of a 32 -> 32 layer at full resolution alone on the chip (through the C ABI, HIP events on the launch
    stream), on operands with the statistics the step's own tensors have (post-ELU activations, small zero-mean gradients: the
    clock the chip holds under matrix load depends on the operands)."""
    from mednet_hip import _lib as L
    lib = L.lib()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp16x2": torch.float16}.get(precision)
    if dt is None:
        return float("nan"), 0
    code = L.BF16 if dt == torch.bfloat16 else L.F16
    c = F_MAPS[0]
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.nn.functional.elu(torch.randn(batch, c, patch, patch, patch, device=dev, generator=g)).to(dt).contiguous(memory_format=torch.channels_last_3d)
    dy = (torch.randn(batch, c, patch, patch, patch, device=dev, generator=g) * 1e-3).to(dt).contiguous(memory_format=torch.channels_last_3d)
    dw = torch.empty(c, c, 3, 3, 3, device=dev)
    ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(batch, patch, patch, patch, c, c, 3, 0), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, batch, patch, patch, patch, c, c, 3, code,
                                                  L.NDHWC, code, L.NDHWC, L.ALGO_AUTO, 0, ws.data_ptr(), ws.numel(), st), "conv3d_wgrad")
    for _ in range(20):  # (the chip idled while the host assembled the record: bring the clocks back before timing)
        run()
    torch.cuda.synchronize()
    rounds, per_round = [], 20
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per_round):
            run()
        e1.record()
        torch.cuda.synchronize()
        rounds.append(e0.elapsed_time(e1) / per_round)
    # median of three rounds of 20 launches (each launch = the kernel + its fixed-order reduce), and how many launches were timed
    return sorted(rounds)[1], len(rounds) * per_round


def fp32_parity_mode(dev, batch: int, patch: int, steps: int):
    """The SAME step in the fp32 storage mode -- the mode that meets the north star's 1e-3 on logits and gradients
    (tests: test_cfg2_128_against_reference_golden) -- timed the same way, as a sub-record."""
    import mednet_hip
    from mednet_hip.train import SegmentationStep
    from mednet_hip.unet.model import ResidualUNet3D
    from mednet_hip.synth import keyed_init_, synthetic_batch
    with mednet_hip.precision("fp32"):
        model = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=F_MAPS)).to(dev)
        step = SegmentationStep(model, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        b = {k: v.to(dev) for k, v in synthetic_batch(batch, 1, (patch, patch, patch), 4, 0, seed=1234).items()}
        for _ in range(2):  # (untimed: the first step sizes the allocator's blocks, the second reuses them)
            step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        step.flat.release()
    pps = batch * steps / dt
    rec = {"value": round(pps, 3), "unit": "patches/s", "ms_per_step": round(1e3 * dt / steps, 2), "steps": steps, "warmup": 2,
           "dtype": "fp32", "loss": round(float(loss), 6),
           "arithmetic": "fp32 storage of activations, gradients and parameters; 3x3x3 contractions as split-bf16 (hi*hi + "
                         "hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulation: ~2^-16 per product) for every "
                         "convolution, ConvTranspose and weight gradient incl. the first layer; GroupNorm, head and losses fp32",
           "tolerance_met": "1e-3 rel-L2 on logits and every gradient vs the reference "
                            "(tests: test_cfg2_128_against_reference_golden, test_cfg4_128_landmark_...)"}
    if patch == 128:  # three bf16 MFMAs per product
        rec["frac_of_bf16_mfma_peak_counting_3_mfma_per_product"] = round(
            3 * pps * FLOP_PER_PATCH / (MFMA_PEAK_TFLOPS["bf16"] * 1e12), 4)
    del model, step
    torch.cuda.empty_cache()
    return rec


def fp16_storage_mode(dev, batch: int, patch: int, steps: int, mode: str = "fp16"):
    """The SAME step in fp16 storage with the device-side dynamic loss scaler (the 16-bit kernels instantiated for the other
    element type).  Three more mantissa bits than bf16: logits and gradient norms within 1e-3 of the reference
    (tests: test_cfg2_128_fp16_storage_against_reference_golden) at the 16-bit modes' speed.  Timed the same way, as a sub-record."""
    import mednet_hip
    from mednet_hip.train import SegmentationStep
    from mednet_hip.unet.model import ResidualUNet3D
    from mednet_hip.synth import keyed_init_, synthetic_batch
    with mednet_hip.precision(mode):
        model = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=F_MAPS)).to(dev)
        step = SegmentationStep(model, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        b = {k: v.to(dev) for k, v in synthetic_batch(batch, 1, (patch, patch, patch), 4, 0, seed=1234).items()}
        for _ in range(3):
            step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        scale, _, taken, _ = step.scaler.snapshot()
        step.flat.release()
    pps = batch * steps / dt
    rec = {"value": round(pps, 3), "unit": "patches/s", "ms_per_step": round(1e3 * dt / steps, 2), "steps": steps, "warmup": 3,
           "dtype": "fp16", "loss": round(float(loss), 6), "loss_scale": float(scale), "optimizer_steps_taken": int(taken),
           "arithmetic": "fp16 storage of activations and gradients, fp16 matrix-core operands (v_mfma_f32_32x32x16_f16), fp32 "
                         "accumulation, fp32 master parameters, GroupNorm statistics and losses; dynamic loss scaling on the device",
           "tolerance_met": "not measured by this run; the bounds are the tests': test_cfg2_128_fp16_storage_against_reference_golden "
                            "holds strided logits and every gradient tensor's norm to 1e-3 vs the reference (N=1), "
                            "test_cfg2_timed_workload_against_the_live_oracle[fp16] holds the full tensors at this batch to the "
                            "16-bit bounds written there (full-tensor gradient rel-L2 of the worst tensors is ABOVE 1e-3)"}
    if patch == 128:
        rec["frac_of_fp16_mfma_peak"] = round(pps * FLOP_PER_PATCH / (MFMA_PEAK_TFLOPS["fp16"] * 1e12), 4)
    if mode == "fp16x2":
        rec["dtype"] = "fp16 storage, split weights"
        rec["arithmetic"] = ("fp16 storage of activations and gradients; every 3x3x3 forward / data-gradient convolution, ConvTranspose3d "
                             "and the first layer multiply the HIGH and the LOW fp16 image of the fp32 master weights (two "
                             "v_mfma_f32_32x32x16_f16 per product, fp32 accumulation; weight gradients one); GroupNorm statistics, head "
                             "and losses fp32; dynamic loss scaling on the device")
        rec["tolerance_met"] = ("not measured by this run; test_cfg2_timed_workload_against_the_live_oracle[fp16x2] holds logits and EVERY "
                                "gradient tensor of this batch to 1e-3 full-tensor rel-L2 against the fp32 oracle run live")
        if patch == 128:  # two MFMAs per product in 2/3 of the contractions
            rec["frac_of_fp16_mfma_peak_counting_the_low_image"] = round(pps * FLOP_PER_PATCH * (5.0 / 3.0) / (MFMA_PEAK_TFLOPS["fp16"] * 1e12), 4)
    del model, step
    torch.cuda.empty_cache()
    return rec


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, RCCL over xGMI) as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>` -- the command line the driver itself uses -- as a child process and return its exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="patches per GPU per step (config 2: 4)")
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp16", "fp16x2"])
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fp32-steps", type=int, default=10,
                    help="timed steps of the fp32 (1e-3 parity) mode for the fp32_parity_mode sub-record (0 = skip)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("MEDNET_GRAPH", "0")),
                    help="1: replay forward+loss+backward as one captured hipGraph per step (train._GraphedStep)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="every rank prints its RANK/LOCAL_RANK/WORLD_SIZE as one JSON line and exits (no GPU call): "
                         "checks the launcher on a box without GPUs")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Bare `python bench.py --gpus N`: this process becomes the launcher.  It has made no GPU call (importing torch
        # does not initialise HIP) and never will: the N ranks are CHILD processes of torch.distributed.run, rank 0's
        # JSON line passes through the inherited stdout, and the launcher exits with the children's return code.
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU, launch with "
                         f"--nproc-per-node {a.gpus} (or run bare `python bench.py --gpus {a.gpus}`, which launches the ranks)")
    if a.dry_launch:
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world_size": world,
                          "master": f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}",
                          "pid": os.getpid()}), flush=True)
        return
    assert torch.cuda.is_available(), (f"bench.py rank {rank}/{world} needs an MI355X (no CPU fallback for the product path)")
    # Rehearsal of an N > 1 run on a ONE-GPU box (never a measurement): MEDNET_REHEARSE_ONE_GPU=1 puts every rank on cuda:0 and
    # exchanges over gloo (RCCL refuses two ranks on one device); launcher, barriers, bucketed exchange, MAX-over-ranks and the
    # JSON line are the real ones.  The line is marked `rehearsal`.
    rehearse = os.environ.get("MEDNET_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("MEDNET_FORCE_DIST") == "1"  # the latter: 1-rank RCCL rehearsal
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    import mednet_hip
    from mednet_hip import _lib, ops
    from mednet_hip.train import SegmentationStep
    from mednet_hip.unet.model import ResidualUNet3D
    from mednet_hip.synth import keyed_init_, synthetic_batch

    assert _lib.lib().mednet_device_ok() == 1, "libmednet_hip.so sees no gfx950 device"
    mednet_hip.set_precision(a.precision)
    model = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=F_MAPS)).to(dev)
    step = SegmentationStep(model, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3, world_size=world, graph=bool(a.graph))
    P = a.patch
    batch = {k: v.to(dev) for k, v in synthetic_batch(a.batch, 1, (P, P, P), 4, 0, seed=1234 + rank).items()}

    if use_dist and world == 1:
        step.world = 1
        step.force_allreduce = True

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step(batch)
    sync()
    if not a.no_roofline:
        # dominant kernel: the 3x3x3 conv at full resolution with f0 -> f0 channels (forward launches; the data
        # gradient runs the same kernel).  HIP events on the launch stream (= torch's current stream).
        ops.PROFILE.update(enabled=True, events=[], wgrad_events=[], dgrad_events=[],
                           match=lambda k, ci, co, d, h, w: k == 3 and ci == F_MAPS[0] and co == F_MAPS[0] and d == P)
    if use_dist:
        from mednet_hip.train import BucketedExchange
        BucketedExchange.TIMING = []
    t0 = time.perf_counter()
    t_issue, n_issue = 0.0, min(a.steps, 5)
    for i in range(a.steps):
        loss = step(batch)
        if i + 1 == n_issue:  # host time to ENQUEUE the first steps (later the launch queue is full and the host waits)
            t_issue = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    ops.PROFILE["enabled"] = False
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss)

    if rank == 0:
        patches = a.batch * world * a.steps
        out = {
            "metric": "128^3 patches/sec training, 3D U-Net 32-base-ch", "value": round(patches / dt, 4),
            "unit": "patches/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"ResidualUNet3D f_maps={F_MAPS} 4-class, {P}^3 patches, batch {a.batch}/GPU, "
                                   f"fwd+DiceLoss+bwd+allreduce+Adam (BASELINE config {'2' if world == 1 else '3'})",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "loss": round(final_loss, 6),
                       "graph_replay": bool(a.graph),
                       "gradient_exchange": step._exchange.describe() if step._exchange is not None else "none",
                       # does the weight-gradient stream run beside the compute stream (its own hardware queue)?  ops.side_stream
                       "side_stream": {"enabled": bool(ops.SIDE["enabled"]), "overlaps": ops.SIDE.get("overlaps"),
                                       "candidates_tried": ops.SIDE.get("candidates_tried"),
                                       "wgrad_workgroups": ops.SIDE["wgrad_wgs"] or "one per CU"}},
            "host_enqueue_ms_per_step": round(1e3 * t_issue / n_issue, 3),
        }
        if use_dist:
            from mednet_hip.train import BucketedExchange
            ev = BucketedExchange.TIMING or []
            xms = [e0.elapsed_time(e1) for e0, e1 in ev]
            nbytes = step.flat.total * 4
            x_avg = sum(xms) / len(xms) if xms else None
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:  # (gloo rehearsal builds, or a torch without the binding)
                ver = f"unavailable ({type(exc).__name__})"
            out["rccl"] = {"world": dist.get_world_size(), "backend": dist.get_backend(), "nccl_version": ver,
                           "exchange_form": step._exchange.describe() if step._exchange is not None else "none",
                           "bytes": nbytes, "exchanges_timed": len(xms),
                           "allreduce_ms_per_step": None if x_avg is None else round(x_avg, 4),
                           # all-reduce bus bandwidth (the figure rccl-tests prints): bytes x 2 (n - 1) / n / time; 0 at one rank
                           "alg_GBps": None if not x_avg else round(nbytes / (x_avg * 1e-3) / 1e9, 2),
                           "bus_GBps": None if not x_avg else round(nbytes * 2 * (world - 1) / world / (x_avg * 1e-3) / 1e9, 2),
                           "timed_with": "HIP events on the compute stream around BucketedExchange.finish() of every timed step "
                                         "(what the step waits for; the early bucket of the two-bucket form runs inside backward)"}
        if rehearse:
            out["rehearsal"] = f"{world} ranks on ONE GPU over gloo: exercises the multi-rank code path, not a throughput"
        if P == 128 and a.precision == "bf16":
            out["model_flops_utilization"] = round(patches / dt * FLOP_PER_PATCH / (world * MFMA_PEAK_TFLOPS["bf16"] * 1e12), 4)
        if not a.no_roofline and ops.PROFILE["events"]:
            ev = ops.PROFILE["events"]
            ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
            flops = ev[0][2]
            avg = sum(ms) / len(ms)
            peak = MFMA_PEAK_TFLOPS[a.precision]
            ach = flops / (avg * 1e-3) / 1e12
            traffic, traffic_source = pmc_traffic(a.batch, P)
            out["roofline"] = {"kernel": "conv32_mfma_kernel<4>: conv3d 3x3x3 32->32 @128^3 (fwd launches)", "bound": "mfma",
                               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "traffic_source": traffic_source, "launches": len(ms),
                               "avg_ms": round(avg, 4), "flop_per_launch": flops}
            dev_ = ops.PROFILE.get("dgrad_events") or []
            if dev_:
                # the data gradients of the same layers run the OTHER epilogue variants of the same kernel (<1>: GroupNorm-backward
                # sums, <3>: + the summed residual gradient): slower than <4>.  `roofline_dgrad` reports them per variant and
                # `roofline.family` the FLOP-weighted fraction over every launch of the kernel family in the timed steps, so that
                # the forward variant's fraction does not stand in for the family.
                by = {}
                for e0, e1, fl, var in dev_:
                    by.setdefault(var, []).append(e0.elapsed_time(e1))
                names = {"gn": "conv32_mfma_kernel<1> (data gradient + GroupNorm-backward sums)",
                         "add+gn": "conv32_mfma_kernel<3> (data gradient + summed residual gradient + GroupNorm-backward sums)",
                         "add": "conv32_mfma_kernel<2> (data gradient + summed residual gradient)",
                         "plain": "conv32_mfma_kernel<0> (plain data gradient)"}
                var = {}
                for k, v in sorted(by.items()):
                    a_ms = sum(v) / len(v)
                    a_tf = flops / (a_ms * 1e-3) / 1e12
                    var[k] = {"kernel": names.get(k, k), "avg_ms": round(a_ms, 4), "launches": len(v),
                              "achieved": round(a_tf, 2), "frac": round(a_tf / peak, 4)}
                tot_ms = sum(ms) + sum(sum(v) for v in by.values())
                tot_n = len(ms) + sum(len(v) for v in by.values())
                fam = flops * tot_n / (tot_ms * 1e-3) / 1e12
                out["roofline"]["family"] = {"what": "all conv32_mfma_kernel launches of the timed steps (forward + data gradients), "
                                                     "FLOP-weighted", "launches": tot_n, "avg_ms": round(tot_ms / tot_n, 4),
                                             "achieved": round(fam, 2), "frac": round(fam / peak, 4)}
                out["roofline_dgrad"] = {"bound": "mfma", "peak": peak, "unit": "TFLOP/s", "flop_per_launch": flops, "variants": var}
        if not a.no_roofline and ops.PROFILE.get("wgrad_events"):
            # the kernel with the largest share of the step's kernel time: the weight gradient of the same layers.  In the step it
            # runs on the second stream, on all CUs, BESIDE the main stream's bandwidth-bound GroupNorm-backward passes (which is
            # what it was rebuilt for in round 5), so its in-step duration includes that sharing; `alone` is the same launch by
            # itself on the chip, on the step's own operands (timed after the loop, 20 launches).
            ev = ops.PROFILE["wgrad_events"]
            ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
            flops = ev[0][2]
            avg = sum(ms) / len(ms)
            peak = MFMA_PEAK_TFLOPS[a.precision]
            alone, alone_launches = wgrad_alone_ms(dev, a.batch, P, a.precision)
            if alone == alone:  # (NaN in the fp32 storage mode: another kernel runs there and this record does not describe it)
                out["roofline_wgrad"] = {"kernel": "wgrad_mfma4_kernel: weight gradient of conv3d 3x3x3 32->32 @128^3", "bound": "mfma",
                                         "achieved": round(flops / (alone * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                         "frac": round(flops / (alone * 1e-3) / 1e12 / peak, 4), "avg_ms": round(alone, 4),
                                         "launches": alone_launches, "flop_per_launch": flops,
                                         "in_step_beside_the_main_stream": {"avg_ms": round(avg, 4), "launches": len(ms),
                                                                            "achieved": round(flops / (avg * 1e-3) / 1e12, 2)}}
        if a.fp32_steps > 0 and world == 1 and a.precision == "bf16":
            del step, model
            torch.cuda.empty_cache()
            out["fp32_parity_mode"] = fp32_parity_mode(dev, a.batch, P, a.fp32_steps)
            out["fp16_mode"] = fp16_storage_mode(dev, a.batch, P, a.fp32_steps)
            out["fp16x2_mode"] = fp16_storage_mode(dev, a.batch, P, a.fp32_steps, "fp16x2")
        if a.cpu_steps > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.cpu_steps)
            out["cfg1_plumbing"] = cfg1_plumbing(dev)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
