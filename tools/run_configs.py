"""Throughput of the BASELINE configs on one MI355X (config 2; config 4: landmark path; config 5: 5-level/64-ch stress at
160x160x96; UNet3D variant of config 2) in the 16-bit modes and in the fp32 (1e-3) mode, each with the roofline of its dominant
convolution timed live.  Prints one JSON line per config."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip.synth import keyed_init_, synthetic_batch
from mednet_hip.train import SegmentationStep, LandmarkStep
from mednet_hip.unet.model import ResidualUNet3D, UNet3D

dev = torch.device("cuda", 0)
mednet_hip.set_precision("bf16")


# TFLOP/s dense; fp32 storage = three bf16 MFMAs per product, fp16x2 (fp16 storage + split weights) = two fp16 MFMAs per product
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp16x2": 2500.0 / 2, "fp32": 2500.0 / 3}


def run(name, make_step, batch, steps=10, warmup=3, precision="bf16", dominant=None):
    """`dominant` = (cin, cout, depth) of the config's dominant 3x3x3 convolution: its launches are timed live with HIP events
    on the launch stream (ops.PROFILE, as bench.py does) -> `roofline` of that kernel for this config."""
    from mednet_hip import ops
    mednet_hip.set_precision(precision)
    step = make_step()
    for _ in range(warmup):
        out = step(batch)
    torch.cuda.synchronize()
    if dominant is not None:
        ci, co, dd = dominant
        ops.PROFILE.update(enabled=True, events=[], match=lambda k, a, b, d, h, w: k == 3 and a == ci and b == co and d == dd)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.PROFILE["enabled"] = False
    roof = None
    if dominant is not None and ops.PROFILE["events"]:
        ev = ops.PROFILE["events"]
        ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
        avg = sum(ms) / len(ms)
        ach = ev[0][2] / (avg * 1e-3) / 1e12
        roof = {"kernel": f"conv3d 3x3x3 {dominant[0]}->{dominant[1]} at depth {dominant[2]} (forward launches)", "bound": "mfma",
                "achieved": round(ach, 1), "peak": round(PEAK[precision], 1), "unit": "TFLOP/s (real FLOP)",
                "frac": round(ach / PEAK[precision], 4), "launches": len(ms), "avg_ms": round(avg, 4), "traffic": None}
    n = batch["data"].shape[0]
    loss = float(out[0] if isinstance(out, tuple) else out)
    extra = {}
    if getattr(step, "scaler", None) is not None:
        sc = step.scaler.snapshot()
        extra = {"loss_scale": sc[0], "optimizer_steps_taken": int(sc[2]), "steps_run": steps + warmup}
    print(json.dumps({"config": name, "dtype": precision, "patches_per_s": round(n / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                      "loss": round(loss, 5), **extra, **({"roofline": roof} if roof else {}),
                      "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    step.flat.release()
    del step
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


which = os.environ.get("RC_WHICH", "cfg2,cfg4,unet3d,cfg5")
# fp32 = the 1e-3 mode with fp32 storage (split-bf16 contraction); fp16x2 = the 1e-3 mode with fp16 storage (split weights, round 6)
precs = os.environ.get("RC_PREC", "bf16,fp16x2,fp32").split(",")
if "cfg2" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1234).items()}
    for prec in precs:
        run(f"cfg2: ResidualUNet3D [32,64,128,256] 4-class, 128^3, batch 4, {prec} storage",
            lambda: SegmentationStep(keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev), [0.05, 1, 1, 1.0]), b,
            precision=prec, dominant=(32, 32, 128))
if "cfg4" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 2, 16, seed=1234).items()}
    for prec in precs:
        run(f"cfg4 landmark: ResidualUNet3D [32,64,128,256] out=18 (16 heat maps + 2 classes), 128^3, batch 4, {prec} storage",
            lambda: LandmarkStep(keyed_init_(ResidualUNet3D(1, 18, False, f_maps=[32, 64, 128, 256])).to(dev), [0.05, 1.0], [0.015] * 16, "L2"), b,
            precision=prec, dominant=(32, 32, 128))
if "unet3d" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1234).items()}
    run("cfg2 with UNet3D [32,64,128,256] 4-class, 128^3, batch 4, bf16",
        lambda: SegmentationStep(keyed_init_(UNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev), [0.05, 1, 1, 1.0]), b,
        dominant=(32, 32, 128))
    if "fp32" in precs:
        # the slow corner of the 1e-3 mode: UNet3D's default order 'gcr' has ReLU, so every contraction asks for EXACT fp32
        # products (config.exact_products -> MEDNET_ALGO_EXACT: v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) instead of the
        # split-bf16 contraction the smooth 'cge' networks take
        PEAK["fp32"] = 157.3
        run("cfg2 with UNet3D [32,64,128,256] 'gcr' 4-class, 128^3, batch 4, fp32 storage, exact fp32 products (ALGO_EXACT)",
            lambda: SegmentationStep(keyed_init_(UNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev), [0.05, 1, 1, 1.0]), b,
            steps=3, warmup=1, precision="fp32", dominant=(32, 32, 128))
        PEAK["fp32"] = 2500.0 / 3
if "cfg5" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(2, 1, (160, 160, 96), 4, 0, seed=1234).items()}
    for prec in (precs if "cfg5only" in which else ["fp16"] + precs):
        run(f"cfg5: ResidualUNet3D [64,128,256,512,1024] 4-class, 160x160x96, batch 2, {prec} storage"
            + (" + dynamic loss scaling (the mode BASELINE names)" if prec == "fp16" else ""),
            lambda: SegmentationStep(keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[64, 128, 256, 512, 1024])).to(dev), [0.05, 1, 1, 1.0]),
            b, steps=5, warmup=2, precision=prec, dominant=(64, 64, 160))
