"""Throughput of the other BASELINE configs on one MI355X (config 4: landmark path; config 5: 5-level/64-ch stress at
160x160x96; UNet3D variant of config 2).  Prints one line per config."""
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


def run(name, make_step, batch, steps=5, warmup=2, precision="bf16"):
    mednet_hip.set_precision(precision)
    step = make_step()
    for _ in range(warmup):
        out = step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = batch["data"].shape[0]
    loss = float(out[0] if isinstance(out, tuple) else out)
    extra = {}
    if getattr(step, "scaler", None) is not None:
        sc = step.scaler.snapshot()
        extra = {"loss_scale": sc[0], "optimizer_steps_taken": int(sc[2]), "steps_run": steps + warmup}
    print(json.dumps({"config": name, "dtype": precision, "patches_per_s": round(n / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                      "loss": round(loss, 5), **extra,
                      "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    step.flat.release()
    del step
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


which = os.environ.get("RC_WHICH", "cfg4,unet3d,cfg5")
if "cfg4" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 2, 16, seed=1234).items()}
    run("cfg4 landmark: ResidualUNet3D [32,64,128,256] out=18 (16 heat maps + 2 classes), 128^3, batch 4, bf16",
        lambda: LandmarkStep(keyed_init_(ResidualUNet3D(1, 18, False, f_maps=[32, 64, 128, 256])).to(dev), [0.05, 1.0], [0.015] * 16, "L2"), b)
if "unet3d" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=1234).items()}
    run("cfg2 with UNet3D [32,64,128,256] 4-class, 128^3, batch 4, bf16",
        lambda: SegmentationStep(keyed_init_(UNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev), [0.05, 1, 1, 1.0]), b)
if "cfg5" in which:
    b = {k: v.to(dev) for k, v in synthetic_batch(2, 1, (160, 160, 96), 4, 0, seed=1234).items()}
    for prec in ("fp16", "bf16"):
        run(f"cfg5: ResidualUNet3D [64,128,256,512,1024] 4-class, 160x160x96, batch 2, {prec} storage"
            + (" + dynamic loss scaling (the mode BASELINE names)" if prec == "fp16" else ""),
            lambda: SegmentationStep(keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[64, 128, 256, 512, 1024])).to(dev), [0.05, 1, 1, 1.0]),
            b, steps=5, warmup=2, precision=prec)
