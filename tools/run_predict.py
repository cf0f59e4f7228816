"""Throughput of the inference path on one MI355X: BASELINE config-2 network, a 320^3 fp16 volume,
128^3 patches with a 16-voxel overlap (96^3 kept per patch, 64 patches), bf16 storage.  Prints one JSON line; with
--cpu also times the CPU oracle's forward on ONE patch (bounded sample) on the box's host cores."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import numpy as np
import torch
import mednet_hip
from mednet_hip import predict as HP
from mednet_hip.synth import keyed_init_
from mednet_hip.unet.model import ResidualUNet3D

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=320)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
mednet_hip.set_precision("bf16")
net = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev)
vol = torch.from_numpy(np.random.default_rng(0).standard_normal((1, a.size, a.size, a.size)).astype(np.float16)).to(dev)
pred = HP.GridPredictor(net, [128] * 3, [16] * 3, num_heatmaps=0, pad_mode="constant", batch_size=a.batch)
n_patches = len(HP.grid_positions(vol.shape[1:], [128] * 3, [16] * 3))
r = pred(vol)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    r = pred(vol)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
out = {"metric": "128^3 patches/sec inference (grid gather + forward + arg-max/stitch)", "value": round(n_patches / dt, 2),
       "unit": "patches/s", "volume": [a.size] * 3, "patches": n_patches, "s_per_volume": round(dt, 4),
       "Mvoxel_per_s": round(a.size ** 3 / dt / 1e6, 1), "dtype": "bf16", "labels_hist": torch.bincount(r[0].flatten().int(), minlength=4).tolist()}
if a.cpu:
    sys.path.insert(0, ROOT)
    from bench import host_cores
    torch.set_num_threads(host_cores())
    from oracle import ref_cpu as O
    ora = O.keyed_init_(O.ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).eval()
    x = torch.randn(1, 1, 128, 128, 128)
    with torch.no_grad():
        ora(x)
        t0 = time.perf_counter()
        ora(x)
        ora(x)
        dtc = (time.perf_counter() - t0) / 2
    out["cpu_baseline"] = {"value": round(1.0 / dtc, 4), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": "oracle forward of one 128^3 patch, fp32, 1 warm-up + 2 timed"}
print(json.dumps(out))
