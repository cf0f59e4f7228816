"""Throughput of the training-patch sampler on one MI355X: 8 subjects of 256^3 (fp16 image, uint8
3-class label map), 128^3 patches, class probabilities [0.2, 0.4, 0.4], batches of 4.  Prints one JSON line with the
device sampler's rate, the CPU oracle's (numpy crop, the reference's procedure) and the rate of sampler + training step."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import numpy as np
import torch
import mednet_hip
from mednet_hip.sampler import DevicePatchSampler
from mednet_hip.synth import keyed_init_
from mednet_hip.train import SegmentationStep
from mednet_hip.unet.model import ResidualUNet3D
from oracle import ref_sampler as S

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n_subj, size, patch, probs, B = 8, 256, [128] * 3, [0.2, 0.4, 0.4], 4
images, labels = [], []
for s in range(n_subj):
    images.append(rng.standard_normal((1, size, size, size), dtype=np.float32).astype(np.float16))
    lab = np.zeros((1, size, size, size), dtype=np.uint8)
    for c in (1, 2):
        for _ in range(6):
            z, y, x = rng.integers(0, size - 24, 3)
            lab[0, z:z + 24, y:y + 24, x:x + 24] = c
    labels.append(lab)
ds = DevicePatchSampler(images, labels, patch, samples_per_subject=16, class_probabilities=probs, device=dev)
np.random.seed(1)
for _ in range(3):
    ds.batch(range(B))
torch.cuda.synchronize()
t0 = time.perf_counter()
nb = 50
for i in range(nb):
    b = ds.batch(range(i * B, i * B + B))
torch.cuda.synchronize()
dt_dev = (time.perf_counter() - t0) / nb
# with the reference's intensity augmentation (train_seg.py:82-86) on the device
dsa = DevicePatchSampler(images, labels, patch, samples_per_subject=16, class_probabilities=probs, device=dev, augment=True)
for _ in range(3):
    dsa.batch(range(B))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(nb):
    b = dsa.batch(range(i * B, i * B + B))
torch.cuda.synchronize()
dt_aug = (time.perf_counter() - t0) / nb
from oracle import ref_augment as A
ora = S.PatchSampler(images, labels, patch, samples_per_subject=16, class_probabilities=probs)
np.random.seed(1)
t0 = time.perf_counter()
for i in range(2):
    items = [ora[j] for j in range(i * B, i * B + B)]
    d = np.stack([a["data"] for a in items])
    A.apply(d, A.draw_parameters(B, d.shape[1]))
dt_cpu_aug = (time.perf_counter() - t0) / 2
np.random.seed(1)
t0 = time.perf_counter()
nc = 6
for i in range(nc):
    items = [ora[j] for j in range(i * B, i * B + B)]
    batch_cpu = {"data": np.stack([a["data"] for a in items]), "label": np.stack([a["label"] for a in items])}
dt_cpu = (time.perf_counter() - t0) / nc
# sampler feeding the training step (config-2 network, bf16 storage)
mednet_hip.set_precision("bf16")
net = keyed_init_(ResidualUNet3D(1, 4, False, f_maps=[32, 64, 128, 256])).to(dev)
step = SegmentationStep(net, loss_weight=[0.05, 1, 1, 1.0])
for i in range(3):
    step(ds.batch(range(B)))
torch.cuda.synchronize()
t0 = time.perf_counter()
ns = 20
for i in range(ns):
    step(ds.batch(range(i * B, i * B + B)))
torch.cuda.synchronize()
dt_train = (time.perf_counter() - t0) / ns
print(json.dumps({"metric": "128^3 training patches/sec sampled (position + crop + cast into batch tensors)",
                  "value": round(B / dt_dev, 1), "unit": "patches/s", "ms_per_batch_of_4": round(dt_dev * 1e3, 3),
                  "cpu_baseline": {"value": round(B / dt_cpu, 2), "unit": "patches/s", "kind": "port",
                                   "sample": f"oracle PatchSampler (numpy crop + stack), {nc} batches of 4, 1 process"},
                  "with_augmentation": {"value": round(B / dt_aug, 1), "unit": "patches/s", "ms_per_batch_of_4": round(dt_aug * 1e3, 3),
                                        "what": "+ brightness / gamma / contrast of train_seg.py:82-86 (mednet_augment_patches)",
                                        "cpu_baseline": {"value": round(B / dt_cpu_aug, 2), "unit": "patches/s", "kind": "port",
                                                         "sample": "oracle sampler + oracle/ref_augment.py (numpy), 2 batches of 4, 1 process"}},
                  "sampler_plus_training_step": {"value": round(B / dt_train, 2), "unit": "patches/s",
                                                 "ms_per_step": round(dt_train * 1e3, 2)}}))
