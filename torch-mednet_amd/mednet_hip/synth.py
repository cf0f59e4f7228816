"""Deterministic synthetic inputs and name-keyed weights for benchmarks and demos.  Same streams as the
test oracle's generators (tests/test_ddp_gloo.py::test_synth_generators_match_oracle pins that), but owned by the product so bench.py's GPU
leg never touches oracle/."""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(tag: str, seed: int = 0):
    return np.random.Generator(np.random.PCG64((zlib.crc32(tag.encode()) << 8) ^ seed))


def keyed_init_(module: torch.nn.Module, seed: int = 0):
    """Every parameter from a PCG64 stream keyed by crc32(parameter name): conv weights/biases U(-1/sqrt(fan_in), ..),
    norm weight U(0.5,1.5), norm bias U(-0.3,0.3)."""
    with torch.no_grad():
        named = dict(module.named_parameters())
        for name, p in named.items():
            g = _rng(name, seed)
            if p.dim() == 5:
                fan_in = (p.shape[0] if "upsample" in name else p.shape[1]) * p.shape[2] * p.shape[3] * p.shape[4]
                b = 1.0 / np.sqrt(fan_in)
                v = g.uniform(-b, b, size=tuple(p.shape))
            elif name.endswith("norm.weight"):
                v = g.uniform(0.5, 1.5, size=tuple(p.shape))
            elif name.endswith("norm.bias"):
                v = g.uniform(-0.3, 0.3, size=tuple(p.shape))
            else:
                w = named.get(name[:-4] + "weight")
                if w is None:
                    fan_in = p.numel()
                elif "upsample" in name:
                    fan_in = w.shape[0] * int(np.prod(w.shape[2:]))
                else:
                    fan_in = int(np.prod(w.shape[1:]))
                b = 1.0 / np.sqrt(max(fan_in, 1))
                v = g.uniform(-b, b, size=tuple(p.shape))
            p.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def synthetic_batch(n, cin, shape, n_classes, n_heatmaps=0, seed=1234):
    """The dict MedDataset emits (dataset.py:332-346): data fp32 N(0,1); label uint8 with `n_heatmaps` heat-map
    channels (uniform 0..255) followed by the class-id channel."""
    g = np.random.Generator(np.random.PCG64(seed))
    data = g.standard_normal((n, cin) + tuple(shape), dtype=np.float32)
    chans = []
    if n_heatmaps:
        chans.append(g.integers(0, 256, size=(n, n_heatmaps) + tuple(shape), dtype=np.uint8))
    chans.append(g.integers(0, n_classes, size=(n, 1) + tuple(shape), dtype=np.uint8))
    return {"data": torch.from_numpy(data), "label": torch.from_numpy(np.concatenate(chans, axis=1))}
