"""Helpers for the -m gpu parity tests (HIP path vs the CPU oracle on identical seeded inputs)."""
import numpy as np
import torch

from oracle import ref_cpu as O

DEV = "cuda:0"
# tolerance of the parity metric ||a-b||2/||b||2: fp32 mode must meet the north-star 1e-3 with margin;
# bf16 mode stores activations/gradients in bf16 (8 significant bits): the reference's own bf16 drift is 8e-3/2e-2.
TOL = {"fp32": 1e-4, "bf16": 2.5e-2, "fp16": 4e-3}  # fp16 storage: 11 significant bits


def rnd(tag, *shape, scale=1.0):
    return torch.from_numpy((O._rng("in:" + tag).standard_normal(shape) * scale).astype(np.float32))


def bf16_round(t):
    return t.bfloat16().float()


def half_round(t, mode):
    """Inputs representable in the mode's storage type, so the comparison sees the kernels' error, not the input cast."""
    return t.bfloat16().float() if mode == "bf16" else (t.half().float() if mode == "fp16" else t)


def rel(a, b):
    return O.rel_l2(a.detach().float().cpu(), b.detach().float().cpu())


def assert_close(a, b, tol, what):
    r = rel(a, b)
    assert r <= tol, f"{what}: rel-L2 {r:.3e} > {tol:.1e}"
    return r


def copy_params(dst, src):
    """Same names => same values (the oracle/HIP module trees are key-compatible)."""
    dst.load_state_dict(src.state_dict())
    return dst
