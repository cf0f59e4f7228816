#!/usr/bin/env python3
"""Which STORED tensor's 16-bit rounding produces the gradient error of the 16-bit storage modes?  

CPU emulation on the oracle (oracle/ref_cpu.py, fp32 arithmetic): a rounding node is put behind every tensor the HIP path stores
in 16 bits, one CLASS at a time --
    y   conv / ConvTranspose outputs (GroupNorm inputs)                    forward
    z   activation outputs (the next conv's operand, the weight gradient's operand)   forward
    w   the 16-bit weight images of the 3x3x3 matrix-core convolutions (forward and data gradient read them; master weights fp32)
    dy  gradient at a conv output (operand of the data gradient and of the weight gradient)   backward
    dz  gradient at an activation output (what the data-gradient conv stores)       backward
-- everything else stays fp32 (the HIP path accumulates in fp32 and keeps statistics, parameters and losses in fp32), and the
per-tensor rel-L2 of every parameter gradient against the unrounded oracle is printed.  `--scale` is the (single, global) loss
scale of the fp16 mode; `--norm-dy` rescales every backward tensor by a power of two chosen from ITS OWN maximum before rounding
(what a per-tensor scale would do: no underflow, no overflow), which separates range effects from mantissa effects.

    python tools/attrib16.py --size 64 --dtype fp16
"""
import argparse
import os
import re
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as O  # noqa: E402

CFG = {"fwd": set(), "bwd": set(), "dtype": torch.float16, "scale": 65536.0, "norm": False, "split": set(), "log": None}


def _round(t, split=False):
    dt = CFG["dtype"]
    hi = t.to(dt).float()
    if split:  # hi + lo pair of 16-bit values
        return hi + (t - hi).to(dt).float()
    return hi


class Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fcls, bcls):
        ctx.bcls = bcls
        if fcls in CFG["fwd"]:
            return _round(x, fcls in CFG["split"])
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        b = ctx.bcls
        if b in CFG["bwd"]:
            s = CFG["scale"]
            if CFG["norm"]:
                m = float(g.abs().max())
                s = 2.0 ** (14 - int(torch.tensor(m).log2().ceil())) if m > 0 else 1.0
            r = _round(g * s, b in CFG["split"]) / s
            if CFG["log"] is not None:
                nz = float((g != 0).sum())
                CFG["log"].append((b, float((r == 0).sum() - (g == 0).sum()) / max(nz, 1.0), float(g.abs().max()) * s))
            return r, None, None
        return g, None, None


class WQ(torch.autograd.Function):
    """The matrix-core kernels read 16-bit WEIGHT images (packed from the fp32 master parameters): the forward convolution reads
    one (class `wf`), the data gradient its transposed twin (class `wb`); the weight gradient lands on the master parameter
    (straight through).  `w` = both."""

    @staticmethod
    def forward(ctx, w, name, cls):
        if cls not in CFG["fwd"] and "w" not in CFG["fwd"]:
            return w
        rx = CFG.get("splitw")
        return _round(w, "w" in CFG["split"] or cls in CFG["split"] or (rx is not None and re.search(rx, name) is not None))

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def _patch_weight(m, name):
    """y = conv(x, Wf) in value; d/dx through Wb; d/dW lands on the master weight with the true x (two convolutions forward:
    A carries the value and the weight gradient, B - B.detach() == 0 carries the data gradient)."""
    if isinstance(m, nn.ConvTranspose3d):
        op = lambda x, w, b: nn.functional.conv_transpose3d(x, w, b, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
    else:
        op = lambda x, w, b: m._conv_forward(x, w, b)

    def fwd(x, m=m):
        separate = bool({"wf", "wb"} & (CFG["fwd"] | CFG["split"]))
        if not separate or not x.requires_grad:
            return op(x, WQ.apply(m.weight, name, "wf"), m.bias)
        a_ = op(x.detach(), WQ.apply(m.weight, name, "wf"), m.bias)
        b_ = op(x, WQ.apply(m.weight, name, "wb").detach(), None)
        return a_ + (b_ - b_.detach())

    m.forward = fwd


def instrument(model):
    for name, m in model.named_modules():
        if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)) and not name.endswith("final_conv") and m.kernel_size[0] == 3 and (
                m.in_channels > 1 or os.environ.get("ATTRIB_FIRST_LAYER") == "1"):
            _patch_weight(m, name)  # (the first layer keeps ~16 mantissa bits of its input and the head runs in fp32: DESIGN section 4)
        if isinstance(m, (nn.ELU, nn.ReLU, nn.LeakyReLU)):
            m.inplace = False
            m.register_forward_hook(lambda mod, inp, out: Q.apply(out, "z", "dz"))
        elif isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)) and not name.endswith("final_conv"):
            m.register_forward_hook(lambda mod, inp, out: Q.apply(out, "y", "dy"))
    return model


def grads(model, batch, crit):
    model.zero_grad()
    loss = O.seg_training_step(model, crit, batch)
    loss.backward()
    return float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--scale", type=float, default=65536.0)
    ap.add_argument("--arms", default="y;z;dy;dz;y,z;dy,dz;y,z,dy,dz;y,z,dy,dz+norm;y,z,dy,dz+split:dy;y,z,dy,dz+split:dy,dz;y,z,dy,dz+split:z;y,z,dy,dz+split:y")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    CFG["dtype"] = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    CFG["scale"] = a.scale if a.dtype == "fp16" else 1.0
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = O.synthetic_batch(a.n, 1, (a.size,) * 3, 4, 0, seed=1234)
    crit = O.DiceLoss(weight=torch.tensor([0.05, 1.0, 1.0, 1.0]))
    model = instrument(O.keyed_init_(O.ResidualUNet3D(**ctor)))
    t0 = time.perf_counter()
    loss0, g0 = grads(model, batch, crit)
    print(f"# ResidualUNet3D cfg2, N={a.n}, {a.size}^3, {a.dtype} storage emulated on the fp32 oracle, loss scale {CFG['scale']:g}; "
          f"baseline pass {time.perf_counter() - t0:.1f} s, loss {loss0:.6f}")
    print("| rounded classes | loss diff | concatenated grad rel-L2 | median tensor | worst five tensors (full-tensor rel-L2) |")
    print("|---|---|---|---|---|")
    for arm in a.arms.split(";"):
        spec, _, opt = arm.partition("+")
        cls = set(spec.split(","))
        CFG["fwd"], CFG["bwd"] = cls & {"y", "z", "w", "wf", "wb"}, cls & {"dy", "dz"}
        CFG["norm"] = opt == "norm"
        CFG["split"] = set(opt[6:].split(",")) if opt.startswith("split:") else set()
        CFG["splitw"] = opt[7:] if opt.startswith("splitw:") else None  # regex over module names: hi + lo weight images there only
        CFG["log"] = []
        loss, g = grads(model, batch, crit)
        per, num, den = [], 0.0, 0.0
        for k in g0:
            d2 = float((g[k].double() - g0[k].double()).pow(2).sum())
            n2 = float(g0[k].double().pow(2).sum())
            per.append(((d2 / n2) ** 0.5, k))
            num += d2
            den += n2
        per.sort(reverse=True)
        fl = ""
        if CFG["log"]:
            worst_flush = max(f for _, f, _ in CFG["log"])
            mx = max(m for _, _, m in CFG["log"])
            fl = f" [flushed to zero: worst tensor {100 * worst_flush:.2f} % of its non-zeros; largest scaled |g| {mx:.3g}]"
        print(f"| {arm} | {abs(loss - loss0):.1e} | {(num / den) ** 0.5:.2e} | {per[len(per) // 2][0]:.2e} | "
              + ", ".join(f"{k} {r:.2e}" for r, k in per[:5]) + fl + " |", flush=True)


if __name__ == "__main__":
    main()
