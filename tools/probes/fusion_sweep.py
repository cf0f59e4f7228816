"""Randomised sweep of the round-4 fusions against the launches they replace: random network widths, depths, volume sizes (odd and
even), batch sizes, pooling types, storage modes.  For every draw a fresh network takes one training step with the fusions on and
one with them off (block.FUSE_POOL, ops.LAZY_POOL, ops.POOL_ACT_MASK, ops.UPCAT_VIEW; option upcat_stats):
  ResidualUNet3D: loss and every gradient must be BIT-identical;
  UNet3D: bit-identical with upcat_stats held fixed, and within 2e-6 (relative L2 per tensor, fp32 mode) with the fused
  concatenation statistics (another summation order).
usage: python tools/probes/fusion_sweep.py [draws] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import ops, block, _lib as L, nn as hnn
from mednet_hip.synth import keyed_init_, synthetic_batch
from mednet_hip.train import SegmentationStep
from mednet_hip.unet.model import ResidualUNet3D, UNet3D

DEV = "cuda:0"
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
lib = L.lib()


def run(cls, ctor, batch, mode, fused, pool, upcat_stats):
    block.FUSE_POOL = ops.LAZY_POOL = ops.POOL_ACT_MASK = ops.UPCAT_VIEW = fused
    lib.mednet_set_option(b"upcat_stats", int(upcat_stats))
    with mednet_hip.precision(mode):
        net = keyed_init_(cls(**ctor))
        if pool == "avg":
            for enc in net.encoders[1:]:
                enc.pooling = hnn.AvgPool3d(kernel_size=(2, 2, 2))
        step = SegmentationStep(net.to(DEV), loss_weight=None, lr=1e-3)
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
        out = (float(loss), step.flat.grad.clone())
        step.flat.release()
    return out


bad = 0
for i in range(draws):
    levels = rng.choice([2, 3, 3, 4])
    base = rng.choice([16, 32, 32, 64])
    f_maps = [base * 2 ** k for k in range(levels)]
    q = 2 ** (levels - 1)
    shape = tuple(rng.choice([q * rng.randint(1, 5), q * rng.randint(1, 5) + rng.randint(0, q - 1)]) for _ in range(3))
    shape = tuple(max(s, q) for s in shape)
    n = rng.choice([1, 2, 3])
    ncls = rng.choice([2, 3, 4])
    mode = rng.choice(["bf16", "bf16", "fp16", "fp32"])
    pool = rng.choice(["max", "max", "avg"])
    cls = rng.choice([ResidualUNet3D, UNet3D])
    if cls is ResidualUNet3D:  # (its ConvTranspose doubles exactly: the reference needs sizes divisible by 2^(levels-1) too)
        shape = tuple(max(q, s // q * q) for s in shape)
    if cls is UNet3D and mode == "fp32" and max(shape) > 24:
        shape = tuple(min(s, 24) for s in shape)  # (exact fp32 products: slow)
        shape = tuple(max(s, q) for s in shape)
    ctor = dict(in_channels=1, out_channels=ncls, final_sigmoid=False, f_maps=f_maps)
    batch = {k: v.to(DEV) for k, v in synthetic_batch(n, 1, shape, ncls, 0, seed=100 + i).items()}
    tag = f"{cls.__name__} f_maps={f_maps} shape={shape} n={n} classes={ncls} {mode} pool={pool}"
    try:
        off = run(cls, ctor, batch, mode, False, pool, False)
        on = run(cls, ctor, batch, mode, True, pool, False)
        ok = off[0] == on[0] and torch.equal(off[1], on[1])
        msg = "bit-identical" if ok else f"DIFFERS: loss {off[0]!r} vs {on[0]!r}, {int((off[1] != on[1]).sum())} gradient values"
        if cls is UNet3D:
            st = run(cls, ctor, batch, mode, True, pool, True)
            rel = float((st[1].double() - on[1].double()).norm() / on[1].double().norm().clamp_min(1e-30))
            lim = 2e-6 if mode == "fp32" else 2e-2
            ok = ok and rel <= lim and abs(st[0] - on[0]) <= 1e-3 * max(1.0, abs(on[0]))
            msg += f"; fused concatenation statistics: gradient rel-L2 {rel:.2e} (limit {lim:g})"
    except Exception as e:  # noqa: BLE001
        ok, msg = False, f"EXCEPTION {type(e).__name__}: {e}"
    bad += not ok
    print(("ok   " if ok else "FAIL ") + tag + " -> " + msg, flush=True)
lib.mednet_set_option(b"upcat_stats", 1)
print(f"{draws - bad} of {draws} draws passed")
sys.exit(1 if bad else 0)
