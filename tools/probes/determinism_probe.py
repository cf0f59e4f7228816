"""Run-to-run determinism of a whole training step (forward + loss + backward, both streams), localised to the kernel.

    python tools/probes/determinism_probe.py --config cfg5 --mode bf16 --plain 60 --traced 8 --poisoned 4

* plain runs: the step exactly as tests/test_gpu_network.py::test_cfg5_full_size_properties runs it (a fresh network and
  trainer per run, nothing else on the device); loss and the flat gradient buffer are compared bit for bit with run 0, per
  parameter tensor.
* traced runs: mednet_hip.debug records a checksum of every tensor the ops produce (activations, GroupNorm partial rows,
  statistics, coefficients, gradients); the first trace point that differs from run 0's names the kernel.
* poisoned runs: the caching allocator's free memory and the workspaces are filled with NaN patterns before the run
  (mednet_hip.debug.poison); a kernel reading memory nobody wrote then yields NaN or a changed result deterministically.
Prints one JSON line per phase; exit code 1 if anything differed.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "torch-mednet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import mednet_hip  # noqa: E402
from mednet_hip import debug  # noqa: E402
from mednet_hip.train import SegmentationStep  # noqa: E402
from mednet_hip.unet import model as HM  # noqa: E402
from mednet_hip import synth as O  # noqa: E402  (the product's own synthetic batch + keyed initialisation: nothing from oracle/)

CONFIGS = {
    "cfg5": dict(f_maps=[64, 128, 256, 512, 1024], size=(160, 160, 96), n=2),
    "cfg2": dict(f_maps=[32, 64, 128, 256], size=(128, 128, 128), n=4),
    "small": dict(f_maps=[32, 64], size=(32, 32, 32), n=2),
}


_WEIGHTS = {}  # keyed initialisation of a config's network, made once on the CPU (141 M parameters take seconds per run)


def fresh_net(cfg):
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=cfg["f_maps"])
    key = tuple(cfg["f_maps"])
    if key not in _WEIGHTS:
        _WEIGHTS[key] = {k: v.clone() for k, v in O.keyed_init_(HM.ResidualUNet3D(**ctor)).state_dict().items()}
    net = HM.ResidualUNet3D(**ctor)
    net.load_state_dict(_WEIGHTS[key])
    return net.to("cuda")


def one_run(cfg, mode, batch, traced=False):
    with mednet_hip.precision(mode):
        net = fresh_net(cfg)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        if traced:
            debug.open_trace()
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
        tr = debug.close_trace() if traced else None
        grads = step.flat.grad.clone()
        names = [(n, off, p.numel()) for (n, p), off in zip(net.named_parameters(), step.flat.offsets)]
        step.flat.release()
        del net, step
    return float(loss), grads, names, tr


def forward_soak(cfg, mode, batch, iters):
    """`iters` forward + loss passes of ONE network (training mode, graph built and dropped); a cheap checksum of every
    encoder / decoder / head output and of the loss per pass, compared with the first pass: a deviation names the module."""
    from mednet_hip.unet import loss as HL
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=cfg["f_maps"])
    marks = []

    def cheap(t):
        if t.dim() == 5 and not t.is_contiguous():
            t = t.permute(0, 2, 3, 4, 1)
        return t.reshape(-1).view(torch.int32).sum(dtype=torch.int64)

    def hook(name):
        def fn(mod, inp, out):
            o = out[-1] if isinstance(out, tuple) else out
            marks.append((name, cheap(o.detach())))
        return fn

    bad = []
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to("cuda")
        for i, m in enumerate(net.encoders):
            m.register_forward_hook(hook(f"encoder{i}"))
        for i, m in enumerate(net.decoders):
            m.register_forward_hook(hook(f"decoder{i}"))
        net.final_conv.register_forward_hook(hook("final_conv"))
        loss_fn = HL.DiceLoss(weight=torch.tensor([0.05, 1.0, 1.0, 1.0], device="cuda")).to("cuda")
        x = batch["data"].float()
        y = batch["label"][:, -1, ...].long()
        ref = None
        for it in range(iters):
            marks.clear()
            loss = loss_fn(net(x), y)
            vals = torch.stack([c for _, c in marks] + [loss.detach().view(1).view(torch.int32).to(torch.int64)[0]])
            del loss
            if ref is None:
                ref, names = vals.clone(), [n for n, _ in marks] + ["loss"]
            elif not torch.equal(vals, ref):
                first = int((vals != ref).nonzero()[0])
                bad.append({"iter": it, "first_module": names[first], "modules_differing": int((vals != ref).sum())})
    return {"phase": "forward_soak", "iters": iters, "modules": len(ref), "deviations": bad[:20], "n_deviations": len(bad)}


def grad_diffs(g0, g1, names):
    bad = []
    for name, off, numel in names:
        a, b = g0[off:off + numel], g1[off:off + numel]
        if not torch.equal(a, b):
            nd = int((a != b).sum())
            bad.append((name, nd, numel, bool(torch.isfinite(b).all())))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg5")
    ap.add_argument("--mode", default="bf16")
    ap.add_argument("--plain", type=int, default=40)
    ap.add_argument("--traced", type=int, default=6)
    ap.add_argument("--poisoned", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--churn", type=int, default=0, help="full steps cycling through the storage modes (allocator churn)")
    ap.add_argument("--soak", type=int, default=0, help="forward-only passes with per-module checksums")
    ap.add_argument("--exit-zero", action="store_true", help="report only (a chain of probes in one GPU call)")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    batch = {k: v.to("cuda") for k, v in O.synthetic_batch(cfg["n"], 1, cfg["size"], 4, 0, seed=a.seed).items()}
    failed = False

    if a.soak:
        rec = forward_soak(cfg, a.mode, batch, a.soak)
        rec.update(config=a.config, mode=a.mode)
        failed |= rec["n_deviations"] > 0
        print(json.dumps(rec), flush=True)

    if a.churn:
        # the driver's failing sequence, many times: runs of the 16-bit mode interleaved with runs in the other storage modes
        # (different tensor sizes: the caching allocator hands every run another layout of recycled blocks)
        refs, bad = {}, []
        order = [a.mode, a.mode, "fp32", "fp16" if a.mode == "bf16" else "bf16", a.mode]
        for i in range(a.churn):
            md = order[i % len(order)]
            l1, g1, nm, _ = one_run(cfg, md, batch)
            if md not in refs:
                refs[md] = (l1, g1)
            elif l1 != refs[md][0] or not torch.equal(g1, refs[md][1]):
                bad.append({"run": i, "mode": md, "loss": l1, "loss_ref": refs[md][0], "grad_tensors_differing": len(grad_diffs(refs[md][1], g1, nm))})
        rec = {"phase": "churn", "config": a.config, "runs": a.churn, "order": order, "mismatches": bad[:20], "n_mismatches": len(bad)}
        failed |= bool(bad)
        print(json.dumps(rec), flush=True)
        del refs

    l0, g0, names, _ = one_run(cfg, a.mode, batch)
    rec = {"phase": "plain", "config": a.config, "mode": a.mode, "runs": a.plain, "loss0": l0, "mismatches": []}
    for i in range(1, a.plain + 1):
        l1, g1, _, _ = one_run(cfg, a.mode, batch)
        bad = grad_diffs(g0, g1, names)
        if l1 != l0 or bad:
            rec["mismatches"].append({"run": i, "loss": l1, "grad_tensors_differing": len(bad), "first": bad[:4]})
    failed |= bool(rec["mismatches"])
    print(json.dumps(rec), flush=True)

    if a.traced:
        _, gt0, _, t0 = one_run(cfg, a.mode, batch, traced=True)
        rec = {"phase": "traced", "config": a.config, "mode": a.mode, "runs": a.traced, "trace_points": len(t0),
               "grad_equal_to_plain": bool(torch.equal(gt0, g0)), "first_differences": []}
        for i in range(1, a.traced + 1):
            l1, g1, _, t1 = one_run(cfg, a.mode, batch, traced=True)
            d = debug.first_difference(t0, t1)
            bad = grad_diffs(g0, g1, names)
            if d is not None or bad or l1 != l0:
                rec["first_differences"].append({"run": i, "trace_point": d, "loss": l1, "grad_tensors_differing": len(bad),
                                                 "first": bad[:4]})
        failed |= bool(rec["first_differences"]) or not rec["grad_equal_to_plain"]
        print(json.dumps(rec), flush=True)

    if a.poisoned:
        rec = {"phase": "poisoned", "config": a.config, "mode": a.mode, "runs": a.poisoned, "mismatches": []}
        for i in range(a.poisoned):
            debug.poison(pattern=0x7FC07FC0 if i % 2 == 0 else 0x7F7F7F7F)  # NaNs / huge finite values (3.4e38, 3.4e38 bf16)
            l1, g1, _, _ = one_run(cfg, a.mode, batch)
            bad = grad_diffs(g0, g1, names)
            if l1 != l0 or bad:
                rec["mismatches"].append({"run": i, "loss": l1, "grad_tensors_differing": len(bad), "first": bad[:6]})
        failed |= bool(rec["mismatches"])
        print(json.dumps(rec), flush=True)
    sys.exit(1 if (failed and not a.exit_zero) else 0)


if __name__ == "__main__":
    main()
