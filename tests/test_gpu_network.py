"""Whole-network and block parity on the GPU: HIP modules vs (a) the CPU oracle run on the same seeded inputs and
(b) the golden vectors captured from the reference itself.  fp32 storage must meet the north-star 1e-3 on logits and
every gradient; bf16 storage is held to the reference's own bf16 drift."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn as nn

import mednet_hip
from mednet_hip import nn as hnn
from mednet_hip.unet import components as HC
from mednet_hip.unet import loss as HL
from mednet_hip.unet import model as HM
from oracle import ref_cpu as O

from gpu_util import DEV, assert_close, rel
from test_oracle_golden import NETS

pytestmark = pytest.mark.gpu

# (logits, per-tensor gradient / cond) rel-L2.  bf16: measured on MI355X over all residual-network cases: logits <= 6.8e-3,
# per-tensor gradient / cond <= 1.84e-2, concatenated gradient <= 1.24e-2; the bounds are those times two (rounded up).
NET_TOL = {"fp32": (1e-3, 1e-3), "bf16": (1.5e-2, 4e-2)}
NET_TOL_BF16_CONCAT = 3e-2
# UNet3D ('gcr': GroupNorm -> conv -> ReLU, max-pooled ReLU maps) under bf16 STORAGE takes discrete decisions (ReLU masks,
# pooling arg-max) that flip against the fp32 reference near ties; a fraction f of gradient routed differently costs
# ~sqrt(2f) in rel-L2.  That is a property of bf16 storage, not of the kernels: the fp32 mode of the same code meets 1e-3,
# and test_unet3d_bf16_gradients_once_discrete_decisions_are_equalised shows the gradients agree at the residual network's
# tolerance once the oracle takes the same decisions.  Against the UNMODIFIED oracle the perf mode is therefore held to a
# sanity bound for this secondary model; the measured drift is printed (0.25 % .. 36 % depending on the case).
UNET3D_BF16_GRAD_TOL = 0.45
# Logits of the plain (non-residual) 14-conv UNet3D chain with bf16 activations AND bf16 matrix-core operands in every
# 3x3x3 layer (the 16-channel layers included): measured 3.1e-2 at 32^3, against 3e-2 allowed for the residual net.
UNET3D_BF16_LOGIT_TOL = 4e-2
HIP_CLS = {O.ResidualUNet3D: HM.ResidualUNet3D, O.UNet3D: HM.UNet3D}


def _hip_loss(lk, w, nh, logits, y, hm):
    wt = None if w is None else torch.tensor(w, dtype=torch.float32, device=DEV)
    if lk == "dice":
        return HL.DiceLoss(weight=wt).to(DEV)(logits, y)
    if lk == "ce":
        return HL.CrossEntropyLoss(weight=wt).to(DEV)(logits, y)
    kind = "L2" if lk == "ldmk" else "L1"
    return HL.DiceLoss(weight=wt).to(DEV)(logits[:, nh:], y) + HL.HeatmapRegressionLoss([0.015] * nh, kind).to(DEV)(logits[:, :nh], hm)


def _oracle_loss(lk, w, nh, logits, y, hm):
    wt = None if w is None else torch.tensor(w, dtype=torch.float32)
    if lk == "dice":
        return O.DiceLoss(weight=wt)(logits, y)
    if lk == "ce":
        return nn.CrossEntropyLoss(weight=wt)(logits, y)
    reg = nn.MSELoss() if lk == "ldmk" else nn.L1Loss()
    return O.landmark_loss(logits[:, nh:], logits[:, :nh], y, hm, O.DiceLoss(weight=wt), reg, [0.015] * nh)[0]


def _run_both(tag, mode, golden_dir):
    cls, ctor, ncls, nh, lk, w = NETS[tag]
    rec = np.load(os.path.join(golden_dir, tag + ".npz"))
    shape = tuple(int(v) for v in rec["meta.shape"])
    n = int(rec["meta.n"])
    batch = O.synthetic_batch(n, ctor["in_channels"], shape, ncls, nh, seed=int(rec["meta.seed"]))
    x = batch["data"].float()
    y = batch["label"][:, -1].long()
    hm = batch["label"][:, :-1].float() if nh else None
    ora = O.keyed_init_(cls(**ctor))
    lo = ora(x)
    loss_o = _oracle_loss(lk, w, nh, lo, y, hm)
    loss_o.backward()
    # the same oracle in fp64: tells how ill-conditioned each gradient is (how far fp32 rounding alone moves it)
    ora64 = O.keyed_init_(cls(**ctor)).double()
    l64 = ora64(x.double())
    wt64 = None if w is None else torch.tensor(w, dtype=torch.float64)
    if lk == "dice":
        loss64 = O.DiceLoss(weight=wt64)(l64, y)
    elif lk == "ce":
        loss64 = nn.CrossEntropyLoss(weight=wt64)(l64, y)
    else:
        reg = nn.MSELoss() if lk == "ldmk" else nn.L1Loss()
        loss64 = O.landmark_loss(l64[:, nh:], l64[:, :nh], y, hm.double(), O.DiceLoss(weight=wt64), reg, [0.015] * nh)[0]
    loss64.backward()
    ora.cond = {k: max(1.0, rel(p.grad, q.grad) / 1e-6) for (k, p), (_, q) in
                zip(ora.named_parameters(), ora64.named_parameters())}
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HIP_CLS[cls](**ctor)).to(DEV)
        lg = net(x.to(DEV))
        assert lg.dtype == torch.float32 and lg.is_contiguous() and lg.shape == lo.shape
        loss_g = _hip_loss(lk, w, nh, lg, y.to(DEV), None if hm is None else hm.to(DEV))
        loss_g.backward()
    return rec, ora, lo, loss_o, net, lg, loss_g


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["res_cfg1", "res_cfg1_ce", "res_small", "res_odd", "res_cfg2_32", "res_cfg4_32",
                                 "res_ldmk_l1", "unet_cfg1", "unet_small", "unet_oddsize", "unet_cfg2_32"])
def test_network_parity(tag, mode, golden_dir):
    rec, ora, lo, loss_o, net, lg, loss_g = _run_both(tag, mode, golden_dir)
    tl, tg = NET_TOL[mode]
    if mode == "bf16" and tag.startswith("unet"):
        tl = UNET3D_BF16_LOGIT_TOL
    report = {"logits": assert_close(lg, lo, tl, f"{tag} logits")}
    assert abs(float(loss_g) - float(loss_o)) <= (1e-4 if mode == "fp32" else 2e-2) * max(1.0, abs(float(loss_o)))
    grads_o = dict(ora.named_parameters())
    worst = 0.0
    num = den = 0.0
    for name, p in net.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, name
        go = grads_o[name].grad
        r = rel(p.grad, go)
        if ora.cond[name] <= 10.0:  # the concatenated metric covers the well-conditioned tensors
            num += float((p.grad.detach().cpu().double() - go.double()).pow(2).sum())
            den += float(go.double().pow(2).sum())
        if mode == "fp32":
            # a gradient that fp32 rounding alone moves by cond*1e-6 (sums with heavy cancellation, e.g. the affine of
            # a 1-channel GroupNorm over the whole volume) cannot be held tighter than that
            lim = max(tg, ora.cond[name] * 3e-6)
        else:
            # bf16 storage: every activation/gradient element carries 2^-9 relative noise and ReLU/max-pool masks flip
            # near ties; a tensor is held to the per-tensor bound only if it averages enough terms to beat that noise
            # (the reference's own bf16 run drifts 1.8e-2 on the concatenated gradient); all tensors count
            # in the concatenated-gradient metric below.
            lim = tg * ora.cond[name] if p.numel() >= 1024 else float("inf")
            if tag.startswith("unet"):
                lim = float("inf")
        worst = max(worst, r / (lim / tg)) if lim != float("inf") else worst
        assert r <= lim, f"{tag} grad {name}: rel-L2 {r:.3e} > {lim:.1e} (cond {ora.cond[name]:.1f})"
    total = (num / den) ** 0.5
    glob_tol = 1e-3 if mode == "fp32" else (UNET3D_BF16_GRAD_TOL if tag.startswith("unet") else NET_TOL_BF16_CONCAT)
    assert total <= glob_tol, f"{tag}: concatenated-gradient rel-L2 {total:.3e} > {glob_tol:.1e}"
    report["grads"] = total
    # and against what the REFERENCE produced (golden): loss + logits
    if mode == "fp32":
        assert abs(float(loss_g) - float(rec["loss"])) <= 1e-4 * max(1.0, abs(float(rec["loss"])))
        if "logits.full" in rec.files:
            assert_close(lg, torch.from_numpy(rec["logits.full"]), tl, f"{tag} logits vs golden")
    print(f"[parity] {tag} {mode}: logits {report['logits']:.2e} concatenated-grad {report['grads']:.2e} "
          f"worst-vs-limit {worst:.2e}")


def test_cfg2_128_against_reference_golden(golden_dir):
    """BASELINE config 2's model on a full 128^3 patch (N=1), fp32 storage, against the vectors captured from the
    reference: strided logits, loss, and norm / keyed projection of every gradient."""
    rec = np.load(os.path.join(golden_dir, "res_cfg2_128.npz"))
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = O.synthetic_batch(1, 1, (128, 128, 128), 4, 0, seed=int(rec["meta.seed"]))
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        lg = net(batch["data"].float().to(DEV))
        loss = HL.DiceLoss(weight=torch.tensor([0.05, 1, 1, 1.0], device=DEV)).to(DEV)(lg, batch["label"][:, -1].long().to(DEV))
        loss.backward()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), 1e-3, "strided logits")
    assert abs(float(loss) - float(rec["loss"])) <= 1e-4
    assert abs(float(lg.double().norm()) - float(rec["logits.norm"])) <= 1e-3 * float(rec["logits.norm"])
    wn = wp = wh = 0.0
    for name, p in net.named_parameters():
        g = p.grad.detach().double().cpu().numpy().reshape(-1)
        norm = float(rec[f"grad.{name}.norm"])
        dn = abs(np.sqrt((g * g).sum()) - norm) / norm
        assert dn <= 1e-3, name
        pv = np.random.Generator(np.random.PCG64(zlib.crc32(("proj:" + name).encode()))).standard_normal(g.size)
        # <g - g_ref, r> with r ~ N(0, I) has standard deviation ||g - g_ref||: a 1e-3 relative error moves the projection by
        # ~1e-3 * norm (one sigma).  Measured 2.9e-4 (round 3 .. 6): held to the north star's 1e-3, as the leading elements are.
        dp = abs(g @ pv - float(rec[f"grad.{name}.proj"])) / norm
        assert dp <= 1e-3, name
        head = rec[f"grad.{name}.head"]
        dh = np.linalg.norm(g[: head.size] - head) / max(np.linalg.norm(head), 1e-3 * norm / np.sqrt(g.size) * 8)
        assert dh <= 1e-3, name
        wn, wp, wh = max(wn, dn), max(wp, dp), max(wh, dh)
    # (measured on an MI355X in the split-bf16 fp32 mode, round 3: strided logits 1.05e-5, loss diff 0, worst gradient-norm diff
    #  2.6e-4, worst projection diff 2.9e-4, worst leading-elements rel-L2 3.1e-4; cfg4: 9.4e-6 / 2.5e-4 / 7.7e-5)
    print(f"[cfg2 128^3 fp32 mode vs reference] strided logits {rl:.2e} (tol 1e-3)  loss diff {abs(float(loss) - float(rec['loss'])):.1e}"
          f"  worst gradient-norm diff {wn:.2e} (tol 1e-3)  worst projection diff {wp:.2e} (tol 1e-3)  worst leading-elements rel-L2 {wh:.2e} (tol 1e-3)")


def test_cfg4_128_landmark_against_reference_golden(golden_dir):
    """BASELINE config 4 (landmark path: 16 heat maps + 2 classes, landmarks.py:66-83,125-134) on a full 128^3 patch, fp32
    storage, against the reference's golden vectors: strided logits, the three losses, norm / projection of every gradient."""
    from mednet_hip.train import LandmarkStep
    rec = np.load(os.path.join(golden_dir, "res_cfg4_128.npz"))
    ctor = dict(in_channels=1, out_channels=18, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = O.synthetic_batch(1, 1, (128, 128, 128), 2, 16, seed=int(rec["meta.seed"]))
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = LandmarkStep(net, [0.05, 1.0], [0.015] * 16, "L2")
        lg = net(batch["data"].float().to(DEV))
        tot, cl, rg = step._fwd_bwd({k: v.to(DEV) for k, v in batch.items()})
        torch.cuda.synchronize()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), 1e-3, "strided logits")
    for got, key in ((tot, "loss"), (cl, "class_loss"), (rg, "regression_loss")):
        assert abs(float(got) - float(rec[key])) <= 1e-4 * max(1.0, abs(float(rec[key]))), key
    step.flat.grads_as_attr()
    wn = wp = 0.0
    for name, p in net.named_parameters():
        g = p.grad.detach().double().cpu().numpy().reshape(-1)
        norm = float(rec[f"grad.{name}.norm"])
        dn = abs(np.sqrt((g * g).sum()) - norm) / norm
        assert dn <= 1e-3, name
        pv = np.random.Generator(np.random.PCG64(zlib.crc32(("proj:" + name).encode()))).standard_normal(g.size)
        dp = abs(g @ pv - float(rec[f"grad.{name}.proj"])) / norm
        assert dp <= 1e-3, name  # (measured 7.7e-5)
        wn, wp = max(wn, dn), max(wp, dp)
    step.flat.release()
    print(f"[cfg4 128^3 fp32 mode vs reference] strided logits {rl:.2e} (tol 1e-3)  worst gradient-norm diff {wn:.2e} (tol 1e-3)"
          f"  worst projection diff {wp:.2e} (tol 1e-3)")


# ---- the BENCHMARKED path (bf16 storage: persistent matrix-core kernels with statistics accumulated over a workgroup's
# bricks, GroupNorm-backward sums from the data-gradient epilogue, weight gradients on the side stream, flat gradient
# buffers) at the size it is timed at.  Tolerances: what was measured on an MI355X (printed below) times two; the reference's
# own bf16 drift at 32^3 is 8.3e-3 (logits) / 1.8e-2 (gradients), SURVEY F7.
# measured (round 2, cfg2 / cfg4): strided logits 6.7e-3 / 6.2e-3, loss diff 1.2e-5, gradient norms 6.3e-3 / 3.4e-3,
# projections 3.0e-2 / 1.9e-2
BF16_128_LOGITS = 1.5e-2    # strided logits, rel-L2
BF16_128_LOSS = 1e-3        # |loss - golden| (relative to max(1, |golden|))
BF16_128_GRAD_NORM = 1.5e-2 # | ||g|| - ||g_ref|| | / ||g_ref||  per tensor
BF16_128_GRAD_PROJ = 6e-2   # |<g - g_ref, r>| / ||g_ref||, r ~ N(0, I): a relative error e moves it by ~e (one sigma)


def _check_grads_against_golden_summaries(net, rec, norm_tol, proj_tol, what):
    worst_n = worst_p = 0.0
    for name, p in net.named_parameters():
        g = p.grad.detach().double().cpu().numpy().reshape(-1)
        assert np.isfinite(g).all(), name
        norm = float(rec[f"grad.{name}.norm"])
        dn = abs(np.sqrt((g * g).sum()) - norm) / norm
        pv = np.random.Generator(np.random.PCG64(zlib.crc32(("proj:" + name).encode()))).standard_normal(g.size)
        dp = abs(g @ pv - float(rec[f"grad.{name}.proj"])) / norm
        worst_n, worst_p = max(worst_n, dn), max(worst_p, dp)
        assert dn <= norm_tol, f"{what} {name}: gradient norm off by {dn:.3e} > {norm_tol:.1e}"
        assert dp <= proj_tol, f"{what} {name}: gradient projection off by {dp:.3e} > {proj_tol:.1e}"
    return worst_n, worst_p


def test_cfg2_128_bf16_timed_path_against_reference_golden(golden_dir):
    """BASELINE config 2's model on a full 128^3 patch through train.SegmentationStep in bf16 mode -- the code path bench.py
    times -- against the reference's golden vectors (model.py:189-214 forward, DiceLoss, every gradient)."""
    from mednet_hip.train import SegmentationStep
    rec = np.load(os.path.join(golden_dir, "res_cfg2_128.npz"))
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(1, 1, (128, 128, 128), 4, 0, seed=int(rec["meta.seed"])).items()}
    with mednet_hip.precision("bf16"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        with torch.no_grad():
            lg = net(batch["data"].float())
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), BF16_128_LOGITS, "strided logits (bf16)")
    dl = abs(float(loss) - float(rec["loss"]))
    assert dl <= BF16_128_LOSS, dl
    step.flat.grads_as_attr()
    wn, wp = _check_grads_against_golden_summaries(net, rec, BF16_128_GRAD_NORM, BF16_128_GRAD_PROJ, "cfg2 128^3 bf16")
    print(f"[parity-128 bf16] cfg2: strided logits {rl:.2e}  loss diff {dl:.2e}  worst grad-norm diff {wn:.2e}  worst projection diff {wp:.2e}")
    step.flat.release()


def test_cfg4_128_bf16_timed_path_against_reference_golden(golden_dir):
    """BASELINE config 4 (landmarks.py:66-83,125-134: 16 heat maps + 2 classes) at 128^3 through train.LandmarkStep in bf16
    mode against the reference's golden vectors."""
    from mednet_hip.train import LandmarkStep
    rec = np.load(os.path.join(golden_dir, "res_cfg4_128.npz"))
    ctor = dict(in_channels=1, out_channels=18, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(1, 1, (128, 128, 128), 2, 16, seed=int(rec["meta.seed"])).items()}
    with mednet_hip.precision("bf16"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = LandmarkStep(net, [0.05, 1.0], [0.015] * 16, "L2")
        with torch.no_grad():
            lg = net(batch["data"].float())
        tot, cl, rg = step._fwd_bwd(batch)
        torch.cuda.synchronize()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), BF16_128_LOGITS, "strided logits (bf16)")
    for got, key in ((tot, "loss"), (cl, "class_loss"), (rg, "regression_loss")):
        assert abs(float(got) - float(rec[key])) <= BF16_128_LOSS * max(1.0, abs(float(rec[key]))), key
    step.flat.grads_as_attr()
    wn, wp = _check_grads_against_golden_summaries(net, rec, BF16_128_GRAD_NORM, BF16_128_GRAD_PROJ, "cfg4 128^3 bf16")
    print(f"[parity-128 bf16] cfg4: strided logits {rl:.2e}  worst grad-norm diff {wn:.2e}  worst projection diff {wp:.2e}")
    step.flat.release()


# fp16 storage (BASELINE config 5).  Measured on MI355X (round 2): strided logits 7.0e-4, gradient norms 4.1e-4, projections
# 6.3e-3; the bounds are those times two.
FP16_LOGITS = 1.5e-3
FP16_GRAD_NORM = 1e-3
FP16_GRAD_PROJ = 1.3e-2


def test_cfg5_fp16_with_loss_scaling_against_reference_golden(golden_dir):
    """BASELINE config 5's topology (5 levels, 64 .. 1024 channels) in fp16 storage with the device-side dynamic loss scaler,
    through train.SegmentationStep, at 32x32x16 against the reference's golden vectors (res_cfg5_small.npz: strided logits,
    loss, norm and keyed projection of every gradient).  Gradients are compared after dividing by the scale the step used;
    without scaling the reference's own fp16 run loses a third of the gradient to underflow -- the unscaled
    run below must show that loss too, otherwise the test would not be testing the scaler."""
    from mednet_hip.train import SegmentationStep
    rec = np.load(os.path.join(golden_dir, "res_cfg5_small.npz"))
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[64, 128, 256, 512, 1024])
    shape = tuple(int(v) for v in rec["meta.shape"])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(int(rec["meta.n"]), 1, shape, 4, 0, seed=int(rec["meta.seed"])).items()}
    with mednet_hip.precision("fp16"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        assert step.scaler is not None
        with torch.no_grad():
            lg = net(batch["data"].float())
        scale0 = step.scaler.snapshot()[0]
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
        s = int(rec["meta.stride"])
        rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), FP16_LOGITS, "strided logits (fp16)")
        assert abs(float(loss) - float(rec["loss"])) <= 1e-3
        assert bool(torch.isfinite(step.flat.grad).all())
        step.flat.grad.div_(scale0)
        step.flat.grads_as_attr()
        wn, wp = _check_grads_against_golden_summaries(net, rec, FP16_GRAD_NORM, FP16_GRAD_PROJ, "cfg5 small fp16")
        # the same backward WITHOUT the scaler: underflow must be visible (this is what the scaler is for)
        step.flat.grad.zero_()
        loss2 = step.loss(net(batch["data"].float()), batch["label"][:, -1].long())
        loss2.backward()
        from mednet_hip.train import finish_backward
        finish_backward()
        torch.cuda.synchronize()
        lost = 0
        for name, p in net.named_parameters():
            g = p.grad.detach().double().cpu().numpy().reshape(-1)
            norm = float(rec[f"grad.{name}.norm"])
            if abs(np.sqrt((g * g).sum()) - norm) > 0.1 * norm:
                lost += 1
        step.flat.release()
    print(f"[fp16 cfg5 small] strided logits {rl:.2e}  worst grad-norm diff {wn:.2e}  worst projection diff {wp:.2e}; "
          f"without loss scaling {lost} of {len(list(net.parameters()))} gradient tensors are off by more than 10 %")
    assert lost > 0


def test_loss_scaler_skips_overflowing_steps_and_recovers():
    """The device-side scaler (mednet_adam_step_scaled): a step whose gradients overflow leaves parameters, Adam moments and
    the step count untouched and halves the scale; clean steps count up and double it after `growth_interval`."""
    from mednet_hip.train import FlatParams, FlatAdam, LossScaler
    lin = torch.nn.Linear(64, 64).to(DEV)
    flat = FlatParams(lin)
    opt = FlatAdam(flat, lr=1e-2)
    sc = LossScaler(DEV, init_scale=1024.0, growth_interval=3)
    ref = torch.nn.Linear(64, 64).to(DEV)
    ref.load_state_dict(lin.state_dict())
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    g = torch.Generator(device="cpu").manual_seed(0)
    for it in range(8):
        true_grad = torch.randn(flat.total, generator=g).to(DEV) * 1e-3
        flat.grad.copy_(true_grad * sc.state[0])
        overflow = it in (2, 5)
        if overflow:
            flat.grad[17] = float("inf") if it == 2 else float("nan")
        before = (flat.flat.clone(), opt.m.clone(), opt.v.clone(), sc.snapshot())
        opt.step_scaled(sc)
        after = sc.snapshot()
        if overflow:
            assert torch.equal(flat.flat, before[0]) and torch.equal(opt.m, before[1]) and torch.equal(opt.v, before[2])
            assert after[0] == before[3][0] * 0.5 and after[1] == 0 and after[2] == before[3][2] and after[3] == 0
        else:
            for p, off in zip(ref.parameters(), flat.offsets):
                p.grad = true_grad[off:off + p.numel()].view_as(p).clone()
            ropt.step()
            assert after[2] == before[3][2] + 1 and after[3] == 0
    for (k, a), (_, b) in zip(lin.named_parameters(), ref.named_parameters()):
        assert_close(a, b, 1e-5, f"params after scaled Adam steps: {k}")
    # 6 clean steps with interval 3, interleaved with 2 backoffs: 1024 -> (2 clean) -> /2 -> (2 clean) -> /2 -> (2 clean) ...
    assert sc.snapshot()[0] in (256.0, 512.0, 1024.0)
    flat.release()


def test_cfg2_128_fp16_split_weights_against_reference_golden(golden_dir):
    """fp16x2 (round 6: fp16 storage + split weights, MEDNET_ALGO_SPLITW_BIT) against the REFERENCE's golden vectors of BASELINE config
    2 at 128^3 (N = 1): the 16-bit-storage mode that is held to the north star's 1e-3 -- strided logits, loss, every gradient
    tensor's norm, and the random projections at the fp32 mode's former bound.  (Full tensors at the timed batch:
    tests/test_gpu_timed_workload.py[fp16x2].)"""
    from mednet_hip.train import SegmentationStep
    rec = np.load(os.path.join(golden_dir, "res_cfg2_128.npz"))
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(1, 1, (128, 128, 128), 4, 0, seed=int(rec["meta.seed"])).items()}
    with mednet_hip.precision("fp16x2"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        with torch.no_grad():
            lg = net(batch["data"].float())
        scale = step.scaler.snapshot()[0]
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), 1e-3, "strided logits (fp16x2)")
    dl = abs(float(loss) - float(rec["loss"]))
    assert dl <= 1e-4, dl
    step.flat.grad.div_(scale)
    step.flat.grads_as_attr()
    wn, wp = _check_grads_against_golden_summaries(net, rec, 1e-3, 4e-3, "cfg2 128^3 fp16x2")
    step.flat.release()
    print(f"[cfg2 128^3 fp16x2 vs reference] strided logits {rl:.2e} (tol 1e-3)  loss diff {dl:.1e}  worst gradient-norm diff {wn:.2e} "
          f"(tol 1e-3)  worst projection diff {wp:.2e} (tol 4e-3)")


def test_cfg2_128_fp16_storage_against_reference_golden(golden_dir):
    """BASELINE config 2's model at 128^3 through train.SegmentationStep in fp16 storage with the device-side loss scaler -- the
    same kernels as the benchmarked bf16 mode, instantiated for the other 16-bit type -- against the reference's golden vectors.
    fp16 carries 3 more mantissa bits than bf16: this mode is held to the north-star 1e-3 on logits and gradient norms at the
    16-bit modes' speed (bench.py reports its rate as `fp16_mode`); the random projections (which see the whole per-tensor
    error vector) are held to twice what was measured."""
    from mednet_hip.train import SegmentationStep
    rec = np.load(os.path.join(golden_dir, "res_cfg2_128.npz"))
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(1, 1, (128, 128, 128), 4, 0, seed=int(rec["meta.seed"])).items()}
    with mednet_hip.precision("fp16"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        with torch.no_grad():
            lg = net(batch["data"].float())
        scale = step.scaler.snapshot()[0]
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), 1e-3, "strided logits (fp16)")
    dl = abs(float(loss) - float(rec["loss"]))
    assert dl <= 1e-4, dl
    assert bool(torch.isfinite(step.flat.grad).all())
    step.flat.grad.div_(scale)  # (the gradients carry the loss scale)
    step.flat.grads_as_attr()
    # measured on an MI355X (round 5): strided logits 8.7e-4, loss diff 3e-7, gradient norms 4.2e-4, projections 3.6e-3, leading
    # elements: median tensor 8.9e-4, worst (decoders.1 conv2 weight) 1.55e-2
    wn, wp = _check_grads_against_golden_summaries(net, rec, 1e-3, 8e-3, "cfg2 128^3 fp16")
    heads = []
    for name, p in net.named_parameters():
        g = p.grad.detach().double().cpu().numpy().reshape(-1)
        head = rec[f"grad.{name}.head"]
        norm = float(rec[f"grad.{name}.norm"])
        heads.append((np.linalg.norm(g[: head.size] - head) / max(np.linalg.norm(head), 1e-3 * norm / np.sqrt(g.size) * 8), name))
    heads.sort(reverse=True)
    wh = heads[0][0]
    print("[cfg2 128^3 fp16] leading-elements rel-L2, worst five:", ", ".join(f"{k} {v:.2e}" for v, k in heads[:5]),
          " median", f"{heads[len(heads) // 2][0]:.2e}")
    assert wh <= 4e-2, heads[0]  # (element-wise error of the 16-bit stored gradients; the bf16 mode is not held to this at all)
    step.flat.release()
    print(f"[cfg2 128^3 fp16 storage vs reference] strided logits {rl:.2e} (tol 1e-3)  loss diff {dl:.1e}  worst gradient-norm diff "
          f"{wn:.2e} (tol 1e-3)  worst projection diff {wp:.2e} (tol 8e-3)  worst leading-elements rel-L2 {wh:.2e}")


# cfg5 at full size against the reference (tests/golden/res_cfg5_full.npz: the reference's ResidualUNet3D
# [64 .. 1024] on one 160 x 160 x 96 patch, fp32 on the CPU, bit-equal to the oracle; tools/make_golden.py).
# fp32 storage and fp16x2 (fp16 storage + split weights): the north-star 1e-3.  16-bit storage: the bounds of the 128^3 tests of the same modes.
@pytest.mark.parametrize("mode", ["fp32", "fp16x2", "fp16", "bf16"])
def test_cfg5_full_size_against_reference_golden(mode, golden_dir):
    from mednet_hip.train import SegmentationStep
    rec = np.load(os.path.join(golden_dir, "res_cfg5_full.npz"))
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[64, 128, 256, 512, 1024])
    shape = tuple(int(v) for v in rec["meta.shape"])
    assert shape == (160, 160, 96) and int(rec["meta.n"]) == 1
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(1, 1, shape, 4, 0, seed=int(rec["meta.seed"])).items()}
    logit_tol, loss_tol, norm_tol, proj_tol = {"fp32": (1e-3, 1e-4, 1e-3, 4e-3),
                                               "fp16x2": (1e-3, 1e-4, 1e-3, 4e-3),  # (fp16 storage + split weights: the fp32 mode's bounds)
                                               "fp16": (FP16_LOGITS, 1e-3, FP16_GRAD_NORM, FP16_GRAD_PROJ),
                                               "bf16": (BF16_128_LOGITS, BF16_128_LOSS, BF16_128_GRAD_NORM, BF16_128_GRAD_PROJ)}[mode]
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        with torch.no_grad():
            lg = net(batch["data"].float())
        scale = step.scaler.snapshot()[0] if step.scaler is not None else 1.0
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
    s = int(rec["meta.stride"])
    rl = assert_close(lg[..., ::s, ::s, ::s], torch.from_numpy(rec["logits.strided"]), logit_tol, f"strided logits ({mode})")
    dl = abs(float(loss) - float(rec["loss"]))
    assert dl <= loss_tol, (float(loss), float(rec["loss"]))
    assert abs(float(lg.double().norm()) - float(rec["logits.norm"])) <= logit_tol * float(rec["logits.norm"])
    assert bool(torch.isfinite(step.flat.grad).all())
    if scale != 1.0:  # (the fp16 gradients carry the loss scale)
        step.flat.grad.div_(scale)
    step.flat.grads_as_attr()
    wn, wp = _check_grads_against_golden_summaries(net, rec, norm_tol, proj_tol, f"cfg5 full size {mode}")
    step.flat.release()
    print(f"[cfg5 160x160x96 {mode} vs reference] strided logits {rl:.2e} (tol {logit_tol:.1e})  loss diff {dl:.1e}  worst "
          f"gradient-norm diff {wn:.2e} (tol {norm_tol:.1e})  worst projection diff {wp:.2e} (tol {proj_tol:.1e})")


class _ForcedReLU(nn.Module):
    """Test-only stand-in for the oracle's nn.ReLU: value and mask taken from what the HIP path stored for this layer, so
    the oracle takes the same discrete decisions; the gradient still flows to the oracle's own pre-activation."""

    def __init__(self, stored):
        super().__init__()
        self.stored = stored

    def forward(self, pre):
        mask = (self.stored > 0).to(pre.dtype)
        return pre * mask + (self.stored - pre * mask).detach()


@pytest.mark.parametrize("tag", ["unet_small", "unet_oddsize", "unet_cfg2_32"])
def test_unet3d_bf16_gradients_once_discrete_decisions_are_equalised(tag, golden_dir):
    """Why UNet3D's bf16 gradients are only held to a sanity bound in test_network_parity.  UNet3D ('gcr': GroupNorm -> conv
    -> ReLU, max-pooled ReLU maps) takes a DISCRETE decision at every voxel of every layer -- the ReLU mask, and the arg-max
    of each 2x2x2 pooling window.  Under bf16 storage some of them flip against the fp32 reference (a conv output within
    rounding noise of zero; two window entries rounded to the same 8-bit mantissa: the first one wins, as in ATen), and a
    fraction f of gradient routed differently costs ~sqrt(2f) in rel-L2 whatever the kernels do.  Demonstration: the oracle
    is given the HIP path's stored ReLU outputs (values and masks; a level's last one is the pooling input), so both sides
    decide identically, while every GroupNorm, convolution, interpolation/concat and loss in between is still the oracle's own
    fp32 autograd.  The gradients then agree at the residual network's bf16 tolerance, per tensor and concatenated -- i.e.
    the whole 'gcr' backward chain of the HIP path is checked at network level."""
    cls, ctor, ncls, nh, lk, w = NETS[tag]
    rec = np.load(os.path.join(golden_dir, tag + ".npz"))
    shape = tuple(int(v) for v in rec["meta.shape"])
    batch = O.synthetic_batch(int(rec["meta.n"]), ctor["in_channels"], shape, ncls, nh, seed=int(rec["meta.seed"]))
    x, y = batch["data"].float(), batch["label"][:, -1].long()
    stored = {}
    with mednet_hip.precision("bf16"):
        net = O.keyed_init_(HM.UNet3D(**ctor)).to(DEV)
        hooks = []
        for name, m in net.named_modules():
            if isinstance(m, HC.SingleConv):
                hooks.append(m.register_forward_hook(
                    lambda mod, i, o, name=name: stored.__setitem__(name, (o[0] if isinstance(o, tuple) else o).detach().float().cpu())))
        lg = net(x.to(DEV))
        _hip_loss(lk, w, nh, lg, y.to(DEV), None).backward()
        for h in hooks:
            h.remove()
    ora = O.keyed_init_(cls(**ctor))
    swapped = 0
    for name, m in ora.named_modules():
        if isinstance(m, O.SingleConv):
            assert list(m._modules)[-1] == "ReLU" and name in stored, name
            m.ReLU = _ForcedReLU(stored[name])
            swapped += 1
    assert swapped == len(stored) > 0
    _oracle_loss(lk, w, nh, ora(x), y, None).backward()
    num = den = 0.0
    worst = 0.0
    for (name, p), (_, q) in zip(net.named_parameters(), ora.named_parameters()):
        r = rel(p.grad, q.grad)
        num += float((p.grad.detach().cpu().double() - q.grad.double()).pow(2).sum())
        den += float(q.grad.double().pow(2).sum())
        if p.numel() >= 1024:
            worst = max(worst, r)
            assert r <= NET_TOL["bf16"][1], f"{tag} {name}: {r:.3e}"
    total = (num / den) ** 0.5
    print(f"[unet3d decisions equalised] {tag}: concatenated-gradient rel-L2 {total:.2e}, worst tensor (>=1024 elements) {worst:.2e}")
    assert total <= NET_TOL_BF16_CONCAT, total


def test_frozen_checkpoint_model_predicts_like_the_oracle(tmp_path):
    """Row N4 (examples/predict.py:47-50,77-87): a PL-0.9-shaped checkpoint -> `load_from_checkpoint` -> `freeze()` ->
    `.to(device)`, `eval`, `no_grad` forward; logits within the fp32-mode tolerance of the oracle that wrote the weights."""
    from test_host_logic import _pl09_checkpoint, _segmentation_net_like_the_reference
    path = tmp_path / "epoch=3.ckpt"
    ora, hp = _pl09_checkpoint(path, [8, 16])
    model = _segmentation_net_like_the_reference().load_from_checkpoint(str(path))
    model.freeze()
    x = O.synthetic_batch(2, 1, (16, 24, 32), 2, 0, seed=3)["data"].float()
    with mednet_hip.precision("fp32"), torch.no_grad():
        model = model.to(DEV)
        model.eval()
        lg = model(x.to(DEV))
    assert not lg.requires_grad
    assert_close(lg, ora(x), 1e-3, "logits of the frozen checkpoint model")


def test_training_steps_track_oracle(golden_dir):
    """segmentation.py:58-65 + :119-120: three Adam steps on cfg1; parameters must track the oracle's."""
    ctor = dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8])
    batch = O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234)
    w = [0.05, 1.0]
    ora = O.keyed_init_(O.ResidualUNet3D(**ctor))
    opt_o = torch.optim.Adam(ora.parameters(), lr=1e-3)
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        opt_g = torch.optim.Adam(net.parameters(), lr=1e-3)
        crit = HL.DiceLoss(weight=torch.tensor(w, device=DEV)).to(DEV)
        gb = {k: v.to(DEV) for k, v in batch.items()}
        for step in range(3):
            opt_o.zero_grad()
            lo = O.seg_training_step(ora, O.DiceLoss(weight=torch.tensor(w)), batch)
            lo.backward()
            opt_o.step()
            opt_g.zero_grad()
            lg = crit(net(gb["data"].float()), gb["label"][:, -1].long())
            lg.backward()
            opt_g.step()
            assert abs(float(lg) - float(lo)) <= 2e-4, (step, float(lg), float(lo))
    for (k, a), (_, b) in zip(net.named_parameters(), ora.named_parameters()):
        assert rel(a, b) <= 2e-3, k


def test_blocks_against_reference_golden(golden_dir):
    rec = np.load(os.path.join(golden_dir, "blocks.npz"))

    def rnd(tag, *shape):
        return torch.from_numpy(O._rng("in:" + tag).standard_normal(shape).astype(np.float32))

    E, D = HC.ExtResNetBlock, HC.DoubleConv
    cases = {f"single_{o}": (lambda o=o: HC.SingleConv(8, 16, 3, o, 8), [("single_" + o, (2, 8, 6, 10, 12))])
             for o in ["cge", "gcr", "cg", "cr", "cl", "ce", "crg", "bcr", "cbe"]}  # b: stock nn.BatchNorm3d between HIP ops
    cases.update({
        "single_cge_c4": (lambda: HC.SingleConv(4, 4, 3, "cge", 8), [("single_cge_c4", (1, 4, 5, 6, 7))]),
        "double_enc_gcr": (lambda: D(8, 32, True, 3, "gcr", 8), [("double_enc", (1, 8, 8, 8, 8))]),
        "double_dec_gcr": (lambda: D(24, 8, False, 3, "gcr", 8), [("double_dec", (1, 24, 8, 8, 8))]),
        "resblock_cge": (lambda: E(8, 16, order="cge"), [("resblock_cge", (2, 8, 6, 8, 10))]),
        "resblock_cgr": (lambda: E(8, 16, order="cgr"), [("resblock_cgr", (2, 8, 6, 8, 10))]),
        "resblock_cgl": (lambda: E(8, 16, order="cgl"), [("resblock_cgl", (2, 8, 6, 8, 10))]),
        "encoder_res": (lambda: HC.Encoder(8, 16, basic_module=E, conv_layer_order="cge"), [("encoder_res", (2, 8, 8, 12, 10))]),
        "encoder_res_oddpool": (lambda: HC.Encoder(8, 8, basic_module=E, conv_layer_order="cge"), [("encoder_odd", (1, 8, 7, 9, 11))]),
        "encoder_double": (lambda: HC.Encoder(8, 16, basic_module=D, conv_layer_order="gcr"), [("encoder_double", (1, 8, 8, 8, 8))]),
        "decoder_res": (lambda: HC.Decoder(16, 8, basic_module=E, conv_layer_order="cge"),
                        [("decoder_res_e", (2, 8, 8, 12, 10)), ("decoder_res_x", (2, 16, 4, 6, 5))]),
        "decoder_double": (lambda: HC.Decoder(24, 8, basic_module=D, conv_layer_order="gcr"),
                           [("decoder_double_e", (1, 8, 7, 9, 10)), ("decoder_double_x", (1, 16, 3, 4, 5))]),
    })
    with mednet_hip.precision("fp32"):
        for tag, (make, ins) in cases.items():
            m = O.keyed_init_(make()).to(DEV)
            xs = [rnd(t, *s).to(DEV).requires_grad_(True) for t, s in ins]
            y = m(*xs)
            g = torch.from_numpy(O._rng("cot:" + tag).standard_normal(tuple(y.shape)).astype(np.float32)).to(DEV)
            (y.float() * g).sum().backward()
            assert_close(y, torch.from_numpy(rec[f"{tag}.y"]), 2e-4, f"{tag}.y")
            for i, t in enumerate(xs):
                assert_close(t.grad, torch.from_numpy(rec[f"{tag}.dx{i}"]), 5e-4, f"{tag}.dx{i}")
            for k, p in m.named_parameters():
                assert_close(p.grad, torch.from_numpy(rec[f"{tag}.dp.{k}"]), 5e-4, f"{tag}.dp.{k}")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("order", ["bcr", "cbe"])
def test_batchnorm_orders_in_the_16_bit_modes(mode, order):
    """Order char 'b' (components.py:58-63): a stock nn.BatchNorm3d fed by / feeding the HIP conv in 16-bit channels-last
    storage -- the place where a silent layout or dtype slip would hide.  Against the oracle on 16-bit-representable inputs."""
    half = torch.bfloat16 if mode == "bf16" else torch.float16
    x = torch.from_numpy(O._rng(f"bn16{order}").standard_normal((2, 16, 6, 10, 12)).astype(np.float32)).to(half).float()
    ora = O.keyed_init_(O.SingleConv(16, 32, 3, order, 8))
    xo = x.clone().requires_grad_(True)
    yo = ora(xo)
    g = torch.from_numpy(O._rng("bn16cot").standard_normal(tuple(yo.shape)).astype(np.float32))
    (yo * g).sum().backward()
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HC.SingleConv(16, 32, 3, order, 8)).to(DEV)
        xg = x.to(DEV).requires_grad_(True)
        yg = net(xg)
        assert tuple(yg.shape) == tuple(yo.shape)
        (yg.float() * g.to(DEV)).sum().backward()
    tol_y, tol_g = (1.5e-2, 3e-2) if mode == "bf16" else (2e-3, 4e-3)
    assert_close(yg, yo, tol_y, f"{order} y")
    assert_close(xg.grad, xo.grad, tol_g, f"{order} dx")
    for (k, a), (_, b) in zip(net.named_parameters(), ora.named_parameters()):
        assert_close(a.grad, b.grad, tol_g, f"{order} d{k}")
    for (k, a), (_, b) in zip(net.named_buffers(), ora.named_buffers()):  # running statistics moved the same way
        if a.dtype.is_floating_point:
            assert_close(a, b, tol_y, f"{order} buffer {k}")


@pytest.mark.parametrize("groups", [8, 16, 32])
def test_fused_groupnorm_partials_any_group_size(groups):
    """bf16 mode: the conv epilogue's GroupNorm partials are per channel PAIR; with 32 channels that is exact for 8 and
    16 groups (4 / 2 channels per group) and must NOT be used for 32 groups (1 channel per group: stand-alone statistics
    pass).  SingleConv 'cge' and ExtResNetBlock against the oracle on bf16-representable inputs."""
    x = torch.from_numpy(O._rng(f"fgp{groups}").standard_normal((2, 32, 9, 11, 21)).astype(np.float32)).bfloat16().float()
    for make_o, make_h in ((lambda: O.SingleConv(32, 32, 3, "cge", groups), lambda: HC.SingleConv(32, 32, 3, "cge", groups)),
                           (lambda: O.ExtResNetBlock(32, 32, order="cge", num_groups=groups),
                            lambda: HC.ExtResNetBlock(32, 32, order="cge", num_groups=groups))):
        ora = O.keyed_init_(make_o())
        xo = x.clone().requires_grad_(True)
        yo = ora(xo)
        g = torch.from_numpy(O._rng("fgpcot").standard_normal(tuple(yo.shape)).astype(np.float32))
        (yo * g).sum().backward()
        with mednet_hip.precision("bf16"):
            net = O.keyed_init_(make_h()).to(DEV)
            xg = x.to(DEV).requires_grad_(True)
            yg = net(xg)
            (yg.float() * g.to(DEV)).sum().backward()
        assert_close(yg, yo, 2e-2, f"y groups={groups}")
        assert_close(xg.grad, xo.grad, 4e-2, f"dx groups={groups}")
        for (k, a), (_, b) in zip(net.named_parameters(), ora.named_parameters()):
            if a.numel() >= 1024:
                assert_close(a.grad, b.grad, 4e-2, f"{k} groups={groups}")


@pytest.mark.parametrize("order", ["gce", "ceg", "gcr"])
def test_conv_act_orders_bf16_without_pooling(order):
    """bf16 mode, the conv -> activation epilogue (mednet_conv3d_act_fwd) with the NEXT GroupNorm's statistics taken from
    the activated output, forward and backward (ConvActFn), isolated from max-pooling: SingleConv and DoubleConv (encoder
    and decoder form) of UNet3D's order family against the oracle on bf16-representable inputs.  ELU orders are smooth and
    compared as they are; for 'gcr' the oracle takes the HIP path's ReLU decisions (_ForcedReLU, see
    test_unet3d_bf16_gradients_once_discrete_decisions_are_equalised), otherwise mask flips near zero dominate dx."""
    x = torch.from_numpy(O._rng("cab" + order).standard_normal((2, 32, 8, 12, 20)).astype(np.float32)).bfloat16().float()
    cases = [(lambda: O.SingleConv(32, 32, 3, order, 8), lambda: HC.SingleConv(32, 32, 3, order, 8)),
             (lambda: O.DoubleConv(32, 64, True, 3, order, 8), lambda: HC.DoubleConv(32, 64, True, 3, order, 8)),
             (lambda: O.DoubleConv(32, 16, False, 3, order, 8), lambda: HC.DoubleConv(32, 16, False, 3, order, 8))]
    for make_o, make_h in cases:
        stored = {}
        with mednet_hip.precision("bf16"):
            net = O.keyed_init_(make_h()).to(DEV)
            for name, m in net.named_modules():
                if isinstance(m, HC.SingleConv):
                    m.register_forward_hook(lambda mod, i, o, name=name: stored.__setitem__(
                        name, (o[0] if isinstance(o, tuple) else o).detach().float().cpu()))
            xg = x.to(DEV).bfloat16().requires_grad_(True)
            yg = net(xg)
            g = torch.from_numpy(O._rng("cabcot").standard_normal(tuple(yg.shape)).astype(np.float32))
            (yg.float() * g.to(DEV)).sum().backward()
        ora = O.keyed_init_(make_o())
        if order == "gcr":
            for name, m in ora.named_modules():
                if isinstance(m, O.SingleConv):
                    m.ReLU = _ForcedReLU(stored[name])
        xo = x.clone().requires_grad_(True)
        yo = ora(xo)
        (yo * g).sum().backward()
        assert_close(yg, yo, 2e-2, f"{order} y")
        assert_close(xg.grad, xo.grad, 4e-2, f"{order} dx")
        for (k, a), (_, b) in zip(net.named_parameters(), ora.named_parameters()):
            if a.numel() >= 1024:
                assert_close(a.grad, b.grad, 4e-2, f"{order} {k}")


@pytest.mark.parametrize("n,c,shape", [(2, 32, (64, 64, 64)), (1, 32, (9, 11, 21)), (2, 64, (16, 24, 32)), (1, 16, (8, 8, 16))])
def test_groupnorm_backward_sums_from_the_data_gradient_epilogue(n, c, shape):
    """ExtResNetBlock backward (components.py:168-180) in bf16 mode: the first pass of GroupNorm-1/-2's backward (sum du,
    sum du*xhat per channel) taken in the epilogue of the data-gradient conv that produces dz (mednet_conv3d_dgrad_gn +
    mednet_gn_act_bwd_fused) against the stand-alone pass (mednet_gn_act_bwd) on the same stored tensors: same gradients up
    to fp32 summation order.  Shapes cover the persistent kernel's accumulate mode (>= 1024 bricks), one-row-per-brick
    mode, ragged volumes, 64 channels (two channel blocks) and a 16-channel layer (half-filled block)."""
    from mednet_hip import block
    x = torch.from_numpy(O._rng(f"gnb{n}{c}{shape}").standard_normal((n, c) + shape).astype(np.float32))
    g = torch.from_numpy(O._rng("gnbcot").standard_normal((n, c) + shape).astype(np.float32))
    res = {}
    for fused in (True, False):
        old = block.FUSE_GNB
        block.FUSE_GNB = fused
        try:
            with mednet_hip.precision("bf16"):
                net = O.keyed_init_(HC.ExtResNetBlock(c, c, order="cge", num_groups=8)).to(DEV)
                xg = x.to(DEV).bfloat16().requires_grad_(True)
                yg = net(xg)
                (yg.float() * g.to(DEV)).sum().backward()
                res[fused] = [xg.grad.float().clone()] + [p.grad.clone() for p in net.parameters()]
        finally:
            block.FUSE_GNB = old
    names = ["dx"] + [k for k, _ in net.named_parameters()]
    for k, a, b in zip(names, res[True], res[False]):
        # (fp32 summation order moves the per-group coefficients by ~1e-7; a few bf16 roundings of dy flip with them)
        assert_close(a, b, 2e-3, f"fused vs stand-alone {k}")
    if c >= 32:  # the fused path really ran (otherwise both runs are the same code)
        from mednet_hip import _lib as L, config
        assert L.lib().mednet_conv3d_dgrad_gn_rows(n, *shape, c, c, config.conv_algo()) > 0


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,c,shape,act", [(2, 32, (32, 32, 32), "e"), (1, 32, (9, 11, 21), "e"), (2, 64, (16, 24, 32), "l"),
                                           (1, 16, (8, 8, 16), "r"), (3, 32, (5, 3, 2), "e"), (1, 32, (64, 64, 64), "e")])
def test_first_layer_groupnorm_backward_inside_its_weight_gradient(mode, n, c, shape, act):
    """The network's first ExtResNetBlock (model.py:63, Cin = 1, components.py:168-180) needs no gradient of the patch: the
    backward of its first GroupNorm feeds only the first conv's weight gradient.  mednet_gn_bwd_coefficients +
    mednet_conv3d_wgrad_c1_gn (the apply pass inside the weight-gradient kernel's staging, dy never stored) against
    mednet_gn_act_bwd_fused + mednet_conv3d_wgrad on the same tensors: every parameter gradient bit-identical."""
    from mednet_hip import block, _lib as L, config
    x = torch.from_numpy(O._rng(f"c1gn{n}{c}{shape}").standard_normal((n, 1) + shape).astype(np.float32))
    g = torch.from_numpy(O._rng("c1gncot").standard_normal((n, c) + shape).astype(np.float32))
    res = {}
    for fused in (True, False):
        old = block.FUSE_C1GN
        block.FUSE_C1GN = fused
        before = block.C1GN_COUNT["fused"]
        try:
            with mednet_hip.precision(mode):
                net = O.keyed_init_(HC.ExtResNetBlock(1, c, order="cg" + act, num_groups=8)).to(DEV)
                yg = net(x.to(DEV))
                (yg.float() * g.to(DEV)).sum().backward()
                res[fused] = [p.grad.clone() for p in net.parameters()]
        finally:
            block.FUSE_C1GN = old
        with mednet_hip.precision(mode):  # (the fused form needs the sums from conv2's data gradient: 32+ channels)
            rows = L.lib().mednet_conv3d_dgrad_gn_rows_dt(n, *shape, c, c, config.conv_algo(), L.dt(yg))
        assert block.C1GN_COUNT["fused"] - before == (1 if fused and rows > 0 else 0)
        assert rows > 0 or c < 32
    for (k, _), a, b in zip(net.named_parameters(), res[True], res[False]):
        assert torch.isfinite(a).all(), k
        assert torch.equal(a, b), f"{k}: fused differs from the two-kernel form by {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("consumer,shape", [("head", (16, 24, 32)), ("head", (9, 11, 21)), ("pool", (16, 24, 32)), ("pool", (8, 8, 16)),
                                            ("convt", (8, 12, 16)), ("convt", (3, 5, 9))])
def test_groupnorm3_backward_sums_from_the_producer_of_the_block_gradient(mode, consumer, shape):
    """ExtResNetBlock backward (components.py:170-178) in the 16-bit modes: the first pass of GroupNorm-3's backward taken
    by the op that produces the block's output gradient -- the 1x1x1 head's data gradient (model.py:204-207), the pooling
    backward + skip-gradient join (model.py:194-205) or the decoder's ConvTranspose3d data gradient (model.py:202-203) -- against the stand-alone pass (MEDNET_FUSE_GN3 off) on the same
    tensors: same gradients up to fp32 summation order; and the sums are really used (ops.GN3_COUNT)."""
    from mednet_hip import ops as hops
    c = 32
    x = torch.from_numpy(O._rng(f"gn3{consumer}{shape}").standard_normal((2, c) + shape).astype(np.float32))
    res = {}
    for fused in (True, False):
        old = hops.FUSE_GN3
        hops.FUSE_GN3 = fused
        before = dict(hops.GN3_COUNT)
        try:
            with mednet_hip.precision(mode):
                blk = O.keyed_init_(HC.ExtResNetBlock(c, c, order="cge", num_groups=8)).to(DEV)
                xg = x.to(DEV).to(torch.bfloat16 if mode == "bf16" else torch.float16).requires_grad_(True)
                out = blk(xg)
                if consumer == "head":
                    head = hnn.Conv3d(c, 4, 1, planar_output=True).to(DEV)
                    with torch.no_grad():
                        head.weight.copy_(torch.from_numpy(O._rng("gn3hw").standard_normal((4, c, 1, 1, 1)).astype(np.float32)) * 0.2)
                        head.bias.zero_()
                    y = head(out)
                    cot = torch.from_numpy(O._rng("gn3hc").standard_normal(tuple(y.shape)).astype(np.float32)).to(DEV)
                    (y * cot).sum().backward()
                elif consumer == "convt":
                    up = hnn.ConvTranspose3d(c, c).to(DEV)
                    with torch.no_grad():
                        up.weight.copy_(torch.from_numpy(O._rng("gn3cw").standard_normal((c, c, 3, 3, 3)).astype(np.float32)) * 0.05)
                        up.bias.zero_()
                    y = up(out)
                    cot = torch.from_numpy(O._rng("gn3cc").standard_normal(tuple(y.shape)).astype(np.float32)).to(DEV)
                    (y.float() * cot).sum().backward()
                else:
                    skip, pooled = hops.skip_pool2(out)
                    c1 = torch.from_numpy(O._rng("gn3pc1").standard_normal(tuple(skip.shape)).astype(np.float32)).to(DEV)
                    c2 = torch.from_numpy(O._rng("gn3pc2").standard_normal(tuple(pooled.shape)).astype(np.float32)).to(DEV)
                    ((skip.float() * c1).sum() + (pooled.float() * c2).sum()).backward()
                res[fused] = [xg.grad.float().clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            hops.FUSE_GN3 = old
        taken = hops.GN3_COUNT["taken"] - before["taken"]
        assert taken == (1 if fused else 0), f"fused={fused}: the block's backward took the producer's sums {taken} times"
    names = ["dx"] + [k for k, _ in blk.named_parameters()]
    for k, a, b in zip(names, res[True], res[False]):
        assert_close(a, b, 2e-3, f"fused vs stand-alone {k}")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("order,cin,cout,shape", [("gcr", 32, 32, (16, 24, 32)), ("gcr", 96, 32, (9, 11, 21)), ("gcl", 32, 64, (8, 8, 16)),
                                                  ("gce", 64, 64, (8, 16, 16))])
def test_plain_groupnorm_backward_sums_from_the_following_conv(mode, order, cin, cout, shape):
    """SingleConv in the 'g c .' orders (UNet3D's default 'gcr', components.py:12-67): GroupNorm -> conv -> activation.  The
    first pass of that GroupNorm's backward is taken in the epilogue of the conv's data gradient (ops.GNBHook) against the
    stand-alone pass: same gradients up to fp32 summation order, and the sums are really used."""
    from mednet_hip import ops as hops
    x = torch.from_numpy(O._rng(f"gnb2{order}{cin}{shape}").standard_normal((2, cin) + shape).astype(np.float32))
    res = {}
    for fused in (True, False):
        old = hops.FUSE_GN3
        hops.FUSE_GN3 = fused
        before = dict(hops.GN3_COUNT)
        try:
            with mednet_hip.precision(mode):
                net = nn.Sequential(O.keyed_init_(HC.SingleConv(cin, cout, 3, order, 8)), O.keyed_init_(HC.SingleConv(cout, cout, 3, order, 8))).to(DEV)
                xg = x.to(DEV).to(torch.bfloat16 if mode == "bf16" else torch.float16).requires_grad_(True)
                y = net(xg)
                cot = torch.from_numpy(O._rng("gnb2c").standard_normal(tuple(y.shape)).astype(np.float32)).to(DEV)
                (y.float() * cot).sum().backward()
                res[fused] = [xg.grad.float().clone()] + [p.grad.clone() for p in net.parameters()]
        finally:
            hops.FUSE_GN3 = old
        taken = hops.GN3_COUNT["taken"] - before["taken"]
        assert taken == (2 if fused else 0), f"fused={fused}: {taken} GroupNorm backward passes took the conv's sums"
        masked = hops.GN3_COUNT["masked"] - before.get("masked", 0)  # the first layer's act' folded into the second GroupNorm's backward
        # (only ReLU layers are folded: applied twice -- when the conv layer has to decline -- a 0/1 mask stays a mask, ELU's or
        #  LeakyReLU's derivative would not)
        assert masked == (1 if fused and order == "gcr" else 0), f"fused={fused}: {masked} activation-backward passes were folded away"
    names = ["dx"] + [k for k, _ in net.named_parameters()]
    for k, a, b in zip(names, res[True], res[False]):
        # (the folded activation derivative saves one rounding of the intermediate gradient to 16 bits: the two paths differ
        #  by that rounding, 2^-9 relative per element in bf16)
        assert_close(a, b, 4e-3 if mode == "bf16" else 1e-3, f"fused vs stand-alone {k}")


@pytest.mark.parametrize("order", ["gcr", "gcl", "gce"])
def test_activation_fold_with_a_second_consumer_of_the_conv_layer_output(order):
    """ops.ActMaskHook decline path: the output z of a fused conv -> activation layer feeds the next layer's GroupNorm AND a
    second consumer, so autograd sums two gradients of z and the conv layer must run its own activation backward.  Whatever
    the next GroupNorm folded into its dx must then not be applied a second time (ReLU: idempotent; LeakyReLU / ELU: the
    fold is off).  Fused (FUSE_GN3) against unfused launches of the same network, and -- for the smooth activations, where
    16-bit storage flips no discrete decision -- against the CPU oracle modules."""
    from mednet_hip import ops as hops
    cin, shape = 32, (8, 8, 16)
    x = torch.from_numpy(O._rng(f"fold2{order}").standard_normal((1, cin) + shape).astype(np.float32))
    c1 = torch.from_numpy(O._rng("fold2c1").standard_normal((1, cin) + shape).astype(np.float32))
    c2 = torch.from_numpy(O._rng("fold2c2").standard_normal((1, cin) + shape).astype(np.float32))
    res = {}
    for fused in (True, False):
        old = hops.FUSE_GN3
        hops.FUSE_GN3 = fused
        try:
            with mednet_hip.precision("bf16"):
                l1 = O.keyed_init_(HC.SingleConv(cin, cin, 3, order, 8)).to(DEV)
                l2 = O.keyed_init_(HC.SingleConv(cin, cin, 3, order, 8)).to(DEV)
                xg = x.to(DEV).to(torch.bfloat16).requires_grad_(True)
                z = l1(xg)
                y = l2(z)
                ((y.float() * c1.to(DEV)).sum() + (z.float() * c2.to(DEV)).sum()).backward()
                res[fused] = [xg.grad.float().cpu()] + [p.grad.cpu() for p in list(l1.parameters()) + list(l2.parameters())]
        finally:
            hops.FUSE_GN3 = old
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert_close(a, b, 4e-3, f"{order}: gradient {i}, fused vs unfused, second consumer of the conv layer's output")
    if order == "gce":  # (ReLU and LeakyReLU are kinked: bf16 storage flips a few of their decisions against the fp32 oracle)
        o1, o2 = O.keyed_init_(O.SingleConv(cin, cin, 3, order, 8)), O.keyed_init_(O.SingleConv(cin, cin, 3, order, 8))
        xo = x.to(torch.bfloat16).float().requires_grad_(True)
        zo = o1(xo)
        yo = o2(zo)
        ((yo * c1).sum() + (zo * c2).sum()).backward()
        want = [xo.grad] + [p.grad for p in list(o1.parameters()) + list(o2.parameters())]
        for i, (a, b) in enumerate(zip(res[True], want)):
            assert_close(a, b, 3e-2, f"{order}: gradient {i} vs the oracle")


def test_groupnorm3_sums_are_declined_when_the_block_output_has_a_second_consumer():
    """ops.GN3Hook: if autograd accumulates another gradient into the block's output gradient, the producer's sums no longer
    describe it; the block must notice (object identity + version counter) and run its stand-alone pass."""
    from mednet_hip import ops as hops
    c, shape = 32, (8, 8, 16)
    x = torch.from_numpy(O._rng("gn3two").standard_normal((1, c) + shape).astype(np.float32))
    res = {}
    for fused in (True, False):
        old = hops.FUSE_GN3
        hops.FUSE_GN3 = fused
        before = dict(hops.GN3_COUNT)
        try:
            with mednet_hip.precision("bf16"):
                blk = O.keyed_init_(HC.ExtResNetBlock(c, c, order="cge", num_groups=8)).to(DEV)
                xg = x.to(DEV).bfloat16().requires_grad_(True)
                out = blk(xg)
                skip, pooled = hops.skip_pool2(out)
                (skip.float().sum() + 2.0 * pooled.float().sum() + 3.0 * (out.float() ** 2).sum()).backward()  # `out` used twice
                res[fused] = [xg.grad.float().clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            hops.FUSE_GN3 = old
        assert hops.GN3_COUNT["taken"] == before["taken"], "stale sums were used"
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


def test_fp32_tensor_into_a_bf16_mode_conv_act_layer():
    """mednet_conv3d_act_fwd reads bf16 unconditionally; an fp32 tensor handed to a bf16-mode 'crg' layer with
    16/32 input channels (a block called stand-alone, a multi-channel network input) must not be read as bf16."""
    x = torch.from_numpy(O._rng("f32in").standard_normal((1, 16, 6, 8, 16)).astype(np.float32))
    ora = O.keyed_init_(O.SingleConv(16, 32, 3, "crg", 8))
    yo = ora(x)
    with mednet_hip.precision("bf16"):
        net = O.keyed_init_(HC.SingleConv(16, 32, 3, "crg", 8)).to(DEV)
        yg = net(x.to(DEV))  # fp32 on purpose
    assert_close(yg, yo, 2e-2, "crg with an fp32 input")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_batched_repack_after_the_optimizer_step_equals_per_layer_packing(mode):
    """train.BatchedRepack: after Adam, ONE launch rebuilds the packed images of all matrix-core conv layers in place; they
    must be byte-identical to what the per-layer packing (nn._PackedWeightMixin._packed) would build from the new weights,
    and the next forward must not repack them."""
    from mednet_hip import ops as hops, train as T
    from mednet_hip.synth import synthetic_batch
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HM.ResidualUNet3D(1, 3, False, f_maps=[16, 32, 64])).to(DEV)
        step = T.SegmentationStep(net, loss_weight=[0.2, 1.0, 1.0], lr=1e-2)
        batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 1, (16, 16, 32), 3, 0, seed=3).items()}
        step(batch)   # first step packs layer by layer; its optimizer step is followed by the batched repack
        step(batch)
        assert step.repack.table is not None and len(step.repack.mods) >= 10
        up = lambda v, a: (v + a - 1) // a * a
        for m in step.repack.mods:
            fresh = hops.pack_conv_weight(m.weight, 3, m._transposed)
            w = m.weight
            cin, cout = (w.shape[0], w.shape[1]) if m._transposed else (w.shape[1], w.shape[0])
            # the four sections of a pack buffer (include/mednet_hip.h; the gaps between them are never written)
            f32 = up(27 * cin * cout * 4, 256)
            fwd, bwd = 27 * up(cout, 32) * cin * 2, 27 * up(cin, 32) * cout * 2
            for lo, size in ((0, 27 * cin * cout * 4), (f32, 27 * cin * cout * 4), (2 * f32, fwd), (2 * f32 + up(fwd, 256), bwd)):
                assert torch.equal(fresh[lo:lo + size], m._pack_buf[lo:lo + size]), "batched repack differs from the per-layer pack"
            assert m._packed() is m._pack_buf, "the next forward would repack this layer"
        step.flat.release()


def test_trainer_direct_gradients_and_fused_adam(golden_dir):
    """train.SegmentationStep (flat buffers, kernels writing gradients in place, fused Adam) follows the same
    trajectory as the oracle + torch.optim.Adam for three steps (segmentation.py:58-65,119-120)."""
    from mednet_hip.train import SegmentationStep
    ctor = dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8])
    batch = O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234)
    w = [0.05, 1.0]
    ora = O.keyed_init_(O.ResidualUNet3D(**ctor))
    opt_o = torch.optim.Adam(ora.parameters(), lr=1e-3)
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=w, lr=1e-3)
        gb = {k: v.to(DEV) for k, v in batch.items()}
        for i in range(3):
            opt_o.zero_grad()
            lo = O.seg_training_step(ora, O.DiceLoss(weight=torch.tensor(w)), batch)
            lo.backward()
            if i == 0:
                first = {k: p.grad.clone() for k, p in ora.named_parameters()}
            opt_o.step()
            lg = step(gb)
            if i == 0:
                step.flat.grads_as_attr()
                for k, p in net.named_parameters():
                    assert_close(p.grad, first[k], 1e-3, f"direct grad {k}")
            assert abs(float(lg) - float(lo)) <= 2e-4, (i, float(lg), float(lo))
    for (k, a), (_, b) in zip(net.named_parameters(), ora.named_parameters()):
        assert rel(a, b) <= 2e-3, k


def test_reference_caller_fixtures_meet_the_hip_path(golden_dir):
    """The vectors tools/make_golden.py captured from the REFERENCE's own callers, against the HIP path directly (no oracle in
    between): `SegmentationNet.training_step`'s first loss (segmentation.py:58-65), `LandmarkNet.training_step`'s three losses
    (landmarks.py:66-83,125-134) and the parameter delta of one `configure_optimizers()` Adam step (segmentation.py:119-120),
    fp32 storage mode."""
    from mednet_hip.train import LandmarkStep, SegmentationStep
    rec = np.load(os.path.join(golden_dir, "callers.npz"))
    cfg1 = np.load(os.path.join(golden_dir, "res_cfg1.npz"))
    with mednet_hip.precision("fp32"):
        # -- segmentation caller: callers.npz and res_cfg1.npz were captured on the same batch (2 x 1 x 32^3, seed 1234)
        assert tuple(int(v) for v in cfg1["meta.shape"]) == (32, 32, 32) and int(cfg1["meta.n"]) == 2 and int(cfg1["meta.seed"]) == 1234
        net = O.keyed_init_(HM.ResidualUNet3D(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8])).to(DEV)
        before = {k: p.detach().clone() for k, p in net.named_parameters()}
        step = SegmentationStep(net, loss_weight=[0.05, 1.0], lr=float(rec["seg.adam"][0]))
        batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234).items()}
        loss = step(batch)
        assert abs(float(loss) - float(rec["seg.loss"])) <= 1e-4, (float(loss), float(rec["seg.loss"]))
        assert abs(float(loss) - float(cfg1["loss"])) <= 1e-4
        # one fused-Adam step (train.FlatAdam) against the reference optimizer's parameter delta; the first Adam step moves
        # every parameter by lr * g / (|g| + eps): +-lr wherever |g| >> eps, so the bound is the CPU test's (rtol 1e-3)
        bad = total = 0
        for k, p in net.named_parameters():
            want = torch.from_numpy(cfg1["adam_delta." + k]).to(DEV)
            got = p.detach() - before[k]
            bad += int(((got - want).abs() > 2e-6 + 1e-3 * want.abs()).sum())
            total += want.numel()
        # (a gradient within float noise of zero may take the other sign on another device: a handful of the 3 738 elements)
        assert bad <= 4, (bad, total)
        print(f"[callers] seg loss {float(loss):.8f} (reference {float(rec['seg.loss']):.8f}); Adam delta: {bad} of {total} elements outside rtol 1e-3")
        # -- landmark caller (hp2 of tools/make_golden.py: 3 heat maps + 2 classes, f_maps [8], 16^3, seed 4321)
        net2 = O.keyed_init_(HM.ResidualUNet3D(in_channels=1, out_channels=5, final_sigmoid=False, f_maps=[8])).to(DEV)
        step2 = LandmarkStep(net2, class_weight=[0.05, 1.0], regression_weight=[0.015] * 3, regression="L2", lr=1e-3)
        batch2 = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (16, 16, 16), 2, 3, seed=4321).items()}
        tot, cl, rg = step2(batch2)
        assert abs(float(tot) - float(rec["ldmk.loss"])) <= 1e-4 * float(rec["ldmk.loss"]), (float(tot), float(rec["ldmk.loss"]))
        assert abs(float(cl) - float(rec["ldmk.class_loss"])) <= 1e-4, (float(cl), float(rec["ldmk.class_loss"]))
        assert abs(float(rg) - float(rec["ldmk.regression_loss"])) <= 1e-4 * float(rec["ldmk.regression_loss"])
        print(f"[callers] ldmk losses {float(tot):.5f} / {float(cl):.7f} / {float(rg):.5f} (reference {float(rec['ldmk.loss']):.5f} / "
              f"{float(rec['ldmk.class_loss']):.7f} / {float(rec['ldmk.regression_loss']):.5f})")


def test_landmark_step_matches_oracle():
    """LandmarkNet.training_step + .loss (landmarks.py:66-83,125-134) through train.LandmarkStep: uint8 heat maps straight
    from the batch dict, fused regression loss, Dice on the class slice, flat-buffer Adam."""
    from mednet_hip.train import LandmarkStep
    ctor = dict(in_channels=1, out_channels=5, final_sigmoid=False, f_maps=[8, 16])
    batch = O.synthetic_batch(2, 1, (16, 16, 16), 2, 3, seed=4321)
    regw = [0.015, 0.02, 0.001]
    ora = O.keyed_init_(O.ResidualUNet3D(**ctor))
    opt = torch.optim.Adam(ora.parameters(), lr=1e-3)
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = LandmarkStep(net, class_weight=[0.05, 1.0], regression_weight=regw, regression="L2", lr=1e-3)
        gb = {k: v.to(DEV) for k, v in batch.items()}
        for i in range(2):
            opt.zero_grad()
            tot, cl, rg = O.ldmk_training_step(ora, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), nn.MSELoss(), regw, batch)
            tot.backward()
            opt.step()
            gt, gc, gr = step(gb)
            assert abs(float(gc) - float(cl)) <= 2e-4, (i, float(gc), float(cl))
            assert abs(float(gr) - float(rg)) <= 2e-4 * abs(float(rg)), (i, float(gr), float(rg))
            assert abs(float(gt) - float(tot)) <= 2e-4 * abs(float(tot))
    for (k, a), (_, b) in zip(net.named_parameters(), ora.named_parameters()):
        assert rel(a, b) <= 2e-3, k


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,nh,ncls,kind,sigmoid,n", [((32, 32, 32), 16, 2, "L2", False, 2), ((12, 10, 6), 5, 3, "L1", False, 3),
                                                         ((8, 8, 20), 16, 2, "L2", True, 1), ((64, 48, 32), 16, 2, "L2", False, 1),
                                                         ((4, 4, 4), 1, 1, "L2", True, 2)])
def test_landmark_head_on_the_matrix_cores_against_the_unfused_launches(mode, shape, nh, ncls, kind, sigmoid, n):
    """LandmarkNet.training_step + .loss (landmarks.py:66-83, 125-134) in the 16-bit storage modes: head and both loss terms as one
    matrix-core node (ops.head_landmark: logits^T = W z^T, dz^T = W^T dl^T, dW = dl^T z with hi + lo split operands, loss terms
    and logit gradient in registers) against the stock launches (1x1x1 head, heat-map and Dice kernels, head data / weight
    gradient) on the same network and batch: same losses and gradients up to fp32 summation order.  Shapes cover whole and
    partial 128-voxel runs, several samples, fewer than 16 heat maps, 1 .. 3 classes, L1 / L2, softmax / sigmoid Dice."""
    from mednet_hip import ops as hops
    from mednet_hip.train import LandmarkStep
    ctor = dict(in_channels=1, out_channels=nh + ncls, final_sigmoid=sigmoid, f_maps=[32, 64])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(n, 1, shape, ncls, nh, seed=99).items()}
    regw = [0.015 + 0.003 * i for i in range(nh)]
    res = {}
    for fused in (True, False):
        old = hops.FUSE_HEAD_LOSS
        hops.FUSE_HEAD_LOSS = fused
        try:
            with mednet_hip.precision(mode):
                net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
                step = LandmarkStep(net, class_weight=[0.05, 1.0, 0.7][:ncls], regression_weight=regw, regression=kind, lr=1e-3)
                step.loss_class.sigmoid_normalization = sigmoid
                scale = step.scaler.snapshot()[0] if step.scaler is not None else 1.0
                tot, cl, rg = step._fwd_bwd(batch)
                torch.cuda.synchronize()
                res[fused] = (float(tot), float(cl), float(rg), (step.flat.grad / scale).clone())
                with torch.no_grad():
                    took = hops.head_landmark_supported(net.forward_features(batch["data"].float()), 32, nh, ncls,
                                                        batch["label"][:, :-1], batch["label"][:, -1])
                step.flat.release()
        finally:
            hops.FUSE_HEAD_LOSS = old
        assert took == fused
    (t1, c1, r1, g1), (t0, c0, r0, g0) = res[True], res[False]
    assert abs(c1 - c0) <= 2e-6 * max(1.0, abs(c0)), (c1, c0)
    assert abs(r1 - r0) <= 2e-6 * max(1.0, abs(r0)), (r1, r0)
    assert torch.isfinite(g1).all()
    e = float((g1 - g0).norm() / g0.norm())
    print(f"[landmark head fused vs stock, {mode} {shape} nh={nh} ncls={ncls}] losses {c1:.7f}/{c0:.7f} {r1:.5f}/{r0:.5f}  all gradients rel-L2 {e:.2e}")
    assert e <= 3e-3, e


def test_landmark_head_node_without_the_groupnorm_hook():
    """ops.head_landmark when the producing block offers no GroupNorm-3 hook (MEDNET_FUSE_GN3 off: the block runs its own first
    pass): the kernel's gn_y == NULL path -- no sums taken, dz stored as before -- against the stock launches."""
    from mednet_hip import ops as hops
    from mednet_hip.train import LandmarkStep
    ctor = dict(in_channels=1, out_channels=18, final_sigmoid=False, f_maps=[32, 64])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (16, 24, 20), 2, 16, seed=5).items()}
    res = {}
    old_gn3 = hops.FUSE_GN3
    hops.FUSE_GN3 = False
    try:
        for fused in (True, False):
            old = hops.FUSE_HEAD_LOSS
            hops.FUSE_HEAD_LOSS = fused
            try:
                with mednet_hip.precision("bf16"):
                    net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
                    step = LandmarkStep(net, [0.05, 1.0], [0.015] * 16, "L2")
                    tot, cl, rg = step._fwd_bwd(batch)
                    torch.cuda.synchronize()
                    res[fused] = (float(cl), float(rg), step.flat.grad.clone())
                    step.flat.release()
            finally:
                hops.FUSE_HEAD_LOSS = old
    finally:
        hops.FUSE_GN3 = old_gn3
    (c1, r1, g1), (c0, r0, g0) = res[True], res[False]
    assert abs(c1 - c0) <= 2e-6 * max(1.0, abs(c0)) and abs(r1 - r0) <= 2e-6 * max(1.0, abs(r0))
    assert torch.isfinite(g1).all() and float((g1 - g0).norm() / g0.norm()) <= 3e-3


def test_cfg5_shape_smoke_bf16():
    """BASELINE config 5's topology (5 levels, 64 base channels -> 1024 at the bottom) at a reduced patch, bf16 storage:
    exercises the >256-channel paths (GroupNorm columns, wide bias sums, 32x32 channel-block pairs up to 1024x1024)
    against the oracle."""
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[64, 128, 256, 512, 1024])
    batch = O.synthetic_batch(1, 1, (32, 32, 16), 4, 0, seed=1234)
    ora = O.keyed_init_(O.ResidualUNet3D(**ctor))
    lo = ora(batch["data"])
    loss_o = O.DiceLoss(weight=torch.tensor([0.05, 1, 1, 1.0]))(lo, batch["label"][:, -1].long())
    loss_o.backward()
    with mednet_hip.precision("bf16"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        lg = net(batch["data"].to(DEV))
        loss = HL.DiceLoss(weight=torch.tensor([0.05, 1, 1, 1.0], device=DEV)).to(DEV)(lg, batch["label"][:, -1].long().to(DEV))
        loss.backward()
    assert_close(lg, lo, 3e-2, "cfg5 logits")
    num = den = 0.0
    for p, q in zip(net.parameters(), ora.parameters()):
        num += float((p.grad.cpu().double() - q.grad.double()).pow(2).sum())
        den += float(q.grad.double().pow(2).sum())
    assert (num / den) ** 0.5 <= 6e-2


def test_graph_replay_matches_eager_steps():
    """train._GraphedStep: forward+loss+backward captured as one hipGraph (both streams) and replayed gives bit-identical
    losses and parameters to the eager launch sequence over five Adam steps (two eager warm-up calls, capture, replays)."""
    from mednet_hip.train import SegmentationStep
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64])
    batches = [O.synthetic_batch(2, 1, (32, 32, 32), 4, 0, seed=100 + i) for i in range(5)]
    res = {}
    for graph in (False, True):
        with mednet_hip.precision("bf16"):
            net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
            step = SegmentationStep(net, loss_weight=[0.05, 1, 1, 1.0], lr=1e-3, graph=graph)
            losses = []
            for b in batches:
                losses.append(float(step({k: v.to(DEV) for k, v in b.items()})))
            torch.cuda.synchronize()
            res[graph] = (losses, step.flat.flat.clone())
            step.flat.release()
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])


def test_validation_step_matches_reference_golden(golden_dir):
    """Row N3: train.SegmentationValidation (forward + fused DiceLoss + fused dice_metric, device scalars) against the
    oracle and the reference's validation_epoch_end values (callers.npz), fp32 mode."""
    from mednet_hip.train import SegmentationValidation
    rec = np.load(os.path.join(golden_dir, "callers.npz"))
    ora = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8])).eval()
    batches = [O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=600 + i) for i in range(2)]
    want = O.validation_epoch_end([O.seg_validation_step(ora, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), b) for b in batches])
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(1, 2, False, f_maps=[8])).to(DEV)
        val = SegmentationValidation(net, loss_weight=[0.05, 1.0])
        outs = [val.validation_step({k: v.to(DEV) for k, v in b.items()}, i) for i, b in enumerate(batches)]
        end = val.validation_epoch_end(outs)
    assert sorted(end["log"].keys()) == ["val_dice0", "val_dice1", "val_loss"]
    for k, v in want.items():
        assert abs(float(end["log"][k]) - float(v)) <= 2e-5 * max(1.0, abs(float(v))), k
        assert abs(float(end["log"][k]) - float(rec["seg.val." + k])) <= 2e-5 * max(1.0, abs(float(v))), k


def test_bf16_training_mode_tracks_fp32_mode_loss_curve():
    """The benchmarked mode (bf16 storage, matrix-core kernels, side-stream weight gradients, fused Adam) trains: over 12
    steps on one synthetic batch its Dice loss falls and stays within 2 % of the fp32-mode (1e-3 parity) trajectory."""
    from mednet_hip.train import SegmentationStep
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (32, 32, 32), 4, 0, seed=5).items()}
    curves = {}
    for mode in ("fp32", "bf16"):
        with mednet_hip.precision(mode):
            net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
            step = SegmentationStep(net, loss_weight=[0.05, 1, 1, 1.0], lr=1e-3)
            curves[mode] = [float(step(batch)) for _ in range(12)]
            step.flat.release()
    a, b = np.array(curves["fp32"]), np.array(curves["bf16"])
    assert a[-1] < a[0] - 0.02 and b[-1] < b[0] - 0.02, (a, b)
    assert np.all(np.abs(a - b) <= 0.02 * np.abs(a)), (a, b)


@pytest.mark.parametrize("mode", ["bf16", "fp32", "fp16"])
def test_fused_head_and_loss_step_equals_the_two_node_step(mode):
    """train.SegmentationStep with the 1x1x1 head and DiceLoss as one node (ops.head_dice; model.py:207 + loss.py:114-130,
    segmentation.py:61-62) against the same step with the two calls as they stand: loss and every gradient in front of the
    head BIT-identical (the feature gradient and the GroupNorm-3 sums the fused kernel hands to the last decoder block are
    the unfused kernels' values), the head's own weight / bias gradient to 1e-5."""
    from mednet_hip import ops
    from mednet_hip.train import SegmentationStep
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (24, 40, 32), 4, 0, seed=31).items()}
    res = {}
    old = ops.FUSE_HEAD_LOSS
    try:
        for fused in (False, True):
            ops.FUSE_HEAD_LOSS = fused
            with mednet_hip.precision(mode):
                net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
                step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
                before = dict(ops.GN3_COUNT)
                (loss,) = step._fwd_bwd(batch)
                torch.cuda.synchronize()
                taken = ops.GN3_COUNT["taken"] - before["taken"]
                names = [(n, off, p.numel()) for (n, p), off in zip(net.named_parameters(), step.flat.offsets)]
                res[fused] = (float(loss), step.flat.grad.clone(), taken)
                step.flat.release()
    finally:
        ops.FUSE_HEAD_LOSS = old
    assert res[True][0] == res[False][0]
    assert res[True][2] == res[False][2] and res[True][2] > 0  # the last decoder block took its GroupNorm-3 sums both ways
    for name, off, cnt in names:
        a, b = res[True][1][off:off + cnt], res[False][1][off:off + cnt]
        if name.startswith("final_conv."):
            assert_close(a, b, 1e-5, name)
        else:
            assert torch.equal(a, b), f"{name}: gradient differs between the fused and the two-node step"


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
def test_pooling_inside_the_block_output_pass_changes_nothing(mode):
    """Encoder blocks write the next level's pooled input in the pass that writes their output (block.PoolStash, model.py:194-199):
    loss and every gradient of a three-level network bit-identical to the step with a stand-alone pooling launch."""
    from mednet_hip import block
    from mednet_hip.train import SegmentationStep
    ctor = dict(in_channels=1, out_channels=3, final_sigmoid=False, f_maps=[32, 64, 96])
    res = {}
    for shape in ((16, 24, 32), (8, 12, 20)):
        batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, shape, 3, 0, seed=3).items()}
        old = block.FUSE_POOL
        try:
            for fused in (False, True):
                block.FUSE_POOL = fused
                with mednet_hip.precision(mode):
                    net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
                    step = SegmentationStep(net, loss_weight=None, lr=1e-3)
                    (loss,) = step._fwd_bwd(batch)
                    torch.cuda.synchronize()
                    res[fused] = (float(loss), step.flat.grad.clone())
                    step.flat.release()
        finally:
            block.FUSE_POOL = old
        assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1]), shape


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("pool", ["max", "avg"])
def test_the_unwritten_gradient_of_an_encoder_level_changes_nothing(mode, pool):
    """The pooling backward of an encoder level takes GroupNorm-3's sums WITHOUT storing the gradient of the block output; the block's
    apply pass rebuilds it (ops.SkipPool2Fn lazy, mednet_gn_act_bwd_fused_res_pool; model.py:194-205 gives the level's output to the
    pooling and to the skip join only): loss and every gradient bit-identical to the step that materialises it."""
    from mednet_hip.train import SegmentationStep
    from mednet_hip import ops
    ctor = dict(in_channels=1, out_channels=3, final_sigmoid=False, f_maps=[32, 64, 96])
    res = {}
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (16, 24, 32), 3, 0, seed=5).items()}
    old = ops.LAZY_POOL
    try:
        for lazy in (False, True):
            ops.LAZY_POOL = lazy
            for k in ops.GN3_COUNT:
                ops.GN3_COUNT[k] = 0
            with mednet_hip.precision(mode):
                net = O.keyed_init_(HM.ResidualUNet3D(**ctor))
                if pool == "avg":  # (components.py:206-214: Encoder's pool_type; the U-Net classes never pass it)
                    for enc in net.encoders[1:]:
                        enc.pooling = hnn.AvgPool3d(kernel_size=(2, 2, 2))
                step = SegmentationStep(net.to(DEV), loss_weight=None, lr=1e-3)
                (loss,) = step._fwd_bwd(batch)
                torch.cuda.synchronize()
                res[lazy] = (float(loss), step.flat.grad.clone(), dict(ops.GN3_COUNT))
                step.flat.release()
    finally:
        ops.LAZY_POOL = old
    assert res[True][2] == res[False][2] and res[True][2]["declined"] == 0, res[True][2]
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])


def test_a_second_consumer_of_a_sole_consumer_output_is_an_error():
    """skip_pool2(sole_consumer=True) is a contract of the caller; breaking it must not yield a silently wrong gradient."""
    from mednet_hip import ops, _lib as L
    from mednet_hip.unet.components import ExtResNetBlock
    with mednet_hip.precision("bf16"):
        blk = ExtResNetBlock(8, 16).to(DEV)
        x = torch.randn(1, 8, 8, 8, 8, device=DEV)
        out = blk(x)
        skip, pooled = ops.skip_pool2(out, L.POOL_MAX, sole_consumer=True)
        loss = pooled.float().sum() + skip.float().sum() + out.float().sum()  # `out` used again: the contract is broken
        with pytest.raises(RuntimeError, match="another consumer"):
            loss.backward()
        out = blk(x)  # ... and the honest use of the same call works and matches the materialised form
        skip, pooled = ops.skip_pool2(out, L.POOL_MAX, sole_consumer=True)
        (pooled.float().square().sum() + skip.float().sum()).backward()


def test_a_forward_hook_on_an_encoder_level_keeps_its_output_gradient_materialised():
    """the U-Net's forward may call skip_pool2(sole_consumer=True) only when nothing else can have seen the level's
    output.  A forward hook that feeds an encoder output to an auxiliary (deep-supervision) loss is such a consumer: the network
    must notice the hook, keep the gradient materialised, and the gradients must equal those of the run with the optimisation off."""
    from mednet_hip import ops
    ctor = dict(in_channels=1, out_channels=3, final_sigmoid=False, f_maps=[32, 64])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (16, 16, 32), 3, 0, seed=21).items()}
    res = {}
    old = ops.LAZY_POOL
    try:
        for lazy in (False, True):
            ops.LAZY_POOL = lazy
            with mednet_hip.precision("bf16"):
                net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
                seen = []
                h = net.encoders[0].register_forward_hook(lambda m, i, o: seen.append(o))
                lg = net(batch["data"].float())
                loss = HL.DiceLoss().to(DEV)(lg, batch["label"][:, -1].long()) + 1e-3 * seen[0].float().square().mean()
                loss.backward()  # (raised "another consumer" before the hook check existed)
                h.remove()
                res[lazy] = (float(loss), [p.grad.clone() for p in net.parameters()])
    finally:
        ops.LAZY_POOL = old
    assert res[True][0] == res[False][0]
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)


def test_an_in_place_change_of_a_block_output_invalidates_the_pooled_stash():
    """the pooled tensor a block stashes beside its output (block.PoolStash) is that of the values it WROTE; after an
    in-place change of the output (a hook's clamp_, user code between the levels; inference too) the pooling must run on the
    current values."""
    from mednet_hip import ops, _lib as L, block
    from mednet_hip.unet.components import ExtResNetBlock
    with mednet_hip.precision("bf16"), torch.no_grad():
        blk = ExtResNetBlock(8, 32).to(DEV)
        x = torch.randn(2, 8, 8, 16, 16, device=DEV)
        out = blk(x, pool_mode=L.POOL_MAX)
        assert getattr(out, "_mednet_pooled", None) is not None and out._mednet_pooled.pooled is not None
        want_untouched = ops.pool2(out.clone(), L.POOL_MAX)
        assert torch.equal(ops.pool2(out, L.POOL_MAX), want_untouched)          # the stash is used ...
        out = blk(x, pool_mode=L.POOL_MAX)
        out.clamp_(max=0.05)
        want = ops.pool2(out.clone(), L.POOL_MAX)
        got = ops.pool2(out, L.POOL_MAX)                                        # ... but not after the output changed
        assert torch.equal(got, want) and not torch.equal(want, want_untouched)


def test_side_stream_weight_gradients_ask_for_half_the_chip_as_an_argument():
    """A weight gradient launched on the side stream is planned for 128 workgroups (profiles/r04_ab.md section 9).  Since round 5
    that number is an ARGUMENT of mednet_conv3d_wgrad / mednet_convt3d_wgrad and of their workspace queries (ops._OnSide hands it
    to the call it encloses); no library option is touched, so nothing a second thread or device plans can be affected.  One on
    the main stream keeps the one-per-CU plan.  Both step classes switch the side stream on (train.use_side_stream)."""
    from mednet_hip import ops, _lib as L
    from mednet_hip.train import SegmentationStep, LandmarkStep
    lib = L.lib()
    q = lambda: lib.mednet_get_option(b"wgrad_wgs", -7)  # (-7: the option does not exist)
    with ops._OnSide(True, torch.device(DEV)) as side:
        half = side.workgroups
        assert q() == -7
    # ops.side_stream tested its candidates for real overlap with the compute stream: with one found the plan is 128 workgroups,
    # with none (every stream on the compute stream's hardware queue) the one-per-CU plan stays
    want = 128 if ops.SIDE.get("overlaps") else 0
    assert ops.SIDE["wgrad_wgs"] == want and half == want, (ops.SIDE, half)
    with ops._OnSide(False, torch.device(DEV)) as side:
        assert side.workgroups == 0
    ops.join_side_stream()
    with mednet_hip.precision("bf16"):  # and a launch with fewer workgroups computes the same gradient (another summation order)
        x = torch.randn(2, 32, 16, 16, 32, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
        dy = torch.randn(2, 32, 16, 16, 32, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
        dws = []
        for k in (0, 128, 16):
            ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(2, 16, 16, 32, 32, 32, 3, k), dtype=torch.uint8, device=DEV)
            dw = torch.full((32, 32, 3, 3, 3), float("nan"), device=DEV)
            L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, 2, 16, 16, 32, 32, 32, 3, L.BF16, L.NDHWC,
                                            L.BF16, L.NDHWC, L.ALGO_AUTO, k, ws.data_ptr(), ws.numel(), L.stream()), "conv3d_wgrad")
            dws.append(dw)
        assert rel(dws[1], dws[0]) <= 1e-5 and rel(dws[2], dws[0]) <= 1e-5
        assert lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dws[0].data_ptr(), None, 2, 16, 16, 32, 32, 32, 3, L.BF16, L.NDHWC,
                                       L.BF16, L.NDHWC, L.ALGO_AUTO, -1, ws.data_ptr(), ws.numel(), L.stream()) == -1  # MEDNET_E_SHAPE
    for make, nlab, nhm in ((lambda net: SegmentationStep(net, loss_weight=None, lr=1e-3), 3, 0),
                            (lambda net: LandmarkStep(net, [0.05, 1.0], [0.015] * 2, "L2"), 2, 2)):
        ops.SIDE["enabled"] = False
        with mednet_hip.precision("bf16"):
            net = O.keyed_init_(HM.ResidualUNet3D(1, nlab + nhm, False, f_maps=[32, 64])).to(DEV)
            step = make(net)
            assert ops.SIDE["enabled"]
            batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (16, 16, 32), nlab, nhm, seed=11).items()}
            out = step._fwd_bwd(batch)
            torch.cuda.synchronize()
            assert all(torch.isfinite(torch.as_tensor(float(v))) for v in out) and q() == -7
            step.flat.release()


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_relu_mask_inside_the_pooling_join_and_the_head_changes_nothing(mode):
    """UNet3D ('gcr', components.py:57-63): an encoder level's output is the output of a fused conv -> ReLU layer; the join of its
    two gradients (next level's pooling backward + the decoder's skip gradient, model.py:194-205) folds ReLU' in and the layer's
    backward skips its activation pass (mednet_pool2_bwd_act, ops.ActMaskHook); the fused head + Dice backward does the same for
    the last decoder block (mednet_head_dice_bwd): loss and every gradient bit-identical."""
    from mednet_hip import ops
    from mednet_hip.train import SegmentationStep
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (16, 24, 32), 3, 0, seed=7).items()}
    res = {}
    old = ops.POOL_ACT_MASK
    try:
        for fused in (False, True):
            ops.POOL_ACT_MASK = fused
            ops.GN3_COUNT["masked"] = 0
            with mednet_hip.precision(mode):
                net = O.keyed_init_(HM.UNet3D(1, 3, False, f_maps=[32, 64, 128])).to(DEV)
                step = SegmentationStep(net, loss_weight=None, lr=1e-3)
                (loss,) = step._fwd_bwd(batch)
                torch.cuda.synchronize()
                res[fused] = (float(loss), step.flat.grad.clone(), ops.GN3_COUNT["masked"])
                step.flat.release()
    finally:
        ops.POOL_ACT_MASK = old
    assert res[True][2] == res[False][2] + 3, (res[True][2], res[False][2])  # two encoder levels that feed a pooling + the head
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])


@pytest.mark.parametrize("shape", [(16, 24, 32), (12, 20, 18)])
def test_skip_gradient_read_inside_the_concatenation_gradient_changes_nothing(shape):
    """UNet3D: the gradient of a skip tensor is the leading channels of the gradient of the decoder's concatenation
    (components.py:277-280); the pooling join of the encoder level reads it there (mednet_pool2_bwd_act, add_channels) instead
    of from a copy.  Loss and every gradient bit-identical; sizes whose deeper levels are odd take the dense path."""
    from mednet_hip import ops
    from mednet_hip.train import SegmentationStep
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, shape, 3, 0, seed=9).items()}
    res = {}
    old = ops.UPCAT_VIEW
    try:
        for view in (False, True):
            ops.UPCAT_VIEW = view
            with mednet_hip.precision("bf16"):
                net = O.keyed_init_(HM.UNet3D(1, 3, False, f_maps=[32, 64, 128])).to(DEV)
                step = SegmentationStep(net, loss_weight=None, lr=1e-3)
                (loss,) = step._fwd_bwd(batch)
                torch.cuda.synchronize()
                res[view] = (float(loss), step.flat.grad.clone())
                step.flat.release()
    finally:
        ops.UPCAT_VIEW = old
    assert res[True][0] == res[False][0] and torch.equal(res[True][1], res[False][1])
