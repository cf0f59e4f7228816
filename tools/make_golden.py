#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (tobiashepp/torch-mednet at /root/reference).

Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

For every case the reference module and the oracle (oracle/ref_cpu.py) are built, given identical name-keyed
weights and identical synthetic inputs, run forward + loss + backward on CPU fp32, and asserted BIT-IDENTICAL
(logits, loss, every .grad).  Only then is the fixture written.  Fixtures hold data only: inputs are regenerated
from seeds, expected outputs are stored in full for small cases and as (norm, sum, keyed projection, head samples)
for large ones.
"""
from __future__ import annotations

import argparse
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_cpu as O  # noqa: E402


def import_reference():
    """pytorch_lightning is only a base class in model.py:11,113 -> stub it with nn.Module."""
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    sys.modules.setdefault("pytorch_lightning", pl)
    sys.path.insert(0, REF)
    import midasmednet.unet.model as rmodel
    import midasmednet.unet.components as rcomp
    import midasmednet.unet.loss as rloss
    return rmodel, rcomp, rloss


def import_reference_callers():
    """segmentation.py / landmarks.py import logging/plot deps that are absent here; stub them empty."""
    import argparse as _ap
    for name in ["configargparse", "torchvision", "torchvision.transforms", "torchvision.utils", "nibabel",
                 "imageio", "matplotlib", "matplotlib.pyplot", "matplotlib.colors"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    sys.modules["configargparse"].ArgumentParser = getattr(sys.modules["configargparse"], "ArgumentParser",
                                                          _ap.ArgumentParser)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "transforms"):
        tv.transforms = sys.modules["torchvision.transforms"]
    if not hasattr(sys.modules["torchvision.utils"], "make_grid"):
        sys.modules["torchvision.utils"].make_grid = lambda *a, **k: None
    import midasmednet.segmentation as rseg
    import midasmednet.landmarks as rldm
    return rseg, rldm


def proj_vec(name, n):
    g = np.random.Generator(np.random.PCG64(zlib.crc32(("proj:" + name).encode())))
    return g.standard_normal(n)


def summarize(name, t, full_limit):
    a = t.detach().cpu().numpy().astype(np.float32)
    out = {}
    if a.size <= full_limit:
        out["full"] = a
    flat = a.reshape(-1).astype(np.float64)
    out["norm"] = np.float64(np.sqrt((flat * flat).sum()))
    out["sum"] = np.float64(flat.sum())
    out["proj"] = np.float64(flat @ proj_vec(name, flat.size))
    out["head"] = a.reshape(-1)[:64].copy()
    return out


def assert_same(tag, a, b):
    if not torch.equal(a, b):
        d = (a.double() - b.double()).abs().max().item()
        raise SystemExit(f"[make_golden] oracle != reference at {tag}: max|diff|={d:.3e}")


def run_net(model, batch, loss_kind, loss_args):
    model.zero_grad(set_to_none=True)
    x = batch["data"].float()
    y = batch["label"][:, -1, ...].long()
    logits = model(x)
    extra = {}
    if loss_kind == "dice":
        loss = loss_args["fn"](logits, y)
    elif loss_kind == "ce":
        loss = loss_args["fn"](logits, y)
    elif loss_kind == "ldmk":
        hm = batch["label"][:, :-1, ...].float()
        nh = hm.shape[1]
        loss, cl, rg = loss_args["fn"](logits[:, nh:, ...], logits[:, :nh, ...], y, hm)
        extra = {"class_loss": cl.detach(), "regression_loss": rg.detach()}
    else:
        raise ValueError(loss_kind)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    return logits.detach(), loss.detach(), grads, extra


def net_case(out, tag, ref_cls, ora_cls, ctor, shape, n, n_classes, n_heatmaps, loss_kind, weight,
             full_limit, rloss, stride=None, adam=False, regression="L2", seed=1234):
    torch.manual_seed(0)
    ref = O.keyed_init_(ref_cls(**ctor))
    ora = O.keyed_init_(ora_cls(**ctor))
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), tag
    for (ka, va), (kb, vb) in zip(ref.state_dict().items(), ora.state_dict().items()):
        assert va.shape == vb.shape and torch.equal(va, vb), (tag, ka)
    batch = O.synthetic_batch(n, ctor["in_channels"], shape, n_classes, n_heatmaps, seed=seed)
    w = None if weight is None else torch.tensor(weight)
    if loss_kind == "dice":
        la_r = {"fn": rloss.DiceLoss(weight=w)}
        la_o = {"fn": O.DiceLoss(weight=w)}
    elif loss_kind == "ce":
        la_r = {"fn": nn.CrossEntropyLoss(weight=w)}
        la_o = {"fn": nn.CrossEntropyLoss(weight=w)}
    else:
        regw = [0.015] * n_heatmaps
        reg = nn.MSELoss() if regression == "L2" else nn.L1Loss()

        def ref_ldmk(ol, oh, y, hm, _c=rloss.DiceLoss(weight=w)):
            # landmarks.py:125-134 restated inline on the REFERENCE's DiceLoss (LandmarkNet needs hparams)
            cl = _c(ol, y)
            rg = torch.tensor(0.0).type_as(ol)
            for c in range(len(regw)):
                rg += regw[c] * reg(oh[:, c, ...], hm[:, c, ...])
            return rg + cl, cl, rg

        la_r = {"fn": ref_ldmk}
        la_o = {"fn": lambda ol, oh, y, hm: O.landmark_loss(ol, oh, y, hm, O.DiceLoss(weight=w), reg, regw)}
    lr, lossr, gr, er = run_net(ref, batch, loss_kind, la_r)
    lo, losso, go, eo = run_net(ora, batch, loss_kind, la_o)
    assert_same(tag + ":logits", lr, lo)
    assert_same(tag + ":loss", lossr, losso)
    for k in gr:
        assert_same(tag + ":grad:" + k, gr[k], go[k])
    rec = {"meta.shape": np.array(shape), "meta.n": np.array(n), "meta.seed": np.array(seed),
           "loss": lossr.numpy().astype(np.float64)}
    for k, v in er.items():
        rec[k] = v.numpy().astype(np.float64)
    if stride is not None:
        rec["logits.strided"] = lr[..., ::stride, ::stride, ::stride].numpy().copy()
        rec["meta.stride"] = np.array(stride)
    for k, v in summarize("logits", lr, full_limit).items():
        rec["logits." + k] = v
    for name, g in gr.items():
        for k, v in summarize(name, g, full_limit).items():
            rec[f"grad.{name}.{k}"] = v
    if adam:
        # segmentation.py:119-120: Adam(lr) defaults; one step's parameter delta
        opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3)
        opt_o = torch.optim.Adam(ora.parameters(), lr=1e-3)
        before = {k: p.detach().clone() for k, p in ref.named_parameters()}
        opt_r.step()
        opt_o.step()
        for (k, pr), (_, po) in zip(ref.named_parameters(), ora.named_parameters()):
            assert_same(tag + ":adam:" + k, pr.detach(), po.detach())
            rec[f"adam_delta.{k}"] = (pr.detach() - before[k]).numpy()
    np.savez_compressed(os.path.join(out, tag + ".npz"), **rec)
    print(f"[make_golden] {tag}: loss={float(lossr):.8f} ok ({len(gr)} grads bit-identical)")


def block_cases(out, rcomp):
    """Per-block vectors: forward output + grads of input and params for a scalar sum(y*r) objective."""
    rec = {}

    def run(tag, ref_m, ora_m, inputs):
        O.keyed_init_(ref_m)
        O.keyed_init_(ora_m)
        res = []
        for m in (ref_m, ora_m):
            xs = [t.clone().requires_grad_(True) for t in inputs]
            y = m(*xs)
            g = torch.from_numpy(O._rng("cot:" + tag).standard_normal(tuple(y.shape)).astype(np.float32))
            (y * g).sum().backward()
            res.append((y.detach(), [t.grad.clone() for t in xs],
                        {k: p.grad.clone() for k, p in m.named_parameters()}))
        (yr, gxr, gpr), (yo, gxo, gpo) = res
        assert_same(tag + ":y", yr, yo)
        for i, (a, b) in enumerate(zip(gxr, gxo)):
            assert_same(f"{tag}:dx{i}", a, b)
        for k in gpr:
            assert_same(f"{tag}:d{k}", gpr[k], gpo[k])
        rec[f"{tag}.y"] = yr.numpy()
        for i, a in enumerate(gxr):
            rec[f"{tag}.dx{i}"] = a.numpy()
        for k, a in gpr.items():
            rec[f"{tag}.dp.{k}"] = a.numpy()
        print(f"[make_golden] block {tag}: ok")

    def rnd(tag, *shape):
        return torch.from_numpy(O._rng("in:" + tag).standard_normal(shape).astype(np.float32))

    for order in ["cge", "gcr", "cg", "cr", "cl", "ce", "crg", "bcr", "cbe"]:
        tag = f"single_{order}"
        run(tag, rcomp.SingleConv(8, 16, 3, order, 8), O.SingleConv(8, 16, 3, order, 8), [rnd(tag, 2, 8, 6, 10, 12)])
    # num_groups fallback (C < groups -> 1 group), components.py:53-54
    run("single_cge_c4", rcomp.SingleConv(4, 4, 3, "cge", 8), O.SingleConv(4, 4, 3, "cge", 8),
        [rnd("single_cge_c4", 1, 4, 5, 6, 7)])
    run("double_enc_gcr", rcomp.DoubleConv(8, 32, True, 3, "gcr", 8), O.DoubleConv(8, 32, True, 3, "gcr", 8),
        [rnd("double_enc", 1, 8, 8, 8, 8)])
    run("double_dec_gcr", rcomp.DoubleConv(24, 8, False, 3, "gcr", 8), O.DoubleConv(24, 8, False, 3, "gcr", 8),
        [rnd("double_dec", 1, 24, 8, 8, 8)])
    for order in ["cge", "cgr", "cgl"]:
        tag = f"resblock_{order}"
        run(tag, rcomp.ExtResNetBlock(8, 16, order=order), O.ExtResNetBlock(8, 16, order=order),
            [rnd(tag, 2, 8, 6, 8, 10)])
    run("encoder_res", rcomp.Encoder(8, 16, basic_module=rcomp.ExtResNetBlock, conv_layer_order="cge"),
        O.Encoder(8, 16, basic_module=O.ExtResNetBlock, conv_layer_order="cge"), [rnd("encoder_res", 2, 8, 8, 12, 10)])
    run("encoder_res_oddpool", rcomp.Encoder(8, 8, basic_module=rcomp.ExtResNetBlock, conv_layer_order="cge"),
        O.Encoder(8, 8, basic_module=O.ExtResNetBlock, conv_layer_order="cge"), [rnd("encoder_odd", 1, 8, 7, 9, 11)])
    run("encoder_double", rcomp.Encoder(8, 16, basic_module=rcomp.DoubleConv, conv_layer_order="gcr"),
        O.Encoder(8, 16, basic_module=O.DoubleConv, conv_layer_order="gcr"), [rnd("encoder_double", 1, 8, 8, 8, 8)])
    run("decoder_res", rcomp.Decoder(16, 8, basic_module=rcomp.ExtResNetBlock, conv_layer_order="cge"),
        O.Decoder(16, 8, basic_module=O.ExtResNetBlock, conv_layer_order="cge"),
        [rnd("decoder_res_e", 2, 8, 8, 12, 10), rnd("decoder_res_x", 2, 16, 4, 6, 5)])
    run("decoder_double", rcomp.Decoder(24, 8, basic_module=rcomp.DoubleConv, conv_layer_order="gcr"),
        O.Decoder(24, 8, basic_module=O.DoubleConv, conv_layer_order="gcr"),
        [rnd("decoder_double_e", 1, 8, 7, 9, 10), rnd("decoder_double_x", 1, 16, 3, 4, 5)])

    class UpOnly(nn.Module):  # ConvTranspose3d alone, components.py:259-264
        def __init__(self):
            super().__init__()
            self.upsample = nn.ConvTranspose3d(16, 8, kernel_size=3, stride=(2, 2, 2), padding=1, output_padding=1)

        def forward(self, x):
            return self.upsample(x)

    run("convtranspose", UpOnly(), UpOnly(), [rnd("convtranspose", 2, 16, 3, 5, 4)])
    np.savez_compressed(os.path.join(out, "blocks.npz"), **rec)


def loss_cases(out, rloss):
    rec = {}
    g = np.random.Generator(np.random.PCG64(77))
    n, c, shp = 2, 4, (6, 8, 10)
    logits_np = (g.standard_normal((n, c) + shp) * 2).astype(np.float32)
    labels_np = g.integers(0, c, size=(n,) + shp).astype(np.int64)
    rec["logits"] = logits_np
    rec["labels"] = labels_np

    def both(tag, fr, fo, needs_grad=True):
        res = []
        for f in (fr, fo):
            z = torch.from_numpy(logits_np).clone().requires_grad_(needs_grad)
            v = f(z, torch.from_numpy(labels_np))
            gz = None
            if needs_grad and v.dim() == 0:
                v.backward()
                gz = z.grad.clone()
            res.append((v.detach(), gz))
        (vr, gr_), (vo, go_) = res
        assert_same(tag + ":value", vr, vo)
        rec[tag + ".value"] = vr.numpy()
        if gr_ is not None:
            assert_same(tag + ":grad", gr_, go_)
            rec[tag + ".grad"] = gr_.numpy()
        print(f"[make_golden] loss {tag}: ok")

    w = torch.tensor([0.05, 1.0, 1.0, 1.0])
    both("dice_plain", rloss.DiceLoss(), O.DiceLoss())
    both("dice_weight", rloss.DiceLoss(weight=w), O.DiceLoss(weight=w))
    both("dice_sigmoid", rloss.DiceLoss(weight=w, sigmoid_normalization=True),
         O.DiceLoss(weight=w, sigmoid_normalization=True))
    both("dice_ignore", rloss.DiceLoss(weight=w, ignore_index=1), O.DiceLoss(weight=w, ignore_index=1))
    both("dice_eps", rloss.DiceLoss(epsilon=1e-2), O.DiceLoss(epsilon=1e-2))
    both("dice_metric", rloss.dice_metric, O.dice_metric, needs_grad=False)
    both("ce_weight", nn.CrossEntropyLoss(weight=w), nn.CrossEntropyLoss(weight=w))
    both("ce_plain", nn.CrossEntropyLoss(), nn.CrossEntropyLoss())
    both("wce", lambda z, y: rloss.WeightedCrossEntropyLoss(target_one_hot_encoded=False)(z, y),
         lambda z, y: O.WeightedCrossEntropyLoss(target_one_hot_encoded=False)(z, y))
    both("wce_onehot", lambda z, y: rloss.WeightedCrossEntropyLoss(weight=w)(z, rloss.expand_as_one_hot(y, 4)),
         lambda z, y: O.WeightedCrossEntropyLoss(weight=w)(z, O.expand_as_one_hot(y, 4)))
    both("celoss", lambda z, y: rloss.CELoss()(z, y.unsqueeze(1)), lambda z, y: O.CELoss()(z, y.unsqueeze(1)))
    # skip_last_target: 5 target classes, 4 logits channels
    lab5 = g.integers(0, 5, size=(n,) + shp).astype(np.int64)
    rec["labels5"] = lab5
    # the reference builds the one-hot with C = logits channels (loss.py:122) so labels must stay < C; the
    # skip_last_target path then drops target channel C-1 and needs logits with C-1 channels -> shape assert.
    try:
        rloss.DiceLoss(skip_last_target=True)(torch.from_numpy(logits_np), torch.from_numpy(labels_np))
        rec["dice_skip_last.raises"] = np.array(0)
    except AssertionError:
        rec["dice_skip_last.raises"] = np.array(1)
    try:
        O.DiceLoss(skip_last_target=True)(torch.from_numpy(logits_np), torch.from_numpy(labels_np))
        assert rec["dice_skip_last.raises"] == 0
    except AssertionError:
        assert rec["dice_skip_last.raises"] == 1
    # one-hot with and without ignore_index
    lab_t = torch.from_numpy(labels_np)
    for tag, ii in (("onehot", None), ("onehot_ignore", 2)):
        a = rloss.expand_as_one_hot(lab_t, 4, ignore_index=ii)
        b = O.expand_as_one_hot(lab_t, 4, ignore_index=ii)
        assert_same(tag, a, b)
        rec[tag] = a.numpy()
    # BCE wrapper
    tgt = torch.from_numpy(g.integers(0, 2, size=(n, c) + shp).astype(np.float32))
    rec["bce_target"] = tgt.numpy()
    for tag, ii in (("bce_wrap", -1), ("bce_wrap_ignore1", 1)):
        res = []
        for cls in (rloss.BCELossWrapper, O.BCELossWrapper):
            z = torch.from_numpy(logits_np).clone().requires_grad_(True)
            v = cls(nn.BCEWithLogitsLoss(), ignore_index=ii)(z, tgt)
            v.backward()
            res.append((v.detach(), z.grad.clone()))
        assert_same(tag, res[0][0], res[1][0])
        assert_same(tag + ":g", res[0][1], res[1][1])
        rec[tag + ".value"] = res[0][0].numpy()
        rec[tag + ".grad"] = res[0][1].numpy()
    # pixel-wise CE (N=1 only: loss.py:218-219 expands a per-sample weight map)
    z1 = torch.from_numpy(logits_np[:1])
    y1 = torch.from_numpy(labels_np[:1])
    wmap = torch.from_numpy(g.uniform(0.5, 2.0, size=(1,) + shp).astype(np.float32))
    rec["pwce_weights"] = wmap.numpy()
    res = []
    for cls in (rloss.PixelWiseCrossEntropyLoss, O.PixelWiseCrossEntropyLoss):
        z = z1.clone().requires_grad_(True)
        v = cls(class_weights=w)(z, y1, wmap)
        v.backward()
        res.append((v.detach(), z.grad.clone()))
    assert_same("pwce", res[0][0], res[1][0])
    assert_same("pwce:g", res[0][1], res[1][1])
    rec["pwce.value"] = res[0][0].numpy()
    rec["pwce.grad"] = res[0][1].numpy()
    # LandmarkLoss (plain MSE) + the LandmarkNet.loss arithmetic (L2 and L1), 3 heatmaps + 2 classes
    hm = torch.from_numpy(g.integers(0, 256, size=(n, 3) + shp).astype(np.float32))
    rec["heatmaps"] = hm.numpy()
    lab2 = torch.from_numpy(g.integers(0, 2, size=(n,) + shp).astype(np.int64))
    rec["labels2"] = lab2.numpy()
    out5 = torch.from_numpy((g.standard_normal((n, 5) + shp) * 3).astype(np.float32))
    rec["logits5"] = out5.numpy()
    assert_same("landmarkloss", rloss.LandmarkLoss()(out5[:, :3], hm), O.LandmarkLoss()(out5[:, :3], hm))
    rec["landmarkloss.value"] = rloss.LandmarkLoss()(out5[:, :3], hm).numpy()
    regw = [0.001, 0.015, 0.02]
    w2 = torch.tensor([0.05, 1.0])
    for tag, reg in (("ldmk_l2", nn.MSELoss()), ("ldmk_l1", nn.L1Loss())):
        res = []
        for dice_cls in (rloss.DiceLoss, O.DiceLoss):
            z = out5.clone().requires_grad_(True)
            tot, cl, rg = O.landmark_loss(z[:, 3:], z[:, :3], lab2, hm, dice_cls(weight=w2), reg, regw)
            tot.backward()
            res.append((tot.detach(), cl.detach(), rg.detach(), z.grad.clone()))
        for a, b in zip(res[0], res[1]):
            assert_same(tag, a, b)
        rec[tag + ".value"] = res[0][0].numpy()
        rec[tag + ".class"] = res[0][1].numpy()
        rec[tag + ".reg"] = res[0][2].numpy()
        rec[tag + ".grad"] = res[0][3].numpy()
        print(f"[make_golden] loss {tag}: ok")
    np.savez_compressed(os.path.join(out, "losses.npz"), **rec)


def caller_cases(out):
    """SegmentationNet / LandmarkNet training_step on the reference (segmentation.py:58-65, landmarks.py:66-83)."""
    rseg, rldm = import_reference_callers()
    rec = {}
    hp = types.SimpleNamespace(in_channels=1, out_channels=2, fmaps=[8], learning_rate=1e-3, num_workers=0,
                               batch_size=2, loss="DICE", loss_weight=[0.05, 1.0])
    net = O.keyed_init_(rseg.SegmentationNet(hp))
    batch = O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234)
    res = net.training_step(batch, 0)
    ora = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8]))
    lo = O.seg_training_step(ora, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), batch)
    assert_same("seg_step", res["loss"].detach(), lo.detach())
    assert sorted(res.keys()) == ["log", "loss"] and list(res["log"].keys()) == ["train_loss"]
    rec["seg.loss"] = res["loss"].detach().numpy()
    opt = net.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam)
    d = opt.defaults
    rec["seg.adam"] = np.array([d["lr"], d["betas"][0], d["betas"][1], d["eps"], d["weight_decay"]])
    # row N3: validation_step / validation_epoch_end (segmentation.py:94-118); log_interval keeps the plotting branch off
    net.log_interval = 10 ** 9
    net.eval()
    vb = [O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=600 + i) for i in range(2)]
    with torch.no_grad():
        vres = [net.validation_step(b, 1 + i) for i, b in enumerate(vb)]
    vora = [O.seg_validation_step(ora.eval(), O.DiceLoss(weight=torch.tensor([0.05, 1.0])), b) for b in vb]
    for a, b in zip(vres, vora):
        assert sorted(a.keys()) == sorted(b.keys()) == ["val_dice0", "val_dice1", "val_loss"]
        for k in a:
            assert_same("seg_val." + k, a[k].detach(), b[k].detach())
    vend, oend = net.validation_epoch_end(vres), O.validation_epoch_end(vora)
    for k in oend:
        assert_same("seg_val_end." + k, vend["log"][k], oend[k])
        rec["seg.val." + k] = oend[k].numpy()
    net.train()
    hp2 = types.SimpleNamespace(in_channels=1, out_channels=5, fmaps=[8], learning_rate=1e-3, num_workers=0,
                                batch_size=2, loss_class="DICE", loss_class_weight=[0.05, 1.0],
                                loss_regression="L2", loss_regression_weight=[0.015, 0.015, 0.015])
    net2 = O.keyed_init_(rldm.LandmarkNet(hp2))
    batch2 = O.synthetic_batch(2, 1, (16, 16, 16), 2, 3, seed=4321)
    res2 = net2.training_step(batch2, 0)
    ora2 = O.keyed_init_(O.ResidualUNet3D(1, 5, False, f_maps=[8]))
    tot, cl, rg = O.ldmk_training_step(ora2, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), nn.MSELoss(),
                                       [0.015] * 3, batch2)
    assert_same("ldmk_step", res2["loss"].detach(), tot.detach())
    assert sorted(res2["log"].keys()) == ["class_loss", "regression_loss", "train_loss"]
    rec["ldmk.loss"] = res2["loss"].detach().numpy()
    rec["ldmk.class_loss"] = np.float64(res2["log"]["class_loss"])
    rec["ldmk.regression_loss"] = np.float64(res2["log"]["regression_loss"])
    np.savez_compressed(os.path.join(out, "callers.npz"), **rec)
    print(f"[make_golden] callers: seg loss={float(res['loss']):.8f} ldmk loss={float(res2['loss']):.6f} ok")


def predict_cases(out):
    """Row N2: the reference's grid_patch_generator / GridPatchSampler.add_processed_batch (dataset.py) and
    the post-processing lines of examples/predict.py, run on deterministic inputs; oracle/ref_predict.py must agree
    exactly.  h5py / zarr / nibabel are absent here: dataset.py imports them at module level only, so they are stubbed
    (zarr.group() -> a dict of numpy-backed datasets, which is all add_processed_batch uses)."""
    import torch.nn.functional as F
    from oracle import ref_predict as P
    PREDICT_CASES, predict_inputs = P.PREDICT_CASES, P.predict_inputs

    class _DS:
        def __init__(self, shape, dtype):
            self.a, self.attrs = np.zeros(tuple(int(v) for v in shape), dtype=dtype), {}

        def __setitem__(self, k, v):
            self.a[k] = v

        def __getitem__(self, k):
            return self.a[k]

    class _Group(dict):
        def require_dataset(self, key, shape, dtype, chunks=False):
            if key not in self:
                self[key] = _DS(shape, dtype)
            return self[key]

    for name in ["nibabel", "h5py", "zarr"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["zarr"].group = _Group
    sys.path.insert(0, REF)
    import midasmednet.dataset as rds

    rec = {}
    for tag, shape, patch, ov, mode, nh, ncls, bs in PREDICT_CASES:
        img, logits_for = predict_inputs(tag, shape, patch, nh, ncls)
        ref_p = list(rds.grid_patch_generator(img, patch, ov, mode=mode))
        ora_p = list(P.grid_patch_generator(img, patch, ov, mode=mode))
        assert len(ref_p) == len(ora_p) > 0
        for (a, ia, ca), (b, ib, cb) in zip(ref_p, ora_p):
            assert np.array_equal(a, b) and np.array_equal(ia, ib) and ca == cb, tag
        # the reference sampler without its file readers: only what add_processed_batch touches (dataset.py:446-474)
        gs = object.__new__(rds.GridPatchSampler)
        gs.patch_overlap, gs.out_channels, gs.out_dtype = ov, nh + 1, np.uint8
        gs.data_shape, gs.data_affine, gs.results = {"s": np.array(shape)}, {"s": np.eye(4)}, _Group()
        result_o = np.zeros((nh + 1,) + tuple(shape[1:]), dtype=np.uint8)
        for b0 in range(0, len(ref_p), bs):
            chunk = ref_p[b0:b0 + bs]
            logits = torch.from_numpy(np.stack([logits_for(c) for _, _, c in chunk]))
            # examples/predict.py:88-95, verbatim semantics
            oc = torch.argmax(F.softmax(logits[:, nh:, ...], dim=1), dim=1, keepdim=True).numpy()
            oh = np.clip(logits[:, :nh, ...].numpy(), 0.0, 255.0)
            output = np.concatenate([oh.astype(np.uint8), oc.astype(np.uint8)], axis=1)
            assert np.array_equal(output, P.postprocess(logits.numpy(), nh)), tag
            pos = np.stack([i for _, i, _ in chunk])
            gs.add_processed_batch({"subject_key": ["s"] * len(chunk), "pos": pos, "data": output})
            P.add_processed_batch(result_o, output, pos, ov)
        result_r = gs.results["s"].a
        assert np.array_equal(result_r, result_o), tag
        rec[f"{tag}.result"] = result_r
        rec[f"{tag}.npatches"] = np.array(len(ref_p))
        rec[f"{tag}.pos"] = np.stack([i for _, i, _ in ref_p])
        rec[f"{tag}.patch_sums"] = np.array([float(np.asarray(a, dtype=np.float64).sum()) for a, _, _ in ref_p])
        rec[f"{tag}.first_patch"] = np.asarray(ref_p[0][0])
        rec[f"{tag}.last_patch"] = np.asarray(ref_p[-1][0])
        print(f"predict[{tag}]: {len(ref_p)} patches, result {result_r.shape} bit-identical reference == oracle")
    np.savez_compressed(os.path.join(out, "predict.npz"), **rec)


def sampler_cases(out):
    """Row N1: the reference's MedDataset.__getitem__ / get_labeled_position / get_random_patch_indices
    (dataset.py) with numpy's global generator seeded, against oracle/ref_sampler.py seeded the same way.  The reference
    uses `np.int`, which numpy 2 no longer has: it is restored as the builtin int for the import.  MedDataset's readers
    are bypassed (object.__new__ + the attributes __getitem__ touches): the volumes are given in memory."""
    from oracle import ref_sampler as S
    if not hasattr(np, "int"):
        np.int = int
    for name in ["nibabel", "h5py", "zarr"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import midasmednet.dataset as rds

    rec = {}
    for tag, shapes, c_img, n_hm, patch, probs, draws, seed in S.SAMPLER_CASES:
        n_classes = len(probs) if probs else 3
        images, labels, heatmaps = S.sampler_volumes(tag, shapes, c_img, n_hm, n_classes)
        ora = S.PatchSampler(images, labels, patch, samples_per_subject=4, heatmaps=heatmaps, class_probabilities=probs)
        ref = object.__new__(rds.MedDataset)
        ref.images, ref.labels, ref.transform = images, labels, None
        ref.heatmap_group = "heatmaps" if heatmaps is not None else None
        if heatmaps is not None:
            ref.heatmaps = heatmaps
        ref.patch_size = np.array(patch, dtype=int)
        ref.subject_keys = [str(i) for i in range(len(images))]
        ref.samples_per_subject = 4
        ref.class_probabilities = None if probs is None else probs / np.sum(probs)
        ref._label_ax2_any = []
        if probs:
            for idx in range(len(labels)):
                ref._label_ax2_any.append([np.any(labels[idx][-1, ...] == c, axis=2) for c in range(len(probs))])
        np.random.seed(seed)
        r_items = [ref[i] for i in range(draws)]
        np.random.seed(seed)
        o_items = [ora[i] for i in range(draws)]
        for a, b in zip(r_items, o_items):
            assert a["subject_key"] == b["subject_key"] and int(a["selected_class"]) == int(b["selected_class"]), tag
            assert np.array_equal(a["patch_position"], b["patch_position"]), tag
            assert a["data"].dtype == b["data"].dtype == np.float32 and np.array_equal(a["data"], b["data"]), tag
            assert a["label"].dtype == b["label"].dtype == np.uint8 and np.array_equal(a["label"], b["label"]), tag
        rec[f"{tag}.pos"] = np.stack([a["patch_position"] for a in r_items])
        rec[f"{tag}.cls"] = np.array([int(a["selected_class"]) for a in r_items])
        rec[f"{tag}.subj"] = np.array([int(a["subject_key"]) for a in r_items])
        rec[f"{tag}.data_sum"] = np.array([float(a["data"].astype(np.float64).sum()) for a in r_items])
        rec[f"{tag}.label_sum"] = np.array([int(a["label"].astype(np.int64).sum()) for a in r_items])
        rec[f"{tag}.first_data"], rec[f"{tag}.first_label"] = r_items[0]["data"], r_items[0]["label"]
        print(f"sampler[{tag}]: {draws} draws bit-identical reference == oracle; classes {sorted(set(rec[f'{tag}.cls'].tolist()))}")
    np.savez_compressed(os.path.join(out, "sampler.npz"), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--skip-large", action="store_true", help="skip the 128^3 and cfg5 cases")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    rmodel, rcomp, rloss = import_reference()
    assert rmodel.create_feature_maps(32, 5) == O.create_feature_maps(32, 5) == [32, 64, 128, 256, 512]

    def want(tag):
        return a.only is None or a.only in tag

    if want("blocks"):
        block_cases(a.out, rcomp)
    if want("losses"):
        loss_cases(a.out, rloss)
    if want("callers"):
        caller_cases(a.out)
    if want("predict"):
        predict_cases(a.out)
    if want("sampler"):
        sampler_cases(a.out)
    R, U = (rmodel.ResidualUNet3D, O.ResidualUNet3D), (rmodel.UNet3D, O.UNet3D)
    seg_w4 = [0.05, 1.0, 1.0, 1.0]
    nets = [
        # tag, classes, ctor, shape, n, n_classes, n_heatmaps, loss, weight, full_limit, stride, adam, large
        ("res_cfg1", R, dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8]), (32, 32, 32), 2, 2, 0,
         "dice", [0.05, 1.0], 1 << 20, None, True, False),
        ("res_cfg1_ce", R, dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8]), (32, 32, 32), 2, 2,
         0, "ce", [0.05, 1.0], 1 << 20, None, False, False),
        ("res_small", R, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64]), (16, 16, 16), 2,
         4, 0, "dice", seg_w4, 1 << 18, None, False, False),
        ("res_odd", R, dict(in_channels=2, out_channels=3, final_sigmoid=False, f_maps=[8, 16, 24]), (12, 20, 8), 1,
         3, 0, "dice", None, 1 << 18, None, False, False),
        ("res_cfg2_32", R, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]),
         (32, 32, 32), 1, 4, 0, "dice", seg_w4, 1 << 18, None, False, False),
        ("res_cfg4_32", R, dict(in_channels=1, out_channels=18, final_sigmoid=False, f_maps=[32, 64, 128, 256]),
         (32, 32, 32), 1, 2, 16, "ldmk", [0.05, 1.0], 1 << 16, None, False, False),
        ("res_ldmk_l1", R, dict(in_channels=1, out_channels=5, final_sigmoid=False, f_maps=[8, 16]), (16, 16, 16), 2,
         2, 3, "ldmk", [0.05, 1.0], 1 << 18, None, False, False),
        ("unet_cfg1", U, dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8]), (32, 32, 32), 2, 2, 0,
         "dice", [0.05, 1.0], 1 << 20, None, False, False),
        ("unet_small", U, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64]), (16, 16, 16), 2,
         4, 0, "dice", seg_w4, 1 << 18, None, False, False),
        ("unet_oddsize", U, dict(in_channels=1, out_channels=3, final_sigmoid=False, f_maps=[8, 16, 32]),
         (13, 10, 9), 1, 3, 0, "dice", None, 1 << 18, None, False, False),
        ("unet_cfg2_32", U, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]),
         (32, 32, 32), 1, 4, 0, "dice", seg_w4, 1 << 18, None, False, False),
        ("res_cfg2_128", R, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]),
         (128, 128, 128), 1, 4, 0, "dice", seg_w4, 0, 16, False, True),
        ("res_cfg4_128", R, dict(in_channels=1, out_channels=18, final_sigmoid=False, f_maps=[32, 64, 128, 256]),
         (128, 128, 128), 1, 2, 16, "ldmk", [0.05, 1.0], 0, 16, False, True),
        ("res_cfg5_small", R, dict(in_channels=1, out_channels=4, final_sigmoid=False,
                                   f_maps=[64, 128, 256, 512, 1024]), (32, 32, 16), 1, 4, 0, "dice", seg_w4, 0, 8,
         False, True),
        # BASELINE config 5 at its real size (one sample: the CPU run peaks at ~30 GB and takes minutes)
        ("res_cfg5_full", R, dict(in_channels=1, out_channels=4, final_sigmoid=False,
                                  f_maps=[64, 128, 256, 512, 1024]), (160, 160, 96), 1, 4, 0, "dice", seg_w4, 0, 16,
         False, True),
    ]
    for (tag, (rc, oc), ctor, shape, n, ncls, nh, lk, w, fl, stride, adam, large) in nets:
        if not want(tag) or (large and a.skip_large):
            continue
        reg = "L1" if tag == "res_ldmk_l1" else "L2"
        net_case(a.out, tag, rc, oc, ctor, shape, n, ncls, nh, lk, w, fl, rloss, stride=stride, adam=adam,
                 regression=reg)
    print("[make_golden] done")


if __name__ == "__main__":
    main()
