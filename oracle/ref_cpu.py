"""CPU oracle for the 3D U-Net hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

This file is a plain ``torch.nn`` (ATen, fp32, CPU) restatement of the algorithm that
tobiashepp/torch-mednet runs for the path named in BASELINE.json:

* network:    midasmednet/unet/model.py:7-8 (feature maps), :36-110 (UNet3D), :140-214 (ResidualUNet3D)
* blocks:     midasmednet/unet/components.py:12-67 (order grammar), :70-90 (SingleConv), :93-133 (DoubleConv),
              :136-180 (ExtResNetBlock), :183-226 (Encoder), :229-287 (Decoder)
* losses:     midasmednet/unet/loss.py:10-21 (flatten), :24-48 (per-channel dice), :51-55 (dice_metric),
              :58-88 (one-hot), :91-130 (DiceLoss); midasmednet/segmentation.py:43-49 (CE), landmarks.py:125-134
* step:       midasmednet/segmentation.py:58-65, landmarks.py:66-83, segmentation.py:119-120 (Adam)

The arithmetic itself lives in ATen (the reference calls torch.nn.Conv3d/GroupNorm/ELU/...), so the
oracle calls the same ATen ops in the same order; module tree and parameter names are identical so a
state_dict moves between the reference, this oracle and the HIP product unchanged.

Pinning: ``tools/make_golden.py`` imports the real reference from /root/reference (with a 3-line
pytorch_lightning stub), runs both on identical inputs/weights and asserts BIT-equality of logits, loss and
every gradient before writing ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks the oracle
against those committed vectors without the reference being present.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(`mednet_hip`) never does: it fails loudly when its HIP library is missing.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- layer grammar
_ACTS = {
    "r": ("ReLU", lambda: nn.ReLU(inplace=True)),
    "l": ("LeakyReLU", lambda: nn.LeakyReLU(negative_slope=0.1, inplace=True)),
    "e": ("ELU", lambda: nn.ELU(inplace=True)),
}


def layer_list(cin, cout, ksize, order, groups, padding=1):
    """(name, module) pairs for one order string; components.py:12-67."""
    if "c" not in order:
        raise AssertionError("Conv layer MUST be present")
    if order[0] in "rle":
        raise AssertionError("Non-linearity cannot be the first operation in the layer")
    conv_at = order.index("c")
    has_norm = ("g" in order) or ("b" in order)
    out = []
    for pos, ch in enumerate(order):
        if ch in _ACTS:
            name, make = _ACTS[ch]
            out.append((name, make()))
        elif ch == "c":
            out.append(("conv", nn.Conv3d(cin, cout, ksize, padding=padding, bias=not has_norm)))
        elif ch == "g":
            nch = cin if pos < conv_at else cout
            g = 1 if nch < groups else groups
            if nch % g:
                raise AssertionError(
                    f"Expected number of channels in input to be divisible by num_groups. "
                    f"num_channels={nch}, num_groups={g}")
            out.append(("groupnorm", nn.GroupNorm(num_groups=g, num_channels=nch)))
        elif ch == "b":
            out.append(("batchnorm", nn.BatchNorm3d(cin if pos < conv_at else cout)))
        else:
            raise ValueError(f"Unsupported layer type '{ch}'. MUST be one of ['b', 'g', 'r', 'l', 'e', 'c']")
    return out


class SingleConv(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, order="crg", num_groups=8, padding=1):
        super().__init__(OrderedDict(layer_list(cin, cout, kernel_size, order, num_groups, padding)))


class DoubleConv(nn.Sequential):
    def __init__(self, cin, cout, encoder, kernel_size=3, order="crg", num_groups=8):
        if encoder:
            mid = max(cout // 2, cin)
            a, b = (cin, mid), (mid, cout)
        else:
            a, b = (cin, cout), (cout, cout)
        super().__init__(OrderedDict([
            ("SingleConv1", SingleConv(a[0], a[1], kernel_size, order, num_groups)),
            ("SingleConv2", SingleConv(b[0], b[1], kernel_size, order, num_groups)),
        ]))


class ExtResNetBlock(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, order="cge", num_groups=8, **_):
        super().__init__()
        bare = "".join(c for c in order if c not in "rel")
        self.conv1 = SingleConv(cin, cout, kernel_size, order, num_groups)
        self.conv2 = SingleConv(cout, cout, kernel_size, order, num_groups)
        self.conv3 = SingleConv(cout, cout, kernel_size, bare, num_groups)
        key = "l" if "l" in order else ("e" if "e" in order else "r")
        self.non_linearity = _ACTS[key][1]()

    def forward(self, x):
        skip = self.conv1(x)
        y = self.conv3(self.conv2(skip))
        y += skip
        return self.non_linearity(y)


class Encoder(nn.Module):
    def __init__(self, cin, cout, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=(2, 2, 2),
                 pool_type="max", basic_module=DoubleConv, conv_layer_order="crg", num_groups=8):
        super().__init__()
        assert pool_type in ("max", "avg")
        self.pooling = None
        if apply_pooling:
            self.pooling = (nn.MaxPool3d if pool_type == "max" else nn.AvgPool3d)(kernel_size=pool_kernel_size)
        self.basic_module = basic_module(cin, cout, encoder=True, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups)

    def forward(self, x):
        return self.basic_module(x if self.pooling is None else self.pooling(x))


class Decoder(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, scale_factor=(2, 2, 2), basic_module=DoubleConv,
                 conv_layer_order="crg", num_groups=8):
        super().__init__()
        self.upsample = None
        if basic_module is not DoubleConv:
            self.upsample = nn.ConvTranspose3d(cin, cout, kernel_size=kernel_size, stride=scale_factor,
                                               padding=1, output_padding=1)
            cin = cout
        self.basic_module = basic_module(cin, cout, encoder=False, kernel_size=kernel_size,
                                         order=conv_layer_order, num_groups=num_groups)

    def forward(self, encoder_features, x):
        if self.upsample is None:
            x = F.interpolate(x, size=encoder_features.shape[2:], mode="nearest")
            x = torch.cat((encoder_features, x), dim=1)
        else:
            x = self.upsample(x)
            x += encoder_features
        return self.basic_module(x)


def create_feature_maps(init_channel_number, number_of_fmaps):
    return [init_channel_number * 2 ** k for k in range(number_of_fmaps)]


class _UNetBase(nn.Module):
    _block = None
    _int_levels = None

    def _build(self, in_channels, out_channels, f_maps, order, num_groups, concat_join):
        if isinstance(f_maps, int):
            f_maps = create_feature_maps(f_maps, self._int_levels)
        f_maps = list(f_maps)
        enc = []
        for i, f in enumerate(f_maps):
            enc.append(Encoder(in_channels if i == 0 else f_maps[i - 1], f, apply_pooling=i > 0,
                               basic_module=self._block, conv_layer_order=order, num_groups=num_groups))
        self.encoders = nn.ModuleList(enc)
        rev = f_maps[::-1]
        dec = []
        for i in range(len(rev) - 1):
            cin = rev[i] + rev[i + 1] if concat_join else rev[i]
            dec.append(Decoder(cin, rev[i + 1], basic_module=self._block, conv_layer_order=order,
                               num_groups=num_groups))
        self.decoders = nn.ModuleList(dec)
        self.final_conv = nn.Conv3d(f_maps[0], out_channels, 1)

    def forward(self, x):
        feats = []
        for e in self.encoders:
            x = e(x)
            feats.insert(0, x)
        for d, f in zip(self.decoders, feats[1:]):
            x = d(f, x)
        x = self.final_conv(x)
        if self.testing and self.final_activation is not None:
            x = self.final_activation(x)
        return x


class UNet3D(_UNetBase):
    _block = DoubleConv
    _int_levels = 4

    def __init__(self, in_channels, out_channels, final_sigmoid, f_maps=64, layer_order="gcr", num_groups=8,
                 **kwargs):
        super().__init__()
        self.testing = kwargs.get("testing", False)
        self._build(in_channels, out_channels, f_maps, layer_order, num_groups, concat_join=True)
        self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)


class ResidualUNet3D(_UNetBase):
    _block = ExtResNetBlock
    _int_levels = 5

    def __init__(self, in_channels, out_channels, final_sigmoid, f_maps=32, conv_layer_order="cge",
                 num_groups=8, skip_final_activation=False, **kwargs):
        super().__init__()
        self.testing = kwargs.get("testing", False)
        self._build(in_channels, out_channels, f_maps, conv_layer_order, num_groups, concat_join=False)
        if skip_final_activation:
            self.final_activation = None
        else:
            self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)


# --------------------------------------------------------------------------- losses
def flatten(t):
    c = t.size(1)
    perm = (1, 0) + tuple(range(2, t.dim()))
    return t.permute(perm).contiguous().view(c, -1)


def expand_as_one_hot(labels, C, ignore_index=None):
    assert labels.dim() == 4
    idx = labels.unsqueeze(1)
    shape = list(idx.size())
    shape[1] = C
    if ignore_index is None:
        return torch.zeros(shape).to(idx.device).scatter_(1, idx, 1)
    mask = idx.expand(shape) == ignore_index
    idx = idx.clone()
    idx[idx == ignore_index] = 0
    out = torch.zeros(shape).to(idx.device).scatter_(1, idx, 1)
    out[mask] = ignore_index
    return out


def compute_per_channel_dice(probs, target, epsilon=1e-5, ignore_index=None, weight=None):
    assert probs.size() == target.size(), "'input' and 'target' must have the same shape"
    if ignore_index is not None:
        keep = target.clone().ne_(ignore_index)
        keep.requires_grad = False
        probs = probs * keep
        target = target * keep
    p = flatten(probs)
    t = flatten(target).float()
    inter = (p * t).sum(-1)
    if weight is not None:
        inter = weight * inter
    denom = (p + t).sum(-1)
    return 2.0 * inter / denom.clamp(min=epsilon)


def dice_metric(logits, labels):
    probs = torch.softmax(logits, dim=1)
    return compute_per_channel_dice(probs, expand_as_one_hot(labels, C=probs.size(1)))


class DiceLoss(nn.Module):
    def __init__(self, epsilon=1e-5, weight=None, ignore_index=None, sigmoid_normalization=False,
                 skip_last_target=False):
        super().__init__()
        self.epsilon = epsilon
        self.register_buffer("weight", weight)
        self.ignore_index = ignore_index
        self.normalization = nn.Sigmoid() if sigmoid_normalization else nn.Softmax(dim=1)
        self.skip_last_target = skip_last_target

    def forward(self, logits, target):
        probs = self.normalization(logits)
        onehot = expand_as_one_hot(target, C=probs.size(1))
        if self.skip_last_target:
            onehot = onehot[:, :-1, ...]
        dice = compute_per_channel_dice(probs, onehot, epsilon=self.epsilon, ignore_index=self.ignore_index,
                                        weight=self.weight)
        return torch.mean(1.0 - dice)


class CELoss(nn.Module):
    """loss.py:135-142 (softmax fed to CrossEntropyLoss; target channel 0)."""

    def __init__(self):
        super().__init__()
        self.ce = nn.CrossEntropyLoss()

    def forward(self, inputs, targets):
        return self.ce(torch.softmax(inputs, dim=1), targets[:, 0, ...])


class WeightedCrossEntropyLoss(nn.Module):
    """loss.py:144-172."""

    def __init__(self, weight=None, ignore_index=-1, target_one_hot_encoded=True):
        super().__init__()
        self.register_buffer("weight", weight)
        self.ignore_index = ignore_index
        self.target_one_hot_encoded = target_one_hot_encoded

    @staticmethod
    def _class_weights(logits):
        p = flatten(F.softmax(logits, dim=1))
        return ((1.0 - p).sum(-1) / p.sum(-1)).detach()

    def forward(self, logits, target):
        cw = self._class_weights(logits)
        if self.weight is not None:
            cw = cw * self.weight
        if self.target_one_hot_encoded:
            target = torch.argmax(target, dim=1)
        return F.cross_entropy(logits, target, weight=cw, ignore_index=self.ignore_index)


class BCELossWrapper:
    """loss.py:175-202."""

    def __init__(self, loss_criterion, ignore_index=-1, skip_last_target=False):
        if hasattr(loss_criterion, "ignore_index"):
            raise RuntimeError(f"Cannot wrap {type(loss_criterion)}. Use 'ignore_index' attribute instead")
        self.loss_criterion = loss_criterion
        self.ignore_index = ignore_index
        self.skip_last_target = skip_last_target

    def __call__(self, logits, target):
        if self.skip_last_target:
            target = target[:, :-1, ...]
        assert logits.size() == target.size()
        if self.ignore_index is None:
            return self.loss_criterion(logits, target)
        keep = target.clone().ne_(self.ignore_index)
        keep.requires_grad = False
        return self.loss_criterion(logits * keep, target * keep)


class PixelWiseCrossEntropyLoss(nn.Module):
    """loss.py:204-241."""

    def __init__(self, class_weights=None, ignore_index=None):
        super().__init__()
        self.register_buffer("class_weights", class_weights)
        self.ignore_index = ignore_index

    def forward(self, logits, target, weights):
        assert target.size() == weights.size()
        logp = F.log_softmax(logits, dim=1)
        onehot = expand_as_one_hot(target, C=logits.size(1), ignore_index=self.ignore_index)
        weights = weights.unsqueeze(0).expand_as(logits)
        if self.ignore_index is not None:
            keep = onehot.detach().ne(self.ignore_index).float()
            logp = logp * keep
            onehot = onehot * keep
        if self.class_weights is None:
            self.register_buffer("class_weights", torch.ones(logits.size(1)).float().to(logits.device))
        weights = self.class_weights.view(1, -1, 1, 1, 1) * weights
        return (-weights * onehot * logp).mean()


class LandmarkLoss(nn.Module):
    """loss.py:243-252 (plain MSE)."""

    def forward(self, logits, heatmaps):
        return F.mse_loss(logits, heatmaps)


def landmark_loss(out_labels, out_heatmaps, labels, heatmaps, loss_class, loss_regression, reg_weights):
    """landmarks.py:125-134: class loss + sum_c w_c * regression(out[:, c], hm[:, c])."""
    class_loss = loss_class(out_labels, labels)
    reg = torch.tensor(0.0).type_as(out_labels)
    for c, w in enumerate(reg_weights):
        reg += w * loss_regression(out_heatmaps[:, c, ...], heatmaps[:, c, ...])
    return reg + class_loss, class_loss, reg


# --------------------------------------------------------------------------- callers' step contract
def seg_training_step(model, loss_fn, batch):
    """segmentation.py:58-65 without the logging dict."""
    x = batch["data"].float()
    y = batch["label"][:, -1, ...].long()
    return loss_fn(model(x), y)


def seg_validation_step(model, loss_fn, batch):
    """segmentation.py:94-110 without the sample logging: {'val_loss', 'val_dice0', ...} (unnormalised network output,
    DiceLoss / CE as configured, dice_metric = unweighted per-channel Dice of the softmax)."""
    x = batch["data"].float()
    y = batch["label"][:, -1, ...].long()
    with torch.no_grad():
        out = model(x)
        res = {"val_loss": loss_fn(out, y)}
        dm = dice_metric(out, y)
    for c in range(out.shape[1]):
        res[f"val_dice{c}"] = dm[c]
    return res


def validation_epoch_end(outputs):
    """segmentation.py:112-118: plain means over the epoch's step results."""
    return {k: torch.stack([o[k] for o in outputs]).mean() for k in outputs[0]}


def ldmk_training_step(model, loss_class, loss_regression, reg_weights, batch):
    """landmarks.py:66-83."""
    x = batch["data"].float()
    hm = batch["label"][:, :-1, ...].float()
    nh = hm.shape[1]
    y = batch["label"][:, -1, ...].long()
    out = model(x)
    return landmark_loss(out[:, nh:, ...], out[:, :nh, ...], y, hm, loss_class, loss_regression, reg_weights)


# --------------------------------------------------------------------------- deterministic, name-keyed data
def _rng(tag: str, seed: int = 0):
    return np.random.Generator(np.random.PCG64((zlib.crc32(tag.encode()) << 8) ^ seed))


def keyed_init_(module: nn.Module, seed: int = 0):
    """Fill every parameter from a PCG64 stream keyed by crc32(parameter name).

    conv / conv-transpose weights: U(-b, b), b = 1/sqrt(fan_in)   (PyTorch's default bound)
    biases: U(-b, b) with the same b;  norm weight: U(0.5, 1.5);  norm bias: U(-0.3, 0.3)
    The same function is used for the reference, the oracle and the HIP modules, so equal names => equal values.
    """
    with torch.no_grad():
        sd = dict(module.named_parameters())
        for name, p in sd.items():
            g = _rng(name, seed)
            if p.dim() == 5:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3] * p.shape[4]
                if "upsample" in name:  # ConvTranspose3d weight is (Cin, Cout, k, k, k)
                    fan_in = p.shape[0] * p.shape[2] * p.shape[3] * p.shape[4]
                b = 1.0 / np.sqrt(fan_in)
                v = g.uniform(-b, b, size=tuple(p.shape))
            elif name.endswith("norm.weight"):
                v = g.uniform(0.5, 1.5, size=tuple(p.shape))
            elif name.endswith("norm.bias"):
                v = g.uniform(-0.3, 0.3, size=tuple(p.shape))
            else:  # conv biases
                w = sd.get(name[:-4] + "weight")
                fan_in = int(np.prod(w.shape[1:])) if w is not None else p.numel()
                if "upsample" in name and w is not None:
                    fan_in = w.shape[0] * int(np.prod(w.shape[2:]))
                b = 1.0 / np.sqrt(max(fan_in, 1))
                v = g.uniform(-b, b, size=tuple(p.shape))
            p.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def synthetic_batch(n, cin, shape, n_classes, n_heatmaps=0, seed=1234):
    """The batch dict MedDataset emits (dataset.py:332-346): data fp32 N x Cin x D x H x W ~ N(0,1);
    label uint8 N x (n_heatmaps+1) x D x H x W, heatmaps uniform 0..255, last channel class ids."""
    g = np.random.Generator(np.random.PCG64(seed))
    data = g.standard_normal((n, cin) + tuple(shape), dtype=np.float32)
    chans = []
    if n_heatmaps:
        chans.append(g.integers(0, 256, size=(n, n_heatmaps) + tuple(shape), dtype=np.uint8))
    chans.append(g.integers(0, n_classes, size=(n, 1) + tuple(shape), dtype=np.uint8))
    label = np.concatenate(chans, axis=1)
    return {"data": torch.from_numpy(data), "label": torch.from_numpy(label)}


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if n > 0 else float(d)
