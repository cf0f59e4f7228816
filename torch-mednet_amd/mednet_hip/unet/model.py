"""UNet3D / ResidualUNet3D with the constructor signatures, attributes and state_dict keys of
midasmednet/unet/model.py, running on the MI355X kernels.

Boundary: forward(x: N x Cin x D x H x W, fp32) -> logits N x out x D x H x W, fp32, NCDHW-contiguous.
The 1x1x1 head writes the logits planar itself; everything between input and head is channels-last.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import config
from .. import nn as hnn
from .. import ops
from .components import Decoder, DoubleConv, Encoder, ExtResNetBlock, SingleConv  # noqa: F401  (re-exported like the reference)

try:  # the reference derives from pl.LightningModule (model.py:11,113); keep that base when it is installed
    import pytorch_lightning as _pl

    _Base = _pl.LightningModule
except Exception:  # pragma: no cover - pytorch_lightning is not part of this image
    _Base = nn.Module


def create_feature_maps(init_channel_number, number_of_fmaps):
    return [init_channel_number * 2 ** k for k in range(number_of_fmaps)]


class _Softmax1(nn.Module):
    """nn.Softmax(dim=1) applied at test time only (model.py:107-108,211-212); off the training path."""

    def forward(self, x):
        return torch.softmax(x, dim=1)


class _CheckpointCompat:
    """What the reference's callers use from pl.LightningModule besides nn.Module when pytorch_lightning is NOT installed
    (it is only the base class, model.py:11,113): `load_from_checkpoint` and `freeze` / `unfreeze` (examples/predict.py:
    47-50), with PL 0.9's semantics -- the class is rebuilt from the checkpoint's `hyper_parameters` (passed as the
    constructor argument named by `hparams_name`, as an argparse.Namespace) and `state_dict` is loaded strictly."""

    CHECKPOINT_HYPER_PARAMS_KEY = "hyper_parameters"
    CHECKPOINT_HYPER_PARAMS_NAME = "hparams_name"

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, *args, map_location=None, **kwargs):
        import argparse
        import inspect
        ckpt = torch.load(checkpoint_path, map_location=map_location if map_location is not None else "cpu",
                          weights_only=False)
        init_args = inspect.getfullargspec(cls.__init__).args[1:]
        hp = ckpt.get(cls.CHECKPOINT_HYPER_PARAMS_KEY, ckpt.get("hparams"))
        if hp is not None:
            hp = dict(vars(hp)) if isinstance(hp, argparse.Namespace) else dict(hp)
            hp.update(kwargs)
            name = ckpt.get(cls.CHECKPOINT_HYPER_PARAMS_NAME)
            if name is None and "hparams" in init_args:
                name = "hparams"
            if name == "kwargs" or name is None:
                kwargs = {k: v for k, v in hp.items() if k in init_args} if name is None else hp
            else:
                kwargs = {name: argparse.Namespace(**hp)}
        model = cls(*args, **kwargs)
        model.load_state_dict(ckpt["state_dict"])
        hook = getattr(model, "on_load_checkpoint", None)
        if callable(hook):
            hook(ckpt)
        return model

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad = True
        self.train()


class _UNetCore(*((_Base,) if _Base is not nn.Module else (_CheckpointCompat, nn.Module))):
    def _assemble(self, in_channels, out_channels, f_maps, order, num_groups, block, default_levels):
        if isinstance(f_maps, int):
            f_maps = create_feature_maps(f_maps, number_of_fmaps=default_levels)
        f_maps = list(f_maps)
        concat = block is DoubleConv
        self.encoders = nn.ModuleList(
            Encoder(in_channels if i == 0 else f_maps[i - 1], f, apply_pooling=(i != 0), basic_module=block,
                    conv_layer_order=order, num_groups=num_groups) for i, f in enumerate(f_maps))
        rev = f_maps[::-1]
        self.decoders = nn.ModuleList(
            Decoder(rev[i] + rev[i + 1] if concat else rev[i], rev[i + 1], basic_module=block, conv_layer_order=order,
                    num_groups=num_groups) for i in range(len(rev) - 1))
        self.final_conv = hnn.Conv3d(f_maps[0], out_channels, 1, planar_output=True)

    def forward(self, x):
        # ReLU / LeakyReLU anywhere in the order string: the whole network asks for exact fp32 products in the fp32 storage
        # mode (config.exact_products); the default 'cge' runs the split-bf16 contraction
        kinked = any(getattr(m, "_kinked", False) for m in self.modules())
        with config.exact_products(kinked):
            return self._forward(x)

    def forward_features(self, x):
        """Everything in front of `final_conv` (model.py:189-206): the last decoder block's output, channels-last.  The fused
        head + Dice step (train.SegmentationStep, ops.head_dice) continues from here."""
        kinked = any(getattr(m, "_kinked", False) for m in self.modules())
        with config.exact_products(kinked):
            return self._features(x)

    def _forward(self, x):
        x = self.final_conv(self._features(x))
        if self.testing and self.final_activation is not None:
            x = self.final_activation(x)
        return x

    def _features(self, x):
        skips = []
        for i, enc in enumerate(self.encoders):
            # the next level pools this level's output (model.py:194-199): tell the block, it writes the pooled tensor as well
            nxt = self.encoders[i + 1] if i + 1 < len(self.encoders) else None
            pool_next = nxt.pooling.mode if (nxt is not None and isinstance(nxt.pooling, hnn._Pool2)) else None
            if i == 0:
                x = enc(x, pool_next=pool_next)
            else:  # x is both the previous level's skip tensor and this level's pooling input (model.py:194-199)
                # (x is re-bound to the level's output and skips[0] to the returned skip tensor: the previous output has no
                #  other consumer, which lets the pooling backward leave its gradient unwritten, ops.SkipPool2Fn)
                # ... provided nobody else SAW that output: a forward hook on the level that produced it (or a global module
                # hook) may have handed it to an auxiliary / deep-supervision loss, and then its gradient must be materialised
                skips[0], x = enc(x, with_skip=True, pool_next=pool_next, sole_consumer=not _has_forward_hooks(self.encoders[i - 1]))
            skips.insert(0, x)
        for dec, skip in zip(self.decoders, skips[1:]):
            x = dec(skip, x)
        return x


def _has_forward_hooks(module) -> bool:
    """Could anything but this package's own forward have received `module`'s output (or that of one of its children)?"""
    import torch.nn.modules.module as M
    if getattr(M, "_global_forward_hooks", None) or getattr(M, "_global_forward_hooks_always_called", None):
        return True
    return any(m._forward_hooks for m in module.modules())


class UNet3D(_UNetCore):
    def __init__(self, in_channels, out_channels, final_sigmoid, f_maps=64, layer_order="gcr", num_groups=8, **kwargs):
        super().__init__()
        self.testing = kwargs.get("testing", False)
        self._assemble(in_channels, out_channels, f_maps, layer_order, num_groups, DoubleConv, default_levels=4)
        self.final_activation = nn.Sigmoid() if final_sigmoid else _Softmax1()


class ResidualUNet3D(_UNetCore):
    def __init__(self, in_channels, out_channels, final_sigmoid, f_maps=32, conv_layer_order="cge", num_groups=8,
                 skip_final_activation=False, **kwargs):
        super().__init__()
        self.testing = kwargs.get("testing", False)
        self._assemble(in_channels, out_channels, f_maps, conv_layer_order, num_groups, ExtResNetBlock,
                       default_levels=5)
        if skip_final_activation:
            self.final_activation = None
        else:
            self.final_activation = nn.Sigmoid() if final_sigmoid else _Softmax1()
