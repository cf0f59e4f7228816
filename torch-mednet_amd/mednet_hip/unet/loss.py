"""Losses of midasmednet/unet/loss.py.  DiceLoss / dice_metric (the training path), CrossEntropyLoss, HeatmapRegressionLoss and
LandmarkLoss run the fused HIP kernels and are defined HERE; the classes of that file no caller of the reference uses live in
loss_compat.py (torch-op restatements of the reference's semantics, kept only so the module's public names resolve) and are
re-exported below -- product code and restated reference code are separate files."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


from .loss_compat import (flatten, expand_as_one_hot, compute_per_channel_dice, CELoss, WeightedCrossEntropyLoss,  # noqa: F401,E402
                          BCELossWrapper, PixelWiseCrossEntropyLoss)  # reference-semantics restatements, off the hot path: loss_compat.py


def dice_metric(logits, labels):
    """Per-channel soft Dice of softmax(logits) vs labels; one fused pass on the GPU (loss.py:51-55)."""
    return ops.per_channel_dice(logits, labels)


class DiceLoss(nn.Module):
    def __init__(self, epsilon=1e-5, weight=None, ignore_index=None, sigmoid_normalization=False,
                 skip_last_target=False):
        super().__init__()
        self.epsilon = epsilon
        self.register_buffer("weight", weight)
        self.ignore_index = ignore_index
        self.sigmoid_normalization = sigmoid_normalization
        self.normalization = nn.Sigmoid() if sigmoid_normalization else nn.Softmax(dim=1)
        self.skip_last_target = skip_last_target

    def forward(self, input, target):
        if self.skip_last_target:
            # loss.py:124-128: the one-hot target loses its last channel, the probabilities do not -> the reference's
            # shape assert fires for every input; keep that behaviour.
            raise AssertionError("'input' and 'target' must have the same shape")
        return ops.dice_loss(input, target, self.weight, self.epsilon, self.sigmoid_normalization, self.ignore_index)


class LandmarkLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.ce_loss = WeightedCrossEntropyLoss(target_one_hot_encoded=False)

    def forward(self, logits, heatmaps):
        return ops.heatmap_loss(logits, heatmaps, None, "L2")  # (loss.py:251 F.mse_loss; raises on CPU tensors like every op)


class CrossEntropyLoss(nn.Module):
    """Drop-in for the torch.nn.CrossEntropyLoss(weight) the callers build (segmentation.py:49, landmarks.py:49)."""

    def __init__(self, weight=None, ignore_index=-100):
        super().__init__()
        self.register_buffer("weight", weight)
        self.ignore_index = ignore_index

    def forward(self, input, target):
        return ops.cross_entropy(input, target, self.weight, self.ignore_index)


class HeatmapRegressionLoss(nn.Module):
    """Fused form of the per-channel loop of LandmarkNet.loss (landmarks.py:129-132)."""

    def __init__(self, channel_weights, kind="L2"):
        super().__init__()
        self.register_buffer("channel_weights", torch.as_tensor(channel_weights, dtype=torch.float32))
        self.kind = kind

    def forward(self, output_heatmaps, heatmaps):
        return ops.heatmap_loss(output_heatmaps, heatmaps, self.channel_weights, self.kind)
