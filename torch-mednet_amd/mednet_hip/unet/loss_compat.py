"""API-parity restatements of the part of midasmednet/unet/loss.py that NO caller of the reference uses (the "loss zoo",
/root/reference/midasmednet/unet/loss.py:58-88 one-hot helper, :24-48 generic per-channel Dice, :144-241 CELoss /
WeightedCrossEntropyLoss / BCELossWrapper / PixelWiseCrossEntropyLoss).

Kept apart from the product module on purpose: these are plain torch-op formulations with the reference's semantics, written so
that `from midasmednet.unet.loss import <any public name>` keeps resolving after the switch; they are OFF the HIP hot path, counted
as no component, and nothing in mednet_hip's training / inference path imports them (DiceLoss, dice_metric, CrossEntropyLoss,
HeatmapRegressionLoss and LandmarkLoss in loss.py are the fused HIP paths).  `WeightedCrossEntropyLoss` forms its class weights
with torch ops and then calls the HIP cross-entropy kernel (ops.cross_entropy)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def flatten(tensor):
    """(N, C, D, H, W) -> (C, N*D*H*W)."""
    c = tensor.size(1)
    order = (1, 0) + tuple(range(2, tensor.dim()))
    return tensor.permute(order).contiguous().view(c, -1)


def expand_as_one_hot(input, C, ignore_index=None):
    """N x D x H x W labels -> N x C x D x H x W one-hot (float32); loss.py:58-88."""
    assert input.dim() == 4
    idx = input.unsqueeze(1)
    shape = list(idx.size())
    shape[1] = C
    if ignore_index is None:
        return torch.zeros(shape, device=idx.device).scatter_(1, idx, 1)
    keep_ignored = idx.expand(shape) == ignore_index
    idx = idx.clone()
    idx[idx == ignore_index] = 0
    result = torch.zeros(shape, device=idx.device).scatter_(1, idx, 1)
    result[keep_ignored] = ignore_index
    return result


def compute_per_channel_dice(input, target, epsilon=1e-5, ignore_index=None, weight=None):
    """Generic (probabilities, one-hot) form of loss.py:24-48, kept for API parity; the fused path is `DiceLoss`."""
    assert input.size() == target.size(), "'input' and 'target' must have the same shape"
    if ignore_index is not None:
        mask = target.clone().ne_(ignore_index)
        mask.requires_grad = False
        input = input * mask
        target = target * mask
    p = flatten(input)
    t = flatten(target).float()
    intersect = (p * t).sum(-1)
    if weight is not None:
        intersect = weight * intersect
    return 2.0 * intersect / (p + t).sum(-1).clamp(min=epsilon)


class CELoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.ce = nn.CrossEntropyLoss()

    def forward(self, inputs, targets):
        return self.ce(torch.softmax(inputs, dim=1), targets[:, 0, ...])


class WeightedCrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, ignore_index=-1, target_one_hot_encoded=True):
        super().__init__()
        self.register_buffer("weight", weight)
        self.ignore_index = ignore_index
        self.target_one_hot_encoded = target_one_hot_encoded

    @staticmethod
    def _class_weights(input):
        p = flatten(F.softmax(input, dim=1))
        return ((1.0 - p).sum(-1) / p.sum(-1)).detach()

    def forward(self, input, target):
        class_weights = self._class_weights(input)
        if self.weight is not None:
            class_weights = class_weights * self.weight
        if self.target_one_hot_encoded:
            target = torch.argmax(target, dim=1)
        return ops.cross_entropy(input, target, class_weights, self.ignore_index)


class BCELossWrapper:
    def __init__(self, loss_criterion, ignore_index=-1, skip_last_target=False):
        if hasattr(loss_criterion, "ignore_index"):
            raise RuntimeError(f"Cannot wrap {type(loss_criterion)}. Use 'ignore_index' attribute instead")
        self.loss_criterion = loss_criterion
        self.ignore_index = ignore_index
        self.skip_last_target = skip_last_target

    def __call__(self, input, target):
        if self.skip_last_target:
            target = target[:, :-1, ...]
        assert input.size() == target.size()
        if self.ignore_index is None:
            return self.loss_criterion(input, target)
        mask = target.clone().ne_(self.ignore_index)
        mask.requires_grad = False
        return self.loss_criterion(input * mask, target * mask)


class PixelWiseCrossEntropyLoss(nn.Module):
    def __init__(self, class_weights=None, ignore_index=None):
        super().__init__()
        self.register_buffer("class_weights", class_weights)
        self.ignore_index = ignore_index
        self.log_softmax = nn.LogSoftmax(dim=1)

    def forward(self, input, target, weights):
        assert target.size() == weights.size()
        logp = self.log_softmax(input)
        target = expand_as_one_hot(target, C=input.size(1), ignore_index=self.ignore_index)
        weights = weights.unsqueeze(0).expand_as(input)
        if self.ignore_index is not None:
            mask = target.detach().ne(self.ignore_index).float()
            logp = logp * mask
            target = target * mask
        if self.class_weights is None:
            self.register_buffer("class_weights", torch.ones(input.size(1), device=input.device))
        weights = self.class_weights.view(1, -1, 1, 1, 1) * weights
        return (-weights * target * logp).mean()
