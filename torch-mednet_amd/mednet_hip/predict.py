"""Patch-based inference on the device: what examples/predict.py:83-97 does around `model(inputs)`
with GridPatchSampler (midasmednet/dataset.py:349-474), without leaving HBM.

    predictor = GridPredictor(model, patch_size, patch_overlap, num_heatmaps, pad_mode="constant", batch_size=4)
    result = predictor(volume)        # volume: C x D x H x W (numpy / torch, float16 or float32) -> uint8 (H + 1) x D x H x W

The volume is uploaded once; the overlapping grid patches are gathered on the device (np.pad semantics: constant or
symmetric), run through the network in batches (forward kernels only, `torch.no_grad`), and the logits are turned into
the uint8 result -- arg-max class, heat maps clipped to 0..255 -- and stitched into the result volume by ONE kernel that
applies the reference's crop rule (including its asymmetric first-axis window, dataset.py:453).  Bit-identical to the
reference's procedure for identical logits (tests/test_gpu_predict.py)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L

_PAD = {"constant": L.PAD_CONSTANT, "symmetric": L.PAD_SYMMETRIC}


def grid_positions(img_size, patch_size, patch_overlap):
    """Grid of dataset.py:372-389: positions (in the padded volume) of every patch, in the reference's order."""
    patch_size, img_size, ov = np.array(patch_size), np.array(img_size), np.array(patch_overlap)
    cropped = patch_size - 2 * ov
    if np.any(cropped <= 0):
        raise ValueError(f"patch_size {patch_size.tolist()} leaves nothing after removing the overlap {ov.tolist()} twice")
    n_patches = np.ceil(img_size / cropped).astype(int)
    axes = [np.arange(0, n_patches[k]) * cropped[k] for k in range(3)]
    return np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.int32)


def crop_window(patch_size, patch_overlap):
    """(start[3], shape[3]) of the reference's `data[:, o0:-o1, o1:-o1, o2:-o2]` (dataset.py:452-455)."""
    o = [int(v) for v in patch_overlap]
    sl = [slice(o[0], -o[1]), slice(o[1], -o[1]), slice(o[2], -o[2])]
    idx = [range(int(p))[s] for p, s in zip(patch_size, sl)]
    return [r.start if len(r) else 0 for r in idx], [len(r) for r in idx]


def gather_patches(volume: torch.Tensor, pos: torch.Tensor, patch_size, patch_overlap, pad_mode="symmetric"):
    """volume C x D x H x W fp32 (device), pos B x 3 int32 (device) -> B x C x pD x pH x pW fp32."""
    L.require_gpu(volume, "grid_gather")
    if pad_mode not in _PAD:
        raise NotImplementedError(f"pad mode {pad_mode!r} (supported: constant, symmetric)")
    c, d, h, w = volume.shape
    pd, ph, pw = (int(v) for v in patch_size)
    out = torch.empty((pos.shape[0], c, pd, ph, pw), dtype=torch.float32, device=volume.device)
    o = [int(v) for v in patch_overlap]
    L.check(L.lib().mednet_grid_gather(volume.data_ptr(), pos.data_ptr(), out.data_ptr(), pos.shape[0], c, d, h, w, pd, ph, pw,
                                       o[0], o[1], o[2], _PAD[pad_mode], L.stream()), "grid_gather")
    return out


def assemble(logits: torch.Tensor, pos: torch.Tensor, result: torch.Tensor, num_heatmaps, patch_overlap):
    """logits B x (H + classes) x pD x pH x pW fp32 planar -> writes uint8 result (H + 1) x D x H x W in place."""
    L.require_gpu(logits, "predict_assemble")
    lg = logits.float().contiguous()
    b, ch, pd, ph, pw = lg.shape
    _, d, h, w = result.shape
    start, shape = crop_window((pd, ph, pw), patch_overlap)
    L.check(L.lib().mednet_predict_assemble(lg.data_ptr(), pos.data_ptr(), result.data_ptr(), b, num_heatmaps, ch - num_heatmaps,
                                            d, h, w, pd, ph, pw, start[0], start[1], start[2], shape[0], shape[1], shape[2],
                                            L.stream()), "predict_assemble")
    return result


class GridPredictor:
    def __init__(self, model, patch_size, patch_overlap, num_heatmaps=0, pad_mode="constant", batch_size=4,
                 channel_selection=None):
        self.model, self.patch_size, self.patch_overlap = model, list(patch_size), list(patch_overlap)
        self.num_heatmaps, self.pad_mode, self.batch_size, self.channel_selection = num_heatmaps, pad_mode, batch_size, channel_selection

    @torch.no_grad()
    def __call__(self, volume):
        dev = next(self.model.parameters()).device
        vol = torch.as_tensor(np.asarray(volume) if not torch.is_tensor(volume) else volume)
        if self.channel_selection is not None:
            vol = vol[self.channel_selection]
        vol = vol.to(dev, non_blocking=True).float().contiguous()  # predict.py:85 `.float()`
        pos_all = torch.from_numpy(grid_positions(vol.shape[1:], self.patch_size, self.patch_overlap)).to(dev)
        result = torch.zeros((self.num_heatmaps + 1,) + tuple(vol.shape[1:]), dtype=torch.uint8, device=dev)
        was_training = self.model.training
        self.model.eval()
        try:
            for b0 in range(0, pos_all.shape[0], self.batch_size):
                pos = pos_all[b0:b0 + self.batch_size].contiguous()
                inputs = gather_patches(vol, pos, self.patch_size, self.patch_overlap, self.pad_mode)
                logits = self.model(inputs)
                assemble(logits, pos, result, self.num_heatmaps, self.patch_overlap)
        finally:
            self.model.train(was_training)
        return result
