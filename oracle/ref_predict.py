"""CPU oracle of the inference path -- TEST INFRASTRUCTURE ONLY, never imported by the product.

numpy restatement of what the reference does around the network at prediction time:
  * grid_patch_generator        midasmednet/dataset.py:349-390   pad + overlapping grid of patches
  * GridPatchSampler.add_processed_batch   dataset.py:439-474     crop the overlap, clip at the volume edge, stitch
    (including its asymmetric first-axis crop `patch_overlap[0]:-patch_overlap[1]`, dataset.py:453)
  * the post-processing of examples/predict.py:87-95              argmax(softmax(class logits)), clip heat maps to uint8
Pinned by tools/make_golden.py (predict cases): the reference's own functions are run on the same inputs and the
results are asserted identical before tests/golden/predict.npz is written.
"""
from __future__ import annotations

import numpy as np

from . import ref_cpu as _O

# deterministic cases shared by tools/make_golden.py (which runs the reference on them) and the tests
PREDICT_CASES = [
    # tag, image shape (C, D, H, W), patch size, overlap, pad mode, heat maps, classes, batch
    ("sym", (1, 20, 17, 23), [16, 16, 16], [2, 3, 4], "symmetric", 3, 2, 3),
    ("const", (1, 33, 40, 29), [16, 24, 16], [4, 4, 4], "constant", 1, 4, 4),
    ("divisible", (2, 16, 16, 16), [8, 8, 8], [2, 2, 2], "constant", 0, 3, 5),
    ("deep_pad", (1, 5, 9, 6), [12, 12, 12], [4, 3, 2], "symmetric", 2, 2, 2),  # padding wider than the volume
]


def predict_inputs(tag, shape, patch, nh, ncls):
    """Deterministic volume (float16 like the reference's readers, dataset.py:431) and a per-patch logits maker."""
    img = _O._rng("predict:img:" + tag).standard_normal(shape).astype(np.float16)

    def logits_for(count):
        return (_O._rng(f"predict:logits:{tag}:{count}").standard_normal((nh + ncls,) + tuple(patch)) * 90.0).astype(np.float32)

    return img, logits_for


def grid_patch_generator(img, patch_size, patch_overlap, **pad_kwargs):
    """dataset.py:349-390.  img: C x D x H x W.  Yields (patch C x pD x pH x pW, idx[3], count)."""
    patch_size = np.array(patch_size)
    img_size = np.array(img.shape[1:])
    patch_overlap = np.array(patch_overlap)
    cropped = patch_size - 2 * patch_overlap
    n_patches = np.ceil(img_size / cropped).astype(int)
    overhead = cropped - img_size % cropped
    padded = np.pad(img, [[0, 0]] + [[patch_overlap[k], patch_overlap[k] + overhead[k]] for k in range(3)], **pad_kwargs)
    count = -1
    for p0 in np.arange(0, n_patches[0]) * cropped[0]:
        for p1 in np.arange(0, n_patches[1]) * cropped[1]:
            for p2 in np.arange(0, n_patches[2]) * cropped[2]:
                idx = np.array([p0, p1, p2])
                end = idx + patch_size
                count += 1
                yield padded[:, idx[0]:end[0], idx[1]:end[1], idx[2]:end[2]], idx, count


def crop_window(patch_size, patch_overlap):
    """(start[3], shape[3]) of `data[:, o0:-o1, o1:-o1, o2:-o2]` (dataset.py:452-455) for a patch of `patch_size`."""
    o = [int(v) for v in patch_overlap]
    probe = [np.arange(int(p)) for p in patch_size]
    sl = [probe[0][o[0]:-o[1]], probe[1][o[1]:-o[1]], probe[2][o[2]:-o[2]]]
    start = [int(s[0]) if len(s) else 0 for s in sl]
    return start, [len(s) for s in sl]


def add_processed_batch(result, data, pos, patch_overlap):
    """dataset.py:446-474 for one batch.  result: C_out x D x H x W (modified in place), data: B x C_out x pD x pH x pW,
    pos: B x 3 grid positions as yielded by grid_patch_generator."""
    o = patch_overlap
    img_size = np.array(result.shape[1:])
    for i in range(data.shape[0]):
        cropped = np.array(data[i, :, o[0]:-o[1], o[1]:-o[1], o[2]:-o[2]])
        p = np.array(pos[i])
        p_end = p + np.array(cropped.shape[1:])
        over = np.maximum(p_end - np.minimum(p_end, img_size), [0, 0, 0])
        new = np.array(cropped.shape[1:]) - over
        result[:, p[0]:p_end[0], p[1]:p_end[1], p[2]:p_end[2]] = cropped[:, :new[0], :new[1], :new[2]].astype(result.dtype)


def postprocess(logits, num_heatmaps):
    """examples/predict.py:88-95.  logits: B x (H + classes) x ... float32 -> B x (H + 1) x ... uint8."""
    cls = logits[:, num_heatmaps:]
    m = cls.max(axis=1, keepdims=True)
    e = np.exp((cls - m).astype(np.float32))
    sm = e / e.sum(axis=1, keepdims=True)                       # F.softmax(dim=1)
    lab = np.argmax(sm, axis=1)[:, None]                        # torch.argmax(dim=1, keepdim=True): first maximum
    hm = np.clip(logits[:, :num_heatmaps], 0.0, 255.0)
    return np.concatenate([hm.astype(np.uint8), lab.astype(np.uint8)], axis=1)


def predict_volume(forward, img, patch_size, patch_overlap, num_heatmaps, batch_size=2, pad_kwargs=None):
    """The loop of predict.py:83-97 on one volume: img C x D x H x W (float16/32) -> uint8 (H + 1) x D x H x W.
    `forward`: float32 array B x C x pD x pH x pW -> logits B x (H + classes) x pD x pH x pW."""
    pad_kwargs = {"mode": "symmetric"} if pad_kwargs is None else pad_kwargs
    result = np.zeros((num_heatmaps + 1,) + tuple(img.shape[1:]), dtype=np.uint8)
    patches, poss = [], []

    def flush():
        if patches:
            out = postprocess(forward(np.stack(patches).astype(np.float32)), num_heatmaps)
            add_processed_batch(result, out, np.stack(poss), patch_overlap)
            patches.clear()
            poss.clear()

    for patch, idx, _ in grid_patch_generator(img, patch_size, patch_overlap, **pad_kwargs):
        patches.append(np.array(patch))
        poss.append(idx)
        if len(patches) == batch_size:
            flush()
    flush()
    return result
