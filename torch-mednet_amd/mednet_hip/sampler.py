"""Class-balanced random patch sampler on device-resident volumes: MedDataset of
midasmednet/dataset.py:208-346 without the per-patch numpy crop, pickling and host-to-device copy.

The volumes (images float16, labels / heat maps uint8 -- the dtypes the reference preloads, dataset.py:254-257) are
uploaded once.  WHERE to crop is decided on the host by the reference's own procedure -- get_labeled_position (:18-52) and
get_random_patch_indices (:55-88), drawing from numpy's global generator in the reference's order, so a seeded run visits
the same patches as the reference -- and the crop + cast (:313-331) of a whole batch is a few launches of
mednet_crop_patches writing straight into the batch tensors the training step consumes
({'data': fp32 B x C x pD x pH x pW, 'label': uint8 B x (heat maps + 1) x ...}, segmentation.py:58-61).
The optional `transform` of the reference -- the batchgenerators Compose of examples/train_seg.py:82-86 (additive brightness,
gamma, contrast on `data`), applied per sample at dataset.py:340-341 -- is `augment=True`: its random numbers are drawn on
the host right after the sample's position (the reference's order of calls into numpy's global generator), the arithmetic
runs on the device (mednet_augment_patches).  batchgenerators itself is not available here: see oracle/ref_augment.py
("parity unpinned")."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def get_labeled_position(label, class_value, label_any=None):
    """dataset.py:18-52 (host, numpy).  The reference keeps only the first matching index along axis 2."""
    if label_any is None:
        label_any = np.any(label == class_value, axis=2)
    valid_idx = np.argwhere(label_any == True)  # noqa: E712
    if valid_idx.size:
        rnd = np.random.randint(0, valid_idx.shape[0])
        idx = valid_idx[rnd]
        first = np.argwhere(label[idx[0], idx[1], :] == class_value)[0]
        return [idx[0], idx[1], np.random.choice(first)]
    return None


def get_random_patch_indices(patch_size, img_shape, pos=None):
    """dataset.py:55-88 (host, numpy)."""
    if pos:
        pos = np.array(pos, dtype=int)
        min_index = np.maximum(pos - patch_size + 1, 0)
        max_index = np.minimum(img_shape - patch_size + 1, pos + 1)
    else:
        min_index = np.array([0, 0, 0])
        max_index = img_shape - patch_size + 1
    index_ini = np.random.randint(low=min_index, high=max_index)
    return index_ini, index_ini + patch_size


# examples/train_seg.py:84-86
AUG_BRIGHTNESS = (0.0, 0.3)      # BrightnessTransform(mu, sigma)
AUG_GAMMA_RANGE = (0.7, 1.3)     # GammaTransform(gamma_range)
AUG_CONTRAST_RANGE = (0.3, 1.7)  # ContrastAugmentationTransform(contrast_range)


def _range_draw(lo_hi):
    if np.random.random() < 0.5 and lo_hi[0] < 1:
        return np.random.uniform(lo_hi[0], 1)
    return np.random.uniform(max(lo_hi[0], 1), lo_hi[1])


def draw_augmentation(channels):
    """The draws of one Compose call on ONE sample, in batchgenerators' order (per-sample and per-channel probability
    draws included, all probabilities 1) -> (C, 3) float32: additive brightness, gamma (one per sample), contrast factor."""
    out = np.zeros((channels, 3), dtype=np.float32)
    np.random.uniform()
    for c in range(channels):
        np.random.uniform()
        out[c, 0] = np.random.normal(*AUG_BRIGHTNESS)
    np.random.uniform()
    out[:, 1] = _range_draw(AUG_GAMMA_RANGE)
    np.random.uniform()
    for c in range(channels):
        out[c, 2] = _range_draw(AUG_CONTRAST_RANGE)
    return out


def augment_(data, params):
    """In place on a B x C x ... fp32 device tensor; params: (B, C, 3) array / tensor as from draw_augmentation."""
    L.require_gpu(data, "augment")
    assert data.dtype == torch.float32 and data.is_contiguous()
    b, c = data.shape[:2]
    spatial = data[0, 0].numel()
    prm = torch.as_tensor(np.asarray(params, dtype=np.float32)).reshape(b, c, 3).to(data.device).contiguous()
    lib = L.lib()
    ws = L.workspace(lib.mednet_augment_ws_bytes(b, c, spatial), data.device)
    L.check(lib.mednet_augment_patches(data.data_ptr(), prm.data_ptr(), b, c, spatial, ws.data_ptr(), ws.numel(), L.stream()),
            "augment_patches")
    return data


_DT = {torch.float16: L.F16, torch.float32: L.F32, torch.uint8: L.U8}


class DevicePatchSampler:
    """images[i]: C x D x H x W float16/float32, labels[i]: L x D x H x W uint8 (class map last), heatmaps[i] optional uint8
    (numpy arrays or tensors).  `batch(indices)` -> the collated batch dict on the device."""

    def __init__(self, images, labels, patch_size, samples_per_subject=1, heatmaps=None, class_probabilities=None,
                 subject_keys=None, device="cuda:0", augment=False):
        self.device = torch.device(device)
        self.augment = augment  # the reference's `transform` (train_seg.py:82-86) on 'data'
        self.last_augmentation = None
        self.patch_size = np.array(patch_size, dtype=int)
        self.samples_per_subject = samples_per_subject
        self.subject_keys = subject_keys if subject_keys is not None else [str(i) for i in range(len(images))]
        assert len(images) == len(labels)
        self._labels_host = [np.asarray(l) for l in labels]        # the position sampling reads label columns on the host
        self.images = [torch.as_tensor(np.asarray(v)).to(self.device).contiguous() for v in images]
        self.labels = [torch.as_tensor(np.asarray(v)).to(self.device).contiguous() for v in labels]
        self.heatmaps = None if heatmaps is None else [torch.as_tensor(np.asarray(v)).to(self.device).contiguous() for v in heatmaps]
        self.class_probabilities = class_probabilities
        self._label_ax2_any = []
        if class_probabilities:
            self.class_probabilities = class_probabilities / np.sum(class_probabilities)
            for lab in self._labels_host:
                self._label_ax2_any.append([np.any(lab[-1, ...] == c, axis=2) for c in range(len(class_probabilities))])

    def __len__(self):
        return len(self.images) * self.samples_per_subject

    def position(self, idx):
        """(subject, index_ini, selected_class) of dataset.py:286-310 for dataset index idx."""
        idx = idx % len(self.images)
        pos, selected_class = None, 0
        if self.class_probabilities is not None:
            selected_class = np.random.choice(range(len(self.class_probabilities)), p=self.class_probabilities)
            if selected_class > 0:
                pos = get_labeled_position(self._labels_host[idx][-1], selected_class,
                                           label_any=self._label_ax2_any[idx][selected_class])
        ini, _ = get_random_patch_indices(self.patch_size, np.array(self.images[idx].shape[1:]), pos=pos)
        return idx, ini, selected_class

    def batch(self, indices):
        L.require_gpu(self.images[0], "patch sampler")
        plan, aug = [], []
        for i in indices:  # per sample: position draws, then (optionally) the transform's draws -- the reference's order
            plan.append(self.position(i))
            if self.augment:
                aug.append(draw_augmentation(self.images[0].shape[0]))
        b = len(plan)
        pd, ph, pw = (int(v) for v in self.patch_size)
        c_img = self.images[0].shape[0]
        n_hm = 0 if self.heatmaps is None else self.heatmaps[0].shape[0]
        n_lab = self.labels[0].shape[0]
        data = torch.empty((b, c_img, pd, ph, pw), dtype=torch.float32, device=self.device)
        label = torch.empty((b, n_hm + n_lab, pd, ph, pw), dtype=torch.uint8, device=self.device)
        lib = L.lib()
        for subj in sorted({p[0] for p in plan}):
            slots = [k for k, p in enumerate(plan) if p[0] == subj]
            pos = torch.tensor(np.stack([plan[k][1] for k in slots]).astype(np.int32), device=self.device)
            slot = torch.tensor(slots, dtype=torch.int32, device=self.device)
            jobs = [(self.images[subj], data, L.F32, c_img, 0)]
            if n_hm:
                jobs.append((self.heatmaps[subj], label, L.U8, n_hm + n_lab, 0))
            jobs.append((self.labels[subj], label, L.U8, n_hm + n_lab, n_hm))
            for vol, out, dst, c_total, c_off in jobs:
                c, d, h, w = vol.shape
                L.check(lib.mednet_crop_patches(vol.data_ptr(), _DT[vol.dtype], pos.data_ptr(), slot.data_ptr(), len(slots),
                                                out.data_ptr(), dst, c, d, h, w, c_total, c_off, pd, ph, pw, L.stream()),
                        "crop_patches")
        if self.augment:
            self.last_augmentation = np.stack(aug)
            augment_(data, self.last_augmentation)
        return {"subject_key": [self.subject_keys[p[0]] for p in plan],
                "patch_position": np.stack([p[1] for p in plan]), "selected_class": np.array([int(p[2]) for p in plan]),
                "data": data, "label": label}
