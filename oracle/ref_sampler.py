"""CPU oracle of the training-patch sampler -- TEST INFRASTRUCTURE ONLY, never imported by the product.

numpy restatement of midasmednet/dataset.py: get_labeled_position (:18-52), get_random_patch_indices (:55-88) and
MedDataset.__getitem__ (:285-346, without the optional `transform`).  All randomness comes from numpy's GLOBAL generator in
exactly the reference's call order (np.random.choice / randint / choice / randint), so seeding np.random reproduces the
reference's sequence of patches.  Pinned by tools/make_golden.py (sampler cases): the reference's functions are run on the
same volumes with the same seed and must agree exactly before tests/golden/sampler.npz is written.
"""
from __future__ import annotations

import numpy as np

from . import ref_cpu as _O


def get_labeled_position(label, class_value, label_any=None):
    """dataset.py:18-52.  NOTE the reference's second draw: `np.argwhere(column == class_value)[0]` keeps only the FIRST
    matching index along axis 2, so np.random.choice picks from a one-element array (it still advances the generator)."""
    if label_any is None:
        label_any = np.any(label == class_value, axis=2)
    valid_idx = np.argwhere(label_any == True)  # noqa: E712
    if valid_idx.size:
        rnd = np.random.randint(0, valid_idx.shape[0])
        idx = valid_idx[rnd]
        column = label[idx[0], idx[1], :]
        first = np.argwhere(column == class_value)[0]
        rnd = np.random.choice(first)
        return [idx[0], idx[1], rnd]
    return None


def get_random_patch_indices(patch_size, img_shape, pos=None):
    """dataset.py:55-88 (`np.int` of the reference is the builtin int)."""
    if pos:
        pos = np.array(pos, dtype=int)
        min_index = np.maximum(pos - patch_size + 1, 0)
        max_index = np.minimum(img_shape - patch_size + 1, pos + 1)
    else:
        min_index = np.array([0, 0, 0])
        max_index = img_shape - patch_size + 1
    index_ini = np.random.randint(low=min_index, high=max_index)
    return index_ini, index_ini + patch_size


class PatchSampler:
    """MedDataset (dataset.py:208-346) on in-memory volumes: images[i] C x D x H x W float16, labels[i] L x D x H x W uint8
    (class map = last channel), heatmaps[i] Hm x D x H x W uint8 or None."""

    def __init__(self, images, labels, patch_size, samples_per_subject=1, heatmaps=None, class_probabilities=None,
                 subject_keys=None):
        self.images, self.labels, self.heatmaps = images, labels, heatmaps
        self.patch_size = np.array(patch_size, dtype=int)
        self.samples_per_subject = samples_per_subject
        self.subject_keys = subject_keys if subject_keys is not None else [str(i) for i in range(len(images))]
        self.class_probabilities = class_probabilities
        self._label_ax2_any = []
        if class_probabilities:
            self.class_probabilities = class_probabilities / np.sum(class_probabilities)   # :246-248
            for idx in range(len(labels)):                                                   # :267-272
                self._label_ax2_any.append([np.any(labels[idx][-1, ...] == c, axis=2) for c in range(len(class_probabilities))])

    def __len__(self):
        return len(self.images) * self.samples_per_subject

    def __getitem__(self, idx):
        idx = idx % len(self.images)
        imgs, lbls = self.images[idx], self.labels[idx]
        pos, selected_class = None, 0
        if self.class_probabilities is not None:
            selected_class = np.random.choice(range(len(self.class_probabilities)), p=self.class_probabilities)
            if selected_class > 0:
                pos = get_labeled_position(lbls[-1], selected_class, label_any=self._label_ax2_any[idx][selected_class])
        ini, fin = get_random_patch_indices(self.patch_size, np.array(imgs.shape[1:]), pos=pos)
        sl = (slice(None), slice(ini[0], fin[0]), slice(ini[1], fin[1]), slice(ini[2], fin[2]))
        data = imgs[sl].astype(np.float32)
        label = lbls[sl].astype(np.uint8)
        if self.heatmaps is not None:
            label = np.concatenate([self.heatmaps[idx][sl].astype(np.uint8), label], axis=0)
        return {"subject_key": self.subject_keys[idx], "patch_position": ini, "selected_class": selected_class,
                "data": data, "label": label}


# deterministic volumes shared by tools/make_golden.py (which runs the reference on them) and the tests
SAMPLER_CASES = [
    # tag, volume shapes (D, H, W) per subject, image channels, heat maps, patch, class probabilities, draws, seed
    ("seg", [(24, 30, 20), (32, 18, 26)], 1, 0, [16, 16, 16], [0.2, 0.4, 0.4], 24, 11),
    ("ldmk", [(20, 20, 24)], 2, 3, [12, 16, 8], [0.1, 0.9], 16, 5),
    ("uniform", [(18, 17, 19), (16, 16, 16), (21, 16, 30)], 1, 0, [16, 16, 16], None, 18, 3),
]


def sampler_volumes(tag, shapes, c_img, n_hm, n_classes):
    rng = _O._rng("sampler:" + tag)
    images, labels, heatmaps = [], [], []
    for s in shapes:
        images.append(rng.standard_normal((c_img,) + tuple(s)).astype(np.float16))
        lab = np.zeros((1,) + tuple(s), dtype=np.uint8)
        for c in range(1, n_classes):  # a few small blobs per class, so label sampling has something to find
            for _ in range(2):
                z, y, x = (int(rng.integers(0, d - 3)) for d in s)
                lab[0, z:z + 3, y:y + 3, x:x + 3] = c
        labels.append(lab)
        heatmaps.append(rng.integers(0, 256, size=(n_hm,) + tuple(s), dtype=np.uint8))
    return images, labels, (heatmaps if n_hm else None)
