"""TEST INFRASTRUCTURE (oracle): numpy restatement of the three batchgenerators transforms the reference composes for its
training patches (/root/reference/examples/train_seg.py:82-86, applied per sample at midasmednet/dataset.py:340-341 to
`patch['data']` of shape 1 x C x D x H x W):

    Compose([BrightnessTransform(mu=0.0, sigma=0.3), GammaTransform(gamma_range=(0.7, 1.3)),
             ContrastAugmentationTransform(contrast_range=(0.3, 1.7))])

PARITY UNPINNED.  The arithmetic lives in the third-party package `batchgenerators` (unpinned in requirements.txt:7, not
vendored under /root/reference, not installed in this image, no network), and the reference holds no fixture for it.  The
functions below restate the published algorithm of batchgenerators' `color_augmentations.py` (0.20-0.21 line:
`augment_brightness_additive`, `augment_gamma`, `augment_contrast`, and the `p_per_sample = 1` loops of the three Transform
classes) including the order in which they draw from numpy's global generator; they are pinned by property tests only
(tests/test_oracle_golden.py::test_augment_oracle_properties).  Only tests/ may import this module.
"""
import numpy as np

BRIGHTNESS = dict(mu=0.0, sigma=0.3)       # train_seg.py:84
GAMMA_RANGE = (0.7, 1.3)                   # train_seg.py:85
CONTRAST_RANGE = (0.3, 1.7)                # train_seg.py:86
EPSILON = 1e-7                             # augment_gamma's default


def _range_draw(lo_hi):
    """`if np.random.random() < 0.5 and r[0] < 1: uniform(r[0], 1) else uniform(max(r[0], 1), r[1])` (gamma and contrast)."""
    if np.random.random() < 0.5 and lo_hi[0] < 1:
        return np.random.uniform(lo_hi[0], 1)
    return np.random.uniform(max(lo_hi[0], 1), lo_hi[1])


def draw_parameters(batch, channels):
    """The random numbers of one Compose call per sample, in batchgenerators' order -> (B, C, 3) float32 array of
    {additive brightness, gamma (the same for every channel of a sample), contrast factor}."""
    out = np.zeros((batch, channels, 3), dtype=np.float32)
    for b in range(batch):
        # BrightnessTransform: per sample `uniform() < p_per_sample`, then per channel `uniform() <= p_per_channel`, normal
        np.random.uniform()
        for c in range(channels):
            np.random.uniform()
            out[b, c, 0] = np.random.normal(BRIGHTNESS["mu"], BRIGHTNESS["sigma"])
        # GammaTransform (per_channel=False, retain_stats=False)
        np.random.uniform()
        out[b, :, 1] = _range_draw(GAMMA_RANGE)
        # ContrastAugmentationTransform (per_channel=True, preserve_range=True)
        np.random.uniform()
        for c in range(channels):
            out[b, c, 2] = _range_draw(CONTRAST_RANGE)
    return out


def apply(data, params):
    """data: (B, C, D, H, W) float32 -> augmented copy, arithmetic in float32 like numpy on the reference's float32 patches."""
    out = np.array(data, dtype=np.float32, copy=True)
    for b in range(out.shape[0]):
        s = out[b]
        for c in range(s.shape[0]):                       # augment_brightness_additive
            s[c] += params[b, c, 0]
        gamma = np.float32(params[b, 0, 1])               # augment_gamma over the whole sample
        minm = s.min()
        rnge = s.max() - minm
        s[...] = np.power((s - minm) / np.float32(float(rnge) + EPSILON), gamma) * rnge + minm
        for c in range(s.shape[0]):                       # augment_contrast, per channel, range preserved
            mn = s[c].mean()
            lo, hi = s[c].min(), s[c].max()
            s[c] = (s[c] - mn) * np.float32(params[b, c, 2]) + mn
            s[c][s[c] < lo] = lo
            s[c][s[c] > hi] = hi
    return out
