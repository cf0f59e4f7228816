"""Row N2 (inference path) on the GPU: the device-side patch gather and the fused post-processing + stitching kernel
against the CPU oracle (oracle/ref_predict.py) and the reference's golden result -- bit-exact (byte / index work)."""
import os

import numpy as np
import pytest
import torch

import mednet_hip
from mednet_hip import predict as HP
from mednet_hip.unet import model as HM
from oracle import ref_cpu as O
from oracle import ref_predict as P

from gpu_util import DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", P.PREDICT_CASES, ids=[c[0] for c in P.PREDICT_CASES])
def test_grid_gather_matches_reference_patches(case, golden_dir):
    tag, shape, patch, ov, mode, nh, ncls, bs = case
    rec = np.load(os.path.join(golden_dir, "predict.npz"))
    img, _ = P.predict_inputs(tag, shape, patch, nh, ncls)
    ora = list(P.grid_patch_generator(img, patch, ov, mode=mode))
    pos = torch.from_numpy(HP.grid_positions(shape[1:], patch, ov)).to(DEV)
    assert np.array_equal(pos.cpu().numpy(), rec[f"{tag}.pos"])
    got = HP.gather_patches(torch.from_numpy(img).to(DEV).float(), pos, patch, ov, mode).cpu().numpy()
    assert got.shape[0] == int(rec[f"{tag}.npatches"])
    for i, (p, _, _) in enumerate(ora):
        assert np.array_equal(got[i], np.asarray(p, dtype=np.float32)), (tag, i)
    assert np.array_equal(got[0], rec[f"{tag}.first_patch"].astype(np.float32))
    assert np.array_equal(got[-1], rec[f"{tag}.last_patch"].astype(np.float32))


@pytest.mark.parametrize("case", P.PREDICT_CASES, ids=[c[0] for c in P.PREDICT_CASES])
def test_postprocess_and_stitch_match_reference_result(case, golden_dir):
    tag, shape, patch, ov, mode, nh, ncls, bs = case
    rec = np.load(os.path.join(golden_dir, "predict.npz"))
    _, logits_for = P.predict_inputs(tag, shape, patch, nh, ncls)
    pos_all = HP.grid_positions(shape[1:], patch, ov)
    result = torch.zeros((nh + 1,) + tuple(shape[1:]), dtype=torch.uint8, device=DEV)
    for b0 in range(0, len(pos_all), bs):
        logits = torch.from_numpy(np.stack([logits_for(c) for c in range(b0, min(b0 + bs, len(pos_all)))])).to(DEV)
        HP.assemble(logits, torch.from_numpy(pos_all[b0:b0 + bs]).to(DEV), result, nh, ov)
    assert np.array_equal(result.cpu().numpy(), rec[f"{tag}.result"]), tag


def test_ties_and_clip_edges():
    """first maximum on ties; clip at 0 and 255 with truncation (np.clip(..).astype(uint8))."""
    lg = torch.zeros(1, 3, 8, 8, 8)
    lg[0, 0] = torch.linspace(-3.0, 300.0, 512).reshape(8, 8, 8)   # heat map
    lg[0, 1] = 1.5
    lg[0, 2] = 1.5                                                # tie -> class 0
    lg[0, 2, 4:] = 1.5000001
    out = P.postprocess(lg.numpy(), 1)
    res = np.zeros((2, 8, 8, 8), dtype=np.uint8)
    P.add_processed_batch(res, out, np.zeros((1, 3), dtype=int), [1, 1, 1])
    got = torch.zeros((2, 8, 8, 8), dtype=torch.uint8, device=DEV)
    HP.assemble(lg.to(DEV), torch.zeros((1, 3), dtype=torch.int32, device=DEV), got, 1, [1, 1, 1])
    assert np.array_equal(got.cpu().numpy(), res)


def test_grid_predictor_end_to_end_matches_oracle_loop():
    """GridPredictor (device gather -> HIP forward -> fused arg-max / clip / stitch) against the oracle's restatement of
    predict.py's loop around the CPU oracle network, fp32 mode: the uint8 volumes agree except where a logit sits within
    float noise of a decision boundary."""
    ctor = dict(in_channels=1, out_channels=5, final_sigmoid=False, f_maps=[8, 16])
    nh, patch, ov = 2, [16, 16, 16], [2, 3, 4]
    img = (O._rng("predict:e2e").standard_normal((1, 21, 30, 19)) * 2).astype(np.float16)
    ora = O.keyed_init_(O.ResidualUNet3D(**ctor)).eval()

    def fwd(x):
        with torch.no_grad():
            return (ora(torch.from_numpy(x)) * 40.0).numpy()

    want = P.predict_volume(fwd, img, patch, ov, nh, batch_size=3, pad_kwargs={"mode": "symmetric"})
    with mednet_hip.precision("fp32"):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)

        class Scaled(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, x):
                return self.m(x) * 40.0

        got = HP.GridPredictor(Scaled(net), patch, ov, num_heatmaps=nh, pad_mode="symmetric", batch_size=3)(img).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.uint8
    assert (got[nh] != want[nh]).mean() < 2e-3                      # labels
    d = np.abs(got[:nh].astype(int) - want[:nh].astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3                 # heat maps: truncation flips only


# ---------------------------------------------------------------------------------------------- row N1: patch sampler
@pytest.mark.parametrize("case", __import__("oracle.ref_sampler", fromlist=["x"]).SAMPLER_CASES,
                         ids=[c[0] for c in __import__("oracle.ref_sampler", fromlist=["x"]).SAMPLER_CASES])
def test_device_patch_sampler_reproduces_reference_batches(case, golden_dir):
    """DevicePatchSampler (host position logic + mednet_crop_patches) seeded like the reference: same subjects, positions,
    classes, and bit-identical 'data' (fp32) / 'label' (uint8) tensors as the oracle, whose run equals the reference's."""
    from mednet_hip.sampler import DevicePatchSampler
    from oracle import ref_sampler as S
    tag, shapes, c_img, n_hm, patch, probs, draws, seed = case
    rec = np.load(os.path.join(golden_dir, "sampler.npz"))
    images, labels, heatmaps = S.sampler_volumes(tag, shapes, c_img, n_hm, len(probs) if probs else 3)
    ora = S.PatchSampler(images, labels, patch, samples_per_subject=4, heatmaps=heatmaps, class_probabilities=probs)
    dev = DevicePatchSampler(images, labels, patch, samples_per_subject=4, heatmaps=heatmaps, class_probabilities=probs, device=DEV)
    np.random.seed(seed)
    items = [ora[i] for i in range(draws)]
    np.random.seed(seed)
    got = []
    for b0 in range(0, draws, 5):  # batches of 5 (mixing subjects inside a batch)
        got.append(dev.batch(list(range(b0, min(b0 + 5, draws)))))
    pos = np.concatenate([g["patch_position"] for g in got])
    assert np.array_equal(pos, rec[f"{tag}.pos"])
    assert np.array_equal(np.concatenate([g["selected_class"] for g in got]), rec[f"{tag}.cls"])
    data = torch.cat([g["data"] for g in got]).cpu().numpy()
    label = torch.cat([g["label"] for g in got]).cpu().numpy()
    assert data.dtype == np.float32 and label.dtype == np.uint8
    for i, it in enumerate(items):
        assert np.array_equal(data[i], it["data"]) and np.array_equal(label[i], it["label"]), (tag, i)
    assert np.array_equal(data[0], rec[f"{tag}.first_data"]) and np.array_equal(label[0], rec[f"{tag}.first_label"])


def test_device_augmentation_matches_the_numpy_restatement():
    """Row N1, the reference's `transform` (examples/train_seg.py:82-86: batchgenerators brightness -> gamma -> contrast on
    'data', dataset.py:340-341): mednet_augment_patches against oracle/ref_augment.py on the same parameters (parity of the
    oracle itself is UNPINNED: batchgenerators is not available; see the oracle's header), plus the properties the
    transforms guarantee: the contrast step keeps every channel inside its range, identity parameters leave the patch alone."""
    from mednet_hip import sampler as HS
    from oracle import ref_augment as A
    g = np.random.Generator(np.random.PCG64(11))
    data = (g.standard_normal((3, 2, 9, 17, 23)) * 40 + 100).astype(np.float32)
    np.random.seed(5)
    params = A.draw_parameters(3, 2)
    want = A.apply(data, params)
    got = HS.augment_(torch.from_numpy(data.copy()).to(DEV), params).cpu().numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err <= 2e-5, err  # (powf against numpy's float32 power)
    for b in range(3):
        for c in range(2):
            assert got[b, c].min() >= want[b, c].min() - 1e-3 and got[b, c].max() <= want[b, c].max() + 1e-3
    ident = np.zeros((3, 2, 3), dtype=np.float32)
    ident[..., 1:] = 1.0
    same = HS.augment_(torch.from_numpy(data.copy()).to(DEV), ident).cpu().numpy()
    assert np.abs(same - data).max() <= 1e-4 * np.abs(data).max()


def test_device_patch_sampler_with_augmentation_follows_the_reference_call_order():
    """DevicePatchSampler(augment=True): per sample the position draws come first, then the transform's draws (the order of
    MedDataset.__getitem__, dataset.py:285-341), so the crop positions differ from an un-augmented run exactly as the
    reference's would; the batch equals the oracle sampler's crop pushed through the augmentation oracle."""
    from mednet_hip.sampler import DevicePatchSampler
    from oracle import ref_augment as A
    from oracle import ref_sampler as S
    tag, shapes, c_img, n_hm, patch, probs, draws, seed = S.SAMPLER_CASES[0]
    images, labels, heatmaps = S.sampler_volumes(tag, shapes, c_img, n_hm, len(probs) if probs else 3)
    ora = S.PatchSampler(images, labels, patch, samples_per_subject=4, heatmaps=heatmaps, class_probabilities=probs)
    dev = DevicePatchSampler(images, labels, patch, samples_per_subject=4, heatmaps=heatmaps, class_probabilities=probs,
                             device=DEV, augment=True)
    np.random.seed(seed)
    want = []
    for i in range(4):
        item = ora[i]                                  # position draws + crop (the oracle sampler)
        prm = A.draw_parameters(1, item["data"].shape[0])  # then the transform's draws for this sample
        want.append((item, A.apply(item["data"][None], prm)[0]))
    np.random.seed(seed)
    got = dev.batch([0, 1, 2, 3])
    for i, (item, aug) in enumerate(want):
        assert np.array_equal(got["patch_position"][i], item["patch_position"])
        assert np.array_equal(got["label"][i].cpu().numpy(), item["label"])
        d = got["data"][i].cpu().numpy()
        assert np.abs(d - aug).max() <= 2e-5 * max(1.0, np.abs(aug).max()), i
