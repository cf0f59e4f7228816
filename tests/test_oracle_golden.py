"""The CPU oracle (oracle/ref_cpu.py) against the committed golden vectors that tools/make_golden.py captured from
the REAL reference (bit-identical at capture time).  Runs without /root/reference."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import ref_cpu as O

RTOL = 2e-5  # other host CPUs may pick other oneDNN kernels than the capture box; fp32 reassociation noise only


def _proj(name, n):
    return np.random.Generator(np.random.PCG64(zlib.crc32(("proj:" + name).encode()))).standard_normal(n)


def _check_summary(rec, key, t):
    a = t.detach().numpy().astype(np.float64).reshape(-1)
    norm = float(rec[key + ".norm"])
    assert abs(np.sqrt((a * a).sum()) - norm) <= RTOL * max(norm, 1e-12), key
    head = rec[key + ".head"]
    np.testing.assert_allclose(a[: head.size], head, rtol=1e-3, atol=RTOL * max(norm, 1e-12) )
    proj = float(rec[key + ".proj"])
    assert abs(a @ _proj(key.split(".", 1)[1] if key.startswith("grad.") else key, a.size) - proj) <= 1e-4 * max(norm, 1e-12) * np.sqrt(a.size) ** 0 + 1e-3 * max(norm, 1e-12), key
    if key + ".full" in rec.files:
        assert O.rel_l2(t, rec[key + ".full"]) <= RTOL, key


NETS = {
    "res_cfg1": (O.ResidualUNet3D, dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8]), 2, 0, "dice", [0.05, 1.0]),
    "res_cfg1_ce": (O.ResidualUNet3D, dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8]), 2, 0, "ce", [0.05, 1.0]),
    "res_small": (O.ResidualUNet3D, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64]), 4, 0, "dice", [0.05, 1, 1, 1]),
    "res_odd": (O.ResidualUNet3D, dict(in_channels=2, out_channels=3, final_sigmoid=False, f_maps=[8, 16, 24]), 3, 0, "dice", None),
    "res_cfg2_32": (O.ResidualUNet3D, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]), 4, 0, "dice", [0.05, 1, 1, 1]),
    "res_cfg4_32": (O.ResidualUNet3D, dict(in_channels=1, out_channels=18, final_sigmoid=False, f_maps=[32, 64, 128, 256]), 2, 16, "ldmk", [0.05, 1.0]),
    "res_ldmk_l1": (O.ResidualUNet3D, dict(in_channels=1, out_channels=5, final_sigmoid=False, f_maps=[8, 16]), 2, 3, "ldmk_l1", [0.05, 1.0]),
    "unet_cfg1": (O.UNet3D, dict(in_channels=1, out_channels=2, final_sigmoid=False, f_maps=[8]), 2, 0, "dice", [0.05, 1.0]),
    "unet_small": (O.UNet3D, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64]), 4, 0, "dice", [0.05, 1, 1, 1]),
    "unet_oddsize": (O.UNet3D, dict(in_channels=1, out_channels=3, final_sigmoid=False, f_maps=[8, 16, 32]), 3, 0, "dice", None),
    "unet_cfg2_32": (O.UNet3D, dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]), 4, 0, "dice", [0.05, 1, 1, 1]),
}


def run_oracle_case(tag, golden_dir):
    cls, ctor, ncls, nh, lk, w = NETS[tag]
    rec = np.load(os.path.join(golden_dir, tag + ".npz"))
    shape = tuple(int(v) for v in rec["meta.shape"])
    n = int(rec["meta.n"])
    model = O.keyed_init_(cls(**ctor))
    batch = O.synthetic_batch(n, ctor["in_channels"], shape, ncls, nh, seed=int(rec["meta.seed"]))
    wt = None if w is None else torch.tensor(w, dtype=torch.float32)
    x = batch["data"].float()
    y = batch["label"][:, -1].long()
    logits = model(x)
    if lk == "dice":
        loss = O.DiceLoss(weight=wt)(logits, y)
    elif lk == "ce":
        loss = nn.CrossEntropyLoss(weight=wt)(logits, y)
    else:
        hm = batch["label"][:, :-1].float()
        reg = nn.MSELoss() if lk == "ldmk" else nn.L1Loss()
        loss, _, _ = O.landmark_loss(logits[:, nh:], logits[:, :nh], y, hm, O.DiceLoss(weight=wt), reg, [0.015] * nh)
    loss.backward()
    return rec, model, logits, loss


@pytest.mark.parametrize("tag", sorted(NETS))
def test_oracle_matches_reference_vectors(tag, golden_dir):
    rec, model, logits, loss = run_oracle_case(tag, golden_dir)
    assert abs(float(loss) - float(rec["loss"])) <= 1e-5 * max(abs(float(rec["loss"])), 1.0)
    _check_summary(rec, "logits", logits)
    for name, p in model.named_parameters():
        _check_summary(rec, "grad." + name, p.grad)


def test_oracle_adam_delta(golden_dir):
    rec, model, _, _ = run_oracle_case("res_cfg1", golden_dir)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    torch.optim.Adam(model.parameters(), lr=1e-3).step()
    for k, p in model.named_parameters():
        np.testing.assert_allclose((p.detach() - before[k]).numpy(), rec["adam_delta." + k], rtol=1e-3, atol=2e-6)


def test_oracle_blocks(golden_dir):
    rec = np.load(os.path.join(golden_dir, "blocks.npz"))
    def rnd(tag, *shape):
        return torch.from_numpy(O._rng("in:" + tag).standard_normal(shape).astype(np.float32))
    cases = {
        "resblock_cge": (O.ExtResNetBlock(8, 16, order="cge"), [rnd("resblock_cge", 2, 8, 6, 8, 10)]),
        "single_gcr": (O.SingleConv(8, 16, 3, "gcr", 8), [rnd("single_gcr", 2, 8, 6, 10, 12)]),
        "decoder_res": (O.Decoder(16, 8, basic_module=O.ExtResNetBlock, conv_layer_order="cge"),
                        [rnd("decoder_res_e", 2, 8, 8, 12, 10), rnd("decoder_res_x", 2, 16, 4, 6, 5)]),
        "decoder_double": (O.Decoder(24, 8, basic_module=O.DoubleConv, conv_layer_order="gcr"),
                           [rnd("decoder_double_e", 1, 8, 7, 9, 10), rnd("decoder_double_x", 1, 16, 3, 4, 5)]),
    }
    for tag, (m, inputs) in cases.items():
        O.keyed_init_(m)
        xs = [t.clone().requires_grad_(True) for t in inputs]
        y = m(*xs)
        g = torch.from_numpy(O._rng("cot:" + tag).standard_normal(tuple(y.shape)).astype(np.float32))
        (y * g).sum().backward()
        assert O.rel_l2(y, rec[f"{tag}.y"]) <= RTOL
        for i, t in enumerate(xs):
            assert O.rel_l2(t.grad, rec[f"{tag}.dx{i}"]) <= RTOL
        for k, p in m.named_parameters():
            assert O.rel_l2(p.grad, rec[f"{tag}.dp.{k}"]) <= 1e-4


def test_oracle_losses(golden_dir):
    rec = np.load(os.path.join(golden_dir, "losses.npz"))
    z = torch.from_numpy(rec["logits"])
    y = torch.from_numpy(rec["labels"])
    w = torch.tensor([0.05, 1.0, 1.0, 1.0])
    table = {
        "dice_plain": O.DiceLoss(), "dice_weight": O.DiceLoss(weight=w),
        "dice_sigmoid": O.DiceLoss(weight=w, sigmoid_normalization=True),
        "dice_ignore": O.DiceLoss(weight=w, ignore_index=1), "dice_eps": O.DiceLoss(epsilon=1e-2),
        "ce_weight": nn.CrossEntropyLoss(weight=w), "ce_plain": nn.CrossEntropyLoss(),
    }
    for tag, fn in table.items():
        zz = z.clone().requires_grad_(True)
        v = fn(zz, y)
        v.backward()
        assert abs(float(v) - float(rec[tag + ".value"])) <= 1e-6, tag
        assert O.rel_l2(zz.grad, rec[tag + ".grad"]) <= RTOL, tag
    np.testing.assert_allclose(O.dice_metric(z, y).numpy(), rec["dice_metric.value"], rtol=1e-5)
    np.testing.assert_array_equal(O.expand_as_one_hot(y, 4).numpy(), rec["onehot"])
    np.testing.assert_array_equal(O.expand_as_one_hot(y, 4, ignore_index=2).numpy(), rec["onehot_ignore"])
    assert int(rec["dice_skip_last.raises"]) == 1


def test_oracle_caller_steps(golden_dir):
    rec = np.load(os.path.join(golden_dir, "callers.npz"))
    ora = O.keyed_init_(O.ResidualUNet3D(1, 2, False, f_maps=[8]))
    batch = O.synthetic_batch(2, 1, (32, 32, 32), 2, 0, seed=1234)
    loss = O.seg_training_step(ora, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), batch)
    assert abs(float(loss) - float(rec["seg.loss"])) <= 1e-5
    np.testing.assert_allclose(rec["seg.adam"], [1e-3, 0.9, 0.999, 1e-8, 0.0])
    ora2 = O.keyed_init_(O.ResidualUNet3D(1, 5, False, f_maps=[8]))
    batch2 = O.synthetic_batch(2, 1, (16, 16, 16), 2, 3, seed=4321)
    tot, cl, rg = O.ldmk_training_step(ora2, O.DiceLoss(weight=torch.tensor([0.05, 1.0])), nn.MSELoss(), [0.015] * 3, batch2)
    assert abs(float(tot) - float(rec["ldmk.loss"])) <= 1e-4 * float(rec["ldmk.loss"])
    assert abs(float(cl) - float(rec["ldmk.class_loss"])) <= 1e-5
    assert abs(float(rg) - float(rec["ldmk.regression_loss"])) <= 1e-4 * float(rec["ldmk.regression_loss"])


def test_predict_oracle_matches_reference_golden(golden_dir):
    """Row N2: oracle/ref_predict.py (grid patches, post-processing, stitching with the reference's crop rule) against
    tests/golden/predict.npz, which tools/make_golden.py wrote only after the reference's own functions agreed."""
    import os
    from oracle import ref_predict as P
    rec = np.load(os.path.join(golden_dir, "predict.npz"))
    for tag, shape, patch, ov, mode, nh, ncls, bs in P.PREDICT_CASES:
        img, logits_for = P.predict_inputs(tag, shape, patch, nh, ncls)
        patches = list(P.grid_patch_generator(img, patch, ov, mode=mode))
        assert len(patches) == int(rec[f"{tag}.npatches"])
        assert np.array_equal(np.stack([i for _, i, _ in patches]), rec[f"{tag}.pos"])
        assert np.allclose([float(np.asarray(a, dtype=np.float64).sum()) for a, _, _ in patches], rec[f"{tag}.patch_sums"])
        assert np.array_equal(patches[0][0], rec[f"{tag}.first_patch"]) and np.array_equal(patches[-1][0], rec[f"{tag}.last_patch"])
        result = np.zeros((nh + 1,) + tuple(shape[1:]), dtype=np.uint8)
        for b0 in range(0, len(patches), bs):
            chunk = patches[b0:b0 + bs]
            out = P.postprocess(np.stack([logits_for(c) for _, _, c in chunk]), nh)
            P.add_processed_batch(result, out, np.stack([i for _, i, _ in chunk]), ov)
        assert np.array_equal(result, rec[f"{tag}.result"]), tag


def test_sampler_oracle_matches_reference_golden(golden_dir):
    """Row N1: oracle/ref_sampler.py seeded like the reference run of tools/make_golden.py reproduces its patches."""
    import os
    from oracle import ref_sampler as S
    rec = np.load(os.path.join(golden_dir, "sampler.npz"))
    for tag, shapes, c_img, n_hm, patch, probs, draws, seed in S.SAMPLER_CASES:
        images, labels, heatmaps = S.sampler_volumes(tag, shapes, c_img, n_hm, len(probs) if probs else 3)
        ora = S.PatchSampler(images, labels, patch, samples_per_subject=4, heatmaps=heatmaps, class_probabilities=probs)
        np.random.seed(seed)
        items = [ora[i] for i in range(draws)]
        assert np.array_equal(np.stack([a["patch_position"] for a in items]), rec[f"{tag}.pos"])
        assert np.array_equal([int(a["selected_class"]) for a in items], rec[f"{tag}.cls"])
        assert np.array_equal([int(a["subject_key"]) for a in items], rec[f"{tag}.subj"])
        assert np.allclose([float(a["data"].astype(np.float64).sum()) for a in items], rec[f"{tag}.data_sum"])
        assert np.array_equal([int(a["label"].astype(np.int64).sum()) for a in items], rec[f"{tag}.label_sum"])
        assert np.array_equal(items[0]["data"], rec[f"{tag}.first_data"]) and np.array_equal(items[0]["label"], rec[f"{tag}.first_label"])


def test_augment_oracle_properties():
    """oracle/ref_augment.py (batchgenerators' brightness / gamma / contrast as composed at examples/train_seg.py:82-86) is
    PARITY UNPINNED -- the library is neither vendored nor installed.  What can be pinned: the documented properties of
    the three transforms, and that the product's host-side draw routine consumes numpy's generator exactly like the oracle's."""
    from oracle import ref_augment as A
    from mednet_hip import sampler as HS
    g = np.random.Generator(np.random.PCG64(3))
    data = (g.standard_normal((2, 3, 6, 7, 8)) * 10 + 50).astype(np.float32)
    np.random.seed(9)
    prm = A.draw_parameters(2, 3)
    assert np.all((prm[..., 1] >= 0.7) & (prm[..., 1] <= 1.3)) and np.all((prm[..., 2] >= 0.3) & (prm[..., 2] <= 1.7))
    assert np.all(prm[:, :, 1] == prm[:, :1, 1])  # one gamma per sample
    out = A.apply(data, prm)
    # identity parameters: nothing changes (up to the 1e-7 epsilon of the gamma map)
    ident = np.zeros_like(prm)
    ident[..., 1:] = 1.0
    assert np.abs(A.apply(data, ident) - data).max() <= 1e-4 * np.abs(data).max()
    # brightness alone shifts a channel; gamma keeps the sample's range; contrast keeps each channel's range and (unclipped) mean
    only_b = ident.copy()
    only_b[..., 0] = prm[..., 0]
    assert np.allclose(A.apply(data, only_b) - data, prm[..., 0][..., None, None, None], atol=2e-4)
    only_g = ident.copy()
    only_g[..., 1] = prm[..., 1]
    og = A.apply(data, only_g)
    for b in range(2):
        assert abs(og[b].min() - data[b].min()) <= 1e-3 and abs(og[b].max() - data[b].max()) <= 1e-3
    only_c = ident.copy()
    only_c[..., 2] = np.minimum(prm[..., 2], 1.0)  # shrinking never clips: the mean survives exactly
    oc = A.apply(data, only_c)
    assert np.allclose(oc.mean(axis=(2, 3, 4)), data.mean(axis=(2, 3, 4)), rtol=1e-4)
    for b in range(2):  # the full Compose: every channel stays inside the sample's range after the brightness shift
        lo = (data[b] + prm[b, :, 0][:, None, None, None]).min()
        hi = (data[b] + prm[b, :, 0][:, None, None, None]).max()
        assert out[b].min() >= lo - 1e-3 and out[b].max() <= hi + 1e-3
    # same consumption of numpy's generator as the product's per-sample routine
    np.random.seed(21)
    a = A.draw_parameters(3, 2)
    np.random.seed(21)
    b_ = np.stack([HS.draw_augmentation(2) for _ in range(3)])
    assert np.array_equal(a, b_)
    assert (HS.AUG_BRIGHTNESS, HS.AUG_GAMMA_RANGE, HS.AUG_CONTRAST_RANGE) == ((A.BRIGHTNESS["mu"], A.BRIGHTNESS["sigma"]), A.GAMMA_RANGE, A.CONTRAST_RANGE)
