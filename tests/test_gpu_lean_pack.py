"""MEDNET_PACK_HIGH_ONLY (round 6): after an optimizer step the trainer rewrites only the weight images the 16-bit matrix-core kernels
read (train.BatchedRepack); the fp32 images of the direct / fp32-matrix kernels and unrequested low images stay STALE.  The step must
not notice, and a call that takes another path with such a pack -- fp32 tensors (split-bf16 kernels: high + LOW images), the direct
kernels -- must get the layer packed in full first (nn._PackedWeightMixin._packed(x)) and agree bit for bit with a layer that never
saw a lean pack."""
import copy

import pytest
import torch

import mednet_hip
from mednet_hip import nn as hnn
from mednet_hip import train as T
from mednet_hip.unet import model as HM
from oracle import ref_cpu as O

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _steps(mode, lean, n_steps=3):
    old = T.BatchedRepack.LEAN
    T.BatchedRepack.LEAN = lean
    try:
        with mednet_hip.precision(mode):
            net = O.keyed_init_(HM.ResidualUNet3D(1, 3, False, f_maps=[32, 64])).to(DEV)
            step = T.SegmentationStep(net, loss_weight=[0.1, 1.0, 1.0], lr=1e-2)
            b = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (32, 32, 32), 3, 0, seed=5).items()}
            losses = [float(step(b)) for _ in range(n_steps)]
            torch.cuda.synchronize()
            params = {k: p.detach().clone() for k, p in net.named_parameters()}
            flags = {name: bool(getattr(m._pack_buf, "_mednet_lean", False)) for name, m in net.named_modules()
                     if isinstance(m, hnn._PackedWeightMixin) and getattr(m, "_pack_buf", None) is not None}
        return net, step, losses, params, flags
    finally:
        T.BatchedRepack.LEAN = old


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp16x2"])
def test_lean_packs_change_nothing_in_the_step_and_full_images_come_back_when_needed(mode):
    net_a, step_a, loss_a, par_a, flags_a = _steps(mode, True)
    net_b, step_b, loss_b, par_b, flags_b = _steps(mode, False)
    assert loss_a == loss_b, (loss_a, loss_b)
    for k in par_a:
        assert torch.equal(par_a[k], par_b[k]), f"{k}: parameters after three steps differ between lean and full packs"
    assert any(flags_a.values()) and not any(flags_b.values()), (flags_a, flags_b)
    # the first layer (one input channel) and the 1x1x1 head are not in the table: never lean
    assert not flags_a["encoders.0.basic_module.conv1.conv"] and not flags_a["final_conv"], flags_a

    # a 3x3x3 layer of the trained model, its pack lean: call it (a) with fp32 tensors, (b) through the direct kernels
    name, conv = next((n, m) for n, m in net_a.named_modules() if isinstance(m, hnn.Conv3d) and flags_a.get(n))
    fresh = hnn.Conv3d(conv.in_channels, conv.out_channels, 3, bias=False).to(DEV)  # never saw a lean pack
    with torch.no_grad():
        fresh.weight.copy_(conv.weight)
    g = torch.Generator(device=DEV).manual_seed(3)
    x32 = torch.randn(1, conv.in_channels, 8, 16, 16, device=DEV, generator=g)
    with mednet_hip.precision(mode):
        assert conv._pack_buf._mednet_lean
        y_a, y_b = conv(x32), fresh(x32)  # fp32 input in a 16-bit mode: the unfused conv passes its dtypes (split-bf16 / fp32 kernels)
        assert not conv._pack_buf._mednet_lean, "the fp32 call did not re-pack the layer"
        assert torch.equal(y_a, y_b), f"{name}: fp32 call after lean packs differs from a freshly packed layer"
        step_a({k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (32, 32, 32), 3, 0, seed=5).items()})  # lean again
        assert conv._pack_buf._mednet_lean
        with torch.no_grad():
            fresh.weight.copy_(conv.weight)
        mednet_hip.set_conv_algo("direct")
        try:
            x16 = x32.to(mednet_hip.config.act_dtype())
            y_a, y_b = conv(x16), fresh(x16)
        finally:
            mednet_hip.set_conv_algo("auto")
        assert not conv._pack_buf._mednet_lean, "the direct-kernel call did not re-pack the layer"
        assert torch.equal(y_a, y_b), f"{name}: direct kernels after lean packs differ from a freshly packed layer"
        # ConvTranspose3d of the decoder: lean, and its 16-bit call leaves it lean
        ct = next(m for m in net_a.modules() if isinstance(m, hnn.ConvTranspose3d))
        step_a({k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (32, 32, 32), 3, 0, seed=5).items()})
        assert ct._pack_buf._mednet_lean and conv._pack_buf._mednet_lean
        xt = torch.randn(1, ct.in_channels, 4, 8, 8, device=DEV, generator=g).to(mednet_hip.config.act_dtype())
        fresh_t = hnn.ConvTranspose3d(ct.in_channels, ct.out_channels).to(DEV)
        with torch.no_grad():
            fresh_t.weight.copy_(ct.weight)
            fresh_t.bias.copy_(ct.bias)
        assert torch.equal(ct(xt), fresh_t(xt)) and ct._pack_buf._mednet_lean
    step_a.flat.release()
    step_b.flat.release()
