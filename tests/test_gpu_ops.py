"""Per-op parity of every libmednet_hip entry point against the CPU oracle's ATen ops (fp32 storage: <= 1e-4,
bf16 storage: <= 2.5e-2 rel-L2), including the edge cases the U-Net hits: Cin=1, odd channel counts, odd spatial
sizes, GroupNorm's single-group fallback, pooling tails and ties, strided logits slices."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import mednet_hip
from mednet_hip import _lib as L
from mednet_hip import nn as hnn
from mednet_hip import ops
from mednet_hip.unet import loss as HL
from oracle import ref_cpu as O

from gpu_util import DEV, TOL, assert_close, bf16_round, half_round, rnd, rel

pytestmark = pytest.mark.gpu
MODES = ["fp32", "bf16", "fp16"]


def _prep(mode, *ts):
    return [half_round(t, mode) for t in ts]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,cin,cout,shape,bias", [
    (2, 1, 8, (6, 10, 12), False), (1, 8, 16, (5, 7, 9), True), (1, 32, 32, (8, 8, 16), False),
    (2, 3, 5, (4, 6, 7), True), (1, 16, 64, (4, 4, 4), False), (1, 64, 32, (3, 5, 8), False)])
def test_conv3d_k3(mode, n, cin, cout, shape, bias):
    tag = f"conv{n}{cin}{cout}{shape}"
    x, w, cot = _prep(mode, rnd(tag + "x", n, cin, *shape), rnd(tag + "w", cout, cin, 3, 3, 3, scale=0.2),
                      rnd(tag + "g", n, cout, *shape))
    b = rnd(tag + "b", cout) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv3d(xr, wr, br, padding=1)
    (yr * cot).sum().backward()
    with mednet_hip.precision(mode):
        conv = hnn.Conv3d(cin, cout, 3, bias=bias).to(DEV)
        with torch.no_grad():
            conv.weight.copy_(w)
            if bias:
                conv.bias.copy_(b)
        xg = x.to(DEV).requires_grad_(True)
        y = conv(xg)
        assert y.shape == yr.shape and y.dtype == mednet_hip.config.act_dtype()
        (y.float() * cot.to(DEV)).sum().backward()
    tol = TOL[mode]
    assert_close(y, yr, tol, "y")
    assert_close(xg.grad, xr.grad, tol, "dx")
    assert_close(conv.weight.grad, wr.grad, tol, "dw")
    if bias:
        assert_close(conv.bias.grad, br.grad, tol, "db")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,cin,cout,shape", [(2, 32, 4, (6, 8, 10)), (1, 8, 18, (5, 6, 7)), (1, 8, 3, (4, 4, 5)),
                                              (1, 8, 2, (8, 8, 8))])
def test_conv1x1_head_planar_logits(mode, n, cin, cout, shape):
    tag = f"head{n}{cin}{cout}"
    x, w = _prep(mode, rnd(tag + "x", n, cin, *shape), rnd(tag + "w", cout, cin, 1, 1, 1, scale=0.3))
    b, cot = rnd(tag + "b", cout), rnd(tag + "g", n, cout, *shape)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br)
    (yr * cot).sum().backward()
    with mednet_hip.precision(mode):
        conv = hnn.Conv3d(cin, cout, 1, planar_output=True).to(DEV)
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.copy_(b)
        xg = x.to(DEV).to(mednet_hip.config.act_dtype()).requires_grad_(True)
        y = conv(xg)
        assert y.dtype == torch.float32 and y.is_contiguous()
        (y * cot.to(DEV)).sum().backward()
    tol = TOL[mode]
    assert_close(y, yr, 1e-4 if mode == "fp32" else 1e-2, "logits")
    assert_close(xg.grad, xr.grad, tol, "dx")
    assert_close(conv.weight.grad, wr.grad, tol, "dw")
    assert_close(conv.bias.grad, br.grad, 1e-4, "db")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,cin,cout,shape,skip", [(2, 16, 8, (3, 5, 4), True), (1, 8, 8, (4, 4, 4), False),
                                                   (1, 64, 32, (2, 3, 5), True)])
def test_conv_transpose3d_with_skip(mode, n, cin, cout, shape, skip):
    tag = f"ct{n}{cin}{cout}"
    oshape = tuple(2 * s for s in shape)
    x, w, sk, cot = _prep(mode, rnd(tag + "x", n, cin, *shape), rnd(tag + "w", cin, cout, 3, 3, 3, scale=0.2),
                          rnd(tag + "s", n, cout, *oshape), rnd(tag + "g", n, cout, *oshape))
    b = rnd(tag + "b", cout)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    skr = sk.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2, padding=1, output_padding=1)
    if skip:
        yr = yr + skr
    (yr * cot).sum().backward()
    with mednet_hip.precision(mode):
        up = hnn.ConvTranspose3d(cin, cout).to(DEV)
        with torch.no_grad():
            up.weight.copy_(w)
            up.bias.copy_(b)
        dt = mednet_hip.config.act_dtype()
        xg = x.to(DEV).to(dt).requires_grad_(True)
        skg = sk.to(DEV).to(dt).requires_grad_(True)
        y = up(xg, skip=skg if skip else None)
        (y.float() * cot.to(DEV)).sum().backward()
    tol = TOL[mode]
    assert_close(y, yr, tol, "y")
    assert_close(xg.grad, xr.grad, tol, "dx")
    assert_close(up.weight.grad, wr.grad, tol, "dw")
    assert_close(up.bias.grad, br.grad, 1e-4 if mode == "fp32" else 1e-3, "db")
    if skip:
        assert_close(skg.grad, skr.grad, tol, "dskip")


def _set_option(name, value):
    L.check(L.lib().mednet_set_option(name.encode(), int(value)), "set_option")


@pytest.mark.parametrize("n,cin,cout,shape", [(2, 32, 32, (9, 11, 21)), (1, 48, 16, (5, 6, 7)), (3, 16, 48, (4, 9, 17)),
                                              (1, 64, 96, (8, 8, 16))])
def test_split_bf16_conv_family_in_the_fp32_mode(n, cin, cout, shape):
    """fp32 storage mode, channel counts that are multiples of 16: forward, data gradient and weight gradient run as
    split-bf16 contractions (csrc/conv_x3_mfma.hip: hi*hi + hi*lo + lo*hi on the bf16 matrix cores).  Against ATen in fp64
    at 3e-5 (the mode's budget is 1e-3 after 21 layers), ragged bricks and several samples included, and against the exact
    fp32 matrix-core kernels (option x3=0) to show that a different kernel really ran."""
    tag = f"x3{n}{cin}{cout}{shape}"
    x, w, cot = rnd(tag + "x", n, cin, *shape), rnd(tag + "w", cout, cin, 3, 3, 3, scale=0.1), rnd(tag + "g", n, cout, *shape)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, padding=1)
    (yr * cot.double()).sum().backward()
    res = {}
    try:
        for x3 in (1, 0):
            _set_option("x3", x3)
            with mednet_hip.precision("fp32"):
                conv = hnn.Conv3d(cin, cout, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                xg = x.to(DEV).requires_grad_(True)
                y = conv(xg)
                (y * cot.to(DEV)).sum().backward()
                res[x3] = (y.detach().cpu(), xg.grad.cpu(), conv.weight.grad.cpu())
    finally:
        _set_option("x3", 1)
    for name, a, b, ref in zip(("y", "dx", "dw"), res[1], res[0], (yr, xr.grad, wr.grad)):
        assert_close(a.double(), ref, 3e-5, f"split-bf16 {name}")
        assert_close(b.double(), ref, 3e-6, f"fp32 mfma {name}")
        assert not torch.equal(a, b), f"{name}: option x3 selected no other kernel"


@pytest.mark.parametrize("n,cout,shape", [(2, 32, (9, 11, 21)), (1, 8, (5, 6, 7)), (3, 64, (4, 9, 17))])
def test_split_bf16_first_layer_in_the_fp32_mode(n, cout, shape):
    """The network's first convolution (one input channel, model.py:171-174) in the fp32 mode: contraction over the 27 taps as a
    split-bf16 product (conv_c1_x3_kernel), with the GroupNorm partial sums of its output; against ATen in fp64."""
    tag = f"x3c1{n}{cout}{shape}"
    x, w = rnd(tag + "x", n, 1, *shape), rnd(tag + "w", cout, 1, 3, 3, 3, scale=0.3)
    yr = F.conv3d(x.double(), w.double(), None, padding=1)
    res = {}
    try:
        for x3 in (1, 0):
            _set_option("x3", x3)
            with mednet_hip.precision("fp32"):
                conv = hnn.Conv3d(1, cout, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                y, partial = conv.forward_with_stats(x.to(DEV))
                res[x3] = (y.detach().cpu(), None if partial is None else partial.detach().cpu())
    finally:
        _set_option("x3", 1)
    assert_close(res[1][0].double(), yr, 3e-5, "split-bf16 first layer")
    assert_close(res[0][0].double(), yr, 3e-6, "fp32 mfma first layer")
    assert not torch.equal(res[1][0], res[0][0])
    part = res[1][1]
    assert part is not None and part.shape[0] == n and part.shape[2] == cout
    sums = part.double().sum(1)  # [n][cout][2]
    ys = res[1][0].double()
    assert_close(sums[..., 0], ys.sum((2, 3, 4)), 1e-5, "sum y from the kernel's partial rows")
    assert_close(sums[..., 1], (ys * ys).sum((2, 3, 4)), 1e-5, "sum y^2 from the kernel's partial rows")


@pytest.mark.parametrize("n,shape", [(2, (5, 9, 17)), (1, (8, 16, 32)), (3, (3, 7, 5))])
def test_split_bf16_first_layer_weight_gradient_in_the_fp32_mode(n, shape):
    """Weight gradient of the first convolution (one input channel, 32 outputs) in the fp32 mode: contraction over the voxels as
    a split-bf16 product (wgrad_c1_x3_kernel: bricks of 4x8x16, several bricks per workgroup, ragged edges), against ATen in
    fp64 and against the exact-product kernel (option x3=0), from which it must differ in the last bits."""
    tag = f"x3c1w{n}{shape}"
    x, w, cot = rnd(tag + "x", n, 1, *shape), rnd(tag + "w", 32, 1, 3, 3, 3, scale=0.3), rnd(tag + "g", n, 32, *shape)
    wr = w.double().requires_grad_(True)
    F.conv3d(x.double(), wr, None, padding=1).backward(cot.double())
    res = {}
    try:
        for x3 in (1, 0):
            _set_option("x3", x3)
            with mednet_hip.precision("fp32"):
                conv = hnn.Conv3d(1, 32, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                conv(x.to(DEV)).backward(cot.to(DEV))
                res[x3] = conv.weight.grad.detach().cpu()
    finally:
        _set_option("x3", 1)
    assert_close(res[1].double(), wr.grad, 3e-5, "split-bf16 first-layer weight gradient")
    assert_close(res[0].double(), wr.grad, 3e-6, "exact-product first-layer weight gradient")
    assert not torch.equal(res[1], res[0])


@pytest.mark.parametrize("n,cin,cout,shape", [(2, 32, 16, (3, 5, 9)), (1, 64, 32, (4, 4, 8)), (1, 16, 48, (5, 3, 17)),
                                              (2, 128, 64, (3, 6, 16))])
def test_split_bf16_conv_transpose_in_the_fp32_mode(n, cin, cout, shape):
    """ConvTranspose3d + bias + skip (components.py:259-264,283-284) in the fp32 mode: forward, data gradient and weight
    gradient (output-parity planes of the gradient, one or two 32-channel blocks of x per workgroup) on the split-bf16
    kernels."""
    tag = f"x3ct{n}{cin}{cout}{shape}"
    oshape = tuple(2 * s for s in shape)
    x, w = rnd(tag + "x", n, cin, *shape), rnd(tag + "w", cin, cout, 3, 3, 3, scale=0.1)
    b, sk, cot = rnd(tag + "b", cout), rnd(tag + "s", n, cout, *oshape), rnd(tag + "g", n, cout, *oshape)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2, padding=1, output_padding=1) + sk.double()
    (yr * cot.double()).sum().backward()
    res = {}
    try:
        for x3 in (1, 0):
            _set_option("x3", x3)
            with mednet_hip.precision("fp32"):
                up = hnn.ConvTranspose3d(cin, cout).to(DEV)
                with torch.no_grad():
                    up.weight.copy_(w)
                    up.bias.copy_(b)
                xg = x.to(DEV).requires_grad_(True)
                y = up(xg, skip=sk.to(DEV))
                (y * cot.to(DEV)).sum().backward()
                res[x3] = (y.detach().cpu(), xg.grad.cpu(), up.weight.grad.cpu())
    finally:
        _set_option("x3", 1)
    for name, a, b2, ref in zip(("y", "dx", "dw"), res[1], res[0], (yr, xr.grad, wr.grad)):
        assert_close(a.double(), ref, 3e-5, f"split-bf16 {name}")
        assert_close(b2.double(), ref, 3e-6, f"fp32 mfma {name}")
    assert not torch.equal(res[1][0], res[0][0]) and not torch.equal(res[1][1], res[0][1]) and not torch.equal(res[1][2], res[0][2])


@pytest.mark.parametrize("n,c,shape", [(1, 32, (32, 64, 64)), (2, 64, (16, 32, 64)), (2, 256, (16, 32, 32))])
def test_split_bf16_kernels_keep_the_groupnorm_sums(n, c, shape):
    """fp32 mode, enough bricks per workgroup: the split-bf16 forward kernel also writes the GroupNorm partial sums of what it
    stores, and its data-gradient form sums the residual-branch gradient in and takes the first pass of the previous
    GroupNorm's backward (no stand-alone statistics / partial passes).  One, two and eight channel blocks (the row plan differs:
    a group of bricks per item range, or several); ExtResNetBlock against the oracle, with and without the fusions."""
    from mednet_hip.unet import components as HC
    assert L.lib().mednet_conv3d_fused_stats_chunks(n, *shape, c, c, 3, L.F32, L.F32, L.ALGO_AUTO) > 0
    assert L.lib().mednet_conv3d_dgrad_gn_rows_dt(n, *shape, c, c, L.ALGO_AUTO, L.F32) > 0
    x = rnd(f"x3stats{c}", n, c, *shape)
    ora = O.keyed_init_(O.ExtResNetBlock(c, c, order="cge"))
    xo = x.clone().requires_grad_(True)
    yo = ora(xo)
    g = rnd("x3statsg", *yo.shape)
    (yo * g).sum().backward()
    res = {}
    try:
        for fuse in (1, 0):
            _set_option("x3_stats", fuse)
            _set_option("conv_fuse_gnb", fuse)
            with mednet_hip.precision("fp32"):
                blk = O.keyed_init_(HC.ExtResNetBlock(c, c, order="cge")).to(DEV)
                xg = x.to(DEV).requires_grad_(True)
                y = blk(xg)
                (y * g.to(DEV)).sum().backward()
                res[fuse] = [y.detach().cpu(), xg.grad.cpu()] + [p.grad.cpu() for p in blk.parameters()]
    finally:
        _set_option("x3_stats", 1)
        _set_option("conv_fuse_gnb", 1)
    want = [yo, xo.grad] + [p.grad for p in ora.parameters()]
    for i, (a, b, r) in enumerate(zip(res[1], res[0], want)):
        assert_close(a, r, 1e-4, f"fused sums, tensor {i}")
        assert_close(b, r, 1e-4, f"stand-alone passes, tensor {i}")


_ACTS = {"none": (L.ACT_NONE, lambda u: u), "relu": (L.ACT_RELU, F.relu), "leaky": (L.ACT_LEAKY, lambda u: F.leaky_relu(u, 0.1)),
         "elu": (L.ACT_ELU, F.elu)}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("c,groups,shape,act,res", [
    (8, 8, (6, 10, 12), "elu", False), (32, 8, (8, 8, 8), "elu", True), (4, 1, (5, 6, 7), "relu", False),
    (24, 8, (4, 6, 5), "leaky", True), (6, 3, (3, 4, 5), "none", False), (64, 8, (4, 4, 4), "elu", True),
    (16, 8, (7, 9, 11), "none", True), (256, 8, (2, 3, 4), "elu", False)])
def test_group_norm_act(mode, c, groups, shape, act, res):
    tag = f"gn{c}{groups}{act}{res}"
    n = 2
    x, r, cot = _prep(mode, rnd(tag + "x", n, c, *shape, scale=2.0) + 0.5, rnd(tag + "r", n, c, *shape), rnd(tag + "g", n, c, *shape))
    gamma, beta = rnd(tag + "ga", c) * 0.3 + 1.0, rnd(tag + "be", c) * 0.3
    code, fn = _ACTS[act]
    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = F.group_norm(xr, groups, gr, br, 1e-5)
    if res:
        u = u + rr
    zr = fn(u)
    (zr * cot).sum().backward()
    with mednet_hip.precision(mode):
        gn = hnn.GroupNorm(groups, c).to(DEV)
        with torch.no_grad():
            gn.weight.copy_(gamma)
            gn.bias.copy_(beta)
        dt = mednet_hip.config.act_dtype()
        xg = x.to(DEV).to(dt).requires_grad_(True)
        rg = r.to(DEV).to(dt).requires_grad_(True)
        z = gn(xg, act=code, residual=rg if res else None)
        (z.float() * cot.to(DEV)).sum().backward()
    tol = TOL[mode]
    assert_close(z, zr, tol, "z")
    assert_close(xg.grad, xr.grad, tol, "dx")
    assert_close(gn.weight.grad, gr.grad, tol, "dgamma")
    assert_close(gn.bias.grad, br.grad, tol, "dbeta")
    if res:
        assert_close(rg.grad, rr.grad, tol, "dres")


@pytest.mark.parametrize("c,groups,shape", [(32, 8, (9, 10, 12)), (64, 8, (6, 7, 9)), (256, 8, (3, 4, 5)), (8, 1, (5, 6, 7))])
def test_group_norm_backward_in_two_launches_equals_four(c, groups, shape):
    """GroupNorm backward with the reduction of the partial rows, the per-group coefficients and the parameter gradients in two
    launches (gn_bwd_reduce_finalize_kernel + the apply kernel's first workgroup; default) against the four-launch form
    (option gn_bwd_one_launch=0): same gradients up to the summation order of the rows."""
    tag = f"gn2l{c}{groups}"
    x, cot = rnd(tag + "x", 2, c, *shape, scale=2.0) + 0.5, rnd(tag + "g", 2, c, *shape)
    res = {}
    try:
        for one in (1, 0):
            _set_option("gn_bwd_one_launch", one)
            with mednet_hip.precision("fp32"):
                gn = hnn.GroupNorm(groups, c).to(DEV)
                with torch.no_grad():
                    gn.weight.copy_(rnd(tag + "ga", c) * 0.3 + 1.0)
                    gn.bias.copy_(rnd(tag + "be", c) * 0.3)
                xg = x.to(DEV).requires_grad_(True)
                (gn(xg, act=L.ACT_ELU) * cot.to(DEV)).sum().backward()
                res[one] = (xg.grad.cpu(), gn.weight.grad.cpu(), gn.bias.grad.cpu())
    finally:
        _set_option("gn_bwd_one_launch", 1)
    for a, b, name in zip(res[1], res[0], ("dx", "dgamma", "dbeta")):
        assert_close(a, b, 2e-6, name)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("act", ["relu", "leaky", "elu"])
def test_standalone_activation(mode, act):
    x, cot = _prep(mode, rnd("act" + act, 2, 5, 3, 7, 9), rnd("actg" + act, 2, 5, 3, 7, 9))
    code, fn = _ACTS[act]
    xr = x.clone().requires_grad_(True)
    zr = fn(xr)
    (zr * cot).sum().backward()
    with mednet_hip.precision(mode):
        xg = x.to(DEV).to(mednet_hip.config.act_dtype()).requires_grad_(True)
        z = ops.activation(xg, code)
        (z.float() * cot.to(DEV)).sum().backward()
    assert_close(z, zr, TOL[mode], "z")
    assert_close(xg.grad, xr.grad, TOL[mode], "dx")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("c,shape,kind", [(8, (8, 12, 10), "max"), (32, (4, 4, 4), "max"), (8, (7, 9, 11), "max"),
                                          (3, (6, 5, 4), "max"), (16, (6, 6, 6), "avg"), (5, (5, 7, 6), "avg")])
def test_pool2(mode, c, shape, kind):
    n = 2
    (x,) = _prep(mode, rnd(f"pool{c}{shape}", n, c, *shape))
    oshape = tuple(s // 2 for s in shape)
    (cot,) = _prep(mode, rnd(f"poolg{c}{shape}", n, c, *oshape))
    xr = x.clone().requires_grad_(True)
    yr = (F.max_pool3d if kind == "max" else F.avg_pool3d)(xr, 2)
    (yr * cot).sum().backward()
    with mednet_hip.precision(mode):
        xg = x.to(DEV).to(mednet_hip.config.act_dtype()).requires_grad_(True)
        y = ops.pool2(xg, L.POOL_MAX if kind == "max" else L.POOL_AVG)
        (y.float() * cot.to(DEV)).sum().backward()
    tol = 1e-6 if (mode == "fp32" or kind == "max") else TOL[mode]
    assert_close(y, yr, tol, "y")
    assert_close(xg.grad, xr.grad, max(tol, 1e-6) if kind == "max" else TOL[mode], "dx")


def test_max_pool_ties_route_to_first_maximum():
    x = torch.zeros(1, 8, 4, 4, 4)
    x[0, :, 0, 1, 1] = 1.0
    x[0, :, 1, 0, 0] = 1.0  # two equal maxima inside the first window; ReLU outputs produce many such ties
    cot = torch.ones(1, 8, 2, 2, 2)
    xr = x.clone().requires_grad_(True)
    F.max_pool3d(xr, 2).backward(cot)
    xg = x.to(DEV).requires_grad_(True)
    with mednet_hip.precision("fp32"):
        ops.pool2(xg, L.POOL_MAX).backward(cot.to(DEV))
    assert torch.equal(xg.grad.cpu(), xr.grad)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("ce,cx,eshape,xshape", [(8, 16, (8, 12, 10), (4, 6, 5)), (8, 16, (7, 9, 10), (3, 4, 5)),
                                                 (3, 5, (5, 5, 5), (2, 2, 2))])
def test_upsample_concat(mode, ce, cx, eshape, xshape):
    e, x = _prep(mode, rnd(f"uce{ce}{eshape}", 2, ce, *eshape), rnd(f"ucx{cx}{xshape}", 2, cx, *xshape))
    (cot,) = _prep(mode, rnd(f"ucg{ce}{cx}{eshape}", 2, ce + cx, *eshape))
    er, xr = e.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr = torch.cat((er, F.interpolate(xr, size=eshape, mode="nearest")), dim=1)
    (yr * cot).sum().backward()
    with mednet_hip.precision(mode):
        dt = mednet_hip.config.act_dtype()
        eg, xg = e.to(DEV).to(dt).requires_grad_(True), x.to(DEV).to(dt).requires_grad_(True)
        y = ops.upsample_concat(eg, xg)
        (y.float() * cot.to(DEV)).sum().backward()
    assert_close(y, yr, 1e-6, "y")
    assert_close(eg.grad, er.grad, 1e-6, "denc")
    assert_close(xg.grad, xr.grad, 1e-6 if mode == "fp32" else 1e-2, "dx")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (32, 64), (16, 48)])
def test_matrix_core_conv_family_in_both_16bit_storage_types(mode, cin, cout):
    """The 16-bit matrix-core kernels (conv forward / data gradient / weight gradient, ConvTranspose3d forward with bias /
    data gradient / weight gradient) with activations ALREADY in the mode's storage type, so the MFMA paths run (an fp32
    input takes the direct kernels): conv_mfma.hip is compiled once per element type (bf16: v_mfma_f32_32x32x16_bf16, fp16:
    v_mfma_f32_32x32x16_f16, BASELINE config 5)."""
    tol = TOL[mode]
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    x = half_round(rnd(f"mc{cin}{cout}x", 1, cin, 6, 10, 16), mode)
    w = rnd(f"mc{cin}{cout}w", cout, cin, 3, 3, 3, scale=0.2)
    g = rnd(f"mc{cin}{cout}g", 1, cout, 6, 10, 16)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    (F.conv3d(xr, wr, None, padding=1) * g).sum().backward()
    with mednet_hip.precision(mode):
        cv = hnn.Conv3d(cin, cout, 3, bias=False).to(DEV)
        with torch.no_grad():
            cv.weight.copy_(w)
        xg = x.to(DEV).to(dt).requires_grad_(True)
        y = cv(xg)
        assert y.dtype == dt
        (y.float() * g.to(DEV)).sum().backward()
    assert_close(y, F.conv3d(x, w, None, padding=1), tol, "conv y")
    assert_close(xg.grad, xr.grad, tol, "conv dx")
    assert_close(cv.weight.grad, wr.grad, tol, "conv dw")
    if cin % 32 or cout % 32:
        return  # (the ConvTranspose3d matrix-core kernels take multiples of 32)
    xt = half_round(rnd(f"mt{cin}{cout}x", 1, cin, 3, 5, 8), mode)
    wt = rnd(f"mt{cin}{cout}w", cin, cout, 3, 3, 3, scale=0.2)
    b = rnd(f"mt{cin}{cout}b", cout)
    gt = rnd(f"mt{cin}{cout}g", 1, cout, 6, 10, 16)
    xr, wr, br = xt.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2, padding=1, output_padding=1)
    (yr * gt).sum().backward()
    with mednet_hip.precision(mode):
        up = hnn.ConvTranspose3d(cin, cout).to(DEV)
        with torch.no_grad():
            up.weight.copy_(wt)
            up.bias.copy_(b)
        xg = xt.to(DEV).to(dt).requires_grad_(True)
        y = up(xg)
        (y.float() * gt.to(DEV)).sum().backward()
    assert_close(y, yr, tol, "convT y")
    assert_close(xg.grad, xr.grad, tol, "convT dx")
    assert_close(up.weight.grad, wr.grad, tol, "convT dw")
    assert_close(up.bias.grad, br.grad, 5e-3 if mode == "bf16" else 1e-3, "convT db")


# ----------------------------------------------------------------------------------------------------- losses
@pytest.mark.parametrize("kw", [dict(), dict(weight=[0.05, 1, 1, 1]), dict(weight=[0.05, 1, 1, 1], sigmoid_normalization=True),
                                dict(weight=[0.05, 1, 1, 1], ignore_index=1), dict(ignore_index=0), dict(epsilon=1e-2)])
def test_dice_loss_matches_golden_and_oracle(kw, golden_dir):
    import os
    rec = np.load(os.path.join(golden_dir, "losses.npz"))
    z, y = torch.from_numpy(rec["logits"]), torch.from_numpy(rec["labels"])
    kwt = {k: (torch.tensor(v) if k == "weight" else v) for k, v in kw.items()}
    zr = z.clone().requires_grad_(True)
    lr = O.DiceLoss(**kwt)(zr, y)
    lr.backward()
    zg = z.to(DEV).requires_grad_(True)
    kwg = {k: (torch.tensor(v).to(DEV) if k == "weight" else v) for k, v in kw.items()}
    lg = HL.DiceLoss(**kwg).to(DEV)(zg, y.to(DEV))
    lg.backward()
    assert abs(float(lg) - float(lr)) <= 2e-6
    assert_close(zg.grad, zr.grad, 1e-5, "dlogits")
    tags = {(): "dice_plain", ("weight",): "dice_weight", ("sigmoid_normalization", "weight"): "dice_sigmoid",
            ("ignore_index", "weight"): "dice_ignore", ("epsilon",): "dice_eps"}
    tag = tags.get(tuple(sorted(kw)))
    if tag and not (tag == "dice_ignore" and kw.get("ignore_index") != 1):
        assert abs(float(lg) - float(rec[tag + ".value"])) <= 2e-6      # golden = the reference's own output
        assert_close(zg.grad, torch.from_numpy(rec[tag + ".grad"]), 1e-5, "dlogits vs golden")


def test_dice_on_channel_slice_and_large_batch():
    z = rnd("dslice", 3, 7, 9, 10, 11, scale=2.0)
    y = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).integers(0, 2, size=(3, 9, 10, 11)))
    w = torch.tensor([0.05, 1.0])
    zr = z.clone().requires_grad_(True)
    lr = O.DiceLoss(weight=w)(zr[:, 5:], y)
    lr.backward()
    zg = z.to(DEV).requires_grad_(True)
    lg = HL.DiceLoss(weight=w.to(DEV)).to(DEV)(zg[:, 5:], y.to(DEV))
    lg.backward()
    assert abs(float(lg) - float(lr)) <= 2e-6
    assert_close(zg.grad, zr.grad, 1e-5, "dlogits")
    assert torch.count_nonzero(zg.grad[:, :5]) == 0


def test_dice_metric_and_shape_assert(golden_dir):
    import os
    rec = np.load(os.path.join(golden_dir, "losses.npz"))
    z, y = torch.from_numpy(rec["logits"]), torch.from_numpy(rec["labels"])
    d = HL.dice_metric(z.to(DEV), y.to(DEV))
    np.testing.assert_allclose(d.cpu().numpy(), rec["dice_metric.value"], rtol=2e-5)
    with pytest.raises(AssertionError, match="same shape"):
        HL.DiceLoss().to(DEV)(z.to(DEV), y[:, :-1].to(DEV))
    with pytest.raises(AssertionError, match="same shape"):
        HL.DiceLoss(skip_last_target=True).to(DEV)(z.to(DEV), y.to(DEV))


@pytest.mark.parametrize("weight", [None, [0.05, 1, 1, 1]])
def test_cross_entropy(weight, golden_dir):
    import os
    rec = np.load(os.path.join(golden_dir, "losses.npz"))
    z, y = torch.from_numpy(rec["logits"]), torch.from_numpy(rec["labels"])
    w = None if weight is None else torch.tensor(weight)
    tag = "ce_plain" if weight is None else "ce_weight"
    zg = z.to(DEV).requires_grad_(True)
    lg = HL.CrossEntropyLoss(weight=None if w is None else w.to(DEV)).to(DEV)(zg, y.to(DEV))
    lg.backward()
    assert abs(float(lg) - float(rec[tag + ".value"])) <= 2e-6
    assert_close(zg.grad, torch.from_numpy(rec[tag + ".grad"]), 1e-5, "dlogits")


def test_out_of_range_labels_poison_the_loss():
    """The reference raises on a label outside [0, C) (scatter_ in expand_as_one_hot, loss.py:81-86; nll_loss for CE).  A
    kernel cannot raise without a host sync per step, so the fused losses turn NaN instead (loss and every gradient):
    corrupt label volumes cannot train silently.  ignore_index stays legal for cross-entropy."""
    z = rnd("oorz", 2, 4, 6, 7, 8).to(DEV).requires_grad_(True)
    y = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).integers(0, 4, size=(2, 6, 7, 8))).to(DEV)
    ok = HL.DiceLoss().to(DEV)(z, y)
    assert torch.isfinite(ok)
    for badval in (4, -1, 255):
        yb = y.clone()
        yb[1, 2, 3, 4] = badval
        zb = z.detach().clone().requires_grad_(True)
        lb = HL.DiceLoss().to(DEV)(zb, yb)
        lb.backward()
        assert torch.isnan(lb) and torch.isnan(zb.grad).any(), badval
        assert torch.isnan(HL.CrossEntropyLoss().to(DEV)(z, yb)), badval
        assert torch.isnan(HL.dice_metric(z, yb)).any()
    yi = y.clone()
    yi[0, 0, 0, 0] = -100  # nn.CrossEntropyLoss's default ignore_index
    assert torch.isfinite(HL.CrossEntropyLoss().to(DEV)(z, yi))


@pytest.mark.parametrize("kind", ["L2", "L1"])
def test_landmark_loss_composition(kind, golden_dir):
    """LandmarkNet.loss (landmarks.py:125-134): Dice on the class slice + weighted per-channel regression."""
    import os
    rec = np.load(os.path.join(golden_dir, "losses.npz"))
    out = torch.from_numpy(rec["logits5"]).to(DEV).requires_grad_(True)
    hm = torch.from_numpy(rec["heatmaps"]).to(DEV)
    lab = torch.from_numpy(rec["labels2"]).to(DEV)
    regw = [0.001, 0.015, 0.02]
    cl = HL.DiceLoss(weight=torch.tensor([0.05, 1.0]).to(DEV)).to(DEV)(out[:, 3:], lab)
    rg = HL.HeatmapRegressionLoss(regw, kind).to(DEV)(out[:, :3], hm)
    (cl + rg).backward()
    tag = "ldmk_l2" if kind == "L2" else "ldmk_l1"
    assert abs(float(cl) - float(rec[tag + ".class"])) <= 2e-6
    assert abs(float(rg) - float(rec[tag + ".reg"])) <= 2e-5 * abs(float(rec[tag + ".reg"]))
    assert_close(out.grad, torch.from_numpy(rec[tag + ".grad"]), 1e-5, "dout")
    # uint8 heat-map targets (what MedDataset emits, dataset.py:327) give the same numbers without the .float() copy
    out2 = torch.from_numpy(rec["logits5"]).to(DEV).requires_grad_(True)
    rg2 = HL.HeatmapRegressionLoss(regw, kind).to(DEV)(out2[:, :3], hm.to(torch.uint8))
    assert abs(float(rg2) - float(rg)) <= 1e-6 * abs(float(rg))


@pytest.mark.parametrize("cls_loss", ["dice", "ce"])
def test_channel_split_losses_write_one_gradient_buffer(cls_loss):
    """LandmarkNet.training_step slices the network output into heat-map and class channels (landmarks.py:71-72).  The two
    loss backward kernels write their slices of ONE full-size gradient (no concatenation): the gradient of the unsplit logits
    must equal autograd's over plain slicing, bit for bit, and really be that shared buffer."""
    from mednet_hip import ops as hops
    n, nh, nc, sp = 2, 5, 3, (6, 10, 12)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(n, nh + nc, *sp, generator=g)
    hm = torch.randint(0, 256, (n, nh, *sp), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, nc, (n, *sp), generator=g)
    wts = [0.3, 1.0, 0.7, 0.2, 1.5]
    cw = torch.tensor([0.1, 1.0, 2.0])

    def run(split):
        x = logits.to(DEV).requires_grad_(True)
        a, b = split(x)
        reg = hops.heatmap_loss(a, hm.to(DEV), wts, "L2")
        cl = hops.dice_loss(b, lab.to(DEV), cw.to(DEV)) if cls_loss == "dice" else hops.cross_entropy(b, lab.to(DEV), cw.to(DEV))
        (reg + 2.0 * cl).backward()
        return x.grad

    g_fused = run(lambda x: hops.split_channels(x, nh))
    g_plain = run(lambda x: (x[:, :nh], x[:, nh:]))
    assert torch.equal(g_fused, g_plain)
    assert g_fused.is_contiguous()


def test_channel_split_with_two_losses_on_one_slice():
    """Dice AND cross-entropy on the class slice (and the regression loss applied twice on the heat-map slice): only the
    first backward of a slice may write the shared buffer; the others get tensors of their own and autograd sums them.
    Must equal plain slicing."""
    from mednet_hip import ops as hops
    n, nh, nc, sp = 2, 3, 3, (4, 6, 8)
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(n, nh + nc, *sp, generator=g)
    hm = torch.randint(0, 256, (n, nh, *sp), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, nc, (n, *sp), generator=g)
    cw = torch.tensor([0.1, 1.0, 2.0])

    def run(split):
        x = logits.to(DEV).requires_grad_(True)
        a, b = split(x)
        loss = hops.dice_loss(b, lab.to(DEV), cw.to(DEV)) + 3.0 * hops.cross_entropy(b, lab.to(DEV), cw.to(DEV)) \
            + hops.heatmap_loss(a, hm.to(DEV), [0.3, 1.0, 0.7], "L2") + 0.5 * hops.heatmap_loss(a, hm.to(DEV), [1.0, 0.2, 0.4], "L1")
        loss.backward()
        return x.grad

    g_fused = run(lambda x: hops.split_channels(x, nh))
    g_plain = run(lambda x: (x[:, :nh], x[:, nh:]))
    assert_close(g_fused, g_plain, 1e-6, "two losses per slice")
    # and against ATen on the CPU
    x = logits.clone().requires_grad_(True)
    a, b = x[:, :nh], x[:, nh:]
    p = torch.softmax(b, 1)
    oh = torch.nn.functional.one_hot(lab, nc).permute(0, 4, 1, 2, 3).float()
    inter = (p * oh).transpose(0, 1).flatten(1).sum(-1) * cw
    den = (p + oh).transpose(0, 1).flatten(1).sum(-1)
    dice = torch.mean(1 - 2 * inter / den.clamp(min=1e-5))
    ce = torch.nn.functional.cross_entropy(b, lab, weight=cw)
    l2 = sum(w * torch.nn.functional.mse_loss(a[:, c], hm[:, c].float()) for c, w in enumerate([0.3, 1.0, 0.7]))
    l1 = sum(w * torch.nn.functional.l1_loss(a[:, c], hm[:, c].float()) for c, w in enumerate([1.0, 0.2, 0.4]))
    (dice + 3.0 * ce + l2 + 0.5 * l1).backward()
    assert_close(g_fused, x.grad, 1e-5, "two losses per slice vs ATen")


def test_adam_step_matches_torch():
    g = np.random.Generator(np.random.PCG64(3))
    p0 = torch.from_numpy(g.standard_normal(10007).astype(np.float32))
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        grad = torch.from_numpy(g.standard_normal(10007).astype(np.float32))
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step_(p, grad.to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
    assert_close(p, ref, 1e-6, "params after 3 Adam steps")


# ----------------------------------------------------------------------------------------------------- MFMA kernels
def _conv_case(n, cin, cout, shape, tag):
    x = bf16_round(rnd(tag + "x", n, cin, *shape))
    w = bf16_round(rnd(tag + "w", cout, cin, 3, 3, 3, scale=0.08))
    cot = bf16_round(rnd(tag + "g", n, cout, *shape))
    return x, w, cot


@pytest.mark.parametrize("n,cin,cout,shape", [
    (1, 32, 32, (4, 8, 16)),      # exactly one brick
    (2, 32, 32, (9, 11, 21)),     # ragged bricks in every dimension, two samples
    (1, 64, 32, (8, 16, 16)),     # 4 K-chunks
    (1, 32, 64, (6, 8, 32)),      # two output-channel blocks
    (1, 128, 128, (5, 6, 7)),     # volume smaller than a brick
    (1, 256, 256, (4, 4, 4)),
    (2, 64, 64, (16, 16, 16)),
    (2, 16, 32, (9, 11, 21)),     # 16-channel sides (UNet3D's first DoubleConv): one k-step / half a channel block
    (1, 32, 16, (8, 8, 16)),
    (1, 16, 16, (5, 9, 17)),
    (1, 48, 80, (4, 8, 16)),      # any multiple of 16: the last channel block is half full
    (2, 64, 96, (10, 10, 6)),     # config 5's deepest level: 8-wide bricks (FwdTile<3>, wgrad_mfma2_kernel<8>), 39 % instead of 10 % filled
    (1, 32, 64, (12, 20, 24)),    # 24 columns: three 8-wide bricks instead of two 16-wide ones
    (2, 32, 32, (7, 9, 40)),      # 40 columns: 16 + 16 + 8
])
def test_conv3d_mfma_fwd_dgrad_wgrad(n, cin, cout, shape):
    """The bf16 MFMA kernels (forced) against the fp32 oracle AND against the direct kernels on identical bf16 inputs
    (weights pre-rounded to bf16, so the only differences are accumulation order and the bf16 rounding of outputs)."""
    tag = f"mfma{n}{cin}{cout}{shape}"
    x, w, cot = _conv_case(n, cin, cout, shape, tag)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, None, padding=1)
    (yr * cot).sum().backward()
    res = {}
    for algo in ("mfma", "direct"):
        mednet_hip.set_conv_algo(algo)
        try:
            with mednet_hip.precision("bf16"):
                conv = hnn.Conv3d(cin, cout, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                xg = x.to(DEV).bfloat16().requires_grad_(True)
                y = conv(xg)
                y.backward(cot.to(DEV).bfloat16())
                res[algo] = (y.detach().float().cpu(), xg.grad.float().cpu(), conv.weight.grad.cpu())
        finally:
            mednet_hip.set_conv_algo("auto")
    y, dx, dw = res["mfma"]
    assert_close(y, yr, 6e-3, "y vs oracle")          # bf16 output rounding only (2^-9 relative per element)
    assert_close(dx, xr.grad, 6e-3, "dx vs oracle")
    assert_close(dw, wr.grad, 2e-4, "dw vs oracle")   # fp32 output, fp32 accumulation
    yd, dxd, dwd = res["direct"]
    assert_close(y, yd, 5e-3, "y vs direct kernel")
    assert_close(dx, dxd, 5e-3, "dx vs direct kernel")
    assert_close(dw, dwd, 2e-4, "dw vs direct kernel")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,shape", [(2, 64, 64, (10, 10, 6)), (1, 32, 64, (12, 20, 24)), (2, 32, 32, (7, 9, 40)), (1, 128, 64, (5, 6, 7))])
def test_narrow_bricks_change_no_bit_of_the_convolution(mode, n, cin, cout, shape):
    """8-wide bricks (FwdTile<3>, option conv_narrow; round 5) only re-partition the output voxels: every output element is the same
    chain of MFMA accumulations (K chunk outer, tap inner) as with 16-wide bricks, so forward and data gradient are bit-identical;
    the weight gradient (wgrad_mfma2_kernel<8>: other partial slabs) agrees to fp32 summation order."""
    lib = L.lib()
    tag = f"narrow{n}{cin}{cout}{shape}"
    x, w, cot = _conv_case(n, cin, cout, shape, tag)
    res = {}
    try:
        for narrow in (2, 0):  # (2: 8-wide bricks wherever they need fewer voxel slots; the default 1 takes them up to 8 columns only)
            lib.mednet_set_option(b"conv_narrow", narrow)
            mednet_hip.set_conv_algo("mfma")
            with mednet_hip.precision(mode):
                conv = hnn.Conv3d(cin, cout, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                xg = x.to(DEV).to(mednet_hip.config.act_dtype()).requires_grad_(True)
                y = conv(xg)
                y.backward(cot.to(DEV).to(y.dtype))
                res[narrow] = (y.detach().clone(), xg.grad.clone(), conv.weight.grad.clone())
    finally:
        lib.mednet_set_option(b"conv_narrow", 1)
        mednet_hip.set_conv_algo("auto")
    assert torch.equal(res[2][0], res[0][0]) and torch.equal(res[2][1], res[0][1])
    assert_close(res[2][2], res[0][2], 2e-5, "dw narrow vs wide bricks")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,shape,wgs", [
    (1, 32, 32, (8, 8, 16), 0),        # one column, one slab
    (2, 32, 32, (9, 11, 21), 0),       # ragged in every dimension, two samples
    (1, 32, 32, (40, 16, 32), 0),      # two z-slabs (20 planes each)
    (2, 64, 64, (16, 16, 32), 0),      # four channel-block pairs
    (1, 16, 48, (12, 9, 17), 0),       # 16-channel sides: half-filled channel blocks
    (4, 32, 32, (32, 32, 64), 128),    # the trainer's side-stream plan
    (2, 64, 32, (24, 24, 48), 24),     # a workgroup count that does not divide the items (and not a multiple of 8)
    (1, 128, 96, (8, 8, 16), 0),       # more pairs than items per pair: one workgroup per pair
])
def test_co_resident_weight_gradient_kernel(mode, n, cin, cout, shape, wgs):
    """wgrad_mfma4_kernel (round 5: one wave per SIMD, z-columns through an LDS-DMA ring, option wgrad_v4) against ATen in fp32 on
    the same 16-bit inputs and against wgrad_mfma2_kernel (same products, another summation order; wgrad_v4=0 selects it)."""
    from mednet_hip import _lib as L
    lib = L.lib()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    code = {"bf16": L.BF16, "fp16": L.F16}[mode]
    tag = f"wg4{n}{cin}{cout}{shape}"
    x = rnd(tag + "x", n, cin, *shape).to(dt)
    dy = rnd(tag + "g", n, cout, *shape).to(dt)
    xr = x.float()
    wr = torch.zeros(cout, cin, 3, 3, 3, requires_grad=True)
    (F.conv3d(xr, wr, None, padding=1) * dy.float()).sum().backward()
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last_3d)
    dyg = dy.to(DEV).contiguous(memory_format=torch.channels_last_3d)
    d, h, w = shape
    out = {}
    try:
        for v4 in (1, 0):
            lib.mednet_set_option(b"wgrad_v4", v4)
            ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, 3, wgs), dtype=torch.uint8, device=DEV)
            ws.fill_(0xFF)  # (NaN pattern: a slab element nobody wrote would show)
            dw = torch.full((cout, cin, 3, 3, 3), float("nan"), device=DEV)
            L.check(lib.mednet_conv3d_wgrad(xg.data_ptr(), dyg.data_ptr(), dw.data_ptr(), None, n, d, h, w, cin, cout, 3, code, L.NDHWC,
                                            code, L.NDHWC, L.ALGO_MFMA, wgs, ws.data_ptr(), ws.numel(), L.stream()), "conv3d_wgrad")
            torch.cuda.synchronize()
            out[v4] = dw.cpu()
    finally:
        lib.mednet_set_option(b"wgrad_v4", 1)  # (the default)
    assert_close(out[1], wr.grad, 1e-4, "dw (co-resident kernel) vs ATen fp32")
    assert_close(out[1], out[0], 2e-5, "dw (co-resident kernel) vs wgrad_mfma2_kernel")


def _fuzz_cases_ct(k, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 3)), int(rng.choice([32, 64, 96])), int(rng.choice([32, 64])),
             tuple(int(v) for v in rng.integers(1, 21, 3))) for _ in range(k)]


@pytest.mark.parametrize("n,cin,cout,shape", [(1, 64, 32, (2, 4, 16)), (2, 64, 32, (3, 5, 9)), (1, 256, 128, (4, 4, 4)),
                                              (1, 32, 32, (5, 9, 17))] + _fuzz_cases_ct(10, 77))
def test_conv_transpose_mfma_dgrad_wgrad(n, cin, cout, shape):
    tag = f"ctm{n}{cin}{cout}{shape}"
    oshape = tuple(2 * s for s in shape)
    x = bf16_round(rnd(tag + "x", n, cin, *shape))
    w = bf16_round(rnd(tag + "w", cin, cout, 3, 3, 3, scale=0.08))
    cot = bf16_round(rnd(tag + "g", n, cout, *oshape))
    b = rnd(tag + "b", cout)
    sk = bf16_round(rnd(tag + "s", n, cout, *oshape))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, b, stride=2, padding=1, output_padding=1) + sk
    (yr * cot).sum().backward()
    mednet_hip.set_conv_algo("mfma")
    try:
        with mednet_hip.precision("bf16"):
            up = hnn.ConvTranspose3d(cin, cout).to(DEV)
            with torch.no_grad():
                up.weight.copy_(w)
                up.bias.copy_(b)
            xg = x.to(DEV).bfloat16().requires_grad_(True)
            y = up(xg, skip=sk.to(DEV).bfloat16())
            y.backward(cot.to(DEV).bfloat16())
    finally:
        mednet_hip.set_conv_algo("auto")
    assert_close(y, yr, 6e-3, "y (MFMA, 8 parity classes, bias + skip epilogue)")
    assert_close(xg.grad, xr.grad, 6e-3, "dx (MFMA stride-2 gather)")
    assert_close(up.weight.grad, wr.grad, 2e-4, "dw (MFMA, transposing LDS reads)")


@pytest.mark.parametrize("n,cin,cout,shape", [(3, 32, 32, (64, 64, 128)),    # accumulate mode, sample changes inside a workgroup's item list
                                              (2, 32, 64, (64, 64, 64)),     # accumulate mode with two channel blocks
                                              (2, 32, 32, (20, 24, 36))])    # ragged bricks: one row per wave and brick
def test_fused_groupnorm_partials_sum_to_the_output_statistics(n, cin, cout, shape):
    """The conv epilogue's GroupNorm partials (per channel PAIR; in the persistent kernel accumulated over a workgroup's bricks
    of a sample) must add up to sum(y) and sum(y^2) of the STORED bf16 output, per sample and channel pair."""
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(n, cin, *shape, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) * 0.05)
    mednet_hip.set_conv_algo("mfma")
    try:
        with mednet_hip.precision("bf16"):
            y, partial = ops.conv3d_with_stats(x, w, None, ops.pack_conv_weight(w, 3, False), 3)
    finally:
        mednet_hip.set_conv_algo("auto")
    assert partial is not None and partial.shape[0] == n and partial.shape[2] == cout
    tot = partial.double().sum(dim=1)                               # n x cout x 2, pair sums at even channels
    yd = y.double()
    s = yd.sum(dim=(2, 3, 4)).reshape(n, cout // 2, 2).sum(-1)
    q = (yd * yd).sum(dim=(2, 3, 4)).reshape(n, cout // 2, 2).sum(-1)
    assert torch.all(tot[:, 1::2] == 0)
    scale = q.sqrt() * (yd[0, 0].numel() ** 0.5)
    assert torch.all((tot[:, 0::2, 0] - s).abs() <= 2e-5 * scale + 1e-3)
    assert torch.all((tot[:, 0::2, 1] - q).abs() <= 2e-5 * q)


def _fuzz_cases(k, seed):
    rng = np.random.default_rng(seed)
    chans = [16, 32, 48, 64, 96]
    out = []
    for _ in range(k):
        out.append((int(rng.integers(1, 4)), int(rng.choice(chans)), int(rng.choice(chans)),
                    tuple(int(v) for v in rng.integers(1, 41, 3))))
    return out


@pytest.mark.parametrize("n,cin,cout,shape", _fuzz_cases(16, 2026))
def test_conv3d_mfma_random_shapes_against_direct_kernels(n, cin, cout, shape):
    """Random channel counts (multiples of 16) and volume sizes (1..40 per axis: thinner than a brick, ragged, many bricks):
    the matrix-core conv / data gradient / weight gradient / fused statistics against the direct kernels on the same bf16
    inputs.  Catches indexing slips of the persistent item walk, the padded channel blocks and the hardware zero fill."""
    tag = f"fz{n}{cin}{cout}{shape}"
    x, w, cot = _conv_case(n, cin, cout, shape, tag)
    res = {}
    for algo in ("mfma", "direct"):
        mednet_hip.set_conv_algo(algo)
        try:
            with mednet_hip.precision("bf16"):
                conv = hnn.Conv3d(cin, cout, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                xg = x.to(DEV).bfloat16().requires_grad_(True)
                y, partial = conv.forward_with_stats(xg)
                y.backward(cot.to(DEV).bfloat16())
                res[algo] = (y.detach().float().cpu(), xg.grad.float().cpu(), conv.weight.grad.cpu(), partial)
        finally:
            mednet_hip.set_conv_algo("auto")
    y, dx, dw, partial = res["mfma"]
    yd, dxd, dwd, _ = res["direct"]
    assert_close(y, yd, 5e-3, "y vs direct kernel")
    assert_close(dx, dxd, 5e-3, "dx vs direct kernel")
    assert_close(dw, dwd, 3e-4, "dw vs direct kernel")
    assert partial is not None
    tot = partial.double().sum(dim=1).cpu()
    s_ref = y.double().sum(dim=(2, 3, 4)).reshape(n, cout // 2, 2).sum(-1)
    q_ref = (y.double() ** 2).sum(dim=(2, 3, 4)).reshape(n, cout // 2, 2).sum(-1)
    nv = float(np.prod(shape))
    assert torch.all((tot[:, 0::2, 0] - s_ref).abs() <= 1e-4 * (q_ref * nv).sqrt() + 1e-3)
    assert torch.all((tot[:, 0::2, 1] - q_ref).abs() <= 1e-4 * q_ref + 1e-6)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,shape", [(1, (66, 60, 50)),    # 544 bricks (a multiple of 8: one slab per XCD), ragged in z, y and x
                                     (1, (76, 56, 64)),    # 532 bricks: interleaved brick order
                                     (3, (32, 64, 48)),    # 576 bricks, the sample changes inside a workgroup's brick list
                                     (2, (32, 60, 62))])   # 512 bricks, 2 z-layers of bricks per XCD: the x-z-y slab walk
def test_conv32_specialisation_is_bit_identical_to_the_general_kernel(mode, n, shape):
    """32 -> 32 channels on >= 512 bricks runs conv32_mfma_kernel (weights in registers, whole-row double-buffered bricks).
    It accumulates in the general kernel's order (K chunk outer, tap inner, one fp32 accumulator per output), so outputs and
    data gradients must be IDENTICAL bit for bit to the general kernel's (option conv32=0), and the fused GroupNorm partials
    must add up to the same totals (their row layout differs)."""
    tag = f"c32{n}{shape}"
    x, w, cot = _conv_case(n, 32, 32, shape, tag)
    x, w, cot = (half_round(t, mode) for t in (x, w, cot))
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    res = {}
    mednet_hip.set_conv_algo("mfma")
    try:
        for special in (1, 0):
            assert L.lib().mednet_set_option(b"conv32", special) == 0
            with mednet_hip.precision(mode):
                conv = hnn.Conv3d(32, 32, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                xg = x.to(DEV).to(dt).requires_grad_(True)
                y, partial = conv.forward_with_stats(xg)
                y.backward(cot.to(DEV).to(dt))
                res[special] = (y.detach().clone(), xg.grad.clone(), partial.double().sum(dim=1))
    finally:
        L.lib().mednet_set_option(b"conv32", 1)
        mednet_hip.set_conv_algo("auto")
    assert res[1][2].shape == res[0][2].shape
    assert torch.equal(res[1][0], res[0][0]), "y differs from the general kernel"
    assert torch.equal(res[1][1], res[0][1]), "dx differs from the general kernel"
    tot1, tot0 = res[1][2], res[0][2]
    nv = float(np.prod(shape))
    q = tot0[:, 0::2, 1]
    assert torch.all((tot1[:, 0::2, 0] - tot0[:, 0::2, 0]).abs() <= 2e-5 * (q * nv).sqrt() + 1e-3)
    assert torch.all((tot1[:, 0::2, 1] - q).abs() <= 2e-5 * q)
    assert torch.all(tot1[:, 1::2] == 0)
    # ... and the general kernel is itself checked against the oracle; one direct comparison here as well
    yr = F.conv3d(x, w, None, padding=1)
    assert_close(res[1][0], yr, 6e-3 if mode == "bf16" else 1e-3, "y vs oracle")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,shape", [(2, (32, 60, 62)), (1, (76, 56, 64))])
def test_conv32_epilogue_variants_match_the_general_kernel(mode, n, shape):
    """The specialised kernel's epilogue comes in compile-time variants (activation, summed second gradient, fused GroupNorm
    statistics, fused first pass of a GroupNorm backward, with ragged rows / planes / x ends): each through its C-ABI entry
    point against the general kernel (option conv32=0) -- tensors bit for bit, partial sums up to fp32 summation order."""
    lib = L.lib()
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    dcode = L.dt(torch.empty(0, dtype=dt))
    CL = torch.channels_last_3d
    g = torch.Generator(device=DEV).manual_seed(11)
    def rand(*sh, scale=1.0):
        return (torch.randn(*sh, device=DEV, generator=g) * scale).to(dt).contiguous(memory_format=CL)
    x, add, gy = rand(n, 32, *shape), rand(n, 32, *shape), rand(n, 32, *shape)
    w = torch.randn(32, 32, 3, 3, 3, device=DEV, generator=g) * 0.05
    coef = torch.randn(n, 32, 2, device=DEV, generator=g).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ALGO_MFMA = 2
    with mednet_hip.precision(mode):
        pk = ops.pack_conv_weight(w, 3, False)
    d, h, wd = shape

    def run(special):
        assert lib.mednet_set_option(b"conv32", special) == 0
        out = {}
        for act in (1, 3):   # ReLU, ELU; with and without the statistics
            for stats in (False, True):
                y = torch.empty_like(x)
                rows = lib.mednet_conv3d_fused_stats_chunks(n, d, h, wd, 32, 32, 3, dcode, dcode, ALGO_MFMA)
                part = torch.zeros(n, rows, 32, 2, device=DEV) if stats else None
                L.check(lib.mednet_conv3d_act_fwd(x.data_ptr(), pk.data_ptr(), y.data_ptr(), n, d, h, wd, 32, 32, act, ALGO_MFMA,
                                                  part.data_ptr() if stats else None, dcode, st), "act_fwd")
                out[f"act{act}{stats}"] = (y, part.double().sum(1) if stats else None)
        dx = torch.empty_like(x)
        L.check(lib.mednet_conv3d_dgrad_add(x.data_ptr(), pk.data_ptr(), add.data_ptr(), dx.data_ptr(), n, d, h, wd, 32, 32,
                                            ALGO_MFMA, dcode, st), "dgrad_add")
        out["dgrad_add"] = (dx, None)
        rows = lib.mednet_conv3d_dgrad_gn_rows(n, d, h, wd, 32, 32, ALGO_MFMA)
        assert rows > 0
        for gact in (0, 1, 2, 3):
            for with_add in (False, True):
                dx = torch.empty_like(x)
                part = torch.zeros(n, rows, 32, 2, device=DEV)
                L.check(lib.mednet_conv3d_dgrad_gn(x.data_ptr(), pk.data_ptr(), add.data_ptr() if with_add else None, dx.data_ptr(),
                                                   gy.data_ptr(), coef.data_ptr(), gact, part.data_ptr(), n, d, h, wd, 32, 32,
                                                   ALGO_MFMA, dcode, st), "dgrad_gn")
                out[f"gn{gact}{with_add}"] = (dx, part.double().sum(1))
        torch.cuda.synchronize()
        return out

    try:
        a, b = run(1), run(0)
    finally:
        lib.mednet_set_option(b"conv32", 1)
    nv = float(np.prod(shape))
    for k in a:
        assert torch.equal(a[k][0], b[k][0]), f"{k}: tensor differs from the general kernel"
        if a[k][1] is not None:
            ta, tb = a[k][1], b[k][1]
            assert ta.shape == tb.shape
            scale = tb.abs().amax(dim=(1,), keepdim=True) + 1e-3
            assert torch.all((ta - tb).abs() <= 5e-5 * scale + 2e-5 * tb.abs() + 1e-3 * nv ** 0.5 * 1e-2), f"{k}: partial sums differ"


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,shape", [
    (1, 64, 64, (32, 64, 64)),    # 256 bricks x 1 block pair: one item per workgroup, accumulate mode
    (1, 64, 64, (30, 60, 50)),    # the same grid, ragged in z, y and x
    (2, 32, 128, (24, 40, 48)),   # 184 brick slots (180 real: padding items) x 2 pairs, two rounds, the sample changes
    (1, 128, 128, (36, 40, 48)),  # 135 bricks (not a multiple of 8): one statistics row per wave and brick; 8 K chunks
    (3, 48, 64, (32, 32, 64)),    # 3 K chunks, 384 bricks over three samples: 1.5 rounds, sample changes inside a workgroup's list
    (1, 16, 64, (32, 64, 64)),    # ONE K chunk: every chunk is first and last
])
def test_two_block_kernel_is_bit_identical_to_the_general_kernel(mode, n, cin, cout, shape):
    """Layers with >= 64 output channels run conv2b_mfma_kernel (round 6: two channel blocks per wave, one wave per SIMD, inputs and a
    ring of weight slots by LDS-DMA).  It accumulates in the general kernel's order (K chunk outer, tap inner), so every tensor it
    writes -- forward with / without activation and statistics, plain data gradient, data gradient with the summed second gradient
    and / or the first pass of a GroupNorm backward -- must equal conv_mfma_kernel<1>'s (option conv2b=0) BIT FOR BIT through the
    same C-ABI entry points; the fused partial sums agree up to fp32 summation order (their row layout differs)."""
    lib = L.lib()
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    dcode = L.dt(torch.empty(0, dtype=dt))
    CL = torch.channels_last_3d
    g = torch.Generator(device=DEV).manual_seed(23)

    def rand(c, scale=1.0):
        return (torch.randn(n, c, *shape, device=DEV, generator=g) * scale).to(dt).contiguous(memory_format=CL)
    x, dy, add, gy = rand(cin), rand(cout), rand(cin), rand(cin)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=g) * 0.05
    coef = torch.randn(n, cin, 2, device=DEV, generator=g).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ALGO_MFMA = 2
    with mednet_hip.precision(mode):
        pk = ops.pack_conv_weight(w, 3, False)
    d, h, wd = shape
    plan = (ctypes.c_int * 13)()

    def run(two_block):
        assert lib.mednet_set_option(b"conv2b", two_block) == 0
        out = {}
        if two_block:
            assert lib.mednet_conv3d_stats_plan(n, d, h, wd, cin, cout, dcode, 0, 1, plan) == 0 and plan[0] in (5, 6), list(plan)
        for act in (0, 3):   # none, ELU; with and without the statistics
            for stats in (False, True):
                y = torch.empty(n, cout, *shape, device=DEV, dtype=dt).contiguous(memory_format=CL)
                rows = lib.mednet_conv3d_fused_stats_chunks(n, d, h, wd, cin, cout, 3, dcode, dcode, ALGO_MFMA)
                assert rows > 0
                part = torch.full((n, rows, cout, 2), float("nan"), device=DEV) if stats else None
                L.check(lib.mednet_conv3d_act_fwd(x.data_ptr(), pk.data_ptr(), y.data_ptr(), n, d, h, wd, cin, cout, act, ALGO_MFMA,
                                                  part.data_ptr() if stats else None, dcode, st), "act_fwd")
                out[f"act{act}{stats}"] = (y, part.double().sum(1) if stats else None)
        if cin % 64 == 0:  # data gradients write `cin` channels: the two-block kernel takes them when those come in pairs of blocks
            # (dy has cout channels; the packed image's second section is the transposed + mirrored one)
            dx = torch.empty_like(x)
            L.check(lib.mednet_conv3d_fwd(dy.data_ptr(), pk.data_ptr(), None, dx.data_ptr(), n, d, h, wd, cout, cin, 3, dcode, L.NDHWC,
                                          dcode, L.NDHWC, 1, ALGO_MFMA, None, st), "dgrad")
            out["dgrad"] = (dx, None)
            dx = torch.empty_like(x)
            L.check(lib.mednet_conv3d_dgrad_add(dy.data_ptr(), pk.data_ptr(), add.data_ptr(), dx.data_ptr(), n, d, h, wd, cin, cout,
                                                ALGO_MFMA, dcode, st), "dgrad_add")
            out["dgrad_add"] = (dx, None)
            rows = lib.mednet_conv3d_dgrad_gn_rows_dt(n, d, h, wd, cin, cout, ALGO_MFMA, dcode)
            assert rows > 0
            for gact in (0, 1, 2, 3):
                for with_add in (False, True):
                    dx = torch.empty_like(x)
                    part = torch.full((n, rows, cin, 2), float("nan"), device=DEV)
                    L.check(lib.mednet_conv3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), add.data_ptr() if with_add else None, dx.data_ptr(),
                                                       gy.data_ptr(), coef.data_ptr(), gact, part.data_ptr(), n, d, h, wd, cin, cout,
                                                       ALGO_MFMA, dcode, st), "dgrad_gn")
                    out[f"gn{gact}{with_add}"] = (dx, part.double().sum(1))
        torch.cuda.synchronize()
        return out

    assert lib.mednet_set_option(b"conv2b_min_fill", 0) == 0 and lib.mednet_set_option(b"conv2b_min_cin", 16) == 0
    try:
        a, b = run(1), run(0)
    finally:
        lib.mednet_set_option(b"conv2b", 1)
        lib.mednet_set_option(b"conv2b_min_fill", 80)
        lib.mednet_set_option(b"conv2b_min_cin", 64)
    nv = float(np.prod(shape))
    for k in a:
        assert torch.equal(a[k][0], b[k][0]), f"{k}: tensor differs from the general kernel ({(a[k][0] != b[k][0]).float().mean().item():.3%} of the elements)"
        if a[k][1] is not None:
            ta, tb = a[k][1], b[k][1]
            assert ta.shape == tb.shape and bool(torch.isfinite(ta).all()), f"{k}: a row of the partial buffer was not written"
            scale = tb.abs().amax(dim=(1,), keepdim=True) + 1e-3
            assert torch.all((ta - tb).abs() <= 5e-5 * scale + 2e-5 * tb.abs() + 1e-3 * nv ** 0.5 * 1e-2), f"{k}: partial sums differ"
    yr = F.conv3d(x.float().cpu(), w.to(dt).float().cpu(), None, padding=1)
    assert_close(a["act0False"][0], yr, 6e-3 if mode == "bf16" else 1e-3, "y vs ATen")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,shape", [
    (1, (32, 32, 64)),    # 1024 bricks of 2 x 4 x 16: four per workgroup, a contiguous run per XCD
    (2, (16, 32, 32)),    # 512 bricks over two samples: the sample changes between XCDs
    (3, (14, 30, 40)),    # ragged in z, y and x (7 x 8 x 3 = 168 brick slots per sample, 504 in all: interleaved walk, the sample
                          # changes inside a workgroup's list, bricks that hang over every face)
    (2, (15, 34, 24)),    # odd depth: the last brick layer holds one z-plane; 2 x 8 x 9 x 2 = 288 brick slots, not a multiple of 8 per XCD run... (288 = 8 x 36: contiguous runs)
    (1, (17, 22, 50)),    # 9 x 6 x 4 = 216 slots < 256: general kernel; with n = 1 only -- see the next case
    (3, (17, 22, 50)),    # 648 slots = 8 x 81, odd depth, ragged y and x, three samples inside one XCD run
    (1, (10, 12, 100)),   # 105 brick slots: fewer than CUs -> the general kernel keeps the call (plan kind 2)
])
def test_convt_dgrad32_is_bit_identical_to_the_general_kernel(mode, n, shape):
    """The data gradient of the decoder's last ConvTranspose3d (64 -> 32; components.py:259-264) runs convt_dgrad32_mfma_kernel
    (round 6: one wave per SIMD, the wave's 54 weight fragments in registers, whole 64-byte gradient rows staged once for both
    channel blocks).  Same accumulation order as conv_mfma_kernel<2> (option convt_dgrad32=0): dx must be equal BIT FOR BIT, with
    and without the fused first pass of the GroupNorm-3 backward, whose partial sums agree up to fp32 summation order."""
    lib = L.lib()
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    dcode = L.dt(torch.empty(0, dtype=dt))
    CL = torch.channels_last_3d
    g = torch.Generator(device=DEV).manual_seed(29)
    cin, cout = 64, 32
    d, h, wd = shape

    def rand(c, shp):
        return torch.randn(n, c, *shp, device=DEV, generator=g).to(dt).contiguous(memory_format=CL)
    dy = rand(cout, (2 * d, 2 * h, 2 * wd))
    gy, gz = rand(cin, shape), rand(cin, shape)
    w = torch.randn(cin, cout, 3, 3, 3, device=DEV, generator=g) * 0.05
    st = torch.cuda.current_stream().cuda_stream
    ALGO_MFMA = 2
    with mednet_hip.precision(mode):
        pk = ops.pack_conv_weight(w, 3, True)
    plan = (ctypes.c_int * 13)()
    tiles = n * ((d + 1) // 2) * ((h + 3) // 4) * ((wd + 15) // 16)

    def run(special):
        assert lib.mednet_set_option(b"convt_dgrad32", special) == 0
        assert lib.mednet_conv3d_stats_plan(n, d, h, wd, cin, cout, dcode, 1, 2, plan) == 0
        assert plan[0] == (7 if special and tiles >= 256 else 2), list(plan)
        out = {}
        dx = torch.full((n, cin, *shape), float("nan"), device=DEV).to(dt).contiguous(memory_format=CL)
        L.check(lib.mednet_convt3d_dgrad(dy.data_ptr(), pk.data_ptr(), dx.data_ptr(), n, d, h, wd, cin, cout, dcode, dcode, ALGO_MFMA, st),
                "convt3d_dgrad")
        out["plain"] = (dx, None)
        rows = lib.mednet_convt3d_dgrad_gn_rows(n, d, h, wd, cin, cout, dcode, ALGO_MFMA)
        assert rows == plan[7] > 0, (rows, list(plan))
        for gact in (1, 2, 3):
            dx = torch.full((n, cin, *shape), float("nan"), device=DEV).to(dt).contiguous(memory_format=CL)
            part = torch.full((n, rows, cin, 2), float("nan"), device=DEV)
            L.check(lib.mednet_convt3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), dx.data_ptr(), gy.data_ptr(), gz.data_ptr(), gact,
                                                part.data_ptr(), n, d, h, wd, cin, cout, dcode, ALGO_MFMA, st), "convt3d_dgrad_gn")
            out[f"gn{gact}"] = (dx, part.double().sum(1))
        torch.cuda.synchronize()
        return out

    try:
        a, b = run(1), run(0)
    finally:
        lib.mednet_set_option(b"convt_dgrad32", 1)
    nv = float(np.prod(shape))
    for k in a:
        assert bool(torch.isfinite(a[k][0].float()).all()), f"{k}: part of dx was not written"
        assert torch.equal(a[k][0], b[k][0]), f"{k}: dx differs from the general kernel ({(a[k][0] != b[k][0]).float().mean().item():.3%} of the elements)"
        if a[k][1] is not None:
            ta, tb = a[k][1], b[k][1]
            assert ta.shape == tb.shape and bool(torch.isfinite(ta).all()), f"{k}: a row of the partial buffer was not written"
            scale = tb.abs().amax(dim=(1,), keepdim=True) + 1e-3
            assert torch.all((ta - tb).abs() <= 5e-5 * scale + 2e-5 * tb.abs() + 1e-3 * nv ** 0.5 * 1e-2), f"{k}: partial sums differ"
    # ... and against ATen: dx = conv3d(dy, W^T, stride 2, padding 1) -- the adjoint of conv_transpose3d
    xr = torch.zeros(n, cin, *shape, dtype=torch.float64, requires_grad=True)
    F.conv_transpose3d(xr, w.to(dt).double().cpu(), None, stride=2, padding=1, output_padding=1).backward(dy.double().cpu())
    assert_close(a["plain"][0], xr.grad.float(), 6e-3 if mode == "bf16" else 1e-3, "dx vs ATen")


@pytest.mark.parametrize("kind,n,cin,cout,shape", [
    ("conv", 1, 32, 32, (32, 64, 64)),     # conv2b's (high, low) form in place of the 32 -> 32 specialisation, 256 bricks
    ("conv", 2, 64, 128, (16, 24, 48)),    # four channel blocks, 36 bricks per sample: fewer items than CUs, padding items
    ("conv", 1, 32, 64, (8, 8, 8)),        # narrow volume: conv_mfma_kernel<3> with the low image as extra K chunks
    ("convt", 2, 64, 32, (8, 16, 16)),     # ConvTranspose3d forward (extra chunks) and data gradient (conv_mfma_kernel<2>, extra chunks)
    ("first", 2, 1, 32, (24, 32, 32)),     # first layer: the weights' low parts as a third MFMA
])
def test_split_weights_remove_the_weight_rounding(kind, n, cin, cout, shape):
    """fp16x2 (round 6, MEDNET_ALGO_SPLITW_BIT): fp16 storage, every matrix-core convolution multiplies fp16(w) AND fp16(w - fp16(w)).
    Against the exact (fp64) convolution of the SAME fp16-representable activations with the fp32 weights, plain fp16 mode carries
    the weights' 2^-12 rounding (~3e-4 rel-L2 of the output); the split mode must be at the level of the output's own fp16 rounding
    -- forward and data gradient, through conv2b's (high, low) form, the general kernel's extra K chunks, the ConvTranspose3d
    kernels and the first layer."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, *shape, generator=g).half().float()
    res = {}
    for mode in ("fp16", "fp16x2"):
        with mednet_hip.precision(mode):
            if kind == "convt":
                mod = hnn.ConvTranspose3d(cin, cout).to(DEV)
            else:
                mod = hnn.Conv3d(cin, cout, 3, bias=False).to(DEV)
            gw = torch.Generator().manual_seed(7)
            w = (torch.rand(mod.weight.shape, generator=gw) * 2 - 1) * (1.0 / (27 * cin) ** 0.5)
            with torch.no_grad():
                mod.weight.copy_(w)
                if kind == "convt":
                    mod.bias.zero_()
            xg = x.to(DEV).requires_grad_(kind != "first")
            y = mod(xg if kind == "first" else xg.half())
            cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).half()
            if kind != "first":
                y.backward(cot.to(DEV))
            res[mode] = (y.detach().float().cpu(), None if kind == "first" else xg.grad.float().cpu())
    xd, wd = x.double(), w.double()
    xd.requires_grad_(True)
    yr = F.conv_transpose3d(xd, wd, None, stride=2, padding=1, output_padding=1) if kind == "convt" else F.conv3d(xd, wd, None, padding=1)
    yr.backward(cot.double())
    for what, i, ref in (("forward", 0, yr.detach()), ("data gradient", 1, xd.grad)):
        if res["fp16"][i] is None:
            continue
        # the outputs themselves are stored in fp16 (~2e-4 rel-L2 of rounding): compared with the exact result ROUNDED THE SAME WAY,
        # the split result differs only where the fp32 accumulation tips a rounding; the single-image result carries the weights'
        # rounding (~2e-4) in full
        ref16 = ref.float().half().float()
        e1, e2 = rel(res["fp16"][i], ref16), rel(res["fp16x2"][i], ref16)
        print(f"[split weights] {kind} {cin}->{cout} {what}: against the fp16-rounded exact result: fp16 {e1:.2e}  fp16x2 {e2:.2e}")
        assert e2 <= 8e-5, f"{kind} {what}: split-weight error {e2:.3e}"
        assert e2 <= 0.4 * e1, f"{kind} {what}: the low image did not help ({e1:.3e} -> {e2:.3e})"


def test_conv3d_mfma_batch_larger_than_4GB():
    """A batch whose activation tensor exceeds 4 GB (34 x 128^3 x 32 ch bf16 = 4.6 GB): the MFMA kernels address one
    SAMPLE per buffer resource, so the last sample must come out bit-identical to the same sample run alone, and the
    weight gradient of the whole batch must equal the sum of the two half-batch gradients."""
    n, c, e = 34, 32, 128
    g = torch.Generator(device=DEV).manual_seed(7)
    mednet_hip.set_conv_algo("mfma")
    try:
        with mednet_hip.precision("bf16"):
            conv = hnn.Conv3d(c, c, 3, bias=False).to(DEV)
            x = torch.randn(n, c, e, e, e, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last_3d)
            cot = torch.randn(n, c, e, e, e, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last_3d)
            assert x.numel() * 2 > 2 ** 32

            def run(xs, cs):
                conv.weight.grad = None
                xs = xs.detach().requires_grad_(True)
                y = conv(xs)
                y.backward(cs)
                torch.cuda.synchronize()
                return y.detach(), xs.grad, conv.weight.grad.clone()

            y, dx, dw = run(x, cot)
            y1, dx1, _ = run(x[n - 1:], cot[n - 1:])
            assert torch.equal(y[n - 1:], y1) and torch.equal(dx[n - 1:], dx1)
            y0, dx0, _ = run(x[:1], cot[:1])
            assert torch.equal(y[:1], y0) and torch.equal(dx[:1], dx0)
            del y, dx, y0, y1, dx0, dx1
            h = n // 2
            _, _, dwa = run(x[:h], cot[:h])
            _, _, dwb = run(x[h:], cot[h:])
            assert_close(dw, dwa + dwb, 1e-5, "dw of the 4.6 GB batch vs the sum of its halves")
    finally:
        mednet_hip.set_conv_algo("auto")


@pytest.mark.parametrize("n,cout,shape", [(1, 32, (4, 8, 16)), (2, 32, (9, 11, 21)), (1, 64, (5, 6, 7)),
                                          (2, 16, (9, 11, 21)), (1, 48, (4, 8, 16))])  # 16 / 48: half-filled channel block (UNet3D: 1 -> 16)
def test_first_layer_mfma_keeps_fp32_input_precision(n, cout, shape):
    """Cin=1 forward on the matrix cores (contraction over the 27 taps, x split into bf16 hi+lo): with bf16-representable
    weights the only rounding left is the bf16 store of the output; x itself is NOT rounded to 8 bits."""
    tag = f"c1{n}{cout}{shape}"
    x = rnd(tag + "x", n, 1, *shape)                       # full fp32 input, not pre-rounded
    w = bf16_round(rnd(tag + "w", cout, 1, 3, 3, 3, scale=0.2))
    yr = F.conv3d(x, w, None, padding=1)
    with mednet_hip.precision("bf16"):
        conv = hnn.Conv3d(1, cout, 3, bias=False).to(DEV)
        with torch.no_grad():
            conv.weight.copy_(w)
        y = conv(x.to(DEV))
    assert y.dtype == torch.bfloat16
    assert_close(y, yr, 3.0e-3, "y (bf16 store rounding only)")
    assert_close(y, bf16_round(yr), 4e-4, "y vs correctly rounded reference")


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp16x2"])
@pytest.mark.parametrize("n,cout,shape", [
    (3, 32, (40, 72, 80)),    # 3 x 450 bricks > 1024 workgroups: the persistent walk, the sample changes inside a workgroup's list
    (2, 64, (36, 60, 70)),    # two channel blocks: 2 x 360 x 2 items, ragged in x, y and z
    (2, 48, (30, 64, 64)),    # three halves-of-blocks: ncb = 2 with the second block half filled
    (1, 32, (9, 11, 21)),     # fewer items than workgroups: one brick per workgroup
])
def test_first_layer_persistent_walk_and_fused_statistics(mode, n, cout, shape):
    """conv_c1_mfma_kernel (model.py:171-174's first convolution, one input channel) as a persistent kernel (round 6): at most four
    workgroups per CU walk the (brick, channel block) items and a wave keeps its GroupNorm sums over its bricks of a sample.  The
    output must be BIT-identical to the one-workgroup-per-item launch (option conv_c1_persist=0), every partial row must be written
    (NaN-filled buffer), and the rows must add up to the sums of the STORED output per channel pair."""
    lib = L.lib()
    x = rnd(f"c1p{n}{cout}{shape}x", n, 1, *shape)
    w = rnd(f"c1p{n}{cout}{shape}w", cout, 1, 3, 3, 3, scale=0.2)
    res = {}
    try:
        for persist in (1, 0):
            lib.mednet_set_option(b"conv_c1_persist", persist)
            with mednet_hip.precision(mode):
                conv = hnn.Conv3d(1, cout, 3, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w)
                # (the partial buffer comes from torch.empty: fill the allocator's block with NaNs first so that an unwritten row shows)
                rows = lib.mednet_conv3d_fused_stats_chunks(n, *shape, 1, cout, 3, L.F32, L.dt(torch.empty(0, dtype=mednet_hip.config.act_dtype())), mednet_hip.config.conv_algo())
                assert rows > 0
                poison = torch.full((n, rows, cout, 2), float("nan"), device=DEV)
                del poison
                y, partial = conv.forward_with_stats(x.to(DEV))
                torch.cuda.synchronize()
                assert partial is not None and tuple(partial.shape) == (n, rows, cout, 2)
                res[persist] = (y.detach().clone(), partial.detach().double().sum(1).cpu(), rows)
    finally:
        lib.mednet_set_option(b"conv_c1_persist", 1)
    nitems = n * ((shape[0] + 3) // 4) * ((shape[1] + 7) // 8) * ((shape[2] + 15) // 16) * ((cout + 31) // 32)
    assert res[0][2] == 4 * nitems // ((cout + 31) // 32)
    assert res[1][2] <= min(res[0][2], 4 * 1024 // ((cout + 31) // 32)), (res[1][2], res[0][2])  # at most 4 workgroups per CU
    assert torch.equal(res[1][0], res[0][0]), "persistent walk changed the output"
    ys = res[1][0].double().cpu()
    s_ref = ys.sum(dim=(2, 3, 4)).reshape(n, cout // 2, 2).sum(-1)
    q_ref = (ys ** 2).sum(dim=(2, 3, 4)).reshape(n, cout // 2, 2).sum(-1)
    nv = float(np.prod(shape))
    for persist in (1, 0):
        tot = res[persist][1]
        assert bool(torch.isfinite(tot).all()), f"persist={persist}: a partial row was not written"
        assert bool((tot[:, 1::2] == 0).all()), "odd entries of the pair format must be zero"
        assert torch.all((tot[:, 0::2, 0] - s_ref).abs() <= 2e-5 * (q_ref * nv).sqrt() + 1e-3), f"persist={persist}: sum y"
        assert torch.all((tot[:, 0::2, 1] - q_ref).abs() <= 2e-5 * q_ref + 1e-6), f"persist={persist}: sum y^2"
    yr = F.conv3d(x, w, None, padding=1)
    assert_close(res[1][0], yr, 6e-3 if mode == "bf16" else 1e-3, "y vs ATen")


@pytest.mark.parametrize("n,cout,shape", [(1, 32, (4, 8, 16)), (2, 32, (9, 11, 21)), (2, 64, (5, 6, 7)), (1, 32, (20, 24, 40)),
                                          (2, 16, (9, 11, 21))])  # 16: UNet3D's first DoubleConv (components.py:119-121)
def test_first_layer_weight_gradient_on_matrix_cores(n, cout, shape):
    """Cin=1 weight gradient: contraction over voxels with x gathered per tap and split into bf16 hi+lo.  The cotangent is
    bf16-representable, so the result must match the fp32 oracle to fp32-accumulation accuracy, and the VALU kernel."""
    tag = f"w1{n}{cout}{shape}"
    x = rnd(tag + "x", n, 1, *shape)                      # full fp32 input
    cot = bf16_round(rnd(tag + "g", n, cout, *shape))
    wr = torch.zeros(cout, 1, 3, 3, 3, requires_grad=True)
    (F.conv3d(x, wr, None, padding=1) * cot).sum().backward()
    lib = L.lib()
    res = {}
    for opt in (1, 0):
        lib.mednet_set_option(b"wgrad_c1_mfma", opt)
        try:
            with mednet_hip.precision("bf16"):
                conv = hnn.Conv3d(1, cout, 3, bias=False).to(DEV)
                y = conv(x.to(DEV))
                y.backward(cot.to(DEV).bfloat16())
                res[opt] = conv.weight.grad.cpu().clone()
        finally:
            lib.mednet_set_option(b"wgrad_c1_mfma", 1)
    assert_close(res[1], wr.grad, 2e-5, "dw (matrix cores) vs oracle")
    assert_close(res[1], res[0], 2e-5, "dw (matrix cores) vs VALU kernel")


@pytest.mark.parametrize("cin,cout", [(32, 4), (32, 18), (64, 2)])
def test_head_dgrad_kernel(cin, cout):
    x = bf16_round(rnd(f"hd{cin}{cout}x", 2, cin, 6, 8, 10))
    w = rnd(f"hd{cin}{cout}w", cout, cin, 1, 1, 1, scale=0.3)
    cot = rnd(f"hd{cin}{cout}g", 2, cout, 6, 8, 10)
    xr = x.clone().requires_grad_(True)
    (F.conv3d(xr, w) * cot).sum().backward()
    with mednet_hip.precision("bf16"):
        conv = hnn.Conv3d(cin, cout, 1, planar_output=True).to(DEV)
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.zero_()
        xg = x.to(DEV).bfloat16().requires_grad_(True)
        (conv(xg) * cot.to(DEV)).sum().backward()
    assert_close(xg.grad, xr.grad, 4e-3, "dz of the 1x1x1 head")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cin,cout,shape,labels_u8,sigmoid,ignore", [
    (32, 4, (9, 11, 21), True, False, None), (32, 4, (16, 16, 32), False, False, None), (64, 4, (8, 12, 10), True, False, None),
    (16, 2, (5, 7, 33), True, False, None), (32, 3, (12, 8, 8), False, True, None), (32, 4, (6, 10, 12), True, False, 0),
    (64, 1, (4, 6, 7), False, True, None)])
def test_fused_head_dice_against_the_unfused_launches(mode, cin, cout, shape, labels_u8, sigmoid, ignore):
    """ops.head_dice (mednet_head_dice_fwd / _bwd: model.py:207 + loss.py:114-130 in one node) against final_conv + DiceLoss as
    two nodes: logits, loss and the feature gradient BIT-identical (every per-voxel expression is the unfused kernels', in
    their order), the head's weight / bias gradients to 1e-5 (they are summed in another fixed order), and all of it within
    the mode's tolerance of ATen on the CPU.  Label forms: int64 N x D x H x W, and the last channel of a uint8 N x C x D x H
    x W volume consumed where it lies (segmentation.py:60)."""
    n = 2
    tag = f"hd{cin}{cout}{shape}"
    x, w, b = _prep(mode, rnd(tag + "x", n, cin, *shape))[0], rnd(tag + "w", cout, cin, 1, 1, 1, scale=0.3), rnd(tag + "b", cout)
    g = np.random.Generator(np.random.PCG64(77))
    lab_vol = torch.from_numpy(g.integers(0, max(cout, 2) if cout > 1 else 2, size=(n, 3) + shape).astype(np.uint8))
    if cout == 1:
        lab_vol.zero_()  # (one sigmoid channel: every voxel's label is class 0)
    wt = torch.tensor([0.05, 1.0, 1.0, 1.0][:cout]) if cout > 1 else None
    # ATen on the CPU (fp32) from the same rounded features
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lr = F.conv3d(xr, wr, br)
    loss_r = O.DiceLoss(weight=wt, sigmoid_normalization=sigmoid, ignore_index=ignore)(lr, lab_vol[:, -1].long())
    loss_r.backward()
    res = {}
    # (the feature gradient is STORED in the mode's 16-bit type: a Dice gradient of ~1e-7 per voxel sits in fp16's subnormals, which
    #  is what train.LossScaler is for -- the test scales the loss like it does)
    gscale = 3.0 * 16384.0 if mode == "fp16" else 3.0
    with mednet_hip.precision(mode):
        for fused in (False, True):
            conv = hnn.Conv3d(cin, cout, 1, planar_output=True).to(DEV)
            with torch.no_grad():
                conv.weight.copy_(w)
                conv.bias.copy_(b)
            xg = ops.to_cl(x.to(DEV).to(mednet_hip.config.act_dtype())).requires_grad_(True)
            lab_dev = lab_vol.to(DEV)
            lab = lab_dev[:, -1] if labels_u8 else lab_dev[:, -1].long()
            wd = None if wt is None else wt.to(DEV)
            if fused:
                assert ops.head_dice_supported(xg, cin, cout, lab)
                lg, loss = ops.head_dice(xg, conv.weight, conv.bias, conv._packed(), lab, wd, 1e-5, sigmoid, ignore)
            else:
                lg = conv(xg)
                loss = ops.dice_loss(lg, lab.long(), wd, 1e-5, sigmoid, ignore)
            (loss * gscale).backward()
            res[fused] = (lg.detach(), loss.detach(), xg.grad, conv.weight.grad, conv.bias.grad)
    (l0, s0, dx0, dw0, db0), (l1, s1, dx1, dw1, db1) = res[False], res[True]
    assert torch.equal(l0, l1), "logits differ between the fused and the unfused head"
    assert torch.equal(s0, s1), (float(s0), float(s1))
    assert torch.equal(dx0, dx1), "feature gradient differs between the fused and the unfused path"
    assert_close(dw1, dw0, 1e-5, "dW fused vs unfused")
    assert_close(db1, db0, 1e-5, "db fused vs unfused")
    tol = TOL[mode]
    assert_close(l1, lr, tol, "logits vs ATen")
    assert abs(float(s1) - float(loss_r)) <= tol
    assert_close(dx1.float() / gscale, xr.grad, max(tol, 1e-4), "dx vs ATen")
    assert_close(dw1 / gscale, wr.grad, max(tol, 1e-4), "dW vs ATen")
    assert_close(db1 / gscale, br.grad, max(tol, 1e-4), "db vs ATen")


@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (9, 11, 21)), (16, 48, (8, 8, 16)), (64, 32, (4, 8, 16))])
def test_plain_bf16_pack_serves_an_fp32_storage_call(cin, cout, shape):
    """mednet_conv3d_pack (the round-2 entry point, bf16 images) followed by an fp32-storage convolution.  The
    split-bf16 kernels contract against the high AND the low weight images; every bf16 pack now writes both, so the call is the
    1e-3 product, not a contraction against whatever the buffer held before."""
    n = 1
    tag = f"pk{cin}{cout}"
    x, w = rnd(tag + "x", n, cin, *shape), rnd(tag + "w", cout, cin, 3, 3, 3, scale=0.2)
    yr = F.conv3d(x.double(), w.double(), padding=1)
    lib = L.lib()
    wd = w.to(DEV).contiguous()
    buf = torch.empty(lib.mednet_conv3d_pack_bytes(cin, cout, 3), dtype=torch.uint8, device=DEV)
    buf.view(torch.int32)[: buf.numel() // 4].fill_(0x7FC07FC0)  # what an unwritten low image would look like: NaNs
    L.check(lib.mednet_conv3d_pack(wd.data_ptr(), buf.data_ptr(), cin, cout, 3, 0, L.stream()), "conv3d_pack")
    xd = ops.to_cl(x.to(DEV))
    y = ops.empty_cl(n, cout, *shape, torch.float32, DEV)
    L.check(lib.mednet_conv3d_fwd(xd.data_ptr(), buf.data_ptr(), None, y.data_ptr(), n, *shape, cin, cout, 3, L.F32, L.NDHWC, L.F32,
                                  L.NDHWC, 0, L.ALGO_AUTO, None, L.stream()), "conv3d_fwd")
    assert bool(torch.isfinite(y).all())
    assert_close(y, yr.float(), 3e-5, "fp32-storage conv on a plain bf16 pack")


def test_losses_take_the_label_volume_where_it_lies():
    """landmarks.py:68-70 / segmentation.py:60 slice a uint8 N x (H + 1) x D x H x W label volume into heat maps [:, :-1] and class
    labels [:, -1].  ops.dice_loss takes the uint8 label view as it is (no `.long()` cast kernel), ops.heatmap_loss the strided
    heat-map view (no `.contiguous()` copy); both must equal the int64 / contiguous forms bit for bit, forward and backward."""
    n, nh, nc, shape = 2, 3, 2, (9, 10, 21)
    g = np.random.Generator(np.random.PCG64(5))
    vol = torch.from_numpy(g.integers(0, 256, size=(n, nh + 1) + shape).astype(np.uint8))
    vol[:, -1] = torch.from_numpy(g.integers(0, nc, size=(n,) + shape).astype(np.uint8))
    vol = vol.to(DEV)
    lg = rnd("lvl", n, nh + nc, *shape).to(DEV)
    w = torch.tensor([0.05, 1.0], device=DEV)
    res = []
    for direct in (False, True):
        x = lg.clone().requires_grad_(True)
        out_hm, out_cls = x[:, :nh], x[:, nh:]
        labels = vol[:, -1] if direct else vol[:, -1].long()
        heat = vol[:, :-1] if direct else vol[:, :-1].contiguous()
        if direct:
            assert labels.dtype == torch.uint8 and not heat.is_contiguous()
        loss = ops.dice_loss(out_cls, labels, w) + ops.heatmap_loss(out_hm, heat, [0.015, 0.02, 0.03], "L2")
        loss.backward()
        res.append((loss.detach().clone(), x.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[0][1].abs().max()) > 0


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("n,c,shape,pool,residual", [(2, 32, (8, 12, 16), "max", True), (1, 64, (6, 4, 10), "avg", True),
                                                     (2, 16, (4, 8, 6), "max", False), (1, 128, (2, 2, 2), "max", True)])
def test_groupnorm_apply_fused_with_the_pooling_that_follows(mode, n, c, shape, pool, residual):
    """mednet_gn_act_pool_fwd (an encoder block's last GroupNorm apply + residual + ELU and the next level's 2x2x2 pooling in one
    pass, components.py:177-178 -> :222-224) against mednet_gn_act_fwd followed by mednet_pool2_fwd: block output and pooled tensor
    bit-identical (max with ties and avg)."""
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[mode]
    d, h, w = shape
    y = ops.to_cl((rnd("gp_y", n, c, *shape) * 2).round().div(2).to(dt).to(DEV))  # (coarse values: ties inside pooling windows)
    r = ops.to_cl(rnd("gp_r", n, c, *shape).to(dt).to(DEV)) if residual else None
    coef = torch.stack((torch.ones(n, c), torch.zeros(n, c)), dim=-1).to(DEV).contiguous()  # identity affine keeps the ties
    lib = L.lib()
    pm = L.POOL_MAX if pool == "max" else L.POOL_AVG
    assert lib.mednet_gn_act_pool_supported(d, h, w, c, L.dt_of(dt))
    z0, z1 = torch.empty_like(y, memory_format=ops.CL), torch.empty_like(y, memory_format=ops.CL)
    p0, p1 = (ops.empty_cl(n, c, d // 2, h // 2, w // 2, dt, DEV) for _ in range(2))
    L.check(lib.mednet_gn_act_fwd(y.data_ptr(), coef.data_ptr(), L.ptr(r), z0.data_ptr(), n, d * h * w, c, L.ACT_ELU, L.dt_of(dt),
                                  L.dt_of(dt), L.stream()), "gn_act_fwd")
    L.check(lib.mednet_pool2_fwd(z0.data_ptr(), p0.data_ptr(), n, d, h, w, c, pm, L.dt_of(dt), L.stream()), "pool2_fwd")
    L.check(lib.mednet_gn_act_pool_fwd(y.data_ptr(), coef.data_ptr(), L.ptr(r), z1.data_ptr(), p1.data_ptr(), n, d, h, w, c, L.ACT_ELU,
                                       pm, L.dt_of(dt), L.stream()), "gn_act_pool_fwd")
    assert torch.equal(z0, z1) and torch.equal(p0, p1)
    assert not lib.mednet_gn_act_pool_supported(5, h, w, c, L.dt_of(dt))  # odd extents keep the two launches


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("n,c,groups,shape,pool,skip", [(2, 32, 8, (8, 12, 16), "max", True), (1, 64, 8, (6, 4, 10), "avg", True),
                                                        (2, 16, 2, (4, 8, 6), "max", False), (1, 128, 8, (4, 4, 4), "max", True)])
def test_groupnorm_backward_rebuilds_the_pooled_gradient_it_was_not_given(mode, n, c, groups, shape, pool, skip):
    """mednet_pool2_bwd_gn(dx = NULL) + mednet_gn_act_bwd_fused_res_pool (the gradient of an encoder block's output is never in
    memory; components.py:177-178 <- :222-224) against mednet_pool2_bwd_gn + mednet_gn_act_bwd_fused_res: the sums, dy3, dres,
    dgamma and dbeta bit-identical, max pooling with ties in the windows.  (The 128-channel case runs the two-launch reducer.)"""
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[mode]
    d, h, w = shape
    sp = d * h * w
    lib = L.lib()
    pm = L.POOL_MAX if pool == "max" else L.POOL_AVG
    y3 = ops.to_cl(rnd("lp_y", n, c, *shape).to(dt).to(DEV))
    out = ops.to_cl((rnd("lp_o", n, c, *shape) * 2).round().div(2).to(dt).to(DEV))  # block output: coarse values, ties
    dyp = ops.to_cl(rnd("lp_g", n, c, d // 2, h // 2, w // 2).to(dt).to(DEV))
    dsk = ops.to_cl(rnd("lp_s", n, c, *shape).to(dt).to(DEV)) if skip else None
    gamma = (1 + 0.1 * rnd("lp_ga", c)).to(DEV)
    stats = torch.stack((0.1 * rnd("lp_m", n, groups), 1 + 0.1 * rnd("lp_r", n, groups).abs()), dim=-1).to(DEV).contiguous()
    coef = torch.stack((torch.ones(n, c), torch.zeros(n, c)), dim=-1).to(DEV).contiguous()
    rows = lib.mednet_pool2_bwd_gn_rows(n, d, h, w, c, L.dt_of(dt))
    assert rows > 0
    ws = L.workspace(lib.mednet_gn_ws_bytes(n, c, sp), DEV)
    res = []
    lib.mednet_set_option(b"gn_bwd_one_launch", 0 if c == 128 else 1)
    for lazy in (False, True):
        part = torch.full((n, rows, c, 2), float("nan"), device=DEV)
        dx = torch.full_like(out, float("nan"))
        L.check(lib.mednet_pool2_bwd_gn(dyp.data_ptr(), out.data_ptr(), L.ptr(dsk), None if lazy else dx.data_ptr(), y3.data_ptr(),
                                        L.ACT_ELU, part.data_ptr(), n, d, h, w, c, pm, L.dt_of(dt), L.stream()), "pool2_bwd_gn")
        dy3, dres = torch.full_like(out, float("nan")), torch.full_like(out, float("nan"))
        dg, db = torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
        if lazy:
            L.check(lib.mednet_gn_act_bwd_fused_res_pool(dyp.data_ptr(), L.ptr(dsk), y3.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                                         gamma.data_ptr(), part.data_ptr(), rows, dy3.data_ptr(), dres.data_ptr(),
                                                         dg.data_ptr(), db.data_ptr(), n, d, h, w, c, groups, L.ACT_ELU, pm,
                                                         L.dt_of(dt), ws.data_ptr(), ws.numel(), L.stream()), "fused_res_pool")
            assert torch.isnan(dx.float()).all()  # nothing wrote the gradient tensor
        else:
            L.check(lib.mednet_gn_act_bwd_fused_res(dx.data_ptr(), y3.data_ptr(), out.data_ptr(), coef.data_ptr(), stats.data_ptr(),
                                                    gamma.data_ptr(), part.data_ptr(), rows, dy3.data_ptr(), dres.data_ptr(),
                                                    dg.data_ptr(), db.data_ptr(), n, sp, c, groups, L.ACT_ELU, L.dt_of(dt),
                                                    ws.data_ptr(), ws.numel(), L.stream()), "fused_res")
        torch.cuda.synchronize()
        res.append((part, dy3, dres, dg, db))
    lib.mednet_set_option(b"gn_bwd_one_launch", 1)
    for name, a, b in zip(("partial", "dy3", "dres", "dgamma", "dbeta"), *res):
        assert not torch.isnan(a.float()).any(), name
        if not torch.equal(a, b):
            bad = (a != b).nonzero()
            i = tuple(bad[0].tolist())
            raise AssertionError(f"{name}: {len(bad)} of {a.numel()} differ, first at {i}: {a[i].item()!r} vs {b[i].item()!r}; "
                                 f"max |diff| {(a.float() - b.float()).abs().max().item():.3e}")


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("n,ce,cx,shape,xshape", [(2, 32, 64, (8, 12, 16), (4, 6, 8)), (1, 64, 128, (6, 6, 10), (3, 3, 5)),
                                                  (2, 16, 32, (9, 7, 11), (4, 3, 5))])
def test_concatenation_kernel_takes_the_groupnorm_sums_of_what_it_writes(mode, n, ce, cx, shape, xshape):
    """mednet_upcat_fwd_stats (UNet3D's decoder: interpolate + cat, components.py:277-280, followed by the GroupNorm that opens its
    'gcr' block, :46-57) against mednet_upcat_fwd + the stand-alone statistics pass: the concatenated tensor bit-identical, the
    finalised statistics and coefficients equal to 1e-6 (the sums run in another order), odd sizes included."""
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[mode]
    groups, eps = 8, 1e-5
    d, h, w = shape
    sp = d * h * w
    ct = ce + cx
    enc = ops.to_cl(rnd("uc_e", n, ce, *shape).to(dt).to(DEV))
    x = ops.to_cl(rnd("uc_x", n, cx, *xshape).to(dt).to(DEV))
    gamma, beta = (1 + 0.1 * rnd("uc_g", ct)).to(DEV), (0.1 * rnd("uc_b", ct)).to(DEV)
    lib = L.lib()
    out0, out1 = (ops.empty_cl(n, ct, d, h, w, dt, DEV) for _ in range(2))
    L.check(lib.mednet_upcat_fwd(enc.data_ptr(), x.data_ptr(), out0.data_ptr(), n, d, h, w, ce, *xshape, cx, L.dt_of(dt), L.stream()), "upcat")
    chunks = lib.mednet_upcat_stats_chunks(n, d, h, w, ce, cx, L.dt_of(dt))
    assert chunks > 0
    part = torch.full((n, chunks, ct, 2), float("nan"), device=DEV)
    L.check(lib.mednet_upcat_fwd_stats(enc.data_ptr(), x.data_ptr(), out1.data_ptr(), part.data_ptr(), n, d, h, w, ce, *xshape, cx,
                                       L.dt_of(dt), L.stream()), "upcat_stats")
    assert torch.equal(out0, out1) and not torch.isnan(part).any()
    ws = L.workspace(lib.mednet_gn_ws_bytes(n, ct, sp), DEV)
    st0, cf0, st1, cf1 = (torch.empty(n, groups, 2, device=DEV), torch.empty(n, ct, 2, device=DEV),
                          torch.empty(n, groups, 2, device=DEV), torch.empty(n, ct, 2, device=DEV))
    L.check(lib.mednet_gn_stats(out0.data_ptr(), gamma.data_ptr(), beta.data_ptr(), st0.data_ptr(), cf0.data_ptr(), n, sp, ct, groups, eps,
                                L.dt_of(dt), ws.data_ptr(), ws.numel(), L.stream()), "gn_stats")
    L.check(lib.mednet_gn_finalize(part.data_ptr(), chunks, gamma.data_ptr(), beta.data_ptr(), st1.data_ptr(), cf1.data_ptr(), n, sp, ct,
                                   groups, eps, ws.data_ptr(), ws.numel(), L.stream()), "gn_finalize")
    torch.cuda.synchronize()
    assert_close(st1, st0, 1e-6, "statistics")
    assert_close(cf1, cf0, 1e-6, "coefficients")
    assert lib.mednet_upcat_stats_chunks(n, d, h, w, ce + 4, cx, L.dt_of(dt)) == 0  # channel counts that are not multiples of 8
