"""Building blocks of the 3D U-Nets with the reference's names, signatures, child-module names and parameter shapes
(midasmednet/unet/components.py), executing on libmednet_hip.

Fusions applied inside SingleConv / ExtResNetBlock / Decoder (the reference runs each as a separate ATen op):
  conv -> [GroupNorm -> activation]           one GN-stats + one apply pass      (components.py:44,57,36-40)
  GroupNorm(conv3) + residual -> activation   folded into that same apply pass   (components.py:175-178)
  ConvTranspose3d + `x += encoder_features`   skip added in the conv epilogue    (components.py:283-284)
"""
from __future__ import annotations

import torch
import torch.nn as nn

import os

from .. import _lib as L
from .. import block
from .. import config
from .. import nn as hnn
from .. import ops

_ACT_MODULES = {"r": ("ReLU", hnn.ReLU, L.ACT_RELU), "l": ("LeakyReLU", hnn.LeakyReLU, L.ACT_LEAKY),
                "e": ("ELU", hnn.ELU, L.ACT_ELU)}


def conv3d(in_channels, out_channels, kernel_size, bias, padding=1):
    return hnn.Conv3d(in_channels, out_channels, kernel_size, padding=padding, bias=bias)


def create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding=1):
    """Order-string grammar of components.py:12-67: c conv, g groupnorm, b batchnorm, r/l/e activations."""
    assert "c" in order, "Conv layer MUST be present"
    assert order[0] not in "rle", "Non-linearity cannot be the first operation in the layer"
    conv_pos = order.index("c")
    modules = []
    for i, ch in enumerate(order):
        if ch in _ACT_MODULES:
            name, cls, _ = _ACT_MODULES[ch]
            modules.append((name, cls(inplace=True)))
        elif ch == "c":
            modules.append(("conv", conv3d(in_channels, out_channels, kernel_size,
                                           bias=not ("g" in order or "b" in order), padding=padding)))
        elif ch == "g":
            nch = in_channels if i < conv_pos else out_channels
            groups = 1 if nch < num_groups else num_groups
            assert nch % groups == 0, (f"Expected number of channels in input to be divisible by num_groups. "
                                       f"num_channels={nch}, num_groups={groups}")
            modules.append(("groupnorm", hnn.GroupNorm(num_groups=groups, num_channels=nch)))
        elif ch == "b":
            # BatchNorm is reachable through the grammar but used by no caller of the reference; it stays a stock
            # torch module (off the hot path, components.py:58-63).
            modules.append(("batchnorm", nn.BatchNorm3d(in_channels if i < conv_pos else out_channels)))
        else:
            raise ValueError(f"Unsupported layer type '{ch}'. MUST be one of ['b', 'g', 'r', 'l', 'e', 'c']")
    return modules


def _kinked(order: str) -> bool:
    """ReLU / LeakyReLU in the layer: the network is piecewise linear and asks for exact fp32 products in the fp32 storage
    mode (config.exact_products says why)."""
    return "r" in order or "l" in order


class SingleConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, order="crg", num_groups=8, padding=1):
        super().__init__()
        for name, module in create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding=padding):
            self.add_module(name, module)
        self._kinked = _kinked(order)

    def forward(self, x, residual=None, final_act=L.ACT_NONE, partial=None, stats_for=None):
        with config.exact_products(self._kinked or final_act in (L.ACT_RELU, L.ACT_LEAKY)):
            return self._forward(x, residual, final_act, partial, stats_for)

    def _forward(self, x, residual=None, final_act=L.ACT_NONE, partial=None, stats_for=None):
        """`residual`/`final_act` let ExtResNetBlock fold `out += residual; act(out)` into the last GroupNorm.
        `partial`: GroupNorm partial sums of the INPUT x (from the kernel that produced it), used when this layer opens with
        a GroupNorm ('gcr').  `stats_for`: the GroupNorm module that will consume this layer's output (DoubleConv passes
        the next SingleConv's); when the layer ends in conv -> activation fused into one kernel, that kernel also writes
        the partial sums for it and forward returns (out, partial)."""
        mods = list(self._modules.values())
        i = 0
        fused_res = False
        out_partial = None
        while i < len(mods):
            m = mods[i]
            if isinstance(m, hnn.GroupNorm):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, hnn._Act):
                    x = m(x, act=nxt.code, partial=partial)
                    i += 2
                elif nxt is None and residual is not None:
                    x = m(x, act=final_act, residual=residual, partial=partial)
                    fused_res = True
                    i += 1
                else:
                    x = m(x, partial=partial)
                    i += 1
                partial = None
                continue
            partial = None
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, hnn.Conv3d) and isinstance(nxt, hnn.GroupNorm) \
                    and m.bias is None and x.is_cuda and (m.out_channels // nxt.num_groups) % 2 == 0:
                x, partial = m.forward_with_stats(x)
            elif isinstance(m, hnn.Conv3d) and isinstance(nxt, hnn._Act) and m.bias is None and m.kernel_size[0] == 3 \
                    and not m.planar_output and ops.conv3d_act_supported(x, m.in_channels, m.out_channels):
                # conv -> activation in one kernel; if it is the end of the layer, also the next GroupNorm's statistics
                last = i + 2 == len(mods)
                want = last and stats_for is not None and stats_for.num_channels == m.out_channels \
                    and (m.out_channels // stats_for.num_groups) % 2 == 0
                x, p = ops.conv3d_act(x, m.weight, m._packed(x), nxt.code, want)
                if last:
                    out_partial = p
                i += 2
                continue
            elif isinstance(m, nn.BatchNorm3d):
                x = m(x.float().contiguous()).to(memory_format=ops.CL)
            else:
                x = m(x)
            i += 1
        if residual is not None and not fused_res:
            x = ops.activation(ops.AddFn.apply(x, residual), final_act)
        if stats_for is not None:
            return x, out_partial
        return x


class DoubleConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, encoder, kernel_size=3, order="crg", num_groups=8):
        super().__init__()
        if encoder:
            c1_in, c1_out = in_channels, max(out_channels // 2, in_channels)
            c2_in, c2_out = c1_out, out_channels
        else:
            c1_in, c1_out, c2_in, c2_out = in_channels, out_channels, out_channels, out_channels
        self.add_module("SingleConv1", SingleConv(c1_in, c1_out, kernel_size, order, num_groups))
        self.add_module("SingleConv2", SingleConv(c2_in, c2_out, kernel_size, order, num_groups))

    def forward(self, x, partial=None):
        """SingleConv1 -> SingleConv2 (components.py:93-133).  When SingleConv2 opens with a GroupNorm ('gcr'), the kernel
        that ends SingleConv1 (conv + activation) also produces that GroupNorm's partial sums.  `partial` (not part of the
        reference's signature): GroupNorm partial sums of x from the kernel that wrote it, for a SingleConv1 that opens with a
        GroupNorm (the decoder's concatenation kernel, Decoder.forward)."""
        sc1, sc2 = self.SingleConv1, self.SingleConv2
        first2 = next(iter(sc2._modules.values()))
        if isinstance(first2, hnn.GroupNorm) and torch.is_tensor(x) and x.is_cuda:
            x, p2 = sc1(x, partial=partial, stats_for=first2)
            return sc2(x, partial=p2)
        return sc2(sc1(x, partial=partial))


class ExtResNetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, order="cge", num_groups=8, **kwargs):
        super().__init__()
        self.conv1 = SingleConv(in_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups)
        self.conv2 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups)
        stripped = order.translate({ord(c): None for c in "rel"})
        self.conv3 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=stripped,
                                num_groups=num_groups)
        key = "l" if "l" in order else ("e" if "e" in order else "r")
        self.non_linearity = _ACT_MODULES[key][1](inplace=True)
        self._kinked = key != "e"

    def _plain(self):
        """(convs, norms) when the block is 3 x [3^3 conv without bias -> GroupNorm (-> activation)] with one group count:
        the shape every caller of the reference builds ('cge' and friends) and the one the single-node path takes."""
        convs, norms = [], []
        for sc, want_act in ((self.conv1, True), (self.conv2, True), (self.conv3, False)):
            mods = list(sc._modules.values())
            if len(mods) != (3 if want_act else 2) or not isinstance(mods[0], hnn.Conv3d) or not isinstance(mods[1], hnn.GroupNorm):
                return None
            if want_act and (not isinstance(mods[2], hnn._Act) or mods[2].code != self.non_linearity.code):
                return None
            if mods[0].bias is not None or mods[0].kernel_size[0] != 3 or not mods[1].affine:
                return None
            convs.append(mods[0])
            norms.append(mods[1])
        if len({m.num_groups for m in norms}) != 1 or len({m.eps for m in norms}) != 1:
            return None
        return convs, norms

    def forward(self, x, pool_mode=None):
        """`pool_mode` (not part of the reference's signature; components.py:167-180 takes x only): the caller will pool this
        block's output with a 2x2x2 pooling of that mode next -- the single-node path then writes the pooled tensor in the pass
        that writes the output.  Purely an optimisation hint: the result is the same tensor either way."""
        with config.exact_products(self._kinked):
            return self._forward(x, pool_mode)

    def _forward(self, x, pool_mode=None):
        plain = self._plain() if (x.is_cuda and block.ENABLED) else None
        if plain is not None:
            convs, norms = plain
            return block.res_block(x, convs, norms, norms[0].num_groups, norms[0].eps, self.non_linearity.code, pool_mode=pool_mode)
        residual = self.conv1(x)
        out = self.conv2(residual)
        return self.conv3(out, residual=residual, final_act=self.non_linearity.code)


def _module_kinked(module) -> bool:
    return any(getattr(m, "_kinked", False) for m in module.modules())


_FUSE_SKIP_POOL = os.environ.get("MEDNET_SKIP_POOL", "1") == "1"  # A/B knob


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=(2, 2, 2),
                 pool_type="max", basic_module=DoubleConv, conv_layer_order="crg", num_groups=8):
        super().__init__()
        assert pool_type in ["max", "avg"]
        if apply_pooling:
            self.pooling = (hnn.MaxPool3d if pool_type == "max" else hnn.AvgPool3d)(kernel_size=pool_kernel_size)
        else:
            self.pooling = None
        self.basic_module = basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups)

    def forward(self, x, with_skip=False, pool_next=None, sole_consumer=False):
        """forward(x) is the reference's Encoder.forward (components.py:222-226).  `with_skip=True` -> (skip, out): `skip` is
        x as the decoder will use it, `out` = forward(x).  With the 2x2x2 pooling of this package the split is one autograd
        node (ops.SkipPool2Fn), so the two gradients of x meet inside the pooling backward kernel; any other pooling module
        takes the plain path (autograd adds them).  Both forms go through `__call__`, so module hooks see every level."""
        # `pool_next`: the mode of the NEXT level's 2x2x2 pooling, when the caller knows this level's output goes there (the U-Net's
        # own forward does): an ExtResNetBlock then writes the pooled tensor beside its output (block.PoolStash)
        def body(t):
            if pool_next is not None and isinstance(self.basic_module, ExtResNetBlock):
                return self.basic_module(t, pool_mode=pool_next)
            return self.basic_module(t)

        if not with_skip:
            if self.pooling is not None:
                x = self.pooling(x)
            return body(x)
        if _FUSE_SKIP_POOL and isinstance(self.pooling, hnn._Pool2) and torch.is_tensor(x) and x.is_cuda and x.requires_grad:
            # sole_consumer: the caller uses the returned skip tensor from here on and x nowhere else (ops.skip_pool2)
            skip, pooled = ops.skip_pool2(x, self.pooling.mode, sole_consumer=sole_consumer)
            return skip, body(pooled)
        return x, self.forward(x, pool_next=pool_next)

    def forward_with_skip(self, x):
        return self(x, with_skip=True)


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=(2, 2, 2), basic_module=DoubleConv,
                 conv_layer_order="crg", num_groups=8):
        super().__init__()
        if basic_module == DoubleConv:
            self.upsample = None  # nearest-neighbour interpolation + concatenation joining
        else:
            self.upsample = hnn.ConvTranspose3d(in_channels, out_channels, kernel_size=kernel_size, stride=scale_factor,
                                                padding=1, output_padding=1)
            in_channels = out_channels
        self.basic_module = basic_module(in_channels, out_channels, encoder=False, kernel_size=kernel_size,
                                         order=conv_layer_order, num_groups=num_groups)

    def forward(self, encoder_features, x):
        with config.exact_products(_module_kinked(self.basic_module)):  # (the upsampling follows its block)
            if self.upsample is None:
                bm = self.basic_module
                first = next(iter(bm.SingleConv1._modules.values())) if isinstance(bm, DoubleConv) else None
                if isinstance(first, hnn.GroupNorm) and torch.is_tensor(x) and x.is_cuda and first.num_channels % (2 * first.num_groups) == 0:
                    # 'g c r': the block opens with a GroupNorm over the concatenation -- the kernel that writes it takes the sums
                    x, partial = ops.upsample_concat(encoder_features, x, want_stats=True)
                    return bm(x, partial=partial)
                x = ops.upsample_concat(encoder_features, x)
            else:
                x = self.upsample(x, skip=encoder_features)
            return self.basic_module(x)


class FinalConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, order="crg", num_groups=8):
        super().__init__()
        self.add_module("SingleConv", SingleConv(in_channels, in_channels, kernel_size, order, num_groups))
        self.add_module("final_conv", hnn.Conv3d(in_channels, out_channels, 1))
