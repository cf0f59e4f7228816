"""Leaf nn.Modules of the MI355X path.  They keep the parameter names/shapes/default init of the torch.nn modules the
reference instantiates (components.py:8-9,36-40,57,210-212,259-264), so checkpoints interchange, but their forward
runs libmednet_hip kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib as L
from . import config, ops


class _PackedWeightMixin:
    """Caches the packed (tap-major / MFMA-fragment) image of `self.weight`; rebuilt whenever the parameter's
    version counter moves (optimizer.step, load_state_dict, .copy_)."""

    _transposed = False

    def _packed(self, x=None, converts=False):
        """`x`: the tensor the op is about to read (None: one the op made itself, in the mode's storage type); `converts`: the op
        casts its input to the storage type first.  Needed for packs the trainer rewrote with MEDNET_PACK_HIGH_ONLY (train.
        BatchedRepack): their fp32 and unrequested low images are stale, so a call that would take another path than the 16-bit
        matrix-core kernels gets the layer packed in full first."""
        w = self.weight
        key = (w.data_ptr(), w._version, getattr(w, "_mednet_step", 0), str(w.device), config.act_dtype(), config.pack_elt())
        if getattr(self, "_pack_key", None) != key:
            self._pack_buf = ops.pack_conv_weight(w, self.kernel_size[0], self._transposed)
            self._pack_key = key
        buf = self._pack_buf
        if getattr(buf, "_mednet_lean", False) and not self._lean_call_ok(x, converts):
            ops.pack_conv_weight(w, self.kernel_size[0], self._transposed, out=buf)
            buf._mednet_lean = False
        return buf

    def _lean_layer_ok(self) -> bool:
        """Structurally: does this layer ALWAYS run on the 16-bit matrix-core kernels when it is given 16-bit tensors?"""
        return False

    def _lean_call_ok(self, x, converts) -> bool:
        from . import _lib
        if not (config.is_half_mode() and self._lean_layer_ok()):
            return False
        if config._decompose(config.conv_algo())[0] == _lib.ALGO_DIRECT:
            return False
        if x is not None:
            if x.dim() != 5 or (x.dtype != config.act_dtype() and not converts):
                return False
            vox = x.shape[2] * x.shape[3] * x.shape[4] * (8 if self._transposed else 1)
            if vox * max(self.in_channels, self.out_channels) * 2 >= 4294960000:  # (conv_mfma_fits: one sample below 4 GB)
                return False
        return True


class Conv3d(nn.Module, _PackedWeightMixin):
    """nn.Conv3d(cin, cout, k, padding=k//2) restricted to what the U-Net uses (k in {1,3}, stride 1)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=None, bias=True, planar_output=False):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        if k not in (1, 3):
            raise NotImplementedError(f"mednet_hip.Conv3d: kernel_size={kernel_size} (supported: 1, 3)")
        pad = k // 2 if padding is None else (padding if isinstance(padding, int) else padding[0])
        if pad != k // 2:
            raise NotImplementedError(f"mednet_hip.Conv3d: padding={padding} with kernel_size={k} (only 'same')")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.padding, self.stride = (k, k, k), (pad, pad, pad), (1, 1, 1)
        self.planar_output = planar_output
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k, k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):  # torch.nn.modules.conv._ConvNd.reset_parameters
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] ** 3
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def _lean_layer_ok(self) -> bool:  # (conv_mfma_supported: 3x3x3, channels in 16s, no bias, channels-last 16-bit output)
        return (self.kernel_size[0] == 3 and self.bias is None and not self.planar_output and self.in_channels % 16 == 0
                and self.out_channels % 16 == 0)

    def forward(self, x):
        out_dtype = torch.float32 if self.planar_output else config.act_dtype()
        return ops.conv3d(x, self.weight, self.bias, self._packed(x), self.kernel_size[0], self.planar_output, out_dtype)

    def forward_with_stats(self, x):
        """(y, GroupNorm partial sums of y or None) -- used when a GroupNorm follows (SingleConv fuses the two)."""
        return ops.conv3d_with_stats(x, self.weight, self.bias, self._packed(x), self.kernel_size[0])

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, bias={self.bias is not None}"


class ConvTranspose3d(nn.Module, _PackedWeightMixin):
    """nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1); forward(x, skip=None) adds `skip`."""

    _transposed = True

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=(2, 2, 2), padding=1, output_padding=1):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        st = stride if isinstance(stride, int) else stride[0]
        if (k, st, padding, output_padding) != (3, 2, 1, 1) or (not isinstance(stride, int) and len(set(stride)) != 1):
            raise NotImplementedError("mednet_hip.ConvTranspose3d: only k=3, stride=2, padding=1, output_padding=1")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.output_padding = (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 3, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = self.weight.size(1) * 27  # torch uses weight.size(1)*receptive field for transposed convs too
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.bias, -bound, bound)

    def _lean_layer_ok(self) -> bool:  # (mednet_convt3d_fwd / _dgrad: matrix-core kernels for channels in 32s)
        return self.in_channels % 32 == 0 and self.out_channels % 32 == 0

    def forward(self, x, skip=None):
        return ops.conv_transpose3d(x, self.weight, self.bias, skip, self._packed(x, converts=True))


class GroupNorm(nn.Module):
    """nn.GroupNorm(num_groups, num_channels, eps=1e-5, affine=True); `forward(x, act=..., residual=...)` fuses
    the following activation and residual add."""

    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True):
        super().__init__()
        if num_channels % num_groups:
            raise ValueError("num_channels must be divisible by num_groups")
        self.num_groups, self.num_channels, self.eps, self.affine = num_groups, num_channels, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_channels))
            self.bias = nn.Parameter(torch.zeros(num_channels))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x, act=L.ACT_NONE, residual=None, partial=None):
        return ops.group_norm_act(x, self.weight, self.bias, self.num_groups, self.eps, act, residual, partial)

    def extra_repr(self):
        return f"{self.num_groups}, {self.num_channels}, eps={self.eps}"


class _Act(nn.Module):
    code = L.ACT_NONE

    def __init__(self, inplace=True):
        super().__init__()
        self.inplace = inplace  # kept for signature parity; tensors are never aliased here

    def forward(self, x):
        return ops.activation(x, self.code)


class ReLU(_Act):
    code = L.ACT_RELU


class LeakyReLU(_Act):
    code = L.ACT_LEAKY

    def __init__(self, negative_slope=0.1, inplace=True):
        super().__init__(inplace)
        if abs(negative_slope - 0.1) > 1e-12:
            raise NotImplementedError("mednet_hip.LeakyReLU: only negative_slope=0.1 (components.py:38)")
        self.negative_slope = negative_slope


class ELU(_Act):
    code = L.ACT_ELU


class _Pool2(nn.Module):
    mode = L.POOL_MAX

    def __init__(self, kernel_size=(2, 2, 2)):
        super().__init__()
        ks = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        if ks != (2, 2, 2):
            raise NotImplementedError(f"mednet_hip pooling: kernel_size={kernel_size} (only 2x2x2)")
        self.kernel_size = ks

    def forward(self, x):
        return ops.pool2(x, self.mode)


class MaxPool3d(_Pool2):
    mode = L.POOL_MAX


class AvgPool3d(_Pool2):
    mode = L.POOL_AVG
