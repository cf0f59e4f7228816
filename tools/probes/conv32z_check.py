"""Bit-identity check of conv32z_mfma_kernel (tools/probes/conv32z_kernel.inc, built by tools/probes/conv32z_build.sh into
libmednet_hip_conv32z.so) against conv32_mfma_kernel; run with MEDNET_LIB_PATH pointing at that library:
    MEDNET_LIB_PATH=torch-mednet_amd/mednet_hip/libmednet_hip_conv32z.so python -m pytest tools/probes/conv32z_check.py -q"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import pytest
import torch
import mednet_hip
from mednet_hip import _lib as L
from mednet_hip import ops
from gpu_util import DEV


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("n,shape", [(2, (32, 60, 62)), (1, (76, 56, 64)), (4, (64, 64, 64)), (3, (33, 40, 100)), (1, (128, 128, 128))])
def test_conv32z_columns_are_bit_identical_to_the_brick_kernel(mode, n, shape):
    """conv32z_mfma_kernel (round 5, option conv32z): the 32 -> 32 forward as z-columns through an LDS-DMA ring -- one N-tile and one
    accumulator per wave, the epilogue of plane z inside the tap loop of plane z + 1.  Same accumulation order as the brick kernel
    (K chunk outer, tap inner), so the outputs must be IDENTICAL bit for bit (ragged rows / planes / x ends, slabs that do not
    divide the depth), every output element written (the buffer starts as NaN), and the fused GroupNorm partials must add up to
    the same totals (other rows)."""
    lib = L.lib()
    dt = torch.bfloat16 if mode == "bf16" else torch.float16
    dcode = L.dt(torch.empty(0, dtype=dt))
    CL = torch.channels_last_3d
    g = torch.Generator(device=DEV).manual_seed(23)
    x = torch.randn(n, 32, *shape, device=DEV, generator=g).to(dt).contiguous(memory_format=CL)
    w = torch.randn(32, 32, 3, 3, 3, device=DEV, generator=g) * 0.05
    with mednet_hip.precision(mode):
        pk = ops.pack_conv_weight(w, 3, False)
    d, h, wd = shape
    st = torch.cuda.current_stream().cuda_stream
    rows = lib.mednet_conv3d_fused_stats_chunks(n, d, h, wd, 32, 32, 3, dcode, dcode, 2)
    assert rows == 1024  # (the specialisation's accumulate-mode layout: both kernels write it)
    res = {}
    try:
        for z in (1, 0):
            lib.mednet_set_option(b"conv32z", z)
            for stats in (False, True):
                y = torch.full_like(x, float("nan"))
                part = torch.full((n, rows, 32, 2), float("nan"), device=DEV) if stats else None
                L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), n, d, h, wd, 32, 32, 3, dcode, L.NDHWC, dcode,
                                              L.NDHWC, 0, 2, part.data_ptr() if stats else None, st), "conv3d_fwd")
                torch.cuda.synchronize()
                res[(z, stats)] = (y, part.double().sum(1) if stats else None)
    finally:
        lib.mednet_set_option(b"conv32z", 0)
    for stats in (False, True):
        ya, yb = res[(1, stats)][0], res[(0, stats)][0]
        assert bool(torch.isfinite(ya.float()).all()), "conv32z left output elements unwritten"
        assert torch.equal(ya, yb), f"y differs from the brick kernel (stats={stats}): {int((ya != yb).sum())} elements"
    t1, t0 = res[(1, True)][1], res[(0, True)][1]
    nv = float(np.prod(shape))
    q = t0[:, 0::2, 1]
    assert bool(torch.isfinite(t1).all())
    assert torch.all((t1[:, 0::2, 0] - t0[:, 0::2, 0]).abs() <= 2e-5 * (q * nv).sqrt() + 1e-3)
    assert torch.all((t1[:, 0::2, 1] - q).abs() <= 2e-5 * q)
    assert torch.all(t1[:, 1::2] == 0)


