"""What would fusing GroupNorm-apply + ELU into the consumer convolution's staging cost?  

The fused form never writes z = ELU(ca * y + cb): the convolution that consumes z reads the conv output y of the layer in front
and transforms every staged 16-byte piece while it commits it to LDS.  That saves the stand-alone apply pass (read y, write z)
and costs VALU work inside the matrix-core kernel -- per staged element one fma, one v_exp_f32 and a select, for EVERY channel
block that stages the piece and for the halo overlap, where the apply pass touches each element once.

This probe measures both sides on the device: `conv_mfma_kernel<1, false, 1>` (the general kernel with the transform in its
commit phase, option conv_xform_probe) against the same launch without it, next to the apply pass at the same shape.
One line per shape; `fusion pays` only if (conv + apply) > conv_with_transform (the weight gradient, which would have to
recompute z the same way, is not even counted).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the probe instantiation lives in its own build: make -C torch-mednet_amd/csrc probe
os.environ.setdefault("MEDNET_LIB_PATH", os.path.join(ROOT, "torch-mednet_amd", "mednet_hip", "libmednet_hip_probe.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch  # noqa: E402

import mednet_hip  # noqa: E402
from mednet_hip import _lib as L, ops  # noqa: E402

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
N = int(os.environ.get("XP_N", "4"))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def case(cin, cout, s):
    mednet_hip.set_precision("bf16")
    x = (torch.randn(N, cin, s, s, s, device=dev) * 1.0).bfloat16().contiguous(memory_format=CL)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    pk = ops.pack_conv_weight(w, 3, False)
    y = torch.empty(N, cout, s, s, s, device=dev, dtype=torch.bfloat16).contiguous(memory_format=CL)
    z = torch.empty_like(x, memory_format=CL)
    coef = torch.stack((torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev) * 0.3), dim=-1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    lib.mednet_set_option(b"conv32", 0)  # the general kernel also at 32 -> 32 (the specialisation has no registers for this)

    def conv():
        L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, cin, cout, 3, 1, 0, 1, 0, 0, 2, None, st), "fwd")

    def apply_pass():
        L.check(lib.mednet_gn_act_fwd(x.data_ptr(), coef.data_ptr(), None, z.data_ptr(), N, s ** 3, cin, L.ACT_ELU, L.BF16, L.BF16, st), "gn_act_fwd")

    res = {"plain": 1e9, "xform": 1e9, "apply": 1e9}
    p = coef.data_ptr()
    for _ in range(3):  # interleaved rounds (the parts' clocks drift)
        lib.mednet_set_option(b"conv_xform_probe", 0)
        res["plain"] = min(res["plain"], timeit(conv))
        lib.mednet_set_option(b"conv_xf_hi", (p >> 32) - (1 << 32) if (p >> 32) >= (1 << 31) else (p >> 32))
        lo = p & 0xFFFFFFFF
        lib.mednet_set_option(b"conv_xf_lo", lo - (1 << 32) if lo >= (1 << 31) else lo)
        lib.mednet_set_option(b"conv_xform_probe", 1)
        res["xform"] = min(res["xform"], timeit(conv))
        lib.mednet_set_option(b"conv_xform_probe", 0)
        res["apply"] = min(res["apply"], timeit(apply_pass))
    lib.mednet_set_option(b"conv32", 1)
    # correctness of the probe itself: conv(ELU(ca * y + cb)) through the two-pass path
    lib.mednet_set_option(b"conv_xform_probe", 0)
    apply_pass()
    y2 = torch.empty_like(y, memory_format=CL)
    lib.mednet_set_option(b"conv32", 0)
    L.check(lib.mednet_conv3d_fwd(z.data_ptr(), pk.data_ptr(), None, y2.data_ptr(), N, s, s, s, cin, cout, 3, 1, 0, 1, 0, 0, 2, None, st), "fwd")
    lib.mednet_set_option(b"conv_xform_probe", 1)
    conv()
    lib.mednet_set_option(b"conv_xform_probe", 0)
    lib.mednet_set_option(b"conv32", 1)
    torch.cuda.synchronize()
    err = float((y.float() - y2.float()).norm() / y2.float().norm())
    d = res["xform"] - res["plain"]
    print(f"{cin:4d}->{cout:4d} @{s:3d}^3 N={N}: conv {res['plain']:7.1f} us | conv with the transform in commit() {res['xform']:7.1f} us "
          f"(+{d:6.1f}) | apply pass {res['apply']:6.1f} us | fusion {'pays' if d < res['apply'] else 'loses'} {res['apply'] - d:+7.1f} us "
          f"per layer (before the weight gradient's share) | probe vs two-pass rel-L2 {err:.1e}", flush=True)


if __name__ == "__main__":
    for (ci, co, s) in [(32, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16), (32, 64, 64)]:
        case(ci, co, s)
