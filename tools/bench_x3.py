"""Micro-benchmark of the fp32-storage (1e-3 mode) conv family at BASELINE config-2 shapes: the split-bf16 kernels
(csrc/conv_x3_mfma.hip) against the exact fp32 matrix-core kernels (option x3=0).  TF/s = real FLOP (2 per MAC); the
split-bf16 kernels issue three bf16 MFMAs per product.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
N = int(os.environ.get("KB_N", "4"))
ITERS = int(os.environ.get("KB_ITERS", "10"))
CL = torch.channels_last_3d
F32, NDHWC, AUTO = L.F32, L.NDHWC, L.ALGO_AUTO


def timeit(fn, iters=ITERS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def best(fn, reps=3):
    return min(timeit(fn) for _ in range(reps))


def conv_case(cin, cout, s):
    with mednet_hip.precision("fp32"):
        x = torch.randn(N, cin, s, s, s, device=dev).contiguous(memory_format=CL)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        pk = ops.pack_conv_weight(w, 3, False)
    y = torch.empty(N, cout, s, s, s, device=dev).contiguous(memory_format=CL)
    dx = torch.empty_like(x)
    dw = torch.empty(cout, cin, 3, 3, 3, device=dev)
    ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(N, s, s, s, cin, cout, 3, 0), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * N * s ** 3 * cin * cout * 27
    rows = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, cin, cout, 3, F32, F32, AUTO)
    part = torch.empty(N, max(rows, 1), cout, 2, device=dev)
    fwd = lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, cin, cout, 3, F32, NDHWC,
                                                F32, NDHWC, 0, AUTO, None, st), "fwd")
    fwd_s = lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, cin, cout, 3, F32,
                                                  NDHWC, F32, NDHWC, 0, AUTO, part.data_ptr(), st), "fwd")
    dg = lambda: L.check(lib.mednet_conv3d_fwd(y.data_ptr(), pk.data_ptr(), None, dx.data_ptr(), N, s, s, s, cout, cin, 3, F32, NDHWC,
                                               F32, NDHWC, 1, AUTO, None, st), "dgrad")
    wg = lambda: L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, N, s, s, s, cin, cout, 3, F32, NDHWC,
                                                 F32, NDHWC, AUTO, 0, ws.data_ptr(), ws.numel(), st), "wgrad")
    out = []
    for x3 in (1, 0):
        lib.mednet_set_option(b"x3", x3)
        tf, td, tw = best(fwd), best(dg), best(wg)
        ts = best(fwd_s) if (x3 and rows > 0) else None
        out.append(f"{'split-bf16' if x3 else 'fp32 mfma '}: fwd {tf*1e3:7.1f} us {flop/tf/1e9:6.1f} TF/s"
                   + (f" (+stats {ts*1e3:7.1f} us)" if ts else "")
                   + f" | dgrad {td*1e3:7.1f} us {flop/td/1e9:6.1f} TF/s | wgrad {tw*1e3:7.1f} us {flop/tw/1e9:6.1f} TF/s")
    lib.mednet_set_option(b"x3", 1)
    print(f"conv {cin:3d}->{cout:3d} @{s:3d}^3 N={N} ({flop/1e9:.1f} GFLOP)\n   " + "\n   ".join(out), flush=True)


def convt_case(cin, cout, s):
    with mednet_hip.precision("fp32"):
        x = torch.randn(N, cin, s, s, s, device=dev).contiguous(memory_format=CL)
        w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
        pk = ops.pack_conv_weight(w, 3, True)
    skip = torch.randn(N, cout, 2 * s, 2 * s, 2 * s, device=dev).contiguous(memory_format=CL)
    b = torch.zeros(cout, device=dev)
    y = torch.empty_like(skip)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    ws = torch.empty(lib.mednet_convt3d_wgrad_ws_bytes(N, s, s, s, cin, cout, 0), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * N * s ** 3 * cin * cout * 27
    f = lambda: L.check(lib.mednet_convt3d_fwd(x.data_ptr(), pk.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), N, s, s, s, cin,
                                               cout, F32, F32, AUTO, st), "ctf")
    d = lambda: L.check(lib.mednet_convt3d_dgrad(y.data_ptr(), pk.data_ptr(), dx.data_ptr(), N, s, s, s, cin, cout, F32, F32, AUTO, st), "ctd")
    g = lambda: L.check(lib.mednet_convt3d_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, N, s, s, s, cin, cout, F32, F32, AUTO, 0,
                                                 ws.data_ptr(), ws.numel(), st), "ctw")
    out = []
    for x3 in (1, 0):
        lib.mednet_set_option(b"x3", x3)
        tf, td, tg = best(f), best(d), best(g)
        out.append(f"{'split-bf16' if x3 else 'fp32 mfma '}: fwd {tf*1e3:7.1f} us {flop/tf/1e9:6.1f} TF/s | dgrad {td*1e3:7.1f} us "
                   f"{flop/td/1e9:6.1f} TF/s | wgrad {tg*1e3:7.1f} us {flop/tg/1e9:6.1f} TF/s")
    lib.mednet_set_option(b"x3", 1)
    print(f"convT {cin:3d}->{cout:3d} @{s:3d}^3->{2*s}^3 N={N} ({flop/1e9:.1f} GFLOP)\n   " + "\n   ".join(out), flush=True)


def gn_case(c, s):
    x = torch.randn(N, c, s, s, s, device=dev).contiguous(memory_format=CL).requires_grad_(True)
    g = torch.ones(c, device=dev, requires_grad=True)
    b = torch.zeros(c, device=dev, requires_grad=True)
    with mednet_hip.precision("fp32"):
        z = ops.group_norm_act(x, g, b, 8, 1e-5, L.ACT_ELU)
        dz = torch.randn_like(z)
        tf = best(lambda: ops.group_norm_act(x, g, b, 8, 1e-5, L.ACT_ELU))
        tb = best(lambda: z.backward(dz, retain_graph=True))
    nbytes = N * c * s ** 3 * 4
    print(f"gn+elu fp32 C={c:3d} @{s:3d}^3: fwd (stats + apply) {tf*1e3:7.1f} us ({3*nbytes/tf/1e9:5.2f} TB/s of 3 passes) | bwd {tb*1e3:7.1f} us "
          f"({5*nbytes/tb/1e9:5.2f} TB/s of 5 passes)", flush=True)


def c1_case(cout, s):
    """first layer (one input channel): forward only, with and without the fused GroupNorm sums"""
    with mednet_hip.precision("fp32"):
        x = torch.randn(N, 1, s, s, s, device=dev)
        w = torch.randn(cout, 1, 3, 3, 3, device=dev) * 0.3
        pk = ops.pack_conv_weight(w, 3, False)
    y = torch.empty(N, cout, s, s, s, device=dev).contiguous(memory_format=CL)
    st = torch.cuda.current_stream().cuda_stream
    rows = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, 1, cout, 3, F32, F32, AUTO)
    part = torch.empty(N, max(rows, 1), cout, 2, device=dev)
    fwd = lambda p: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, 1, cout, 3, F32, NDHWC,
                                                  F32, NDHWC, 0, AUTO, p, st), "fwd")
    out = []
    for x3 in (1, 0):
        lib.mednet_set_option(b"x3", x3)
        t0 = best(lambda: fwd(None))
        t1 = best(lambda: fwd(part.data_ptr())) if (x3 and rows > 0) else None
        out.append(f"{'split-bf16' if x3 else 'fp32 mfma '}: fwd {t0*1e3:7.1f} us ({y.numel()*4/t0/1e9:5.2f} TB/s written)"
                   + (f" | with sums {t1*1e3:7.1f} us" if t1 else ""))
    lib.mednet_set_option(b"x3", 1)
    print(f"conv   1-> {cout} @{s}^3 N={N}\n   " + "\n   ".join(out), flush=True)


which = os.environ.get("KB_WHICH", "conv,convt,gn")
if "c1" in which.split(","):
    c1_case(32, 128)
if "conv" in which.split(","):
    for cin, cout, s in ((32, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16), (32, 64, 64)):
        conv_case(cin, cout, s)
if "convt" in which.split(","):
    for cin, cout, s in ((64, 32, 64), (128, 64, 32), (256, 128, 16)):
        convt_case(cin, cout, s)
if "gn" in which.split(","):
    for c, s in ((32, 128), (64, 64)):
        gn_case(c, s)
