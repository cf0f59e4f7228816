"""Micro-benchmark of the hot kernels at BASELINE config-2 shapes (run on the GPU box).  Prints TFLOP/s or TB/s."""
import os, sys, time
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


def conv_case(cin, cout, s):
    x = torch.randn(N, cin, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    pk = ops.pack_conv_weight(w, 3, False)
    y = torch.empty(N, cout, s, s, s, device=dev, dtype=torch.bfloat16).contiguous(memory_format=CL)
    dw = torch.empty(cout, cin, 3, 3, 3, device=dev)
    ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(N, s, s, s, cin, cout, 3, 0), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * N * s ** 3 * cin * cout * 27

    def fwd():
        L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, cin, cout, 3, 1, 0, 1, 0, 0, 2, None, st), "fwd")

    chunks = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, cin, cout, 3, 1, 1, 2)
    part = torch.empty(N, max(chunks, 1), cout, 2, device=dev)

    def fwd_stats():
        L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, cin, cout, 3, 1, 0, 1, 0, 0, 2, part.data_ptr(), st), "fwd")

    def wg():
        L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, N, s, s, s, cin, cout, 3, 1, 0, 1, 0, 2, 0,
                                        ws.data_ptr(), ws.numel(), st), "wgrad")

    res = []
    variants = [0]
    for rnd in range(3):
        for pv in variants:
            res.append((pv, timeit(fwd)))
    ts = min(timeit(fwd_stats) for _ in range(3))
    print(f"   fwd + fused GroupNorm partials: {ts*1e3:6.1f} us {flop/ts/1e9:6.1f} TF/s")
    tw = timeit(wg)
    txt = " | ".join(f" {min(t for p, t in res if p == pv)*1e3:6.1f} us {flop/min(t for p, t in res if p == pv)/1e9:6.1f} TF/s" for pv in variants)
    print(f"conv {cin:3d}->{cout:3d} @{s:3d}^3 N={N}: fwd {txt} | wgrad {tw*1e3:7.1f} us {flop/tw/1e9:7.1f} TF/s", flush=True)


def gn_case(c, s):
    x = torch.randn(N, c, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL).requires_grad_(True)
    g = torch.ones(c, device=dev, requires_grad=True)
    b = torch.zeros(c, device=dev, requires_grad=True)
    with mednet_hip.precision("bf16"):
        z = ops.group_norm_act(x, g, b, 8, 1e-5, L.ACT_ELU)
        dz = torch.randn_like(z)
        tf = timeit(lambda: ops.group_norm_act(x, g, b, 8, 1e-5, L.ACT_ELU))
        tb = timeit(lambda: z.backward(dz, retain_graph=True))
    nbytes = N * c * s ** 3 * 2
    print(f"gn+elu C={c:3d} @{s:3d}^3: fwd {tf*1e3:7.1f} us ({3*nbytes/tf/1e9:5.2f} TB/s of 3 passes) | bwd {tb*1e3:7.1f} us ({7*nbytes/tb/1e9:5.2f} TB/s of 7 passes)", flush=True)


which = os.environ.get("KB_WHICH", "conv,gn")
if "conv" in which:
    for cin, cout, s in ((32, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16), (32, 64, 64)):
        conv_case(cin, cout, s)
if "gn" in which:
    for c, s in ((32, 128), (64, 64)):
        gn_case(c, s)


def convt_case(cin, cout, s):
    """ConvTranspose3d(k3,s2,p1,op1) cin -> cout from s^3 to (2s)^3: forward (+skip), data gradient, weight gradient."""
    x = torch.randn(N, cin, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
    skip = torch.randn(N, cout, 2 * s, 2 * s, 2 * s, device=dev).bfloat16().contiguous(memory_format=CL)
    w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    pk = ops.pack_conv_weight(w, 3, True)
    y = torch.empty_like(skip)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    ws = torch.empty(lib.mednet_convt3d_wgrad_ws_bytes(N, s, s, s, cin, cout, 0), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * N * s ** 3 * cin * cout * 27
    f = lambda: L.check(lib.mednet_convt3d_fwd(x.data_ptr(), pk.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), N, s, s, s, cin, cout, 1, 1, 2, st), "ctf")
    d = lambda: L.check(lib.mednet_convt3d_dgrad(y.data_ptr(), pk.data_ptr(), dx.data_ptr(), N, s, s, s, cin, cout, 1, 1, 2, st), "ctd")
    g = lambda: L.check(lib.mednet_convt3d_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, N, s, s, s, cin, cout, 1, 1, 2, 0, ws.data_ptr(), ws.numel(), st), "ctw")
    tf, td = min(timeit(f) for _ in range(3)), min(timeit(d) for _ in range(3))
    tg = min(timeit(g) for _ in range(3))
    print(f"convT {cin:3d}->{cout:3d} @{s:3d}^3->{2*s}^3 N={N}: fwd {tf*1e3:6.1f} us {flop/tf/1e9:6.1f} TF/s | dgrad {td*1e3:6.1f} us {flop/td/1e9:6.1f} TF/s"
          f" | wgrad {tg*1e3:6.1f} us {flop/tg/1e9:6.1f} TF/s", flush=True)


if "convt" in which:
    for cin, cout, s in ((64, 32, 64), (128, 64, 32), (256, 128, 16)):
        convt_case(cin, cout, s)
