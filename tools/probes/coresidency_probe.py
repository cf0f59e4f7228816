"""Do small kernels on the main stream make progress while the persistent weight-gradient kernel (one 512-thread workgroup per
CU, 100 KB LDS) runs on the side stream?  Times tiny launches (HIP events on the main stream) issued ~60 us after the
weight gradient started, for several tiny-kernel shapes.  Round-2 finding that prompted it: gn_bwd_finalize_kernel (32
workgroups x 64 threads) took 340 us in the step trace whenever it overlapped wgrad_mfma2_kernel and ended when that ended."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
N, c, s = 4, 32, 128
CL = torch.channels_last_3d
x = torch.randn(N, c, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
dy = torch.randn(N, c, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
dw = torch.empty(c, c, 3, 3, 3, device=dev)
ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(N, s, s, s, c, c, 3, 0), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
small = torch.zeros(64, device=dev)
mid = torch.zeros(1 << 20, device=dev)
dbl = torch.zeros(2048, device=dev, dtype=torch.float64)


def wgrad():
    with torch.cuda.stream(side):
        L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, s, s, s, c, c, 3, 1, 0, 1, 0, 0, 0,
                                        ws.data_ptr(), ws.numel(), side.cuda_stream), "wgrad")


def probe(name, fn, with_wgrad):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_wgrad:
        with torch.cuda.stream(side):
            w0.record()
        wgrad()
        with torch.cuda.stream(side):
            w1.record()
    torch.cuda._sleep(150000)  # ~60-80 us at 2 GHz: the weight gradient is resident by now
    e[0].record()
    for i in range(12):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    d = [e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(12)]
    extra = f"   wgrad {w0.elapsed_time(w1) * 1e3:.0f} us" if with_wgrad else ""
    print(f"{name:34s} wgrad={int(with_wgrad)}  us per launch: " + " ".join(f"{v:6.1f}" for v in d) + extra)


part = torch.randn(N, 2048, c, 2, device=dev)
gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
stats, coef = torch.empty(N, 8, 2, device=dev), torch.empty(N, c, 2, device=dev)
gws = torch.empty(lib.mednet_gn_ws_bytes(N, c, s ** 3), dtype=torch.uint8, device=dev)


def gn_finalize():  # 32 workgroups x 256 threads: strided row sums, wave_sum (ds_bpermute shuffles) + an LDS exchange
    L.check(lib.mednet_gn_finalize(part.data_ptr(), 2048, gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), coef.data_ptr(),
                                   N, s ** 3, c, 8, 1e-5, gws.data_ptr(), gws.numel(), main.cuda_stream), "gn_finalize")


for rep in range(2):
    for wg in (False, True):
        probe("64-element add_ (1 wave)", lambda: small.add_(1.0), wg)
        probe("1M-element add_ (4096 waves)", lambda: mid.add_(1.0), wg)
        probe("2048 fp64 mul (32 waves)", lambda: dbl.mul_(1.0000001), wg)
        probe("fp64 div", lambda: dbl.div_(1.0000001), wg)
        probe("torch.sum of 2048 (shuffles + LDS)", lambda: dbl.sum(), wg)
        probe("mednet_gn_finalize (shuffles+LDS)", gn_finalize, wg)
