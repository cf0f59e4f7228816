"""Forward 3x3x3 convolutions of the general kernel (no fused statistics) at config 2 / config 5 layer shapes with the library named
by MEDNET_LIB_PATH (default: the product): microseconds and PFLOP/s.  Used with tools/probes/one_wg_per_cu_build.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
for n, c, shp in ((4, 64, (64, 64, 64)), (4, 128, (32, 32, 32)), (4, 256, (16, 16, 16)), (2, 64, (96, 160, 160)), (2, 128, (48, 80, 80))):
    d, h, w = shp
    x = torch.randn(n, c, d, h, w, device=dev).bfloat16().contiguous(memory_format=CL)
    wt = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
    pk = ops.pack_conv_weight(wt, 3, False)
    y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), n, d, h, w, c, c, 3, 1, 0, 1, 0, 0, 2, None, st), "fwd")
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 10)
    ms = sorted(best)[1]
    flop = 2.0 * 27 * c * c * n * d * h * w
    print(f"{c:4d}->{c:<4d} @{d}x{h}x{w} N={n}: {ms * 1e3:7.1f} us  {flop / ms / 1e12:6.3f} PFLOP/s")
