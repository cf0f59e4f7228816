"""conv32z_mfma_kernel (z-columns, option conv32z=1) against conv32_mfma_kernel (bricks) at the benchmarked shape: stand-alone event
timings with and without the fused statistics, on random operands and on post-ELU-like ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
N, s = int(os.environ.get("CT_N", "4")), int(os.environ.get("CT_S", "128"))
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)
st = torch.cuda.current_stream().cuda_stream
for data in ("randn", "elu"):
    x = torch.randn(N, 32, s, s, s, device=dev)
    if data == "elu":
        x = torch.nn.functional.elu(x)
    x = x.bfloat16().contiguous(memory_format=torch.channels_last_3d)
    y = torch.empty_like(x)
    part = torch.empty(N, 1024, 32, 2, device=dev)
    for rnd in range(2):
        for z in (0, 1):
            lib.mednet_set_option(b"conv32z", z)
            for with_stats in (0, 1):
                fn = lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, 32, 32, 3, 1, 0, 1, 0, 0, 2,
                                                           part.data_ptr() if with_stats else None, st), "fwd")
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                print(f"{data:5s} conv32z={z} stats={with_stats}: {ms * 1e3:7.1f} us  {2.0 * N * s ** 3 * 32 * 32 * 27 / ms / 1e9:7.1f} TFLOP/s", flush=True)
lib.mednet_set_option(b"conv32z", 0)
