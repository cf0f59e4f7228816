"""Stand-alone event timings of conv32_mfma_kernel (the 32 -> 32 channel specialisation) against the general kernel (option
conv32=0): forward with and without the fused statistics, and the data-gradient variants.  (The s_memtime phase stamps this
tool printed until round 4 needed an instrumented build of conv_mfma.hip; that scaffolding left the product source in round 5,
the numbers it produced are in profiles/r02_* / r04_ab.md.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
N, s = int(os.environ.get("CT_N", "4")), int(os.environ.get("CT_S", "128"))
x = torch.randn(N, 32, s, s, s, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)
y = torch.empty(N, 32, s, s, s, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
st = torch.cuda.current_stream().cuda_stream


def launch(part):
    L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, 32, 32, 3, 1, 0, 1, 0, 0, 2,
                                  part.data_ptr() if part is not None else None, st), "fwd")


for special in (1, 0):
    lib.mednet_set_option(b"conv32", special)
    chunks = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, 32, 32, 3, 1, 1, 2)
    part = torch.empty(N, chunks, 32, 2, device=dev)
    for with_stats in (0, 1):
        for _ in range(3):
            launch(part if with_stats else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20):
            launch(part if with_stats else None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"conv32={special} stats={with_stats}: {ms * 1e3:7.1f} us  {2.0 * N * s ** 3 * 32 * 32 * 27 / ms / 1e9:7.1f} TFLOP/s")
# the data-gradient variants: + summed second gradient, + first pass of a GroupNorm backward (ELU), both
add = torch.randn_like(x)
gy = torch.randn_like(x)
coef = torch.randn(N, 32, 2, device=dev)
dx = torch.empty_like(x)
for special in (1, 0):
    lib.mednet_set_option(b"conv32", special)
    rows = lib.mednet_conv3d_dgrad_gn_rows(N, s, s, s, 32, 32, 2)
    part = torch.empty(N, rows, 32, 2, device=dev)
    cases = {
        "dgrad_add": lambda: lib.mednet_conv3d_dgrad_add(x.data_ptr(), pk.data_ptr(), add.data_ptr(), dx.data_ptr(), N, s, s, s, 32, 32, 2, 1, st),
        "dgrad_gn": lambda: lib.mednet_conv3d_dgrad_gn(x.data_ptr(), pk.data_ptr(), None, dx.data_ptr(), gy.data_ptr(), coef.data_ptr(), 3,
                                                       part.data_ptr(), N, s, s, s, 32, 32, 2, 1, st),
        "dgrad_gn+add": lambda: lib.mednet_conv3d_dgrad_gn(x.data_ptr(), pk.data_ptr(), add.data_ptr(), dx.data_ptr(), gy.data_ptr(),
                                                           coef.data_ptr(), 3, part.data_ptr(), N, s, s, s, 32, 32, 2, 1, st),
    }
    for name, fn in cases.items():
        for _ in range(3):
            L.check(fn(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"conv32={special} {name:14s}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
lib.mednet_set_option(b"conv32", 1)
