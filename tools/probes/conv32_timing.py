"""Phase timing of conv32_mfma_kernel (the 32 -> 32 channel specialisation) with s_memtime stamps; needs the instrumented
build:  make -C torch-mednet_amd/csrc timing  ->  mednet_hip/libmednet_hip_timing.so; run with MEDNET_LIB_PATH set to it.
Stamps of wave 0 of every workgroup: 0 start, 1 weights in registers + first brick committed, and for the workgroup's 5th
brick: 3 barrier passed, 5 the 216 MFMAs issued, 6 epilogue done; 15 end of the workgroup.  Also prints plain event timings of
the specialised and the general kernel (option conv32=0) with and without the fused statistics."""
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
if "timing" in os.environ.get("MEDNET_LIB_PATH", ""):
    dbg = torch.zeros(256, 16, dtype=torch.int64, device=dev)
    p = dbg.data_ptr()
    lib.mednet_set_option(b"conv_dbg_lo", (p & 0xFFFFFFFF) - (1 << 32) if (p & 0x80000000) else (p & 0xFFFFFFFF))
    lib.mednet_set_option(b"conv_dbg_hi", p >> 32)
    for _ in range(2):
        dbg.zero_()
        launch(None)
        torch.cuda.synchronize()
    lib.mednet_set_option(b"conv_dbg_lo", 0)
    lib.mednet_set_option(b"conv_dbg_hi", 0)
    t = dbg.cpu().double()
    nitem = N * (s // 4) * (s // 8) * (s // 16) / 256
    print(f"{nitem:.0f} bricks per workgroup; ticks of s_memtime (100 MHz):")
    for nm, a_, b_ in (("prologue (weights, first brick)", 0, 1), ("5th brick: tap loop", 3, 5), ("5th brick: epilogue", 5, 6),
                       ("whole workgroup", 0, 15)):
        d = t[:, b_] - t[:, a_]
        print(f"   {nm:34s} mean {d.mean():9.1f}  p10 {d.quantile(0.1):9.1f}  p90 {d.quantile(0.9):9.1f}")
    print(f"   per brick {((t[:, 15] - t[:, 1]).mean() / nitem):.1f} ticks")
