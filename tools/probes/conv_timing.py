"""Phase timing of conv_mfma_kernel<1> with s_memtime stamps (needs the instrumented build:
   make -C torch-mednet_amd/csrc timing  ->  mednet_hip/libmednet_hip_timing.so; run with MEDNET_LIB_PATH set to it).
Stamps of wave 0 of every workgroup: 0 start, 1 plan done, 2 first loads issued, per chunk k (3+4k: barrier passed,
4+4k: LDS commit done (= global data arrived), 5+4k: second barrier + next prefetch issued, 6+4k: 27 taps issued),
11 epilogue barrier, 12 accumulators in LDS + barrier, 13 rows stored, 14 all stores acknowledged."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
N, cin, cout, s = 4, int(os.environ.get("CT_CIN", "32")), int(os.environ.get("CT_COUT", "32")), int(os.environ.get("CT_S", "128"))
x = torch.randn(N, cin, s, s, s, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)
y = torch.empty(N, cout, s, s, s, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
chunks = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, cin, cout, 3, 1, 1, 2)
part = torch.empty(N, chunks, cout, 2, device=dev)
nwg = ((N * chunks + 7) // 8) * 8 * ((cout + 31) // 32)  # (>= the grid of either launch form)
dbg = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for with_stats in (0, 1):
    for rep in range(2):
        dbg.zero_()
        p = dbg.data_ptr()
        lib.mednet_set_option(b"conv_dbg_lo", (p & 0xFFFFFFFF) - (1 << 32) if (p & 0x80000000) else (p & 0xFFFFFFFF))
        lib.mednet_set_option(b"conv_dbg_hi", p >> 32)
        L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, cin, cout, 3, 1, 0, 1, 0, 0, 2,
                                      part.data_ptr() if with_stats else None, st), "fwd")
        torch.cuda.synchronize()
    t = dbg.cpu().double()
    t = t[t[:, 0] > 0]
    life = (t[:, 15] - t[:, 0])
    span = (t[:, 15].max() - t[:, 0].min()).item()
    print(f"stats={with_stats}: {t.shape[0]} workgroups, kernel span {span:.0f} ticks, mean workgroup life {life.mean():.0f} ticks "
          f"(phases 3..14: the workgroup's 9th item, or its only one)")
    names = ["plan", "issue loads", "barrier", "wait data + commit k0", "barrier + prefetch", "taps k0", "barrier", "commit k1",
             "barrier", "taps k1", "-", "epilogue barrier", "acc->LDS + barrier", "rows -> global (+stats dot2)", "stats reduce + store ack"]
    for i in range(14):
        a, b = i, i + 1
        if i == 10:
            continue
        if i == 9:
            b = 11
            a = 10
        d = (t[:, b] - t[:, a])
        print(f"   {a:2d}->{b:2d} {names[i] if i < 9 else names[i + 1]:32s} mean {d.mean():8.0f}  p10 {d.quantile(0.1):8.0f}  p90 {d.quantile(0.9):8.0f}")
lib.mednet_set_option(b"conv_dbg_lo", 0)
lib.mednet_set_option(b"conv_dbg_hi", 0)
