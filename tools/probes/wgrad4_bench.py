"""Stand-alone timings of the weight-gradient kernels at config 2's layer shapes: wgrad_mfma2_kernel (8 waves, a CU whole) against
wgrad_mfma4_kernel (round 5: 4 waves of < 256 registers, z-columns through an LDS-DMA ring; option wgrad_v4), each with the library's
plan (one workgroup per CU) and with the trainer's side-stream plan (128 workgroups); then the co-residency question itself: a
bandwidth-bound GroupNorm-backward apply pass on a second stream BESIDE each weight gradient (all CUs), together vs one after the
other.  Random bf16 operands (dense: the slow end of the clock range)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L

dev = "cuda:0"
lib = L.lib()
N = int(os.environ.get("WG_N", "4"))
CL = torch.channels_last_3d


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def wgrad_fn(x, dy, dw, cin, cout, s, wgs, stream):
    ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(N, s, s, s, cin, cout, 3, wgs), dtype=torch.uint8, device=dev)
    return lambda: L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, s, s, s, cin, cout, 3, L.BF16, L.NDHWC,
                                                   L.BF16, L.NDHWC, L.ALGO_MFMA, wgs, ws.data_ptr(), ws.numel(), stream), "wgrad")


shapes = [(32, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16)]
if os.environ.get("WG_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["WG_SHAPES"].split(",")]
for cin, cout, s in shapes:
    x = torch.randn(N, cin, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
    dy = torch.randn(N, cout, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
    dw = torch.empty(cout, cin, 3, 3, 3, device=dev)
    flop = 2.0 * N * s ** 3 * cin * cout * 27
    st = torch.cuda.current_stream().cuda_stream
    line = []
    ref = None
    for v4 in (0, 1):
        lib.mednet_set_option(b"wgrad_v4", v4)
        for wgs in (0, 128):
            us = min(timeit(wgrad_fn(x, dy, dw, cin, cout, s, wgs, st)) for _ in range(3))
            line.append(f"v{4 if v4 else 2} wgs={wgs or 256}: {us:7.1f} us {flop / us / 1e6:6.0f} TF/s")
            if wgs == 0:
                if ref is None:
                    ref = dw.clone()
                else:
                    err = float((dw - ref).norm() / ref.norm())
                    line.append(f"rel diff {err:.1e}")
    print(f"wgrad {cin:3d}->{cout:3d} @{s:3d}^3 N={N}: " + " | ".join(line), flush=True)
    # ---- beside a bandwidth-bound pass on another stream (GroupNorm backward apply: reads dz, y; writes dy)
    if s >= 64:
        c = cout
        dz = torch.randn(N, c, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
        y = torch.randn(N, c, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
        out = torch.empty_like(dz)
        coef = torch.randn(N, c, 2, device=dev)
        stats = torch.rand(N, 8, 2, device=dev) + 0.5
        gamma = torch.randn(c, device=dev)
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        spatial = s ** 3
        gws = torch.empty(lib.mednet_gn_ws_bytes(N, c, spatial), dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream()
        rows = 64
        partial = torch.randn(N, rows, c, 2, device=dev) * 1e-3
        def apply_on(stream):  # (the form the step runs: the sums come from the data gradient's epilogue; reduce + finalize + apply)
            return lambda: L.check(lib.mednet_gn_act_bwd_fused(dz.data_ptr(), y.data_ptr(), coef.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                                               partial.data_ptr(), rows, out.data_ptr(), dg.data_ptr(), db.data_ptr(), N, spatial, c, 8,
                                                               L.ACT_ELU, L.ACT_NONE, L.BF16, gws.data_ptr(), gws.numel(), stream), "gn_act_bwd_fused")
        t_apply = min(timeit(apply_on(st)) for _ in range(3))
        for v4 in (0, 1):
            lib.mednet_set_option(b"wgrad_v4", v4)
            for wgs in (0, 128):
                wg_side = wgrad_fn(x, dy, dw, cin, cout, s, wgs, side.cuda_stream)
                ap_main = apply_on(st)
                t_w = min(timeit(wgrad_fn(x, dy, dw, cin, cout, s, wgs, st)) for _ in range(2))
                def both():
                    side.wait_stream(torch.cuda.current_stream())
                    wg_side()
                    ap_main()
                    torch.cuda.current_stream().wait_stream(side)
                t_b = min(timeit(both) for _ in range(3))
                print(f"   v{4 if v4 else 2} wgs={wgs or 256}: wgrad alone {t_w:7.1f} + apply pass alone {t_apply:6.1f} = {t_w + t_apply:7.1f} us; "
                      f"on two streams together {t_b:7.1f} us ({(t_w + t_apply) / t_b:.2f}x)", flush=True)
lib.mednet_set_option(b"wgrad_v4", 1)
