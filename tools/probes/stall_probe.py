"""Which small kernel stalls right after the data-gradient conv while the weight gradient (launched at the same time on the
side stream) still has to run?  Scenario of one backward layer at 32->32 @128^3, N=4:
    side: wgrad (waits for `dy`)            main: dgrad_gn -> [candidate] -> e1
The candidate's time (HIP events) is printed next to the weight gradient's.  See DESIGN.md (round 2, co-scheduling)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
N, c, s = 4, 32, 128
CL = torch.channels_last_3d
mk = lambda: torch.randn(N, c, s, s, s, device=dev).bfloat16().contiguous(memory_format=CL)
x, dy, y_prev, dx, dyo = mk(), mk(), mk(), mk(), mk()
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)
dw = torch.empty(c, c, 3, 3, 3, device=dev)
ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(N, s, s, s, c, c, 3, 0), dtype=torch.uint8, device=dev)
rows = lib.mednet_conv3d_dgrad_gn_rows(N, s, s, s, c, c, 0)
part = torch.empty(N, rows, c, 2, device=dev)
coef = torch.randn(N, c, 2, device=dev)
stats = torch.rand(N, 8, 2, device=dev) + 0.5
gamma = torch.ones(c, device=dev)
dgam, dbet = torch.empty(c, device=dev), torch.empty(c, device=dev)
gws = torch.empty(lib.mednet_gn_ws_bytes(N, c, s ** 3), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
small = torch.zeros(64, device=dev)
dbl = torch.ones(2048, device=dev, dtype=torch.float64)


def gn_bwd_fused():
    L.check(lib.mednet_gn_act_bwd_fused(dx.data_ptr(), y_prev.data_ptr(), coef.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                        part.data_ptr(), rows, dyo.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), N, s ** 3, c, 8,
                                        L.ACT_ELU, L.ACT_NONE, L.BF16, gws.data_ptr(), gws.numel(), main.cuda_stream), "gn_bwd_fused")


def scenario(name, cand, wgrad_after_dgrad=False):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def launch_wgrad():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            w0.record()
            L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, s, s, s, c, c, 3, 1, 0, 1, 0, 0, 0,
                                            ws.data_ptr(), ws.numel(), side.cuda_stream), "wgrad")
            w1.record()

    small.add_(1.0)  # (stands for the apply pass that produced dy)
    ev[0].record()
    if not wgrad_after_dgrad:
        launch_wgrad()
    L.check(lib.mednet_conv3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), None, dx.data_ptr(), y_prev.data_ptr(), coef.data_ptr(),
                                       L.ACT_ELU, part.data_ptr(), N, s, s, s, c, c, 0, main.cuda_stream), "dgrad_gn")
    ev[1].record()
    if wgrad_after_dgrad:
        launch_wgrad()
    cand()
    ev[2].record()
    torch.cuda.synchronize()
    print(f"{name:46s} dgrad {ev[0].elapsed_time(ev[1]) * 1e3:6.0f} us   candidate {ev[1].elapsed_time(ev[2]) * 1e3:6.0f} us   "
          f"wgrad (side, from its start) {w0.elapsed_time(w1) * 1e3:6.0f} us")


def only(mask_keep):  # run only the sub-kernels of gn_act_bwd_fused whose bit is in mask_keep (1 dux, 2 finalize, 4 params, 8 apply)
    def f():
        lib.mednet_set_option(b"gn_bwd_skip", 15 & ~mask_keep)
        gn_bwd_fused()
        lib.mednet_set_option(b"gn_bwd_skip", 0)
    return f


for rep in range(2):
    for keep, nm in ((3, "dux+finalize"), (5, "dux+params"), (7, "dux+finalize+params"), (9, "dux+apply"), (10, "finalize+apply")):
        scenario(nm + " [wgrad queued behind dgrad]", only(keep), True)
    scenario("dux, tiny add_, finalize [wgrad queued behind dgrad]", lambda: (only(1)(), small.add_(1.0), only(2)()), True)
    scenario("dux, finalize as two API calls [wgrad queued..]", lambda: (only(1)(), only(2)()), True)
for rep in range(0):
    for after in (False, True):
        tag = " [wgrad queued behind dgrad]" if after else ""
        scenario("tiny add_" + tag, lambda: small.add_(1.0), after)
        scenario("fp64 div" + tag, lambda: dbl.div_(1.0000001), after)
        scenario("gn_act_bwd_fused (dux,finalize,params,apply)" + tag, gn_bwd_fused, after)
        scenario("tiny add_, then gn_act_bwd_fused" + tag, lambda: (small.add_(1.0), gn_bwd_fused()), after)
