"""Stand-alone time of the fused first-layer backward kernel (mednet_conv3d_wgrad_c1_gn) against the two kernels it replaces
(the apply pass of mednet_gn_act_bwd_fused + mednet_conv3d_wgrad, Cin = 1) at config 2's shape; timing only (random coefficients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L

dev = torch.device("cuda", 0)
lib = L.lib()
n, d, h, w, c = 4, 128, 128, 128, 32
dt = torch.bfloat16
x = torch.randn(n, d, h, w, device=dev)
dz = torch.randn(n, d, h, w, c, device=dev).to(dt)
y = torch.randn(n, d, h, w, c, device=dev).to(dt)
dy = torch.empty_like(y)
coef = torch.randn(n, c, 2, device=dev)
bcoef = torch.randn(n, c, 3, device=dev)
dw = torch.empty(c, 27, device=dev)
ws = torch.empty(lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, 1, c, 3, 0), dtype=torch.uint8, device=dev)
s = L.stream()


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    best = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(best)[1]


fused = lambda: L.check(lib.mednet_conv3d_wgrad_c1_gn(x.data_ptr(), dz.data_ptr(), y.data_ptr(), coef.data_ptr(), bcoef.data_ptr(),
                                                      dw.data_ptr(), n, d, h, w, c, L.ACT_ELU, L.F32, L.BF16, ws.data_ptr(), ws.numel(), s), "c1gn")
plain = lambda: L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, n, d, h, w, 1, c, 3, L.F32, L.NDHWC, L.BF16,
                                                L.NDHWC, L.ALGO_AUTO, 0, ws.data_ptr(), ws.numel(), s), "wgrad_c1")
print(f"wgrad_c1_gn (dz, y -> dW; 1.07 GB): {timed(fused):.1f} us   wgrad_c1 alone (dy -> dW; 0.54 GB): {timed(plain):.1f} us"
      f"   (the apply pass it replaces: 283 us alone, profiles/r05_bf16_single_stream_timeline.txt)")
