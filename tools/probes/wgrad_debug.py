"""Debug: structure of the MFMA wgrad result for one-hot dy (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops

dev = "cuda:0"
n, c, D, H, W = 1, 32, 2, 8, 16
lib = L.lib()

def wgrad(x, dy, algo):
    x = x.to(dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    dy = dy.to(dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    dw = torch.zeros(c, c, 3, 3, 3, device=dev)
    ws = torch.zeros(lib.mednet_conv3d_wgrad_ws_bytes(n, D, H, W, c, c, 3), dtype=torch.uint8, device=dev)
    L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, n, D, H, W, c, c, 3, 1, 0, 1, 0, algo,
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "wgrad")
    torch.cuda.synchronize()
    return dw.cpu()

zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
for name, field in (("ci", None), ("x", xx), ("y", yy), ("z", zz)):
    x = torch.zeros(n, c, D, H, W)
    if field is None:
        x += torch.arange(c).view(1, c, 1, 1, 1).float()
    else:
        x += field.view(1, 1, D, H, W).float()
    for (z0, y0, x0, co0) in ((1, 3, 5, 7), (0, 4, 9, 20)):
        dy = torch.zeros(n, c, D, H, W)
        dy[0, co0, z0, y0, x0] = 1.0
        ref = wgrad(x, dy, 1)
        got = wgrad(x, dy, 2)
        ok = torch.allclose(ref, got)
        print(f"field={name} onehot=({z0},{y0},{x0},co{co0}) match={ok}")
        if not ok:
            nz = (got.abs().sum(dim=(1, 2, 3, 4)) > 0).nonzero().flatten().tolist()
            print("  nonzero co rows (got):", nz, " expected:", [co0])
            print("  ref[co0, ci=3].flatten():", ref[co0, 3].flatten().tolist())
            r = nz[0] if nz else co0
            print(f"  got[{r}, ci=3].flatten():", got[r, 3].flatten().tolist())
            print(f"  got[{r}, :, 1,1,1]:", got[r, :, 1, 1, 1].tolist())
            print(f"  ref[co0, :, 1,1,1]:", ref[co0, :, 1, 1, 1].tolist())
