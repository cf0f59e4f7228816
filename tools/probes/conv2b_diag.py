"""Where does conv2b_mfma_kernel's output differ from the general kernel's?  (development aid, round 6)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops
dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
n, cin, cout, shape = 1, 64, 64, (32, 64, 64)
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(23)
rand = lambda c: torch.randn(n, c, *shape, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
x, dy, add, gy = rand(cin), rand(cout), rand(cin), rand(cin)
w = torch.randn(cout, cin, 3, 3, 3, device=dev, generator=g) * 0.05
coef = torch.randn(n, cin, 2, device=dev, generator=g).contiguous()
st = torch.cuda.current_stream().cuda_stream
with mednet_hip.precision("bf16"):
    pk = ops.pack_conv_weight(w, 3, False)
d, h, wd = shape
code = L.dt(x)
lib.mednet_set_option(b"conv2b_min_fill", 0)


def run(opt, gact, with_add):
    lib.mednet_set_option(b"conv2b", opt)
    rows = lib.mednet_conv3d_dgrad_gn_rows_dt(n, d, h, wd, cin, cout, 2, code)
    dx = torch.empty_like(x)
    part = torch.zeros(n, rows, cin, 2, device=dev)
    L.check(lib.mednet_conv3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), add.data_ptr() if with_add else None, dx.data_ptr(), gy.data_ptr(), coef.data_ptr(),
                                       gact, part.data_ptr(), n, d, h, wd, cin, cout, 2, code, st), "dgrad_gn")
    torch.cuda.synchronize()
    return dx, part


for trial in range(3):
    for gact, with_add in ((0, False), (3, False), (3, True)):
        a, _ = run(1, gact, with_add)
        b, _ = run(0, gact, with_add)
        diff = (a != b)  # N C D H W (logical)
        print(f"trial {trial} gact {gact} add {with_add}: {diff.float().mean().item():.4%} differ")
        if diff.any():
            idx = diff.nonzero()
            for name, col, div in (("channel block", 1, 32), ("z", 2, 1), ("z brick", 2, 4), ("y brick", 3, 8), ("x brick", 4, 16), ("y", 3, 1), ("x", 4, 1)):
                vals, cnt = torch.unique(idx[:, col] // div, return_counts=True)
                print(f"   by {name}: " + " ".join(f"{int(v)}:{int(c)}" for v, c in zip(vals[:40], cnt[:40])))
            e = idx[0]
            print("   first:", e.tolist(), float(a[tuple(e)]), float(b[tuple(e)]))
