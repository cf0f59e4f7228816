"""Race screen of the matrix-core landmark head (head_mfma.hip: wave-private LDS tiles handed between lanes without a barrier,
end-of-kernel reductions through a shared scratch): the same forward + backward N times, alone and beside a bandwidth-bound
kernel on another stream; every run's losses, feature gradient, GroupNorm partial sums, dW and db must equal the first run's bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = torch.device("cuda", 0)
lib = L.lib()
torch.manual_seed(3)
n, d, h, w, cin, nh, ncls = 2, 96, 96, 64, 32, 16, 2
sp = d * h * w
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for dt, code in ((torch.bfloat16, L.BF16), (torch.float16, L.F16)):
    z = torch.randn(n, d, h, w, cin, device=dev).to(dt)
    gy = torch.randn(n, d, h, w, cin, device=dev).to(dt)
    wt = torch.randn(nh + ncls, cin, device=dev) * 0.2
    bias = torch.randn(nh + ncls, device=dev)
    packed = torch.empty(lib.mednet_conv3d_pack_bytes(cin, nh + ncls, 1), dtype=torch.uint8, device=dev)
    L.check(lib.mednet_conv3d_pack_elt(wt.data_ptr(), packed.data_ptr(), cin, nh + ncls, 1, 0, code, L.stream()), "pack")
    hm = torch.randint(0, 256, (n, nh, sp), dtype=torch.uint8, device=dev)
    lab = torch.randint(0, ncls, (n, sp), dtype=torch.uint8, device=dev)
    cw = torch.tensor([0.05, 1.0], device=dev)
    rw = torch.full((nh,), 0.015, device=dev)
    ws = torch.empty(lib.mednet_head_landmark_ws_bytes(n, sp, nh, ncls), dtype=torch.uint8, device=dev)
    rows = lib.mednet_head_landmark_gn_rows(sp)
    one = torch.ones((), device=dev)
    side = torch.cuda.Stream()
    big = torch.randn(256 * 1024 * 1024 // 4, device=dev)

    def run():
        closs, rloss = torch.empty((), device=dev), torch.empty((), device=dev)
        saved = torch.empty(ncls, 2, device=dev)
        dz = torch.empty_like(z)
        part = torch.empty(n, rows, cin, 2, device=dev)
        dw, db = torch.empty(nh + ncls, cin, device=dev), torch.empty(nh + ncls, device=dev)
        L.check(lib.mednet_head_landmark_fwd(z.data_ptr(), packed.data_ptr(), bias.data_ptr(), hm.data_ptr(), nh * sp, lab.data_ptr(), sp,
                                             cw.data_ptr(), rw.data_ptr(), None, closs.data_ptr(), rloss.data_ptr(), saved.data_ptr(), n,
                                             sp, cin, nh, ncls, L.REG_L2, 1e-5, 0, L.NO_IGNORE, code, ws.data_ptr(), ws.numel(), L.stream()), "fwd")
        L.check(lib.mednet_head_landmark_bwd(z.data_ptr(), packed.data_ptr(), bias.data_ptr(), hm.data_ptr(), nh * sp, lab.data_ptr(), sp,
                                             cw.data_ptr(), rw.data_ptr(), saved.data_ptr(), one.data_ptr(), one.data_ptr(), dz.data_ptr(),
                                             gy.data_ptr(), L.ACT_ELU, part.data_ptr(), dw.data_ptr(), db.data_ptr(), n, sp, cin, nh, ncls,
                                             L.REG_L2, 1e-5, 0, L.NO_IGNORE, code, ws.data_ptr(), ws.numel(), L.stream()), "bwd")
        torch.cuda.synchronize()
        return closs, rloss, dz, part, dw, db

    ref = run()
    assert all(torch.isfinite(t.float()).all() for t in ref)
    bad = 0
    for i in range(reps):
        if i % 2:  # a bandwidth-bound neighbour on a second stream
            with torch.cuda.stream(side):
                big.mul_(1.0001)
        got = run()
        bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(got, ref))
    print(f"{dt}: {reps} forward + backward pairs ({reps // 2} beside a streaming kernel), tensors that differ from the first run: {bad}")
    assert bad == 0
