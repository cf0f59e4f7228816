"""Race screen of round 6's last two kernels.  convt_dgrad32_mfma_kernel commits the next brick's 24 staging registers per thread into a
single-buffered LDS image between two barriers while its GroupNorm rows and the brick after are in flight; conv_c1_mfma_kernel (now
persistent) does the same with its halo values.  A protocol error would show as an occasional wrong output, most likely when the
waves of a workgroup drift apart.  SOAK_RUNS launches of every form -- alone, beside a 256 MB copy and beside a 32 -> 32 convolution
on a second stream (what the training step puts next to them) -- must equal the first launch bit for bit, partial rows included."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops, nn as hnn

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
RUNS = int(os.environ.get("SOAK_RUNS", "600"))
side = torch.cuda.Stream()
junk_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk_b = torch.empty_like(junk_a)
bad = 0
for dt, mode in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    code = L.dt(torch.empty(0, dtype=dt))
    g = torch.Generator(device=dev).manual_seed(11)
    # the co-runner: a 32 -> 32 convolution at 128^3 (conv32_mfma_kernel, one workgroup per CU with the whole register file)
    cx = torch.randn(2, 32, 128, 128, 128, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
    cy = torch.empty_like(cx)
    with mednet_hip.precision(mode):
        cpk = ops.pack_conv_weight(torch.randn(32, 32, 3, 3, 3, device=dev, generator=g) * 0.05, 3, False)

    def corunner(i):
        if i % 3 == 1:
            with torch.cuda.stream(side):
                junk_b.copy_(junk_a, non_blocking=True)
        elif i % 3 == 2:
            with torch.cuda.stream(side):
                L.check(lib.mednet_conv3d_fwd(cx.data_ptr(), cpk.data_ptr(), None, cy.data_ptr(), 2, 128, 128, 128, 32, 32, 3, code, L.NDHWC, code, L.NDHWC,
                                              0, 2, None, side.cuda_stream), "co-runner")

    # ---- ConvTranspose3d 64 -> 32 data gradient + GroupNorm-3 sums, N = 4, 64^3 -> 128^3 and a ragged N = 3 volume
    for n, shape in ((4, (64, 64, 64)), (3, (17, 22, 50))):
        d, h, w = shape
        dy = torch.randn(n, 32, 2 * d, 2 * h, 2 * w, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
        gy = torch.randn(n, 64, d, h, w, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
        gz = torch.randn(n, 64, d, h, w, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
        with mednet_hip.precision(mode):
            pk = ops.pack_conv_weight(torch.randn(64, 32, 3, 3, 3, device=dev, generator=g) * 0.05, 3, True)
        rows = lib.mednet_convt3d_dgrad_gn_rows(n, d, h, w, 64, 32, code, 2)
        dx = torch.empty_like(gy)
        part = torch.empty(n, rows, 64, 2, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def fn():
            L.check(lib.mednet_convt3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), dx.data_ptr(), gy.data_ptr(), gz.data_ptr(), 3, part.data_ptr(), n, d, h, w, 64, 32,
                                                code, 2, st), "convt3d_dgrad_gn")
        fn()
        torch.cuda.synchronize()
        ref_x, ref_p = dx.clone(), part.clone()
        mism = 0
        for i in range(RUNS):
            corunner(i)
            dx.fill_(7)
            part.fill_(float("nan"))
            fn()
            torch.cuda.synchronize()
            if not (torch.equal(dx, ref_x) and torch.equal(part, ref_p)):
                mism += 1
        bad += mism
        print(f"{mode} ConvTranspose3d 64->32 data gradient + sums N={n} {shape}: {RUNS} launches (a third alone, a third beside a copy, a third beside a "
              f"32->32 convolution), {mism} differ from the first", flush=True)

    # ---- first layer, 1 -> 32 at 128^3 N = 4 (1024 workgroups x 16 bricks) and 1 -> 64 on a ragged volume
    for n, cout, shape in ((4, 32, (128, 128, 128)), (2, 64, (36, 60, 70))):
        x = torch.randn(n, 1, *shape, device=dev, generator=g)
        with mednet_hip.precision("fp16x2" if mode == "fp16" else mode):  # (fp16: the split-weight form, a third MFMA per k-step)
            conv = hnn.Conv3d(1, cout, 3, bias=False).to(dev)
            y0, p0 = conv.forward_with_stats(x)
            torch.cuda.synchronize()
            ref_y, ref_p = y0.clone(), p0.clone()
            mism = 0
            for i in range(RUNS):
                corunner(i)
                y, p = conv.forward_with_stats(x)
                torch.cuda.synchronize()
                if not (torch.equal(y, ref_y) and torch.equal(p, ref_p)):
                    mism += 1
                del y, p
        bad += mism
        print(f"{mode}{'x2' if mode == 'fp16' else ''} first layer 1->{cout} N={n} {shape}: {RUNS} launches, {mism} differ from the first", flush=True)
print("TOTAL mismatching launches:", bad)
sys.exit(1 if bad else 0)
