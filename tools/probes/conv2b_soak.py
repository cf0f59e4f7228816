"""Race screen of conv2b_mfma_kernel (round 6): its operands arrive by LDS-DMA into a double-buffered input image and a ring of three
weight slots, ordered by one counted s_waitcnt vmcnt + s_barrier per phase -- a protocol error would show as an occasional wrong
output, most likely when the waves of a workgroup drift apart (other work on the CU, memory latency spikes).  300 launches of every
form (forward + statistics, data gradient + GroupNorm sums + summed gradient, the split-weight form), alone and beside a streaming
copy kernel on a second stream, at a 64 -> 64 and a 128 -> 128 layer: every output must equal the first run's bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
RUNS = int(os.environ.get("SOAK_RUNS", "300"))
side = torch.cuda.Stream()
junk_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk_b = torch.empty_like(junk_a)
bad = 0
for dt, mode in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    for n, c, s in ((4, 64, 64), (4, 128, 32)):
        g = torch.Generator(device=dev).manual_seed(3)
        x = torch.randn(n, c, s, s, s, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
        add = torch.randn(n, c, s, s, s, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
        gy = torch.randn(n, c, s, s, s, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
        coef = torch.randn(n, c, 2, device=dev, generator=g).contiguous()
        w = torch.randn(c, c, 3, 3, 3, device=dev, generator=g) * 0.03
        with mednet_hip.precision("fp16x2" if mode == "fp16" else mode):
            pk = ops.pack_conv_weight(w, 3, False)
        code = L.dt(x)
        st = torch.cuda.current_stream().cuda_stream
        y = torch.empty_like(x)
        rows_f = lib.mednet_conv3d_fused_stats_chunks(n, s, s, s, c, c, 3, code, code, 2)
        rows_b = lib.mednet_conv3d_dgrad_gn_rows_dt(n, s, s, s, c, c, 2, code)
        pf = torch.empty(n, rows_f, c, 2, device=dev)
        pb = torch.empty(n, rows_b, c, 2, device=dev)
        forms = {
            "fwd+stats": lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), n, s, s, s, c, c, 3, code, L.NDHWC, code, L.NDHWC,
                                                                0, 2, pf.data_ptr(), st), "fwd"),
            "dgrad+gn+add": lambda: L.check(lib.mednet_conv3d_dgrad_gn(x.data_ptr(), pk.data_ptr(), add.data_ptr(), y.data_ptr(), gy.data_ptr(), coef.data_ptr(), 3,
                                                                      pb.data_ptr(), n, s, s, s, c, c, 2, code, st), "dgrad_gn"),
        }
        if mode == "fp16":
            rows_s = lib.mednet_conv3d_fused_stats_chunks(n, s, s, s, c, c, 3, code, code, 2 | 8)
            ps = torch.empty(n, rows_s, c, 2, device=dev)
            forms["fwd+stats, split weights"] = lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), n, s, s, s, c, c, 3, code, L.NDHWC,
                                                                                      code, L.NDHWC, 0, 2 | 8, ps.data_ptr(), st), "fwd split")
        for name, fn in forms.items():
            part = {"fwd+stats": pf, "dgrad+gn+add": pb}.get(name, None)
            if part is None:
                part = ps
            fn()
            torch.cuda.synchronize()
            ref_y, ref_p = y.clone(), part.clone()
            mism = 0
            for i in range(RUNS):
                if i % 2:
                    with torch.cuda.stream(side):
                        junk_b.copy_(junk_a, non_blocking=True)
                y.fill_(7)
                part.fill_(float("nan"))
                fn()
                torch.cuda.synchronize()
                if not (torch.equal(y, ref_y) and torch.equal(part, ref_p)):
                    mism += 1
            bad += mism
            print(f"{mode} {c}->{c} @{s}^3 {name}: {RUNS} launches (every second one beside a 256 MB copy), {mism} differ from the first", flush=True)
print("TOTAL mismatching launches:", bad)
sys.exit(1 if bad else 0)
