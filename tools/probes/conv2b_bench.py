"""Round 6: conv2b_mfma_kernel (two channel blocks per wave) against conv_mfma_kernel<1> (option conv2b=0) at the layer shapes of
BASELINE configs 2 and 5 that it takes: forward with fused GroupNorm statistics, data gradient with GroupNorm-backward sums, through
the C ABI, HIP events, three interleaved rounds.  Operands with the statistics the step's tensors have (post-ELU activations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
ALGO_MFMA = 2


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def case(n, cin, cout, shape, dt=torch.bfloat16):
    d, h, w = shape
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.nn.functional.elu(torch.randn(n, cin, d, h, w, device=dev, generator=g)).to(dt).contiguous(memory_format=CL)
    dy = (torch.randn(n, cout, d, h, w, device=dev, generator=g) * 1e-3).to(dt).contiguous(memory_format=CL)
    gy = torch.randn(n, cin, d, h, w, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
    coef = torch.randn(n, cin, 2, device=dev, generator=g).contiguous()
    wgt = torch.randn(cout, cin, 3, 3, 3, device=dev, generator=g) * 0.05
    with mednet_hip.precision("bf16" if dt == torch.bfloat16 else "fp16"):
        pk = ops.pack_conv_weight(wgt, 3, False)
    code = L.dt(x)
    y = torch.empty(n, cout, d, h, w, device=dev, dtype=dt).contiguous(memory_format=CL)
    dx = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * n * d * h * w * cin * cout * 27
    res = {}
    for rnd in range(3):
        for opt in (1, 0):
            lib.mednet_set_option(b"conv2b", opt)
            rows = lib.mednet_conv3d_fused_stats_chunks(n, d, h, w, cin, cout, 3, code, code, ALGO_MFMA)
            part = torch.empty(n, rows, cout, 2, device=dev)
            t = timeit(lambda: L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), n, d, h, w, cin, cout, 3, code, L.NDHWC,
                                                              code, L.NDHWC, 0, ALGO_MFMA, part.data_ptr(), st), "fwd"))
            res.setdefault(("fwd+stats", opt), []).append(t)
            if cin % 64 == 0:
                rows = lib.mednet_conv3d_dgrad_gn_rows_dt(n, d, h, w, cin, cout, ALGO_MFMA, code)
                part = torch.empty(n, rows, cin, 2, device=dev)
                t = timeit(lambda: L.check(lib.mednet_conv3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), None, dx.data_ptr(), gy.data_ptr(), coef.data_ptr(), 3,
                                                                       part.data_ptr(), n, d, h, w, cin, cout, ALGO_MFMA, code, st), "dgrad_gn"))
                res.setdefault(("dgrad+gn", opt), []).append(t)
                t = timeit(lambda: L.check(lib.mednet_conv3d_fwd(dy.data_ptr(), pk.data_ptr(), None, dx.data_ptr(), n, d, h, w, cout, cin, 3, code, L.NDHWC,
                                                                  code, L.NDHWC, 1, ALGO_MFMA, None, st), "dgrad"))
                res.setdefault(("dgrad", opt), []).append(t)
    lib.mednet_set_option(b"conv2b", 1)
    for what in ("fwd+stats", "dgrad", "dgrad+gn"):
        if (what, 1) in res:
            a, b = min(res[(what, 1)]), min(res[(what, 0)])
            print(f"conv {cin:4d}->{cout:4d} @{d}x{h}x{w} N={n} {what:10s}: two-block {a:7.1f} us {flop / a / 1e6:7.1f} TF/s | general {b:7.1f} us "
                  f"{flop / b / 1e6:7.1f} TF/s | {b / a:5.3f}x", flush=True)


which = os.environ.get("C2B_WHICH", "cfg2,cfg5")
if "cfg2" in which:
    for n, cin, cout, s in ((4, 64, 64, 64), (4, 32, 64, 64), (4, 128, 128, 32), (4, 64, 128, 32)):
        case(n, cin, cout, (s, s, s))
if "cfg5" in which:
    for n, cin, cout, shape in ((2, 64, 64, (160, 160, 96)), (2, 128, 128, (80, 80, 48)), (2, 256, 256, (40, 40, 24)), (2, 512, 512, (20, 20, 12))):
        case(n, cin, cout, shape)
