"""Round 6: convt_dgrad32_mfma_kernel against conv_mfma_kernel<2> (option convt_dgrad32=0) at the ConvTranspose3d 64 -> 32 data
gradient of BASELINE configs 2 / 4 (N = 4, 64^3 -> 128^3), plain and with the fused GroupNorm-3 sums, through the C ABI, three
interleaved rounds.  Algorithmic bytes: dy read + dx written (+ y3 and z read in the GN form)."""
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


def timeit(fn, iters=20):
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


def case(n, s, dt):
    cin, cout = 64, 32
    g = torch.Generator(device=dev).manual_seed(5)
    dy = (torch.randn(n, cout, 2 * s, 2 * s, 2 * s, device=dev, generator=g) * 1e-3).to(dt).contiguous(memory_format=CL)
    gy = torch.randn(n, cin, s, s, s, device=dev, generator=g).to(dt).contiguous(memory_format=CL)
    gz = torch.nn.functional.elu(torch.randn(n, cin, s, s, s, device=dev, generator=g)).to(dt).contiguous(memory_format=CL)
    wgt = torch.randn(cin, cout, 3, 3, 3, device=dev, generator=g) * 0.05
    with mednet_hip.precision("bf16" if dt == torch.bfloat16 else "fp16"):
        pk = ops.pack_conv_weight(wgt, 3, True)
    code = L.dt(dy)
    dx = torch.empty_like(gy)
    st = torch.cuda.current_stream().cuda_stream
    flop = 2.0 * n * s ** 3 * cin * cout * 27
    byts = {"dgrad": dy.numel() * 2 + dx.numel() * 2, "dgrad+gn": dy.numel() * 2 + 3 * dx.numel() * 2}
    res = {}
    for rnd in range(3):
        for opt in (1, 0):
            lib.mednet_set_option(b"convt_dgrad32", opt)
            rows = lib.mednet_convt3d_dgrad_gn_rows(n, s, s, s, cin, cout, code, ALGO_MFMA)
            part = torch.empty(n, rows, cin, 2, device=dev)
            t = timeit(lambda: L.check(lib.mednet_convt3d_dgrad(dy.data_ptr(), pk.data_ptr(), dx.data_ptr(), n, s, s, s, cin, cout, code, code, ALGO_MFMA, st), "dgrad"))
            res.setdefault(("dgrad", opt), []).append(t)
            t = timeit(lambda: L.check(lib.mednet_convt3d_dgrad_gn(dy.data_ptr(), pk.data_ptr(), dx.data_ptr(), gy.data_ptr(), gz.data_ptr(), 3, part.data_ptr(),
                                                                    n, s, s, s, cin, cout, code, ALGO_MFMA, st), "dgrad_gn"))
            res.setdefault(("dgrad+gn", opt), []).append(t)
    lib.mednet_set_option(b"convt_dgrad32", 1)
    for what in ("dgrad", "dgrad+gn"):
        a, b = min(res[(what, 1)]), min(res[(what, 0)])
        print(f"convT dgrad 64->32 N={n} {s}^3 -> {2 * s}^3 {str(dt)[6:]} {what:9s}: specialised {a:7.1f} us {flop / a / 1e6:6.1f} TF/s {byts[what] / a / 1e6:5.2f} TB/s | "
              f"general {b:7.1f} us {flop / b / 1e6:6.1f} TF/s {byts[what] / b / 1e6:5.2f} TB/s | {b / a:5.3f}x", flush=True)


for dt in (torch.bfloat16, torch.float16):
    case(4, 64, dt)
case(1, 64, torch.bfloat16)
