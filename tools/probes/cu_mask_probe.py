"""Feasibility probe: does an HBM-bound kernel keep its bandwidth on a CU-masked stream (hipExtStreamCreateWithCUMask)?
Runs the GroupNorm-apply kernel and the conv kernel of this library on streams restricted to a fraction of the CUs."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
import mednet_hip
from mednet_hip import _lib as L, ops

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
lib = L.lib()


def masked_stream(keep_every, of):
    """stream on the CUs whose index i satisfies (i % of) < keep_every (evenly spread whatever the bit order is)"""
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if i % of < keep_every:
            words[i // 32] |= 1 << (i % 32)
            n += 1
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(s.value, device=dev), n


N, C, S = 4, 32, 128
x = torch.randn(N, C, S, S, S, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
g = torch.ones(C, device=dev)
b = torch.zeros(C, device=dev)
w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)
y = torch.empty_like(x)
mednet_hip.set_precision("bf16")


def timeit(fn, stream, iters=10):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) / iters


def gn():
    ops.group_norm_act(x, g, b, 8, 1e-5, L.ACT_ELU)


def conv():
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, S, S, S, C, C, 3, 1, 0, 1, 0, 0, 2, None, st), "fwd")


for keep, of in ((4, 4), (3, 4), (2, 4), (1, 4)):
    try:
        s, n = masked_stream(keep, of)
    except Exception as e:
        print("mask failed:", e)
        break
    tg = timeit(gn, s)
    tc = timeit(conv, s)
    nbytes = N * C * S ** 3 * 2
    print(f"CUs {n:3d}: gn+elu fwd {tg*1e3:7.1f} us ({3*nbytes/tg/1e9:5.2f} TB/s of 3 passes) | conv 32->32 {tc*1e3:7.1f} us", flush=True)
