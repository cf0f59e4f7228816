"""Why is the first-layer kernel 6x slower in the step than stand-alone?  Same launch, (a) one output buffer re-used, (b) a ring of
8 output buffers (every launch writes memory that was last touched 8 GB of traffic ago), (c) ring + a 2 GB read-modify pass in between."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch, mednet_hip
from mednet_hip import _lib as L, ops
dev = "cuda:0"; lib = L.lib(); N, s, cout = 4, 128, 32
CL = torch.channels_last_3d
with mednet_hip.precision("fp32"):
    x = torch.randn(N, 1, s, s, s, device=dev)
    w = torch.randn(cout, 1, 3, 3, 3, device=dev) * 0.3
    pk = ops.pack_conv_weight(w, 3, False)
ys = [torch.empty(N, cout, s, s, s, device=dev).contiguous(memory_format=CL) for _ in range(8)]
rows = lib.mednet_conv3d_fused_stats_chunks(N, s, s, s, 1, cout, 3, L.F32, L.F32, L.ALGO_AUTO)
part = torch.empty(N, rows, cout, 2, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(y): L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), N, s, s, s, 1, cout, 3, L.F32, L.NDHWC, L.F32, L.NDHWC, 0, L.ALGO_AUTO, part.data_ptr(), st), "fwd")
def timed(fn, n=16):
    ev = [(torch.cuda.Event(True), torch.cuda.Event(True)) for _ in range(n)]
    for i in range(n):
        pre = fn(i, None)
        ev[i][0].record(); fn(i, 1); ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3
big = torch.empty(512 << 20, dtype=torch.float32, device=dev)
print("same buffer      : %.1f us" % timed(lambda i, go: run(ys[0]) if go else None))
print("ring of 8 buffers: %.1f us" % timed(lambda i, go: run(ys[i % 8]) if go else None))
print("ring + 2 GB pass : %.1f us" % timed(lambda i, go: run(ys[i % 8]) if go else big.add_(1.0)))
x2 = torch.randn(N, 1, s, s, s, device=dev)
print("fresh x each time: %.1f us" % timed(lambda i, go: run(ys[i % 8]) if go else x.copy_(x2)))
