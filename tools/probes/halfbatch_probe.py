"""Would two half-batch pipelines on two streams hide the forward's GroupNorm apply passes under the convolutions?
(They do not: conv32_mfma_kernel holds 483-494 of a lane's 512 registers, accumulation registers included, so nothing shares a CU
with it and the streams take turns -- profiles/r04_ab.md section 11.)
Chain per stream: [conv 32->32 (+ fused statistics) -> GroupNorm apply + ELU] x ROUNDS at 128^3.
  serial: one stream, the whole batch (N = 4) per launch: what the step does today
  halves: two streams, N = 2 each, launches issued alternately"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L, ops

dev = "cuda:0"
lib = L.lib()
s, c, ROUNDS = 128, 32, 6
CL = torch.channels_last_3d


def bufs(n):
    x = (torch.randn(n, c, s, s, s, device=dev) * 0.5).bfloat16().contiguous(memory_format=CL)
    y = torch.empty_like(x)
    chunks = lib.mednet_conv3d_fused_stats_chunks(n, s, s, s, c, c, 3, 1, 1, 2)
    part = torch.empty(n, max(chunks, 1), c, 2, device=dev)
    coef = torch.stack((torch.full((n, c), 0.05), torch.zeros(n, c)), dim=-1).to(dev).contiguous()
    return x, y, part, coef


w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv_weight(w, 3, False)


def conv(b, n, st):
    x, y, part, coef = b
    L.check(lib.mednet_conv3d_fwd(x.data_ptr(), pk.data_ptr(), None, y.data_ptr(), n, s, s, s, c, c, 3, 1, 0, 1, 0, 0, 2, part.data_ptr(),
                                  st.cuda_stream), "fwd")


def apply(b, n, st):
    x, y, part, coef = b
    L.check(lib.mednet_gn_act_fwd(y.data_ptr(), coef.data_ptr(), None, x.data_ptr(), n, s ** 3, c, L.ACT_ELU, 1, 1, st.cuda_stream), "apply")


full = bufs(4)
ha, hb = bufs(2), bufs(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    for _ in range(ROUNDS):
        conv(full, 4, s1)
        apply(full, 4, s1)


def halves():
    for _ in range(ROUNDS):
        conv(ha, 2, s1)
        conv(hb, 2, s2)
        apply(ha, 2, s1)
        apply(hb, 2, s2)


def only(fn_name):
    for _ in range(ROUNDS):
        (conv if fn_name == "conv" else apply)(full, 4, s1)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(s1)
    s2.wait_event(e0)
    for _ in range(3):
        fn()
    s1.wait_stream(s2)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3 / ROUNDS * 1e3


ind = bufs(4)


def independent():  # conv over one batch beside the apply pass of ANOTHER tensor: what co-residency is worth at best
    for _ in range(ROUNDS):
        conv(full, 4, s1)
        apply(ind, 4, s2)


def independent_half():  # ... the apply pass over half as much
    for _ in range(ROUNDS):
        conv(full, 4, s1)
        apply(hb, 2, s2)


for rep in range(2):
    print(f"conv beside an independent apply pass, us per round: same size {timed(independent):7.1f}   half size {timed(independent_half):7.1f}", flush=True)
for rep in range(3):
    print(f"per round (conv + apply over 4 samples), us:  conv only {timed(lambda: only('conv')):7.1f}   apply only {timed(lambda: only('apply')):7.1f}"
          f"   serial {timed(serial):7.1f}   two half-batch streams {timed(halves):7.1f}", flush=True)
