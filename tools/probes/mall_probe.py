"""Does a producer -> consumer chain run faster sample by sample, with the intermediate tensor still in the 256 MiB Infinity
Cache, than over the whole batch (537 MB per tensor at 32 channels x 128^3 x N=4 in bf16: nothing survives between kernels)?

Chain under test = one forward layer of the full-resolution block: conv 3x3x3 32->32 (+ fused statistics) -> gn_finalize ->
GroupNorm apply + ELU.  Timed as (a) three launches over N=4, (b) the same three launches per sample, N times.  Also the
backward pair data gradient -> GroupNorm-backward apply.  Prints us per batch for both forms."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch  # noqa: E402

import mednet_hip  # noqa: E402
from mednet_hip import _lib as L, ops  # noqa: E402

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
mednet_hip.set_precision("bf16")


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
    return e0.elapsed_time(e1) / iters * 1e3


def main(N=4, C=32, S=128, G=8):
    x = torch.randn(N, C, S, S, S, device=dev).bfloat16().contiguous(memory_format=CL)
    y = torch.empty_like(x, memory_format=CL)
    z = torch.empty_like(x, memory_format=CL)
    dz = torch.randn(N, C, S, S, S, device=dev).bfloat16().contiguous(memory_format=CL)
    dy = torch.empty_like(x, memory_format=CL)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    pk = ops.pack_conv_weight(w, 3, False)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    sp = S ** 3
    ws = torch.empty(lib.mednet_gn_ws_bytes(N, C, sp), dtype=torch.uint8, device=dev)

    def layer(n0, nn):
        """conv + statistics + apply for samples [n0, n0 + nn)"""
        off = n0 * sp * C * 2
        chunks = lib.mednet_conv3d_fused_stats_chunks(nn, S, S, S, C, C, 3, L.BF16, L.BF16, L.ALGO_AUTO)
        part = torch.empty(nn, chunks, C, 2, device=dev)
        stats = torch.empty(nn, G, 2, device=dev)
        coef = torch.empty(nn, C, 2, device=dev)
        L.check(lib.mednet_conv3d_fwd(x.data_ptr() + off, pk.data_ptr(), None, y.data_ptr() + off, nn, S, S, S, C, C, 3, L.BF16, L.NDHWC,
                                      L.BF16, L.NDHWC, 0, L.ALGO_AUTO, part.data_ptr(), st), "conv")
        L.check(lib.mednet_gn_finalize(part.data_ptr(), chunks, gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), coef.data_ptr(),
                                       nn, sp, C, G, 1e-5, ws.data_ptr(), ws.numel(), st), "fin")
        L.check(lib.mednet_gn_act_fwd(y.data_ptr() + off, coef.data_ptr(), None, z.data_ptr() + off, nn, sp, C, L.ACT_ELU, L.BF16, L.BF16, st), "apply")
        return stats, coef

    stats, coef = layer(0, N)

    def bwd(n0, nn):
        """data gradient (plain) -> GroupNorm-backward apply for samples [n0, n0 + nn)"""
        off = n0 * sp * C * 2
        L.check(lib.mednet_conv3d_fwd(dz.data_ptr() + off, pk.data_ptr(), None, dy.data_ptr() + off, nn, S, S, S, C, C, 3, L.BF16, L.NDHWC,
                                      L.BF16, L.NDHWC, 1, L.ALGO_AUTO, None, st), "dgrad")
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        L.check(lib.mednet_gn_act_bwd(dy.data_ptr() + off, None, y.data_ptr() + off, None, coef[n0:n0 + nn].data_ptr(), stats[n0:n0 + nn].data_ptr(),
                                      gamma.data_ptr(), z.data_ptr() + off, None, dg.data_ptr(), db.data_ptr(), nn, sp, C, G, L.ACT_ELU,
                                      L.ACT_NONE, L.BF16, ws.data_ptr(), ws.numel(), st), "gn_bwd")

    for name, fn in (("forward layer (conv + statistics + apply)", layer), ("backward pair (data gradient + GroupNorm backward, 2 passes)", bwd)):
        res = {"batch": 1e9, "per_sample": 1e9}
        for _ in range(3):
            res["batch"] = min(res["batch"], timeit(lambda: fn(0, N)))
            res["per_sample"] = min(res["per_sample"], timeit(lambda: [fn(n, 1) for n in range(N)]))
        print(f"{name}: whole batch N={N}: {res['batch']:8.1f} us | sample by sample: {res['per_sample']:8.1f} us "
              f"({res['per_sample'] / res['batch'] - 1:+.1%})", flush=True)


if __name__ == "__main__":
    main()
