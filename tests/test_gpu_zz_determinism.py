"""Race screens -- run LAST (the file name sorts behind every other -m gpu file): bitwise run-to-run identity of whole training
steps at the sizes that are benchmarked, and insensitivity of the step to what free memory holds.

Round 3's GPU run went red on one of these (two bf16 steps of config 5 three fp32 ulp apart in the loss) 30 tests into an
`-x` run, which left 287 parity tests unreached.  The deviation has not been seen again in ~2 100 forward passes and ~670 full
steps on five other MI355X, so these tests now (a) sit behind the parity tests and (b) localise a
failure themselves: on a mismatch the step is re-run with mednet_hip.debug's trace open and the assertion names the first
tensor that differs."""
import numpy as np
import pytest
import torch

import mednet_hip
from mednet_hip.unet import loss as HL
from mednet_hip.unet import model as HM
from oracle import ref_cpu as O

from gpu_util import DEV

pytestmark = pytest.mark.gpu


def _fresh_step_run(ctor, mode, batch, traced=False):
    """One forward + loss + backward of a FRESH network and trainer (both streams) -> (loss, flat gradient buffer, trace)."""
    from mednet_hip import debug
    from mednet_hip.train import SegmentationStep
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
        if traced:
            debug.open_trace()
        (loss,) = step._fwd_bwd(batch)
        torch.cuda.synchronize()
        trace = debug.close_trace() if traced else None
        out = (float(loss), step.flat.grad.clone(), trace)
        step.flat.release()
        del net, step
    return out


def _describe_difference(ctor, mode, a, b, limit=12):
    """Where two flat gradient buffers differ: per parameter (train.FlatParams lays the parameters out in named_parameters()
    order, each padded to the buffer's alignment) the count, first index and values; plus the caching allocator's counters."""
    from mednet_hip.train import FlatParams
    with mednet_hip.precision(mode):
        net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
        flat = FlatParams(net)
        spans = [(k, int(p._mednet_grad.data_ptr() - flat.grad.data_ptr()) // 4, p.numel()) for k, p in net.named_parameters()]
        flat.release()
    diff = (a[1] != b[1]).nonzero().flatten()
    lines = []
    for k, off, cnt in spans:
        idx = diff[(diff >= off) & (diff < off + cnt)]
        if idx.numel():
            i = int(idx[0])
            lines.append(f"{k}: {idx.numel()} of {cnt} (first at {i - off}: {float(a[1][i])!r} vs {float(b[1][i])!r})")
        if len(lines) >= limit:
            lines.append("...")
            break
    st = torch.cuda.memory_stats()
    mem = {k: st.get(k) for k in ("allocated_bytes.all.current", "reserved_bytes.all.current", "num_alloc_retries", "segment.all.current")}
    return f"differing parameters: {lines or 'none (loss only)'}; allocator {mem}"


def _assert_step_is_bitwise_repeatable(ctor, mode, batch, what):
    """Two fresh runs of the same step must agree bit for bit (no atomics, fixed-order reductions, two streams).  If they do
    not, the step is run twice more with mednet_hip.debug's trace open -- a checksum of every tensor the ops produce -- so
    that the failure names the first kernel output that differs, not only the loss."""
    from mednet_hip import debug
    a = _fresh_step_run(ctor, mode, batch)
    b = _fresh_step_run(ctor, mode, batch)
    assert np.isfinite(a[0]) and bool(torch.isfinite(a[1]).all()), what
    assert float(a[1].abs().max()) > 0
    if a[0] == b[0] and torch.equal(a[1], b[1]):
        return a
    # what the failing pair already holds goes on record BEFORE anything is re-run (a rare event will not show in the re-runs):
    # the differing flat-gradient elements mapped to parameter names, and the allocator's state
    evidence = _describe_difference(ctor, mode, a, b)
    t0 = _fresh_step_run(ctor, mode, batch, traced=True)
    t1 = _fresh_step_run(ctor, mode, batch, traced=True)
    where = debug.first_difference(t0[2], t1[2])
    raise AssertionError(f"{what}: two runs differ -- loss {a[0]!r} vs {b[0]!r}, {int((a[1] != b[1]).sum())} of {a[1].numel()} "
                         f"gradient values; {evidence}; traced re-runs: losses {t0[0]!r} / {t1[0]!r}, first differing trace point: {where}")


@pytest.mark.parametrize("MODE16", ["fp16", "bf16"])
def test_cfg5_full_size_properties(MODE16):
    """BASELINE config 5 at its full size and batch (5 levels, 64 base channels, 160x160x96, N = 2) in fp16 storage with loss
    scaling (the mode BASELINE names) and in bf16: every loss and gradient finite, two runs bitwise identical.  (Parity at this
    size is test_cfg5_full_size_against_reference_golden.)"""
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[64, 128, 256, 512, 1024])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (160, 160, 96), 4, 0, seed=1234).items()}
    loss, _, _ = _assert_step_is_bitwise_repeatable(ctor, MODE16, batch, f"cfg5 full size {MODE16}")
    print(f"[cfg5 full size] {MODE16} loss {loss:.6f}, two runs bit-identical")


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_cfg2_benchmarked_shape_is_bitwise_repeatable(mode):
    """The shape bench.py times (BASELINE config 2: [32, 64, 128, 256], 128^3, N = 4, SegmentationStep._fwd_bwd, weight
    gradients on the side stream): the 32 -> 32 specialisation with its register-resident weights, LDS-DMA rows and
    accumulate-mode statistics, the persistent general kernel, the split-bf16 kernels in the fp32 mode."""
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256])
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(4, 1, (128, 128, 128), 4, 0, seed=99).items()}
    loss, _, _ = _assert_step_is_bitwise_repeatable(ctor, mode, batch, f"cfg2 128^3 N=4 {mode}")
    print(f"[cfg2 benchmarked shape] {mode} loss {loss:.6f}, two runs bit-identical")


def test_bitwise_reproducible_step():
    """No float atomics anywhere: two runs of the same step give identical bits (race screen)."""
    ctor = dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64])
    batch = O.synthetic_batch(2, 1, (16, 16, 16), 4, 0, seed=7)
    outs = []
    for mode in ("bf16", "bf16", "fp32", "fp32"):
        with mednet_hip.precision(mode):
            net = O.keyed_init_(HM.ResidualUNet3D(**ctor)).to(DEV)
            lg = net(batch["data"].to(DEV))
            HL.DiceLoss().to(DEV)(lg, batch["label"][:, -1].long().to(DEV)).backward()
            outs.append([lg.detach().clone()] + [p.grad.clone() for p in net.parameters()])
    for a, b in ((outs[0], outs[1]), (outs[2], outs[3])):
        for u, v in zip(a, b):
            assert torch.equal(u, v)


@pytest.mark.parametrize("cfg", ["cfg5_n2_bf16", "cfg2_n4_bf16", "cfg2_n1_fp32"])
def test_the_step_does_not_read_memory_nobody_wrote(cfg):
    """Every byte the caching allocator can hand out, and the library's workspaces, are filled with NaN patterns (then with
    huge finite values) before the step runs (mednet_hip.debug.poison): a GroupNorm partial row, a weight-gradient slab, a
    loss partial or a pack image that a reducer reads but no producer wrote would turn the loss or a gradient into NaN -- or
    simply change it.  Loss and every gradient must equal the unpoisoned run bit for bit."""
    from mednet_hip import debug
    ctor, shape, n, mode = {
        "cfg5_n2_bf16": (dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[64, 128, 256, 512, 1024]), (160, 160, 96), 2, "bf16"),
        "cfg2_n4_bf16": (dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]), (128, 128, 128), 4, "bf16"),
        "cfg2_n1_fp32": (dict(in_channels=1, out_channels=4, final_sigmoid=False, f_maps=[32, 64, 128, 256]), (128, 128, 128), 1, "fp32"),
    }[cfg]
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(n, 1, shape, 4, 0, seed=5).items()}
    ref = _fresh_step_run(ctor, mode, batch)
    assert np.isfinite(ref[0])
    for pattern in (0x7FC07FC0, 0x7F7F7F7F):
        debug.poison(pattern=pattern, big_gb=40.0, small_mb=768)
        got = _fresh_step_run(ctor, mode, batch)
        assert got[0] == ref[0], f"{cfg}: loss {got[0]!r} with poisoned memory (pattern {pattern:#x}), {ref[0]!r} without"
        assert torch.equal(got[1], ref[1]), f"{cfg}: gradients change with what free memory holds (pattern {pattern:#x})"
    torch.cuda.empty_cache()


@pytest.mark.parametrize("which", ["landmark", "unet3d"])
def test_the_other_step_classes_are_bitwise_repeatable(which):
    """LandmarkStep (config 4's path: 16 heat maps + 2 classes, its weight gradients on the side stream since round 4) and a
    SegmentationStep over UNet3D (the 'gcr' blocks: fused concatenation statistics, ReLU' folded into the pooling join and the
    head, the skip gradient read inside the concatenation's gradient) at 64^3, N = 2: two fresh runs agree bit for bit."""
    from mednet_hip.train import LandmarkStep, SegmentationStep
    nh = 16 if which == "landmark" else 0
    ncls = 2 if which == "landmark" else 4
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(2, 1, (64, 64, 64), ncls, nh, seed=77).items()}
    runs = []
    for _ in range(2):
        with mednet_hip.precision("bf16"):
            if which == "landmark":
                net = O.keyed_init_(HM.ResidualUNet3D(1, 18, False, f_maps=[32, 64, 128])).to(DEV)
                step = LandmarkStep(net, [0.05, 1.0], [0.015] * 16, "L2")
            else:
                net = O.keyed_init_(HM.UNet3D(1, 4, False, f_maps=[32, 64, 128])).to(DEV)
                step = SegmentationStep(net, loss_weight=[0.05, 1.0, 1.0, 1.0], lr=1e-3)
            out = step._fwd_bwd(batch)
            torch.cuda.synchronize()
            runs.append(([float(v) for v in out], step.flat.grad.clone()))
            step.flat.release()
            del net, step
    assert all(np.isfinite(v) for v in runs[0][0]) and float(runs[0][1].abs().max()) > 0
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]), which


@pytest.mark.parametrize("shape,wgs", [((4, 32, 32, (64, 64, 64)), 0), ((2, 64, 96, (24, 40, 56)), 24)])
def test_co_resident_weight_gradient_is_timing_independent(shape, wgs):
    """wgrad_mfma4_kernel's ring protocol (a counted vmcnt wait + one barrier per tick, slots recycled six ticks later, hand-written
    LDS-DMA the compiler does not see) must not depend on timing: the launch alone, beside a streaming copy on a second stream (memory
    latencies stretch) and beside a second weight gradient on the same CUs must give the same bits every time.  (The long form:
    tools/probes/wgrad4_soak.py, profiles/r05_wgrad4_race_screen.log -- 300 launches, none differs.)"""
    from mednet_hip import _lib as L
    lib = L.lib()
    n, cin, cout, (d, h, w) = shape
    CL = torch.channels_last_3d
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, cin, d, h, w, device=DEV, generator=g).bfloat16().contiguous(memory_format=CL)
    dy = torch.randn(n, cout, d, h, w, device=DEV, generator=g).bfloat16().contiguous(memory_format=CL)
    big = torch.randn(32 * 1024 * 1024, device=DEV)
    big2 = torch.empty_like(big)
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    assert lib.mednet_conv3d_wgrad_coresident(n, d, h, w, cin, cout, 3, L.BF16, L.BF16, L.ALGO_AUTO) == 1
    nbytes = lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, 3, wgs)
    ws1, ws2 = (torch.empty(nbytes, dtype=torch.uint8, device=DEV) for _ in range(2))

    def launch(stream, dw, ws):
        L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, n, d, h, w, cin, cout, 3, L.BF16, L.NDHWC, L.BF16,
                                        L.NDHWC, L.ALGO_MFMA, wgs, ws.data_ptr(), ws.numel(), stream.cuda_stream), "conv3d_wgrad")

    ref = torch.empty(cout, cin, 3, 3, 3, device=DEV)
    launch(main, ref, ws1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ref).all())
    for i in range(18):
        dw, dwb = torch.full_like(ref, float("nan")), torch.full_like(ref, float("nan"))
        ws1.fill_(0xFF)
        side.wait_stream(main)
        if i % 3 == 1:
            with torch.cuda.stream(side):
                big2.copy_(big)
                big.copy_(big2)
        elif i % 3 == 2:
            launch(side, dwb, ws2)
        launch(main, dw, ws1)
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(dw, ref), f"launch {i} (mode {i % 3}) differs from the first"
        if i % 3 == 2:
            assert torch.equal(dwb, ref), f"the concurrent launch {i} differs from the first"
