"""torch.autograd.Function wrappers around the C ABI: one per fused device op of the U-Net path.

Every forward/backward here is a handful of libmednet_hip launches on torch's current HIP stream; PyTorch only
provides memory (caching allocator), the stream and the autograd graph.  Internal activations are channels-last
(`torch.channels_last_3d`: logical N,C,D,H,W, physical N,D,H,W,C) so the MFMA kernels read 16-byte channel
fragments; the network input (C=1) and the logits (written planar by the 1x1x1 head) are plain NCDHW.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import _lib as L
from . import config, debug

CL = torch.channels_last_3d

# Optional profiling of the dominant kernel (bench.py): list of (start_event, end_event, flops) per matching launch
PROFILE = {"enabled": False, "events": [], "match": None}


# ---- weight gradients on a side stream (trainer mode only) ------------------------------------------------------------
# In the backward pass the weight gradient of a layer is off the critical path (nothing needs dW before the all-reduce /
# Adam), while the data gradient -> GroupNorm backward chain is HBM-bound.  When the trainer provides in-place gradient
# targets, weight-gradient kernels are launched on a second HIP stream so the matrix-core work overlaps the bandwidth
# work of the main stream.  train.py joins the streams before the gradient exchange.
# Workgroups of a weight-gradient launch on the side stream: HALF the CUs.  A weight-gradient workgroup takes a CU's whole
# register file; with one per CU nothing of the main stream -- not even the 32-workgroup reducer in front of a GroupNorm backward
# pass -- starts before they retire, and the two streams merely take turns.  With 128 the other 128 CUs run the bandwidth-bound
# GroupNorm passes of the main stream beside them, and the main stream's persistent kernels (256 / 512 workgroups) still divide
# evenly over what is left (profiles/r04_ab.md section 9: 144 or 192 are worse than 128 AND than 256).  0: the library's default.
SIDE = {"enabled": False, "stream": None, "keepalive": [], "wgrad_wgs": int(os.environ.get("MEDNET_SIDE_WGRAD_WGS", "128"))}


SIDE_MIN_VOXELS = int(os.environ.get("MEDNET_SIDE_MIN_VOXELS", "0"))  # (A/B knob: half-chip weight gradients from this layer size on)
SIDE_CORESIDENT = os.environ.get("MEDNET_SIDE_CORESIDENT", "1") == "1"  # (A/B knob: 0 = half the CUs for the co-resident kernel too)


def wgrad_coresident(n, d, h, w, cin, cout, ksize, x, dy) -> bool:
    return bool(L.lib().mednet_conv3d_wgrad_coresident(n, d, h, w, cin, cout, ksize, L.dt(x), L.dt(dy), config.conv_algo()))


def _runs_beside(main, cand, device) -> bool:
    """Does work queued on `cand` execute while `main` is busy?  HIP multiplexes its streams onto a few hardware queues (4 by
    default) in creation order; two streams that land on the same queue take turns whatever the program says."""
    x = torch.zeros(64, device=device)
    e0, em, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(cand):
        x.add_(1.0)  # (a stream's first launch creates its hardware queue: milliseconds that would read as "did not overlap")
    torch.cuda.synchronize(device)
    with torch.cuda.stream(main):
        e0.record(main)
        torch.cuda._sleep(3_000_000)  # main is busy for a millisecond or two
        em.record(main)
    with torch.cuda.stream(cand):
        x.add_(1.0)
        ec.record(cand)
    torch.cuda.synchronize(device)
    return e0.elapsed_time(ec) < 0.5 * e0.elapsed_time(em)


def side_stream(device):
    """The weight-gradient stream: a stream that really runs beside the current one.  With a process group up (RCCL and c10d have
    taken streams of their own by then) the first new stream shared the compute stream's hardware queue: the one-rank rehearsal
    of the data-parallel step ran every kernel on one queue, 23.7 ms instead of 20.5 (profiles/r04_ab.md section 15).  So the
    candidates are TESTED, once, at the first use (an eager warm-up step): up to 8 new streams, the first that overlaps wins."""
    if SIDE["stream"] is None:
        main = torch.cuda.current_stream(device)
        probe = os.environ.get("MEDNET_SIDE_PROBE", "1") == "1" and not torch.cuda.is_current_stream_capturing()
        cands, chosen = [], None
        prio = int(os.environ.get("MEDNET_SIDE_PRIORITY", "0"))  # (A/B knob: HIP stream priority of the weight-gradient stream)
        for _ in range(8 if probe else 1):
            c = torch.cuda.Stream(device=device, priority=prio) if prio else torch.cuda.Stream(device=device)
            cands.append(c)  # (kept alive until the choice is made: a released stream's slot would be handed out again)
            if not probe or _runs_beside(main, c, device):
                chosen = c
                break
        SIDE["stream"] = chosen if chosen is not None else cands[0]
        # True / False: MEASURED; None: the probe did not run (hipGraph capture, MEDNET_SIDE_PROBE=0) -- the first new stream is
        # taken on trust and may share the compute stream's hardware queue (bench.py reports the tri-state)
        SIDE["overlaps"] = (chosen is not None) if probe else None
        SIDE["candidates_tried"] = len(cands)
        if probe and chosen is None:  # no queue to itself: one workgroup per CU again (half the chip for twice as long gains nothing in turns)
            SIDE["wgrad_wgs"] = 0
    return SIDE["stream"]


def join_side_stream():
    if SIDE["stream"] is not None:
        torch.cuda.current_stream().wait_stream(SIDE["stream"])
    SIDE["keepalive"].clear()


class _OnSide:
    """Context: run the enclosed launches on the side stream after everything queued so far on the main stream."""

    def __init__(self, active, device, *tensors, coresident=False):
        # coresident: the enclosed launch leaves half of every CU free (mednet_conv3d_wgrad_coresident): it gets ALL the CUs, and
        # the main stream's bandwidth-bound passes run on the same CUs beside it (round 5; profiles/r05_ab.md)
        self.active, self.device, self.tensors, self.coresident = active, device, tensors, bool(coresident) and SIDE_CORESIDENT
        # the `workgroups` argument of the weight-gradient launch AND its workspace query inside this context (0: the library's
        # plan, one workgroup per CU -- a weight gradient launched on the main stream keeps the whole chip)
        self.workgroups = 0

    def __enter__(self):
        if self.active:
            main = torch.cuda.current_stream(self.device)
            side = side_stream(self.device)
            side.wait_stream(main)
            for t in self.tensors:
                if t is not None:
                    # Operands stay referenced until join_side_stream(), i.e. until the main stream has been made to
                    # wait for the side stream: (1) the caching allocator cannot recycle memory the side stream still
                    # reads (no Tensor.record_stream: its per-block events made every later allocation poll hundreds of
                    # events -- 9 of the host's 15 ms per step), and (2) autograd cannot accumulate INTO them: a
                    # gradient handed on to autograd (ConvTranspose's skip gradient is `dy` itself) is summed in place
                    # with later arrivals when nobody else holds it.
                    SIDE["keepalive"].append(t)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            t = self.tensors[-1] if self.tensors else None
            vox = (t.shape[0] * t[0, 0].numel()) if (t is not None and t.dim() == 5) else 1 << 40
            if SIDE["wgrad_wgs"] > 0 and vox >= SIDE_MIN_VOXELS and not self.coresident:
                self.workgroups = SIDE["wgrad_wgs"]
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
        return False


class profiled_conv:
    """bench.py's live timing of the dominant kernel: HIP events on the launch stream around matching conv launches."""

    def __init__(self, ksize, cin, cout, n, d, h, w):
        self.on = PROFILE["enabled"] and PROFILE["match"] is not None and PROFILE["match"](ksize, cin, cout, d, h, w)
        self.flops = 2.0 * n * d * h * w * cin * cout * ksize ** 3

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE["events"].append((self.e0, self.e1, self.flops))
        return False


class profiled_wgrad(profiled_conv):
    """The same for the weight gradient of matching layers (bench.py's second roofline object): the events go on the stream the
    launch runs on -- inside _OnSide that is the weight-gradient stream, where the kernel shares the CUs with the main stream's
    bandwidth-bound passes."""

    def __init__(self, ksize, cin, cout, n, d, h, w):
        super().__init__(ksize, cin, cout, n, d, h, w)
        self.on = self.on and PROFILE.get("wgrad_events") is not None

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE["wgrad_events"].append((self.e0, self.e1, self.flops))
        return False


class profiled_dgrad(profiled_conv):
    """The same for the DATA gradient of matching layers (bench.py's `roofline_dgrad` / `roofline.family`): `variant` names the
    epilogue the launch carries ("gn": GroupNorm-backward sums, "add+gn": summed residual gradient + sums, "plain")."""

    def __init__(self, variant, ksize, cin, cout, n, d, h, w):
        super().__init__(ksize, cin, cout, n, d, h, w)
        self.on = self.on and PROFILE.get("dgrad_events") is not None
        self.variant = variant

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE["dgrad_events"].append((self.e0, self.e1, self.flops, self.variant))
        return False


def _with_algo(backward):
    """backward of a conv-family Function: replays the algorithm choice (config.conv_algo()) captured at forward."""
    def wrapped(ctx, *grads):
        with config.algo_scope(getattr(ctx, "algo", None)):
            return backward(ctx, *grads)
    return wrapped


def _grad_target(param, shape):
    """Trainer hook (train.FlatParams): when a Parameter carries `_mednet_grad` (a contiguous fp32 view into the flat
    gradient buffer) the kernels write the gradient there and autograd gets None -> no copy / accumulate kernels."""
    tgt = getattr(param, "_mednet_grad", None)
    if tgt is not None:
        assert tuple(tgt.shape) == tuple(shape) and tgt.dtype == torch.float32 and tgt.is_contiguous()
        return tgt, True
    return torch.empty(shape, dtype=torch.float32, device=param.device), False


def to_cl(x: torch.Tensor) -> torch.Tensor:
    """Physical NDHWC. No-op for tensors produced by this package."""
    return x.contiguous(memory_format=CL)


def empty_cl(n, c, d, h, w, dtype, device):
    return torch.empty((n, c, d, h, w), dtype=dtype, device=device, memory_format=CL)


def _storage_dtype(x: torch.Tensor) -> torch.dtype:
    return config.act_dtype()


def _as_act(x: torch.Tensor) -> torch.Tensor:
    """Inputs that are neither fp32 nor the mode's 16-bit storage type (e.g. fp64 user tensors, bf16 in fp16 mode) are
    brought to the activation dtype."""
    if x.dtype == torch.float32 or x.dtype == config.act_dtype():
        return x
    return x.to(config.act_dtype())


# ------------------------------------------------------------------------------------------------- weights
def pack_conv_weight(weight: torch.Tensor, ksize: int, transposed: bool, out: torch.Tensor = None) -> torch.Tensor:
    """PyTorch-layout fp32 weight -> opaque packed buffer read by the forward / data-gradient kernels (`out`: rewrite that buffer)."""
    L.require_gpu(weight, "pack_conv_weight")
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    nbytes = L.lib().mednet_conv3d_pack_bytes(cin, cout, ksize)
    if out is not None and (out.numel() != nbytes or out.device != w.device):
        raise RuntimeError("pack_conv_weight: `out` is not this layer's pack buffer")
    buf = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    # element type of the matrix-core fragment images; fp32 storage: bf16 high + low images (split-bf16 contraction); fp16x2: fp16 + low
    elt = config.pack_elt()
    L.check(L.lib().mednet_conv3d_pack_elt(w.data_ptr(), buf.data_ptr(), cin, cout, ksize, int(transposed), elt, L.stream()),
            "conv3d_pack")
    return buf


# ------------------------------------------------------------------------------------------------- Conv3d
class GN3Hook:
    """Carried by the output tensor of an ExtResNetBlock (`out._mednet_gn3`): lets the op that consumes `out` -- the 1x1x1
    head or the pooling + skip join -- take the first pass of the block's GroupNorm-3 backward while it produces the block's
    output gradient (mednet_head_dgrad_gn / mednet_pool2_bwd_gn).  The block's backward uses the sums only if the gradient it
    receives IS that tensor, untouched (same object, same version counter): any other consumer of `out` makes autograd
    accumulate into / replace it, and the block falls back to its stand-alone pass."""
    __slots__ = ("gn_in", "act", "partial", "dx", "version", "lazy")

    def __init__(self):
        self.gn_in = None  # the GroupNorm's input (y3 of an ExtResNetBlock; x of a plain GroupNorm)
        self.act = 0
        self.partial = self.dx = None
        self.version = -1
        self.lazy = None  # (dy_pool, skip gradient, pool mode): `dx` was NOT written, the block's apply pass rebuilds it

    def offer(self, dx, partial, lazy=None):
        self.dx, self.partial, self.version, self.lazy = dx, partial, dx._version, lazy

    def take(self, dout):
        """The sums, if `dout` is exactly the tensor they were taken from; releases the references either way.  A LAZY offer (the
        gradient tensor is a placeholder nobody wrote: SkipPool2Fn with sole_consumer) cannot be declined -- the caller asserted
        that the block output has no other consumer; a violated contract is an error, not a silent wrong gradient."""
        partial, dx, version, lazy = self.partial, self.dx, self.version, self.lazy
        self.partial = self.dx = self.lazy = None
        ok = partial is not None and dout is dx and dout._version == version
        if lazy is not None and not ok:
            raise RuntimeError("mednet_hip: the block output handed to skip_pool2(sole_consumer=True) has another consumer (or its "
                               "gradient was replaced by a hook): the pooling backward did not materialise its gradient (MEDNET_LAZY_POOL=0 turns "
                               "the optimisation off)")
        GN3_COUNT["taken" if ok else ("declined" if partial is not None else "absent")] += 1
        self.lazy = lazy if ok else None  # (read and cleared by the block's backward)
        return partial if ok else None


FUSE_GN3 = os.environ.get("MEDNET_FUSE_GN3", "1") == "1"  # A/B knob


class GNBHook(GN3Hook):
    """The same contract for a plain GroupNorm (+ activation) whose output feeds a 3x3x3 convolution directly (the 'gcr'
    orders of UNet3D, components.py:12-67): the conv's data gradient takes {sum du, sum du * x} of that GroupNorm's backward
    in its epilogue (mednet_conv3d_dgrad_gn) and offers them with the gradient tensor it produced."""
    __slots__ = ("coef",)

    def __init__(self):
        super().__init__()
        self.coef = None


class ActMaskHook:
    """Carried by the output z of a fused conv -> activation layer: if z goes straight into a GroupNorm, that GroupNorm's
    backward multiplies the gradient it produces by act'(z) (z is its own input, in registers anyway) and says so here; the
    conv layer's backward then skips its activation-backward pass -- if the gradient it receives is that very tensor."""
    __slots__ = ("act", "dx", "version")

    def __init__(self, act):
        self.act, self.dx, self.version = act, None, -1

    def offer(self, dx):
        self.dx, self.version = dx, dx._version

    def take(self, dz):
        dx, version = self.dx, self.version
        self.dx = None
        ok = dx is not None and dz is dx and dz._version == version
        GN3_COUNT["masked"] += int(ok)
        return ok


def _gnb_hook_of(x, dtype):
    h = getattr(x, "_mednet_gnb", None) if FUSE_GN3 else None
    if h is None or h.gn_in is None or h.gn_in.shape != x.shape or h.gn_in.dtype != dtype or dtype not in config.HALF_TYPES:
        return None
    return h
GN3_COUNT = {"taken": 0, "declined": 0, "absent": 0, "masked": 0}  # (tests: how the blocks' backward passes found their sums)


def _gn3_hook_of(x, dtype):
    # (any storage type: the head's data gradient and the pooling backward take the sums in fp32 storage too; the
    #  ConvTranspose data gradient answers rows = 0 there and the block runs its stand-alone pass)
    h = getattr(x, "_mednet_gn3", None) if FUSE_GN3 else None
    if h is None or h.gn_in is None or h.gn_in.shape != x.shape or h.gn_in.dtype != dtype:
        return None
    return h


class Conv3dFn(Function):
    """nn.Conv3d(k in {1,3}, stride 1, padding k//2)  -- components.py:8-9,44; model.py:77,179."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, ksize, out_planar, out_dtype, want_stats=False):
        L.require_gpu(x, "conv3d")
        ctx.algo = config.conv_algo()
        x = _as_act(x)
        n, cin, d, h, w = x.shape
        cout = weight.shape[0]
        # Cin == 1: NCDHW and NDHWC coincide, so the network input is consumed as it arrives.
        xin = x.contiguous() if cin == 1 else to_cl(x)
        if out_planar:
            y = torch.empty((n, cout, d, h, w), dtype=out_dtype, device=x.device)
        else:
            y = empty_cl(n, cout, d, h, w, out_dtype, x.device)
        # GroupNorm partial sums from the conv epilogue (when the MFMA kernel takes this call): saves a pass over y
        partial = None
        if want_stats and bias is None and not out_planar:
            chunks = L.lib().mednet_conv3d_fused_stats_chunks(n, d, h, w, cin, cout, ksize, L.dt(xin), L.dt(y), config.conv_algo())
            if chunks > 0:
                partial = torch.empty((n, chunks, cout, 2), dtype=torch.float32, device=x.device)
        with profiled_conv(ksize, cin, cout, n, d, h, w):
            L.check(L.lib().mednet_conv3d_fwd(xin.data_ptr(), packed.data_ptr(), L.ptr(bias), y.data_ptr(), n, d, h, w, cin,
                                              cout, ksize, L.dt(xin), L.NDHWC, L.dt(y), L.NCDHW if out_planar else L.NDHWC,
                                              0, config.conv_algo(), L.ptr(partial), L.stream()), "conv3d_fwd")
        ctx.save_for_backward(xin, packed)
        ctx.meta = (ksize, out_planar, cin, cout, bias is not None, x.dtype)
        ctx.params = (weight, bias)
        ctx.gn3 = _gn3_hook_of(x, xin.dtype) if (ksize == 1 and out_planar and xin is x) else None
        if debug.TRACE is not None:
            debug.trace(f"conv3d.fwd k{ksize} {cin}->{cout}", y, partial)
        if not want_stats:
            return y
        if partial is None:
            partial = torch.empty(0, device=x.device)  # "no fused statistics"
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    @_with_algo
    def backward(ctx, dy, _dpartial=None):
        xin, packed = ctx.saved_tensors
        ksize, out_planar, cin, cout, has_bias, x_dtype = ctx.meta
        n, _, d, h, w = xin.shape
        dy = dy.contiguous() if out_planar else to_cl(dy)
        lib = L.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = empty_cl(n, cin, d, h, w, x_dtype, dy.device)
            hook = ctx.gn3
            rows = 0
            if hook is not None and dy.dtype == torch.float32 and dx.dtype == xin.dtype and config.conv_algo() != L.ALGO_DIRECT:
                rows = lib.mednet_head_dgrad_gn_rows(n, d, h, w, cin, L.dt(dx))
            if rows > 0:  # + the first pass of the producing block's GroupNorm-3 backward (xin IS that block's output)
                partial = torch.empty((n, rows, cin, 2), dtype=torch.float32, device=dy.device)
                L.check(lib.mednet_head_dgrad_gn(dy.data_ptr(), packed.data_ptr(), dx.data_ptr(), hook.gn_in.data_ptr(),
                                                 xin.data_ptr(), hook.act, partial.data_ptr(), n, d, h, w, cin, cout, L.dt(dx),
                                                 L.stream()), "head_dgrad_gn")
                hook.offer(dx, partial)
            else:
                L.check(lib.mednet_conv3d_fwd(dy.data_ptr(), packed.data_ptr(), None, dx.data_ptr(), n, d, h, w, cout, cin,
                                              ksize, L.dt(dy), L.NCDHW if out_planar else L.NDHWC, L.dt(dx), L.NDHWC, 1,
                                              config.conv_algo(), None, L.stream()), "conv3d_dgrad")
        direct_w = direct_b = False
        if ctx.needs_input_grad[1]:
            weight, bias = ctx.params
            dw, direct_w = _grad_target(weight, (cout, cin, ksize, ksize, ksize))
            if has_bias and ctx.needs_input_grad[2]:
                db, direct_b = _grad_target(bias, (cout,))
            on_side = SIDE["enabled"] and direct_w and (db is None or direct_b)
            cores = on_side and not out_planar and wgrad_coresident(n, d, h, w, cin, cout, ksize, xin, dy)
            with _OnSide(on_side, dy.device, xin, dy, coresident=cores) as side:
                nbytes = lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, ksize, side.workgroups)
                ws = L.workspace(nbytes, dy.device)
                L.check(lib.mednet_conv3d_wgrad(xin.data_ptr(), dy.data_ptr(), dw.data_ptr(), L.ptr(db), n, d, h, w, cin,
                                                cout, ksize, L.dt(xin), L.NDHWC, L.dt(dy),
                                                L.NCDHW if out_planar else L.NDHWC, config.conv_algo(), side.workgroups,
                                                ws.data_ptr(), ws.numel(), L.stream()), "conv3d_wgrad")
        return dx, (None if direct_w else dw), (None if direct_b else db), None, None, None, None, None


def conv3d(x, weight, bias, packed, ksize, out_planar=False, out_dtype=None):
    return Conv3dFn.apply(x, weight, bias, packed, ksize, out_planar, out_dtype or config.act_dtype())


def conv3d_with_stats(x, weight, bias, packed, ksize):
    """conv + (when the kernel can) the GroupNorm partial sums of its output; returns (y, partial-or-None)."""
    y, partial = Conv3dFn.apply(x, weight, bias, packed, ksize, False, config.act_dtype(), True)
    return y, (partial if partial.numel() else None)


class ConvActFn(Function):
    """conv 3x3x3 (no bias) -> ReLU / LeakyReLU / ELU in ONE kernel (the conv's epilogue): the 'gcr'-style orders of
    components.py:12-67 (UNet3D).  Optionally also the GroupNorm partial sums of the ACTIVATED output, for the GroupNorm
    that opens the next SingleConv.  Backward: activation' from the saved output, then the conv's two gradients."""

    @staticmethod
    def forward(ctx, x, weight, packed, act, want_stats, mask=None):
        L.require_gpu(x, "conv3d+act")
        ctx.algo = config.conv_algo()
        ctx.mask = mask
        xin = to_cl(_as_act(x))
        ctx.gnb = _gnb_hook_of(x, xin.dtype) if xin is x else None
        n, cin, d, h, w = xin.shape
        cout = weight.shape[0]
        lib = L.lib()
        z = empty_cl(n, cout, d, h, w, config.act_dtype(), x.device)
        partial = None
        if want_stats:
            chunks = lib.mednet_conv3d_fused_stats_chunks(n, d, h, w, cin, cout, 3, L.dt(xin), L.dt(z), config.conv_algo())
            if chunks > 0:
                partial = torch.empty((n, chunks, cout, 2), dtype=torch.float32, device=x.device)
        with profiled_conv(3, cin, cout, n, d, h, w):
            L.check(lib.mednet_conv3d_act_fwd(xin.data_ptr(), packed.data_ptr(), z.data_ptr(), n, d, h, w, cin, cout, act,
                                              config.conv_algo(), L.ptr(partial), L.dt(z), L.stream()), "conv3d_act_fwd")
        ctx.save_for_backward(xin, packed, z)
        ctx.act = act
        ctx.weight = weight
        if partial is None:
            partial = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(partial)
        return z, partial

    @staticmethod
    @_with_algo
    def backward(ctx, dz, _dpartial=None):
        from . import block  # (block imports ops)
        xin, packed, z = ctx.saved_tensors
        masked = ctx.mask is not None and ctx.mask.take(dz)  # (before any conversion: identity matters)
        dz = to_cl(dz.to(z.dtype))
        if masked:  # the GroupNorm that consumed z already multiplied its input gradient by act'(z)
            du = dz
        else:
            du = torch.empty_like(z, memory_format=CL)
            L.check(L.lib().mednet_act_bwd(dz.data_ptr(), z.data_ptr(), du.data_ptr(), z.numel(), ctx.act, L.dt(z), L.stream()),
                    "act_bwd")
        hook = ctx.gnb
        dx, dw, partial = block._conv_bwd(xin, du, packed, ctx.weight, ctx.needs_input_grad[0],
                                          gnb=None if hook is None else (hook.gn_in, hook.coef, hook.act))
        if partial is not None:
            hook.offer(dx, partial)
        return dx, dw, None, None, None, None


def conv3d_act_supported(x, cin, cout):
    # mednet_conv3d_act_fwd reads and writes ONE 16-bit type.  An fp32 tensor handed to a bf16-mode layer (a block
    # called stand-alone, a 16/32-channel network input) takes the unfused conv, which passes its dtypes.
    if not (x.is_cuda and config.is_half_mode() and x.dtype == config.act_dtype() and x.dim() == 5):
        return False
    n, _, d, h, w = x.shape
    return bool(L.lib().mednet_conv3d_act_supported(n, d, h, w, cin, cout, config.conv_algo()))


def conv3d_act(x, weight, packed, act, want_stats=False):
    """-> (activated output, GroupNorm partials of it or None)."""
    mask = ActMaskHook(act) if (FUSE_GN3 and torch.is_grad_enabled() and act != L.ACT_NONE) else None
    z, partial = ConvActFn.apply(x, weight, packed, act, want_stats, mask)
    if mask is not None:
        z._mednet_actmask = mask
    return z, (partial if partial.numel() else None)


# ------------------------------------------------------------------------------------------------- ConvTranspose3d (+ skip add)
class ConvT3dFn(Function):
    """nn.ConvTranspose3d(k=3,s=2,p=1,output_padding=1) with the decoder's `x += encoder_features` fused into the
    epilogue  -- components.py:259-264,283-284."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip, packed):
        L.require_gpu(x, "conv_transpose3d")
        ctx.algo = config.conv_algo()
        x0 = x
        x = to_cl(_as_act(x))
        ctx.gn3 = _gn3_hook_of(x0, x.dtype) if x is x0 else None
        n, cin, d, h, w = x.shape
        cout = weight.shape[1]
        y = empty_cl(n, cout, 2 * d, 2 * h, 2 * w, config.act_dtype(), x.device)
        sk = None
        if skip is not None:
            if tuple(skip.shape) != tuple(y.shape):
                raise RuntimeError(f"conv_transpose3d: skip shape {tuple(skip.shape)} != output {tuple(y.shape)}")
            sk = to_cl(skip.to(y.dtype))
        L.check(L.lib().mednet_convt3d_fwd(x.data_ptr(), packed.data_ptr(), L.ptr(bias), L.ptr(sk), y.data_ptr(), n, d, h,
                                           w, cin, cout, L.dt(x), L.dt(y), config.conv_algo(), L.stream()), "convt3d_fwd")
        ctx.save_for_backward(x, packed)
        ctx.meta = (cin, cout, bias is not None, skip is not None, None if skip is None else skip.dtype)
        ctx.params = (weight, bias)
        if debug.TRACE is not None:
            debug.trace(f"convt3d.fwd {cin}->{cout}", y)
        return y

    @staticmethod
    @_with_algo
    def backward(ctx, dy):
        x, packed = ctx.saved_tensors
        cin, cout, has_bias, has_skip, skip_dtype = ctx.meta
        n, _, d, h, w = x.shape
        dy = to_cl(dy)
        lib = L.lib()
        dx = dw = db = dskip = None
        if ctx.needs_input_grad[0]:
            dx = empty_cl(n, cin, d, h, w, x.dtype, dy.device)
            hook = ctx.gn3
            rows = 0
            if hook is not None and dy.dtype == dx.dtype:
                rows = lib.mednet_convt3d_dgrad_gn_rows(n, d, h, w, cin, cout, L.dt(dx), config.conv_algo())
            if rows > 0:  # + the first pass of the producing block's GroupNorm-3 backward (x IS that block's output)
                partial = torch.empty((n, rows, cin, 2), dtype=torch.float32, device=dy.device)
                L.check(lib.mednet_convt3d_dgrad_gn(dy.data_ptr(), packed.data_ptr(), dx.data_ptr(), hook.gn_in.data_ptr(),
                                                    x.data_ptr(), hook.act, partial.data_ptr(), n, d, h, w, cin, cout, L.dt(dx),
                                                    config.conv_algo(), L.stream()), "convt3d_dgrad_gn")
                hook.offer(dx, partial)
            else:
                L.check(lib.mednet_convt3d_dgrad(dy.data_ptr(), packed.data_ptr(), dx.data_ptr(), n, d, h, w, cin, cout,
                                                 L.dt(dy), L.dt(dx), config.conv_algo(), L.stream()), "convt3d_dgrad")
        direct_w = direct_b = False
        if ctx.needs_input_grad[1]:
            weight, bias = ctx.params
            dw, direct_w = _grad_target(weight, (cin, cout, 3, 3, 3))
            if has_bias and ctx.needs_input_grad[2]:
                db, direct_b = _grad_target(bias, (cout,))
            on_side = SIDE["enabled"] and direct_w and (db is None or direct_b)
            with _OnSide(on_side, dy.device, x, dy) as side:
                ws = L.workspace(lib.mednet_convt3d_wgrad_ws_bytes(n, d, h, w, cin, cout, side.workgroups), dy.device)
                L.check(lib.mednet_convt3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), L.ptr(db), n, d, h, w, cin, cout,
                                                 L.dt(x), L.dt(dy), config.conv_algo(), side.workgroups, ws.data_ptr(), ws.numel(),
                                                 L.stream()), "convt3d_wgrad")
        if has_skip and ctx.needs_input_grad[3]:
            dskip = dy if dy.dtype == skip_dtype else dy.to(skip_dtype)
        if debug.TRACE is not None:
            debug.trace(f"convt3d.bwd {cin}->{cout}", dx, ctx.gn3.partial if ctx.gn3 is not None else None)
        return dx, (None if direct_w else dw), (None if direct_b else db), dskip, None


def conv_transpose3d(x, weight, bias, skip, packed):
    return ConvT3dFn.apply(x, weight, bias, skip, packed)


# ------------------------------------------------------------------------------------------------- GroupNorm (+act, +residual)
class GroupNormActFn(Function):
    """z = act(GroupNorm(x) [+ residual])  -- components.py:57, :36-40, :177-178."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, groups, eps, act, partial=None, hook=None):
        L.require_gpu(x, "group_norm_act")
        x0 = x
        x = to_cl(_as_act(x))
        ctx.inmask = getattr(x0, "_mednet_actmask", None) if (x is x0 and FUSE_GN3) else None
        # The fold multiplies dx by act'(x) HERE; the conv layer skips its own activation backward only if the gradient it
        # receives is this very tensor (ActMaskHook.take).  If it declines (x has a second consumer, a tensor hook replaced
        # the gradient) act' is applied twice to this contribution: idempotent for ReLU (a 0/1 mask), wrong for
        # LeakyReLU / ELU -- so only ReLU layers are folded.
        if ctx.inmask is not None and ctx.inmask.act != L.ACT_RELU:
            ctx.inmask = None
        n, c, d, h, w = x.shape
        if c % groups:
            raise RuntimeError(f"group_norm: C={c} not divisible by num_groups={groups}")
        spatial = d * h * w
        lib = L.lib()
        stats = torch.empty((n, groups, 2), dtype=torch.float32, device=x.device)
        coef = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
        ws = L.workspace(lib.mednet_gn_ws_bytes(n, c, spatial), x.device)
        if partial is not None:  # partial sums came out of the producing conv's epilogue
            L.check(lib.mednet_gn_finalize(partial.data_ptr(), partial.shape[1], L.ptr(gamma), L.ptr(beta), stats.data_ptr(),
                                           coef.data_ptr(), n, spatial, c, groups, eps, ws.data_ptr(), ws.numel(),
                                           L.stream()), "gn_finalize")
        else:
            L.check(lib.mednet_gn_stats(x.data_ptr(), L.ptr(gamma), L.ptr(beta), stats.data_ptr(), coef.data_ptr(), n,
                                        spatial, c, groups, eps, L.dt(x), ws.data_ptr(), ws.numel(), L.stream()), "gn_stats")
        res = None
        if residual is not None:
            res = to_cl(residual.to(x.dtype))
        z = torch.empty_like(x, memory_format=CL)
        L.check(lib.mednet_gn_act_fwd(x.data_ptr(), coef.data_ptr(), L.ptr(res), z.data_ptr(), n, spatial, c, act,
                                      L.dt(x), L.dt(z), L.stream()), "gn_act_fwd")
        # the activated output is only read back when a residual entered the pre-activation; otherwise the backward
        # recomputes act' from x and `coef` (one tensor less to read in each of its two passes)
        keep_z = act != L.ACT_NONE and residual is not None
        ctx.save_for_backward(x, z if keep_z else None, stats, gamma, coef)
        ctx.meta = (groups, act, residual is not None, gamma is not None, beta is not None)
        ctx.params = (gamma, beta)
        ctx.gnb = hook
        if hook is not None:
            hook.gn_in, hook.coef, hook.act = x, coef, act
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z, stats, gamma, coef = ctx.saved_tensors
        groups, act, has_res, has_gamma, has_beta = ctx.meta
        n, c, d, h, w = x.shape
        spatial = d * h * w
        fused = ctx.gnb.take(dz) if ctx.gnb is not None else None  # (before any conversion: identity matters)
        dz = to_cl(dz.to(x.dtype))
        lib = L.lib()
        dx = torch.empty_like(x, memory_format=CL)
        dres = torch.empty_like(x, memory_format=CL) if has_res else None
        pg, pb = ctx.params
        dgamma, direct_g = _grad_target(pg, (c,)) if has_gamma else (None, False)
        dbeta, direct_b = _grad_target(pb, (c,)) if has_beta else (None, False)
        ws = L.workspace(lib.mednet_gn_ws_bytes(n, c, spatial), x.device)
        inmask = ctx.inmask  # x is the output of a fused conv -> activation layer: fold act'(x) into dx (ActMaskHook)
        in_act = inmask.act if inmask is not None else L.ACT_NONE
        if inmask is not None:
            inmask.offer(dx)
        if fused is not None and not has_res:  # first pass taken by the conv data gradient that produced dz
            L.check(lib.mednet_gn_act_bwd_fused(dz.data_ptr(), x.data_ptr(), coef.data_ptr(), stats.data_ptr(), L.ptr(gamma),
                                                fused.data_ptr(), fused.shape[1], dx.data_ptr(), L.ptr(dgamma), L.ptr(dbeta), n,
                                                spatial, c, groups, act, in_act, L.dt(x), ws.data_ptr(), ws.numel(), L.stream()),
                    "gn_act_bwd_fused")
            return dx, (None if direct_g else dgamma), (None if direct_b else dbeta), dres, None, None, None, None, None
        L.check(lib.mednet_gn_act_bwd(dz.data_ptr(), None, x.data_ptr(), L.ptr(z), coef.data_ptr(), stats.data_ptr(), L.ptr(gamma),
                                      dx.data_ptr(), L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), n, spatial, c, groups, act, in_act,
                                      L.dt(x), ws.data_ptr(), ws.numel(), L.stream()), "gn_act_bwd")
        return dx, (None if direct_g else dgamma), (None if direct_b else dbeta), dres, None, None, None, None, None


def group_norm_act(x, gamma, beta, groups, eps=1e-5, act=L.ACT_NONE, residual=None, partial=None):
    hook = None
    if FUSE_GN3 and residual is None and config.is_half_mode() and torch.is_grad_enabled() and gamma is not None:
        hook = GNBHook()  # the conv that consumes z may take this GroupNorm's first backward pass (see GNBHook)
    z = GroupNormActFn.apply(x, gamma, beta, residual, groups, eps, act, partial, hook)
    if hook is not None:
        z._mednet_gnb = hook
    return z


# ------------------------------------------------------------------------------------------------- stand-alone activation
class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        L.require_gpu(x, "activation")
        x = _as_act(x)
        xc = x if (x.is_contiguous() or x.is_contiguous(memory_format=CL)) else x.contiguous()
        z = torch.empty_like(xc)  # preserves the memory format
        L.check(L.lib().mednet_act_fwd(xc.data_ptr(), z.data_ptr(), xc.numel(), act, L.dt(xc), L.stream()), "act_fwd")
        ctx.save_for_backward(z)
        ctx.act = act
        return z

    @staticmethod
    def backward(ctx, dz):
        (z,) = ctx.saved_tensors
        if z.dim() == 5 and z.is_contiguous(memory_format=CL) and not z.is_contiguous():
            dz = to_cl(dz)
        else:
            dz = dz.contiguous()
        dz = dz.to(z.dtype)
        dx = torch.empty_like(z)
        L.check(L.lib().mednet_act_bwd(dz.data_ptr(), z.data_ptr(), dx.data_ptr(), z.numel(), ctx.act, L.dt(z),
                                       L.stream()), "act_bwd")
        return dx, None


def activation(x, act):
    return ActFn.apply(x, act)


class AddFn(Function):
    """out = a + b for two same-shape activations (only used when a residual join cannot be folded into a GroupNorm)."""

    @staticmethod
    def forward(ctx, a, b):
        L.require_gpu(a, "add")
        a = to_cl(_as_act(a))
        b = to_cl(b.to(a.dtype))
        out = torch.empty_like(a, memory_format=CL)
        L.check(L.lib().mednet_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), L.dt(a), L.stream()), "add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


# ------------------------------------------------------------------------------------------------- channel split
class _SplitGrad:
    """Shared gradient buffer of a channel split: the loss backward kernels of the two halves write their slices of ONE
    full-size tensor (they index the gradient with the strides of the logits slice they were given), so the split's backward
    is that tensor, not a concatenation."""
    __slots__ = ("shape", "base", "k", "claimed")

    def __init__(self, shape, k):
        self.shape, self.k, self.base = tuple(shape), k, None
        self.claimed = set()  # slice offsets whose view has been handed to a loss backward (one writer per slice)


class SplitChannelsFn(Function):
    """(x[:, :k], x[:, k:]) as ONE autograd node (landmarks.py:71-72 slices the network output into heat-map and class
    channels): backward is the shared buffer both loss kernels wrote into (or, if the gradients came from somewhere else, a
    single concatenation) instead of autograd's zero-fill + copy per slice + add over the full logits tensor."""

    @staticmethod
    def forward(ctx, x, k, holder):
        ctx.k, ctx.shape, ctx.holder = k, x.shape, holder
        return x[:, :k], x[:, k:]

    @staticmethod
    def backward(ctx, d0, d1):
        k, shape, holder = ctx.k, ctx.shape, ctx.holder
        base = holder.base if holder is not None else None
        if holder is not None:
            holder.base = None
            holder.claimed.clear()
        if (base is not None and d0 is not None and d1 is not None and d0.dtype == base.dtype == d1.dtype
                and d0.data_ptr() == base.data_ptr() and d1.data_ptr() == base[:, k:].data_ptr()
                and d0.stride() == base[:, :k].stride() and d1.stride() == base[:, k:].stride()):
            return base, None, None
        ref = d0 if d0 is not None else d1
        if d0 is None:
            d0 = ref.new_zeros((shape[0], k) + tuple(shape[2:]))
        if d1 is None:
            d1 = ref.new_zeros((shape[0], shape[1] - k) + tuple(shape[2:]))
        return torch.cat((d0, d1), dim=1), None, None


SPLIT_SHARED = os.environ.get("MEDNET_SPLIT_SHARED", "1") == "1"  # A/B knob


def split_channels(x, k):
    holder = _SplitGrad(x.shape, k) if (SPLIT_SHARED and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()) else None
    a, b = SplitChannelsFn.apply(x, k, holder)
    if holder is not None:
        a._mednet_split, b._mednet_split = (holder, 0), (holder, k)
    return a, b


def _loss_grad_like(lg, split):
    """fp32 gradient buffer with the STRIDES of the logits view `lg` (the loss backward kernels write with those strides):
    a slice of the shared buffer of a channel split when there is one, else a fresh tensor of that layout."""
    if split is not None:
        holder, c0 = split
        # ONE writer per slice: a second loss on the same slice (Dice + CE on the class logits, a loss applied twice) gets a
        # tensor of its own -- autograd then sums the two, and SplitChannelsFn.backward sees a gradient that is not the
        # shared view and takes its concatenation path.  (Two kernels writing one view would leave 2*G2, not G1+G2.)
        if c0 not in holder.claimed:
            if holder.base is None:
                holder.base = torch.empty(holder.shape, dtype=torch.float32, device=lg.device)
            view = holder.base[:, c0:c0 + lg.shape[1]]
            if tuple(view.shape) == tuple(lg.shape) and view.stride() == lg.stride():
                holder.claimed.add(c0)
                return view
    return torch.empty_strided(tuple(lg.shape), lg.stride(), dtype=torch.float32, device=lg.device)


# ------------------------------------------------------------------------------------------------- pooling
class Pool2Fn(Function):
    """nn.MaxPool3d(2) / nn.AvgPool3d(2)  -- components.py:208-212."""

    @staticmethod
    def forward(ctx, x, mode):
        L.require_gpu(x, "pool3d")
        x0 = x
        x = to_cl(_as_act(x))
        n, c, d, h, w = x.shape
        stash = getattr(x0, "_mednet_pooled", None) if x is x0 else None
        # (the stash pooled the values the block WROTE: an in-place change of the output since then -- a forward hook's clamp_(),
        #  user code between the levels -- shows in the version counter, and the pooling kernel runs on the current values)
        if (stash is not None and stash.mode == mode and stash.pooled is not None and stash.pooled.dtype == x.dtype
                and stash.version == x0._version):
            y, stash.pooled = stash.pooled, None  # written by the producing block's last apply pass (block.PoolStash)
        else:
            y = empty_cl(n, c, d // 2, h // 2, w // 2, x.dtype, x.device)
            L.check(L.lib().mednet_pool2_fwd(x.data_ptr(), y.data_ptr(), n, d, h, w, c, mode, L.dt(x), L.stream()), "pool2_fwd")
        ctx.save_for_backward(x)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, d, h, w = x.shape
        dy = to_cl(dy.to(x.dtype))
        dx = torch.empty_like(x, memory_format=CL)
        L.check(L.lib().mednet_pool2_bwd(dy.data_ptr(), x.data_ptr(), None, dx.data_ptr(), n, d, h, w, c, ctx.mode, L.dt(x),
                                         L.stream()), "pool2_bwd")
        return dx, None


def pool2(x, mode=L.POOL_MAX):
    return Pool2Fn.apply(x, mode)


class SkipPool2Fn(Function):
    """(skip, pooled) = (x, pool(x)): the encoder output that feeds both the next level's pooling and the decoder's
    skip join (model.py:194-205).  As ONE node its backward receives both gradients and sums them inside the pooling
    backward kernel, instead of autograd adding two full-resolution tensors with a separate kernel."""

    @staticmethod
    def forward(ctx, x, mode, sole_consumer=False):
        L.require_gpu(x, "pool3d")
        x0 = x
        x = to_cl(_as_act(x))
        n, c, d, h, w = x.shape
        stash = getattr(x0, "_mednet_pooled", None) if x is x0 else None
        # (the stash pooled the values the block WROTE: an in-place change of the output since then -- a forward hook's clamp_(),
        #  user code between the levels -- shows in the version counter, and the pooling kernel runs on the current values)
        if (stash is not None and stash.mode == mode and stash.pooled is not None and stash.pooled.dtype == x.dtype
                and stash.version == x0._version):
            y, stash.pooled = stash.pooled, None  # written by the producing block's last apply pass (block.PoolStash)
        else:
            y = empty_cl(n, c, d // 2, h // 2, w // 2, x.dtype, x.device)
            L.check(L.lib().mednet_pool2_fwd(x.data_ptr(), y.data_ptr(), n, d, h, w, c, mode, L.dt(x), L.stream()), "pool2_fwd")
        ctx.save_for_backward(x)
        ctx.mode = mode
        ctx.gn3 = _gn3_hook_of(x0, x.dtype) if x is x0 else None
        # x is the output of a fused conv -> activation layer (UNet3D's blocks): the join below folds act'(x) into the gradient it
        # produces and that layer's backward skips its activation pass (ActMaskHook) -- for even extents (mednet_pool2_bwd_act)
        ctx.inmask = getattr(x0, "_mednet_actmask", None) if (x is x0 and FUSE_GN3 and POOL_ACT_MASK and not ((d | h | w) & 1)) else None
        # sole_consumer: the caller (the U-Net's own forward) hands x to nothing else, so the gradient of x goes to the producing
        # block and nowhere else -- the backward may then leave it unwritten and let the block's apply pass rebuild it (LAZY_POOL)
        # (a tensor hook or retain_grad() on x would READ that gradient: then it is written as before)
        ctx.lazy_ok = (bool(sole_consumer) and LAZY_POOL and ctx.gn3 is not None and not x0._backward_hooks
                       and not getattr(x0, "retains_grad", False))
        if debug.TRACE is not None:
            debug.trace(f"skip_pool2.fwd c{c}", y)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dskip, dy):
        (x,) = ctx.saved_tensors
        n, c, d, h, w = x.shape
        if dy is None:
            return dskip, None, None
        dy = to_cl(dy.to(x.dtype))
        hook = ctx.gn3
        add_c = 0
        if dskip is not None:
            # (UNet3D: the skip gradient is the leading channels of the concatenation's gradient -- read in place by the plain join)
            add_c = _channel_slice_of_cl(dskip, c) if (hook is None and dskip.dtype == x.dtype and c % 8 == 0 and not ((d | h | w) & 1)) else 0
            if add_c == 0 or add_c % 8:
                add_c = 0
                dskip = to_cl(dskip.to(x.dtype))
        dx = torch.empty_like(x, memory_format=CL)
        rows = L.lib().mednet_pool2_bwd_gn_rows(n, d, h, w, c, L.dt(x)) if hook is not None else 0
        if rows > 0:  # + the first pass of the producing block's GroupNorm-3 backward (x IS that block's output)
            partial = torch.empty((n, rows, c, 2), dtype=torch.float32, device=x.device)
            lazy = ctx.lazy_ok and bool(L.lib().mednet_gn_act_pool_supported(d, h, w, c, L.dt(x)))
            # lazy: sums only -- dx stays an unwritten placeholder (autograd needs a tensor of the right shape to hand on), the
            # block's GroupNorm-3 apply pass rebuilds its rows from dy, the arg-max of x and dskip (mednet_gn_act_bwd_fused_res_pool)
            L.check(L.lib().mednet_pool2_bwd_gn(dy.data_ptr(), x.data_ptr(), L.ptr(dskip), None if lazy else dx.data_ptr(),
                                                hook.gn_in.data_ptr(), hook.act, partial.data_ptr(), n, d, h, w, c, ctx.mode, L.dt(x),
                                                L.stream()), "pool2_bwd_gn")
            hook.offer(dx, partial, (dy, dskip, ctx.mode) if lazy else None)
            if debug.TRACE is not None:
                debug.trace(f"skip_pool2.bwd c{c}", None if lazy else dx, partial)
            return dx, None, None
        mask = ctx.inmask
        L.check(L.lib().mednet_pool2_bwd_act(dy.data_ptr(), x.data_ptr(), L.ptr(dskip), add_c, dx.data_ptr(), n, d, h, w, c, ctx.mode,
                                             L.ACT_NONE if mask is None else mask.act, L.dt(x), L.stream()), "pool2_bwd")
        if mask is not None:
            mask.offer(dx)
        if debug.TRACE is not None:
            debug.trace(f"skip_pool2.bwd c{c}", dx)
        return dx, None, None


POOL_ACT_MASK = os.environ.get("MEDNET_POOL_ACT_MASK", "1") == "1"  # A/B knob
LAZY_POOL = os.environ.get("MEDNET_LAZY_POOL", "1") == "1"  # A/B knob


def skip_pool2(x, mode=L.POOL_MAX, sole_consumer=False):
    """(skip, pooled).  `sole_consumer`: the caller guarantees that x -- a block output -- is used by nothing but this call (the
    returned `skip` IS the tensor to use as x from here on); the backward then need not materialise the gradient of x."""
    return SkipPool2Fn.apply(x, mode, sole_consumer)


# ------------------------------------------------------------------------------------------------- nearest upsample + concat
class UpCatFn(Function):
    """F.interpolate(x, size=enc.shape[2:], mode='nearest') ; torch.cat((enc, x), 1)  -- components.py:277-280."""

    @staticmethod
    def forward(ctx, enc, x, want_stats=False):
        L.require_gpu(x, "upsample_concat")
        enc = to_cl(_as_act(enc))
        x = to_cl(_as_act(x)).to(enc.dtype)
        n, ce, d, h, w = enc.shape
        _, cx, xd, xh, xw = x.shape
        out = empty_cl(n, ce + cx, d, h, w, enc.dtype, enc.device)
        lib = L.lib()
        chunks = lib.mednet_upcat_stats_chunks(n, d, h, w, ce, cx, L.dt(enc)) if want_stats else 0
        if chunks > 0:  # + the GroupNorm partial sums of the concatenated tensor (the 'g c r' block that follows opens with one)
            partial = torch.empty((n, chunks, ce + cx, 2), dtype=torch.float32, device=enc.device)
            L.check(lib.mednet_upcat_fwd_stats(enc.data_ptr(), x.data_ptr(), out.data_ptr(), partial.data_ptr(), n, d, h, w, ce, xd, xh,
                                               xw, cx, L.dt(enc), L.stream()), "upcat_fwd_stats")
        else:
            partial = torch.empty(0, device=enc.device)
            L.check(lib.mednet_upcat_fwd(enc.data_ptr(), x.data_ptr(), out.data_ptr(), n, d, h, w, ce, xd, xh, xw, cx,
                                         L.dt(enc), L.stream()), "upcat_fwd")
        ctx.dims = (n, ce, d, h, w, cx, xd, xh, xw)
        if not want_stats:
            return out
        ctx.mark_non_differentiable(partial)
        return out, partial

    @staticmethod
    def backward(ctx, dout, _dpartial=None):
        n, ce, d, h, w, cx, xd, xh, xw = ctx.dims
        dout = to_cl(dout)
        # the encoder part of the gradient is a VIEW of dout (its leading channels): the pooling join that consumes it reads it
        # where it lies (SkipPool2Fn, mednet_pool2_bwd_act's add_channels); any other consumer makes it dense itself
        view = UPCAT_VIEW and ce % 8 == 0 and cx % 8 == 0
        denc = dout.narrow(1, 0, ce) if view else empty_cl(n, ce, d, h, w, dout.dtype, dout.device)
        dx = empty_cl(n, cx, xd, xh, xw, dout.dtype, dout.device)
        L.check(L.lib().mednet_upcat_bwd(dout.data_ptr(), None if view else denc.data_ptr(), dx.data_ptr(), n, d, h, w, ce, xd, xh,
                                         xw, cx, L.dt(dout), L.stream()), "upcat_bwd")
        return denc, dx, None


UPCAT_VIEW = os.environ.get("MEDNET_UPCAT_VIEW", "1") == "1"  # A/B knob


def _channel_slice_of_cl(t, c):
    """Channels per voxel of the channels-last tensor `t` is a leading-channel slice of (0: `t` is not such a view)."""
    if t.dim() != 5 or t.shape[1] != c:
        return 0
    n, _, d, h, w = t.shape
    sn, sc, sd, sh, sw = t.stride()
    ct = sw
    if sc == 1 and ct > c and sh == w * ct and sd == h * w * ct and sn == d * h * w * ct:
        return ct
    return 0


def upsample_concat(enc, x, want_stats=False):
    """cat((enc, nearest-upsampled x), 1); with want_stats -> (tensor, GroupNorm partial sums of it or None)."""
    if not want_stats:
        return UpCatFn.apply(enc, x)
    out, partial = UpCatFn.apply(enc, x, True)
    return out, (partial if partial.numel() else None)


# ------------------------------------------------------------------------------------------------- losses
def _planar_logits(logits: torch.Tensor):
    """fp32 logits whose (D,H,W) block is dense: returns (tensor, stride_n, stride_c)."""
    t = logits if logits.dtype == torch.float32 else logits.float()
    sp = t.shape[2:]
    dense = []
    acc = 1
    for s in reversed(sp):
        dense.insert(0, acc)
        acc *= s
    if tuple(t.stride()[2:]) != tuple(dense):
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


def _labels_i64(labels: torch.Tensor, shape):
    if labels.dtype != torch.int64:
        labels = labels.long()
    if tuple(labels.shape) != tuple(shape):
        raise AssertionError("'input' and 'target' must have the same shape")
    return labels.contiguous()


class DiceLossFn(Function):
    """DiceLoss.forward  -- loss.py:114-130 (softmax|sigmoid, one-hot, per-channel dice over the WHOLE batch)."""

    @staticmethod
    def forward(ctx, logits, labels, weight, eps, sigmoid, ignore_index):
        L.require_gpu(logits, "dice_loss")
        lg, sn, sc = _planar_logits(logits)
        n, c = lg.shape[:2]
        spatial = lg[0, 0].numel()
        lab, lab_sn, lab_dt = _label_view(labels, n, tuple(lg.shape[2:]))  # uint8 / int64 as they lie (no cast kernel)
        wt = None if weight is None else weight.to(device=lg.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        saved = torch.empty((c, 2), dtype=torch.float32, device=lg.device)
        lib = L.lib()
        ws = L.workspace(lib.mednet_loss_ws_bytes(n, c, spatial), lg.device)
        ii = L.NO_IGNORE if ignore_index is None else int(ignore_index)
        L.check(lib.mednet_dice_fwd_lt(lg.data_ptr(), lab.data_ptr(), lab_dt, lab_sn, L.ptr(wt), loss.data_ptr(), saved.data_ptr(),
                                       None, n, c, spatial, sn, sc, eps, int(sigmoid), ii, ws.data_ptr(), ws.numel(), L.stream()),
                "dice_fwd")
        ctx.save_for_backward(lg, lab, wt, saved)
        ctx.meta = (eps, int(sigmoid), ii, sn, sc, logits.dtype, lab_sn, lab_dt)
        ctx.split = getattr(logits, "_mednet_split", None) if lg is logits else None
        if debug.TRACE is not None:
            debug.trace("dice.fwd", lg, loss, saved)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lg, lab, wt, saved = ctx.saved_tensors
        eps, sigmoid, ii, sn, sc, in_dtype, lab_sn, lab_dt = ctx.meta
        n, c = lg.shape[:2]
        spatial = lg[0, 0].numel()
        dl = dloss.to(torch.float32).contiguous()
        dlogits = _loss_grad_like(lg, ctx.split)
        L.check(L.lib().mednet_dice_bwd_lt(lg.data_ptr(), lab.data_ptr(), lab_dt, lab_sn, L.ptr(wt), saved.data_ptr(), dl.data_ptr(),
                                           dlogits.data_ptr(), n, c, spatial, sn, sc, eps, sigmoid, ii, L.stream()),
                "dice_bwd")
        if debug.TRACE is not None:
            debug.trace("dice.bwd", dlogits)
        return dlogits.to(in_dtype), None, None, None, None, None


def dice_loss(logits, labels, weight=None, eps=1e-5, sigmoid=False, ignore_index=None):
    return DiceLossFn.apply(logits, labels, weight, eps, sigmoid, ignore_index)


# ------------------------------------------------------------------------------------------------- 1x1x1 head + Dice, one node
FUSE_HEAD_LOSS = os.environ.get("MEDNET_FUSE_HEAD_LOSS", "1") == "1"  # A/B knob


def _label_view(labels: torch.Tensor, n: int, spatial_shape):
    """Labels as the fused kernels take them: uint8 or int64, N x spatial, each sample's block dense; the stride between samples
    is free (the last channel of a uint8 N x C x D x H x W label volume is consumed where it lies, segmentation.py:60)."""
    if labels.dtype not in (torch.uint8, torch.int64):
        labels = labels.long()
    if tuple(labels.shape) != (n,) + tuple(spatial_shape):
        raise AssertionError("'input' and 'target' must have the same shape")
    dense, acc = [], 1
    for sdim in reversed(spatial_shape):
        dense.insert(0, acc)
        acc *= sdim
    if tuple(labels.stride()[1:]) != tuple(dense) or (n > 1 and labels.stride(0) < acc):
        labels = labels.contiguous()
    return labels, (labels.stride(0) if n > 1 else acc), (L.U8 if labels.dtype == torch.uint8 else L.I64)


def head_dice_supported(x: torch.Tensor, cin: int, cout: int, labels: torch.Tensor) -> bool:
    if not (FUSE_HEAD_LOSS and x.is_cuda and x.dim() == 5 and labels.is_cuda):
        return False
    if x.dtype != config.act_dtype() or not x.is_contiguous(memory_format=CL):
        return False
    ld = L.U8 if labels.dtype == torch.uint8 else L.I64
    return bool(L.lib().mednet_head_dice_supported(cin, cout, L.dt(x), ld))


class HeadDiceFn(Function):
    """logits = final_conv(x) (model.py:207, 1x1x1 with bias, planar fp32) and loss = DiceLoss(logits, labels) (loss.py:114-130)
    as ONE autograd node: forward is one pass over the features (mednet_head_dice_fwd), backward one pass that produces the
    feature gradient, the head's weight and bias gradients and the first pass of the producing block's GroupNorm-3 backward
    without ever storing the logit gradient (mednet_head_dice_bwd).  Returns (logits, loss); the logits are a result for the
    caller (metrics, `outputs`), not a differentiable output of this node."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, labels, loss_weight, eps, sigmoid, ignore_index):
        L.require_gpu(x, "head_dice")
        n, cin, d, h, w = x.shape
        cout = weight.shape[0]
        spatial = d * h * w
        lab, lab_sn, lab_dt = _label_view(labels, n, (d, h, w))
        wt = None if loss_weight is None else loss_weight.to(device=x.device, dtype=torch.float32).contiguous()
        logits = torch.empty((n, cout, d, h, w), dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        saved = torch.empty((cout, 2), dtype=torch.float32, device=x.device)
        lib = L.lib()
        ws = L.workspace(lib.mednet_head_dice_ws_bytes(n, spatial, cin, cout), x.device)
        ii = L.NO_IGNORE if ignore_index is None else int(ignore_index)
        L.check(lib.mednet_head_dice_fwd(x.data_ptr(), packed.data_ptr(), L.ptr(bias), lab.data_ptr(), lab_dt, lab_sn, L.ptr(wt),
                                         logits.data_ptr(), loss.data_ptr(), saved.data_ptr(), n, spatial, cin, cout, eps,
                                         int(sigmoid), ii, L.dt(x), ws.data_ptr(), ws.numel(), L.stream()), "head_dice_fwd")
        ctx.save_for_backward(x, packed, logits, lab, wt, saved)
        ctx.meta = (eps, int(sigmoid), ii, lab_sn, lab_dt, cin, cout)
        ctx.params = (weight, bias)
        ctx.gn3 = _gn3_hook_of(x, x.dtype)
        # x is the output of a fused conv -> activation layer (UNet3D's last block): the backward folds act'(x) into the feature
        # gradient it stores and that layer skips its activation pass (ActMaskHook), as SkipPool2Fn does for the encoder levels
        ctx.inmask = getattr(x, "_mednet_actmask", None) if (ctx.gn3 is None and FUSE_GN3 and POOL_ACT_MASK) else None
        ctx.mark_non_differentiable(logits)
        if debug.TRACE is not None:
            debug.trace("head_dice.fwd", logits, loss, saved)
        return logits, loss

    @staticmethod
    def backward(ctx, _dlogits, dloss):
        x, packed, logits, lab, wt, saved = ctx.saved_tensors
        eps, sigmoid, ii, lab_sn, lab_dt, cin, cout = ctx.meta
        weight, bias = ctx.params
        n, _, d, h, w = x.shape
        spatial = d * h * w
        lib = L.lib()
        dl = dloss.to(torch.float32).contiguous()
        dx = torch.empty_like(x, memory_format=CL)
        dw, direct_w = _grad_target(weight, (cout, cin, 1, 1, 1))
        db, direct_b = (None, True) if bias is None else _grad_target(bias, (cout,))
        hook = ctx.gn3
        partial = None
        if hook is not None:
            partial = torch.empty((n, lib.mednet_head_dice_gn_rows(n, spatial, cin), cin, 2), dtype=torch.float32, device=x.device)
        ws = L.workspace(lib.mednet_head_dice_ws_bytes(n, spatial, cin, cout), x.device)
        L.check(lib.mednet_head_dice_bwd(logits.data_ptr(), lab.data_ptr(), lab_dt, lab_sn, packed.data_ptr(), L.ptr(wt),
                                         saved.data_ptr(), dl.data_ptr(), dx.data_ptr(),
                                         None if hook is None else hook.gn_in.data_ptr(), x.data_ptr(),
                                         hook.act if hook is not None else (ctx.inmask.act if ctx.inmask is not None else 0),
                                         L.ptr(partial), dw.data_ptr(), L.ptr(db), n, spatial, cin,
                                         cout, eps, sigmoid, ii, L.dt(x), ws.data_ptr(), ws.numel(), L.stream()), "head_dice_bwd")
        if hook is not None:
            hook.offer(dx, partial)
        elif ctx.inmask is not None:
            ctx.inmask.offer(dx)
        if debug.TRACE is not None:
            debug.trace("head_dice.bwd", dx, partial, dw, db)
        return dx, (None if direct_w else dw), (None if (bias is None or direct_b) else db), None, None, None, None, None, None


def head_dice(x, weight, bias, packed, labels, loss_weight=None, eps=1e-5, sigmoid=False, ignore_index=None):
    """-> (logits N x C x D x H x W fp32, Dice loss)."""
    return HeadDiceFn.apply(x, weight, bias, packed, labels, loss_weight, eps, sigmoid, ignore_index)


def _heatmap_view(heatmaps: torch.Tensor, n: int, nh: int, spatial_shape):
    """uint8 heat-map targets as the fused landmark head takes them: N x nh x spatial, channels dense, any stride between samples
    (the first channels of a uint8 label volume are consumed where they lie, landmarks.py:69)."""
    if tuple(heatmaps.shape) != (n, nh) + tuple(spatial_shape):
        raise RuntimeError(f"heatmap_loss: target shape {tuple(heatmaps.shape)} != output {(n, nh) + tuple(spatial_shape)}")
    dense, acc = [], 1
    for sdim in reversed((nh,) + tuple(spatial_shape)):
        dense.insert(0, acc)
        acc *= sdim
    if tuple(heatmaps.stride()[1:]) != tuple(dense) or (n > 1 and heatmaps.stride(0) < acc):
        heatmaps = heatmaps.contiguous()
    return heatmaps, (heatmaps.stride(0) if n > 1 else acc)


def head_landmark_supported(x: torch.Tensor, cin: int, nh: int, ncls: int, heatmaps: torch.Tensor, labels: torch.Tensor) -> bool:
    if not (FUSE_HEAD_LOSS and x.is_cuda and x.dim() == 5 and labels.is_cuda and heatmaps.is_cuda):
        return False
    if x.dtype != config.act_dtype() or x.dtype == torch.float32 or not x.is_contiguous(memory_format=CL):
        return False
    if heatmaps.dtype != torch.uint8 or labels.dtype != torch.uint8:
        return False
    # (a network whose last block ends in conv -> activation hands the head an activation mask to fold in: the stock path does that)
    if getattr(x, "_mednet_actmask", None) is not None and _gn3_hook_of(x, x.dtype) is None:
        return False
    return bool(L.lib().mednet_head_landmark_supported(cin, nh, ncls, L.dt(x), x[0, 0].numel()))


class HeadLandmarkFn(Function):
    """LandmarkNet's head and loss (landmarks.py:66-83, 125-134) as ONE autograd node in the 16-bit storage modes:
    outputs = final_conv(x) (model.py:207), class_loss = DiceLoss(outputs[:, nh:], labels), regression_loss = sum_c w_c *
    mean f(outputs[:, c] - heatmaps[:, c]).  Forward: one matrix-core pass over the features that writes nothing but loss partials
    (mednet_head_landmark_fwd); backward: one pass that rebuilds the logits the same way and produces the feature gradient, the
    head's weight / bias gradients and the first pass of the producing block's GroupNorm-3 backward
    (mednet_head_landmark_bwd).  No logit or logit-gradient tensor exists.  Returns (class_loss, regression_loss)."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, heatmaps, labels, class_weight, reg_weight, kind, eps, sigmoid, ignore_index):
        L.require_gpu(x, "head_landmark")
        n, cin, d, h, w = x.shape
        nh = heatmaps.shape[1]
        ncls = weight.shape[0] - nh
        spatial = d * h * w
        hm, hm_sn = _heatmap_view(heatmaps, n, nh, (d, h, w))
        lab, lab_sn, _ = _label_view(labels, n, (d, h, w))
        # launch_head_lm_* reads targets and labels four voxels at a time: base pointers and sample strides must be multiples of 4
        # bytes.  A view with an odd storage offset / stride (e.g. a label volume sliced at an odd channel offset) is copied once.
        if hm.data_ptr() % 4 or (n > 1 and hm_sn % 4):
            hm = hm.contiguous().clone() if hm.is_contiguous() else hm.contiguous()
            hm_sn = nh * spatial
        if lab.data_ptr() % 4 or (n > 1 and lab_sn % 4):
            lab = lab.contiguous().clone() if lab.is_contiguous() else lab.contiguous()
            lab_sn = spatial
        cw = None if class_weight is None else class_weight.to(device=x.device, dtype=torch.float32).contiguous()
        rw = None if reg_weight is None else torch.as_tensor(reg_weight, dtype=torch.float32, device=x.device).contiguous()
        closs = torch.empty((), dtype=torch.float32, device=x.device)
        rloss = torch.empty((), dtype=torch.float32, device=x.device)
        saved = torch.empty((ncls, 2), dtype=torch.float32, device=x.device)
        lib = L.lib()
        ws = L.workspace(lib.mednet_head_landmark_ws_bytes(n, spatial, nh, ncls), x.device)
        ii = L.NO_IGNORE if ignore_index is None else int(ignore_index)
        L.check(lib.mednet_head_landmark_fwd(x.data_ptr(), packed.data_ptr(), L.ptr(bias), hm.data_ptr(), hm_sn, lab.data_ptr(), lab_sn,
                                             L.ptr(cw), L.ptr(rw), None, closs.data_ptr(), rloss.data_ptr(), saved.data_ptr(), n,
                                             spatial, cin, nh, ncls, kind, eps, int(sigmoid), ii, L.dt(x), ws.data_ptr(), ws.numel(),
                                             L.stream()), "head_landmark_fwd")
        ctx.save_for_backward(x, packed, hm, lab, cw, rw, saved)
        ctx.meta = (kind, eps, int(sigmoid), ii, hm_sn, lab_sn, cin, nh, ncls)
        ctx.params = (weight, bias)
        ctx.gn3 = _gn3_hook_of(x, x.dtype)
        if debug.TRACE is not None:
            debug.trace("head_landmark.fwd", closs, rloss, saved)
        return closs, rloss

    @staticmethod
    def backward(ctx, dclass, dreg):
        x, packed, hm, lab, cw, rw, saved = ctx.saved_tensors
        kind, eps, sigmoid, ii, hm_sn, lab_sn, cin, nh, ncls = ctx.meta
        weight, bias = ctx.params
        n, _, d, h, w = x.shape
        spatial = d * h * w
        lib = L.lib()
        zero = None
        if dclass is None or dreg is None:
            zero = torch.zeros((), dtype=torch.float32, device=x.device)
        dc = zero if dclass is None else dclass.to(torch.float32).contiguous()
        dr = zero if dreg is None else dreg.to(torch.float32).contiguous()
        dx = torch.empty_like(x, memory_format=CL)
        dw, direct_w = _grad_target(weight, (nh + ncls, cin, 1, 1, 1))
        db, direct_b = (None, True) if bias is None else _grad_target(bias, (nh + ncls,))
        hook = ctx.gn3
        partial = None
        if hook is not None:
            partial = torch.empty((n, lib.mednet_head_landmark_gn_rows(spatial), cin, 2), dtype=torch.float32, device=x.device)
        ws = L.workspace(lib.mednet_head_landmark_ws_bytes(n, spatial, nh, ncls), x.device)
        L.check(lib.mednet_head_landmark_bwd(x.data_ptr(), packed.data_ptr(), L.ptr(bias), hm.data_ptr(), hm_sn, lab.data_ptr(), lab_sn,
                                             L.ptr(cw), L.ptr(rw), saved.data_ptr(), dc.data_ptr(), dr.data_ptr(), dx.data_ptr(),
                                             None if hook is None else hook.gn_in.data_ptr(), hook.act if hook is not None else 0,
                                             L.ptr(partial), dw.data_ptr(), L.ptr(db), n, spatial, cin, nh, ncls, kind, eps, sigmoid,
                                             ii, L.dt(x), ws.data_ptr(), ws.numel(), L.stream()), "head_landmark_bwd")
        if hook is not None:
            hook.offer(dx, partial)
        if debug.TRACE is not None:
            debug.trace("head_landmark.bwd", dx, partial, dw, db)
        return (dx, (None if direct_w else dw), (None if (bias is None or direct_b) else db)) + (None,) * 9


def head_landmark(x, weight, bias, packed, heatmaps, labels, class_weight=None, reg_weight=None, kind="L2", eps=1e-5, sigmoid=False,
                  ignore_index=None):
    """-> (class_loss, regression_loss) of LandmarkNet.loss on final_conv(x)."""
    return HeadLandmarkFn.apply(x, weight, bias, packed, heatmaps, labels, class_weight, reg_weight,
                                L.REG_L2 if kind == "L2" else L.REG_L1, eps, sigmoid, ignore_index)


def per_channel_dice(logits, labels, weight=None, eps=1e-5, sigmoid=False, ignore_index=None):
    """dice_metric  -- loss.py:51-55 (no gradient)."""
    L.require_gpu(logits, "dice_metric")
    lg, sn, sc = _planar_logits(logits.detach())
    n, c = lg.shape[:2]
    spatial = lg[0, 0].numel()
    lab, lab_sn, lab_dt = _label_view(labels, n, tuple(lg.shape[2:]))
    wt = None if weight is None else weight.to(device=lg.device, dtype=torch.float32).contiguous()
    saved = torch.empty((c, 2), dtype=torch.float32, device=lg.device)
    dice = torch.empty((c,), dtype=torch.float32, device=lg.device)
    lib = L.lib()
    ws = L.workspace(lib.mednet_loss_ws_bytes(n, c, spatial), lg.device)
    ii = L.NO_IGNORE if ignore_index is None else int(ignore_index)
    L.check(lib.mednet_dice_fwd_lt(lg.data_ptr(), lab.data_ptr(), lab_dt, lab_sn, L.ptr(wt), None, saved.data_ptr(), dice.data_ptr(),
                                   n, c, spatial, sn, sc, eps, int(sigmoid), ii, ws.data_ptr(), ws.numel(), L.stream()), "dice_fwd")
    return dice


class CrossEntropyFn(Function):
    """nn.CrossEntropyLoss(weight)(logits, labels), mean reduction  -- segmentation.py:49; landmarks.py:49."""

    @staticmethod
    def forward(ctx, logits, labels, weight, ignore_index):
        L.require_gpu(logits, "cross_entropy")
        lg, sn, sc = _planar_logits(logits)
        n, c = lg.shape[:2]
        spatial = lg[0, 0].numel()
        lab = _labels_i64(labels, (n,) + tuple(lg.shape[2:]))
        wt = None if weight is None else weight.to(device=lg.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        saved = torch.empty((2,), dtype=torch.float32, device=lg.device)
        lib = L.lib()
        ws = L.workspace(lib.mednet_loss_ws_bytes(n, c, spatial), lg.device)
        L.check(lib.mednet_ce_fwd(lg.data_ptr(), lab.data_ptr(), L.ptr(wt), loss.data_ptr(), saved.data_ptr(), n, c,
                                  spatial, sn, sc, int(ignore_index), ws.data_ptr(), ws.numel(), L.stream()), "ce_fwd")
        ctx.save_for_backward(lg, lab, wt, saved)
        ctx.meta = (int(ignore_index), sn, sc, logits.dtype)
        ctx.split = getattr(logits, "_mednet_split", None) if lg is logits else None
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lg, lab, wt, saved = ctx.saved_tensors
        ii, sn, sc, in_dtype = ctx.meta
        n, c = lg.shape[:2]
        spatial = lg[0, 0].numel()
        dl = dloss.to(torch.float32).contiguous()
        dlogits = _loss_grad_like(lg, ctx.split)
        L.check(L.lib().mednet_ce_bwd(lg.data_ptr(), lab.data_ptr(), L.ptr(wt), saved.data_ptr(), dl.data_ptr(),
                                      dlogits.data_ptr(), n, c, spatial, sn, sc, ii, L.stream()), "ce_bwd")
        return dlogits.to(in_dtype), None, None, None


def cross_entropy(logits, labels, weight=None, ignore_index=-100):
    return CrossEntropyFn.apply(logits, labels, weight, ignore_index)


class HeatmapLossFn(Function):
    """sum_c w_c * mean_{n,v} f(out[:,c] - hm[:,c])  -- landmarks.py:129-132 (f = square | abs)."""

    @staticmethod
    def forward(ctx, out, target, cweight, kind):
        L.require_gpu(out, "heatmap_loss")
        lg, sn, sc = _planar_logits(out)
        n, c = lg.shape[:2]
        spatial = lg[0, 0].numel()
        if tuple(target.shape) != tuple(lg.shape):
            raise RuntimeError(f"heatmap_loss: target shape {tuple(target.shape)} != output {tuple(lg.shape)}")
        tgt = target if target.dtype in (torch.uint8, torch.float32) else target.float()
        # channels dense, any stride between samples: the heat-map channels of a label volume are read where they lie
        dense, acc = [], 1
        for sdim in reversed(tuple(tgt.shape[1:])):
            dense.insert(0, acc)
            acc *= sdim
        if tuple(tgt.stride()[1:]) != tuple(dense) or (n > 1 and tgt.stride(0) < c * spatial):
            tgt = tgt.contiguous()
        tgt_sn = tgt.stride(0) if n > 1 else c * spatial
        wt = None if cweight is None else torch.as_tensor(cweight, dtype=torch.float32, device=lg.device).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        lib = L.lib()
        ws = L.workspace(lib.mednet_loss_ws_bytes(n, c, spatial), lg.device)
        u8 = int(tgt.dtype == torch.uint8)
        L.check(lib.mednet_heatmap_loss_fwd_strided(lg.data_ptr(), tgt.data_ptr(), tgt_sn, L.ptr(wt), loss.data_ptr(), n, c, spatial,
                                                    sn, sc, kind, u8, ws.data_ptr(), ws.numel(), L.stream()), "heatmap_loss_fwd")
        ctx.save_for_backward(lg, tgt, wt)
        ctx.meta = (kind, u8, sn, sc, out.dtype, tgt_sn)
        ctx.split = getattr(out, "_mednet_split", None) if lg is out else None
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lg, tgt, wt = ctx.saved_tensors
        kind, u8, sn, sc, in_dtype, tgt_sn = ctx.meta
        n, c = lg.shape[:2]
        spatial = lg[0, 0].numel()
        dl = dloss.to(torch.float32).contiguous()
        dout = _loss_grad_like(lg, ctx.split)
        L.check(L.lib().mednet_heatmap_loss_bwd_strided(lg.data_ptr(), tgt.data_ptr(), tgt_sn, L.ptr(wt), dl.data_ptr(), dout.data_ptr(),
                                                        n, c, spatial, sn, sc, kind, u8, L.stream()), "heatmap_loss_bwd")
        return dout.to(in_dtype), None, None, None


def heatmap_loss(out, target, channel_weights=None, kind="L2"):
    return HeatmapLossFn.apply(out, target, channel_weights, L.REG_L2 if kind == "L2" else L.REG_L1)


# ------------------------------------------------------------------------------------------------- optimiser
def adam_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """In-place Adam on flat fp32 buffers (torch.optim.Adam defaults: segmentation.py:119-120)."""
    L.require_gpu(p, "adam_step")
    L.check(L.lib().mednet_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2,
                                     eps, weight_decay, step, grad_scale, L.stream()), "adam_step")
