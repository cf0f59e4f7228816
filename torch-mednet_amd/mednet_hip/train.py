"""Training-step harness for the hot path: the callers' step contract (segmentation.py:58-65, landmarks.py:66-83,
configure_optimizers :119-120) plus what pytorch_lightning's Trainer(gpus=N) adds around it (train_seg.py:126): one
gradient all-reduce per step over RCCL and the Adam update.

MI355X-first layout: all parameters live in ONE flat fp32 buffer and all gradients in another.  Backward kernels
write each parameter gradient straight into its slice (no autograd accumulate kernels), the data-parallel exchange
is a single RCCL all-reduce of the flat gradient buffer (35 MB for config 3), its 1/world averaging is folded into the
fused Adam kernel, and Adam is one launch over the flat buffers.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import ops
from .unet import loss as HL


class FlatParams:
    """Re-homes a module's parameters into one flat fp32 buffer (+ a flat gradient buffer the kernels write into)."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.offsets = []
        total = 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 63) // 64 * 64  # 256-byte aligned slices
        self.total = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            view = self.flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p._mednet_grad = self.grad[off:off + p.numel()].view_as(p)

    def grads_as_attr(self):
        """Expose the flat slices as .grad (for inspection / torch optimizers)."""
        for p in self.params:
            p.grad = p._mednet_grad

    def release(self):
        for p in self.params:
            if hasattr(p, "_mednet_grad"):
                del p._mednet_grad


class FlatAdam:
    """torch.optim.Adam(lr) semantics (betas .9/.999, eps 1e-8, weight_decay 0) as one kernel over FlatParams."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.t = 0  # calls so far.  Plain mode: = optimizer steps taken (bias correction uses it).  Behind a LossScaler the
        #             bias correction uses the DEVICE count scaler.state[2], which a skipped step does not advance: there
        #             `t` only versions the parameters (_bump) and steps_taken() reads the truth.
        self._scaled = None  # which of the two update forms this optimizer runs (they keep different step counts)

    def steps_taken(self, scaler: "LossScaler | None" = None) -> int:
        """Optimizer steps actually applied (synchronises when a scaler holds the count)."""
        return self.t if scaler is None else int(scaler.state[2].item())

    def state_dict(self, scaler: "LossScaler | None" = None):
        sd = {"m": self.m.detach().cpu().clone(), "v": self.v.detach().cpu().clone(), "t": self.steps_taken(scaler),
              "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd}
        if scaler is not None:
            sd["scaler"] = scaler.state_dict()
        return sd

    def load_state_dict(self, sd, scaler: "LossScaler | None" = None):
        """Restores the moments, the step count and the hyper-parameters the state was saved with (torch.optim.Adam's
        load_state_dict restores its param_groups too).  A run that trains behind a loss scaler must find the scaler's state
        in `sd`: resuming fp16 training with a fresh scale of 65536 and a device step count of 0 would silently restart the
        bias correction."""
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        self.t = int(sd["t"])
        self.lr = sd.get("lr", self.lr)
        self.betas = tuple(sd.get("betas", self.betas))
        self.eps = sd.get("eps", self.eps)
        self.wd = sd.get("weight_decay", self.wd)
        if scaler is not None:
            if "scaler" not in sd:
                raise KeyError("FlatAdam.load_state_dict: a LossScaler was passed but the saved state has no 'scaler' entry "
                               "(it was saved by a run without loss scaling)")
            scaler.load_state_dict(sd["scaler"])

    def _form(self, scaled: bool):
        if self._scaled is None:
            self._scaled = scaled
        elif self._scaled != scaled:
            raise RuntimeError("FlatAdam: step() and step_scaled() keep different step counts (host / device) and cannot "
                               "be mixed on one optimizer")

    def step(self, grad_scale=1.0):
        self._form(False)
        self.t += 1
        ops.adam_step_(self.flat.flat, self.flat.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.wd, self.t, grad_scale)
        self._bump()

    def step_scaled(self, scaler: "LossScaler", inv_world=1.0):
        """The same update behind the loss scaler: unscale, skip on overflow, adjust the scale -- all on the device."""
        from . import _lib as L
        self._form(True)
        self.t += 1
        f = self.flat
        L.check(L.lib().mednet_adam_step_scaled(f.flat.data_ptr(), f.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                                f.flat.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                                inv_world, scaler.state.data_ptr(), scaler.growth_factor,
                                                scaler.backoff_factor, scaler.growth_interval, L.stream()), "adam_step_scaled")
        self._bump()

    def _bump(self):
        # torch's version counter only moves on torch in-place ops; a no-op in-place op on the flat buffer does not
        # reach the views' counters, so packed-weight caches are keyed on this step counter as well.
        for p in self.flat.params:
            p._mednet_step = self.t


class BatchedRepack:
    """After an optimizer step every conv layer's packed weight images are stale; instead of one small launch per layer at
    its next forward (23 per step for cfg2) ONE launch rebuilds the matrix-core layers' images (mednet_conv3d_pack_many).
    The device table of (parameter, pack buffer, shape) is built once: parameters live in the flat buffer and the pack
    buffers are rewritten in place.  Layers without matrix-core images (first conv, 1x1x1 head) keep the per-layer path."""

    ENABLED = os.environ.get("MEDNET_BATCHED_PACK", "1") == "1"  # A/B knob
    # 16-bit storage modes: rewrite only the images the step's matrix-core kernels read (MEDNET_PACK_HIGH_ONLY) -- not the fp32
    # images of the direct / fp32-matrix kernels nor unrequested low images: 1.1 GB instead of 2.8 GB per step for config 5's 141 M
    # parameters.  The buffers are marked (`_mednet_lean`) and nn._PackedWeightMixin._packed(x) re-packs a layer in full before a
    # call that would read anything else.  Only when EVERY layer of the table always takes the matrix-core path on 16-bit tensors.
    LEAN = os.environ.get("MEDNET_LEAN_PACK", "1") == "1"

    def __init__(self, model):
        from . import nn as hnn
        self.mods = [m for m in model.modules() if isinstance(m, hnn._PackedWeightMixin) and m.kernel_size[0] == 3
                     and m.weight.shape[0] % 16 == 0 and m.weight.shape[1] % 16 == 0]
        self.sig = None
        self.table = None
        self.max_blocks = 0

    def _signature(self):
        from . import config
        return tuple((m.weight.data_ptr(), m._pack_buf.data_ptr()) for m in self.mods) + (config.act_dtype(), config.split_weights())

    def run(self):
        """Call right after the parameters changed (same stream).  No-op until every layer has been packed once."""
        import ctypes as C
        from . import _lib as L, config
        if not self.ENABLED or not self.mods or any(getattr(m, "_pack_buf", None) is None for m in self.mods):
            return
        lib = L.lib()
        sig = self._signature()
        if sig != self.sig:
            class Job(C.Structure):
                _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("cin", C.c_int), ("cout", C.c_int), ("ksize", C.c_int),
                            ("transposed_src", C.c_int)]
            jobs = (Job * len(self.mods))()
            for j, m in zip(jobs, self.mods):
                w = m.weight
                cin, cout = (w.shape[0], w.shape[1]) if m._transposed else (w.shape[1], w.shape[0])
                j.w, j.packed, j.cin, j.cout, j.ksize, j.transposed_src = w.data_ptr(), m._pack_buf.data_ptr(), cin, cout, 3, int(m._transposed)
            host = torch.empty(lib.mednet_conv3d_pack_table_bytes(len(self.mods)), dtype=torch.uint8)
            mb = C.c_uint(0)
            L.check(lib.mednet_conv3d_pack_table(C.addressof(jobs), len(self.mods), host.data_ptr(), C.addressof(mb)), "pack_table")
            self.table = host.to(self.mods[0].weight.device)
            self.max_blocks, self.sig = mb.value, sig
        elt = config.pack_elt()  # (fp32 storage: bf16 + the low images; fp16x2: fp16 + the low images)
        lean = self.LEAN and config.is_half_mode() and all(m._lean_layer_ok() for m in self.mods)
        L.check(lib.mednet_conv3d_pack_many(self.table.data_ptr(), len(self.mods), self.max_blocks, elt | (L.PACK_HIGH_ONLY if lean else 0),
                                            L.stream()), "pack_many")
        for m in self.mods:  # what _PackedWeightMixin._packed() will compute at the next forward
            w = m.weight
            m._pack_key = (w.data_ptr(), w._version, getattr(w, "_mednet_step", 0), str(w.device), config.act_dtype(), config.pack_elt())
            m._pack_buf._mednet_lean = lean


class LossScaler:
    """Dynamic loss scaling for fp16 storage (BASELINE config 5; the reference's hint is the commented-out `precision=16` of
    examples/train_seg.py:127).  torch.cuda.amp.GradScaler's rule -- scale 2^16, halve on overflow, double after 2000 clean
    steps -- with the whole state on the device: the loss is multiplied by a device scalar, the overflow test, the decision
    to skip the update and the scale update are kernels (mednet_adam_step_scaled), so a step never waits for the host."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    def state_dict(self):
        """Everything a resumed run needs (synchronises): the device state {scale, good steps, optimizer steps taken,
        found_inf, skipped steps} and the rule's constants."""
        return {"state": self.state.detach().cpu().clone(), "growth_factor": self.growth_factor,
                "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval}

    def load_state_dict(self, sd):
        st = sd["state"].to(self.state.device).reshape(-1)
        if st.numel() > self.state.numel():
            raise ValueError(f"LossScaler.load_state_dict: saved state has {st.numel()} entries, this build keeps {self.state.numel()}")
        # (earlier builds kept 4 entries {scale, good steps, steps taken, found_inf}; the counters added since start at 0)
        self.state.zero_()
        self.state[:st.numel()].copy_(st)
        self.growth_factor, self.backoff_factor = sd["growth_factor"], sd["backoff_factor"]
        self.growth_interval = sd["growth_interval"]

    def skipped_steps(self) -> int:
        """Steps whose update was skipped for non-finite gradients (synchronises)."""
        return int(self.state[4].item())

    def scale_loss(self, loss):
        return loss * self.state[0]

    def snapshot(self):
        """(scale, good steps, optimizer steps taken, found_inf) -- synchronises; for tests and logging only."""
        return tuple(float(v) for v in self.state[:4].tolist())


def make_scaler(device):
    from . import config
    return LossScaler(device) if config.act_dtype() == torch.float16 else None


def finish_backward():
    """Join the weight-gradient side stream (ops.SIDE) before anything reads the flat gradient buffer."""
    ops.join_side_stream()


def allreduce_gradients(flat_grad: torch.Tensor, world_size: int, force: bool = False):
    """The data-parallel exchange: ONE all-reduce(sum) of the flat gradient buffer (RCCL over xGMI on GPUs; any
    torch.distributed backend in tests).  Returns the scale the optimizer must apply (1/world) so that the update uses
    the mean of the per-rank-batch gradients -- what PL's DDP does with the reference (Dice is reduced
    over the LOCAL batch, then gradients are averaged)."""
    if world_size > 1 or force:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


def late_bucket_split(model, flat: FlatParams, levels: int = 2):
    """Element offset in the flat buffers that separates the parameters whose gradients arrive LAST in backward -- the
    first `levels` encoders, which are the first parameters in module order -- from all the others.  For the U-Nets of
    the path the late bucket is tiny (cfg2: 0.34 M of 8.77 M parameters) while its backward is the longest stretch (the
    two full-resolution levels), so everything else can be exchanged underneath it.  None when the model has no such
    structure."""
    encs = getattr(model, "encoders", None)
    if encs is None or len(encs) <= levels:
        return None
    late = {id(p) for enc in list(encs)[:levels] for p in enc.parameters() if p.requires_grad}
    k = 0
    while k < len(flat.params) and id(flat.params[k]) in late:
        k += 1
    if k != len(late) or k == 0 or k == len(flat.params):
        return None  # not a prefix of the flat order: keep the single exchange
    return flat.offsets[k]


class BucketedExchange:
    """Two-bucket gradient exchange overlapped with backward.

    When autograd delivers the gradient of encoder[levels-1]'s output, every parameter outside the first `levels` encoders
    has its gradient launched: that slice of the flat buffer (cfg2: 96 % of the bytes) is all-reduced asynchronously --
    issued behind the weight-gradient stream, so RCCL waits for exactly the kernels that produce it -- while the backward
    of the full-resolution encoders runs.  The small remainder is exchanged after backward; `finish()` makes the
    compute stream wait for both.  Sums are identical to the single exchange (disjoint slices)."""

    def __init__(self, model, flat: FlatParams, world_size: int, force: bool = False, levels: int = 2):
        self.flat, self.world, self.work, self.force = flat, world_size, None, bool(force)
        self.split = late_bucket_split(model, flat, levels) if (world_size > 1 or force) else None
        # The overlapped form is OPT-IN (MEDNET_BUCKETS=1) until a multi-GPU measurement exists.  One-rank RCCL rehearsal on an
        # MI355X, round 4 (profiles/r04_ab.md section 15, after the side stream got a hardware queue of its own under a
        # process group): one all-reduce after backward 20.55 ms per step, two buckets 20.69 ms, no process group 20.58 -- the
        # collective inside backward costs the compute stream ~0.14 ms there, against a ring time over xGMI of ~0.4-1.2 ms
        # for cfg3's 35 MB that it would hide (the round-3 figure of 0.6 ms was taken with both streams on one queue).  For
        # cfg5 (565 MB, ~6.5 ms single-ring against a 47 ms step) hiding the exchange under the backward
        # of the full-resolution encoders is the obvious candidate -- to be switched on by a scaling run, not by a guess
        # (ADVICE r3).
        self.enabled = self.split is not None and self.overlap_selected(flat.total * 4)
        if self.enabled:
            list(model.encoders)[levels - 1].register_forward_hook(self._on_forward)

    @classmethod
    def overlap_selected(cls, grad_bytes: int) -> bool:
        """MEDNET_BUCKETS=1 selects the two-bucket overlapped exchange (any size); default: one all-reduce after backward."""
        return os.environ.get("MEDNET_BUCKETS") == "1"

    def describe(self) -> str:
        """For bench records: which exchange this step runs."""
        if not (self.world > 1 or self.force):
            return "none (1 rank)"
        mb = self.flat.total * 4 / 1e6
        if self.enabled:
            early = (self.flat.total - self.split) * 4 / 1e6
            return (f"2 buckets: {early:.1f} MB all-reduced under the backward of the full-resolution encoders, "
                    f"{mb - early:.1f} MB after backward")
        return f"one all-reduce of the flat fp32 gradient buffer ({mb:.1f} MB) after backward"

    def _on_forward(self, module, inputs, output):
        if isinstance(output, tuple):  # Encoder(x, with_skip=True) -> (skip, out): the hook belongs on the level's output
            output = output[-1]
        if self.enabled and torch.is_tensor(output) and output.requires_grad:
            output.register_hook(self._on_grad)

    def _on_grad(self, grad):
        early = self.flat.grad[self.split:]
        if self.work is None and not (early.is_cuda and torch.cuda.is_current_stream_capturing()):
            if early.is_cuda:
                main = torch.cuda.current_stream(early.device)
                side = ops.side_stream(early.device)
                side.wait_stream(main)  # main-stream gradients of the early bucket (GroupNorm affines, biases)
                with torch.cuda.stream(side):
                    self.work = dist.all_reduce(early, op=dist.ReduceOp.SUM, async_op=True)
            else:
                self.work = dist.all_reduce(early, op=dist.ReduceOp.SUM, async_op=True)
        return None

    # bench.py's `rccl` record: a list to which finish() appends (start, end) HIP events around the exchange on the compute
    # stream (the early bucket of the two-bucket form runs inside backward and is not in them: the record says which form ran)
    TIMING = None

    def finish(self):
        """After backward (and the side-stream join): exchange what is left, wait for the early bucket; -> 1/world."""
        timed = (BucketedExchange.TIMING is not None and (self.world > 1 or self.force) and self.flat.grad.is_cuda
                 and not torch.cuda.is_current_stream_capturing())
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scale = self._finish()
        if timed:
            e1.record()
            BucketedExchange.TIMING.append((e0, e1))
        return scale

    def _finish(self):
        if self.work is not None:
            dist.all_reduce(self.flat.grad[:self.split], op=dist.ReduceOp.SUM)
            self.work.wait()
            self.work = None
        elif self.world > 1 or self.force:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM)
        return 1.0 / self.world


class _GraphedStep:
    """hipGraph capture of forward + loss + backward (everything up to the gradient exchange).

    A step is ~330 kernel launches from Python; the GPU stays ahead of the host only while the host is not disturbed.
    After `warm` eager calls the launch sequence of `_fwd_bwd` (both streams: the weight-gradient side stream forks from
    and re-joins the capture stream) is captured once and replayed with one call per step.  The all-reduce and the
    Adam launch stay outside the graph (Adam's bias correction is computed on the host from the step count).
    Shapes and dtypes of the batch must not change after capture; a new batch tensor is copied into the static one."""

    GRAPH_DEFAULT = os.environ.get("MEDNET_GRAPH", "0") == "1"

    def _init_graph(self, graph, warm=2):
        self.use_graph = self.GRAPH_DEFAULT if graph is None else bool(graph)
        self._calls, self._warm, self._graph, self._static = 0, warm, None, None

    def _run(self, batch):
        """-> tuple of detached loss tensors of this step's forward/backward (gradients are in self.flat.grad)."""
        if not self.use_graph:
            return self._fwd_bwd(batch)
        self._calls += 1
        if self._graph is None:
            if self._calls <= self._warm:
                return self._fwd_bwd(batch)
            ops.PROFILE["enabled"] = False  # timing events cannot be recorded into a graph
            self._static = {k: v for k, v in batch.items()}
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._out = self._fwd_bwd(self._static)
        for k, v in batch.items():
            if v is not self._static[k]:
                self._static[k].copy_(v)
        self._graph.replay()
        return self._out


def use_side_stream():
    """Weight gradients on their own stream (ops.SIDE) for every step class: MEDNET_SIDE_STREAM=0 turns it off."""
    ops.SIDE["enabled"] = os.environ.get("MEDNET_SIDE_STREAM", "1") == "1"


def _call_is_hooked(model, final_conv) -> bool:
    """The fused head + loss nodes call model.forward_features() and never final_conv: they bypass `model.__call__` and
    `final_conv.__call__`, and in the landmark form no `outputs` tensor exists at all.  Any forward hook / pre-hook registered on
    those two modules, or a global module hook, would silently not fire -- the steps then take the stock `model(inputs)` path.
    (Hooks on encoder / decoder children still fire in the fused form: forward_features calls them through __call__.)"""
    import torch.nn.modules.module as M
    for name in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_forward_hooks_always_called"):
        if getattr(M, name, None):
            return True
    return any(bool(getattr(mod, attr, None)) for mod in (model, final_conv) for attr in ("_forward_hooks", "_forward_pre_hooks"))


class SegmentationStep(_GraphedStep):
    """One data-parallel training step of SegmentationNet (segmentation.py:58-65) on the MI355X path."""

    def __init__(self, model, loss_weight=None, lr=1e-3, loss="DICE", world_size=1, graph=None):
        self.model = model
        dev = next(model.parameters()).device
        w = None if loss_weight is None else torch.tensor(loss_weight, dtype=torch.float32, device=dev)
        self.loss = (HL.DiceLoss(weight=w) if loss == "DICE" else HL.CrossEntropyLoss(weight=w)).to(dev)
        self.flat = FlatParams(model)
        self.opt = FlatAdam(self.flat, lr=lr)
        self.repack = BatchedRepack(model)
        self.world = world_size
        use_side_stream()
        self.scaler = make_scaler(dev)  # fp16 storage only
        self._init_graph(graph)
        self._exchange = None

    def _head_loss(self, inputs, label_u8):
        """`outputs = self(inputs); loss = self.loss(outputs, labels)` (segmentation.py:61-62).  When the model is this package's
        U-Net with a 1x1x1 head of at most 4 classes and the loss is DiceLoss, the head and the loss run as one fused node
        (ops.head_dice: logits written once, no logit-gradient tensor, the label volume's uint8 channel consumed where it
        lies); anything else takes the two calls as they stand."""
        from .unet.model import _UNetCore
        m, fc = self.model, getattr(self.model, "final_conv", None)
        if (isinstance(m, _UNetCore) and not m.testing and isinstance(self.loss, HL.DiceLoss) and not self.loss.skip_last_target
                and fc is not None and getattr(fc, "planar_output", False) and fc.kernel_size[0] == 1 and inputs.is_cuda
                and not _call_is_hooked(m, fc)):
            feats = m.forward_features(inputs)
            if ops.head_dice_supported(feats, fc.in_channels, fc.out_channels, label_u8):
                return ops.head_dice(feats, fc.weight, fc.bias, fc._packed(), label_u8, self.loss.weight, self.loss.epsilon,
                                     self.loss.sigmoid_normalization, self.loss.ignore_index)
            outputs = fc(feats)
        else:
            outputs = m(inputs)
        return outputs, self.loss(outputs, label_u8.long())

    def _fwd_bwd(self, batch):
        inputs = batch["data"].float()
        _, loss = self._head_loss(inputs, batch["label"][:, -1, ...])
        (loss if self.scaler is None else self.scaler.scale_loss(loss)).backward()
        finish_backward()
        return (loss.detach(),)

    def __call__(self, batch):
        force = getattr(self, "force_allreduce", False)
        if self._exchange is None:  # (built lazily: bench.py sets force_allreduce after construction)
            self._exchange = BucketedExchange(self.model, self.flat, self.world, force)
        (loss,) = self._run(batch)
        scale = self._exchange.finish()  # 1/world is folded into Adam
        if self.scaler is None:
            self.opt.step(grad_scale=scale)
        else:
            self.opt.step_scaled(self.scaler, inv_world=scale)
        self.repack.run()
        return loss


class SegmentationValidation:
    """SegmentationNet.validation_step / validation_epoch_end (segmentation.py:94-118) on the MI355X
    path: forward kernels only, the configured loss and `dice_metric` each as ONE fused pass over the logits, results as
    device scalars (no host synchronisation per batch; the reference's sample plotting is not part of it)."""

    def __init__(self, model, loss_weight=None, loss="DICE"):
        self.model = model
        dev = next(model.parameters()).device
        w = None if loss_weight is None else torch.tensor(loss_weight, dtype=torch.float32, device=dev)
        self.loss = (HL.DiceLoss(weight=w) if loss == "DICE" else HL.CrossEntropyLoss(weight=w)).to(dev)

    @torch.no_grad()
    def validation_step(self, batch, batch_nb=0):
        inputs = batch["data"].float()
        labels = batch["label"][:, -1, ...].long()
        was_training = self.model.training
        self.model.eval()
        try:
            outputs = self.model(inputs)
        finally:
            self.model.train(was_training)
        results = {"val_loss": self.loss(outputs, labels)}
        per_channel_dice = HL.dice_metric(outputs, labels)
        for c in range(outputs.shape[1]):
            results[f"val_dice{c}"] = per_channel_dice[c]
        return results

    @staticmethod
    def validation_epoch_end(outputs):
        logs = {k: torch.stack([o[k] for o in outputs]).mean() for k in outputs[0]}
        return {"val_loss": logs["val_loss"], "log": logs, "progress_bar": logs}


class LandmarkStep(_GraphedStep):
    """LandmarkNet.training_step (landmarks.py:66-83, loss :125-134) with the per-channel regression loop fused."""

    def __init__(self, model, class_weight, regression_weight, regression="L2", lr=1e-3, world_size=1, graph=None):
        self.model = model
        dev = next(model.parameters()).device
        self.loss_class = HL.DiceLoss(weight=torch.tensor(class_weight, dtype=torch.float32, device=dev)).to(dev)
        self.loss_reg = HL.HeatmapRegressionLoss(regression_weight, regression).to(dev)
        self.flat = FlatParams(model)
        self.opt = FlatAdam(self.flat, lr=lr)
        self.repack = BatchedRepack(model)
        self.world = world_size
        use_side_stream()
        self.scaler = make_scaler(dev)
        self._init_graph(graph)
        self._exchange = BucketedExchange(model, self.flat, world_size)

    def _head_losses(self, inputs, heatmaps, labels, nh):
        """`outputs = self(inputs)` and the two terms of LandmarkNet.loss (landmarks.py:71-75, 125-134).  In the 16-bit storage
        modes, with this package's U-Net, a 32-feature 1x1x1 head and uint8 targets, head and losses run as one fused node on the
        matrix cores (ops.head_landmark: no logit tensor); anything else takes the calls as they stand."""
        from .unet.model import _UNetCore
        m, fc = self.model, getattr(self.model, "final_conv", None)
        # (the fused branch bypasses model.__call__ and final_conv.__call__ and never forms `outputs`: with a forward (pre-)hook on
        #  the model, on one of its children or a global module hook, the stock m(inputs) path runs so that the hook sees its tensor)
        if (isinstance(m, _UNetCore) and not m.testing and fc is not None and getattr(fc, "planar_output", False)
                and fc.kernel_size[0] == 1 and inputs.is_cuda and not self.loss_class.skip_last_target and self.loss_reg.kind in ("L1", "L2")
                and not _call_is_hooked(m, fc)):
            feats = m.forward_features(inputs)
            if ops.head_landmark_supported(feats, fc.in_channels, nh, fc.out_channels - nh, heatmaps, labels):
                return ops.head_landmark(feats, fc.weight, fc.bias, fc._packed(), heatmaps, labels, self.loss_class.weight,
                                         self.loss_reg.channel_weights, self.loss_reg.kind, self.loss_class.epsilon,
                                         self.loss_class.sigmoid_normalization, self.loss_class.ignore_index)
            outputs = fc(feats)
        else:
            outputs = m(inputs)
        out_hm, out_cls = ops.split_channels(outputs, nh)  # (= outputs[:, :nh], outputs[:, nh:]; one gradient join)
        return self.loss_class(out_cls, labels), self.loss_reg(out_hm, heatmaps)

    def _fwd_bwd(self, batch):
        inputs = batch["data"].float()
        heatmaps = batch["label"][:, :-1, ...]  # uint8 is consumed directly by the fused regression kernel
        nh = heatmaps.shape[1]
        labels = batch["label"][:, -1, ...]  # uint8, consumed where it lies (ops.dice_loss takes uint8 or int64 labels)
        if not labels.is_cuda:
            labels = labels.long()
        class_loss, regression_loss = self._head_losses(inputs, heatmaps, labels, nh)
        loss = regression_loss + class_loss
        (loss if self.scaler is None else self.scaler.scale_loss(loss)).backward()
        finish_backward()
        return loss.detach(), class_loss.detach(), regression_loss.detach()

    def __call__(self, batch):
        out = self._run(batch)
        scale = self._exchange.finish()
        if self.scaler is None:
            self.opt.step(grad_scale=scale)
        else:
            self.opt.step_scaled(self.scaler, inv_world=scale)
        self.repack.run()
        return out
