"""ExtResNetBlock as ONE autograd node (midasmednet/unet/components.py:136-180).

    z1 = act(GN1(conv1(x)));  z2 = act(GN2(conv2(z1)));  out = act(GN3(conv3(z2)) + z1)

Hand-written backward over the C ABI instead of 9 chained autograd nodes: the post-activation tensor z1 has two consumers
(conv2 and the residual add), so autograd would materialise both gradients and sum them with an ATen kernel (3 full-size
tensor passes per block).  Here GroupNorm-1's backward reads the two gradient streams directly (`dz2` operand of
mednet_gn_act_bwd), the weight gradients go to the side stream exactly as in the per-op path, and nothing but
libmednet_hip kernels runs between the block's input and output.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

import os

from . import _lib as L
from . import config, debug, ops

ENABLED = os.environ.get("MEDNET_BLOCK_NODE", "1") == "1"
WGRAD_FIRST = os.environ.get("MEDNET_WGRAD_FIRST", "1") == "1"  # A/B knob: launch order of the two gradients


def _conv_fwd(x, packed, cout, want_stats):
    n, cin, d, h, w = x.shape
    lib = L.lib()
    y = ops.empty_cl(n, cout, d, h, w, config.act_dtype(), x.device)
    partial = None
    if want_stats:
        chunks = lib.mednet_conv3d_fused_stats_chunks(n, d, h, w, cin, cout, 3, L.dt(x), L.dt(y), config.conv_algo())
        if chunks > 0:
            partial = torch.empty((n, chunks, cout, 2), dtype=torch.float32, device=x.device)
    with ops.profiled_conv(3, cin, cout, n, d, h, w):
        L.check(lib.mednet_conv3d_fwd(x.data_ptr(), packed.data_ptr(), None, y.data_ptr(), n, d, h, w, cin, cout, 3, L.dt(x),
                                      L.NDHWC, L.dt(y), L.NDHWC, 0, config.conv_algo(), L.ptr(partial), L.stream()), "conv3d_fwd")
    return y, partial


FUSE_POOL = os.environ.get("MEDNET_FUSE_POOL", "1") == "1"  # A/B knob: the next level's pooling inside the block's last apply pass


def _gn_fwd(y, partial, gamma, beta, groups, eps, act, residual, pool_mode=None):
    """-> (z, stats, coef[, pooled]).  `pool_mode`: also write the 2x2x2-pooled z in the same pass (mednet_gn_act_pool_fwd) when the
    shape allows; the fourth result is then the pooled tensor, else None."""
    n, c, d, h, w = y.shape
    spatial = d * h * w
    lib = L.lib()
    stats = torch.empty((n, groups, 2), dtype=torch.float32, device=y.device)
    coef = torch.empty((n, c, 2), dtype=torch.float32, device=y.device)
    ws = L.workspace(lib.mednet_gn_ws_bytes(n, c, spatial), y.device)
    if partial is not None:
        L.check(lib.mednet_gn_finalize(partial.data_ptr(), partial.shape[1], gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                       coef.data_ptr(), n, spatial, c, groups, eps, ws.data_ptr(), ws.numel(), L.stream()),
                "gn_finalize")
    else:
        L.check(lib.mednet_gn_stats(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), coef.data_ptr(), n,
                                    spatial, c, groups, eps, L.dt(y), ws.data_ptr(), ws.numel(), L.stream()), "gn_stats")
    z = torch.empty_like(y, memory_format=ops.CL)
    if pool_mode is not None and FUSE_POOL and lib.mednet_gn_act_pool_supported(d, h, w, c, L.dt(y)):
        pooled = ops.empty_cl(n, c, d // 2, h // 2, w // 2, y.dtype, y.device)
        L.check(lib.mednet_gn_act_pool_fwd(y.data_ptr(), coef.data_ptr(), L.ptr(residual), z.data_ptr(), pooled.data_ptr(), n, d, h, w,
                                           c, act, pool_mode, L.dt(y), L.stream()), "gn_act_pool_fwd")
        return z, stats, coef, pooled
    L.check(lib.mednet_gn_act_fwd(y.data_ptr(), coef.data_ptr(), L.ptr(residual), z.data_ptr(), n, spatial, c, act, L.dt(y),
                                  L.dt(z), L.stream()), "gn_act_fwd")
    return (z, stats, coef, None) if pool_mode is not None else (z, stats, coef)


def _gn_bwd(dz, dz2, y, z, coef, stats, gamma_p, beta_p, groups, act, want_dres, partial=None, lazy=None):
    """Returns (dy, dres, dgamma, dbeta).  `partial`: the first pass ({sum du, sum du*y} per channel) already taken by the
    data-gradient kernel that produced dz (mednet_conv3d_dgrad_gn)."""
    n, c, d, h, w = y.shape
    spatial = d * h * w
    lib = L.lib()
    dy = torch.empty_like(y, memory_format=ops.CL)
    dres = torch.empty_like(y, memory_format=ops.CL) if want_dres else None
    dgamma, dg_direct = ops._grad_target(gamma_p, (c,))
    dbeta, db_direct = ops._grad_target(beta_p, (c,))
    ws = L.workspace(lib.mednet_gn_ws_bytes(n, c, spatial), y.device)
    if partial is not None and z is not None and lazy is not None:
        # residual layer of an encoder block; dz = pooling backward + skip join was NOT written (ops.SkipPool2Fn, lazy): the
        # apply pass rebuilds its rows from the pooled gradient, the arg-max of the block output and the skip gradient
        dy_pool, dskip, pool_mode = lazy
        assert dz2 is None and want_dres
        L.check(lib.mednet_gn_act_bwd_fused_res_pool(dy_pool.data_ptr(), L.ptr(dskip), y.data_ptr(), z.data_ptr(), stats.data_ptr(),
                                                     gamma_p.data_ptr(), partial.data_ptr(), partial.shape[1], dy.data_ptr(),
                                                     dres.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), n, d, h, w, c, groups,
                                                     act, pool_mode, L.dt(y), ws.data_ptr(), ws.numel(), L.stream()),
                "gn_act_bwd_fused_res_pool")
        return dy, dres, (None if dg_direct else dgamma), (None if db_direct else dbeta)
    if partial is not None and z is not None:  # residual layer: sums taken by the producer of dz (ops.GN3Hook)
        assert dz2 is None and want_dres
        L.check(lib.mednet_gn_act_bwd_fused_res(dz.data_ptr(), y.data_ptr(), z.data_ptr(), coef.data_ptr(), stats.data_ptr(),
                                                gamma_p.data_ptr(), partial.data_ptr(), partial.shape[1], dy.data_ptr(),
                                                dres.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), n, spatial, c, groups, act,
                                                L.dt(y), ws.data_ptr(), ws.numel(), L.stream()), "gn_act_bwd_fused_res")
        return dy, dres, (None if dg_direct else dgamma), (None if db_direct else dbeta)
    if partial is not None:
        assert dz2 is None and z is None and not want_dres
        L.check(lib.mednet_gn_act_bwd_fused(dz.data_ptr(), y.data_ptr(), coef.data_ptr(), stats.data_ptr(), gamma_p.data_ptr(),
                                            partial.data_ptr(), partial.shape[1], dy.data_ptr(), dgamma.data_ptr(),
                                            dbeta.data_ptr(), n, spatial, c, groups, act, L.ACT_NONE, L.dt(y), ws.data_ptr(),
                                            ws.numel(), L.stream()), "gn_act_bwd_fused")
        return dy, None, (None if dg_direct else dgamma), (None if db_direct else dbeta)
    L.check(lib.mednet_gn_act_bwd(dz.data_ptr(), L.ptr(dz2), y.data_ptr(), L.ptr(z), coef.data_ptr(), stats.data_ptr(),
                                  gamma_p.data_ptr(), dy.data_ptr(), L.ptr(dres), dgamma.data_ptr(), dbeta.data_ptr(), n,
                                  spatial, c, groups, act, L.ACT_NONE, L.dt(y), ws.data_ptr(), ws.numel(), L.stream()),
            "gn_act_bwd")
    return dy, dres, (None if dg_direct else dgamma), (None if db_direct else dbeta)


FUSE_C1GN = os.environ.get("MEDNET_FUSE_C1GN", "1") == "1"  # A/B knob: first layer's GroupNorm backward inside its weight gradient
C1GN_COUNT = {"fused": 0}  # (tests check that the fused form really ran, like ops.GN3_COUNT)


def _c1_gn_applies(xin, y, partial, need_dx):
    """The network's first SingleConv (Cin = 1, no gradient of the input wanted): its GroupNorm backward feeds nothing but the
    weight gradient, so the apply pass can happen inside that kernel's staging (mednet_conv3d_wgrad_c1_gn)."""
    return (FUSE_C1GN and partial is not None and not need_dx and xin.shape[1] == 1 and y.dtype != torch.float32
            and config.conv_algo() != L.ALGO_DIRECT
            and bool(L.lib().mednet_conv3d_wgrad_c1_gn_supported(y.shape[1], L.dt(xin), L.dt(y))))


def _c1_gn_bwd(xin, dz, y, coef, stats, gamma_p, beta_p, weight_p, groups, act, partial):
    """GroupNorm(+activation) backward and 3x3x3 weight gradient of the first layer without the gradient tensor between them.
    Returns (dw-or-None, dgamma-or-None, dbeta-or-None) like _gn_bwd / _conv_bwd.  On the caller's stream: the side stream is
    busy with the second layer's weight gradient at this point, and this kernel is HBM-bound like the pass it replaces."""
    n, c, d, h, w = y.shape
    lib = L.lib()
    C1GN_COUNT["fused"] += 1
    dgamma, dg_direct = ops._grad_target(gamma_p, (c,))
    dbeta, db_direct = ops._grad_target(beta_p, (c,))
    dw, dw_direct = ops._grad_target(weight_p, (c, 1, 3, 3, 3))
    bcoef = torch.empty((n, c, 3), dtype=torch.float32, device=y.device)
    ws = L.workspace(max(lib.mednet_gn_ws_bytes(n, c, d * h * w), lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, 1, c, 3, 0)), y.device)
    L.check(lib.mednet_gn_bwd_coefficients(stats.data_ptr(), gamma_p.data_ptr(), partial.data_ptr(), partial.shape[1],
                                           bcoef.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), n, d * h * w, c, groups,
                                           ws.data_ptr(), ws.numel(), L.stream()), "gn_bwd_coefficients")
    L.check(lib.mednet_conv3d_wgrad_c1_gn(xin.data_ptr(), dz.data_ptr(), y.data_ptr(), coef.data_ptr(), bcoef.data_ptr(),
                                          dw.data_ptr(), n, d, h, w, c, act, L.dt(xin), L.dt(y), ws.data_ptr(), ws.numel(),
                                          L.stream()), "conv3d_wgrad_c1_gn")
    return (None if dw_direct else dw), (None if dg_direct else dgamma), (None if db_direct else dbeta)


FUSE_DRES = os.environ.get("MEDNET_FUSE_DRES", "1") == "1"  # A/B knob: residual gradient summed in conv2's data gradient


FUSE_GNB = os.environ.get("MEDNET_FUSE_GNB", "1") == "1"  # A/B knob: GroupNorm-backward sums in the data-gradient epilogue


def _conv_bwd(x, dy, packed, weight_p, need_dx, add=None, gnb=None):
    """Weight gradient (side stream in trainer mode) + data gradient (+ `add`, a second gradient of x, summed in the
    data-gradient kernel's epilogue when given).  `gnb` = (y_prev, coef_prev, act): x is act(GroupNorm(y_prev)); when the
    kernel can, it also takes the first pass of that GroupNorm's backward over the dx it stores.
    Returns (dx, dw-or-None, partial-or-None)."""
    n, cin, d, h, w = x.shape
    cout = dy.shape[1]
    lib = L.lib()
    # data gradient first (critical path), THEN the weight gradient on the side stream: queued behind the data gradient it
    # overlaps the next GroupNorm backward (HBM-bound) instead of fighting the data gradient for the matrix cores
    dx = None
    partial = None

    def dgrad():
        nonlocal dx, partial
        if need_dx:
            dx = ops.empty_cl(n, cin, d, h, w, x.dtype if x.dtype in (torch.float32, config.act_dtype()) else config.act_dtype(), dy.device)
            rows = 0
            if gnb is not None and FUSE_GNB and dx.dtype == dy.dtype and (add is None or add.dtype == dy.dtype):
                # (16-bit storage: the bf16 / fp16 matrix-core kernel; fp32 storage: the split-bf16 kernel)
                rows = lib.mednet_conv3d_dgrad_gn_rows_dt(n, d, h, w, cin, cout, config.conv_algo(), L.dt(dy))
            variant = ("add+gn" if add is not None else "gn") if rows > 0 else ("add" if add is not None else "plain")
            with ops.profiled_dgrad(variant, 3, cout, cin, n, d, h, w):
                if rows > 0:
                    y_prev, coef_prev, act_prev = gnb
                    partial = torch.empty((n, rows, cin, 2), dtype=torch.float32, device=dy.device)
                    L.check(lib.mednet_conv3d_dgrad_gn(dy.data_ptr(), packed.data_ptr(), L.ptr(add), dx.data_ptr(), y_prev.data_ptr(),
                                                       coef_prev.data_ptr(), act_prev, partial.data_ptr(), n, d, h, w, cin, cout,
                                                       config.conv_algo(), L.dt(dy), L.stream()), "conv3d_dgrad_gn")
                elif add is not None:
                    L.check(lib.mednet_conv3d_dgrad_add(dy.data_ptr(), packed.data_ptr(), add.data_ptr(), dx.data_ptr(), n, d, h, w,
                                                        cin, cout, config.conv_algo(), L.dt(dy), L.stream()), "conv3d_dgrad_add")
                else:
                    L.check(lib.mednet_conv3d_fwd(dy.data_ptr(), packed.data_ptr(), None, dx.data_ptr(), n, d, h, w, cout, cin, 3,
                                                  L.dt(dy), L.NDHWC, L.dt(dx), L.NDHWC, 1, config.conv_algo(), None, L.stream()),
                            "conv3d_dgrad")

    if not WGRAD_FIRST:
        dgrad()
    dw, direct = ops._grad_target(weight_p, (cout, cin, 3, 3, 3))
    on_side = ops.SIDE["enabled"] and direct
    with ops._OnSide(on_side, dy.device, x, dy, coresident=on_side and ops.wgrad_coresident(n, d, h, w, cin, cout, 3, x, dy)) as side:
        ws = L.workspace(lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, 3, side.workgroups), dy.device)
        with ops.profiled_wgrad(3, cin, cout, n, d, h, w):
            L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, n, d, h, w, cin, cout, 3, L.dt(x),
                                            L.NDHWC, L.dt(dy), L.NDHWC, config.conv_algo(), side.workgroups, ws.data_ptr(),
                                            ws.numel(), L.stream()), "conv3d_wgrad")
    if WGRAD_FIRST:
        dgrad()
    return dx, (None if direct else dw), partial


class ResBlockFn(Function):
    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, pk1, pk2, pk3, groups, eps, act, hook=None, pool=None):
        L.require_gpu(x, "ExtResNetBlock")
        ctx.algo = config.conv_algo()
        x = ops._as_act(x)
        xin = x.contiguous() if x.shape[1] == 1 else ops.to_cl(x)  # Cin == 1: NCDHW and NDHWC coincide
        cout = w1.shape[0]
        fuse = (cout // groups) % 2 == 0  # fused partials are per channel pair (include/mednet_hip.h)
        y1, p1 = _conv_fwd(xin, pk1, cout, fuse)
        z1, s1, c1 = _gn_fwd(y1, p1, g1, b1, groups, eps, act, None)
        y2, p2 = _conv_fwd(z1, pk2, cout, fuse)
        z2, s2, c2 = _gn_fwd(y2, p2, g2, b2, groups, eps, act, None)
        y3, p3 = _conv_fwd(z2, pk3, cout, fuse)
        # `pool`: a holder the caller reads the pooled output from (the next encoder level pools this block's output: one pass
        # writes both; ops.SkipPool2Fn uses the stash instead of launching the pooling kernel)
        if pool is not None:
            out, s3, c3, pooled = _gn_fwd(y3, p3, g3, b3, groups, eps, act, z1, pool_mode=pool.mode)
            pool.pooled = pooled
        else:
            out, s3, c3 = _gn_fwd(y3, p3, g3, b3, groups, eps, act, z1)
        if debug.TRACE is not None:
            debug.trace("resblock.fwd", y1, p1, s1, c1, z1, y2, p2, s2, c2, z2, y3, p3, s3, c3, out)
        ctx.save_for_backward(xin, y1, z1, y2, z2, y3, out, s1, c1, s2, c2, s3, c3, pk1, pk2, pk3)
        ctx.params = (w1, g1, b1, w2, g2, b2, w3, g3, b3)
        ctx.meta = (groups, act)
        ctx.gn3 = hook
        if hook is not None:
            hook.gn_in, hook.act = y3, act
        return out

    @staticmethod
    @ops._with_algo
    def backward(ctx, dout):
        xin, y1, z1, y2, z2, y3, out, s1, c1, s2, c2, s3, c3, pk1, pk2, pk3 = ctx.saved_tensors
        w1, g1, b1, w2, g2, b2, w3, g3, b3 = ctx.params
        groups, act = ctx.meta
        part3 = ctx.gn3.take(dout) if ctx.gn3 is not None else None  # (before any conversion: identity matters)
        lazy = None
        if ctx.gn3 is not None and part3 is not None:
            lazy, ctx.gn3.lazy = ctx.gn3.lazy, None  # dout is an unwritten placeholder: see ops.SkipPool2Fn / GN3Hook.take
        if lazy is None:
            dout = ops.to_cl(dout.to(out.dtype))
        # GN3 + residual + activation: act' from the block output; dres = gradient of the residual branch (into z1)
        dy3, dres, dg3, db3 = _gn_bwd(dout, None, y3, out, c3, s3, g3, b3, groups, act, True, partial=part3, lazy=lazy)
        dz2, dw3, part2 = _conv_bwd(z2, dy3, pk3, w3, True, gnb=(y2, c2, act))
        dy2, _, dg2, db2 = _gn_bwd(dz2, None, y2, None, c2, s2, g2, b2, groups, act, False, partial=part2)
        # z1 feeds conv2 AND the residual add: the two gradients are summed in the epilogue of conv2's data gradient (bf16
        # matrix-core path), otherwise inside GroupNorm-1's backward (two more tensor reads)
        n_, c_, d_, h_, w_ = z1.shape
        fuse = FUSE_DRES and dres.dtype == dy2.dtype and bool(
            L.lib().mednet_conv3d_dgrad_add_supported(n_, d_, h_, w_, c_, c_, config.conv_algo(), L.dt(dy2)))
        dz1, dw2, part1 = _conv_bwd(z1, dy2, pk2, w2, True, add=dres if fuse else None, gnb=(y1, c1, act) if fuse else None)
        if fuse and _c1_gn_applies(xin, y1, part1, ctx.needs_input_grad[0]):
            dy1 = dx = None  # (never materialised)
            dw1, dg1, db1 = _c1_gn_bwd(xin, dz1, y1, c1, s1, g1, b1, w1, groups, act, part1)
        else:
            dy1, _, dg1, db1 = _gn_bwd(dz1, None if fuse else dres, y1, None, c1, s1, g1, b1, groups, act, False, partial=part1)
            dx, dw1, _ = _conv_bwd(xin, dy1, pk1, w1, ctx.needs_input_grad[0])
        if debug.TRACE is not None:
            debug.trace("resblock.bwd", None if lazy is not None else dout, part3, dy3, dres, dz2, part2, dy2, dz1, part1, dy1, dx)
        return (dx, dw1, dg1, db1, dw2, dg2, db2, dw3, dg3, db3) + (None,) * 8


class PoolStash:
    """Carried by a block output whose consumer is a 2x2x2 pooling of `mode`: the pooled tensor the block's last apply pass wrote
    beside the output (or None when the shape did not allow it).  ops.SkipPool2Fn / Pool2Fn take it instead of launching."""
    __slots__ = ("mode", "pooled", "version")

    def __init__(self, mode):
        self.mode, self.pooled, self.version = mode, None, -1  # version: the output's `_version` when the stash was attached


def res_block(x, convs, norms, groups, eps, act, pool_mode=None):
    """convs / norms: the three mednet_hip.nn.Conv3d / GroupNorm modules of the block.  `pool_mode`: the block's output goes into a
    2x2x2 pooling of that mode next (the next encoder level): the pooled tensor is produced in the same pass and stashed on the
    output (`_mednet_pooled`)."""
    (k1, k2, k3), (n1, n2, n3) = convs, norms
    hook = ops.GN3Hook() if (ops.FUSE_GN3 and torch.is_grad_enabled()) else None  # (training only)
    stash = PoolStash(pool_mode) if (pool_mode is not None and FUSE_POOL) else None
    out = ResBlockFn.apply(x, k1.weight, n1.weight, n1.bias, k2.weight, n2.weight, n2.bias, k3.weight, n3.weight, n3.bias,
                           k1._packed(x, converts=True), k2._packed(x, converts=True), k3._packed(x, converts=True), groups, eps, act,
                           hook, stash)  # (the node casts x to the storage type; its inner tensors have x's extent)
    if hook is not None:
        out._mednet_gn3 = hook  # see ops.GN3Hook: the consumer of `out` may take GroupNorm-3's first backward pass
    if stash is not None and stash.pooled is not None:
        stash.version = out._version
        out._mednet_pooled = stash
    return out
