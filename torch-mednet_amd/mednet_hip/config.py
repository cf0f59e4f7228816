"""Run-time switches of the MI355X path (precision of stored activations, kernel selection)."""
from __future__ import annotations

import contextlib
import os
import threading

import torch

from . import _lib

_state = {
    # "fp32": activations/gradients stored fp32, fp32 FMA everywhere -> matches the reference within 1e-3 (parity mode)
    # "bf16": activations/gradients stored bf16, conv contraction on bf16 MFMA with fp32 accumulation (perf mode)
    # "fp16": the same kernels with fp16 storage and fp16 MFMA operands (BASELINE config 5); gradients need loss scaling
    #         (train.LossScaler: dynamic, device-side), the reference's own fp16 autocast run loses 33 % of the gradient
    #         without it
    # "fp16x2": fp16 storage + SPLIT WEIGHTS (round 6): every 3x3x3 forward / data-gradient convolution and ConvTranspose3d also
    #         multiplies the low image fp16(w - fp16(w)) of its weights (two MFMAs per product).  In fp16 storage the 11-bit rounding
    #         of the matrix cores' WEIGHT operand, not the 16-bit storage of activations and gradients, is what keeps the reference's
    #         gradients outside 1e-3 (tools/attrib16.py, profiles/r06_ab.md section 2): this mode meets 1e-3 on every gradient tensor.
    "precision": os.environ.get("MEDNET_PRECISION", "fp32"),
    "algo": {"auto": _lib.ALGO_AUTO, "direct": _lib.ALGO_DIRECT, "mfma": _lib.ALGO_MFMA}[
        os.environ.get("MEDNET_CONV_ALGO", "auto")],
}


def set_precision(mode: str):
    if mode not in ("fp32", "bf16", "fp16", "fp16x2"):
        raise ValueError("precision must be 'fp32', 'bf16', 'fp16' or 'fp16x2'")
    _state["precision"] = mode


def get_precision() -> str:
    return _state["precision"]


def act_dtype() -> torch.dtype:
    return {"bf16": torch.bfloat16, "fp16": torch.float16, "fp16x2": torch.float16}.get(_state["precision"], torch.float32)


def split_weights() -> bool:
    """fp16x2: the matrix-core convolutions multiply the high AND the low image of their weights."""
    return _state["precision"] == "fp16x2"


def pack_elt() -> int:
    """Element type argument of the pack calls for the current mode (fp32 storage: bf16 high + low images; fp16x2: fp16 + low)."""
    e = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}.get(act_dtype(), _lib.F32)
    return e | _lib.PACK_LOW if split_weights() else e


HALF_TYPES = (torch.bfloat16, torch.float16)


def is_half_mode() -> bool:
    """bf16 or fp16 storage: the 16-bit matrix-core kernels and their fusions apply."""
    return _state["precision"] in ("bf16", "fp16", "fp16x2")


def set_conv_algo(name: str):
    _state["algo"] = {"auto": _lib.ALGO_AUTO, "direct": _lib.ALGO_DIRECT, "mfma": _lib.ALGO_MFMA}[name]


# Scopes (exact_products, algo_scope) are per THREAD: backward runs on autograd's worker thread while another model may
# be in its forward on the main thread, and two threads driving two models must not see each other's request.
_tls = threading.local()


def _scope():
    if not hasattr(_tls, "exact"):
        # algo: the base choice replayed by a backward (None: the process setting); split: likewise (None: the mode's)
        _tls.exact, _tls.algo, _tls.split = False, None, None
    return _tls


def _compose(base: int, exact: bool, split: bool = False) -> int:
    """The C-ABI `algo` argument of a base choice plus the exact-products and split-weights requests (include/mednet_hip.h)."""
    sp = _lib.ALGO_SPLITW_BIT if split else 0
    if not exact or base == _lib.ALGO_DIRECT:  # (the direct kernels' fp32 products are exact anyway)
        return base | sp
    return (_lib.ALGO_EXACT if base == _lib.ALGO_AUTO else (base | _lib.ALGO_EXACT_BIT)) | sp


def _decompose(algo: int):
    """-> (base choice, exact-products request, split-weights request)"""
    split = bool(algo & _lib.ALGO_SPLITW_BIT)
    algo &= ~_lib.ALGO_SPLITW_BIT
    if algo == _lib.ALGO_EXACT:
        return _lib.ALGO_AUTO, True, split
    if algo > _lib.ALGO_EXACT and algo & _lib.ALGO_EXACT_BIT:
        return algo & 3, True, split
    return algo, False, split


def conv_algo() -> int:
    """Algorithm argument of the conv-family C-ABI calls: the base choice (set_conv_algo / MEDNET_CONV_ALGO, or the one a
    backward replays) and -- kept SEPARATE from it -- the exact-products request of the enclosing scopes.  With the request on,
    fp32-storage contractions run on exact fp32 products (v_mfma_f32_32x32x2_f32) instead of the split-bf16 contraction
    (~2^-16 per product), whether the base choice is 'auto' (ALGO_EXACT) or 'mfma' (ALGO_MFMA | ALGO_EXACT_BIT); nothing
    changes for the 16-bit modes."""
    sc = _scope()
    return _compose(_state["algo"] if sc.algo is None else sc.algo, sc.exact, split_weights() if sc.split is None else sc.split)


@contextlib.contextmanager
def exact_products(on: bool = True):
    """Networks with KINKED activations (ReLU 'r', LeakyReLU 'l' in the order string, components.py:36-38) are piecewise
    linear: their gradient is a discontinuous function of every pre-activation, so a 2^-16 perturbation of a convolution
    output flips a few activation masks and moves first-layer gradients by more than the 1e-3 budget (measured: UNet3D 'gcr'
    at 32^3, 1.4e-3).  Such layers therefore ask for exact fp32 products; the smooth default order 'cge' (ELU) does not.
    A scope only turns the request ON (an inner smooth layer inside a kinked network stays exact)."""
    sc = _scope()
    old = sc.exact
    if on:
        sc.exact = True
    try:
        yield
    finally:
        sc.exact = old


@contextlib.contextmanager
def algo_scope(algo):
    """Backward passes run outside the forward's scopes (and on another thread): replay the choice an op captured at
    forward -- its base algorithm and its exact-products request, whatever scope the caller of backward() is in."""
    if algo is None:
        yield
        return
    sc = _scope()
    old = (sc.algo, sc.exact, sc.split)
    sc.algo, sc.exact, sc.split = _decompose(algo)
    try:
        yield
    finally:
        sc.algo, sc.exact, sc.split = old


@contextlib.contextmanager
def precision(mode: str):
    old = _state["precision"]
    set_precision(mode)
    try:
        yield
    finally:
        _state["precision"] = old
