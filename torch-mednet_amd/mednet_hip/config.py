"""Run-time switches of the MI355X path (precision of stored activations, kernel selection)."""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib

_state = {
    # "fp32": activations/gradients stored fp32, fp32 FMA everywhere -> matches the reference within 1e-3 (parity mode)
    # "bf16": activations/gradients stored bf16, conv contraction on bf16 MFMA with fp32 accumulation (perf mode)
    # "fp16": the same kernels with fp16 storage and fp16 MFMA operands (BASELINE config 5); gradients need loss scaling
    #         (train.LossScaler: dynamic, device-side), the reference's own fp16 autocast run loses 33 % of the gradient
    #         without it
    "precision": os.environ.get("MEDNET_PRECISION", "fp32"),
    "algo": {"auto": _lib.ALGO_AUTO, "direct": _lib.ALGO_DIRECT, "mfma": _lib.ALGO_MFMA}[
        os.environ.get("MEDNET_CONV_ALGO", "auto")],
}


def set_precision(mode: str):
    if mode not in ("fp32", "bf16", "fp16"):
        raise ValueError("precision must be 'fp32', 'bf16' or 'fp16'")
    _state["precision"] = mode


def get_precision() -> str:
    return _state["precision"]


def act_dtype() -> torch.dtype:
    return {"bf16": torch.bfloat16, "fp16": torch.float16}.get(_state["precision"], torch.float32)


HALF_TYPES = (torch.bfloat16, torch.float16)


def is_half_mode() -> bool:
    """bf16 or fp16 storage: the 16-bit matrix-core kernels and their fusions apply."""
    return _state["precision"] in ("bf16", "fp16")


def set_conv_algo(name: str):
    _state["algo"] = {"auto": _lib.ALGO_AUTO, "direct": _lib.ALGO_DIRECT, "mfma": _lib.ALGO_MFMA}[name]


def conv_algo() -> int:
    return _state["algo"]


@contextlib.contextmanager
def precision(mode: str):
    old = _state["precision"]
    set_precision(mode)
    try:
        yield
    finally:
        _state["precision"] = old
