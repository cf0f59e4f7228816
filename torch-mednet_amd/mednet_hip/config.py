"""Run-time switches of the MI355X path (precision of stored activations, kernel selection)."""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib

_state = {
    # "fp32": activations/gradients stored fp32, fp32 FMA everywhere -> matches the reference within 1e-3 (parity mode)
    # "bf16": activations/gradients stored bf16, conv contraction on bf16 MFMA with fp32 accumulation (perf mode)
    "precision": os.environ.get("MEDNET_PRECISION", "fp32"),
    "algo": {"auto": _lib.ALGO_AUTO, "direct": _lib.ALGO_DIRECT, "mfma": _lib.ALGO_MFMA}[
        os.environ.get("MEDNET_CONV_ALGO", "auto")],
}


def set_precision(mode: str):
    if mode not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    _state["precision"] = mode


def get_precision() -> str:
    return _state["precision"]


def act_dtype() -> torch.dtype:
    return torch.bfloat16 if _state["precision"] == "bf16" else torch.float32


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
