"""ctypes binding of libmednet_hip.so (the C ABI declared in include/mednet_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises, so a GPU run can never
silently fall back to eager PyTorch.  The reference binds nothing here (its hot path is torch.nn -> ATen); the
functions below replace the ATen ops listed in the header next to each prototype.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEDNET_LIB_PATH") or os.path.join(_HERE, "libmednet_hip.so")  # (override: A/B of two builds)

F32, BF16 = 0, 1
NDHWC, NCDHW = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3
POOL_MAX, POOL_AVG = 0, 1
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_EXACT = 0, 1, 2, 3
ALGO_EXACT_BIT = 4  # OR-ed onto ALGO_MFMA: matrix-core path required AND exact fp32 products
ALGO_SPLITW_BIT = 8  # OR-ed onto any choice: 16-bit modes multiply the LOW weight images too (include/mednet_hip.h, round 6)
PACK_LOW = 0x100     # OR-ed onto the element type of a pack call: an fp16 pack with the low images
PACK_HIGH_ONLY = 0x200  # mednet_conv3d_pack_many: only the images the 16-bit matrix-core kernels read (see mednet_hip.h)
REG_L2, REG_L1 = 0, 1
PAD_CONSTANT, PAD_SYMMETRIC = 0, 1
F16, U8, I64 = 2, 3, 4  # F16: fp16 storage / resident volumes; U8, I64: label types
NO_IGNORE = -(2 ** 31)

_vp, _i, _sz, _f, _i64 = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_int64

# name -> (restype, argtypes); must list EVERY symbol of include/mednet_hip.h (tests/test_abi.py checks that)
SIGNATURES = {
    "mednet_abi_version": (_i, []),
    "mednet_last_error": (C.c_char_p, []),
    "mednet_device_ok": (_i, []),
    "mednet_set_option": (_i, [C.c_char_p, _i]),
    "mednet_get_option": (_i, [C.c_char_p, _i]),
    "mednet_conv3d_pack_bytes": (_sz, [_i, _i, _i]),
    "mednet_conv3d_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mednet_conv3d_pack_elt": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mednet_conv3d_pack_table_bytes": (_sz, [_i]),
    "mednet_conv3d_pack_table": (_i, [_vp, _i, _vp, _vp]),
    "mednet_conv3d_pack_many": (_i, [_vp, _i, C.c_uint, _i, _vp]),
    "mednet_conv3d_stats_plan": (_i, [_i] * 9 + [_vp]),
    "mednet_conv3d_fused_stats_chunks": (_i, [_i] * 10),
    "mednet_conv3d_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp, _vp]),
    "mednet_gn_finalize": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _sz, _i, _i, _f, _vp, _sz, _vp]),
    "mednet_conv3d_act_supported": (_i, [_i] * 7),
    "mednet_conv3d_act_fwd": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp, _i, _vp]),
    "mednet_conv3d_dgrad_add": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "mednet_conv3d_dgrad_gn_rows": (_i, [_i] * 7),
    "mednet_conv3d_dgrad_gn_rows_dt": (_i, [_i] * 8),
    "mednet_conv3d_dgrad_add_supported": (_i, [_i] * 8),
    "mednet_conv3d_dgrad_gn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp] + [_i] * 8 + [_vp]),
    "mednet_gn_act_bwd_fused": (_i, [_vp] * 6 + [_i, _vp, _vp, _vp, _i, _sz, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mednet_gn_bwd_coefficients": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _sz, _i, _i, _vp, _sz, _vp]),
    "mednet_gn_act_bwd_fused_res": (_i, [_vp] * 7 + [_i, _vp, _vp, _vp, _vp, _i, _sz, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mednet_gn_act_bwd_fused_res_pool": (_i, [_vp] * 7 + [_i, _vp, _vp, _vp, _vp] + [_i] * 9 + [_vp, _sz, _vp]),
    "mednet_head_dgrad_gn_rows": (_i, [_i] * 6),
    "mednet_head_dgrad_gn": (_i, [_vp] * 5 + [_i, _vp] + [_i] * 7 + [_vp]),
    "mednet_pool2_bwd_gn_rows": (_i, [_i] * 6),
    "mednet_convt3d_dgrad_gn_rows": (_i, [_i] * 8),
    "mednet_convt3d_dgrad_gn": (_i, [_vp] * 5 + [_i, _vp] + [_i] * 8 + [_vp]),
    "mednet_pool2_bwd_gn": (_i, [_vp] * 5 + [_i, _vp] + [_i] * 7 + [_vp]),
    "mednet_conv3d_wgrad_ws_bytes": (_sz, [_i] * 8),
    "mednet_conv3d_wgrad_coresident": (_i, [_i] * 10),
    "mednet_conv3d_wgrad_plan": (_i, [_i] * 8 + [_vp]),
    "mednet_conv3d_wgrad_c1_gn_supported": (_i, [_i] * 3),
    "mednet_conv3d_wgrad_c1_gn": (_i, [_vp] * 6 + [_i] * 8 + [_vp, _sz, _vp]),
    "mednet_conv3d_wgrad": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp, _sz, _vp]),
    "mednet_convt3d_fwd": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "mednet_convt3d_dgrad": (_i, [_vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "mednet_convt3d_wgrad_ws_bytes": (_sz, [_i] * 7),
    "mednet_convt3d_wgrad": (_i, [_vp, _vp, _vp, _vp] + [_i] * 10 + [_vp, _sz, _vp]),
    "mednet_gn_ws_bytes": (_sz, [_i, _i, _sz]),
    "mednet_gn_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _i, _i, _f, _i, _vp, _sz, _vp]),
    "mednet_gn_act_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _sz, _i, _i, _i, _i, _vp]),
    "mednet_gn_act_bwd": (_i, [_vp] * 11 + [_i, _sz, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mednet_act_fwd": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "mednet_act_bwd": (_i, [_vp, _vp, _vp, _sz, _i, _i, _vp]),
    "mednet_add": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "mednet_pool2_fwd": (_i, [_vp, _vp] + [_i] * 7 + [_vp]),
    "mednet_gn_act_pool_supported": (_i, [_i] * 5),
    "mednet_gn_act_pool_fwd": (_i, [_vp] * 5 + [_i] * 8 + [_vp]),
    "mednet_pool2_bwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp]),
    "mednet_pool2_bwd_act": (_i, [_vp, _vp, _vp, _i, _vp] + [_i] * 8 + [_vp]),
    "mednet_upcat_fwd": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "mednet_upcat_stats_chunks": (_i, [_i] * 7),
    "mednet_upcat_fwd_stats": (_i, [_vp, _vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "mednet_upcat_bwd": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "mednet_loss_ws_bytes": (_sz, [_i, _i, _sz]),
    "mednet_dice_fwd": (_i, [_vp] * 6 + [_i, _i, _sz, _i64, _i64, _f, _i, _i, _vp, _sz, _vp]),
    "mednet_dice_bwd": (_i, [_vp] * 6 + [_i, _i, _sz, _i64, _i64, _f, _i, _i, _vp]),
    "mednet_head_landmark_supported": (_i, [_i, _i, _i, _i, _sz]),
    "mednet_head_landmark_ws_bytes": (_sz, [_i, _sz, _i, _i]),
    "mednet_head_landmark_gn_rows": (_i, [_sz]),
    "mednet_head_landmark_fwd": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i64] + [_vp] * 6 + [_i, _sz, _i, _i, _i, _i, _f, _i, _i, _i,
                                      _vp, _sz, _vp]),
    "mednet_head_landmark_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i64] + [_vp] * 7 + [_i, _vp, _vp, _vp, _i, _sz, _i, _i, _i,
                                      _i, _f, _i, _i, _i, _vp, _sz, _vp]),
    "mednet_head_dice_supported": (_i, [_i] * 4),
    "mednet_head_dice_ws_bytes": (_sz, [_i, _sz, _i, _i]),
    "mednet_head_dice_gn_rows": (_i, [_i, _sz, _i]),
    "mednet_head_dice_fwd": (_i, [_vp] * 4 + [_i, _i64, _vp, _vp, _vp, _vp, _i, _sz, _i, _i, _f, _i, _i, _i, _vp, _sz, _vp]),
    "mednet_head_dice_bwd": (_i, [_vp, _vp, _i, _i64] + [_vp] * 7 + [_i, _vp, _vp, _vp, _i, _sz, _i, _i, _f, _i, _i, _i, _vp, _sz, _vp]),
    "mednet_dice_fwd_lt": (_i, [_vp, _vp, _i, _i64] + [_vp] * 4 + [_i, _i, _sz, _i64, _i64, _f, _i, _i, _vp, _sz, _vp]),
    "mednet_dice_bwd_lt": (_i, [_vp, _vp, _i, _i64] + [_vp] * 4 + [_i, _i, _sz, _i64, _i64, _f, _i, _i, _vp]),
    "mednet_heatmap_loss_fwd_strided": (_i, [_vp, _vp, _i64, _vp, _vp] + [_i, _i, _sz, _i64, _i64, _i, _i, _vp, _sz, _vp]),
    "mednet_heatmap_loss_bwd_strided": (_i, [_vp, _vp, _i64, _vp, _vp, _vp] + [_i, _i, _sz, _i64, _i64, _i, _i, _vp]),
    "mednet_ce_fwd": (_i, [_vp] * 5 + [_i, _i, _sz, _i64, _i64, _i, _vp, _sz, _vp]),
    "mednet_ce_bwd": (_i, [_vp] * 6 + [_i, _i, _sz, _i64, _i64, _i, _vp]),
    "mednet_heatmap_loss_fwd": (_i, [_vp] * 4 + [_i, _i, _sz, _i64, _i64, _i, _i, _vp, _sz, _vp]),
    "mednet_heatmap_loss_bwd": (_i, [_vp] * 5 + [_i, _i, _sz, _i64, _i64, _i, _i, _vp]),
    "mednet_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "mednet_adam_step_scaled": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _vp, _f, _f, _i, _vp]),
    "mednet_grid_gather": (_i, [_vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "mednet_predict_assemble": (_i, [_vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "mednet_crop_patches": (_i, [_vp, _i, _vp, _vp, _i, _vp] + [_i] * 10 + [_vp]),
    "mednet_augment_ws_bytes": (_sz, [_i, _i, _sz]),
    "mednet_augment_patches": (_i, [_vp, _vp, _i, _i, _sz, _vp, _sz, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the shared object is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"mednet_hip: {LIB_PATH} not found -- build it with `python __graft_entry__.py` "
                "(or make -C torch-mednet_amd/csrc). There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        for item in os.environ.get("MEDNET_OPTIONS", "").split(","):  # e.g. "conv_fuse_stats=0,conv_persist=1" (A/B knobs)
            if "=" in item:
                k, v = item.split("=", 1)
                h.mednet_set_option(k.strip().encode(), int(v))
        _lib = h
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().mednet_last_error().decode(errors="replace")
        raise RuntimeError(f"mednet_hip.{what} failed ({rc}): {msg}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError(f"mednet_hip: unsupported dtype {t.dtype} (float32 / bfloat16 / float16 only)")


def dt_of(dtype: torch.dtype) -> int:
    return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[dtype]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"mednet_hip.{who}: expected a HIP (cuda) tensor, got device '{t.device}'. "
                           "The MI355X path has no CPU fallback.")


_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-(device, stream) scratch that only grows; reused by every op on that stream (stream order makes the
    reuse safe)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
