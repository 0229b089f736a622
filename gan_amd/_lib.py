"""ctypes binding of libgan_amd.so (C ABI declared in include/gan_amd.h).

The library is the product: there is no fallback.  If it is missing, or a tensor is not on a
ROCm device, the calls here raise -- nothing in this package computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p
from typing import Optional

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAN_AMD_LIB: development aid (A/B builds of the same ABI); the product loads the in-tree library
LIB_PATH = os.environ.get("GAN_AMD_LIB") or os.path.join(_HERE, "libgan_amd.so")

GAN_F32, GAN_BF16 = 0, 1
ABI_VERSION = 25

_SIGNATURES = {
    "gan_amd_version": (c_int, []),
    "gan_amd_arch": (c_char_p, []),
    "gan_amd_last_error": (c_char_p, []),
    "gan_upfirdn2d": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 13 + [c_void_p]),
    "gan_affine_act_fwd": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_int, c_void_p],
    ),
    "gan_mask_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_float, c_int, c_void_p]),
    "gan_chan_dot_workspace": (c_int64, [c_int, c_int64, c_int, c_int]),
    "gan_chan_dot": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p],
    ),
    "gan_conv2d_fwd": (
        c_int,
        [c_void_p] * 8 + [c_float] + [c_int] * 13 + [c_float, c_int, c_void_p, c_float, c_float, c_void_p] + [c_int, c_void_p],
    ),
    "gan_conv2d_fwd_workspace": (c_int64, [c_int] * 10),
    "gan_conv2d_fwd_ws": (
        c_int,
        [c_void_p] * 8 + [c_float] + [c_int] * 13 + [c_float, c_int, c_void_p, c_float, c_float, c_void_p] + [c_void_p, c_int64, c_int, c_void_p],
    ),
    "gan_conv2d_wgrad_workspace": (c_int64, [c_int] * 13),
    "gan_conv2d_wgrad": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p] + [c_int] * 12 + [c_void_p, c_int64, c_float, c_int, c_float, c_int, c_int, c_void_p]),
    "gan_conv2d_wgrad_bias_workspace": (c_int64, [c_int] * 14),
    "gan_conv2d_wgrad_bias": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p, c_int, c_void_p] + [c_int] * 12 + [c_void_p, c_int64, c_float, c_int, c_float, c_int, c_int, c_void_p]),
    "gan_amd_init": (c_int, []),
    "gan_conv2d_wgrad_static_max_floats": (c_int64, []),
    "gan_conv2d_wgrad_static": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p] + [c_int] * 12 + [c_float, c_int, c_float, c_int, c_void_p]),
    "gan_modconv_bwd_pre": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_modconv_bwd_post": (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p]),
    "gan_scale_dot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_demod_fwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "gan_demod_bwd": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "gan_demod_multi": (c_int, [c_int, c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "gan_pack_weight": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_int] * 7 + [c_void_p]),
    "gan_pad_channels": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int] + [c_int] * 5 + [c_void_p]),
    "gan_pack_weights_multi": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "gan_pack_weight_pair": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_int] * 5 + [c_void_p]),
    "gan_up2_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gan_mbstd_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_mbstd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_mbstd_bwd2": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_mbstd_concat": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_mbstd_concat_adj": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "gan_chan_stats": (c_int, [c_void_p] * 3 + [c_int, c_int64, c_int, c_float, c_int, c_void_p]),
    "gan_chan_affine": (c_int, [c_void_p] * 4 + [c_int, c_int64, c_int, c_float, c_int, c_void_p]),
    "gan_chan_lin2": (c_int, [c_void_p] * 7 + [c_int, c_int64, c_int, c_float, c_int, c_void_p]),
    "gan_layernorm2d_fwd": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_float, c_float, c_int, c_void_p]),
    "gan_layernorm2d_bwd": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_float, c_int, c_void_p]),
    "gan_space_to_depth": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "gan_rgb_conv1x1_fwd": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 5 + [c_float, c_void_p]),
    "gan_rgb_conv1x1_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "gan_mapping_fwd": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
    "gan_mapping_bwd_workspace": (c_int64, [c_int, c_int]),
    "gan_mapping_bwd": (c_int, [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "gan_split_cols": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gan_diffaug_workspace": (c_int64, [c_int]),
    "gan_diffaug": (c_int, [c_void_p] + [c_int64] * 4 + [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_int64, c_int, c_void_p]),
    "gan_rgb_conv1x1_wgrad_workspace": (c_int64, [c_int, c_int]),
    "gan_rgb_conv1x1_wgrad": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64] + [c_int] * 5 + [c_void_p]),
    "gan_ema_lerp": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "gan_conv2d_fwd_dot_supported": (c_int, [c_int] * 14),
    "gan_conv2d_fwd_dot": (c_int, [c_void_p] * 6 + [c_int] * 13 + [c_float, c_int, c_void_p]),
    "gan_conv2d_fwd_short_supported": (c_int, [c_int] * 12),
    "gan_conv2d_fwd_short": (c_int, [c_void_p] * 4 + [c_float, c_float, c_void_p, c_void_p, c_int, c_float] + [c_int] * 11 + [c_void_p]),
    "gan_scale_rows": (c_int, [c_void_p] * 5 + [c_int, c_int64, c_void_p]),
    "gan_scale_rows_bwd": (c_int, [c_void_p] * 5 + [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "gan_adam_step": (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_int] + [c_float] * 5 + [c_int, c_void_p]),
}

_lib = None


class GanAmdError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libgan_amd.so once; raise loudly if it has not been built (`python __graft_entry__.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GanAmdError(
            f"{LIB_PATH} not found: build the HIP extension first (make -C gan_amd/csrc, or __graft_entry__.build()). "
            "gan_amd has no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.gan_amd_version() != ABI_VERSION:
        raise GanAmdError(f"libgan_amd.so ABI {lib.gan_amd_version()} != expected {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES.keys())


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().gan_amd_last_error().decode(errors="replace")
        raise GanAmdError(f"{what} failed ({status}): {msg}")


def dtype_code(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return GAN_F32
    if t.dtype == torch.bfloat16:
        return GAN_BF16
    raise GanAmdError(f"gan_amd kernels take float32 or bfloat16 activations, got {t.dtype}")


def require_gpu(*tensors: Optional[Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GanAmdError("gan_amd ops run on a ROCm device only (no CPU path); got a tensor on " + str(t.device))


def ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def stream_of(t: Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream
