"""ctypes binding of libmstg_hip.so (include/mstg_hip.h).  No fallback: without the library every op raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSTG_LIB") or os.path.join(HERE, "libmstg_hip.so")  # MSTG_LIB: A/B of two builds in one GPU call

ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_TANH, ACT_GELU = 0, 1, 2, 3, 4
LOSS_L1, LOSS_MSE = 0, 1


class ConvDesc(C.Structure):
    """mirror of mstg_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad", "dil", "transposed",
        "x_nchw", "y_nchw", "x_ctot", "x_coff", "y_ctot", "y_coff", "act", "accumulate")]


class F16ConvDesc(C.Structure):
    """mirror of mstg_f16_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in (
        "kind", "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "K", "stride", "pad", "dil", "src_nchw_f32", "dst_nchw", "act")]


_vp, _fp, _sz, _i, _f = C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_float
_dp = C.POINTER(ConvDesc)
_hp = C.POINTER(F16ConvDesc)

# name -> (restype, argtypes); the test-suite checks that every symbol declared in include/mstg_hip.h is here
SIGNATURES = {
    "mstg_version": (C.c_char_p, []),
    "mstg_arch": (C.c_char_p, []),
    "mstg_last_error": (C.c_char_p, []),
    "mstg_env_refresh": (None, []),
    "mstg_prof_enable": (_i, [_i]),
    "mstg_prof_count": (_i, []),
    "mstg_prof_get": (_i, [_i, C.c_char_p, _sz, C.POINTER(C.c_float)]),
    "mstg_conv2d_kernel_name": (C.c_char_p, [_dp, _i]),
    "mstg_conv2d_workspace_bytes": (_sz, [_dp]),
    "mstg_conv2d_fwd": (_i, [_dp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_conv2d_dgrad": (_i, [_dp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_conv2d_dgrad_bsums_supported": (_i, [_dp]),
    "mstg_conv2d_dgrad_bsums_workspace_bytes": (_sz, [_dp]),
    "mstg_conv2d_dgrad_bsums": (_i, [_dp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _i, _vp]),
    "mstg_conv2d_fwd_cached": (_i, [_dp, _fp, _fp, _fp, _fp, _vp, _sz, _i, _vp]),
    "mstg_conv2d_fwd_norm_cached": (_i, [_dp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _i, _vp]),
    "mstg_conv2d_dgrad_cached": (_i, [_dp, _fp, _fp, _fp, _vp, _sz, _i, _vp]),
    "mstg_msblock_fwd_cached": (_i, [_fp] * 10 + [_i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "mstg_msblock_dgrad_cached": (_i, [_fp] * 7 + [_i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "mstg_conv2d_wgrad_workspace_bytes": (_sz, [_dp]),
    "mstg_conv2d_wgrad": (_i, [_dp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_norm_workspace_bytes": (_sz, [_i, _i, _i]),
    "mstg_norm_act_fwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_norm_act_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_window_attn_core_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_core_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_ws_supported": (_i, [_i, _i]),
    "mstg_window_attn_ws_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_ws_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_fused_supported": (_i, [_i]),
    "mstg_window_attn_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mstg_conv2d_wgrad_norm_supported": (_i, [_dp]),
    "mstg_conv2d_wgrad_norm": (_i, [_dp, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_conv2d_fwd_norm_supported": (_i, [_dp]),
    "mstg_conv2d_fwd_stats_pays": (_i, [_dp]),
    "mstg_conv2d_fwd_norm_workspace_bytes": (_sz, [_dp]),
    "mstg_conv2d_fwd_norm": (_i, [_dp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _sz, _vp]),
    "mstg_window_attn_norm_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_norm_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mstg_window_attn_norm_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_norm_apply_fwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_norm_stats": (_i, [_fp, _fp, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_norm_bwd_apply": (_i, [_fp, _fp, _fp, _fp, _i, _fp, _i, _i, _i, _i, _vp]),
    "mstg_window_attn_norm_sums_split": (_i, []),
    "mstg_window_attn_bwd_direct": (_i, [_fp] * 11 + [_i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_window_attn_norm_bwd_direct": (_i, [_fp] * 12 + [_i, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_window_attn_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_act_fwd": (_i, [_fp, _fp, _sz, _i, _vp]),
    "mstg_act_bwd": (_i, [_fp, _fp, _fp, _sz, _i, _vp]),
    "mstg_loss_workspace_bytes": (_sz, [_sz]),
    "mstg_loss_mean_fwd": (_i, [_fp, _fp, _f, _sz, _i, _fp, _vp, _sz, _vp]),
    "mstg_loss_mean_bwd": (_i, [_fp, _fp, _f, _sz, _i, _fp, _f, _fp, _fp, _vp]),
    "mstg_channel_sum_workspace_bytes": (_sz, [_sz, _i]),
    "mstg_channel_sum": (_i, [_fp, _sz, _i, _i, _i, _f, _fp, _vp, _sz, _vp]),
    "mstg_plane_sum_workspace_bytes": (_sz, [_i, _i, _sz]),
    "mstg_plane_sum": (_i, [_fp, _i, _i, _sz, _f, _fp, _vp, _sz, _vp]),
    "mstg_segment_mean_fwd": (_i, [_fp, _i, _sz, _i, _fp, _vp]),
    "mstg_segment_mean_bwd": (_i, [_fp, _i, _sz, _i, _fp, _vp]),
    "mstg_maxpool2x2_fwd": (_i, [_fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "mstg_maxpool2x2_bwd": (_i, [_fp, _vp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_gram_workspace_bytes": (_sz, [_i, _i, _i]),
    "mstg_gram_fwd": (_i, [_fp, _fp, _i, _i, _i, _f, _vp, _sz, _vp]),
    "mstg_gram_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _f, _vp]),
    "mstg_adam_step_flat": (_i, [_fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _i, _vp, _vp]),
    "mstg_msblock_fused_supported": (_i, [_i]),
    "mstg_msblock_fwd_workspace_bytes": (_sz, [_i]),
    "mstg_msblock_fwd": (_i, [_fp] * 10 + [_i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_msblock_dgrad_workspace_bytes": (_sz, [_i]),
    "mstg_msblock_dgrad": (_i, [_fp] * 7 + [_i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_msblock_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mstg_msblock_wgrad": (_i, [_fp] * 10 + [_i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_spectral_norm_workspace_bytes": (_sz, [_i, _i]),
    "mstg_spectral_norm_fwd": (_i, [_fp] * 7 + [_i, _i, _f, _i, _vp, _sz, _vp]),
    "mstg_spectral_norm_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_spectral_norm_group_max": (_i, []),
    "mstg_spectral_norm_group_workspace_bytes": (_sz, [_i, _vp, _vp]),
    "mstg_spectral_norm_group_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _fp, _vp, _vp, _vp, _vp, _f, _i, _vp, _sz, _vp]),
    "mstg_spectral_norm_group_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mstg_f16_conv_plan_bytes": (_sz, [_hp]),
    "mstg_f16_conv_pack": (_i, [_hp] + [_fp] * 8 + [_vp, _sz, _vp]),
    "mstg_f16_conv_partial_bytes": (_sz, [_hp]),
    "mstg_f16_conv_fwd": (_i, [_hp, _vp, _vp, _fp, _vp, _fp, _vp, _sz, _vp]),
    "mstg_f16_conv_fwd_res": (_i, [_hp, _vp, _vp, _fp, _vp, _vp, _fp, _vp, _sz, _vp]),
    "mstg_f16_norm_residual": (_i, [_vp, _vp, _fp, _vp, _i, _i, _i, _vp]),
    "mstg_f16_attn_plan_bytes": (_sz, [_i]),
    "mstg_f16_attn_pack": (_i, [_fp, _fp, _fp, _fp, _i, _vp, _sz, _vp]),
    "mstg_f16_attn_fwd": (_i, [_vp, _fp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mstg_add": (_i, [_fp, _fp, _fp, _sz, _vp]),
    "mstg_weighted_sum_fwd": (_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_float), _i, _i, _fp, _vp]),
    "mstg_weighted_sum_bwd": (_i, [_fp, C.POINTER(C.c_float), _i, _fp, _vp]),
    "mstg_masked_l1_mean_fwd": (_i, [_fp, _fp, _fp, _sz, _fp, _vp, _sz, _vp]),
    "mstg_masked_l1_mean_bwd": (_i, [_fp, _fp, _fp, _sz, _fp, _fp, _vp]),
    "mstg_clip_grad_norm": (_i, [_fp, _sz, _f, _fp, _vp, _sz, _vp]),
    "mstg_structure_map": (_i, [_fp, _fp, _i, _i, _i, _vp]),
    "mstg_ln_mod_fwd": (_i, [_fp] * 7 + [_i, _i, _i, _f, _vp]),
    "mstg_ln_mod_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "mstg_ln_mod_bwd": (_i, [_fp] * 11 + [_i, _i, _i, _i, _vp, _sz, _vp]),
    "mstg_flash_attn_fwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_flash_attn_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mstg_resample_ksize": (_i, [_i, _i, _i]),
    "mstg_resample_coeffs": (_i, [_i, _i, _i, _vp, _vp]),
    "mstg_resample_h_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mstg_resample_v_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mstg_paste_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mstg_u8_to_tensor": (_i, [_vp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, C.c_ulonglong, _i, _vp]),
    "mstg_tensor_to_u8": (_i, [_fp, _i, _i, _vp, _vp]),
    "mstg_blend_u8": (_i, [_vp, _vp, C.c_double, C.c_double, _vp, _vp, _i, _i, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library once.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m mstg_hip.build` (or __graft_entry__.build()). "
                "This package has no CPU or eager-PyTorch fallback.")
        # torch first: it ships its own libamdhip64 / libhsa-runtime64, and this library must bind to THAT runtime (the one that owns
        # torch's device memory and streams).  Loaded before torch, this library pulls in /opt/rocm's copy, the process then holds two
        # HIP runtimes and every launch from here fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mstg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with code {rc}: {msg}")
