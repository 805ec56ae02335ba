"""ctypes binding of libspk_hip.so (C ABI: include/spk.h).

There is deliberately NO fallback: if the library is missing or a tensor is not on a HIP device
the call raises.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libspk_hip.so")

# epilogue flags (include/spk.h)
EPI_BIAS, EPI_NOISE, EPI_LRELU, EPI_STYLE, CONV_UPSAMPLE2X, EPI_ACCUM, EPI_STATS, CONV_IN_AFFINE_RELU = \
    1, 2, 4, 8, 16, 32, 64, 128

c_float_p = C.c_void_p  # device pointers travel as integers


class Conv2dDesc(C.Structure):
    """Mirror of spk_conv2d_desc (include/spk.h)."""
    _fields_ = [("x", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("noise_w", C.c_void_p),
                ("noise", C.c_void_p), ("style", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
                ("stats", C.c_void_p), ("y", C.c_void_p), ("y_pre", C.c_void_p),
                ("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Hin", C.c_int32), ("Win", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("style_stride", C.c_int32), ("flags", C.c_uint32), ("lrelu_slope", C.c_float),
                ("out_scale", C.c_float), ("config", C.c_int32), ("ksplit", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("out_scale_bc", C.c_void_p), ("act_gain", C.c_float),
                ("groups", C.c_int32), ("group_in_stride", C.c_int32), ("stats_slots", C.c_int32),
                ("accum_half", C.c_void_p), ("out_scale_dev", C.c_void_p),
                ("rgb_w", C.c_void_p), ("rgb_bias", C.c_void_p), ("rgb_y", C.c_void_p), ("rgb_channels", C.c_int32), ("reserved", C.c_int32)]


CONV_IN_BATCH_SCALE = 256
CONV_UP_FIR1331 = 512
CONV_DGRAD_S2 = 1024
CONV_TRANSPOSE4X4_S2 = 2048
CONV_BF16X3 = 4096
EPI_ACCUM_HALF = 8192
CONV_WINOGRAD = 16384
EPI_TORGB = 32768


FC_MAX_GROUPS = 16
DEMOD_MAX_GROUPS = 16


BN_LIST_MAX = 64


class BnReplayItem(C.Structure):
    """``spk_bn_replay_item`` (include/spk.h)."""
    _fields_ = [("stats", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("count", C.c_int64),
                ("C", C.c_int32), ("reserved", C.c_int32)]


class FcGroup(C.Structure):
    """``spk_fc_group`` (include/spk.h)."""
    _fields_ = [("x", C.c_void_p), ("x_stride", C.c_int64), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("out_stride", C.c_int64), ("I", C.c_int32), ("O", C.c_int32), ("wmul", C.c_float), ("bmul", C.c_float),
                ("slope", C.c_float), ("reserved", C.c_int32)]


class FcBwdGroup(C.Structure):
    """``spk_fc_bwd_group`` (include/spk.h)."""
    _fields_ = [("dout", C.c_void_p), ("dout_stride", C.c_int64), ("out", C.c_void_p), ("x", C.c_void_p), ("x_stride", C.c_int64),
                ("w", C.c_void_p),
                ("dx", C.c_void_p), ("dx_stride", C.c_int64), ("dw", C.c_void_p), ("db", C.c_void_p), ("I", C.c_int32),
                ("O", C.c_int32), ("wmul", C.c_float), ("bmul", C.c_float), ("slope", C.c_float), ("reserved", C.c_int32)]


class DemodGroup(C.Structure):
    """``spk_demod_group`` (include/spk.h)."""
    _fields_ = [("w", C.c_void_p), ("s", C.c_void_p), ("d", C.c_void_p), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("taps", C.c_int32), ("scale", C.c_float)]


# ---- launch lists (include/spk.h: spk_launch_list) ----
OP_CONV2D, OP_FC, OP_FC_GROUPED, OP_BIAS_NOISE_STYLE, OP_TORGB, OP_DEMOD_GROUPED, OP_PIXELNORM, OP_UPSAMPLE2X = 1, 2, 3, 4, 5, 6, 7, 8
ALL_OPS = 0xFFFFFFFF


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("desc", C.c_void_p)]


class FcArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_stride", C.c_int64), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("out_stride", C.c_int64), ("B", C.c_int32), ("I", C.c_int32), ("O", C.c_int32), ("wmul", C.c_float),
                ("bmul", C.c_float), ("slope", C.c_float)]


class FcGroupedArgs(C.Structure):
    _fields_ = [("groups", C.c_void_p), ("n_groups", C.c_int32), ("B", C.c_int32)]


class BiasNoiseStyleArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_batch_stride", C.c_int64), ("bias", C.c_void_p), ("noise_w", C.c_void_p),
                ("noise", C.c_void_p), ("style", C.c_void_p), ("style_stride", C.c_int64), ("y", C.c_void_p),
                ("B", C.c_int32), ("C", C.c_int32), ("HW", C.c_int32), ("reserved", C.c_int32)]


class ToRGBArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("mod", C.c_void_p), ("bias", C.c_void_p), ("skip", C.c_void_p),
                ("y", C.c_void_p), ("B", C.c_int32), ("C", C.c_int32), ("O", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("in_scale", C.c_float)]


class DemodGroupedArgs(C.Structure):
    _fields_ = [("groups", C.c_void_p), ("n_groups", C.c_int32), ("B", C.c_int32), ("eps", C.c_float), ("reserved", C.c_int32)]


class Upsample2xArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("planes", C.c_int64), ("Hin", C.c_int32), ("Win", C.c_int32),
                ("zero_border", C.c_int32), ("reserved", C.c_int32)]


class PixelNormArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("B", C.c_int32), ("C", C.c_int32), ("HW", C.c_int64), ("eps", C.c_float),
                ("sqrt_form", C.c_int32)]


SN_MAX_GROUPS = 24


class SnGroup(C.Structure):
    """``spk_sn_group`` (include/spk.h)."""
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("w_hat", C.c_void_p), ("sigma", C.c_void_p),
                ("dw", C.c_void_p), ("R", C.c_int32), ("C", C.c_int32), ("accumulate", C.c_int32), ("reserved", C.c_int32)]


class WgradDesc(C.Structure):
    """Mirror of spk_wgrad_desc (include/spk.h)."""
    _fields_ = [("g", C.c_void_p), ("x", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
                ("dw", C.c_void_p),
                ("B", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Hin", C.c_int32), ("Win", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("flags", C.c_uint32), ("scale", C.c_float), ("accumulate", C.c_int32), ("splits", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("groups", C.c_int32), ("group_in_stride", C.c_int32),
                ("fold", C.c_int32), ("g_scale", C.c_void_p)]


_PROTOTYPES = {
    "spk_modconv_demod": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_void_p]),
    "spk_modconv_demod_grouped": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "spk_upfirdn2d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "spk_conv1x1_small_mod_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "spk_torgb_mod_skip_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "spk_pixelnorm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "spk_instance_norm_affine_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                               C.c_int64, C.c_float, C.c_void_p]),
    "spk_instance_norm_affine_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "spk_blur2d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
    "spk_upscale2d_nearest_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "spk_fade_in_tanh_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "spk_pixelnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "spk_blur2d_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
    "spk_upscale2d_nearest_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "spk_bn_bwd_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 +
                          [C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "spk_bn_bwd_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 +
                         [C.c_int64, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "spk_bn_bwd_apply_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 +
                              [C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "spk_dilate2x": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "spk_maxpool3x3s2_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "spk_conv2d_wgrad_workspace_bytes": (C.c_int64, [C.c_int] * 9),
    "spk_conv2d_wgrad": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "spk_conv2d_wgrad_up_supported": (C.c_int, [C.c_int] * 5),
    "spk_conv2d_wgrad_mod_supported": (C.c_int, [C.c_int] * 6),
    "spk_conv2d_wgrad_wino_supported": (C.c_int, [C.c_int] * 5),
    "spk_conv2d_wgrad_wino_splits": (C.c_int, [C.c_int] * 6),
    "spk_conv2d_wgrad_wino_workspace_bytes": (C.c_int64, [C.c_int] * 6),
    "spk_conv2d_wgrad_wino": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spk_wgrad_reduce_slabs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "spk_modconv_epi_finish": (C.c_int, [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "spk_modconv_dx_finish": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]),
    "spk_modconv_demod_bwd_workspace_bytes": (C.c_int64, [C.c_int] * 3),
    "spk_modconv_demod_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int64] + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "spk_torgb_mod_bwd_data": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "spk_epilogue_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "spk_upsample2x_bilinear_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "spk_conv1x1_small_bwd_blocks": (C.c_int, [C.c_int, C.c_int64]),
    "spk_conv1x1_small_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "spk_fc_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "spk_spectral_norm_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int]),
    "spk_spectral_norm_grouped": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "spk_spectral_norm_bwd_grouped": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "spk_launch_list": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]),
    "spk_version": (C.c_char_p, []),
    "spk_last_error": (C.c_char_p, []),
    "spk_conv2d_num_configs": (C.c_int, []),
    "spk_conv2d_config_valid": (C.c_int, [C.c_int] * 4),
    "spk_conv2d_pick_config": (C.c_int, [C.c_int] * 8),
    "spk_conv2d_dgrad_s2_config": (C.c_int, [C.c_int] * 5),
    "spk_conv2d_dgrad_s2_workspace_bytes": (C.c_int64, [C.c_int] * 8),
    "spk_conv2d_config_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "spk_conv2d_packed_floats": (C.c_int64, [C.c_int] * 5),
    "spk_conv2d_workspace_bytes": (C.c_int64, [C.c_int] * 10),
    "spk_conv2d_workspace_bytes_grouped": (C.c_int64, [C.c_int] * 11),
    "spk_conv2d_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    "spk_conv2d_pack_weights_list": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p]),
    "spk_conv2d_fwd": (C.c_int, [C.POINTER(Conv2dDesc), C.c_void_p]),
    "spk_conv2d_packed_bytes_bf16x3": (C.c_int64, [C.c_int, C.c_int]),
    "spk_conv2d_pack_weights_bf16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spk_conv2d_pack_weights_bf16x3_tf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "spk_conv2d_bf16x3_supported": (C.c_int, [C.c_int] * 5),
    "spk_conv2d_bf16x3_fwd": (C.c_int, [C.POINTER(Conv2dDesc), C.c_void_p]),
    "spk_conv2d_packed_bytes_wino": (C.c_int64, [C.c_int, C.c_int]),
    "spk_conv2d_pack_weights_wino": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "spk_conv2d_pack_weights_wino_list": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "spk_conv2d_wino_supported": (C.c_int, [C.c_int] * 5),
    "spk_conv2d_wino_ksplit": (C.c_int, [C.c_int] * 6),
    "spk_conv2d_wino_workspace_bytes": (C.c_int64, [C.c_int] * 6),
    "spk_conv2d_wino_fwd": (C.c_int, [C.POINTER(Conv2dDesc), C.c_void_p]),
    "spk_conv2d_stats_slots": (C.c_int, [C.c_int] * 9),
    "spk_bn_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                  C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "spk_bn_replay_list": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "spk_bn_add_relu_fwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "spk_maxpool3x3s2_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "spk_global_avgpool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "spk_fc_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                             C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "spk_fc_grouped_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spk_fc_grouped_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "spk_bias_noise_style_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "spk_conv1x1_expand_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "spk_conv1x1_small_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int64, C.c_float, C.c_void_p]),
    "spk_upsample2x_bilinear_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "spk_plane_sums_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "spk_upsample2x_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


class SpkError(RuntimeError):
    pass


def exported_symbols():
    """Names every entry point include/spk.h declares (used by the CPU-side ABI test)."""
    return sorted(_PROTOTYPES)


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise SpkError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                                   f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
                h = C.CDLL(LIB_PATH)
                for name, (res, args) in _PROTOTYPES.items():
                    fn = getattr(h, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = h
    return _lib


def check(code: int, what: str):
    if code != 0:
        raise SpkError(f"{what} failed ({code}): {lib().spk_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dptr(t, name="tensor"):
    """Device pointer of a contiguous fp32 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SpkError(f"{name}: expected a HIP device tensor, got {t.device} -- speak-hack_amd has no CPU path")
    if t.dtype != torch.float32:
        raise SpkError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise SpkError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()
