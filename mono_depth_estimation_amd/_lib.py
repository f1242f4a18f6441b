"""ctypes binding of libmde_hip.so (include/mde_hip.h).

The product path has no CPU fallback: if the shared library is missing or an entry point
is absent this module raises at import/first use, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class MdeError(RuntimeError):
    pass


# MDE_ACT_DTYPE=fp16 selects the build whose 16-bit storage type is IEEE half (libmde_hip_f16.so: the same sources with
# -DMDE_ACT_F16; BASELINE configuration 5's precision) for the whole process; default bf16.  ops.ACT_DTYPE is the matching torch dtype.
ACT_NAME = os.environ.get("MDE_ACT_DTYPE", "bf16").lower()
if ACT_NAME in ("float16", "half", "f16"):
    ACT_NAME = "fp16"
_FP16_NOTED = False


def note_fp16_backward():
    """The fp16 storage build flushes activation gradients below 6e-8 to zero: a backward pass through it needs the caller's
    loss scale (the reference's precision=16 run brings a GradScaler; bench.py applies a static one).  Said once per process,
    on the first backward -- the library cannot see whether the gradient it is handed was scaled."""
    global _FP16_NOTED
    if ACT_NAME == "fp16" and not _FP16_NOTED and os.environ.get("MDE_FP16_QUIET", "0") != "1":
        _FP16_NOTED = True
        import warnings
        warnings.warn("mono_depth_estimation_amd: fp16 storage build (MDE_ACT_DTYPE=fp16) -- backward expects a SCALED loss "
                      "(torch.cuda.amp.GradScaler, or a static loss scale as bench.py's); unscaled fp16 activation gradients "
                      "underflow (INTEGRATION.md, section 1).  MDE_FP16_QUIET=1 silences this note.", stacklevel=3)


if ACT_NAME not in ("bf16", "fp16"):
    raise MdeError("MDE_ACT_DTYPE=%s: bf16 (default) or fp16" % ACT_NAME)
LIB_NAME = "libmde_hip_f16.so" if ACT_NAME == "fp16" else "libmde_hip.so"
LIB_PATH = os.environ.get("MDE_LIB_PATH") or os.path.join(_HERE, LIB_NAME)   # override: diagnostic builds only
ABI_VERSION = 12
MAX_TAPS = 32


class BnRed(C.Structure):
    """mde_bn_red (include/mde_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("save_mean", C.c_void_p), ("save_rstd", C.c_void_p), ("mask_scale", C.c_void_p),
                ("mask_shift", C.c_void_p), ("relu_bits", C.c_void_p), ("part", C.c_void_p), ("x_ld", C.c_int32),
                ("x2", C.c_void_p), ("save_mean2", C.c_void_p), ("save_rstd2", C.c_void_p), ("part2", C.c_void_p), ("x2_ld", C.c_int32),
                ("add", C.c_void_p), ("add_bits", C.c_void_p)]


class BnFin(C.Structure):
    """mde_bn_fin (include/mde_hip.h)."""
    _fields_ = [("part", C.c_void_p), ("part_ld", C.c_int32), ("mean_in", C.c_void_p), ("var_in", C.c_void_p), ("count", C.c_int64),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("rmean", C.c_void_p), ("rvar", C.c_void_p), ("momentum", C.c_float),
                ("eps", C.c_float), ("scale", C.c_void_p), ("shift", C.c_void_p), ("smean", C.c_void_p), ("srstd", C.c_void_p),
                ("zero", C.c_void_p), ("zero_n", C.c_int64)]


class BnBfin(C.Structure):
    """mde_bn_bfin (include/mde_hip.h)."""
    _fields_ = [("part", C.c_void_p), ("part_ld", C.c_int32), ("count", C.c_int64), ("gamma", C.c_void_p), ("srstd", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("zero", C.c_void_p), ("zero_n", C.c_int64)]


class ConvDesc(C.Structure):
    """mde_conv_desc (include/mde_hip.h)."""
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("ld_in", C.c_int32), ("C", C.c_int32),
        ("in_bytes", C.c_uint32),
        ("GH", C.c_int32), ("GW", C.c_int32), ("sy", C.c_int32), ("sx", C.c_int32), ("ntaps", C.c_int32),
        ("dy", C.c_int16 * MAX_TAPS), ("dx", C.c_int16 * MAX_TAPS), ("wtap", C.c_int16 * MAX_TAPS),
        ("wtaps_total", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32), ("ld_out", C.c_int32),
        ("osy", C.c_int32), ("osx", C.c_int32), ("ooy", C.c_int32), ("oox", C.c_int32),
        ("ncols", C.c_int32), ("accumulate", C.c_int32), ("grouped", C.c_int32),
    ]


class WgradDesc(C.Structure):
    """mde_wgrad_desc (include/mde_hip.h)."""
    _fields_ = [
        ("N", C.c_int32), ("GH", C.c_int32), ("GW", C.c_int32),
        ("ld_d", C.c_int32), ("Cd", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("ld_g", C.c_int32), ("Cg", C.c_int32),
        ("d_bytes", C.c_uint32), ("g_bytes", C.c_uint32),
        ("sy", C.c_int32), ("sx", C.c_int32), ("ntaps", C.c_int32),
        ("dy", C.c_int16 * MAX_TAPS), ("dx", C.c_int16 * MAX_TAPS), ("otap", C.c_int16 * MAX_TAPS),
        ("otaps_total", C.c_int32), ("rows_from_gathered", C.c_int32), ("ksplit", C.c_int32), ("group_size", C.c_int32),
    ]


_P, _I, _L, _F, _Z, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint

# name -> (restype, argtypes); every symbol include/mde_hip.h declares
SIGNATURES = {
    "mde_last_error": (C.c_char_p, []),
    "mde_act_dtype": (_I, []),
    "mde_abi_version": (_I, []),
    "mde_device_cu_count": (_I, [C.POINTER(_I)]),
    "mde_det_scratch_bytes": (_Z, [_L]),
    "mde_set_deterministic": (_I, [_I, _P, _P, _L]),
    "mde_deterministic": (_I, []),
    "mde_det_flush": (_I, [_P]),
    "mde_conv_gemm": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "mde_conv_gemm_act": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, _P]),
    "mde_conv_gemm_bnred": (_I, [C.POINTER(ConvDesc), _P, _P, _P, C.POINTER(BnRed), _P]),
    "mde_conv_wgrad": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P]),
    "mde_conv_wgrad_ws": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _L, _P]),
    "mde_conv_wgrad_ws_bytes": (_L, [C.POINTER(WgradDesc)]),
    "mde_stem_conv_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "mde_stem_conv_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mde_stem_conv_fwd_c": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mde_stem_conv_wgrad_c": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mde_head_conv_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mde_head_conv_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mde_stat_slots": (_I, []),
    "mde_bn_stats": (_I, [_P, _L, _I, _I, _P, _P]),
    "mde_bn_finalize": (_I, [_P, _L, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mde_bn_moments": (_I, [_P, _L, _I, _P, _P, _P]),
    "mde_bn_finalize_moments": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mde_bn_eval_scale_shift": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "mde_bn_apply": (_I, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _L, _I, _I, _P]),
    "mde_bn_bwd_reduce": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "mde_bn_bwd_reduce2": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "mde_bn_bwd_apply2": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P, _I, _P]),
    "mde_bn_bwd_finalize": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P]),
    "mde_bn_bwd_apply": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _I, _P, _I, _P]),
    "mde_bn_apply_fin": (_I, [_P, _I, C.POINTER(BnFin), _P, _I, C.POINTER(BnFin), _P, _I, _P, _L, _I, _I, _P]),
    "mde_bn_bwd_apply_fin": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, C.POINTER(BnBfin), _L, _I, _I, _P, _I, _I, _P, _I, _P]),
    "mde_bn_bwd_apply2_fin": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, C.POINTER(BnBfin), C.POINTER(BnBfin), _L, _I, _P, _I, _P, _I, _P]),
    "mde_pixel_shuffle2": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mde_maxpool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mde_maxpool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mde_upsample_sigmoid_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mde_upsample_sigmoid_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mde_pw_fwd": (_I, [_P, _I, _P, _P, _I, _P, _I, _L, _I, _I, _P]),
    "mde_pw_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, _P, _L, _I, _I, _P]),
    "mde_spatial_sum": (_I, [_P, _I, _I, _L, _I, _F, _P, _I, _P]),
    "mde_spatial_bcast": (_I, [_P, _I, _F, _P, _I, _I, _L, _I, _I, _P]),
    "mde_gate_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _L, _I, _P]),
    "mde_gate_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _L, _I, _P]),
    "mde_resize_bilinear_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_resize_bilinear_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_nearest2_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "mde_sum2x2": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mde_spread2x2": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mde_softmax_head_fwd": (_I, [_P, _I, _P, _P, _P, _I, _L, _I, _P]),
    "mde_softmax_head_bwd": (_I, [_P, _P, _P, _P, _I, _P, _I, _L, _I, _P]),
    "mde_map_act_fwd": (_I, [_P, _P, _L, _I, _F, _P]),
    "mde_map_act_bwd": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "mde_to_nchw_act_fwd": (_I, [_P, _I, _P, _P, _I, _L, _I, _I, _F, _P]),
    "mde_to_nchw_act_bwd": (_I, [_P, _P, _P, _I, _P, _I, _L, _I, _I, _F, _P]),
    "mde_image_residual_fwd": (_I, [_P, _P, _P, _I, _L, _I, _P]),
    "mde_image_residual_bwd": (_I, [_P, _P, _P, _P, _I, _L, _I, _P]),
    "mde_plane_depth_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _P]),
    "mde_plane_depth_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "mde_map_to_slot": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mde_slot_to_map_add": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "mde_pack_grouped": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mde_dwconv3x3_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_dwconv3x3_dgrad": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_dwconv3x3_wgrad": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mde_maxpool_view_fwd": (_I, [_P, _I, _I, _L, _I, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "mde_maxpool_view_bwd": (_I, [_P, _I, _P, _P, _I, _I, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_maxpool_fwd2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mde_maxpool_bwd2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mde_chan_scale": (_I, [_P, _I, _P, _P, _I, _I, _L, _I, _I, _P]),
    "mde_avgpool_flat_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_avgpool_flat_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mde_ordinal_fwd": (_I, [_P, _I, _P, _P, _I, _L, _I, _P]),
    "mde_ordinal_bwd": (_I, [_P, _P, _I, _P, _I, _I, _L, _I, _P]),
    "mde_weighted_pool_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _L, _I, _P]),
    "mde_weighted_pool_bwd": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _I, _L, _I, _P]),
    "mde_combine3_fwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _L, _P, _P]),
    "mde_combine3_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _L, _P, _P, _P, _P, _P]),
    "mde_aug_to_u8": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "mde_aug_resample_u8": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "mde_aug_affine_nearest_u8": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "mde_aug_crop_flip_to_float": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mde_aug_crop_flip_to_float_c": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "mde_aug_flip_pad_crop": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mde_ord_loss_ws_bytes": (_Z, []),
    "mde_ord_loss_fwd": (_I, [_P, _P, _I, _I, _L, _P, _P, _P]),
    "mde_ord_loss_bwd": (_I, [_P, _P, _I, _I, _L, _P, _P, _P]),
    "mde_silog_ws_bytes": (_Z, []),
    "mde_silog_fwd": (_I, [_P, _P, _L, _F, _P, _P, _P]),
    "mde_silog_bwd": (_I, [_P, _P, _L, _F, _P, _P, _P, _P]),
    "mde_masked_loss_ws_bytes": (_Z, []),
    "mde_masked_loss_fwd": (_I, [_I, _P, _P, _L, _P, _P, _P]),
    "mde_masked_loss_bwd": (_I, [_I, _P, _P, _L, _P, _P, _P, _P]),
    "mde_masked_depth_ws_bytes": (_Z, [_I]),
    "mde_masked_depth_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mde_masked_depth_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mde_midas_ws_bytes": (_Z, [_I]),
    "mde_midas_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _I, _P, _P, _P]),
    "mde_midas_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mde_procrustes_ws_bytes": (_Z, [_I]),
    "mde_procrustes_fwd": (_I, [_P, _P, _I, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P]),
    "mde_procrustes_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mde_scale_and_shift": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mde_wcel_ws_bytes": (_Z, [_I]),
    "mde_wcel_fwd": (_I, [_P, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P]),
    "mde_wcel_bwd": (_I, [_P, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P]),
    "mde_bins_to_depth_fwd": (_I, [_P, _P, _I, _I, _L, _P, _P]),
    "mde_bins_to_depth_bwd": (_I, [_P, _P, _P, _I, _I, _L, _P, _P]),
    "mde_depth_to_bins": (_I, [_P, _L, _F, _F, _F, _F, _I, _P, _P]),
    "mde_vnl_ws_bytes": (_Z, [_I, _I]),
    "mde_vnl_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P]),
    "mde_vnl_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "mde_vnl_head_depth_fwd": (_I, [_P, _I, _P, _P, _L, _I, _P, _P, _P, _P]),
    "mde_vnl_head_wcel_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "mde_vnl_head_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P]),
    "mde_stdepth_ws_bytes": (_Z, []),
    "mde_stdepth_scratch_elems": (_Z, [_I, _I, _I, _I, _U]),
    "mde_stdepth_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _U, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "mde_stdepth_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _U, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    "mde_metrics_ws_bytes": (_Z, []),
    "mde_depth_metrics": (_I, [_P, _P, _L, _P, _P, _P]),
    "mde_ssim_metric_ws_bytes": (_Z, []),
    "mde_ssim_metric": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mde_adam_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _P]),
    "mde_adamw_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _P]),
    "mde_sgd_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P]),
    "mde_cast_bf16": (_I, [_P, _P, _L, _P]),
    "mde_param_fingerprint_state_bytes": (_Z, []),
    "mde_param_fingerprint": (_I, [_P, _L, _P, _P]),
    "mde_refresh_if_changed": (_I, [_P, _P, _P, _P, _I, _L, _L, _P, _P]),
    "mde_pack_wt": (_I, [_P, _P, _I, _I, _I, _P]),
    "mde_pack_wt_batch": (_I, [_P, _P, _P, _I, _L, _P]),
    "mde_pack_split_batch": (_I, [_P, _P, _P, _I, _L, _I, _P]),
    "mde_pack_grouped_split": (_I, [_P, _P, _I, _I, _I, _P]),
    "mde_nchw_to_nhwc_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mde_nchw_to_nhwc_bf16_pad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mde_nhwc_bf16_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mde_nchw_to_nhwc_split16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mde_stem_weight_split16": (_I, [_P, _P, _L, _I, _I, _P]),
}

_lib = None


def load():
    """Load libmde_hip.so once and bind every declared symbol. Raises MdeError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MdeError(
            "%s not found at %s — build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or mono_depth_estimation_amd/csrc/build.sh). There is no CPU fallback." % (LIB_NAME, LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise MdeError("libmde_hip.so lacks symbols declared in include/mde_hip.h: %s" % ", ".join(missing))
    if lib.mde_abi_version() != ABI_VERSION:
        raise MdeError("%s ABI %d != binding ABI %d" % (LIB_NAME, lib.mde_abi_version(), ABI_VERSION))
    if lib.mde_act_dtype() != (1 if ACT_NAME == "fp16" else 0):
        raise MdeError("%s stores %s, the process asked for %s" % (LIB_NAME, ("bf16", "fp16")[lib.mde_act_dtype()], ACT_NAME))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mde_last_error()
        raise MdeError("%s failed (%d): %s" % (what or "libmde_hip call", rc, (msg or b"").decode()))
