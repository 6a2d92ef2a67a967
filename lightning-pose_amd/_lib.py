"""ctypes binding of liblp_hip.so (include/lp_hip.h).  Fails loudly when the library is absent."""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LP_HIP_LIB") or os.path.join(_HERE, "liblp_hip.so")  # LP_HIP_LIB: A/B builds of the same ABI


class LpHipUnavailable(RuntimeError):
    """liblp_hip.so could not be loaded, or an op was given a non-ROCm tensor.  There is no fallback path."""


class LpHipError(RuntimeError):
    pass


class DecodeTables(C.Structure):
    _fields_ = [("row_base", C.c_void_p), ("row_taps", C.c_void_p), ("col_start", C.c_void_p),
                ("col_taps", C.c_void_p), ("colT_start", C.c_void_p), ("colT_taps", C.c_void_p),
                ("ty", C.c_int), ("tx", C.c_int), ("tc", C.c_int)]


class FrameMap(C.Structure):
    _fields_ = [("transforms", C.c_void_p), ("tf_mode", C.c_int), ("bbox", C.c_void_p), ("bbox_stride", C.c_int),
                ("kp_per_view", C.c_int), ("model_h", C.c_float), ("model_w", C.c_float)]


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "Hi", "Wi", "Ci", "Ho", "Wo", "Co", "R", "S", "stride", "pad")]


class BnFuse(C.Structure):
    _fields_ = [("z", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("mask_from_z", C.c_int), ("relu_bits", C.c_void_p), ("sums", C.c_void_p), ("seg_images", C.c_int), ("addend_half", C.c_int)]


class FrameNorm(C.Structure):
    _fields_ = [("mean", C.c_float * 3), ("std", C.c_float * 3)]


class FrameAugment(C.Structure):
    _fields_ = [("has_matrix", C.c_int), ("matrix", C.c_float * 6), ("brightness", C.c_float), ("contrast", C.c_float),
                ("contrast_center", C.c_float), ("shot_factor", C.c_float), ("seed", C.c_ulonglong)]


class GemmBatch(C.Structure):
    _fields_ = [("nb", C.c_int), ("nh", C.c_int)] + [(n, C.c_longlong) for n in ("a_b", "a_h", "b_b", "b_h", "c_b", "c_h")]


HM_MSE, HM_KL, HM_JS = 0, 1, 2
CONV_KERNEL_IGEMM, CONV_KERNEL_PIPE, CONV_KERNEL_WGRAD, CONV_KERNEL_WGRAD_PIPE, CONV_KERNEL_PIPE_HALO, CONV_KERNEL_RES2D = 0, 1, 2, 3, 4, 5   # lp_conv_last_kernel()
CONV_KERNEL_STEM_WGRAD_NB = 8
BORDER_RENORM, BORDER_CLAMP = 0, 1
TF_NONE, TF_SINGLE, TF_PER_FRAME, TF_PER_VIEW = 0, 1, 2, 3

ABI_VERSION = 142   # include/lp_hip.h: LP_HIP_ABI_VERSION - the header these PROTOTYPES were written against (tests/test_abi_and_failloud.py)

_P, _I, _F, _L, _Z = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t

# name -> (restype, argtypes); mirrors include/lp_hip.h one to one
PROTOTYPES = {
    "lp_version": (_I, []),
    "lp_strerror": (C.c_char_p, [_I]),
    "lp_config_reload_env": (_I, []),
    "lp_decode_window": (_I, [_I, _I]),
    "lp_decode_fwd": (_I, [_P, _I, _I, _I, _I, _I, _F, C.POINTER(DecodeTables), C.POINTER(FrameMap), _P, _P, _P, _P, _I, _P]),
    "lp_decode_bwd": (_I, [_P, _I, _I, _I, _I, _I, _F, C.POINTER(DecodeTables), C.POINTER(FrameMap), _P, _P, _P, _P, _I, _I, _P]),
    "lp_frame_map_apply": (_I, [_P, _I, _I, C.POINTER(FrameMap), _I, _P, _P]),
    "lp_heatmap_gen": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "lp_heatmap_gen_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "lp_temporal_heatmap_workspace_bytes": (_Z, [_I, _I]),
    "lp_temporal_heatmap_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _F, _P, _P, _P]),
    "lp_temporal_heatmap_bwd": (_I, [_I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "lp_heatmap_confidence": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "lp_heatmap_mse_workspace_bytes": (_Z, [_I, _I]),
    "lp_heatmap_mse_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_heatmap_mse_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "lp_heatmap_loss_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_heatmap_loss_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "lp_unimodal_mse_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "lp_unimodal_mse_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P]),
    "lp_softmax2d_fwd": (_I, [_P, _L, _L, _L, _I, _I, _I, _P, _P]),
    "lp_softmax2d_bwd": (_I, [_P, _P, _I, _I, _I, _P, _L, _L, _L, _P]),
    "lp_temporal_fwd_bwd": (_I, [_P, _P, _I, _I, _P, _F, _P, _P, _P]),
    "lp_pca_fwd_bwd": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _F, _P, _P, _P]),
    "lp_rmse_fwd": (_I, [_P, _P, _I, _P, _P]),
    "lp_loss_combine": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "lp_loss_combine_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P]),
    "lp_conv_fwd": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P, _P, _I, _I, _P]),
    "lp_conv_dgrad": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lp_conv_dgrad_bits": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P, _P, _I, _P]),
    "lp_gemm_nt": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, C.POINTER(GemmBatch), _P]),
    "lp_gemm_nt_gelu_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "lp_gemm_nt_gelu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "lp_gemm_tn": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, C.POINTER(GemmBatch), _P]),
    "lp_attn_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P, _I, _P]),
    "lp_attn_bwd_kv": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, C.c_float, _P, _P, _I, _I, _I, _P]),
    "lp_attn_rowdot": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "lp_attn_dscores": (_I, [_P, _I, _P, _I, _P, _P, _I, C.c_longlong, C.c_longlong, C.c_float, _P, _I, _I, _I, _I, C.POINTER(GemmBatch), _P]),
    "lp_conv_fwd_bn": (_I, [_P, _P, C.POINTER(ConvGeom), _P, C.POINTER(BnFuse), _P]),
    "lp_conv_last_kernel": (_I, []),
    "lp_stem_fwd_bn": (_I, [_P, _P, C.POINTER(ConvGeom), _P, C.POINTER(BnFuse), _P]),
    "lp_conv_dgrad_bn": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P, _P, C.POINTER(BnFuse), _P]),
    "lp_bn_fold": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P]),
    "lp_conv_fwd_act": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P, _I, _P, _P]),
    "lp_conv_wgrad_workspace_bytes": (_Z, [C.POINTER(ConvGeom), _I]),
    "lp_conv_wgrad": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _I, _P, _Z, _P]),
    "lp_conv_wgrad_bias": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P, _I, _P, C.c_size_t, _P]),
    "lp_stem_fwd": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _P]),
    "lp_stem_wgrad": (_I, [_P, _P, C.POINTER(ConvGeom), _P, _I, _P, _Z, _P]),
    "lp_bn_reduce_workspace_bytes": (_Z, [_I, _I]),
    "lp_bn_stats": (_I, [_P, _I, _I, _P, _P, _Z, _P]),
    "lp_bn_finalize": (_I, [_P, _F, _I, _F, _F, _P, _P, _P, _P, _P]),
    "lp_fxsum_accumulate": (_I, [_P, _I, _P, _P]),
    "lp_bn_finalize_f32": (_I, [_P, _F, _I, _F, _F, _P, _P, _P, _P, _P]),
    "lp_bn_finalize2": (_I, [_P, _F, _F, _I, _F, _F, _P, _P, _P, _P, _P]),
    "lp_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "lp_bn_apply_seg": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_bn_apply_seg_lo": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "lp_bn_apply_seg_rbn": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_bn_bwd_apply_seg": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "lp_bn_bwd_ds_workspace_bytes": (_Z, [_I, _I]),
    "lp_bn_bwd_apply_seg_ds": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "lp_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "lp_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "lp_maxpool_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_maxpool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "lp_bn_relu_maxpool_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_bn_pool_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "lp_bn_pool_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "lp_images_to_nhwc4": (_I, [_P, _I, _I, _I, _P, _P]),
    "lp_pixel_shuffle": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "lp_vit_patchify": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "lp_vit_tokens_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "lp_vit_tokens_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "lp_small_matmul": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "lp_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P]),
    "lp_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "lp_layernorm_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "lp_gelu_fwd": (_I, [_P, _Z, _P, _P]),
    "lp_gelu_bwd": (_I, [_P, _P, _Z, _P, _P]),
    "lp_gelu_bwd_colsum": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "lp_layernorm_bwd_bf16_colsum": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "lp_softmax_rows_fwd": (_I, [_P, _I, _I, _I, _F, _P]),
    "lp_softmax_rows_bwd": (_I, [_P, _P, _I, _I, _I, _F, _P]),
    "lp_transpose_batched": (_I, [_P, _I, _I, _I, C.c_longlong, C.c_longlong, _P, _I, C.c_longlong, C.c_longlong, _I, _I, _P]),
    "lp_frames_resize": (_I, [_P, _I, _I, _I, C.c_longlong, _I, _I, _I, _I, C.POINTER(FrameNorm), _P, _P]),
    "lp_frames_resize_cubic": (_I, [_P, _I, _I, _I, C.c_longlong, _I, _I, _I, _I, C.POINTER(FrameNorm), _P, _P]),
    "lp_frames_augment": (_I, [_P, _I, _I, _I, C.POINTER(FrameAugment), C.POINTER(FrameNorm), _P, _P]),
    "lp_labeled_keypoints": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "lp_f32_conv_fwd": (_I, [_P, _P, C.POINTER(ConvGeom), _I, _I, _I, _P, _P, _P, _P]),
    "lp_f32_conv_dgrad": (_I, [_P, _P, C.POINTER(ConvGeom), _I, _I, _I, _P, _P, _P, _P]),
    "lp_f32_conv_wgrad": (_I, [_P, _P, C.POINTER(ConvGeom), _I, _I, _I, _P, _P]),
    "lp_f32_bn_stats": (_I, [_P, _I, _I, _P, _P]),
    "lp_f32_bn_stats_workspace_bytes": (C.c_size_t, [_I, _I]),
    "lp_f32_bn_stats_ordered": (_I, [_P, _I, _I, _P, _P, C.c_size_t, _P]),
    "lp_f32_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "lp_f32_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "lp_f32_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P]),
    "lp_f32_maxpool_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "lp_f32_maxpool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "lp_f32_images_to_nhwc4": (_I, [_P, _I, _I, _I, _P, _P]),
    "lp_f32_pixel_shuffle": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "lp_f32_softmax2d_bwd": (_I, [_P, _P, _I, _I, _I, _P, _L, _L, _L, _P]),
    "lp_f32_vit_patchify": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "lp_f32_vit_tokens_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "lp_f32_vit_tokens_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "lp_f32_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P]),
    "lp_f32_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "lp_f32_gelu_fwd": (_I, [_P, _Z, _P, _P]),
    "lp_f32_gelu_bwd": (_I, [_P, _P, _Z, _P, _P]),
    "lp_f32_attn_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _I, _P]),
    "lp_f32_attn_bwd": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _I, _I, _F, _P, _P, _I, _P]),
    "lp_adam_step": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _I, _F, _P, _P]),
    "lp_cast_bf16": (_I, [_P, _Z, _P, _P]),
    "lp_permute_cba": (_I, [_P, _I, _I, _I, _P, _P]),
}


def declare(lib: C.CDLL) -> C.CDLL:
    """Attach the lp_hip.h prototypes to an opened library (every symbol must exist)."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    have = int(lib.lp_version())
    if have != ABI_VERSION:   # same symbols, other signatures: a stale build would take e.g. the stream for an inserted argument
        raise LpHipUnavailable(f"{getattr(lib, '_name', 'liblp_hip.so')} was built for ABI {have}, this package binds ABI {ABI_VERSION} "
                               "(include/lp_hip.h: LP_HIP_ABI_VERSION) - rebuild it: python -c 'import __graft_entry__ as g; g.build()'")
    return lib


_lib = None


def lib() -> C.CDLL:
    """The loaded liblp_hip.so; raises LpHipUnavailable (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LpHipUnavailable(
                f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(lightning-pose_amd/csrc/build.sh).  lightning_pose_amd has no CPU fallback.")
        try:
            _lib = declare(C.CDLL(LIB_PATH))
        except OSError as e:  # e.g. no ROCm runtime on this host
            raise LpHipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    return _lib


def check(code: int, what: str) -> None:
    """Map C-ABI return codes to the reference's exception conventions."""
    if code == 0:
        return
    msg = lib().lp_strerror(code).decode()
    if code == -1:
        raise ValueError(f"{what}: {msg}")
    if code == -2:
        raise NotImplementedError(f"{what}: {msg}")
    raise LpHipError(f"{what}: HIP error {code}: {msg}")
