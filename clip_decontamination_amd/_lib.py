"""ctypes binding of libsegearth_hip.so (the C ABI declared in include/segearth_hip.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load, importing
any op raises immediately with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SEGEARTH_HIP_LIB: another build of the same library (same-box A/B measurements of a kernel change); the default is the in-tree build
LIB_PATH = os.environ.get("SEGEARTH_HIP_LIB") or os.path.join(HERE, "libsegearth_hip.so")

# enums (include/segearth_hip.h)
PREC_F32, PREC_BF16, PREC_FP8, PREC_F16, PREC_F16X2 = 0, 1, 2, 3, 4
MODEL_TYPES = {"vanilla": 0, "MaskCLIP": 1, "ClearCLIP": 2, "SCLIP": 3, "SegEarth": 4, "SFP": 5, "Experimental": 6,
               "NACLIP": 7, "NOnly": 8, "GAV": 9, "GEM": 10}
IMG_F32_NCHW, IMG_U8_NHWC = 0, 1


class VitDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "layers", "heads", "patch", "embed_dim", "grid0", "mlp_width",
                                          "quick_gelu", "precision")]


class ForwardOpts(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("ignore_residual", C.c_int32), ("similarity_enabled", C.c_int32),
                ("similarity_weight", C.c_float), ("similarity_temperature", C.c_float), ("similarity_add_self", C.c_int32),
                ("outlier_enabled", C.c_int32), ("outlier_top_k", C.c_int32), ("outlier_contamination_temp", C.c_float),
                ("selfattn_enabled", C.c_int32), ("selfattn_mode", C.c_int32), ("selfattn_top_k", C.c_int32),
                ("selfattn_strength", C.c_float), ("selfattn_threshold", C.c_float), ("gem_depth", C.c_int32),
                ("layer_fusion_enabled", C.c_int32), ("layer_fusion_lambda", C.c_float)]


class TileBatch(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("format", C.c_int32), ("scene_h", C.c_int32), ("scene_w", C.c_int32),
                ("windows", C.c_void_p), ("scene_index", C.c_void_p), ("scene_stride", C.c_int64), ("n_tiles", C.c_int32), ("tile_h", C.c_int32), ("tile_w", C.c_int32),
                ("pad_l", C.c_int32), ("pad_t", C.c_int32), ("grid_h", C.c_int32), ("grid_w", C.c_int32)]


P, I, F, Z, L = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "sg_last_error": (C.c_char_p, []),
    "sg_version": (I, []),
    "sg_profile_enable": (I, [I]),
    "sg_profile_disable": (I, []),
    "sg_set_gemm_config": (I, [I]),
    "sg_profile_read": (I, [I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(L), C.POINTER(L)]),
    "sg_create": (I, [C.POINTER(P), I, C.POINTER(VitDesc)]),
    "sg_destroy": (None, [P]),
    "sg_vit_set_tensor": (I, [P, C.c_char_p, P, L, P]),
    "sg_vit_finalize": (I, [P, P]),
    "sg_vit_workspace_bytes": (Z, [P, I, I, I, C.POINTER(ForwardOpts)]),
    "sg_vit_forward": (I, [P, C.POINTER(TileBatch), C.POINTER(ForwardOpts), P, P, P, Z, P]),
    "sg_cosine_logits": (I, [P, P, P, I, I, I, I, F, F, P, P]),
    "sg_cosine_logits_two_plane": (I, [P, P, P, I, I, I, I, F, P, P]),
    "sg_stitch": (I, [P, P, I, I, I, I, I, I, I, I, I, I, P, P]),
    "sg_resize_bilinear": (I, [P, I, I, I, P, I, I, P]),
    "sg_postprocess": (I, [P, P, I, I, I, I, F, F, I, P, P, P]),
    "sg_outlier_scratch_bytes": (Z, [I, I, I]),
    "sg_outlier_suppress": (I, [P, P, P, I, I, I, I, I, F, P, P, P]),
    "sg_cross_tile_scratch_bytes": (Z, [I, I, I, I, I]),
    "sg_cross_tile_fusion": (I, [P, I, I, I, I, I, I, I, F, P, P]),
    "sg_cross_tile_pack": (I, [P, P, I, I, I, I, I, I, I, I, P, P]),
    "sg_cross_tile_fuse": (I, [P, P, I, I, I, I, I, I, I, I, F, I, P, P]),
    "sg_cross_tile_apply": (I, [P, P, P, I, I, I, I, I, I, I, P]),
    "sg_weak_token_replace": (I, [P, P, I, I, I, I, I, P, P, P]),
    "sg_similarity_map": (I, [P, L, I, I, I, I, F, I, I, P, P, Z, P]),
    "sg_op_linear": (I, [P, P, P, P, P, I, I, I, I, I, P, Z, P]),
    "sg_gemm_bf16_raw": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "sg_gemm_fp8_raw": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "sg_op_ln_chain_scratch_bytes": (Z, [I, I, I, I]),
    "sg_op_ln_chain": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "sg_gemm_fp8_mx_raw": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "sg_quantize_rows_fp8": (I, [P, L, I, P, P, P]),
    "sg_op_layernorm": (I, [P, P, P, P, I, I, F, P]),
    "sg_op_attention_scratch_bytes": (Z, [I, I, I, I, I]),
    "sg_op_attention": (I, [P, I, I, I, I, I, P, F, P, P, P, I, P, Z, P]),
    "sg_adaptive_conv": (I, [P, P, I, I, I, I, I, P, P]),
    "sg_render_maps": (I, [P, P, P, I, I, I, P, P, P]),
    "sg_ctd_scratch_bytes": (Z, [I, I, I]),
    "sg_ctd_debias": (I, [P, P, I, I, I, C.c_double, I, F, I, P, P, Z, P]),
    "sg_text_create": (I, [C.POINTER(P), I, I, I, I, I, I, I, I, I]),
    "sg_text_destroy": (None, [P]),
    "sg_text_set_tensor": (I, [P, C.c_char_p, P, L, P]),
    "sg_text_workspace_bytes": (Z, [P, I]),
    "sg_text_encode": (I, [P, P, I, P, P, Z, P]),
    "sg_jbu_create": (I, [C.POINTER(P), I, I, I]),
    "sg_jbu_destroy": (None, [P]),
    "sg_jbu_set_tensor": (I, [P, C.c_char_p, P, L, P]),
    "sg_jbu_workspace_bytes": (Z, [P, I, I, I]),
    "sg_jbu_upsample": (I, [P, P, P, I, I, I, I, I, I, P, P, Z, P]),
    "sg_jbu_logits": (I, [P, P, P, I, I, I, I, I, I, P, I, P, F, P, P, Z, P]),
    "sg_extract_tiles": (I, [C.POINTER(TileBatch), I, I, P, P]),
    "sg_global_debias": (I, [P, P, I, I, I, F, P, P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or fail loudly -- there is no other implementation of the hot path."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch owns device memory and streams, so the library must bind to the SAME HIP runtime instance:
    # import torch first (its bundled libamdhip64 is then the one already loaded when ours resolves).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -m clip_decontamination_amd.build` (needs hipcc, cross-compiles for gfx950). "
            "There is no CPU fallback for the segmentation hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().sg_last_error().decode(errors="replace")
        raise RuntimeError(f"libsegearth_hip {what} failed ({rc}): {msg}")
