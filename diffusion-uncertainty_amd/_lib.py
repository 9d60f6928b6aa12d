"""ctypes binding of libdua_hip.so (the C ABI declared in include/dua_hip.h).

The library is the product: if it is missing, or a tensor is not a contiguous fp32 CUDA(HIP) tensor,
the call raises - there is no eager/CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdua_hip.so")

_c = ctypes
_P, _I, _F, _Z, _L, _U64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t, _c.c_long, _c.c_uint64


class GemmArgs(_c.Structure):
    _fields_ = [("A1", _P), ("A2", _P), ("B", _P), ("bias", _P), ("residual", _P), ("C", _P),
                ("M", _I), ("N", _I), ("K1", _I), ("K2", _I),
                ("lda1", _L), ("lda2", _L), ("ldb", _L), ("ldc", _L), ("ldr", _L),
                ("sA1", _L), ("sA2", _L), ("sB", _L), ("sC", _L), ("sR", _L),
                ("batch", _I), ("batch_inner", _I),
                ("sA1i", _L), ("sBi", _L), ("sCi", _L),
                ("transB", _I), ("act_in", _I), ("act_out", _I), ("alpha", _F), ("colsum", _P)]


class ResBlockArgs(_c.Structure):
    _fields_ = [("x1", _P), ("C1", _I), ("x2", _P), ("C2", _I), ("N", _I), ("H", _I), ("W", _I), ("Cout", _I),
                ("gn1_w", _P), ("gn1_b", _P), ("conv1_packed", _P), ("conv1_b", _P), ("film", _P), ("film_ld", _L),
                ("gn2_w", _P), ("gn2_b", _P), ("conv2_packed", _P), ("conv2_b", _P), ("skip_packed", _P), ("skip_b", _P),
                ("x1_colsums", _P), ("x2_colsums", _P), ("updown", _I), ("dropmask", _P),
                ("precision", _I), ("act_scale", _F), ("raw_scale", _F), ("workspace", _P), ("workspace_bytes", _Z),
                ("y", _P), ("y_colsums", _P), ("conv1_wino", _P), ("conv2_wino", _P), ("conv1_wino_m", _I), ("conv2_wino_m", _I)]


class AttnBlockArgs(_c.Structure):
    _fields_ = [("x", _P), ("C", _I), ("N", _I), ("H", _I), ("W", _I), ("heads", _I), ("gn_w", _P), ("gn_b", _P),
                ("qkv_packed", _P), ("qkv_b", _P), ("proj_packed", _P), ("proj_b", _P), ("x_colsums", _P),
                ("precision", _I), ("act_scale", _F), ("raw_scale", _F), ("workspace", _P), ("workspace_bytes", _Z),
                ("y", _P), ("y_colsums", _P)]


class UViTBlockArgs(_c.Structure):
    _fields_ = [("x", _P), ("skip", _P), ("B", _I), ("L", _I), ("D", _I), ("heads", _I), ("hidden", _I),
                ("skip_packed", _P), ("skip_b", _P), ("norm1_w", _P), ("norm1_b", _P), ("norm2_w", _P), ("norm2_b", _P), ("eps", _F),
                ("qkv_packed", _P), ("qkv_b", _P), ("proj_packed", _P), ("proj_b", _P), ("fc1_packed", _P), ("fc1_b", _P),
                ("fc2_packed", _P), ("fc2_b", _P), ("precision", _I), ("act_scale", _F), ("raw_scale", _F),
                ("workspace", _P), ("workspace_bytes", _Z), ("y", _P), ("x_planes", _P), ("skip_planes", _P), ("y_planes", _P)]


# name -> (restype, argtypes); every symbol include/dua_hip.h declares
SIGNATURES = {
    "dua_version": (_I, []),
    "dua_last_error": (_c.c_char_p, []),
    "dua_ddim_step": (_I, [_P, _P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _P]),
    "dua_axpby": (_I, [_P, _P, _P, _Z, _F, _F, _P]),
    "dua_x0_from_eps": (_I, [_P, _P, _P, _Z, _F, _F, _P]),
    "dua_dpmpp_2m_update": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _P]),
    "dua_mc_centered_accum": (_I, [_P, _P, _P, _Z, _I, _I, _P]),
    "dua_mc_reduce_centered": (_I, [_P, _P, _P, _Z, _I, _P]),
    "dua_mc_reduce_var": (_I, [_P, _P, _Z, _I, _P]),
    "dua_quantile_mask": (_I, [_P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "dua_set_quantile_stream": (_I, [_I]),
    "dua_quantile_workspace_release": (_I, [_P, _I]),
    "dua_flip_h": (_I, [_P, _P, _Z, _I, _I, _P]),
    "dua_posterior_blend": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "dua_threshold_map": (_I, [_P, _P, _P, _I, _Z, _I, _I, _P]),
    "dua_batch_sum": (_I, [_P, _P, _I, _Z, _P]),
    "dua_posterior_score": (_I, [_P, _P, _P, _I, _Z, _I, _F, _P]),
    "dua_map_blend": (_I, [_P, _P, _P, _P, _Z, _P]),
    "dua_pixelwise_order_statistic": (_I, [_P, _I, _L, _I, _P, _P]),
    "dua_to_uint8": (_I, [_P, _P, _Z, _P]),
    "dua_randn": (_I, [_P, _Z, _U64, _U64, _P]),
    "dua_dropout_mask": (_I, [_P, _Z, _F, _U64, _U64, _P]),
    "dua_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dua_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dua_groupnorm_ws_floats": (_Z, [_I, _I, _I, _I]),
    "dua_groupnorm_stats": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "dua_groupnorm_apply": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dua_add_channel_bias": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "dua_geglu": (_I, [_P, _P, _L, _I, _P]),
    "dua_geglu_split_f16x3": (_I, [_P, _P, _L, _I, _F, _P]),
    "dua_pick_even": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dua_resample2x": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dua_conv3x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dua_groupnorm_stats_from_colsums": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _P, _P]),
    "dua_groupnorm_stats_from_colsums_bias": (_I, [_P, _I, _I, _I, _I, _F, _P, _L, _P, _P]),
    "dua_groupnorm_apply_prebias_split_f16x3": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _F, _P, _P]),
    "dua_conv3x3_packed_floats": (_Z, [_I, _I]),
    "dua_pack_conv3x3_weight": (_I, [_P, _P, _I, _I, _P]),
    "dua_gemm": (_I, [_c.POINTER(GemmArgs), _P]),
    "dua_softmax_rows": (_I, [_P, _L, _I, _L, _P]),
    "dua_range_check": (_I, [_I]),
    "dua_range_check_read": (_I, [_c.POINTER(_c.c_ulonglong), _c.POINTER(_c.c_ulonglong), _I]),
    "dua_set_splitk_workspace": (_I, [_P, _Z]),
    "dua_groupnorm_bwd": (_I, [_P, _I, _P, _I, _I, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "dua_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "dua_geglu_bwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "dua_softmax_bwd_rows": (_I, [_P, _P, _L, _I, _L, _P]),
    "dua_softmax_fwd_bwd_rows": (_I, [_P, _P, _L, _I, _L, _P]),
    "dua_place_even": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dua_mc_var_grad": (_I, [_P, _P, _I, _Z, _F, _P]),
    "dua_attention": (_I, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "dua_attention_bf16x3": (_I, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "dua_layernorm": (_I, [_P, _P, _P, _P, _L, _I, _F, _P]),
    "dua_timestep_embedding": (_I, [_P, _P, _I, _I, _P]),
    "dua_embedding_add": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dua_bf16x3_plane_elems": (_Z, [_I, _I]),
    "dua_pack_weight_bf16x3": (_I, [_P, _P, _I, _I, _I, _P]),
    "dua_conv3x3_bf16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dua_conv3x3_bf16x3_presplit": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dua_groupnorm_apply_split": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dua_f16x3_weight_bytes": (_Z, [_I, _I]),
    "dua_pack_weight_f16x3": (_I, [_P, _P, _I, _I, _I, _P]),
    "dua_groupnorm_apply_split_f16x3": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _F, _P, _P]),
    "dua_conv3x3_f16x3_presplit": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P]),
    "dua_gemm_f16x3_presplit": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P]),
    "dua_gemm_f16x3_presplit2": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _F, _F, _P]),
    "dua_set_hgemm_mode": (_I, [_I, _I]),
    "dua_set_hgemm_concurrent": (_I, [_I]),
    "dua_set_sgemm_mode": (_I, [_I]),
    "dua_conv3x3_tap_gather": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "dua_f16x3_wino_act_bytes": (_Z, [_I, _I, _I, _I]),
    "dua_f16x3_wino_weight_bytes": (_Z, [_I, _I]),
    "dua_pack_weight_f16x3_wino": (_I, [_P, _P, _I, _I, _P]),
    "dua_groupnorm_apply_wino_f16x3": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _F, _P, _P]),
    "dua_conv3x3_f16x3_wino_supported": (_I, [_I, _I, _I, _I, _I]),
    "dua_conv3x3_f16x3_wino": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _F, _P]),
    "dua_groupnorm_apply_wino_f16w": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _F, _P, _P]),
    "dua_conv3x3_f16w_wino": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _F, _P]),
    "dua_f16x3_wino4_act_bytes": (_Z, [_I, _I, _I, _I]),
    "dua_f16x3_wino4_weight_bytes": (_Z, [_I, _I]),
    "dua_pack_weight_f16x3_wino4": (_I, [_P, _P, _I, _I, _P]),
    "dua_groupnorm_apply_wino4_f16x3": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _F, _P, _P]),
    "dua_conv3x3_f16x3_wino4_supported": (_I, [_I, _I, _I, _I, _I]),
    "dua_conv3x3_f16x3_wino_pick": (_I, [_I, _I, _I, _I, _I]),
    "dua_conv3x3_f16x3_wino4": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _F, _P]),
    "dua_f32_wino_weight_bytes": (_Z, [_I, _I]),
    "dua_pack_weight_f32_wino": (_I, [_P, _P, _I, _I, _P]),
    "dua_groupnorm_apply_wino_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dua_conv3x3_f32_wino": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dua_f32_wino4_weight_bytes": (_Z, [_I, _I]),
    "dua_pack_weight_f32_wino4": (_I, [_P, _P, _I, _I, _P]),
    "dua_groupnorm_apply_wino4_f32": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dua_conv3x3_f32_wino4": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "dua_gemm_f16x3": (_I, [_P, _I, _L, _P, _I, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _F, _P, _F, _P]),
    "dua_attention_f16x3": (_I, [_P, _P, _P, _F, _I, _I, _I, _I, _F, _P]),
    "dua_cross_attention_f16x3": (_I, [_P, _L, _P, _P, _L, _P, _P, _F, _I, _I, _I, _I, _I, _F, _P]),
    "dua_adm_resblock_workspace_bytes": (_Z, [_c.POINTER(ResBlockArgs)]),
    "dua_adm_resblock_fwd": (_I, [_c.POINTER(ResBlockArgs), _P]),
    "dua_adm_attnblock_workspace_bytes": (_Z, [_c.POINTER(AttnBlockArgs)]),
    "dua_adm_attnblock_fwd": (_I, [_c.POINTER(AttnBlockArgs), _P]),
    "dua_gemm_f16x3_presplit_planes": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "dua_layernorm_split_f16x3": (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _P]),
    "dua_uvit_block_workspace_bytes": (_Z, [_c.POINTER(UViTBlockArgs)]),
    "dua_uvit_block_fwd": (_I, [_c.POINTER(UViTBlockArgs), _P]),
    "dua_conv_profile": (_I, [_I]),
    "dua_conv_profile_collect": (_I, [_c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_c.c_long), _c.POINTER(_c.c_double)]),
    "dua_kernel_profile": (_I, [_I]),
    "dua_kernel_profile_collect": (_c.c_long, [_c.c_char_p, _Z]),
    "dua_set_dma_conv": (_I, [_I]),
    "dua_set_gn_apply_wide": (_I, [_I]),
    "dua_set_conv_stagger": (_I, [_I, _I, _I]),
    "dua_gemm_bf16x3_presplit": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dua_layernorm_split": (_I, [_P, _P, _P, _P, _L, _I, _F, _P]),
    "dua_gemm_bf16x3": (_I, [_P, _I, _L, _P, _I, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _F, _P, _P]),
    "dua_gemm_nt_bf16x3_batched": (_I, [_P, _L, _L, _L, _P, _L, _L, _L, _P, _L, _L, _L, _I, _I, _I, _I, _I, _F, _P]),
    "dua_transpose_batched": (_I, [_P, _P, _I, _I, _L, _L, _L, _L, _L, _L, _I, _I, _P]),
    "dua_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dua_uvit_assemble": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dua_unpatchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().dua_last_error().decode()}")
