"""ctypes binding of libgaot_hip.so (include/gaot_hip.h).  Fails loudly: there is NO CPU or eager fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAOT_HIP_LIB: another build of the same ABI (same-box A/B runs of a kernel change: tools/ only)
LIB_PATH = os.environ.get("GAOT_HIP_LIB") or os.path.join(_HERE, "lib", "libgaot_hip.so")

ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD, ACT_SWIGLU_BWD, ACT_SWIGLU = 0, 1, 2, 3, 4, 5, 6

_f = C.c_void_p   # device float*
_i = C.c_void_p   # device int*
_s = C.c_void_p   # hipStream_t


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("A", _f), ("lda", C.c_int64), ("a_kmajor", C.c_int32),
        ("A2", _f), ("lda2", C.c_int64), ("k_split", C.c_int32),
        ("B", _f), ("ldb", C.c_int64), ("b_kmajor", C.c_int32),
        ("C", _f), ("ldc", C.c_int64),
        ("bias", _f),
        ("rowbias", _f), ("rowbias_period", C.c_int32), ("ld_rowbias", C.c_int64),
        ("rowscale", _f),
        ("act", C.c_int32),
        ("aux_in", _f), ("aux_out", _f), ("ld_aux", C.c_int64),
        ("residual", _f), ("ldr", C.c_int64),
        ("split_k", C.c_int32), ("workspace", _f),
        ("colsum", _f),
        ("pieces", C.c_int32),
        ("a_absmax", _f), ("b_absmax", _f),
        ("b_planes", C.c_void_p), ("ld_bplanes", C.c_int64), ("b_plane_stride", C.c_int64),
        ("c_absmax", _f),
        ("a2_absmax", _f),
        ("raw_slabs", C.c_int32),
    ]


class WgradItem(C.Structure):
    """gaot_wgrad_item: out[M,N] = g[K,M]^T x[K,N] (+ colsum[m] = sum_k g[k,m])"""
    _fields_ = [("g", _f), ("ldg", C.c_int64), ("x", _f), ("ldx", C.c_int64), ("out", _f), ("ldo", C.c_int64), ("colsum", _f),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("g_absmax", _f), ("x_absmax", _f)]


class AbsmaxItem(C.Structure):
    """gaot_absmax_item: out[0] = max(out[0], max |x[r * ld + c]|)"""
    _fields_ = [("x", _f), ("ld", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("out", _f)]


class F16PlanesItem(C.Structure):
    """gaot_f16_planes_item: weight matrix -> two fp16 planes of the scaled weight, as stored and transposed"""
    _fields_ = [("src", _f), ("ld", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("absmax", _f), ("planes_k", C.c_void_p), ("planes_t", C.c_void_p)]


class KmlpDesc(C.Structure):
    """gaot_kmlp_desc: one chain of gaot_kernel_mlp_fwd_pair / _bwd_pair"""
    _fields_ = [("x", _f), ("E", C.c_int32), ("cin", C.c_int32), ("n_layers", C.c_int32), ("w", C.POINTER(C.c_void_p)), ("b", C.POINTER(C.c_void_p)),
                ("act", C.c_int32), ("widths", C.POINTER(C.c_int32)), ("ldw", C.POINTER(C.c_int32)), ("pieces", C.c_int32), ("out", _f),
                ("dk", _f), ("grads", _f), ("workspace", _f)]


class UnionPart(C.Structure):
    """gaot_union_part: one sample of a block-diagonal union composed from a device-side table (gaot_union_compose)"""
    _fields_ = [("index", _i), ("edge_query", _i), ("t_edge", _i), ("splits", _i), ("t_splits", _i), ("src", _f), ("dst", _f),
                ("e_begin", C.c_int32), ("e_count", C.c_int32)]


class UnionPartRaw(C.Structure):
    """gaot_union_part_raw: one sample of a union composed straight from int64 CSR lists (gaot_union_compose_raw)"""
    _fields_ = [("index", _i), ("splits", _i), ("src", _f), ("dst", _f), ("e_begin", C.c_int32), ("e_count", C.c_int32), ("reserved", C.c_int64 * 3)]


class ColsumItem(C.Structure):
    """gaot_colsum_item: out[n] = sum_m x[m * ld + n]"""
    _fields_ = [("x", _f), ("ld", C.c_int64), ("out", _f), ("M", C.c_int32), ("N", C.c_int32), ("out_cols", C.c_int32), ("out_ld", C.c_int64)]


# name -> (restype, argtypes): every symbol include/gaot_hip.h (data path) and include/gaot_hip_debug.h (tuning hooks) declare
PROTOTYPES = {
    "gaot_abi_version": (C.c_int, []),
    "gaot_last_error": (C.c_char_p, []),
    "gaot_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _s]),
    "gaot_gemm_tn_grouped_workspace": (C.c_int64, [C.POINTER(WgradItem), C.c_int32, C.POINTER(C.c_int32)]),
    "gaot_gemm_tn_grouped": (C.c_int, [C.POINTER(WgradItem), C.c_int32, C.c_int32, _f, _i, _s]),
    "gaot_debug_set_gemm_tile": (C.c_int, [C.c_int]),
    "gaot_debug_set_gemm_ablate": (C.c_int, [C.c_int]),
    "gaot_debug_last_gemm_path": (C.c_int, []),
    "gaot_debug_set_gemm_glds": (C.c_int, [C.c_int]),
    "gaot_debug_set_gemm_pieces": (C.c_int, [C.c_int]),
    "gaot_gemm_path": (C.c_int, [C.POINTER(GemmDesc)]),
    "gaot_gemm_slab_count": (C.c_int32, [C.c_int32, C.c_int32]),
    "gaot_absmax_grouped": (C.c_int, [C.POINTER(AbsmaxItem), C.c_int32, _s]),
    "gaot_split_f16_planes_grouped": (C.c_int, [C.POINTER(F16PlanesItem), C.c_int32, _s]),
    "gaot_debug_set_gemm_planes": (C.c_int, [C.c_int]),
    "gaot_debug_set_gemm_ad": (C.c_int, [C.c_int]),
    "gaot_debug_set_gemm_ad_narrow": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_qsplit": (C.c_int, [C.c_int]),
    "gaot_debug_split_redo_count": (C.c_uint, [C.c_int]),
    "gaot_debug_set_wgrad_kslab": (C.c_int, [C.c_int]),
    "gaot_debug_set_wgrad_tile_rows": (C.c_int, [C.c_int]),
    "gaot_debug_set_gemm_ad_flush": (C.c_int, [C.c_int]),
    "gaot_debug_set_wgrad_slab_rule": (C.c_int, [C.c_int]),
    "gaot_csr_prepare": (C.c_int, [_i, _i, C.c_int32, C.c_int32, C.c_int32, _i, _i, _i, _i, _s]),
    "gaot_csr_transpose": (C.c_int, [_i, C.c_int32, C.c_int32, _i, _i, _i, _s]),
    "gaot_union_compose": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, _i, _i, _i, _f, _f, _i, _s]),
    "gaot_union_compose_raw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, _i, _f, _f, _i, _i, _s]),
    "gaot_csr_transpose_dev_scratch": (C.c_int64, [C.c_int32, C.c_int32]),
    "gaot_csr_transpose_dev": (C.c_int, [_i, C.c_int32, _i, C.c_int32, _i, _i, _i, _s]),
    "gaot_edge_drop_scratch": (C.c_int64, [C.c_int32]),
    "gaot_edge_drop": (C.c_int, [_i, _i, _i, _i, _i, C.c_int32, C.c_int32, C.c_int32, _i, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                 _i, _i, _i, _i, _i, _i, _i, _s]),
    "gaot_edge_inv_degree": (C.c_int, [_i, _i, C.c_int32, _i, _f, _s]),
    "gaot_edge_zero_pads": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, _i, C.c_int32, _s]),
    "gaot_guard_begin": (C.c_int, [_i, _s]),
    "gaot_guard_compare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _i, _s]),
    "gaot_guard_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _i, _s]),
    "gaot_guard_sync2": (C.c_int, [_f, _f, C.c_int64, _f, _f, C.c_int64, _i, _s]),
    "gaot_edge_attention_cosine": (C.c_int, [_f, _f, C.c_int32, _i, _i, C.c_int32, _f, _i, _s]),
    "gaot_segment_softmax_fwd": (C.c_int, [_f, _i, C.c_int32, _f, _s]),
    "gaot_segment_softmax_bwd": (C.c_int, [_f, _f, _i, C.c_int32, _f, _s]),
    "gaot_edge_features": (C.c_int, [_f, _f, C.c_int32, _i, _i, C.c_int32, _f, _i, _s]),
    "gaot_geo_stats": (C.c_int, [_f, _f, C.c_int32, _i, _i, C.c_int32, _f, C.c_void_p, _i, C.c_int32, _s]),
    "gaot_concat_offset": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _i, _s]),
    "gaot_cells_build": (C.c_int, [_f, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_int32), _i, _i, _i, _s]),
    "gaot_radius_count": (C.c_int, [_f, C.c_int32, _f, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_int32),
                                    _i, _i, _i, _i, C.c_int32, C.c_int32, _s]),
    "gaot_radius_fill": (C.c_int, [_f, C.c_int32, _f, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_int32),
                                   _i, _i, _i, _i, _i, C.c_int32, C.c_int32, _s]),
    "gaot_gno_gather_reduce": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_int32, _i, _i, _i, C.c_int32, _f, _f, _s]),
    "gaot_gno_edge_grad": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, C.c_int32, _f, _f, _s]),
    "gaot_gno_segment_sum": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, _i, C.c_int32, _f, _f, _s]),
    "gaot_rmsnorm_fwd": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_float, _f, _f, _f, _s]),
    "gaot_rmsnorm_bwd_partials": (C.c_int, [C.c_int32]),
    "gaot_rmsnorm_bwd": (C.c_int, [_f, _f, _f, _f, _f, _f, C.c_int32, C.c_int32, _f, _f, _f, _s]),
    "gaot_rmsnorm_bwd_slabs": (C.c_int, [_f, _f, _f, _f, C.c_int32, C.c_int64, _f, _f, _f, C.c_int32, C.c_int32, _f, _f, _f, _s]),
    "gaot_swiglu_fwd": (C.c_int, [_f, C.c_int32, C.c_int32, _f, _s]),
    "gaot_swiglu_bwd": (C.c_int, [_f, _f, C.c_int32, C.c_int32, _f, _s]),
    "gaot_act_bwd": (C.c_int, [_f, _f, C.c_int64, C.c_int32, _f, _s]),
    "gaot_attention_fwd": (C.c_int, [_f, _f, _f, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, _f, C.c_int64, _f, C.c_int32, _f, _s]),
    "gaot_attention_fwd_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gaot_attention_fwd_ws": (C.c_int, [_f, _f, _f, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, _f, C.c_int64, _f, C.c_int32, _f, _f, _s]),
    "gaot_attention_bwd_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gaot_attention_bwd": (C.c_int, [_f, _f, _f, C.c_int64, C.c_int64, C.c_int64, _f, _f, C.c_int64, _f,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     _f, _f, _f, C.c_int64, C.c_int64, C.c_int64, _f, C.c_int32, _f, _f, _f, _s]),
    "gaot_attention_seed_next": (C.c_int, [_i, C.c_uint64, _i, _s]),
    "gaot_attention_fwd_dropout": (C.c_int, [_f, _f, _f, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, _f, C.c_int64, _f, C.c_float, _i, _s]),
    "gaot_attention_bwd_dropout": (C.c_int, [_f, _f, _f, C.c_int64, C.c_int64, C.c_int64, _f, _f, C.c_int64, _f,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             _f, _f, _f, C.c_int64, C.c_int64, C.c_int64, _f, C.c_float, _i, _s]),
    "gaot_colsum_scratch": (C.c_int64, [C.c_int32, C.c_int32]),
    "gaot_colsum": (C.c_int, [_f, C.c_int64, C.c_int32, C.c_int32, _f, _f, _s]),
    "gaot_colsum_grouped": (C.c_int, [C.POINTER(ColsumItem), C.c_int32, _s]),
    "gaot_batchsum": (C.c_int, [_f, C.c_int32, C.c_int64, _f, _s]),
    "gaot_gno_lift_gather_reduce": (C.c_int, [_f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, _f, C.c_int32, _f, _f, _f, _s]),
    "gaot_gno_lift_edge_grad_parts": (C.c_int32, [C.c_int32, C.c_int32]),
    "gaot_gno_lift_edge_grad": (C.c_int, [_f, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, _f, C.c_int32, _f, _f, _f, _s]),
    "gaot_gno_proj_gather_reduce": (C.c_int, [_f, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, _f, C.c_int32, _f, _f, _s]),
    "gaot_gno_proj_backward": (C.c_int, [_f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, _f, C.c_int32, _f, _f, _f, _f, _f, _f, _s]),
    "gaot_kernel_mlp_fwd": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, _f, _s]),
    "gaot_debug_set_kernel_mlp_ablate": (C.c_int, [C.c_int]),
    "gaot_debug_set_kernel_mlp_split": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_split": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_h16": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_dh8": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_keysplit": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_pipe": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_p_pieces": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_operand_pieces": (C.c_int, [C.c_int]),
    "gaot_debug_set_attention_tr": (C.c_int, [C.c_int]),
    "gaot_kernel_mlp_fwd_w": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.c_int32, _f, _s]),
    "gaot_kernel_mlp_bwd_w": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.c_int32, _f, _f, _f, _s]),
    "gaot_kernel_mlp_fwd_pair": (C.c_int, [C.POINTER(KmlpDesc), C.POINTER(KmlpDesc), _s]),
    "gaot_kernel_mlp_bwd_pair": (C.c_int, [C.POINTER(KmlpDesc), C.POINTER(KmlpDesc), _s]),
    "gaot_kernel_mlp_bwd_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gaot_kernel_mlp_bwd_rows": (C.c_int32, [C.c_int32]),
    "gaot_kernel_mlp_bwd": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, _f, _f, _f, _s]),
    "gaot_mse_loss_fwd": (C.c_int, [_f, _f, C.c_int64, _f, _f, _s]),
    "gaot_mse_loss_bwd": (C.c_int, [_f, _f, C.c_int64, _f, _f, _s]),
    "gaot_mse_loss_fwd_bwd": (C.c_int, [_f, _f, C.c_int64, _f, C.c_void_p, _f, _f, _f, _s]),
    "gaot_adamw_step": (C.c_int, [_f, _f, _f, _f, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _f, _s]),
    "gaot_adamw_step_dev": (C.c_int, [_f, _f, _f, _f, C.c_int64, _f, _f, _s]),
    "gaot_adamw_apply_dev": (C.c_int, [_f, _f, _f, _f, C.c_int64, _f, _f, _s]),
    "gaot_debug_set_ep_chunk": (C.c_int, [C.c_int]),
    "gaot_gno_ep_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "gaot_gno_lift_gather_reduce_ep": (C.c_int, [_f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, _i, C.c_int32, C.c_int32, _f, _f, _f, _i, _s]),
    "gaot_gno_proj_gather_t_ep": (C.c_int, [_f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, C.c_int32, _i, _i, _f, _f, _f, _i, _s]),
    "gaot_gno_proj_gather_t_ep_w": (C.c_int, [_f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, C.c_int32, _i, _i, _f, _f, _f, _i, _f, _s]),
    "gaot_proj_fold_workspace": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "gaot_proj_fold_fwd": (C.c_int, [_f, C.c_int64, _f, _f, C.c_int64, _f, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, _f, _s]),
    "gaot_proj_fold_bwd": (C.c_int, [_f, _f, _f, C.c_int64, _f, C.c_int64, _f, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f,
                                     _f, C.c_int64, _f, _f, C.c_int64, _f, C.c_void_p, _s]),
    "gaot_gno_proj_gather_reduce_bin": (C.c_int, [_f, _f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, C.c_int32, _f, _f, _i, _s]),
    "gaot_rope_inplace": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _f, C.c_int32, _s]),
    "gaot_edge_dot_score": (C.c_int, [_f, _f, C.c_int32, _i, _i, C.c_int32, C.c_float, _f, _s]),
    "gaot_edge_rowdot_scale": (C.c_int, [_f, _f, _f, C.c_int32, C.c_int32, _f, _s]),
    "gaot_segment_max_fwd": (C.c_int, [_f, C.c_int32, _i, C.c_int32, _f, _s]),
    "gaot_segment_max_bwd": (C.c_int, [_f, _f, _f, C.c_int32, _i, C.c_int32, _f, _s]),
    "gaot_segment_broadcast": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _f, _f, _s]),
    "gaot_scale_mix_fwd": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, _f, C.c_int32, C.c_int32, C.c_int32, _f, _s]),
    "gaot_scale_mix_bwd": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _s]),
    "gaot_edge_cat": (C.c_int, [_f, C.c_int32, _f, C.c_int32, C.c_int32, C.c_int32, _i, C.c_int32, _f, _s]),
    "gaot_gno_bk_reduce": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, C.c_int32, _f, C.c_int32, _f, _s]),
    "gaot_gno_bk_backward": (C.c_int, [_f, _f, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i, _i, _i, _i, _f,
                                       C.c_int32, _f, _f, _f, _s]),
    "gaot_cond_affine_fwd": (C.c_int, [_f, _f, _f, C.c_int32, C.c_int64, C.c_int32, _f, _s]),
    "gaot_cond_affine_bwd_chunks": (C.c_int32, [C.c_int64]),
    "gaot_cond_affine_bwd": (C.c_int, [_f, _f, _f, C.c_int32, C.c_int64, C.c_int32, _f, _f, _s]),
    "gaot_rollout_input": (C.c_int, [_f, C.c_int32, _f, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int64, _f, _s]),
    "gaot_rollout_update": (C.c_int, [_f, _f, C.c_int32, _f, _f, _f, _f, C.c_float, C.c_int32, C.c_int64, _f, _s]),
    "gaot_patchify": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f, C.c_int32, _f, _s]),
}

_lib = None


class GaotLibraryError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and bind every prototype.  Raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must load ITS HIP runtime (libamdhip64) first: libgaot_hip.so then binds to the same runtime instance, so
    # streams and device pointers are interchangeable.  Loading us first would pull in a second runtime by path.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GaotLibraryError(
            f"{LIB_PATH} not found: build it with `python -m gaot_amd.build` (hipcc --offload-arch=gfx950). "
            "gaot_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().gaot_last_error()
        raise GaotLibraryError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
