"""ctypes binding of libkmunet_hip.so (the C ABI declared in include/kmunet_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError
is raised.  The product never computes the hot path on the CPU or through PyTorch ops.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkmunet_hip.so")
if os.environ.get("KMU_LIB_VARIANT"):      # development: an alternative build made by tools/build_variant.py (never a fallback)
    LIB_PATH = os.path.join(_HERE, "lib", "variants", os.environ["KMU_LIB_VARIANT"], "libkmunet_hip.so")

_c = ctypes
_P, _I, _Z = _c.c_void_p, _c.c_int, _c.c_size_t

class DagemArgs(_c.Structure):
    """struct kmu_dagem_args of include/kmunet_hip.h (field for field)."""
    _fields_ = ([(n, _I) for n in ("B", "C", "H", "W", "training")] + [("eps", _c.c_float * 5), ("momentum", _c.c_float * 5)] +
                [(n, _P) for n in ("x", "dconv", "wa", "ba", "wv", "bv", "we", "be", "wr", "br", "wf")] +
                [(n, _P * 5) for n in ("gamma", "beta", "running_mean", "running_var", "num_batches_tracked")] +
                [(n, _P) for n in ("a_pre", "u_pre", "v_pre", "r_pre", "z", "out", "bnstat", "part", "part_bwd", "agg_out", "u_out", "vert_out",
                                   "ue_out", "g_out", "g_dd", "gv", "gr", "ga", "dr_pre", "dxb", "de", "dx")] +
                [(n, _P * 5) for n in ("d_gamma", "d_beta")] +
                [(n, _P) for n in ("p_wf", "p_wv", "p_bv", "p_we", "p_be", "p_wa", "p_wr")])


# name -> (restype, argtypes); must list every symbol of include/kmunet_hip.h
SIGNATURES = {
    "kmu_version": (_I, []),
    "kmu_last_error": (_c.c_char_p, []),
    "kmu_kan_pack_fwd_elems": (_Z, [_I, _I]),
    "kmu_kan_pack_bwd_elems": (_Z, [_I, _I]),
    "kmu_kan_pack_weights": (_I, [_P] * 5 + [_I, _I, _P]),
    "kmu_kan_conv2d_fwd": (_I, [_P] * 5 + [_I] * 6 + [_P]),
    "kmu_kan_bwd_ws_bytes": (_Z, [_I] * 5),
    "kmu_kan_conv2d_bwd_input": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "kmu_kan_conv2d_bwd_weights": (_I, [_P] * 9 + [_Z] + [_I] * 5 + [_P]),
    "kmu_layernorm1d_fwd": (_I, [_P] * 5 + [_I] * 3 + [_c.c_float, _P]),
    "kmu_layernorm1d_bwd": (_I, [_P] * 7 + [_I] * 3 + [_P]),
    "kmu_layernorm1d_partials": (_I, [_I, _I, _I]),
    "kmu_hsmssd_state_elems": (_Z, [_I] * 3),
    "kmu_hsmssd_fwd_ws_bytes": (_Z, [_I] * 4),
    "kmu_hsmssd_fwd": (_I, [_P] * 10 + [_Z] + [_I] * 4 + [_P]),
    "kmu_hsmssd_fwd_stage": (_I, [_P] * 10 + [_Z] + [_I] * 5 + [_P]),
    "kmu_hsmssd_fwd_stage_x3": (_I, [_P] * 10 + [_Z] + [_I] * 5 + [_P]),
    "kmu_hsmssd_bwd_stage": (_I, [_P] * 16 + [_Z] + [_I] * 5 + [_P]),
    "kmu_hsmssd_bwd_ws_bytes": (_Z, [_I] * 4),
    "kmu_hsmssd_bwd_ws_bytes_x3": (_Z, [_I] * 4),
    "kmu_hsmssd_bwd_partials_x3": (_I, [_I] * 3),
    "kmu_hsmssd_bwd_stage_x3": (_I, [_P] * 16 + [_Z] + [_I] * 5 + [_P]),
    "kmu_hsmssd_bwd_partials": (_I, [_I] * 3),
    "kmu_hsmssd_gate_partials": (_I, [_I]),
    "kmu_hsmssd_bwd": (_I, [_P] * 16 + [_Z] + [_I] * 4 + [_P]),
    "kmu_dysample_lp_fwd": (_I, [_P] * 6 + [_I] * 4 + [_P]),
    "kmu_dysample_lp_bwd": (_I, [_P] * 6 + [_I] * 4 + [_P]),
    "kmu_deform_conv2d_fwd": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "kmu_deform_conv2d_bwd": (_I, [_P] * 8 + [_I] * 5 + [_P]),
    "kmu_deform_sample_fwd": (_I, [_P] * 3 + [_I] * 4 + [_P]),
    "kmu_deform_sample_bwd": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_deform_sample_bwd_lds_supported": (_I, [_I] * 4),
    "kmu_deform_sample_bwd_lds": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_bn_blend_splits": (_I, [_I, _I]),
    "kmu_bn_blend_fwd": (_I, [_P] * 7 + [_c.c_float, _c.c_float, _I, _I] + [_P] * 4 + [_I] * 3 + [_P]),
    "kmu_bn_blend_bwd": (_I, [_P] * 7 + [_I, _I] + [_P] * 6 + [_I] * 3 + [_P]),
    "kmu_qkv_gate_fwd": (_I, [_P] * 2 + [_I] * 3 + [_P]),
    "kmu_qkv_gate_bwd": (_I, [_P] * 3 + [_I] * 3 + [_P]),
    "kmu_pwconv_fwd": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "kmu_pwconv_bwd_input": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "kmu_pwconv_bwd_weight_ws_bytes": (_Z, [_I] * 4),
    "kmu_pwconv_bwd_weight": (_I, [_P] * 5 + [_Z] + [_I] * 5 + [_P]),
    "kmu_dwconv3x3_scaled_fwd": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_scaled_bwd_data": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_scaled_finish": (_I, [_P] * 8 + [_I] * 2 + [_P]),
    "kmu_dwconv3x3_scaled_bwd_all": (_I, [_P] * 7 + [_I] * 4 + [_P]),
    "kmu_qkv_dw_scaled_supported": (_I, [_I] * 4),
    "kmu_qkv_dw_scaled_fwd": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_qkv_dw_scaled_bwd": (_I, [_P] * 7 + [_I] * 4 + [_P]),
    "kmu_gauss11_filter": (_I, [_P] * 3 + [_I] * 4 + [_P]),
    "kmu_mix3_blocks": (_I, [_I]),
    "kmu_mix3_fwd": (_I, [_P] * 7 + [_I] * 2 + [_P]),
    "kmu_mix3_bwd": (_I, [_P] * 10 + [_I] * 2 + [_P]),
    "kmu_mean_rows": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "kmu_mix3_bwd_dg": (_I, [_P] * 7 + [_I] * 2 + [_P]),
    "kmu_mix3_bwd_apply": (_I, [_P] * 7 + [_I] * 3 + [_P]),
    "kmu_shift3_fwd": (_I, [_P] * 2 + [_I] * 5 + [_P]),
    "kmu_shift3_bwd": (_I, [_P] * 2 + [_I] * 5 + [_P]),
    "kmu_hybrid_loss_blocks": (_I, [_I] * 3),
    "kmu_hybrid_loss_stats": (_I, [_P] * 4 + [_I] * 3 + [_P]),
    "kmu_hybrid_loss_stack": (_I, [_P] * 4 + [_I] * 3 + [_P]),
    "kmu_hybrid_loss_combine": (_I, [_P] * 3 + [_I] * 3 + [_c.c_float, _P]),
    "kmu_hybrid_loss_grad_maps": (_I, [_P] * 3 + [_I] * 3 + [_c.c_float, _P]),
    "kmu_hybrid_loss_grad_input": (_I, [_P] * 6 + [_I] * 3 + [_c.c_float, _P]),
    "kmu_iwp_front_fwd": (_I, [_P] * 2 + [_I] * 5 + [_P]),
    "kmu_iwp_front_bwd": (_I, [_P] * 2 + [_I] * 5 + [_P]),
    "kmu_gate_mlp_fwd": (_I, [_P] * 7 + [_I] * 6 + [_P]),
    "kmu_gate_mlp_bwd": (_I, [_P] * 11 + [_I] * 6 + [_P]),
    "kmu_colsum_multi": (_I, [_I, _P, _P, _P, _P, _P]),
    "kmu_colsum_multi_strided": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "kmu_copy_multi": (_I, [_I, _P, _P, _P, _P]),
    "kmu_add_n": (_I, [_P] * 5 + [_c.c_longlong, _P]),
    "kmu_relu_mask": (_I, [_P] * 3 + [_c.c_longlong, _P]),
    "kmu_bias_sum_multi": (_I, [_I] + [_P] * 5 + [_P]),
    "kmu_group_norm_splits": (_I, [_I]),
    "kmu_group_norm_fwd": (_I, [_P] * 6 + [_I] * 4 + [_c.c_float, _P]),
    "kmu_group_norm_bwd": (_I, [_P] * 8 + [_I] * 4 + [_P]),
    "kmu_group_norm_act_fwd": (_I, [_P] * 6 + [_I] * 4 + [_c.c_float, _I, _P]),
    "kmu_group_norm_act_bwd": (_I, [_P] * 9 + [_I] * 5 + [_P]),
    "kmu_dwconv3x3_fwd": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_bwd_data": (_I, [_P] * 3 + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_bwd_data_add": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "kmu_pwconv_bwd_input_add": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "kmu_pwconv_bwd_input_rowadd": (_I, [_P] * 3 + [_c.c_float, _P] + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_partials": (_I, [_I]),
    "kmu_dwconv3x3_bwd_weight": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "kmu_conv3x3_x3_pack_elems": (_Z, [_I, _I, _I]),
    "kmu_kan_pack_weights_x3": (_I, [_P] * 4 + [_I, _I, _P]),
    "kmu_conv3x3_pack_weights_x3": (_I, [_P] * 2 + [_I, _I, _P]),
    "kmu_conv3x3_pack_weights_dgrad_x3": (_I, [_P] * 2 + [_I, _I, _P]),
    "kmu_kan_conv2d_fwd_x3": (_I, [_P] * 5 + [_I] * 6 + [_P]),
    "kmu_conv3x3_fwd_x3": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "kmu_kan_dgrad_x3_pack_elems": (_Z, [_I, _I]),
    "kmu_kan_pack_weights_dgrad_x3": (_I, [_P] * 4 + [_I, _I, _P]),
    "kmu_kan_conv2d_bwd_input_x3": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "kmu_conv3x3_x3_wgrad_ws_bytes": (_Z, [_I] * 6),
    "kmu_kan_conv2d_bwd_weights_x3": (_I, [_P] * 9 + [_Z] + [_I] * 5 + [_P]),
    "kmu_conv3x3_bwd_weight_x3": (_I, [_P] * 4 + [_Z] + [_I] * 5 + [_P]),
    "kmu_conv2d_x3_pack_elems": (_Z, [_I, _I, _I]),
    "kmu_conv2d_pack_weights_x3": (_I, [_P] * 2 + [_I] * 4 + [_P]),
    "kmu_conv2d_fwd_x3": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "kmu_conv2d_x3_wgrad_ws_bytes": (_Z, [_I] * 6),
    "kmu_conv2d_bwd_weight_x3": (_I, [_P] * 4 + [_Z] + [_I] * 6 + [_P]),
    "kmu_resize_bilinear_ac_fwd": (_I, [_P] * 2 + [_I] * 6 + [_P]),
    "kmu_resize_bilinear_ac_bwd": (_I, [_P] * 2 + [_I] * 6 + [_P]),
    # grouped variants: the three direction branches of EnhancedViMBlock stacked along the channel axis, one launch per layer
    "kmu_pwconv_fwd_g": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "kmu_pwconv_bwd_input_g": (_I, [_P] * 5 + [_I] * 6 + [_P]),
    "kmu_pwconv_bwd_weight_g": (_I, [_P] * 5 + [_Z] + [_I] * 7 + [_P]),
    "kmu_layernorm1d_fwd_g": (_I, [_P] * 5 + [_I] * 3 + [_c.c_float, _I, _P]),
    "kmu_layernorm1d_bwd_g": (_I, [_P] * 7 + [_I] * 4 + [_P]),
    "kmu_layernorm1d_bwd_add": (_I, [_P] * 8 + [_I] * 4 + [_P]),
    "kmu_hsmssd_fwd_ws_bytes_g": (_Z, [_I] * 5),
    "kmu_hsmssd_fwd_stage_x3_g": (_I, [_P] * 10 + [_Z] + [_I] * 6 + [_P]),
    "kmu_hsmssd_bwd_ws_bytes_x3_g": (_Z, [_I] * 5),
    "kmu_hsmssd_bwd_stage_x3_g": (_I, [_P] * 16 + [_Z] + [_I] * 6 + [_P]),
    "kmu_pack_job_bytes": (_Z, []),
    "kmu_conv_pack_job": (_I, [_P, _I, _I] + [_P] * 4 + [_I] * 3),
    "kmu_conv_pack_multi": (_I, [_P, _I, _P]),
    "kmu_hsmssd_pack_elems": (_Z, [_I, _I]),
    "kmu_hsmssd_pack_x3": (_I, [_P] * 3 + [_I, _I, _P]),
    "kmu_hsm_pack_job": (_I, [_P, _I] + [_P] * 3 + [_I, _I]),
    "kmu_hsm_pack_multi": (_I, [_P, _I, _P]),
    "kmu_hsmssd_fwd_stage_x3_pk": (_I, [_P] * 10 + [_Z] + [_I] * 6 + [_P, _P]),
    "kmu_hsmssd_bwd_stage_x3_pk": (_I, [_P] * 16 + [_Z] + [_I] * 6 + [_P, _P]),
    "kmu_mixer_fwd_ws_bytes": (_Z, [_I] * 4),
    "kmu_mixer_fwd_stage": (_I, [_P] * 3 + [_c.c_float] + [_P] * 11 + [_Z, _P] + [_I] * 6 + [_P]),
    "kmu_mixer_bwd_ws_bytes": (_Z, [_I] * 4),
    "kmu_mixer_bwd_partials": (_I, [_I] * 2),
    "kmu_mixer_bwd_stage": (_I, [_P] * 18 + [_Z] + [_I] * 6 + [_P, _P]),
    "kmu_mixer_debug_passb": (None, [_I]),
    "kmu_mixer_debug_rows": (None, [_I]),
    "kmu_conv_debug_split": (None, [_I]),
    "kmu_gate_mlp_fwd_g": (_I, [_P] * 7 + [_I] * 7 + [_P]),
    "kmu_gate_mlp_bwd_g": (_I, [_P] * 11 + [_I] * 7 + [_P]),
    "kmu_mix3_fwd_stacked": (_I, [_P] * 5 + [_I] * 2 + [_P]),
    "kmu_mix3_bwd_dg_stacked": (_I, [_P] * 5 + [_I] * 2 + [_P]),
    "kmu_mix3_bwd_apply_stacked": (_I, [_P] * 5 + [_I] * 3 + [_P]),
    "kmu_pwconv_bwd_weight_partial": (_I, [_P] * 3 + [_Z] + [_I] * 6 + [_P]),
    "kmu_pwconv_bwd_weight_partial_multi": (_I, [_I] + [_P] * 3 + [_Z] + [_I] * 6 + [_P]),
    "kmu_pwconv_bwd_weight_reduce_multi": (_I, [_I] + [_P] * 7 + [_P]),
    "kmu_dwconv3x3_stats_partials": (_I, [_I] * 4),
    "kmu_dwconv3x3_fwd_stats": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_pwconv_stats_partials": (_I, [_I] * 2),
    "kmu_pwconv_fwd_stats": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "kmu_bn_blend_fwd_pre": (_I, [_P] * 7 + [_c.c_float, _c.c_float, _I] + [_P] * 3 + [_I, _P] + [_I] * 3 + [_P]),
    "kmu_lca_fwd": (_I, [_P] * 3 + [_I] * 3 + [_P]),
    "kmu_lca_bwd": (_I, [_P] * 5 + [_I] * 3 + [_P]),
    "kmu_dagem_edges_fwd": (_I, [_P] * 2 + [_I] * 4 + [_P]),
    "kmu_dagem_edges_bwd": (_I, [_P] * 3 + [_I] * 4 + [_P]),
    "kmu_dagem_args_bytes": (_Z, []),
    "kmu_dagem_supported": (_I, [_I]),
    "kmu_dagem_tiles": (_I, [_I] * 3),
    "kmu_dagem_part_floats": (_Z, [_I] * 4),
    "kmu_dagem_stage": (_I, [_P, _I, _P]),
    "kmu_triple_norm_supported": (_I, [_I, _I]),
    "kmu_triple_norm_splits": (_I, [_I]),
    "kmu_triple_norm_partials": (_I, [_I, _I, _I]),
    "kmu_triple_norm_fwd": (_I, [_P] * 10 + [_I] * 3 + [_c.c_float, _c.c_float, _P]),
    "kmu_triple_norm_bwd": (_I, [_P] * 12 + [_I] * 3 + [_c.c_float, _P]),
    "kmu_pwconv_fwd_res": (_I, [_P] * 6 + [_I] * 5 + [_P]),
    "kmu_pwconv_bwd_input_s": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "kmu_ffn_fused_supported": (_I, [_I] * 3),
    "kmu_ffn_fused_rows": (_I, [_I] * 4),
    "kmu_ffn_fused_fwd_ws_bytes": (_Z, [_I] * 3),
    "kmu_ffn_fused_bwd_ws_bytes": (_Z, [_I] * 3),
    "kmu_ffn_fused_fwd": (_I, [_P] * 7 + [_c.c_float] * 2 + [_P] * 6 + [_c.c_float] * 2 + [_P, _I] + [_P] * 6 + [_Z] + [_I] * 4 + [_P]),
    "kmu_ffn_fused_bwd": (_I, [_P] * 12 + [_I] + [_P] * 9 + [_Z] + [_I] * 4 + [_P]),
    "kmu_bn_blend_bwd_partials": (_I, [_P] * 7 + [_I, _P] + [_I] * 3 + [_P]),
    "kmu_dwconv3x3_bn_bwd_data": (_I, [_P] * 7 + [_I, _I] + [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_bn_bwd_weight": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_dwconv3x3_bn_bwd_all": (_I, [_P] * 8 + [_I, _I] + [_P] * 5 + [_I] * 4 + [_P]),
    "kmu_tail_ffn_fwd": (_I, [_P] * 8 + [_I] * 3 + [_P]),
    "kmu_tail_ffn_bwd": (_I, [_P] * 10 + [_I] * 3 + [_P]),
    "kmu_contingency_counts": (_I, [_P] * 3 + [_Z, _P, _I, _c.c_float, _P]),
}

_lock = threading.Lock()
_lib = None


def load():
    """Return the loaded library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "libkmunet_hip.so is missing (%s). Build it with `python km-unet_amd/build.py` "
                    "(or __graft_entry__.build()); there is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)      # AttributeError => symbol missing from the build
                fn.restype, fn.argtypes = res, args
            if lib.kmu_version() != 1:
                raise RuntimeError("libkmunet_hip.so ABI version mismatch")
            _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().kmu_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))
