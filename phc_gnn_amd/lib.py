"""ctypes binding of libphc_hip.so (the C ABI declared in include/phc_hip.h).

There is deliberately no fallback here: if the library is missing or a call
fails, a ``PhcError`` is raised.  Tensors are passed as raw device pointers and
the caller's current HIP stream; PyTorch only provides memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libphc_hip.so")

ABI_VERSION = 8
ACT_IDS = {"identity": 0, "relu": 1, "lrelu": 2, "elu": 3, "swish": 4, "selu": 5}


class PhcError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/phc_hip.h one to one
_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_S = C.c_size_t
class GradJob(C.Structure):
    """phc_phm_grad_job (include/phc_hip.h)."""
    _fields_ = [("x", _P), ("dy", _P), ("A", _P), ("W", _P), ("dW", _P), ("dA", _P), ("db", _P),
                ("N", _L), ("n", C.c_int32), ("kin", C.c_int32), ("kout", C.c_int32), ("flags", C.c_int32),
                ("xf_stat", _P), ("xf_n_valid", _P), ("xf_act", C.c_int32), ("reserved", C.c_int32)]


class HJob(C.Structure):
    """phc_phm_H_job (include/phc_hip.h)."""
    _fields_ = [("A", _P), ("W", _P), ("H", _P), ("HT", _P),
                ("n", C.c_int32), ("kin", C.c_int32), ("kout", C.c_int32), ("flags", C.c_int32),
                ("Hk4", _P), ("HTk4", _P), ("gate_w", _P), ("gate_b", _P), ("phm_b", _P), ("gate_c", _P), ("gate_k", C.c_int32)]


class DownstreamNorm(C.Structure):
    """phc_downstream_norm (include/phc_hip.h)."""
    _fields_ = [("weight", _P), ("bias", _P), ("running_mean", _P), ("running_var", _P), ("num_batches_tracked", _P),
                ("n_comp", C.c_int), ("momentum", C.c_float), ("eps", C.c_float)]


class DownstreamDesc(C.Structure):
    """phc_downstream_desc (include/phc_hip.h)."""
    _fields_ = [("B", _L), ("d0", C.c_int), ("d1", C.c_int), ("d2", C.c_int), ("d3", C.c_int), ("T", C.c_int), ("act", C.c_int),
                ("n_valid_dev", _P),
                ("x", _P), ("H1", _P), ("b1", _P), ("H2", _P), ("b2", _P), ("H3", _P), ("b3", _P), ("H1T", _P), ("H2T", _P),
                ("H3T", _P), ("Wr", _P), ("br", _P),
                ("norm1", DownstreamNorm), ("norm2", DownstreamNorm),
                ("z1", _P), ("z2", _P), ("z3", _P), ("logits", _P), ("stat1", _P), ("stat2", _P),
                ("dlogits", _P),
                ("dx", _P), ("dz1", _P), ("dz2", _P), ("dz3", _P), ("dwb1", _P), ("dwb2", _P), ("dWr", _P),
                ("partial", _P), ("sync", _P),
                ("H1k4", _P), ("H2k4", _P), ("H1Tk4", _P), ("H2Tk4", _P),
                ("drop_p1", C.c_float), ("drop_p2", C.c_float), ("drop_same_k1", C.c_int32), ("drop_same_k2", C.c_int32),
                ("drop_site1", C.c_int32), ("drop_site2", C.c_int32), ("rng_state", _P), ("a1", _P), ("a2", _P),
                ("loss_y", _P), ("loss_kind", C.c_int32), ("loss_out", _P), ("loss_dlogits", _P), ("loss_granules", _P),
                ("fused_bwd", C.c_int32), ("partial_bwd", _P), ("sync_bwd", _P)]


class MlpNormBwdDesc(C.Structure):
    """phc_mlp_norm_bwd_desc (include/phc_hip.h)."""
    _fields_ = [("dout", _P), ("y2", _P), ("z1", _P), ("save_o", _P), ("save_i", _P), ("H2Tk4", _P), ("H1Tk4", _P),
                ("dy2", _P), ("dz1", _P), ("ds", _P), ("dwb_o", _P), ("dwb_i", _P), ("acc_o", _P), ("acc_i", _P),
                ("n_valid_dev", _P), ("N", _L), ("h", C.c_int32), ("act_o", C.c_int32), ("act_i", C.c_int32), ("barrier", _P),
                ("agg_src", _P), ("agg_residual", _P), ("agg_rowptr_t", _P), ("agg_col_t", _P), ("agg_ell_t", _P), ("dx_out", _P),
                ("agg_add_self", C.c_int32)]


class MlpNormFwdDesc(C.Structure):
    """phc_mlp_norm_fwd_desc (include/phc_hip.h)."""
    _fields_ = [("s", _P), ("H1k4", _P), ("b1", _P), ("H2k4", _P), ("b2", _P), ("z1", _P), ("y2", _P), ("acc1", _P), ("acc2", _P),
                ("weight", _P), ("bias", _P), ("running_mean", _P), ("running_var", _P), ("num_batches_tracked", _P),
                ("n_comp", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float), ("save_stat", _P), ("act", C.c_int32),
                ("n_valid_dev", _P), ("N", _L), ("h", C.c_int32), ("barrier", _P),
                ("x", _P), ("codes", _P), ("tables", _P), ("vocab", _P), ("rowptr", _P), ("col", _P), ("ell", _P), ("tables_host", _P), ("F", C.c_int32), ("n", C.c_int32),
                ("add_self", C.c_int32), ("s_out", _P), ("norm_acc", _P), ("weight_o", _P), ("bias_o", _P), ("running_mean_o", _P),
                ("running_var_o", _P), ("num_batches_tracked_o", _P), ("n_comp_o", C.c_int32), ("norm_act", C.c_int32),
                ("momentum_o", C.c_float), ("eps_o", C.c_float), ("skip", _P), ("save_stat_o", _P), ("x_out", _P),
                ("softmax_beta", _P), ("agg_out", _P), ("lse", _P)]


class HeadStage(C.Structure):
    """phc_head_stage (include/phc_hip.h)."""
    _fields_ = [("M", _P), ("bias", _P), ("K", C.c_int32), ("N", C.c_int32), ("post", C.c_int32), ("out", _P), ("z", _P), ("stat", _P),
                ("dwb", _P), ("norm", DownstreamNorm), ("drop_p", C.c_float), ("drop_same_k", C.c_int32), ("drop_site", C.c_int32),
                ("a_out", _P)]


class HeadChainDesc(C.Structure):
    """phc_head_chain_desc (include/phc_hip.h)."""
    _fields_ = [("B", _L), ("n_stages", C.c_int32), ("act", C.c_int32), ("n_valid_dev", _P), ("valid_rows_max", _L), ("in_", _P), ("stage", HeadStage * 4),
                ("work", _P), ("sync", _P), ("rng_state", _P)]


class GemmBaDesc(C.Structure):
    """phc_gemm_ba_desc (include/phc_hip.h)."""
    _fields_ = [("dout", _P), ("bn_x", _P), ("bn_stat", _P), ("sums_acc", _P), ("n_valid_dev", _P), ("dz", _P), ("dwb", _P),
                ("Mk4", _P), ("y", _P), ("N", _L), ("K", C.c_int32), ("Nout", C.c_int32), ("act", C.c_int32), ("bw_act", C.c_int32),
                ("bw_x", _P), ("bw_stat", _P), ("bw_sums_acc", _P), ("count", _L),
                ("drop_p", C.c_float), ("drop_same_k", C.c_int32), ("drop_site", C.c_int32), ("rng_state", _P)]


class GemmAgDesc(C.Structure):
    """phc_gemm_ag_desc (include/phc_hip.h)."""
    _fields_ = [("x", _P), ("codes", _P), ("tables", _P), ("vocab", _P), ("rowptr", _P), ("col", _P), ("N", _L),
                ("F", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("add_self", C.c_int32), ("s_out", _P),
                ("norm_acc", _P), ("weight", _P), ("bias", _P), ("running_mean", _P), ("running_var", _P), ("num_batches_tracked", _P),
                ("n_comp", C.c_int32), ("norm_act", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float),
                ("skip", _P), ("save_stat", _P), ("x_out", _P),
                ("Mk4", _P), ("b", _P), ("Nout", C.c_int32), ("y", _P), ("n_valid_dev", _P), ("stat_acc", _P)]


class ReadoutDesc(C.Structure):
    """phc_readout_desc (include/phc_hip.h)."""
    _fields_ = [("x", _P), ("batch_sorted_i64", _P), ("N", _L), ("num_graphs", _L), ("h", C.c_int32), ("n_comp", C.c_int32), ("kg", C.c_int32),
                ("M", _P), ("c", _P), ("gate", _P), ("out", _P), ("n_valid_dev", _P),
                ("norm_acc", _P), ("weight", _P), ("bias", _P), ("running_mean", _P), ("running_var", _P), ("num_batches_tracked", _P),
                ("norm_n_comp", C.c_int32), ("norm_act", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float),
                ("skip", _P), ("save_stat", _P), ("x_out", _P)]


class EncodeFront(C.Structure):
    """phc_encode_front (include/phc_hip.h)."""
    _fields_ = [("idx_i64", _P), ("tables", _P), ("vocab", _P), ("out", _P), ("N", _L), ("F", C.c_int32), ("n", C.c_int32), ("k", C.c_int32)]


class TailExtras(C.Structure):
    """phc_tail_extras (include/phc_hip.h)."""
    _fields_ = [("gate_job", C.c_int32), ("gate_n", C.c_int32), ("gate_kin", C.c_int32), ("gate_kout", C.c_int32), ("gate_kg", C.c_int32),
                ("gate_H", _P), ("gate_w", _P), ("gate_phm_b", _P), ("gate_A", _P), ("gate_W", _P),
                ("gate_dA", _P), ("gate_dW", _P), ("gate_db", _P), ("gate_dWg", _P),
                ("reg_num", C.c_int32), ("reg_p", C.c_int32), ("reg_coef", C.c_float),
                ("reg_W", _P), ("reg_n", _P), ("reg_cells", _P), ("reg_job", _P),
                ("reg_value", _P), ("reg_loss_inout", _P), ("reg_ws", _P)]


class Copy2dJob(C.Structure):
    """phc_copy2d_job (include/phc_hip.h)."""
    _fields_ = [("src", _P), ("dst", _P), ("src_off", C.c_int64), ("dst_off", C.c_int64), ("src_stride", C.c_int64),
                ("dst_stride", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32)]


class TableGradJob(C.Structure):
    """phc_table_grad_job (include/phc_hip.h)."""
    _fields_ = [("dout", _P), ("x", _P), ("tables", _P), ("dtables_flat", _P)]


SIGNATURES = {
    "phc_version": (_I, []),
    "phc_error_string": (C.c_char_p, [_I]),
    "phc_last_error": (C.c_char_p, []),
    "phc_set_error_word": (_I, [_I, _P]),
    "phc_device_info": (_I, [C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    "phc_phm_build_H": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "phc_phm_linear_fwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "phc_phm_linear_bwd_ws_bytes": (_S, [_L, _I, _I, _I]),
    "phc_phm_linear_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _S, _P]),
    "phc_phm_build_H_batched": (_I, [C.POINTER(HJob), _I, _P]),
    "phc_encode_front_supported": (_I, [_L, _I, _I, _I]),
    "phc_phm_build_H_batched_ex": (_I, [C.POINTER(HJob), _I, C.POINTER(EncodeFront), _P]),
    "phc_gemm_prebuilt": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P]),
    "phc_gemm_stats_max_partials": (_I, []),
    "phc_stat_acc_words": (_S, [_I]),
    "phc_stat_acc_replicas": (_I, []),
    "phc_gemm_prebuilt_stats": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, C.POINTER(_I), _P]),
    "phc_gemm_prebuilt_stats_res": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, C.POINTER(_I), _P]),
    "phc_bn_fwd_from_partials": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _L, _I, C.c_float, C.c_float, _I, _P, _P, _P,
                                      C.c_float, _I, _I, _P, _P]),
    "phc_bn_finalize_from_partials": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _L, _I, C.c_float, C.c_float, _P, _P]),
    "phc_gemm_prebuilt_xf": (_I, [_P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, C.POINTER(_I), _P]),
    "phc_gemm_prebuilt_xf_acc": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, C.c_float, C.c_float, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P,
                                      C.POINTER(_I), _P]),
    "phc_gemm_prebuilt_bwsums": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _I, _P, _P, C.POINTER(_I), _P]),
    "phc_gemm_bwd_apply_supported": (_I, [_I, _I]),
    "phc_stat_acc_read": (_I, [_P, _I, _L, _P, _P]),
    "phc_gate_param_grads": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "phc_debug_hs_stamps": (_I, [_P]),
    "phc_gemm_hs_enabled": (_I, []),
    "phc_gemm_hs_set_enabled": (_I, [_I]),
    "phc_gemm_bwd_apply": (_I, [C.POINTER(GemmBaDesc), _P]),
    "phc_gemm_aggregate_supported": (_I, [_L, _I, _I, _I, _I]),
    "phc_readout_fused_supported": (_I, [_L, _L, _I, _I, _I]),
    "phc_readout_fused_fwd": (_I, [C.POINTER(ReadoutDesc), _P]),
    "phc_debug_readout_stamps": (_I, [_P]),
    "phc_gemm_aggregate_stats": (_I, [C.POINTER(GemmAgDesc), _P]),
    "phc_bn_bwd_partial_acc": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "phc_bn_bwd_partial_acc_drop": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, C.c_float, _I, _I, _P, _P]),
    "phc_bn_bwd_from_partials": (_I, [_P, _P, _P, _P, _I, _P, _L, _I, _I, _P, _P, _P, _S, _P]),
    "phc_phm_param_grads_ws_bytes": (_S, [C.POINTER(GradJob), _I]),
    "phc_phm_param_grads": (_I, [C.POINTER(GradJob), _I, _P, _S, _P]),
    "phc_bn_moments": (_I, [_P, _L, _I, _P, _P, _S, _P]),
    "phc_bn_fwd_from_moments": (_I, [_P, _L, _P, _P, _I, _L, _P, _P, _P, _P, _P, _I, _I, C.c_float, C.c_float, _I, _P, _P, _P, _P]),
    "phc_bn_bwd_sums": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _S, _P]),
    "phc_bn_bwd_apply": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _S, _P]),
    "phc_mlp_norm_fwd_fused_supported": (_I, [_L, _I]),
    "phc_mlp_norm_fwd_fused_agg_supported": (_I, [_L, _I, _I, _I, _I]),
    "phc_mlp_norm_fwd_fused": (_I, [C.POINTER(MlpNormFwdDesc), _P]),
    "phc_mlp_norm_bwd_fused_supported": (_I, [_L, _I]),
    "phc_mlp_norm_bwd_fused_agg_supported": (_I, [_L, _I]),
    "phc_mlp_norm_bwd_fused": (_I, [C.POINTER(MlpNormBwdDesc), _P]),
    "phc_mlp_norm_bwd_fused_barrier_words": (_S, []),
    "phc_debug_mlp_bwd_stamps": (_I, [_P]),
    "phc_debug_mlp_all_stamps": (_I, [_P]),
    "phc_downstream_supported": (_I, [_L, _I, _I, _I, _I, _I]),
    "phc_downstream_supported_k4": (_I, [_L, _I, _I, _I, _I, _I]),
    "phc_downstream_dropout_supported": (_I, [_L, _I, _I, _I, _I, _I]),
    "phc_downstream_loss_supported": (_I, [_L, _I, _I, _I, _I, _I]),
    "phc_downstream_fused_bwd_supported": (_I, [_L, _I, _I, _I, _I, _I]),
    "phc_downstream_partial_floats": (_S, [_L]),
    "phc_downstream_fwd": (_I, [C.POINTER(DownstreamDesc), _P]),
    "phc_downstream_bwd": (_I, [C.POINTER(DownstreamDesc), _P]),
    "phc_head_chain_supported": (_I, [_L, _L, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "phc_head_chain_work_floats": (_S, [_L, _I, C.POINTER(C.c_int32)]),
    "phc_head_chain": (_I, [C.POINTER(HeadChainDesc), _P]),
    "phc_pack_grads": (_I, [_P, _P, _I, _P, _I, _P, _P, C.c_float, _P]),
    "phc_phm_weight_reg_max_jobs": (_I, []),
    "phc_phm_weight_reg_max_blocks": (_I, []),
    "phc_phm_weight_reg_ws_bytes": (_S, []),
    "phc_phm_weight_reg": (_I, [_P, _P, _P, _P, _I, C.c_float, _I, _P, _P, _P, _S, _P]),
    "phc_csr_bucketed_ws_bytes": (_S, [_L, _L]),
    "phc_csr_build_bucketed": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _S, _P]),
    "phc_csr_ws_bytes": (_S, [_L, _L]),
    "phc_csr_build": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _S, _P]),
    "phc_csr_pair_ws_bytes": (_S, [_L, _L, _L]),
    "phc_csr_build_pair": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "phc_csr_build_pair_codes": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _S, _P]),
    "phc_csr_build_pair_codes_resident": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _S, _P]),
    "phc_csr_build_pair_codes_resident_ell": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _S, _P]),
    "phc_csr_build_pair_codes_resident_ell_src": (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _S, _P]),
    "phc_remove_isolated_ws_bytes": (_S, [_L]),
    "phc_remove_isolated_nodes": (_I, [_P, _P, _L, _L, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "phc_segment_ptr_sorted": (_I, [_P, _L, _L, _P, _P]),
    "phc_aggregate_fwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "phc_aggregate_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _P, _P, _P]),
    "phc_gather_rows_i64": (_I, [_P, _P, _L, _I, _P, _P]),
    "phc_aggregate_softmax_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "phc_softmax_bwd_num_partials": (_I, [_L]),
    "phc_aggregate_softmax_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "phc_aggregate_softmax_fwd_self": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "phc_aggregate_softmax_bwd_self": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _L, _P, _P, _P, _P]),
    "phc_aggregate_softmax_enc_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "phc_aggregate_softmax_norm_supported": (_I, [_L, _I, _I, _I]),
    "phc_aggregate_softmax_norm_enc_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I,
                                                _P, _P, _P, _L, _P, _P, _P, _P]),
    "phc_aggregate_softmax_enc_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _L, _I, _L, _P, _P, _P, _P]),
    "phc_aggregate_softmax_enc_bwd_res": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _L, _I, _L, _P, _I, _P, _P, _P, _P]),
    "phc_encode_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "phc_encode_bwd_ws_bytes": (_S, [_L, _I, _P, _I, _I]),
    "phc_encode_bwd": (_I, [_P, _P, _L, _I, _P, _I, _I, _P, _P, _S, _P]),
    "phc_encode_bwd_batched_ws_bytes": (_S, [_L, _I, _P, _I, _I, _I]),
    "phc_encode_bwd_batched": (_I, [C.POINTER(TableGradJob), _I, _P, _L, _I, _P, _I, _I, _P, _S, _P]),
    "phc_aggregate_enc_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "phc_pack_edge_codes": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "phc_aggregate_encp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "phc_aggregate_enc_bwd_ws_bytes": (_S, [_L, _I, _P, _I, _I]),
    "phc_aggregate_enc_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P,
                                   _S, _P]),
    "phc_table_grads_ws_bytes": (_S, [_L, _I, _P, _I, _I, _I]),
    "phc_table_grads_batched": (_I, [C.POINTER(TableGradJob), _I, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _S, _P]),
    "phc_table_grads_counted_ws_bytes": (_S, [_L, _I, _P, _I, _I, _I]),
    "phc_table_grads_counted": (_I, [C.POINTER(TableGradJob), _I, _P, _P, _L, _L, _P, _I, _I, _I, _P, _S, _P]),
    "phc_step_tail_ws_bytes": (_S, [C.POINTER(GradJob), _I, _I, _L, _L, _I, _P, _L, _I, _P, _I, _I]),
    "phc_step_tail": (_I, [C.POINTER(GradJob), _I, C.POINTER(TableGradJob), _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _L, _P, _P, _L, _I,
                           _P, _P, _I, _I, _P, _S, _P]),
    "phc_tail_reg_max_jobs": (_I, []),
    "phc_tail_extras_ws_floats": (_S, [C.POINTER(TailExtras)]),
    "phc_step_tail_ex": (_I, [C.POINTER(GradJob), _I, C.POINTER(TableGradJob), _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _L, _P, _P, _L, _I,
                              _P, _P, _I, _I, _P, _S, C.POINTER(TailExtras), _P]),
    "phc_bn_ws_bytes": (_S, [_L, _I]),
    "phc_bn_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _L, _I, _I, C.c_float, C.c_float, _I, _P, _P, _P, _P, _S, C.c_float, _I, _I,
                        _P, _P]),
    "phc_bn_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _S, C.c_float, _I, _I, _P, _P]),
    "phc_rng_advance": (_I, [_P, _P]),
    "phc_bce_masked_ws_bytes": (_S, [_L]),
    "phc_bce_masked_fwd": (_I, [_P, _P, _L, _P, _P, _P, _S, _P]),
    "phc_bce_masked_bwd": (_I, [_P, _P, _P, _P, _L, _P, _P]),
    "phc_l1_masked_fwd": (_I, [_P, _P, _L, _P, _P, _P, _S, _P]),
    "phc_l1_masked_fwd_unit": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "phc_l1_masked_bwd": (_I, [_P, _P, _P, _P, _L, _P, _P]),
    "phc_bce_masked_fwd_unit": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "phc_gated_pool_fwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P]),
    "phc_gated_pool_fwd_sorted": (_I, [_P, _P, _P, _L, _L, _I, _I, _P, _P]),
    "phc_gated_pool_bwd": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    "phc_gated_pool_bwd_gate_supported": (_I, [_I, _I]),
    "phc_gated_pool_bwd_gate": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P]),
    "phc_sum_n": (_I, [_P, _I, _L, _P, _P]),
    "phc_copy2d_batched": (_I, [_P, _I, _L, _P, _P, _P]),
    "phc_aggregate_dx_rows": (_I, [_P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    "phc_aggregate_norm_supported": (_I, [_L, _I, _I, _I]),
    "phc_aggregate_norm_encp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L,
                                         _I, _I, _I, _I, _P, _P]),
    "phc_aggregate_dx_rows_multi": (_I, [_P, _P, _P, _L, _L, _I, _I, _P, _I, _P, _P]),
    "phc_vec_sums": (_I, [_P, _P, _P, _I, _P]),
    "phc_small_linear_max_weights": (_I, []),
    "phc_small_linear_fwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P]),
    "phc_small_linear_bwd": (_I, [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "phc_clip_adam_max_tensors": (_I, []),
    "phc_clip_adam_chunk": (_I, []),
    "phc_clip_adam_step": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P,
                                _I, _P]),
    "phc_clip_adam_fused_ws_bytes": (C.c_size_t, [_I]),
    "phc_clip_adam_fused_supported": (_I, [_I, _I]),
    "phc_clip_adam_fused": (_I, [_P, _P, _I, _P, _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P, C.c_size_t, _I, _P]),
    "phc_prof_enable": (_I, [_I]),
    "phc_prof_reset": (_I, []),
    "phc_prof_num_kernels": (_I, []),
    "phc_prof_read": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libphc_hip.so (once).  Raises PhcError when it has not been built."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("PHC_LIB_PATH"):                      # experiments: a differently built engine (scripts/exp)
        LIB_PATH = os.environ["PHC_LIB_PATH"]
    if not os.path.exists(LIB_PATH):
        raise PhcError(
            f"{LIB_PATH} not found: the HIP engine is not built (run `python -m phc_gnn_amd.build`). "
            "phc_gnn_amd has no CPU or PyTorch fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise PhcError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise PhcError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.phc_version() != ABI_VERSION:
        raise PhcError(f"ABI version mismatch: library {lib.phc_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        raise PhcError(f"{what} failed: {lib.phc_error_string(rc).decode()} ({rc}): {lib.phc_last_error().decode()}")


def require_device(*tensors: Optional[torch.Tensor]) -> torch.device:
    """All tensors must live on one ROCm device; anything else is a hard error."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise PhcError("phc_gnn_amd runs on MI355X only: got a tensor on "
                           f"'{t.device}'. Move the module and its inputs to a ROCm device "
                           "(there is no CPU fallback).")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise PhcError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise PhcError("no tensor given")
    if dev not in _ERR_WORDS:
        _register_error_word(dev)
    if dev not in _SELF_CHECKED:
        # first engine op on this device: the H-stationary GEMM is compared with the LDS-panel kernel once (selfcheck.py)
        from . import selfcheck
        if not torch.cuda.is_current_stream_capturing():
            _SELF_CHECKED.add(dev)
            selfcheck.hs_self_check(dev)
    return dev


# ---- index validation (include/phc_hip.h: phc_set_error_word) ---------------------------------------------------
# One sticky int32 per device.  Kernels never dereference an out-of-range index (it reads as 0) and OR a bit into the
# word; the host looks at it at its check points: GraphedTrainStep after warm-up, bench.py / smoke at the end,
# `phc_gnn_amd.check_indices()` whenever the caller wants (one 4-byte read = one synchronisation), or after EVERY
# structure build / encoder call with PHC_CHECK_INDICES=1 (debugging aid: the reference's nn.Embedding / scatter raise
# at the call, a device-side assert in their CUDA form at the next synchronisation).
_ERR_WORDS: dict = {}
_SELF_CHECKED: set = set()
_ERR_BITS = {1: "edge_index entry outside [0, num_nodes)", 2: "node feature index outside its embedding's vocabulary",
             4: "edge feature index outside its embedding's vocabulary", 8: "segment id (batch vector) outside [0, num_graphs)"}
ERR_RENDEZVOUS = 16                # PHC_ERR_RENDEZVOUS (include/phc_hip.h)
CHECK_EAGERLY = os.environ.get("PHC_CHECK_INDICES", "0") == "1"


def _register_error_word(dev: torch.device) -> None:
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    check(load().phc_set_error_word(dev.index if dev.index is not None else torch.cuda.current_device(), word.data_ptr()),
          "phc_set_error_word")
    _ERR_WORDS[dev] = word
    if dev not in _EMPTY_BUF and not torch.cuda.is_current_stream_capturing():
        _make_empty_buf(dev)


def check_indices(device: Optional[torch.device] = None) -> None:
    """Raise IndexError if any kernel since the last call met an out-of-range index (synchronises the device)."""
    for dev, word in list(_ERR_WORDS.items()):
        if device is not None and torch.device(device) != dev:
            continue
        bits = int(word.item())
        if bits & ERR_RENDEZVOUS:
            word.zero_()
            raise PhcError(f"phc_gnn_amd ({dev}): the grid-wide exchange of a fused launch (phc_downstream_fwd / _bwd) timed out — its "
                           "workgroups were not co-resident (CU mask / partitioned device / CUs owned by another stream); the step's "
                           "results are invalid")
        if bits:
            word.zero_()
            what = "; ".join(msg for bit, msg in _ERR_BITS.items() if bits & bit)
            raise IndexError(f"phc_gnn_amd ({dev}): index out of range in self: {what} (the offending entries were read as 0)")


def maybe_check_indices(dev: torch.device) -> None:
    if CHECK_EAGERLY:
        check_indices(dev)


_EMPTY_BUF: dict = {}


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address for the C ABI.  An EMPTY tensor (a batch without edges: no sources, no codes) has address 0, which the entry
    points' argument checks would take for a missing argument: it is represented by a small zeroed buffer of its device instead —
    the kernels never index it (their counts are zero), and a clamped read of element 0 would find zeros."""
    if t is None:
        return None
    p = t.data_ptr()
    if p == 0 and t.is_cuda:
        buf = _EMPTY_BUF.get(t.device)
        if buf is None:
            buf = _make_empty_buf(t.device)
        return buf.data_ptr()
    return p


def _make_empty_buf(dev: torch.device) -> torch.Tensor:
    """The device's stand-in for empty tensors (ptr): created on the device's first engine op (_register_error_word), never inside
    a stream capture — there the buffer would land in the graph's private pool and its memset would only be recorded, not run."""
    if torch.cuda.is_current_stream_capturing():
        raise PhcError("an empty tensor reached the engine inside a hipGraph capture before any engine op had run on this device "
                       "eagerly (run one eager warm-up step first: GraphedTrainStep does)")
    buf = _EMPTY_BUF[dev] = torch.zeros(256, dtype=torch.uint8, device=dev)
    return buf


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Contiguous fp32 view/copy (the ABI is dense row-major fp32)."""
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise PhcError(f"fp32 expected, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ---- profiler helpers (used by bench.py) -------------------------------------
# While the launch profiler is on, the operators also book the ALGORITHMIC work of
# each launch (flops / bytes from the formulas in DESIGN.md §4) under the kernel's
# name, so that achieved = booked work / measured time can be formed per kernel.
PROF_ON = False
WORK: dict = {}


def account(kernel: str, flops: float = 0.0, nbytes: float = 0.0) -> None:
    if PROF_ON:
        w = WORK.setdefault(kernel, {"flops": 0.0, "bytes": 0.0, "calls": 0})
        w["flops"] += flops
        w["bytes"] += nbytes
        w["calls"] += 1


def prof_enable(on: bool) -> None:
    global PROF_ON
    check(load().phc_prof_enable(1 if on else 0), "phc_prof_enable")
    PROF_ON = bool(on)


def prof_reset() -> None:
    check(load().phc_prof_reset(), "phc_prof_reset")
    WORK.clear()


def prof_read() -> dict:
    lib = load()
    out = {}
    for k in range(lib.phc_prof_num_kernels()):
        name, n, ms = C.c_char_p(), C.c_uint64(), C.c_double()
        check(lib.phc_prof_read(k, C.byref(name), C.byref(n), C.byref(ms)), "phc_prof_read")
        if n.value:
            out[name.value.decode()] = {"launches": int(n.value), "total_ms": float(ms.value)}
    return out
