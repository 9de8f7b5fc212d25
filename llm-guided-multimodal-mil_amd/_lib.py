"""ctypes binding of ``libmil_hip.so`` (C ABI: include/mil_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  ``lib()`` raises
if the shared object has not been built (``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C llm-guided-multimodal-mil_amd/csrc``)."""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_long, c_float, c_int, c_int32, c_size_t, c_uint32, c_uint64, c_void_p
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmil_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mil_hip.h")
ABI_VERSION = 6

_P = c_void_p
# name -> (restype, argtypes); mirrors include/mil_hip.h one to one
SIGNATURES: Dict[str, Tuple[object, List[object]]] = {
    "mil_abi_version": (c_int, []),
    "mil_dropout_keep_bits": (c_int, [_P, c_int, c_int, c_float, c_uint64, c_uint64, _P, _P]),
    "mil_dropout_keep_bits_pair": (c_int, [_P, c_int, _P, c_int, c_int, c_uint64, c_uint64, c_uint64, _P, _P, _P, c_int, _P]),
    "mil_dropout_apply_bits": (c_int, [_P, _P, c_int, c_int, c_float, _P]),
    "mil_counter_add": (c_int, [_P, c_int, _P]),
    "mil_gate_scores_fwd": (c_int, [_P] * 9 + [c_int, c_int, c_int, _P, c_float, _P]),
    "mil_gate_scores_fwd_draw": (c_int, [_P] * 9 + [c_int, c_int, c_int, _P, c_float, _P, c_int, c_uint64, c_uint64, c_uint64, _P, _P]),
    "mil_attn_pool_fwd": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P]),
    "mil_attn_pool_partial": (c_int, [_P] * 3 + [c_int, c_int, _P, _P, c_float, _P]),
    "mil_attn_pool_partial_h": (c_int, [_P] * 3 + [c_int, c_int, _P, _P, c_int, _P, _P, c_float, _P, c_float, _P]),
    "mil_attn_pool_bwd_from_h": (c_int, [_P] * 6 + [c_int, c_int, _P, _P]),
    "mil_pool_merge_head": (c_int, [_P] * 2 + [c_int, c_int, c_int, _P, _P, c_int, _P, c_float] + [_P] * 12
                            + [_P, c_float, _P, c_int, _P]),
    "mil_pool_merge_head_ws": (c_int, [_P] * 2 + [c_int, c_int, c_int, _P, _P, c_int, _P, c_float] + [_P] * 12
                               + [_P, c_float, _P, c_int, _P, _P]),
    "mil_pool_tail_workspace_floats": (c_size_t, [c_int]),
    "mil_set_i32": (c_int, [_P, _P, c_int, _P]),
    "mil_head_fwd": (c_int, [_P] * 5 + [c_int, c_int, c_int, _P]),
    "mil_bce_fwd_bwd": (c_int, [_P] * 4 + [c_int, c_int, c_float, _P]),
    "mil_head_bwd": (c_int, [_P] * 8 + [c_int, c_int, c_int, _P]),
    "mil_head_bwd_params": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, _P, _P]),
    "mil_head_bwd_params_acc": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, _P, c_int, _P]),
    "mil_clip_contrastive_loss": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "mil_cosine_embedding_loss": (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P, _P]),
    "mil_rowdot": (c_int, [_P] * 3 + [c_int, c_int, _P]),
    "mil_attn_pool_bwd": (c_int, [_P] * 6 + [c_int, c_int, _P, _P, _P, c_float, _P]),
    "mil_gate_bwd_workspace_floats": (c_size_t, [c_int, c_int]),
    "mil_gate_bwd_params": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, c_size_t] + [_P] * 6 + [c_int, _P, c_float, _P]),
    "mil_gate_bwd_params_head": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, c_size_t] + [_P] * 6 + [c_int] + [_P] * 4
                                 + [c_int, c_int, _P, _P, _P, c_float, _P]),
    "mil_gate_bwd_partials": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "mil_gate_bwd_partials_rows": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "mil_build_tile_map": (c_int, [_P, c_int, _P, _P, _P, c_int, _P]),
    "mil_gate_bwd_reduce": (c_int, [_P, c_int, c_int] + [_P] * 6 + [c_int, c_float, _P]),
    "mil_gate_bwd_reduce_head": (c_int, [_P, c_int, c_int] + [_P] * 6 + [c_int, c_float] + [_P] * 4 + [c_int, c_int, _P, _P, _P]),
    "mil_gate_bwd_reduce_head_adam": (c_int, [_P, c_int, c_int] + [_P] * 6 + [c_int, c_float] + [_P] * 4 + [c_int, c_int, _P, _P]
                                      + [_P, _P, c_size_t, _P, _P, c_int] + [c_float] * 6 + [_P]),
    "mil_gate_bwd_input": (c_int, [_P] * 5 + [c_int, c_int, c_int, _P, _P, c_float, _P]),
    "mil_gate_bwd_input_pool": (c_int, [_P] * 5 + [c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, _P]),
    "mil_image_only_step_run": (c_int, [_P, _P]),
    "mil_image_only_step_time": (c_int, [_P, c_uint32, c_int, c_int, _P, _P]),
    "mil_image_only_step_profile": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "mil_image_only_step_profile_rot": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "mil_patch_drop_select": (c_int, [_P, _P, _P, c_int, c_int, c_uint64, c_uint64, _P, _P]),
    "mil_cohort_feed": (c_int, [_P, _P, _P, _P, _P, _P]),
    "mil_cast_bf16": (c_int, [_P, _P, c_size_t, _P]),
    "mil_gate_scores_fwd_bf16": (c_int, [_P] * 9 + [c_int, c_int, c_int, _P, _P, c_float, _P]),
    "mil_attn_pool_partial_bf16": (c_int, [_P] * 3 + [c_int, c_int, _P, _P, c_float, _P]),
    "mil_attn_pool_partial_h_bf16": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int, _P, _P, c_float, _P, c_float, _P]),
    "mil_attn_pool_bwd_bf16": (c_int, [_P] * 6 + [c_int, c_int, _P, _P, c_float, _P]),
    "mil_gate_bwd_workspace_floats_bf16": (c_size_t, [c_int, c_int]),
    "mil_gate_bwd_params_bf16": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, c_size_t] + [_P] * 6 + [c_int, _P, c_float, _P]),
    "mil_gate_bwd_params_x16": (c_int, [_P] * 4 + [c_int, c_int, c_int, _P, c_size_t] + [_P] * 6 + [c_int, _P, c_float, _P]),
    "mil_gemm_workspace_floats": (c_size_t, [c_int] * 4),
    "mil_gemm": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int,
                         _P, c_size_t, _P]),
    "mil_gemm_rows": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int,
                         _P, c_size_t, _P, _P]),
    "mil_gemm_nt2_ok": (c_int, [c_int] * 5),
    "mil_gemm_tn2_ok": (c_int, [c_int] * 6),
    "mil_gemm_tn2_splits": (c_int, [c_int] * 3),
    "mil_gemm_tn2": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "mil_gemm_nt2": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "mil_split_bf16": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "mil_gemm_split": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int,
                               c_int, _P]),
    "mil_gemm_grouped": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int,
                                 c_long, c_long, _P, c_long, _P, c_int, _P, c_size_t, _P]),
    "mil_gemm_grouped_pad": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int,
                                 c_long, c_long, _P, c_long, _P, c_int, _P, c_size_t, c_int, _P]),
    "mil_gemm_grouped_workspace_floats": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mil_gemm_aux": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int,
                         _P, c_size_t, _P, c_int, c_int, _P]),
    "mil_colsum_workspace_floats": (c_size_t, [c_int, c_int]),
    "mil_colsum": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P]),
    "mil_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "mil_linear_small_fwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "mil_linear_small_bwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P,
                                     c_int, c_int, c_int, _P]),
    "mil_linear_small_fwd_add": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int,
                                         c_int, _P]),
    "mil_linear_small_bwd_sum": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P,
                                         c_int, _P, c_int, c_int, c_int, _P]),
    "mil_linear_small_bwd_split": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "mil_linear_small_ln_bwd5": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P,
                                         _P, c_int, c_int, _P]),
    "mil_sum4": (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    "mil_linear_small_ln_bwd3": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, _P,
                                         c_int, c_int, _P]),
    "mil_linear_small_dw_grouped": (c_int, [_P, c_int, _P]),
    "mil_linear_small_ln_fwd": (c_int, [_P, c_int, _P, _P, c_float, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, _P, _P,
                                        c_int, c_int, _P]),
    "mil_linear_small_ln_bwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, _P]),
    "mil_attn_rows_fwd": (c_int, [_P] * 6 + [c_int] * 4 + [_P, _P, _P]),
    "mil_attn_rows_bwd": (c_int, [_P] * 9 + [c_int] * 4 + [_P] * 4 + [_P]),
    "mil_attn_rows_bwd_general": (c_int, [_P] * 10 + [c_int, c_int, c_int, c_int] + [_P] * 5),
    "mil_attn_pool_fwd_mh": (c_int, [_P] * 6 + [c_int] * 5 + [_P] * 3 + [_P]),
    "mil_attn_pool_bwd_mh": (c_int, [_P] * 9 + [c_int] * 5 + [_P] * 4 + [_P]),
    "mil_attn_seq_fwd": (c_int, [_P] * 3 + [c_int, _P] + [c_int] * 5 + [_P, _P, _P]),
    "mil_attn_seq_bwd": (c_int, [_P] * 3 + [c_int] + [_P] * 4 + [c_int] * 5 + [_P] * 3 + [c_int, _P]),
    "mil_quickgelu": (c_int, [_P, _P, _P, c_size_t, _P]),
    "mil_absorb_query": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "mil_absorb_query_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "mil_absorb_query_pad": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "mil_absorb_query_bwd_pad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P]),
    "mil_absorbed_pool_fwd": (c_int, [_P] * 6 + [c_int] * 5 + [_P] * 3 + [_P]),
    "mil_absorbed_pool_bwd": (c_int, [_P] * 9 + [c_int] * 6 + [_P] * 4 + [_P]),
    "mil_lnbr_absorbed_pool_value_fwd": (c_int, [_P] * 4 + [c_float] + [_P] * 5 + [c_int] * 5 + [_P] * 8 + [_P]),
    "mil_lnbr_absorbed_pool_bwd": (c_int, [_P] * 14 + [c_int] * 6 + [_P] * 7 + [_P]),
    "mil_absorbed_pool_value_fwd": (c_int, [_P] * 6 + [c_int] * 5 + [_P] * 6 + [_P]),
    "mil_value_proj_bwd": (c_int, [_P] * 3 + [c_int] * 4 + [_P] * 3 + [_P]),
    "mil_grp_col_softmax": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "mil_grp_col_softmax_bwd": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "mil_grp_col_softmax_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "mil_grp_col_softmax_ws": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "mil_grp_col_softmax_bwd_ws": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P]),
    "mil_row_softmax_t": (c_int, [_P, c_int, c_int, c_int, c_int, _P]),
    "mil_row_softmax_t_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "mil_value_proj": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "mil_value_proj_pad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "mil_layernorm_fwd": (c_int, [_P] * 3 + [c_int, c_int, c_float, _P, _P, _P]),
    "mil_layernorm_bwd_blocks": (c_int, [c_int]),
    "mil_layernorm_bwd": (c_int, [_P] * 4 + [c_int, c_int] + [_P] * 4 + [_P]),
    "mil_layernorm_bwd_res": (c_int, [_P] * 5 + [c_int, c_int] + [_P] * 4 + [_P]),
    "mil_layernorm_bagrow_fwd": (c_int, [_P] * 5 + [c_int, c_int, c_float, _P, _P, _P]),
    "mil_layernorm_bagrow_bwd": (c_int, [_P] * 4 + [c_int] + [_P] * 3 + [c_int, c_int] + [_P] * 5 + [_P]),
    "mil_layernorm_bagrow_rows_per_block": (c_int, [c_int]),
    "mil_build_fusion_segs": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P]),
    "mil_build_fusion_segs_tail": (c_int, [_P, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P]),
    "mil_add_pe": (c_int, [_P] * 4 + [c_int, c_int, _P, _P]),
    "mil_add_bag_row": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "mil_segment_colsum": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "mil_sinusoid_pe": (c_int, [_P, c_int, c_int, _P]),
    "mil_ct_map_tokens": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "mil_embed_tokens": (c_int, [_P] * 3 + [c_int] * 3 + [_P, _P]),
    "mil_gather_eot": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    "mil_adam_step": (c_int, [_P] * 4 + [c_size_t, c_int] + [c_float] * 6 + [_P]),
    "mil_adam_step_counted": (c_int, [_P] * 4 + [c_size_t, _P] + [c_float] * 6 + [_P]),
    "mil_adam_step_counted_noinc": (c_int, [_P] * 4 + [c_size_t, _P] + [c_float] * 6 + [_P]),
    "mil_adam_step_dev": (c_int, [_P] * 4 + [c_size_t, _P, _P] + [c_float] * 5 + [c_int, _P]),
    "mil_adam_step_dev_segs": (c_int, [_P] * 4 + [_P, _P, c_int, _P, _P, _P] + [c_float] * 5 + [c_int, _P]),
    "mil_sgd_step": (c_int, [_P, _P, c_size_t] + [c_float] * 3 + [_P]),
    "mil_linear_mid_fwd": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "mil_linear_mid_bwd": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int,
                                   c_int, _P]),
    "mil_linear_bwd_params_workspace_floats": (c_size_t, [c_int] * 3),
    "mil_linear_bwd_params": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P,
                                      c_size_t, _P]),
    "mil_linear_bwd_params_rows": (c_int, [_P, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P,
                                      c_size_t, _P, _P]),
}

STAGE_DROPBITS, STAGE_GATE_FWD, STAGE_POOL, STAGE_TAIL, STAGE_GATE_BWD, STAGE_REDUCE, STAGE_ADAM = 1, 2, 4, 8, 16, 32, 64
STAGE_TILEMAP = 0x80
STAGE_ALL = 0xff
STAGE_POOL_FUSED = 0x100      # hint: all tiles full and aligned (include/mil_hip.h) - pool partial pass inside the forward launch


class SmallDwDesc(ctypes.Structure):
    """Mirror of mil_small_dw_desc (include/mil_hip.h); layout checked against the header by tests/test_abi.py."""
    _fields_ = [("dy", c_void_p), ("yv", c_void_p), ("x", c_void_p), ("dW", c_void_p), ("db", c_void_p),
                ("lddy", c_int32), ("ldyv", c_int32), ("ldx", c_int32), ("lddw", c_int32), ("act", c_int32),
                ("M", c_int32), ("N", c_int32), ("K", c_int32)]


SMALL_DW_MAX = 32
FEED_MAX_BAGS, FEED_MAX_AUX = 8, 4


class CohortFeedDesc(ctypes.Structure):
    """Mirror of mil_cohort_feed_desc (include/mil_hip.h); layout checked against the header by tests/test_abi.py."""
    _fields_ = [("struct_bytes", c_uint32), ("nb", c_int32), ("L", c_int32), ("dst_row0", c_int32), ("dst_bag0", c_int32),
                ("naux", c_int32), ("sel_off", c_int32 * FEED_MAX_BAGS), ("src_row0", c_int32 * FEED_MAX_BAGS),
                ("rows", c_int32 * FEED_MAX_BAGS), ("bag_id", c_int32 * FEED_MAX_BAGS), ("aux_words", c_int32 * FEED_MAX_AUX),
                ("aux_table", c_void_p * FEED_MAX_AUX), ("aux_dst", c_void_p * FEED_MAX_AUX)]


class ImageOnlyStep(ctypes.Structure):
    """`mil_image_only_step` of include/mil_hip.h, field for field."""
    _fields_ = (
        [("struct_bytes", c_uint32), ("stages", c_uint32),
         ("x", _P), ("y", _P), ("tile_map", _P), ("bag_tile_off", _P),
         ("R", c_int32), ("L", c_int32), ("B", c_int32), ("C", c_int32), ("T", c_int32),
         ("bag_len_dev", _P), ("rows_dev", _P),
         ("x_bf16", c_int32), ("loss_scale", c_float), ("loss_kind", c_int32), ("accumulate", c_int32)]
        + [(n, _P) for n in ("Wv", "bv", "Wu", "bu", "w", "b", "Wf", "bf", "Wv16", "Wu16")]
        + [(n, _P) for n in ("dWv", "dbv", "dWu", "dbu", "dw", "db", "dWf", "dbf", "loss_out")]
        + [(n, _P) for n in ("scores", "gates", "gates16", "partials", "hrow", "ds", "dw_ws")]
        + [("dw_ws_floats", c_uint64)]
        + [(n, _P) for n in ("M", "Mdrop", "lse", "logits", "prob", "loss_bag", "dz", "dM", "cdot")]
        + [("train", c_int32), ("bf16_grad_mfma", c_int32), ("xbits", _P), ("mbits", _P), ("seed", c_uint64),
           ("offset", c_uint64), ("offset_dev", _P)]
        + [(n, _P) for n in ("param_flat", "grad_flat", "exp_avg", "exp_avg_sq")]
        + [("n_param", c_uint64), ("adam_step", c_int32), ("adam_step_dev", _P)]
        + [(n, c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "grad_scale")]
        + [("lr_dev", _P), ("tail_ws", _P), ("done_dev", _P)])


_lib = None


class MilHipError(RuntimeError):
    pass


def header_symbols() -> List[str]:
    """Function names declared in include/mil_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mil_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        # MIL_HIP_LIB: load another build of the same library (tools/variants/*.so: A/B timing of kernel variants)
        path = os.environ.get("MIL_HIP_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise MilHipError(
                f"{path} is missing: the HIP library is not built and there is no fallback path. "
                "Run __graft_entry__.build() or `make -C llm-guided-multimodal-mil_amd/csrc`.")
        import torch  # noqa: F401  (loads torch's libamdhip64 first so the kernels share its runtime/streams)
        handle = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        got = handle.mil_abi_version()
        if got != ABI_VERSION:
            raise MilHipError(f"libmil_hip.so ABI {got} != expected {ABI_VERSION}")
        _lib = handle
    return _lib


SHIM_DIR = os.path.join(_HERE, "csrc", "shim_build")
SHIM_SRC = os.path.join(_HERE, "csrc", "torch_shim.cpp")
_shim = False


def build_shim(verbose: bool = False):
    """Compile csrc/torch_shim.cpp (a torch cpp_extension: pybind11 + at::Tensor, no kernels) against libmil_hip.so into
    csrc/shim_build/mil_torch_shim.so - in-tree, so it travels to the GPU box with the library."""
    from torch.utils.cpp_extension import load
    os.makedirs(SHIM_DIR, exist_ok=True)
    return load(name="mil_torch_shim", sources=[SHIM_SRC], extra_ldflags=[f"-L{_HERE}", "-lmil_hip", f"-Wl,-rpath,{_HERE}"],
                extra_cflags=["-O2"], build_directory=SHIM_DIR, with_cuda=False, verbose=verbose)


def shim():
    """The torch cpp_extension binding of the token-side entries (csrc/torch_shim.cpp) when MIL_TORCH_SHIM=1 and it has been
    built, else None: ops.py then reaches the same C functions through ctypes.  Opt-in because it measured no faster: the
    eager fusion step (32 bags x 1024 x 768) takes 3.01 ms through it and 2.91 ms through ctypes - what the host spends per
    launch is autograd-node bookkeeping, not argument marshalling."""
    global _shim
    if _shim is False:
        _shim = None
        path = os.path.join(SHIM_DIR, "mil_torch_shim.so")
        if os.environ.get("MIL_TORCH_SHIM", "0") == "1" and os.path.exists(path) and not os.environ.get("MIL_HIP_LIB"):
            import importlib.util
            lib()                                     # libmil_hip.so first: the extension's DT_NEEDED resolves to it by soname
            try:
                spec = importlib.util.spec_from_file_location("mil_torch_shim", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.abi_version() == ABI_VERSION:
                    _shim = mod
            except (ImportError, OSError, AttributeError):
                _shim = None                          # stale build (another torch / ABI): the ctypes binding serves
    return _shim


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-22: "invalid argument (MIL_EINVAL)", -28: "workspace too small (MIL_ENOSPC)"}.get(
            rc, f"hipError {rc}" if rc > 0 else f"error {rc}")
        raise MilHipError(f"{what}: {kind}")
