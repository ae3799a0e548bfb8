"""ctypes binding of libgnnmp.so (include/gnnmp.h).

The product path has NO fallback: if the shared library is missing or a symbol
is absent this module raises, and every operator in the package goes through it.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgnnmp.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gnnmp.h")

OK = 0


class GnnmpError(RuntimeError):
    pass


STEP_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gnnmp_step.h")


def declared_symbols() -> List[str]:
    """Every function name declared in include/gnnmp.h and include/gnnmp_step.h (used by the export test)."""
    text = open(HEADER_PATH).read() + open(STEP_HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmp_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnnmpError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gnn_pretraining_amd/csrc`). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


p, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t


class AugMasksJob(C.Structure):          # gmp_aug_masks_job (include/gnnmp.h)
    _fields_ = [("ptr", C.c_void_p), ("out_ptr", C.c_void_p), ("num_graphs", C.c_int32), ("stream_id", C.c_uint32), ("out_idx", C.c_void_p)]


class AugViewsJob(C.Structure):          # gmp_aug_views_job
    _fields_ = [("ptr", C.c_void_p), ("eptr", C.c_void_p), ("edge_index", C.c_void_p), ("num_nodes", C.c_int64), ("num_edges", C.c_int64),
                ("view_ptr", C.c_void_p), ("num_graphs", C.c_int32), ("num_features", C.c_int32), ("stream_id", C.c_uint32),
                ("rows1", C.c_void_p), ("rows2", C.c_void_p), ("rowmask1", C.c_void_p), ("rowmask2", C.c_void_p), ("edges1", C.c_void_p),
                ("edges2", C.c_void_p), ("edge_capacity", C.c_int64), ("common1", C.c_void_p), ("common2", C.c_void_p), ("counts", C.c_void_p),
                ("totals_and_flags", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class BnConfig(C.Structure):
    _fields_ = [("training", C.c_int), ("relu", C.c_int), ("eps", C.c_float), ("momentum", C.c_float),
                ("dropout_p", C.c_float), ("seed", C.c_uint64), ("stream_id", C.c_uint32), ("seed_dev", C.c_void_p),
                ("sync", C.c_void_p), ("sync_words", C.c_uint32)]


_SIGS: Dict[str, tuple] = {
    "gmp_version": (C.c_int, []),
    "gmp_last_error_string": (C.c_char_p, []),
    "gmp_csr_build_workspace_bytes": (sz, [i64, i64]),
    "gmp_csr_build": (C.c_int, [p, i64, i64, p, p, p, p, p, p, p, p, sz, p]),
    "gmp_csr_build_segmented": (C.c_int, [p, i64, i64, p, p, i32, i64, i64, p, p, p, p, p, p, p, p]),
    "gmp_gin_aggregate_fwd": (C.c_int, [p, p, p, p, p, i64, i32, p]),
    "gmp_gin_aggregate_fwd_rows": (C.c_int, [p, p, p, p, p, i64, i64, i32, p]),
    "gmp_gin_aggregate_bwd_workspace_bytes": (sz, [i64, i32]),
    "gmp_gin_aggregate_bwd": (C.c_int, [p, p, p, p, p, p, p, i64, i32, p, sz, p]),
    "gmp_segment_sum": (C.c_int, [p, p, p, p, i64, i32, i32, i32, p]),
    "gmp_row_gather": (C.c_int, [p, p, p, p, i64, i64, i32, p]),
    "gmp_segment_max_fwd": (C.c_int, [p, p, p, i64, i32, p]),
    "gmp_segment_max_bwd": (C.c_int, [p, p, p, p, p, i64, i32, i32, p]),
    "gmp_gemm_f32_workspace_bytes": (sz, [i32, i64, i64, i64]),
    "gmp_gemm_f32": (C.c_int, [i32, p, p, p, p, i64, i64, i64, i64, i64, i64, f32, i32, i32, p, sz, p]),
    "gmp_gemm_f32_grouped": (C.c_int, [i32, p, p, p, p, i32, p, p, p, p, p, p, i64, i64, i64, i64, i64, i64, f32, i32, i32, p, sz, p]),
    "gmp_gin_aggregate_bwd_ex": (C.c_int, [p, p, p, p, p, p, p, p, i64, i32, p]),
    "gmp_group_sum_1d": (C.c_int, [p, i32, p, p, p, p]),
    "gmp_colsum_workspace_bytes": (sz, [i64, i64]),
    "gmp_colsum": (C.c_int, [p, p, i64, i64, i64, i32, p, sz, p]),
    "gmp_bn_workspace_bytes": (sz, [i64, i32, i32, i64]),
    "gmp_bn_sync_bytes": (sz, [i32, i32]),
    "gmp_bn_fwd": (C.c_int, [p, p, p, p, i32, i64, i64, i32, p, p, p, p, p, p, p, C.POINTER(BnConfig), p, sz, p]),
    "gmp_bn_param_grads": (C.c_int, [p, i32, i32, p, p, p, p, p, i32, p]),
    "gmp_bn_running_update_batch": (C.c_int, [i32, p, i32, p, p, p, p, p, p, p, p]),
    "gmp_bn_running_update": (C.c_int, [p, p, i32, i32, p, p, p, p, p, p]),
    "gmp_bn_bwd": (C.c_int, [p, p, p, p, p, i32, i64, i64, i32, p, p, p, p, p, p, p, p, p, p, p, p, i32,
                             C.POINTER(BnConfig), p, sz, p]),
    "gmp_lp_edge_features_fwd": (C.c_int, [p, p, p, i64, i64, i32, p]),
    "gmp_lp_edge_features_bwd": (C.c_int, [p, p, p, p, p, i64, i64, i32, p]),
    "gmp_nt_xent_workspace_bytes": (sz, [i64, i32]),
    "gmp_nt_xent_fwd": (C.c_int, [p, p, i64, i32, f32, p, p, sz, p]),
    "gmp_nt_xent_bwd": (C.c_int, [p, p, i64, i32, f32, p, p, p, p, sz, p]),
    "gmp_nt_xent_grouped_workspace_bytes": (sz, [i32, i64, i32]),
    "gmp_nt_xent_grouped": (C.c_int, [p, p, i32, p, p, i32, f32, p, p, p, p, sz, p]),
    "gmp_dropout_fwd": (C.c_int, [p, p, i64, f32, C.c_uint64, C.c_uint32, p]),
    "gmp_relu_dropout_bwd": (C.c_int, [p, p, p, i64, f32, C.c_uint64, C.c_uint32, p]),
    "gmp_dropout_rowdot_fwd": (C.c_int, [p, p, p, p, p, i64, i32, C.c_float, C.c_uint64, C.c_uint32, p]),
    "gmp_outer_relu_dropout_bwd": (C.c_int, [p, p, p, p, i64, i32, C.c_float, C.c_uint64, C.c_uint32, p]),
    "gmp_weighted_colsum_workspace_bytes": (sz, [i64, i32]),
    "gmp_weighted_colsum": (C.c_int, [p, p, p, p, i64, i32, p, sz, p]),
    "gmp_lp_pair_rowdot_fwd": (C.c_int, [p, p, p, p, p, i64, i32, C.c_float, C.c_uint64, C.c_uint32, p]),
    "gmp_lp_pair_sigmoid_bce_fwd_bwd": (C.c_int, [p, p, p, i64, p, p, p, p, p, sz, p]),
    "gmp_lp_pair_outer_bwd": (C.c_int, [p, p, p, p, p, i64, i32, C.c_float, C.c_uint64, C.c_uint32, p]),
    "gmp_lp_pair_colsum_workspace_bytes": (sz, [i64, i32]),
    "gmp_lp_pair_weighted_colsum": (C.c_int, [p, p, p, p, p, i64, i32, C.c_float, C.c_uint64, C.c_uint32, p, sz, p]),
    "gmp_loss_workspace_bytes": (sz, [i64]),
    "gmp_mse_sum_fwd": (C.c_int, [p, p, i64, p, p, sz, p]),
    "gmp_mse_sum_bwd": (C.c_int, [p, p, p, p, i64, p]),
    "gmp_sigmoid_fwd": (C.c_int, [p, p, i64, p]),
    "gmp_sigmoid_bwd": (C.c_int, [p, p, p, i64, p]),
    "gmp_bce_sum_fwd": (C.c_int, [p, p, i64, p, p, sz, p]),
    "gmp_bce_sum_bwd": (C.c_int, [p, p, p, p, i64, p]),
    "gmp_sigmoid_bce_sum_fwd_bwd": (C.c_int, [p, p, i64, p, p, p, p, p, sz, p]),
    "gmp_sigmoid_bce_signed_sum_fwd_bwd": (C.c_int, [p, p, i64, p, p, p, p, p, sz, p]),
    "gmp_cross_entropy_sum_fwd": (C.c_int, [p, p, i64, i32, p, p, sz, p]),
    "gmp_cross_entropy_sum_bwd": (C.c_int, [p, p, i64, i32, p, p, p]),
    "gmp_row_fill": (C.c_int, [p, p, p, i64, i64, i32, i32, p]),
    "gmp_hard_negative_workspace_bytes": (sz, [i64, i64]),
    "gmp_hard_negative_topk": (C.c_int, [p, i64, i64, p, i64, i64, p, p, p, p, sz, p]),
    "gmp_streams_share_queue": (C.c_int, [p, p, p]),
    "gmp_spin_us": (C.c_int, [i32, p]),
    "gmp_gate_wait": (C.c_int, [p, C.c_uint64, i32, p, p]),
    "gmp_gate_open": (C.c_int, [p, i32, p]),
    "gmp_gate_open_by_next_gemm": (C.c_int, [p, i32]),
    "gmp_gate_open_pending": (C.c_int, []),
    "gmp_gate_set_timeout": (C.c_int, [C.c_double]),
    "gmp_counter_add": (C.c_int, [p, C.c_uint64, p]),
    "gmp_aug_workspace_bytes": (sz, [i64, i64, i32]),
    "gmp_aug_node_masks": (C.c_int, [p, p, i32, i64, C.c_uint64, C.c_uint32, p, p]),
    "gmp_aug_node_masks_batch": (C.c_int, [C.POINTER(AugMasksJob), i32, i64, C.c_uint64, p]),
    "gmp_aug_two_views_batch": (C.c_int, [C.POINTER(AugViewsJob), i32, i64, i64, C.c_uint64, p]),
    "gmp_aug_two_views": (C.c_int, [p, p, p, i64, i64, p, i32, i64, i64, i32, C.c_uint64, C.c_uint32, p, p, p, p, p, p, i64, p, p, p, p, p, sz, p]),
    "gmp_upload": (C.c_int, [i32, p, p, p, p]),
    "gmp_segments_pack": (C.c_int, [p, p, p, i32, i64, p]),
    "gmp_segments_unpack": (C.c_int, [p, p, p, i32, i64, f32, p]),
    "gmp_encoder_fwd": (C.c_int, [p, i64, i64, i32, p, p, p, p, p, i32, p, i32, p, p, p, i32, p, p]),
    "gmp_encoder_bwd": (C.c_int, [p, i64, i64, i32, p, p, p, p, p, i32, p, i32, i32, p, p, p, p, p, sz, p]),
    "gmp_step_desc_size": (sz, []),
    "gmp_pretrain_step_fwd_bwd": (C.c_int, [p, p, p, p]),
    "gmp_step_wait_grads": (C.c_int, [i32, p]),
    "gmp_step_phase_ms": (C.c_int, [p]),
    "gmp_step_phase_detail_ms": (C.c_int, [p]),
    "gmp_step_head_ms": (C.c_int, [p, i32]),
    "gmp_mt_workspace_bytes": (sz, [i32]),
    "gmp_mt_pcgrad_clip_adamw": (C.c_int, [p, i64, i32, i32, p, p, p, p, i32, i32, i32, p, p, p, p, p, p, f32, f32, f32,
                                           f32, p, p, p, p, p, sz, i32, p]),
    "gmp_mt_pcgrad_clip_adamw_ex": (C.c_int, [p, i64, i32, i32, p, p, p, p, i32, i32, i32, p, p, p, p, p, p, f32, f32, f32,
                                              f32, p, p, p, p, p, sz, i32, i32, i32, i32, p, p]),
}


def _declare(l: C.CDLL) -> None:
    missing = [s for s in declared_symbols() if not hasattr(l, s)]
    if missing:
        raise GnnmpError(f"libgnnmp.so lacks symbols declared in gnnmp.h: {missing}")
    for name, (res, args) in _SIGS.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    from ._step_desc import StepDesc
    if l.gmp_step_desc_size() != C.sizeof(StepDesc):
        raise GnnmpError(f"gnnmp_step.h / _step_desc.py disagree: sizeof(gmp_step_desc) = {l.gmp_step_desc_size()} "
                         f"but the ctypes mirror is {C.sizeof(StepDesc)} bytes")


def check(rc: int, what: str) -> None:
    if rc != OK:
        msg = lib().gmp_last_error_string()
        raise GnnmpError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
