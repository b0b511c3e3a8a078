"""ctypes binding of libmsgat_hip.so -- the only door between Python and the HIP kernels.

The structures and prototypes mirror include/msgat_hip.h one to one.  There is no CPU
fallback anywhere in this package: if the library is missing, `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libmsgat_hip.so")

MSGAT_OK = 0
ABI_VERSION = 7  # MSGAT_ABI_VERSION of include/msgat_hip.h
MODE_PLAIN, MODE_AGG_FIRST, MODE_PROJ_FIRST = 0, 1, 2

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


SELL_SLACK = 512  # MSGAT_SELL_SLACK


class Sell(C.Structure):
    """msgat_sell_t: degree-sorted sliced-ELLPACK form of the CSR rows / CSC columns (n_slices = 0: absent)."""
    _fields_ = [
        ("n_slices", C.c_int32), ("n_pos", C.c_int32),
        ("slice_off", C.c_void_p), ("lane_row", C.c_void_p), ("idx", C.c_void_p), ("src", C.c_void_p),
        ("pos", C.c_void_p), ("prefer", C.c_int32), ("pair_trips", C.c_int32),
    ]


class Graph(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("nnz", C.c_int32),
        ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p), ("erow", C.c_void_p),
        ("colptr", C.c_void_p), ("crow", C.c_void_p), ("cperm", C.c_void_p), ("cpos", C.c_void_p),
        ("sell_rows", Sell), ("sell_cols", Sell),
    ]


class Shape(C.Structure):
    _fields_ = [("R", C.c_int32), ("Bg", C.c_int32), ("C", C.c_int32), ("Co", C.c_int32),
                ("N", C.c_int32), ("T", C.c_int32)]


class Fwd(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "alpha", "Wg", "W", "z", "q", "kW", "lse", "pq", "E", "u")] + [
        ("need_bwd", C.c_int32), ("edge_scratch", C.c_void_p), ("Ec", C.c_void_p), ("dense_scratch", C.c_void_p)]


class Bwd(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "x", "alpha", "Wg", "W", "q", "kW", "lse", "pq", "E", "u", "dz", "dx", "dalpha", "dWg", "dW",
        "workspace")] + [("workspace_bytes", C.c_size_t), ("dz_group_channels", C.c_int32), ("Ec", C.c_void_p)]


class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("channels", C.c_int32), ("group_stride", C.c_int32)]


_PROTOTYPES = {
    "msgat_abi_version": (C.c_int, []),
    "msgat_status_string": (C.c_char_p, [C.c_int]),
    "msgat_gacn_mode": (C.c_int, [C.c_int32, C.c_int32]),
    "msgat_graph_count": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, c_int_p]),
    "msgat_graph_build": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 8),
    "msgat_graph_validate": (C.c_int, [C.POINTER(Graph)]),
    "msgat_graph_sell_count": (C.c_int, [C.c_void_p, C.c_int32, c_int_p, c_int_p, c_int_p]),
    "msgat_graph_sell_build": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p] * 5),
    "msgat_edge_scratch_floats": (C.c_size_t, [C.POINTER(Shape), C.POINTER(Graph)]),
    "msgat_gacn_forward": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph), C.POINTER(Fwd), C.c_void_p]),
    "msgat_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(Shape), C.POINTER(Graph)]),
    "msgat_gacn_backward": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph), C.POINTER(Bwd), C.c_void_p]),
    "msgat_stage_project": (C.c_int, [C.POINTER(Shape)] + [C.c_void_p] * 6),
    "msgat_stage_scores": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph)] + [C.c_void_p] * 9),
    "msgat_stage_dense_column_pass": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph)] + [C.c_void_p] * 8),
    "msgat_dense_scratch_bytes": (C.c_size_t, [C.POINTER(Shape)]),
    "msgat_stage_aggregate": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph), C.c_int32] + [C.c_void_p] * 5),
    "msgat_stage_aggregate_project": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph)] + [C.c_void_p] * 6),
    "msgat_stage_mix": (C.c_int, [C.POINTER(Shape), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "msgat_contract_partial_floats": (C.c_size_t, [C.POINTER(Shape), C.c_int32, C.c_int32]),
    "msgat_stage_contract": (C.c_int, [C.POINTER(Shape), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "msgat_stage_mix_epilogue": (C.c_int, [C.POINTER(Shape), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "msgat_time_mix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 8
                       + [C.c_void_p]),
    "msgat_time_mix_partial_floats": (C.c_size_t, [C.c_int32] * 3),
    "msgat_causal_conv_grad_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 8
                                      + [C.c_void_p]),
    "msgat_causal_conv_fused": (C.c_int, [C.c_int32, C.c_int32]),
    "msgat_causal_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 9 + [C.c_void_p]),
    "msgat_time_mix_grad_matrix": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_node_pool": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "msgat_node_pool_grad_signal": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "msgat_node_pool_partial_floats": (C.c_size_t, [C.c_int32] * 3),
    "msgat_node_pool_grad_weight": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 5 + [C.c_void_p]),
    "msgat_mix_segments": (C.c_int, [C.c_int32] * 4 + [C.POINTER(Seg), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                     C.POINTER(Seg), C.c_int32, C.c_int32, C.POINTER(Seg), C.c_int32, C.c_void_p]),
    "msgat_contract_segments_partial_floats": (C.c_size_t, [C.c_int32] * 3),
    "msgat_contract_segments": (C.c_int, [C.c_int32] * 4 + [C.POINTER(Seg), C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "msgat_stage_project_backward": (C.c_int, [C.POINTER(Shape)] + [C.c_void_p] * 10),
    "msgat_contract_mix_segments": (C.c_int, [C.c_int32] * 4 + [C.POINTER(Seg), C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "msgat_contract_mix_partial_floats": (C.c_size_t, [C.c_int32] * 6),
    "msgat_contract_form_name": (C.c_int, [C.c_int32] * 5 + [C.c_char_p, C.c_int32]),
    "msgat_attention_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(Shape), C.POINTER(Graph)]),
    "msgat_attention_backward": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph), C.c_void_p, C.c_void_p, C.c_int32] +
                                 [C.c_void_p] * 11 + [C.c_size_t, C.c_void_p]),
    "msgat_attention_bwd_accepts_strided_dv": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph)]),
    "msgat_bwd_accepts_strided_dz": (C.c_int, [C.POINTER(Shape), C.POINTER(Graph)]),
    "msgat_head_forward_partial_floats": (C.c_size_t, [C.c_int32] * 4),
    "msgat_head_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_head_forward_ln": (C.c_int, [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_gate_sum": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "msgat_gate_sum_backward": (C.c_int, [C.c_void_p] * 9 + [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "msgat_head_grad_signal": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_head_grad_weight_partial_floats": (C.c_size_t, [C.c_int32] * 4),
    "msgat_head_grad_weight": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_layernorm_pool_partial_floats": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "msgat_layernorm_forward_pooled": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                                C.c_int32, C.c_void_p]),
    "msgat_layernorm_backward_pooled": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] + [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_float,
                                                 C.c_int32, C.c_int32, C.c_void_p]),
    "msgat_layernorm_head_backward_partial_floats": (C.c_size_t, [C.c_int32] * 4),
    "msgat_layernorm_head_backward": (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 6 + [C.c_float, C.c_int32, C.c_void_p]),
    "msgat_layernorm_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "msgat_layernorm_partial_floats": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "msgat_layernorm_backward": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "msgat_channel_attention_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p]),
    "msgat_channel_attention_partial_floats": (C.c_size_t, [C.c_int32] * 4),
    "msgat_channel_attention_backward": (C.c_int, [C.c_void_p] * 9 + [C.c_int32] * 5 + [C.c_void_p]),
    "msgat_temporal_attention_forward": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_temporal_attention_partial_floats": (C.c_size_t, [C.c_int32] * 3),
    "msgat_temporal_attention_backward": (C.c_int, [C.c_void_p] * 10 + [C.c_int32] * 6 + [C.c_void_p]),
    "msgat_huber_partial_doubles": (C.c_size_t, [C.c_int64]),
    "msgat_huber_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float] + [C.c_void_p] * 3
                            + [C.c_float, C.c_void_p]),
    "msgat_huber_grad": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "msgat_adam_chunk_elems": (C.c_int, []),
    "msgat_adam_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_double] * 4 + [C.c_void_p, C.c_void_p]),
    "msgat_gather_scaled": (C.c_int, [C.c_void_p] * 3 + [C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
}

_lock = threading.Lock()
_handle = None


class MsgatError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built (no fallback path exists)."""
    global _handle
    if _handle is not None:
        return _handle
    with _lock:
        if _handle is None:
            if not os.path.exists(LIB_PATH):
                raise MsgatError(
                    f"{LIB_PATH} is missing: build it with `python -m ms_gat_amd.build` "
                    "(hipcc --offload-arch=gfx950). ms_gat_amd has no CPU or eager fallback.")
            h = C.CDLL(LIB_PATH)
            for name, (res, args) in _PROTOTYPES.items():
                fn = getattr(h, name)  # AttributeError here = header/library mismatch
                fn.restype, fn.argtypes = res, args
            if h.msgat_abi_version() != ABI_VERSION:
                raise MsgatError("libmsgat_hip.so ABI version mismatch; rebuild")
            _handle = h
    return _handle


def check(status: int, what: str) -> None:
    if status != MSGAT_OK:
        msg = lib().msgat_status_string(status)
        raise MsgatError(f"{what} failed: {msg.decode() if msg else status} (status {status})")


def contract_form_name(Ca: int, Cb: int, with_ones: bool, n_positions: int, with_mix: bool) -> str:
    """Which kernel form the backward of a 1x1 convolution (with_mix) or a plain channel-pair contraction takes for this
    shape: `msgat_contract_form_name` (host only, no device needed)."""
    buf = C.create_string_buffer(160)
    check(lib().msgat_contract_form_name(Ca, Cb, int(with_ones), n_positions, int(with_mix), buf, 160), "msgat_contract_form_name")
    return buf.value.decode()


def exported_symbols():
    return sorted(_PROTOTYPES)
