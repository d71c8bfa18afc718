"""ctypes loader for librmi_hip.so (the C ABI of include/rmi_hip.h).  Fails loudly when the
library is missing -- there is no CPU fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librmi_hip.so")


class ModelParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("p", C.c_double * 4), ("ip", C.c_uint64 * 4)]


class TrainConfig(C.Structure):
    _fields_ = [("root", ModelParams), ("leaf_kind", C.c_int32), ("_pad", C.c_int32), ("num_leaves", C.c_uint64),
                ("root_table", C.c_void_p), ("root_table_entries", C.c_uint64)]


class Shard(C.Structure):
    _fields_ = [("n_global", C.c_uint64), ("read_lo", C.c_uint64), ("read_hi", C.c_uint64),
                ("key_lo", C.c_uint64), ("key_hi", C.c_uint64), ("leaf_lo", C.c_uint64), ("leaf_hi", C.c_uint64),
                ("split_idx", C.c_uint64), ("split_target", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [
        ("num_rows", C.c_uint64), ("num_leaves", C.c_uint64),
        ("leaf_kind", C.c_int32), ("params_per_leaf", C.c_int32), ("row_bytes", C.c_uint64),
        ("model_avg_error", C.c_double), ("model_avg_l2_error", C.c_double),
        ("model_avg_log2_error", C.c_double), ("model_max_log2_error", C.c_double),
        ("model_max_error", C.c_uint64), ("model_max_error_idx", C.c_uint64),
        ("split_idx", C.c_uint64), ("split_target", C.c_uint64),
        ("shard_leaf_lo", C.c_uint64), ("shard_leaves", C.c_uint64),
        ("sum_n_err", C.c_uint64), ("sum_l2", C.c_double), ("sum_log2", C.c_double),
        ("device_ns", C.c_uint64), ("kernel_ns", C.c_uint64 * 8),
        ("long_leaves", C.c_uint64),
        ("fit_mode_used", C.c_int32), ("merged_leaves", C.c_int32),
        ("exact_leaves", C.c_uint64), ("guard_leaves", C.c_uint64),
        ("generation", C.c_uint64),
    ]


# every symbol include/rmi_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("rmi_hip_abi_version", C.c_int, []),
    ("rmi_hip_last_pipeline", C.c_int, [C.c_void_p]),
    ("rmi_hip_device_count", C.c_int, []),
    ("rmi_hip_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("rmi_hip_destroy", None, [C.c_void_p]),
    ("rmi_hip_last_error", C.c_char_p, [C.c_void_p]),
    ("rmi_hip_strerror", C.c_char_p, [C.c_int]),
    ("rmi_hip_key_buffer", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    ("rmi_hip_train_many", C.c_int, [C.c_void_p, C.POINTER(TrainConfig), C.c_uint64, C.c_int, C.POINTER(Result), C.POINTER(C.c_int)]),
    ("rmi_hip_release_views", C.c_int, [C.c_void_p]),
    ("rmi_hip_measure_read_bandwidth_ex", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("rmi_hip_set_profile_level", C.c_int, [C.c_void_p, C.c_int]),
    ("rmi_hip_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_set_fit_mode", C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    ("rmi_hip_model_from_name", C.c_int, [C.c_char_p]),
    ("rmi_hip_model_name", C.c_char_p, [C.c_int]),
    ("rmi_hip_parse_spec", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("rmi_hip_upload_keys", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    ("rmi_hip_upload_keys_async", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    ("rmi_hip_upload_wait", C.c_int, [C.c_void_p]),
    ("rmi_hip_attach_device_keys", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    ("rmi_hip_num_keys", C.c_uint64, [C.c_void_p]),
    ("rmi_hip_generate_keys", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    ("rmi_hip_download_keys", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_device_keys", C.c_void_p, [C.c_void_p]),
    ("rmi_hip_cache_fix", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rmi_hip_download_cache_fix", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_root_table_entries", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("rmi_hip_download_root_table", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_set_root_table", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("rmi_hip_selftest_div", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rmi_hip_selftest_host_div", C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rmi_hip_selftest_recip", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rmi_hip_measure_read_bandwidth", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    ("rmi_hip_set_shard", C.c_int, [C.c_void_p, C.POINTER(Shard)]),
    ("rmi_hip_set_rows_output", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_fit_root", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(ModelParams)]),
    ("rmi_hip_fit_root_host", C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(ModelParams)]),
    ("rmi_hip_fit_root_fast", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(ModelParams)]),
    ("rmi_hip_root_target", C.c_int, [C.POINTER(ModelParams), C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rmi_hip_root_stream_begin", C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    ("rmi_hip_root_stream_push", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("rmi_hip_root_stream_finish", C.c_int, [C.c_void_p, C.POINTER(ModelParams)]),
    ("rmi_hip_train_two_layer", C.c_int, [C.c_void_p, C.POINTER(ModelParams), C.c_int, C.c_uint64, C.POINTER(Result)]),
    ("rmi_hip_train_streamed", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(ModelParams), C.c_int, C.c_uint64, C.c_int, C.POINTER(Result)]),
    ("rmi_hip_download_leaf_params", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_download_leaf_errors", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_download_leaf_counts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_download_leaf_starts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_download_rows", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rmi_hip_download_checked", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64]),
    ("rmi_hip_device_rows", C.c_void_p, [C.c_void_p]),
    ("rmi_hip_plan_shards", C.c_int, [C.c_void_p, C.POINTER(ModelParams), C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                      C.c_void_p, C.c_void_p, C.POINTER(Shard)]),
    ("rmi_hip_generated_key", C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rmi_hip_fit_root_from_source", C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(ModelParams)]),
    ("rmi_hip_comm_unique_id", C.c_int, [C.c_void_p]),
    ("rmi_hip_comm_init", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("rmi_hip_comm_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("rmi_hip_comm_destroy", C.c_int, [C.c_void_p]),
    ("rmi_hip_train_sharded", C.c_int, [C.c_void_p, C.POINTER(ModelParams), C.c_int, C.c_uint64, C.POINTER(Result)]),
    ("rmi_hip_device_rows_full", C.c_void_p, [C.c_void_p]),
    ("rmi_hip_peer_export", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p]),
    ("rmi_hip_peer_import", C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    ("rmi_hip_set_exchange", C.c_int, [C.c_void_p, C.c_int]),
    ("rmi_hip_download_rows_full", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
]

KEY_AT_FN = C.CFUNCTYPE(C.c_uint64, C.c_void_p, C.c_uint64)
COMM_ID_BYTES = 128
PEER_HANDLE_BYTES = 256

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError(
            f"{SO} is missing: build it with `python -m rmi_amd.build` (hipcc --offload-arch=gfx950). "
            "rmi_amd has no CPU fallback for the training hot path.")
    # RMI_HIP_LIB: development aid for A/B timing against another build of the same ABI
    alt = os.environ.get("RMI_HIP_LIB")
    L = C.CDLL(alt or SO)
    for name, res, args in SYMBOLS:
        if alt and not hasattr(L, name):
            continue
        fn = getattr(L, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
