"""ctypes binding for the CPU oracle (oracle/rmi_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never from ``rmi_amd`` (the product).
PARITY UNPINNED: see oracle/rmi_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librmi_oracle.so")

KEY_U64, KEY_U32, KEY_F64 = 0, 1, 2
MODEL_LINEAR, MODEL_LINEAR_SPLINE, MODEL_CUBIC, MODEL_RADIX, MODEL_ROBUST_LINEAR = 0, 1, 2, 3, 4
MODEL_IDS = {
    "linear": MODEL_LINEAR,
    "linear_spline": MODEL_LINEAR_SPLINE,
    "cubic": MODEL_CUBIC,
    "radix": MODEL_RADIX,
    "robust_linear": MODEL_ROBUST_LINEAR,
    "loglinear": 5, "normal": 6,
    "radix8": 8, "radix18": 9, "radix22": 10, "radix26": 11, "radix28": 12,      # RadixTable (train/mod.rs:46-50)
    "bradix": 13,                                                                 # BalancedRadixModel
}
ERRORS = {
    -1: "unknown model (train/mod.rs:53)",
    -2: "layer restriction (train/mod.rs:69-82)",
    -3: "non-monotone root (two_layer.rs:50)",
    -4: "degenerate split (two_layer.rs:27/144)",
    -5: "root prediction out of bounds (two_layer.rs:45)",
    -6: "bad argument",
    -7: "negative variance (linear.rs:48)",
    -8: "robust_linear needs more data (linear.rs:248)",
    -9: "num_bits assert (utils.rs:18)",
    -10: "cubic unwrap on None (cubic_spline.rs:50/61)",
}


class OracleError(RuntimeError):
    def __init__(self, code: int):
        super().__init__(f"oracle error {code}: {ERRORS.get(code, '?')}")
        self.code = code


class _Model(C.Structure):
    _fields_ = [("kind", C.c_int), ("p", C.c_double * 4), ("ip", C.c_uint64 * 4),
                ("table", C.POINTER(C.c_uint32)), ("table_len", C.c_uint64)]


class _Trained(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("num_leaves", C.c_uint64),
        ("root", _Model),
        ("leaf_kind", C.c_int),
        ("params_per_leaf", C.c_int),
        ("leaf_params", C.POINTER(C.c_double)),
        ("leaf_err", C.POINTER(C.c_uint64)),
        ("leaf_count", C.POINTER(C.c_uint64)),
        ("leaf_start", C.POINTER(C.c_uint64)),
        ("model_avg_error", C.c_double),
        ("model_avg_l2_error", C.c_double),
        ("model_avg_log2_error", C.c_double),
        ("model_max_error", C.c_uint64),
        ("model_max_error_idx", C.c_uint64),
        ("model_max_log2_error", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc) next to its source; returns the .so path."""
    src = os.path.join(_HERE, "rmi_oracle.c")
    hdr = os.path.join(_HERE, "rmi_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librmi_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_fit_pairs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.POINTER(_Model)]
        L.orc_fit_pairs.restype = C.c_int
        L.orc_predict_to_float.argtypes = [C.POINTER(_Model), C.c_int, C.c_uint64]
        L.orc_predict_to_float.restype = C.c_double
        L.orc_predict_to_int.argtypes = [C.POINTER(_Model), C.c_int, C.c_uint64]
        L.orc_predict_to_int.restype = C.c_uint64
        L.orc_num_bits.argtypes = [C.c_uint64]
        L.orc_num_bits.restype = C.c_int
        L.orc_common_prefix_size.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.orc_common_prefix_size.restype = C.c_int
        L.orc_fit_root.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(_Model)]
        L.orc_fit_root.restype = C.c_int
        L.orc_train_two_layer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64,
                                          C.POINTER(_Model), C.c_int, C.POINTER(_Trained)]
        L.orc_train_two_layer.restype = C.c_int
        L.orc_bucket_ids.argtypes = [C.POINTER(_Model), C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_bucket_ids.restype = C.c_int
        L.orc_check_lookup_property.argtypes = [C.POINTER(_Trained), C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_check_lookup_property.restype = C.c_uint64
        L.orc_cache_fix.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
        L.orc_cache_fix.restype = C.c_int
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        L.orc_check_bounded_property.argtypes = [C.POINTER(_Trained), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                                 C.POINTER(C.c_uint64)]
        L.orc_check_bounded_property.restype = C.c_uint64
        L.orc_model_free.argtypes = [C.POINTER(_Model)]
        L.orc_model_free.restype = None
        L.orc_version.restype = C.c_char_p
        _lib = L
    return _lib


def dtype_of(keys: np.ndarray) -> int:
    if keys.dtype == np.uint64:
        return KEY_U64
    if keys.dtype == np.uint32:
        return KEY_U32
    if keys.dtype == np.float64:
        return KEY_F64
    raise TypeError(f"unsupported key dtype {keys.dtype}")


def _kind(name_or_id) -> int:
    if isinstance(name_or_id, str):
        if name_or_id not in MODEL_IDS:
            raise OracleError(-1)
        return MODEL_IDS[name_or_id]
    return int(name_or_id)


@dataclass
class Model:
    kind: int
    p: tuple
    ip: tuple
    table: np.ndarray | None = None       # radix tables: hint_table (u32), radix.rs:83-88

    def __post_init__(self):
        self.ip = tuple(int(v) for v in self.ip) + (0,) * (4 - len(self.ip))

    def _c(self) -> _Model:
        m = _Model()
        m.kind = self.kind
        for i in range(4):
            m.p[i] = self.p[i]
        for i in range(4):
            m.ip[i] = self.ip[i]
        if self.table is not None:
            m.table = self.table.ctypes.data_as(C.POINTER(C.c_uint32))     # borrowed: self keeps it alive
            m.table_len = self.table.size
        return m

    def predict_to_int(self, key, dtype=KEY_U64) -> int:
        return int(lib().orc_predict_to_int(C.byref(self._c()), dtype, _key_bits(key, dtype)))

    def predict_to_float(self, key, dtype=KEY_U64) -> float:
        return float(lib().orc_predict_to_float(C.byref(self._c()), dtype, _key_bits(key, dtype)))


def _key_bits(key, dtype) -> int:
    if dtype == KEY_F64:
        return int(np.array([key], dtype=np.float64).view(np.uint64)[0])
    return int(key)


def _from_c(m: _Model, own_table: bool = False) -> Model:
    table = None
    if m.table_len:
        table = np.ctypeslib.as_array(m.table, shape=(int(m.table_len),)).copy()
        if own_table:
            lib().orc_model_free(C.byref(m))
    return Model(int(m.kind), tuple(float(x) for x in m.p), tuple(int(x) for x in m.ip), table)


def fit_pairs(kind, keys, ys, scale: float = 1.0) -> Model:
    keys = np.ascontiguousarray(keys)
    ys = np.ascontiguousarray(ys, dtype=np.uint64)
    m = _Model()
    rc = lib().orc_fit_pairs(_kind(kind), dtype_of(keys), keys.ctypes.data, ys.ctypes.data, len(keys), scale, C.byref(m))
    if rc:
        raise OracleError(rc)
    return _from_c(m, own_table=True)


def fit_root(kind, keys: np.ndarray, num_leaves: int) -> Model:
    keys = np.ascontiguousarray(keys)
    m = _Model()
    rc = lib().orc_fit_root(_kind(kind), dtype_of(keys), keys.ctypes.data, len(keys), num_leaves, C.byref(m))
    if rc:
        raise OracleError(rc)
    return _from_c(m, own_table=True)


def num_bits(t: int) -> int:
    return int(lib().orc_num_bits(t))


def common_prefix_size(keys: np.ndarray) -> int:
    keys = np.ascontiguousarray(keys)
    return int(lib().orc_common_prefix_size(dtype_of(keys), keys.ctypes.data, len(keys)))


def bucket_ids(root: Model, keys: np.ndarray, num_leaves: int) -> np.ndarray:
    keys = np.ascontiguousarray(keys)
    out = np.empty(len(keys), dtype=np.uint64)
    lib().orc_bucket_ids(C.byref(root._c()), dtype_of(keys), keys.ctypes.data, len(keys), num_leaves, out.ctypes.data)
    return out


@dataclass
class TrainedRMI:
    """Mirror of TrainedRMI (rmi_lib/src/train/mod.rs:18-33) as produced by the oracle."""
    n: int
    num_leaves: int
    root: Model
    leaf_kind: int
    params_per_leaf: int
    leaf_params: np.ndarray   # [L, ppl] f64
    leaf_err: np.ndarray      # [L] u64  (last_layer_max_l1s)
    leaf_count: np.ndarray    # [L] u64
    leaf_start: np.ndarray    # [L+1] u64
    model_avg_error: float
    model_avg_l2_error: float
    model_avg_log2_error: float
    model_max_error: int
    model_max_error_idx: int
    model_max_log2_error: float
    _c: object = field(default=None, repr=False)


def train_two_layer(root_kind, leaf_kind, keys: np.ndarray, num_leaves: int,
                    root: Model | None = None, threads: int = 1) -> TrainedRMI:
    keys = np.ascontiguousarray(keys)
    L = int(num_leaves)
    lk = _kind(leaf_kind)
    ppl = 4 if lk == MODEL_CUBIC else 2
    leaf_params = np.zeros((L, ppl), dtype=np.float64)
    leaf_err = np.zeros(L, dtype=np.uint64)
    leaf_count = np.zeros(L, dtype=np.uint64)
    leaf_start = np.zeros(L + 1, dtype=np.uint64)
    t = _Trained()
    t.leaf_params = leaf_params.ctypes.data_as(C.POINTER(C.c_double))
    t.leaf_err = leaf_err.ctypes.data_as(C.POINTER(C.c_uint64))
    t.leaf_count = leaf_count.ctypes.data_as(C.POINTER(C.c_uint64))
    t.leaf_start = leaf_start.ctypes.data_as(C.POINTER(C.c_uint64))
    if root is None and 8 <= _kind(root_kind) <= 12:
        root = fit_root(root_kind, keys, L)              # the table then lives in numpy, not in C
    rootc = root._c() if root is not None else None
    rc = lib().orc_train_two_layer(_kind(root_kind), lk, dtype_of(keys), keys.ctypes.data, len(keys), L,
                                   C.byref(rootc) if rootc is not None else None, threads, C.byref(t))
    if rc:
        raise OracleError(rc)
    return TrainedRMI(
        n=int(t.n), num_leaves=L, root=(root if root is not None and root.table is not None else _from_c(t.root)), leaf_kind=lk, params_per_leaf=ppl,
        leaf_params=leaf_params, leaf_err=leaf_err, leaf_count=leaf_count, leaf_start=leaf_start,
        model_avg_error=float(t.model_avg_error), model_avg_l2_error=float(t.model_avg_l2_error),
        model_avg_log2_error=float(t.model_avg_log2_error), model_max_error=int(t.model_max_error),
        model_max_error_idx=int(t.model_max_error_idx), model_max_log2_error=float(t.model_max_log2_error),
        _c=(t, root, leaf_params, leaf_err, leaf_count, leaf_start),
    )


def check_lookup_property(rmi: TrainedRMI, keys: np.ndarray):
    """Returns (num_violations, first_bad_index)."""
    keys = np.ascontiguousarray(keys)
    fb = C.c_uint64(0)
    bad = lib().orc_check_lookup_property(C.byref(rmi._c[0]), dtype_of(keys), keys.ctypes.data, len(keys), C.byref(fb))
    return int(bad), int(fb.value)


def cache_fix(keys: np.ndarray, line_size: int) -> np.ndarray:
    """cache_fix.rs:109-150 -> [m, 2] uint64 (key, offset) spline points."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    ptr = C.POINTER(C.c_uint64)()
    cnt = C.c_uint64(0)
    rc = lib().orc_cache_fix(keys.ctypes.data, len(keys), line_size, C.byref(ptr), C.byref(cnt))
    if rc:
        raise OracleError(rc)
    out = np.ctypeslib.as_array(ptr, shape=(int(cnt.value) * 2,)).copy().reshape(-1, 2)
    lib().orc_free(ptr)
    return out


def train_bounded(root_kind, leaf_kind, keys: np.ndarray, num_leaves: int, line_size: int):
    """train_bounded (train/mod.rs:156-184): RMI over the re-indexed spline points.  Returns (rmi, spline)."""
    spline = cache_fix(keys, line_size)
    rmi = train_two_layer(root_kind, leaf_kind, np.ascontiguousarray(spline[:, 0]), num_leaves)
    return rmi, spline


def check_bounded_property(rmi: TrainedRMI, spline: np.ndarray, line_size: int, keys: np.ndarray):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    sp = np.ascontiguousarray(spline, dtype=np.uint64)
    fb = C.c_uint64(0)
    bad = lib().orc_check_bounded_property(C.byref(rmi._c[0]), sp.ctypes.data, len(sp), line_size, keys.ctypes.data, len(keys), C.byref(fb))
    return int(bad), int(fb.value)
