"""Host-side mirror of the reference's training surface for two-layer RMIs.

Reference (all under /root/reference/rmi_lib/src): ``train()`` train/mod.rs:100-126,
``TrainedRMI`` train/mod.rs:18-33, model registry train/mod.rs:35-57, ``RMITrainingData``
models/mod.rs:233-317.  The arithmetic runs in the HIP kernels behind include/rmi_hip.h.
"""
from __future__ import annotations

import ctypes as C
import re
import threading
import time
from dataclasses import dataclass, field

import numpy as np

from . import _lib

KEY_U64, KEY_U32, KEY_F64 = 0, 1, 2
_DTYPES = {np.dtype(np.uint64): KEY_U64, np.dtype(np.uint32): KEY_U32, np.dtype(np.float64): KEY_F64}

# train/mod.rs:37-54
MODEL_NAMES = ["linear", "linear_spline", "cubic", "radix", "robust_linear", "loglinear", "normal",
               "lognormal", "radix8", "radix18", "radix22", "radix26", "radix28", "bradix", "histogram"]


class RMIError(RuntimeError):
    """A condition on which the reference panics, reported as a code (include/rmi_hip.h)."""

    def __init__(self, code: int, detail: str = ""):
        lib = _lib.load()
        msg = lib.rmi_hip_strerror(code).decode()
        super().__init__(f"rmi_hip error {code}: {msg}" + (f" [{detail}]" if detail else ""))
        self.code = code


def _check(rc: int, ctx=None):
    if rc != 0:
        detail = ""
        if ctx is not None and rc == -14:
            detail = _lib.load().rmi_hip_last_error(ctx).decode()
        raise RMIError(rc, detail)


def parse_spec(spec: str):
    """validate() + the two-layer check of train() (train/mod.rs:59-85, 104-125)."""
    lib = _lib.load()
    r, l = C.c_int(), C.c_int()
    _check(lib.rmi_hip_parse_spec(spec.encode(), C.byref(r), C.byref(l)))
    return r.value, l.value


@dataclass
class Model:
    """Parameters of one model in `params()` order (models/mod.rs:742)."""
    kind: int
    p: tuple = (0.0, 0.0, 0.0, 0.0)
    ip: tuple = (0, 0, 0, 0)       # radix: (prefix, bits); bradix: (prefix, bits, clamp, high)
    table: object = field(default=None, repr=False, compare=False)   # radix tables: hint_table, np.uint32 (radix.rs:83-88)

    def __post_init__(self):
        self.ip = tuple(int(v) for v in self.ip) + (0,) * (4 - len(self.ip))

    @property
    def name(self) -> str:
        return MODEL_NAMES[self.kind]

    @property
    def is_radix_table(self) -> bool:
        return 8 <= self.kind <= 12

    def _c(self) -> _lib.ModelParams:
        key = (self.kind, self.p, self.ip)
        cached = self.__dict__.get("_cstruct")
        if cached is not None and cached[0] == key:            # (train_leaves is called ~1000x per second)
            return cached[1]
        m = _lib.ModelParams()
        m.kind = self.kind
        for i in range(4):
            m.p[i] = self.p[i]
        for i in range(4):
            m.ip[i] = self.ip[i]
        self.__dict__["_cstruct"] = (key, m)
        return m

    @staticmethod
    def _from_c(m) -> "Model":
        return Model(int(m.kind), tuple(float(x) for x in m.p), tuple(int(x) for x in m.ip))


@dataclass
class TrainedRMI:
    """train/mod.rs:18-33.  Per-leaf arrays are downloaded lazily from HBM."""
    num_rmi_rows: int
    num_data_rows: int
    model_avg_error: float
    model_avg_l2_error: float
    model_avg_log2_error: float
    model_max_error: int
    model_max_error_idx: int
    model_max_log2_error: float
    models: str
    branching_factor: int
    root: Model
    leaf_kind: int
    params_per_leaf: int
    build_time: int = 0            # ns, train/mod.rs:103-118
    device_ns: int = 0
    kernel_ns: tuple = ()
    long_leaves: int = 0           # leaves handed to the one-lane-per-leaf kernel (skew diagnostic)
    split_idx: int = 0
    split_target: int = 0
    shard_leaf_lo: int = 0          # multi-GPU: this object covers leaves [shard_leaf_lo, +shard_leaves)
    shard_leaves: int = 0
    partial: dict = field(default_factory=dict)   # per-shard partial sums of the aggregates
    cache_fix: object = None
    fit_mode_used: int = 0          # 0 exact, 1 one pass + guard, 2 one pass (rmi_hip_set_fit_mode)
    exact_leaves: int = 0           # one-pass modes: leaves re-fitted by the exact kernels
    merged_leaves: int = 0          # one-pass mode 2: long leaves fitted from merged partial sums
    guard_leaves: int = 0           # ... of which flagged by the guard (mode 2: counted only)
    generation: int = 0             # which train call of the trainer's context produced the per-leaf arrays
    pipeline: int = 0               # leaf kernels that ran: 4 k_leaf_regs (one read of the keys), 3 k_leaf_lanes, 2 / 1 the older pipelines
    _trainer: object = field(default=None, repr=False)
    _cache: dict = field(default_factory=dict, repr=False)

    def _get(self, what: str):
        """Per-leaf arrays live in the trainer's context until its next train call: fetch them before
        (``materialize()``), or get a ``RuntimeError`` instead of another training's arrays."""
        if what not in self._cache:
            self._cache[what] = self._trainer._download(what, self)
        return self._cache[what]

    def materialize(self) -> "TrainedRMI":
        """Download every per-leaf array now (the object then no longer depends on the trainer)."""
        _ = self.leaf_params, self.last_layer_max_l1s, self.leaf_counts, self.leaf_starts, self.rows
        return self

    @property
    def leaf_params(self) -> np.ndarray:        # rmi[1][j].params()
        return self._get("params")

    @property
    def last_layer_max_l1s(self) -> np.ndarray:
        return self._get("errors")

    @property
    def leaf_counts(self) -> np.ndarray:
        return self._get("counts")

    @property
    def leaf_starts(self) -> np.ndarray:
        return self._get("starts")

    @property
    def rows(self) -> np.ndarray:
        """Byte image of the reference's L1_PARAMETERS file (codegen.rs:288-315)."""
        return self._get("rows")


class Trainer:
    """Owns one device context with the key array resident in HBM (the role of
    RMITrainingData + soft_copy in the reference: many train() calls over one data set)."""

    def __init__(self, keys=None, device: int = 0):
        self._lib = _lib.load()
        self._device = device
        h = C.c_void_p()
        _check(self._lib.rmi_hip_create(device, C.byref(h)))
        self._h = h
        self._host_keys = None
        self._keepalive = None
        self._table_in_ctx = None
        self._ctx_lock = threading.RLock()
        self.n = 0
        if keys is not None:
            self.set_keys(keys)

    def view(self) -> "Trainer":
        """A second context on the same resident keys (borrowed, not copied): trainings issued through
        different views can be in flight together, one host thread each."""
        self.wait_keys()
        ptr, n, dt = C.c_void_p(), C.c_uint64(), C.c_int()
        _check(self._lib.rmi_hip_key_buffer(self._h, C.byref(ptr), C.byref(n), C.byref(dt)), self._h)
        v = Trainer(device=self._device)
        _check(self._lib.rmi_hip_attach_device_keys(v._h, ptr, n.value, dt.value), v._h)
        v._keepalive = self                       # the keys belong to this trainer
        v._host_keys = self._host_keys
        v.n = int(n.value)
        if hasattr(self, "_np_dtype"):
            v._np_dtype = self._np_dtype
        return v

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmi_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_keys(self, keys, wait: bool = True):
        """numpy array (copied to HBM) or a torch CUDA tensor (borrowed in place).  wait=False: the copy of a host
        array runs on a thread of the library; ``wait_keys()`` (called by fit_root / train_leaves) joins it -- so
        that the host root fit of a new key set overlaps its upload."""
        if isinstance(keys, np.ndarray) or isinstance(keys, np.memmap):
            arr = np.ascontiguousarray(keys)
            if arr.dtype not in _DTYPES:
                raise TypeError(f"unsupported key dtype {arr.dtype}")
            self.wait_keys()
            if wait:
                _check(self._lib.rmi_hip_upload_keys(self._h, arr.ctypes.data, arr.size, _DTYPES[arr.dtype]), self._h)
            else:
                _check(self._lib.rmi_hip_upload_keys_async(self._h, arr.ctypes.data, arr.size, _DTYPES[arr.dtype]), self._h)
                self._uploading = True
            self._host_keys = arr
            self._keepalive = None
            self.n = arr.size
            return
        # torch tensor on the GPU: uint64 is carried as int64 bit patterns
        import torch
        if isinstance(keys, torch.Tensor) and keys.is_cuda:
            self.wait_keys()                       # (an upload still running on the library's thread would replace these keys)
            t = keys.contiguous()
            if t.dtype in (torch.int64, torch.uint64):
                dt = KEY_U64
            elif t.dtype in (torch.int32, torch.uint32):
                dt = KEY_U32
            elif t.dtype == torch.float64:
                dt = KEY_F64
            else:
                raise TypeError(f"unsupported tensor dtype {t.dtype}")
            _check(self._lib.rmi_hip_attach_device_keys(self._h, C.c_void_p(t.data_ptr()), t.numel(), dt), self._h)
            self._keepalive = t
            self._host_keys = None
            self.n = t.numel()
            return
        raise TypeError("keys must be a numpy array or a CUDA torch tensor")

    def wait_keys(self):
        if getattr(self, "_uploading", False):
            self._uploading = False
            _check(self._lib.rmi_hip_upload_wait(self._h), self._h)

    def generate_keys(self, generator: str, dtype, n_global: int, start: int = 0, count: int | None = None, seed: int = 0):
        """Synthetic sorted keys produced directly in HBM (datagen.uniform_* / dups_* shards)."""
        gen = {"uniform": 0, "dups": 1}[generator]
        self.wait_keys()
        dt = _DTYPES[np.dtype(dtype)]
        count = n_global - start if count is None else count
        _check(self._lib.rmi_hip_generate_keys(self._h, gen, dt, n_global, start, count, seed), self._h)
        self._host_keys = None
        self._keepalive = None
        self._np_dtype = np.dtype(dtype)
        self.n = count

    def download_keys(self) -> np.ndarray:
        if self._host_keys is not None:
            return self._host_keys
        self.wait_keys()
        a = np.empty(self.n, dtype=getattr(self, "_np_dtype", np.dtype(np.uint64)))
        _check(self._lib.rmi_hip_download_keys(self._h, a.ctypes.data), self._h)
        self._host_keys = a
        return a

    def measure_read_bandwidth(self, iters: int = 10, pattern: int = 0) -> float:
        """GB/s of a read-only streaming kernel over the resident keys (the box's achievable HBM rate).  pattern 0: grid-stride
        16-byte loads; 1: contiguous 8 KB pieces per wave, non-temporal loads (the pattern of the one-read kernels)."""
        v = C.c_double()
        self.wait_keys()
        _check(self._lib.rmi_hip_measure_read_bandwidth_ex(self._h, iters, int(pattern), C.byref(v)), self._h)
        return float(v.value)

    def set_profile_level(self, level: int):
        """-1: no events (device_ns = 0); 0: device_ns only; 1: + kernel_ns[0] (the first, dominant kernel); 2: every kernel group."""
        _check(self._lib.rmi_hip_set_profile_level(self._h, int(level)), self._h)

    def set_fit_mode(self, mode: int | str, guard_k: float = 0.0):
        """How linear leaves are fitted (include/rmi_hip.h, rmi_hip_set_fit_mode): "exact" (0, default),
        "onepass_guarded" (1: one pass over the keys, error integers still bit-identical), "onepass" (2)."""
        m = {"exact": 0, "onepass_guarded": 1, "onepass": 2}.get(mode, mode)
        _check(self._lib.rmi_hip_set_fit_mode(self._h, int(m), float(guard_k)), self._h)

    def set_stream(self, stream_ptr: int | None):
        _check(self._lib.rmi_hip_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def fit_root(self, root: str | int, num_leaves: int, mode: str = "exact") -> Model:
        """mode="exact" (default): the reference's fit, bit for bit.  mode="fast": `linear` /
        `robust_linear` from parallel sums on the device (rmi_hip_fit_root_fast) -- the same line to
        ~1e-12, not bit-identical; the other kinds are exact either way."""
        kind = root if isinstance(root, int) else self._lib.rmi_hip_model_from_name(root.encode())
        if kind < 0:
            raise RMIError(kind)
        m = _lib.ModelParams()
        if mode == "fast" and kind in (0, 4):
            self.wait_keys()
            with self._ctx_lock:                    # device work on the context's stream
                _check(self._lib.rmi_hip_fit_root_fast(self._h, kind, num_leaves, C.byref(m)), self._h)
            return Model._from_c(m)
        if mode not in ("exact", "fast"):
            raise ValueError("mode must be 'exact' or 'fast'")
        hk = C.c_void_p(self._host_keys.ctypes.data) if self._host_keys is not None else None
        if kind in (0, 4) and hk is not None and getattr(self, "_uploading", False):
            # sequential host recurrence over the host copy: no device involved, runs beside the upload
            _check(self._lib.rmi_hip_fit_root_host(kind, _DTYPES[self._host_keys.dtype], hk, self.n, num_leaves, C.byref(m)))
            return Model._from_c(m)
        self.wait_keys()
        if kind in (2, 13):                         # cubic, bradix: device reductions / scans on the context's stream
            with self._ctx_lock:
                _check(self._lib.rmi_hip_fit_root(self._h, kind, num_leaves, hk, C.byref(m)), self._h)
            return Model._from_c(m)
        if not (8 <= kind <= 12):
            _check(self._lib.rmi_hip_fit_root(self._h, kind, num_leaves, hk, C.byref(m)), self._h)
            return Model._from_c(m)
        with self._ctx_lock:                        # a table root's fit writes the context's table
            _check(self._lib.rmi_hip_fit_root(self._h, kind, num_leaves, hk, C.byref(m)), self._h)
            root = Model._from_c(m)
            ent = C.c_uint64()
            _check(self._lib.rmi_hip_root_table_entries(self._h, C.byref(ent)))
            root.table = np.empty(int(ent.value), dtype=np.uint32)
            _check(self._lib.rmi_hip_download_root_table(self._h, root.table.ctypes.data))
            self._table_in_ctx = root.table
        return root

    def train_leaves(self, root: Model, leaf: str | int, num_leaves: int) -> TrainedRMI:
        leaf_kind = leaf if isinstance(leaf, int) else self._lib.rmi_hip_model_from_name(leaf.encode())
        if leaf_kind < 0:
            raise RMIError(leaf_kind)
        self.wait_keys()
        with self._ctx_lock:
            return self._train_leaves_locked(root, leaf_kind, num_leaves)

    def _train_leaves_locked(self, root: Model, leaf_kind: int, num_leaves: int) -> TrainedRMI:
        if root.is_radix_table and self._table_in_ctx is not root.table:
            if root.table is None:
                raise ValueError("a radix-table root needs its hint table (Model.table)")
            t = np.ascontiguousarray(root.table, dtype=np.uint32)
            _check(self._lib.rmi_hip_set_root_table(self._h, t.ctypes.data, t.size), self._h)
            self._table_in_ctx = root.table
        res = _lib.Result()
        rc = self._lib.rmi_hip_train_two_layer(self._h, C.byref(root._c()), leaf_kind, num_leaves, C.byref(res))
        _check(rc, self._h)
        return self._result(res, root, leaf_kind, num_leaves)

    def _result(self, res, root: Model, leaf_kind: int, num_leaves: int) -> TrainedRMI:
        return TrainedRMI(
            num_rmi_rows=int(res.num_rows), num_data_rows=int(res.num_rows),
            model_avg_error=res.model_avg_error, model_avg_l2_error=res.model_avg_l2_error,
            model_avg_log2_error=res.model_avg_log2_error, model_max_error=int(res.model_max_error),
            model_max_error_idx=int(res.model_max_error_idx), model_max_log2_error=res.model_max_log2_error,
            models=f"{root.name},{MODEL_NAMES[leaf_kind]}", branching_factor=int(num_leaves), root=root,
            leaf_kind=leaf_kind, params_per_leaf=int(res.params_per_leaf),
            device_ns=int(res.device_ns), kernel_ns=tuple(res.kernel_ns), long_leaves=int(res.long_leaves),
            split_idx=int(res.split_idx), split_target=int(res.split_target),
            shard_leaf_lo=int(res.shard_leaf_lo), shard_leaves=int(res.shard_leaves),
            partial={"max_error": int(res.model_max_error), "max_error_idx": int(res.model_max_error_idx),
                     "sum_n_err": int(res.sum_n_err), "sum_l2": float(res.sum_l2), "sum_log2": float(res.sum_log2)},
            fit_mode_used=int(res.fit_mode_used), exact_leaves=int(res.exact_leaves), merged_leaves=int(res.merged_leaves), guard_leaves=int(res.guard_leaves),
            generation=int(res.generation), pipeline=int(self._lib.rmi_hip_last_pipeline(self._h)), _trainer=self)

    def train_many(self, configs, in_flight: int = 4):
        """rmi_hip_train_many: `configs` = [(root Model, leaf kind or name, branching factor), ...] on the resident keys, in ONE call
        of the library (its own threads and contexts: what optimizer.rs:220-231 does with par_iter).  Returns a list of
        (return code, TrainedRMI with the aggregates only -- the per-leaf arrays of these trainings are not kept)."""
        self.wait_keys()
        n = len(configs)
        cfg = (_lib.TrainConfig * max(n, 1))()
        kinds, keep = [], []
        for i, (root, leaf, L) in enumerate(configs):
            lk = leaf if isinstance(leaf, int) else MODEL_NAMES.index(leaf)
            kinds.append(lk)
            cfg[i].root = root._c()
            cfg[i].leaf_kind = lk
            cfg[i].num_leaves = int(L)
            if root.is_radix_table:
                if root.table is None:
                    raise ValueError("a radix-table root needs its hint table (Model.table)")
                t = np.ascontiguousarray(root.table, dtype=np.uint32)
                keep.append(t)
                cfg[i].root_table = t.ctypes.data
                cfg[i].root_table_entries = t.size
        res = (_lib.Result * max(n, 1))()
        rcs = (C.c_int * max(n, 1))()
        with self._ctx_lock:                                      # (root fits that use the device run on this context too)
            rc = self._lib.rmi_hip_train_many(self._h, cfg, n, int(in_flight), res, rcs)
            self._table_in_ctx = None                             # (the context's table is whatever its last configuration set)
        if rc not in (0,) and all(int(r) == 0 for r in rcs[:n]):
            _check(rc, self._h)                                   # (the call itself failed, not a configuration)
        self.last_many_error = (self._lib.rmi_hip_last_error(self._h) or b"").decode() if rc else ""   # every failing configuration's message
        # ... and by configuration ("configuration <i>: <message>", joined by "; ")
        self.last_many_errors = {int(m.group(1)): m.group(2).strip() for m in
                                 re.finditer(r"configuration (\d+): (.*?)(?=; configuration \d+: |$)", self.last_many_error, re.S)}
        out = []
        for i, (root, _leaf, L) in enumerate(configs):
            if int(rcs[i]) != 0:
                out.append((int(rcs[i]), None))
                continue
            t = self._result(res[i], root, kinds[i], int(L))
            t._trainer = None
            t.pipeline = 0                                        # (rmi_hip_last_pipeline speaks of the caller's context only: not known per configuration)
            out.append((0, t))
        return out

    def release_views(self):
        """Frees the worker contexts train_many keeps between calls (each holds per-leaf buffers of the largest leaf count it trained)."""
        _check(self._lib.rmi_hip_release_views(self._h), self._h)

    def fit_root_host(self, keys: np.ndarray, root: str | int, num_leaves: int) -> Model:
        """The exact root fit from keys in HOST memory, no device involved (linear, robust_linear; linear_spline and radix
        through the same entry point): what train_streamed needs before the keys are uploaded."""
        kind = root if isinstance(root, int) else self._lib.rmi_hip_model_from_name(root.encode())
        if kind < 0:
            raise RMIError(kind)
        arr = np.ascontiguousarray(keys)
        m = _lib.ModelParams()
        _check(self._lib.rmi_hip_fit_root_host(kind, _DTYPES[arr.dtype], C.c_void_p(arr.ctypes.data), arr.size, num_leaves, C.byref(m)))
        return Model._from_c(m)

    def train_streamed(self, keys: np.ndarray, root: Model, leaf: str | int, num_leaves: int, chunks: int = 16) -> TrainedRMI:
        """Upload + train, overlapped (rmi_hip_train_streamed): the keys go to HBM in chunks through pinned staging
        buffers and every leaf-aligned shard is trained as soon as its keys have arrived.  Afterwards the keys are
        resident like after set_keys; the result is that of set_keys + train_leaves."""
        leaf_kind = leaf if isinstance(leaf, int) else self._lib.rmi_hip_model_from_name(leaf.encode())
        if leaf_kind < 0:
            raise RMIError(leaf_kind)
        arr = np.ascontiguousarray(keys)
        if arr.dtype not in _DTYPES:
            raise TypeError(f"unsupported key dtype {arr.dtype}")
        self.wait_keys()
        with self._ctx_lock:
            if root.is_radix_table and self._table_in_ctx is not root.table:
                if root.table is None:
                    raise ValueError("a radix-table root needs its hint table (Model.table)")
                t = np.ascontiguousarray(root.table, dtype=np.uint32)
                _check(self._lib.rmi_hip_set_root_table(self._h, t.ctypes.data, t.size), self._h)
                self._table_in_ctx = root.table
            res = _lib.Result()
            rc = self._lib.rmi_hip_train_streamed(self._h, C.c_void_p(arr.ctypes.data), arr.size, _DTYPES[arr.dtype], C.byref(root._c()),
                                                  leaf_kind, num_leaves, chunks, C.byref(res))
            _check(rc, self._h)
            self._host_keys = arr
            self._keepalive = None
            self.n = int(arr.size)
            return self._result(res, root, leaf_kind, num_leaves)

    def train_from_host(self, keys: np.ndarray, model_spec: str, branch_factor: int, chunks: int = 16) -> TrainedRMI:
        """rmi_lib::train for keys in host memory with the upload overlapped: the exact root fit on the host (no device),
        then train_streamed.  Root kinds whose fit needs the device (radix tables, bradix) and leaf counts that no power
        of two up to `chunks` divides take the plain path: set_keys + train."""
        t0 = time.perf_counter_ns()
        root_kind, leaf_kind = parse_spec(model_spec)
        c = 1
        while c * 2 <= chunks and branch_factor % (c * 2) == 0:
            c *= 2
        try:
            root = self.fit_root_host(keys, root_kind, branch_factor)
        except RMIError as e:
            if e.code != -11:                       # RMI_ERR_UNSUPPORTED_MODEL: a root the host alone cannot fit
                raise
            self.set_keys(keys)
            return self.train(model_spec, branch_factor)
        out = self.train_streamed(keys, root, leaf_kind, branch_factor, chunks=c)
        out.build_time = time.perf_counter_ns() - t0
        return out

    def train(self, model_spec: str, branch_factor: int, root_mode: str = "exact") -> TrainedRMI:
        """rmi_lib::train (train/mod.rs:100-126).  With a key set whose upload is still running (set_keys(wait=False)) the
        sequential host fit of a linear / robust_linear root runs beside the upload."""
        t0 = time.perf_counter_ns()
        root_kind, leaf_kind = parse_spec(model_spec)
        root = self.fit_root(root_kind, branch_factor, mode=root_mode)
        out = self.train_leaves(root, leaf_kind, branch_factor)
        out.build_time = time.perf_counter_ns() - t0
        return out

    def cache_fix(self, line_size: int) -> np.ndarray:
        """cache_fix.rs:109-150 over the (u64) keys of this trainer -> [m, 2] uint64 (key, offset)."""
        keys = self.download_keys()
        if keys.dtype != np.uint64:
            raise TypeError("Can only construct a bounded RMI on u64 data.")        # src/main.rs:281-282
        cnt = C.c_uint64()
        _check(self._lib.rmi_hip_cache_fix(self._h, keys.ctypes.data, keys.size, int(line_size), C.byref(cnt)), self._h)
        out = np.empty((int(cnt.value), 2), dtype=np.uint64)
        _check(self._lib.rmi_hip_download_cache_fix(self._h, out.ctypes.data), self._h)
        return out

    def train_bounded(self, model_spec: str, branch_factor: int, line_size: int, device: int | None = None) -> TrainedRMI:
        """rmi_lib::train_bounded (train/mod.rs:156-184): spline on the host, RMI over the re-indexed
        spline points on the device (a key set of its own; the returned object keeps its trainer)."""
        t0 = time.perf_counter_ns()
        spline = self.cache_fix(line_size)
        sub = Trainer(np.ascontiguousarray(spline[:, 0]), device=self._device if device is None else device)
        res = sub.train(model_spec, branch_factor)
        res.cache_fix = (int(line_size), spline)
        res.num_data_rows = self.n
        res.build_time = time.perf_counter_ns() - t0
        return res

    def _download(self, what: str, rmi: TrainedRMI):
        L, ppl = (rmi.shard_leaves or rmi.branching_factor), rmi.params_per_leaf
        code, a = {
            "params": (0, lambda: np.empty((L, ppl), dtype=np.float64)),
            "errors": (1, lambda: np.empty(L, dtype=np.uint64)),
            "counts": (2, lambda: np.empty(L, dtype=np.uint64)),
            "starts": (3, lambda: np.empty(L + 1, dtype=np.uint64)),
            "rows": (4, lambda: np.empty(L * (ppl * 8 + 8), dtype=np.uint8)),
        }[what]
        a = a()
        with self._ctx_lock:
            if self._h is None:
                raise RuntimeError("the trainer of this result has been closed: download the arrays first (materialize())")
            rc = self._lib.rmi_hip_download_checked(self._h, code, rmi.generation, a.ctypes.data, a.nbytes)
        if rc == -6:
            raise RuntimeError("this trainer has trained again since: the per-leaf arrays of the earlier result are gone "
                               "(call materialize() on a result before the next train call)")
        _check(rc, self._h)
        return a


def train(keys, model_spec: str, branch_factor: int, device: int = 0) -> TrainedRMI:
    """One-shot convenience mirror of ``rmi_lib::train(data, model_spec, branch_factor)``."""
    if isinstance(keys, np.ndarray):
        tr = Trainer(device=device)
        out = tr.train_from_host(keys, model_spec, branch_factor)
        out.materialize()
        tr.close()
        return out
    tr = Trainer(keys, device=device)
    out = tr.train(model_spec, branch_factor)
    out.materialize()                  # before the context goes away
    tr.close()
    return out
