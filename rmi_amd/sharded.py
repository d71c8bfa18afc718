"""Multi-GPU form of the leaf hot path: one process per GPU, leaf-aligned contiguous shards, one
all-gather of the packed leaf rows over RCCL (``torch.distributed`` backend ``nccl``).

Given the root parameters every leaf's container, fit and error bound depend only on a contiguous
key range plus a one/two-key halo and on global indices.  Rank r owns the leaves
``[r*L/G, (r+1)*L/G)`` and the keys the root maps to them; the cut points are found by binary
search with exactly the bucketing the kernels use (``rmi_hip_root_target``).  The G-GPU result is
byte-identical to the 1-GPU result.

The reference has no distributed path at all (one process, Rayon threads: two_layer.rs:161-169);
this module is the MI355X-side replacement of that 2-way join, not a translation of anything.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

from . import _lib
from . import train as T

NO_SPLIT = (1 << 64) - 1


@dataclass
class ShardPlan:
    n_global: int
    num_leaves: int
    rank: int
    world: int
    leaf_lo: int
    leaf_hi: int
    key_lo: int
    key_hi: int
    read_lo: int
    read_hi: int
    split_idx: int
    split_target: int

    def c_struct(self) -> _lib.Shard:
        s = _lib.Shard()
        s.n_global, s.read_lo, s.read_hi = self.n_global, self.read_lo, self.read_hi
        s.key_lo, s.key_hi, s.leaf_lo, s.leaf_hi = self.key_lo, self.key_hi, self.leaf_lo, self.leaf_hi
        s.split_idx, s.split_target = self.split_idx, self.split_target
        return s


def _key_bits(k, np_dtype) -> int:
    if np.dtype(np_dtype) == np.float64:
        return int(np.array([k], dtype=np.float64).view(np.uint64)[0])
    return int(k)


class Planner:
    """Host-side shard planning over any global key source ``key_at(i) -> key`` (a closed-form
    synthetic generator, or an mmap'd key file): O(G log N) probes, no pass over the data."""

    def __init__(self, key_at, n_global: int, np_dtype, root: T.Model, num_leaves: int):
        self.key_at = key_at
        self.n = int(n_global)
        self.np_dtype = np.dtype(np_dtype)
        self.dt = T._DTYPES[self.np_dtype]
        self.root = root
        self.L = int(num_leaves)
        self._lib = _lib.load()
        self._rootc = root._c()

    def target(self, i: int) -> int:
        if self.root.is_radix_table:                      # radix.rs:124-134, clamp two_layer.rs:49 (the table is host data)
            k = self.key_at(i)
            v = min(max(int(k), 0), (1 << 64) - 1)
            prefix, bits = int(self.root.ip[0]), int(self.root.ip[1])
            nb = 0 if prefix + bits > 64 else 64 - (prefix + bits)
            slot = (((v << (prefix & 63)) & ((1 << 64) - 1)) >> (prefix & 63)) >> (nb & 63)
            return min(int(self.root.table[slot]), self.L - 1)
        out = C.c_uint64()
        rc = self._lib.rmi_hip_root_target(C.byref(self._rootc), self.dt, _key_bits(self.key_at(i), self.np_dtype),
                                           self.L, C.byref(out))
        if rc:
            raise T.RMIError(rc)
        return int(out.value)

    def first_index_with_target_ge(self, leaf: int) -> int:
        """lower_bound over the (monotone) targets -- the search of two_layer.rs:132-136."""
        lo, hi = 0, self.n
        while lo < hi:
            mid = (lo + hi) // 2
            if self.target(mid) < leaf:
                lo = mid + 1
            else:
                hi = mid
        return lo

    def first_occurrence(self, i: int) -> int:
        v = self.key_at(i)
        if i == 0 or self.key_at(i - 1) != v:
            return i
        lo, hi = 0, i - 1
        while lo < hi:
            mid = (lo + hi) // 2
            if self.key_at(mid) < v:
                lo = mid + 1
            else:
                hi = mid
        return lo

    def plan(self, world: int):
        if self.L % world != 0:
            raise ValueError("the number of leaves must be divisible by the number of GPUs")
        per = self.L // world
        cuts = [0] + [self.first_index_with_target_ge(r * per) for r in range(1, world)] + [self.n]
        cuts[0] = 0
        split = self.first_index_with_target_ge(self.L // 2)
        if split >= self.n:
            split_idx, split_target = NO_SPLIT, 0
        else:
            split_idx, split_target = split, self.target(split)
        plans = []
        for r in range(world):
            key_lo, key_hi = cuts[r], cuts[r + 1]
            # left halo: the whole duplicate run of key[key_lo-1] plus one more key (prev-last point
            # with its first-occurrence offset; "was the previous key the split key" needs key_lo-2)
            if key_lo == 0:
                read_lo = 0
            else:
                read_lo = max(0, min(self.first_occurrence(key_lo - 1), key_lo - 1) - 1)
            # right halo: key[key_hi] (next-first point / upper lower-bound error) plus one spare
            read_hi = min(self.n, key_hi + 2)
            plans.append(ShardPlan(self.n, self.L, r, world, r * per, (r + 1) * per, key_lo, key_hi,
                                   read_lo, read_hi, split_idx, split_target))
        return plans


def run_shard(tr: T.Trainer, plan: ShardPlan, root: T.Model, leaf, rows_ptr: int | None = None):
    """Train the leaves of one shard on the trainer's device (keys [read_lo, read_hi) resident)."""
    lib = tr._lib
    sh = plan.c_struct()
    T._check(lib.rmi_hip_set_shard(tr._h, C.byref(sh)), tr._h)
    T._check(lib.rmi_hip_set_rows_output(tr._h, C.c_void_p(rows_ptr or 0)), tr._h)
    try:
        return tr.train_leaves(root, leaf, plan.num_leaves)
    finally:
        lib.rmi_hip_set_shard(tr._h, None)
        lib.rmi_hip_set_rows_output(tr._h, None)      # (a later unsharded training must not write into the caller's slot)


def combine_stats(parts, n_global: int) -> dict:
    """Aggregates of two_layer.rs:267-287 from per-shard partial sums (rmi_hip_result)."""
    mx, idx = 0, 0
    for p in parts:
        if p["max_error"] > mx or (p["max_error"] == mx and p["max_error_idx"] >= idx):
            mx, idx = p["max_error"], p["max_error_idx"]
    s_err = sum(p["sum_n_err"] for p in parts) & ((1 << 64) - 1)
    return {
        "model_max_error": mx, "model_max_error_idx": idx,
        "model_avg_error": s_err / n_global,
        "model_avg_l2_error": float(sum(p["sum_l2"] for p in parts)),
        "model_avg_log2_error": float(sum(p["sum_log2"] for p in parts)) / n_global,
    }


def exchange_rows(dist, full_rows, mine, rank: int, world: int):
    """One all-gather of the packed rows: every rank contributes `mine` (its L/G rows) and receives
    the full table in `full_rows` (flat uint8 tensors).  nccl == RCCL over xGMI; gloo on CPU tests."""
    per = full_rows.numel() // world
    assert mine.numel() == per
    if dist.get_backend() == "gloo":
        outs = [full_rows.new_empty(per) for _ in range(world)]
        dist.all_gather(outs, mine.contiguous())
        for r in range(world):
            full_rows[r * per:(r + 1) * per] = outs[r]
    else:
        dist.all_gather_into_tensor(full_rows, mine)
    return full_rows


class ShardedTrainer:
    """Driver of the N-GPU form used by bench.py: every rank generates its own shard of the global synthetic key
    array in HBM; rank 0 fits the root exactly (streamed) and broadcasts it; planning, the RCCL communicator and the
    all-gather of the rows are the library's (rmi_hip_plan_shards, rmi_hip_comm_*, rmi_hip_train_sharded) --
    torch.distributed only carries the 128-byte communicator id and the root parameters.  With the gloo backend
    (functional tests without RCCL) the rows are exchanged through torch instead."""

    def __init__(self, tr: T.Trainer, dist, rank: int, world: int, dataset: str, np_dtype,
                 n_global: int, num_leaves: int, spec: str, chunk: int = 50_000_000, fit_mode: int = 0, exchange: str = "rccl"):
        import torch
        self.tr, self.dist, self.rank, self.world = tr, dist, rank, world
        self.n_global, self.L = n_global, num_leaves
        root_kind, self.leaf_kind = T.parse_spec(spec)
        lib = tr._lib
        np_dtype = np.dtype(np_dtype)
        dt = T._DTYPES[np_dtype]
        gen_id = {"uniform": 0, "dups": 1}[dataset]
        t0 = time.perf_counter()
        self.on_gpu = dist.get_backend() != "gloo"
        dev = "cuda" if self.on_gpu else "cpu"
        # ---- root: exact fit on rank 0 (the linear recurrence is sequential: SURVEY.md section 7, H1), broadcast ----
        pbuf = torch.zeros(8, dtype=torch.float64, device=dev)
        if rank == 0:
            if root_kind == 0:
                rs = C.c_void_p()
                T._check(lib.rmi_hip_root_stream_begin(root_kind, dt, n_global, num_leaves, C.byref(rs)))
                gen = T.Trainer(device=torch.cuda.current_device())
                done = 0
                while done < n_global:
                    cnt = min(chunk, n_global - done)
                    gen.generate_keys(dataset, np_dtype, n_global, done, cnt)
                    host = gen.download_keys()
                    T._check(lib.rmi_hip_root_stream_push(rs, host.ctypes.data, cnt))
                    gen._host_keys = None
                    done += cnt
                gen.close()
                m = _lib.ModelParams()
                T._check(lib.rmi_hip_root_stream_finish(rs, C.byref(m)))
                root = T.Model._from_c(m)
            elif root_kind in (1, 3):               # linear_spline, radix: a handful of keys, through the key source
                kb0 = C.c_uint64()

                def key_src(user, i):
                    lib.rmi_hip_generated_key(gen_id, dt, n_global, 0, i, C.byref(kb0))
                    return kb0.value
                cb0 = _lib.KEY_AT_FN(key_src)
                m = _lib.ModelParams()
                T._check(lib.rmi_hip_fit_root_from_source(root_kind, dt, n_global, num_leaves, C.cast(cb0, C.c_void_p), None, C.byref(m)))
                root = T.Model._from_c(m)
            else:
                raise ValueError("the sharded driver fits linear, linear_spline and radix roots")
            pbuf.copy_(torch.tensor(list(root.p) + [float(v) for v in root.ip], dtype=torch.float64))
        dist.broadcast(pbuf, src=0)
        vals = pbuf.cpu().tolist()
        self.root = T.Model(root_kind, tuple(vals[:4]), tuple(int(v) for v in vals[4:8]))
        self.root_seconds = time.perf_counter() - t0
        # ---- plan (the library's planner over the generator's closed form) ----
        kb = C.c_uint64()

        def key_at(user, i):
            lib.rmi_hip_generated_key(gen_id, dt, n_global, 0, i, C.byref(kb))
            return kb.value
        self._cb = _lib.KEY_AT_FN(key_at)
        shards = (_lib.Shard * world)()
        T._check(lib.rmi_hip_plan_shards(None, C.byref(self.root._c()), dt, n_global, num_leaves, world,
                                         C.cast(self._cb, C.c_void_p), None, shards))
        sh = shards[rank]
        self.plan = ShardPlan(n_global, num_leaves, rank, world, int(sh.leaf_lo), int(sh.leaf_hi), int(sh.key_lo), int(sh.key_hi),
                              int(sh.read_lo), int(sh.read_hi), int(sh.split_idx), int(sh.split_target))
        tr.generate_keys(dataset, np_dtype, n_global, self.plan.read_lo, self.plan.read_hi - self.plan.read_lo)
        tr.set_fit_mode(fit_mode)
        self.row_bytes = 24 if self.leaf_kind != 2 else 40
        # ---- communicator: rank 0 makes the id, torch carries the 128 bytes ----
        self.exchange = "library (ncclAllGather in rmi_hip_train_sharded)"
        self.auto_report = None
        self._torch = torch

        def setup_direct():
            # peer stores over xGMI (rmi_hip_peer_export / _import): needs no RCCL; torch.distributed carries the handles
            hb = (C.c_ubyte * _lib.PEER_HANDLE_BYTES)()
            T._check(lib.rmi_hip_peer_export(tr._h, rank, world, self.leaf_kind, num_leaves, hb), tr._h)
            mine = torch.tensor(list(hb), dtype=torch.uint8, device=dev)
            allh = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allh, mine)
            for r in range(world):
                pb = (C.c_ubyte * _lib.PEER_HANDLE_BYTES)(*allh[r].cpu().tolist())
                T._check(lib.rmi_hip_peer_import(tr._h, r, pb), tr._h)

        if exchange == "direct":
            setup_direct()
            T._check(lib.rmi_hip_set_exchange(tr._h, 1), tr._h)
            self._shard_c = self.plan.c_struct()
            T._check(lib.rmi_hip_set_shard(tr._h, C.byref(self._shard_c)), tr._h)
            dist.barrier()
            self.on_gpu = True
            self.exchange = "library (direct peer stores into every rank's table + epoch flags: rmi_hip_peer_*)"
        elif self.on_gpu:
            # the library's own communicator; should RCCL not be loadable or the communicator not come up on some rank, EVERY
            # rank falls back to torch.distributed's all-gather of device tensors (the kernels are the same)
            ok = torch.ones(1, dtype=torch.int32, device=dev)
            idt = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
                if lib.rmi_hip_comm_unique_id(buf) != 0:
                    ok.zero_()
                idt.copy_(torch.tensor(list(buf), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                idb = (C.c_ubyte * _lib.COMM_ID_BYTES)(*idt.cpu().tolist())
                if lib.rmi_hip_comm_init(tr._h, rank, world, idb) != 0:
                    ok.zero_()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self._shard_c = self.plan.c_struct()
                T._check(lib.rmi_hip_set_shard(tr._h, C.byref(self._shard_c)), tr._h)
            else:
                lib.rmi_hip_comm_destroy(tr._h)
                self.on_gpu = False
                self.exchange = "torch.distributed all_gather_into_tensor (the library's communicator did not come up)"
        if exchange == "auto" and self.on_gpu and world > 1:
            # A/B on the machine itself: the direct exchange is taken only if it completes, gives the SAME table as the
            # RCCL exchange on every rank, and is faster (max over ranks of the median step)
            rep = {"rccl_ms": None, "direct_ms": None, "same_table": None, "chosen": "rccl"}

            def all_ok(ok: bool) -> bool:
                """Every rank reports; every rank takes the same branch (a rank that failed locally must not leave the others
                waiting in the next collective)."""
                v = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(v, op=dist.ReduceOp.MIN)
                return bool(int(v.item()))

            def med(k=12):
                """Median step time (max over the ranks).  A step that raises on this rank is remembered, the collectives go on."""
                ts, ok = [], True
                for _ in range(k):
                    dist.barrier(); torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    try:
                        self.step(); torch.cuda.synchronize()
                    except Exception as ex:
                        ok = False
                        rep["error"] = str(ex)
                    ts.append(time.perf_counter() - t1)
                    if not all_ok(ok):                           # (outside the timed window; a failing exchange costs ONE peer time-out, not k)
                        ok = False
                        break
                v = torch.tensor([sorted(ts)[len(ts) // 2]], dtype=torch.float64, device=dev)
                dist.all_reduce(v, op=dist.ReduceOp.MAX)
                return float(v.item()), all_ok(ok)

            ok = True
            try:
                setup_direct()
            except Exception as ex:
                ok = False
                rep["error"] = str(ex)
            if all_ok(ok):
                t_rccl, ok_r = med()
                rep["rccl_ms"] = t_rccl * 1e3
                ref_rows = self.full_rows().copy() if ok_r else None
                ok_s = lib.rmi_hip_set_exchange(tr._h, 1) == 0
                if all_ok(ok_r and ok_s):
                    t_dir, ok_d = med()
                    rep["direct_ms"] = t_dir * 1e3
                    same = ok_d and bool(np.array_equal(ref_rows, self.full_rows()))
                    rep["same_table"] = all_ok(same)
                    # the choice is rank 0's, sent to all (the medians are maxima over the ranks already: same on every rank)
                    pick = torch.tensor([1 if (rep["same_table"] and rep["direct_ms"] < rep["rccl_ms"]) else 0], dtype=torch.int32, device=dev)
                    dist.broadcast(pick, src=0)
                    if int(pick.item()) == 1:
                        rep["chosen"] = "direct"
                        self.exchange = "library (direct peer stores: chosen by the A/B at start-up)"
                if rep["chosen"] != "direct":
                    lib.rmi_hip_set_exchange(tr._h, 0)
            else:
                lib.rmi_hip_set_exchange(tr._h, 0)
            self.auto_report = rep
        if not self.on_gpu:
            if dist.get_backend() == "gloo":
                self.exchange = "torch.distributed all_gather through host memory (gloo: functional runs only)"
            per = (self.plan.leaf_hi - self.plan.leaf_lo) * self.row_bytes
            self._full = torch.empty(num_leaves * self.row_bytes, dtype=torch.uint8, device="cuda")
            self._mine = torch.empty(per, dtype=torch.uint8, device="cuda")
            self._host = (torch.empty(num_leaves * self.row_bytes, dtype=torch.uint8), torch.empty(per, dtype=torch.uint8))

    def step(self):
        """One training: when it returns, every rank holds the full row table (the step of SURVEY 8d for N > 1)."""
        if self.on_gpu:
            res = _lib.Result()
            with self.tr._ctx_lock:
                rc = self.tr._lib.rmi_hip_train_sharded(self.tr._h, C.byref(self.root._c()), self.leaf_kind, self.L, C.byref(res))
            T._check(rc, self.tr._h)
            return res
        res = run_shard(self.tr, self.plan, self.root, self.leaf_kind, self._mine.data_ptr())
        if self.dist.get_backend() != "gloo":                  # device tensors straight through torch's RCCL
            self._torch.cuda.synchronize()
            exchange_rows(self.dist, self._full, self._mine, self.rank, self.world)
            return res
        full_h, mine_h = self._host
        mine_h.copy_(self._mine)
        exchange_rows(self.dist, full_h, mine_h, self.rank, self.world)
        self._full.copy_(full_h)
        return res

    def full_rows(self) -> np.ndarray:
        if self.on_gpu:
            out = np.empty(self.L * self.row_bytes, dtype=np.uint8)
            T._check(self.tr._lib.rmi_hip_download_rows_full(self.tr._h, out.ctypes.data, out.nbytes), self.tr._h)
            return out
        return self._full.cpu().numpy()

    def finish(self):
        self._torch.cuda.synchronize()
