"""N>1 path on CPU (gloo, world_size 2): shard planning with the library's host-side bucketing,
the row exchange, and the statistics combination.  The per-shard compute needs a GPU, so here each
rank fills its slice of the row buffer from the oracle's result for its own leaf range; what is
under test is everything around it: cuts, halos, the all-gather and the assembled buffer."""
import os
import socket

import numpy as np
import pytest

from rmi_amd import datagen as dg


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, L, q):
    import torch
    import torch.distributed as dist
    from oracle import binding as oracle
    from rmi_amd import sharded, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys = dg.books_u64(120_000)
    o = oracle.train_two_layer("linear", "linear", keys, L)
    root = train.Model(0, o.root.p, o.root.ip)
    plan = sharded.Planner(lambda i: keys[i], len(keys), keys.dtype, root, L).plan(world)[rank]
    # cuts == the oracle's bucket boundaries
    assert plan.key_lo == int(o.leaf_start[plan.leaf_lo]) and plan.key_hi == int(o.leaf_start[plan.leaf_hi])
    assert plan.read_lo <= max(plan.key_lo - 2, 0) and plan.read_hi >= min(plan.key_hi + 1, len(keys))
    ref_rows = np.empty((L, 3), dtype=np.uint64)
    ref_rows[:, :2] = o.leaf_params.view(np.uint64)
    ref_rows[:, 2] = o.leaf_err
    ref_bytes = ref_rows.view(np.uint8).reshape(-1)
    full = torch.zeros(L * 24, dtype=torch.uint8)
    lo, hi = plan.leaf_lo * 24, plan.leaf_hi * 24
    mine = torch.from_numpy(ref_bytes[lo:hi].copy())              # stand-in for this rank's device rows
    sharded.exchange_rows(dist, full, mine, rank, world)
    ok = bool(np.array_equal(full.numpy(), ref_bytes))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_exchange_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2048, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_planner_cuts_and_split(oracle):
    from rmi_amd import sharded, train
    for gen in ("uniform_u64", "dups_u64", "uniform_u32"):
        keys = dg.GENERATORS[gen](50_000)
        L = 1024
        o = oracle.train_two_layer("linear", "linear", keys, L)
        root = train.Model(0, o.root.p, o.root.ip)
        pl = sharded.Planner(lambda i: keys[i], len(keys), keys.dtype, root, L)
        ids = oracle.bucket_ids(o.root, keys, L)
        for i in (0, 1, 777, len(keys) - 1):
            assert pl.target(i) == int(ids[i])             # host bucketing == oracle bucketing
        plans = pl.plan(8)
        assert plans[0].key_lo == 0 and plans[-1].key_hi == len(keys)
        for p in plans:
            assert p.key_lo == int(o.leaf_start[p.leaf_lo])
        split = int(np.searchsorted(ids, L // 2, side="left"))
        assert plans[0].split_idx == split and plans[0].split_target == int(ids[split])


def test_combine_stats():
    from rmi_amd import sharded
    parts = [{"max_error": 5, "max_error_idx": 3, "sum_n_err": 10, "sum_l2": 1.0, "sum_log2": 2.0},
             {"max_error": 5, "max_error_idx": 9, "sum_n_err": 20, "sum_l2": 2.0, "sum_log2": 4.0}]
    st = sharded.combine_stats(parts, 10)
    assert st["model_max_error"] == 5 and st["model_max_error_idx"] == 9    # last maximum wins (max_by_key)
    assert st["model_avg_error"] == 3.0 and st["model_avg_l2_error"] == 3.0 and st["model_avg_log2_error"] == 0.6


def test_c_planner_matches_python(oracle):
    """rmi_hip_plan_shards (the planner behind the C ABI) against the Python planner it replaces, with a host key
    array as the key source and with the closed form of the synthetic generators (rmi_hip_generated_key)."""
    import ctypes as C
    from rmi_amd import _lib, sharded, train
    lib = _lib.load()
    for gen, gid, dt in (("uniform_u64", 0, 0), ("dups_u64", 1, 0), ("uniform_u32", 0, 1), ("dups_u32", 1, 1), ("books_u64", None, 0)):
        keys = dg.GENERATORS[gen](60_000)
        n, L = len(keys), 2048
        o = oracle.train_two_layer("linear", "linear", keys, L)
        root = train.Model(0, o.root.p, o.root.ip)
        if gid is not None:                                  # the closed form reproduces the generator
            kb = C.c_uint64()
            for i in (0, 1, 7, 8, 4097, n - 1):
                assert lib.rmi_hip_generated_key(gid, dt, n, 0, i, C.byref(kb)) == 0 and kb.value == int(keys[i])
        cb = _lib.KEY_AT_FN(lambda user, i: int(keys[i]))
        for world in (1, 2, 8):
            want = sharded.Planner(lambda i: keys[i], n, keys.dtype, root, L).plan(world)
            got = (_lib.Shard * world)()
            rc = lib.rmi_hip_plan_shards(None, C.byref(root._c()), dt, n, L, world, C.cast(cb, C.c_void_p), None, got)
            assert rc == 0
            for w, g in zip(want, got):
                assert (g.n_global, g.read_lo, g.read_hi, g.key_lo, g.key_hi, g.leaf_lo, g.leaf_hi, g.split_idx, g.split_target) == \
                       (w.n_global, w.read_lo, w.read_hi, w.key_lo, w.key_hi, w.leaf_lo, w.leaf_hi, w.split_idx, w.split_target)
        bad = (_lib.Shard * 3)()
        assert lib.rmi_hip_plan_shards(None, C.byref(root._c()), dt, n, L, 3, C.cast(cb, C.c_void_p), None, bad) == -6   # L % world != 0


def test_comm_entry_points_without_a_gpu():
    """No GPU here: the communicator calls must fail with a code, never crash (RCCL is bound at run time)."""
    import ctypes as C
    from rmi_amd import _lib
    lib = _lib.load()
    buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
    assert lib.rmi_hip_comm_unique_id(buf) in (0, -16, -17)
    assert lib.rmi_hip_comm_unique_id(None) == -6
    assert lib.rmi_hip_comm_init(None, 0, 1, None) == -6
