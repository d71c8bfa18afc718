"""Multi-GPU row on one GPU: the shards of a G-way split are run one after the
other on the same device, each with only its own keys + halo resident, and the concatenation must
be byte-identical to the unsharded result (and therefore to the oracle)."""
import os

import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu


def _run_sharded(train, sharded, keys, root_name, leaf, L, G):
    tr = train.Trainer(keys)
    root = tr.fit_root(root_name, L)
    full = tr.train_leaves(root, leaf, L)
    full_rows = full.rows.copy()
    full_params, full_err = full.leaf_params.copy(), full.last_layer_max_l1s.copy()
    full_counts, full_starts = full.leaf_counts.copy(), full.leaf_starts.copy()
    tr.close()
    plans = sharded.Planner(lambda i: keys[i], len(keys), keys.dtype, root, L).plan(G)
    rows, params, errs, counts, starts, parts = [], [], [], [], [], []
    for pl in plans:
        assert pl.key_lo == full_starts[pl.leaf_lo] and pl.key_hi == full_starts[pl.leaf_hi]
        t = train.Trainer(np.ascontiguousarray(keys[pl.read_lo:pl.read_hi]))
        res = sharded.run_shard(t, pl, root, leaf)
        rows.append(res.rows.copy()); params.append(res.leaf_params.copy()); errs.append(res.last_layer_max_l1s.copy())
        counts.append(res.leaf_counts.copy()); starts.append(res.leaf_starts[:-1].copy()); parts.append(res.partial)
        t.close()
    assert np.array_equal(np.concatenate(starts), full_starts[:-1])
    assert np.array_equal(np.concatenate(params), full_params)
    assert np.array_equal(np.concatenate(errs), full_err)
    assert np.array_equal(np.concatenate(counts), full_counts)
    assert np.array_equal(np.concatenate(rows), full_rows), "sharded rows differ from the 1-GPU rows"
    st = sharded.combine_stats(parts, len(keys))
    assert st["model_max_error"] == full.model_max_error
    assert st["model_max_error_idx"] == full.model_max_error_idx
    assert st["model_avg_error"] == full.model_avg_error
    assert abs(st["model_avg_log2_error"] - full.model_avg_log2_error) <= 1e-9 * max(1.0, abs(full.model_avg_log2_error))
    return full


@pytest.mark.parametrize("pipeline", ["3", "2"])
@pytest.mark.parametrize("gen", ["uniform_u64", "books_u64", "dups_u64", "dups_u32"])
@pytest.mark.parametrize("G", [2, 8])
def test_sharded_equals_single(monkeypatch, pipeline, gen, G):
    monkeypatch.setenv("RMI_HIP_PIPELINE", pipeline)
    from rmi_amd import train, sharded
    keys = dg.GENERATORS[gen](200_000)
    _run_sharded(train, sharded, keys, "linear", "linear", 4096, G)
    _run_sharded(train, sharded, keys, "radix", "linear_spline", 8192, G)


def test_sharded_tiny_leaves_split_at_cut():
    """More leaves than keys per wave row, and the 2-way-join split exactly on a shard cut."""
    from rmi_amd import train, sharded
    keys = dg.uniform_u64(300_000)
    _run_sharded(train, sharded, keys, "linear", "linear", 1 << 17, 2)
    _run_sharded(train, sharded, keys, "linear", "linear", 1 << 17, 4)
    _run_sharded(train, sharded, dg.dups_u64(300_000), "linear", "linear", 1 << 16, 8)


def test_sharded_long_leaves():
    """Leaves longer than pass A's table (k_fit_long) inside shards, incl. a long leaf at a shard cut."""
    from rmi_amd import train, sharded
    for gen in ("dups_u64", "books_u64"):
        keys = dg.GENERATORS[gen](300_000)
        _run_sharded(train, sharded, keys, "linear", "linear", 16, 2)
        _run_sharded(train, sharded, keys, "linear", "linear", 64, 8)
        _run_sharded(train, sharded, keys, "linear", "linear_spline", 32, 4)


def test_sharded_radix_table_root():
    """Radix-table root: the hint table travels with the root (rmi_hip_set_root_table per shard context)."""
    from rmi_amd import train, sharded
    keys = dg.dups_u64(200_000)
    _run_sharded(train, sharded, keys, "radix18", "linear", 4096, 4)
    _run_sharded(train, sharded, dg.dups_u32(200_000), "radix8", "linear_spline", 128, 2)


def test_sharded_cubic_leaves():
    from rmi_amd import train, sharded
    keys = dg.dups_u64(200_000)
    _run_sharded(train, sharded, keys, "linear", "cubic", 4096, 4)
    _run_sharded(train, sharded, keys, "cubic", "cubic", 512, 2)


def test_sharded_matches_oracle(oracle):
    from rmi_amd import train, sharded
    keys = dg.books_u64(150_000)
    full = _run_sharded(train, sharded, keys, "linear", "linear", 1024, 4)
    o = oracle.train_two_layer("linear", "linear", keys, 1024)
    assert np.array_equal(full.leaf_params, o.leaf_params)
    assert np.array_equal(full.last_layer_max_l1s, o.leaf_err)


def _sharded_trainer_worker(rank, world, port, n_global, L, q, exchange="rccl", listed_rank=-1):
    import os
    import torch
    import torch.distributed as dist
    from rmi_amd import sharded, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("RMI_HIP_PEER_TIMEOUT_S", "400")        # (processes that share ONE GPU on a slow box: the library's default is 60 s)
    if rank == listed_rank:
        os.environ["RMI_HIP_LONG_MIN"] = "64"                     # this rank hands (nearly) every leaf to the list kernels
    torch.cuda.set_device(0)                                      # both ranks share the one GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = train.Trainer(device=0)
    sh = sharded.ShardedTrainer(tr, dist, rank, world, "uniform", np.uint64, n_global, L, "linear,linear", chunk=700_000, exchange=exchange)
    sh.step()
    res = sh.step()                                               # (a second step re-uses every buffer)
    if exchange == "direct":
        res = sh.step()                                           # (direct exchange: both halves of the double-buffered table)
        dist.barrier()
    rows = sh.full_rows().copy()
    ok = True
    if listed_rank >= 0:                                          # (the rank concerned really took the list kernels, the other none)
        ok = (int(res.long_leaves) > 1000) == (rank == listed_rank)
    if rank == 0 or exchange == "direct":                         # against the unsharded result on the same keys
        t1 = train.Trainer(device=0)
        t1.generate_keys("uniform", np.uint64, n_global)
        ref = t1.train("linear,linear", L)
        ok = ok and bool(np.array_equal(rows, ref.rows)) and ref.root.p == sh.root.p
        t1.close()
    q.put((rank, ok))
    tr.close()
    dist.destroy_process_group()


def test_sharded_trainer_two_ranks_one_gpu():
    """bench.py's N>1 driver (ShardedTrainer: streamed exact root on rank 0, broadcast, plan, device-side
    key generation per shard, per-step exchange) with two gloo ranks on one GPU: every rank ends with
    the byte-identical table of the unsharded run."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_trainer_worker, args=(r, 2, port, 2_000_000, 8192, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_sharded_direct_exchange_two_ranks_one_gpu():
    """The exchange as peer stores (rmi_hip_peer_export / _import, RMI_EXCHANGE_DIRECT): two processes on the one GPU of the
    box map each other's row table and mailbox through IPC handles, every rank stores its slice into the other's table,
    publishes aggregates + epoch flag, waits for the peers' flags.  Every rank ends with the byte-identical table of the
    unsharded run (three steps: both halves of the double buffer).  Across devices this path has not run yet."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_trainer_worker, args=(r, 2, port, 2_000_000, 8192, q, "direct")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_sharded_direct_exchange_listed_leaves_on_one_rank():
    """A training publishes its result without the list kernels; if SOME rank handed leaves to them, every rank reads it
    from the exchanged records, the rank concerned finishes its leaves and all exchange once more (the next epoch of the
    peer-store tables).  Rank 1 lists nearly every leaf here, rank 0 none: same table as the unsharded run on both."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_trainer_worker, args=(r, 2, port, 2_000_000, 8192, q, "direct", 1)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _rccl_worker(rank, world, port, n_global, L, mode, q, exchange="rccl", listed_rank=-1):
    """The N>1 path through the C ABI only: rmi_hip_plan_shards, rmi_hip_comm_init, rmi_hip_train_sharded (kernels +
    ncclAllGather on the library's stream).  torch.distributed carries the communicator id and the root parameters."""
    import os
    import torch
    import torch.distributed as dist
    from rmi_amd import sharded, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if rank == listed_rank:
        os.environ["RMI_HIP_LONG_MIN"] = "64"                     # this rank hands (nearly) every leaf to the list kernels
    if world > 1:
        os.environ.setdefault("NCCL_DEBUG", "INFO")               # (the first multi-GPU run logs which algorithm RCCL picks)
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL")
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    tr = train.Trainer(device=dev)
    sh = sharded.ShardedTrainer(tr, dist, rank, world, "uniform", np.uint64, n_global, L, "linear,linear", chunk=700_000, fit_mode=mode, exchange=exchange)
    import ctypes as C
    cw, cr = C.c_int(), C.c_int()
    assert tr._lib.rmi_hip_comm_info(tr._h, C.byref(cw), C.byref(cr)) == 0
    info_ok = (cw.value, cr.value) == (world, rank) or exchange == "direct"     # the communicator itself counts `world` ranks
    if sh.auto_report is not None:
        print(f"[rank {rank}] exchange A/B: {sh.auto_report}", flush=True)
    res = None
    for _ in range(3):                                            # (later steps re-use every buffer)
        res = sh.step()
    rows = sh.full_rows().copy()
    t1 = train.Trainer(device=dev)
    t1.generate_keys("uniform", np.uint64, n_global)
    ref = t1.train("linear,linear", L)
    ref_rows = ref.rows.view(np.uint64).reshape(L, 3)
    got = rows.view(np.uint64).reshape(L, 3)
    ok = info_ok and bool(np.array_equal(got[:, 2], ref_rows[:, 2])) and ref.root.p == sh.root.p        # error integers: always
    if listed_rank >= 0:
        ok = ok and (int(res.long_leaves) > 1000) == (rank == listed_rank)
    if mode == 0:
        ok = ok and bool(np.array_equal(got, ref_rows))                                     # exact mode: byte-identical table
    ok = ok and int(res.model_max_error) == ref.model_max_error and int(res.model_max_error_idx) == ref.model_max_error_idx
    ok = ok and float(res.model_avg_error) == ref.model_avg_error
    ok = ok and abs(float(res.model_avg_log2_error) - ref.model_avg_log2_error) <= 1e-9 * abs(ref.model_avg_log2_error)
    q.put((rank, bool(ok)))
    t1.close(); tr.close()
    dist.destroy_process_group()


def _spawn(world, n_global, L, mode, exchange="rccl", listed_rank=-1):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, n_global, L, mode, q, exchange, listed_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


@pytest.mark.parametrize("mode", [0, 1])
def test_train_sharded_rccl_single_rank(mode):
    """The library's own RCCL path (communicator from a real unique id, in-place ncclAllGather of the rows and of the
    aggregates on the context's stream) with the one rank a 1-GPU box allows."""
    assert _spawn(1, 3_000_000, 4096, mode) == [(0, True)]


def test_train_sharded_rccl_single_rank_listed_leaves():
    """... and with leaves handed to the list kernels: the result is published without them, the rank reads its own `pending`
    from the gathered records, runs them and gathers once more."""
    assert _spawn(1, 3_000_000, 4096, 0, listed_rank=0) == [(0, True)]


def test_sharded_trainer_falls_back_to_torch_all_gather(monkeypatch):
    """Without RCCL for the library (RMI_HIP_NO_RCCL) every rank takes torch.distributed's all-gather of device tensors:
    the same table."""
    monkeypatch.setenv("RMI_HIP_NO_RCCL", "1")
    assert _spawn(1, 3_000_000, 4096, 0) == [(0, True)]


@pytest.mark.parametrize("mode", [0, 1])
def test_train_sharded_rccl_two_ranks(mode):
    """Two processes, two GPUs, RCCL over xGMI: runs wherever the box has them (the first multi-GPU box proves that the
    collective saw N ranks); every rank ends with the table of the unsharded run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    assert _spawn(2, 6_000_000, 8192, mode) == [(0, True), (1, True)]


@pytest.mark.parametrize("exchange", ["direct", "auto"])
def test_train_sharded_peer_stores_two_gpus(exchange):
    """The direct exchange across two DEVICES (stores over xGMI into the peer's table, system-scope flags), and the A/B
    against ncclAllGather that bench.py --exchange auto runs: wherever the box has two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    assert _spawn(2, 6_000_000, 8192, 0, exchange) == [(0, True), (1, True)]


def test_upload_overlaps_the_host_root_fit(oracle):
    """set_keys(wait=False): the copy runs on a thread of the library while the host fits the root; same result."""
    from rmi_amd import train
    keys = dg.uniform_u64(3_000_000)
    tr = train.Trainer()
    tr.set_keys(keys, wait=False)
    g = tr.train("linear,linear", 4096).materialize()
    o = oracle.train_two_layer("linear", "linear", keys, 4096)
    assert g.root.p == o.root.p and np.array_equal(g.leaf_params, o.leaf_params) and np.array_equal(g.last_layer_max_l1s, o.leaf_err)
    tr.close()


def _run_ranks(world, args_of, timeout=600):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_trainer_worker, args=args_of(r, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


@pytest.mark.parametrize("exchange", ["rccl", "direct"])
def test_sharded_trainer_eight_ranks_one_gpu(exchange):
    """Eight ranks (eight processes on the one GPU of the box, gloo for the rendezvous): the shard planner at G = 8, and with the
    direct exchange the rows of every rank stored to SEVEN peers' IPC-mapped tables from the kernels themselves (PeerRows),
    the 8-flag mailbox wait, three steps through both halves of the double buffer.  Every rank ends with the byte-identical
    table of the unsharded run.  (Across eight devices this has not run: no such box was available.)"""
    res = _run_ranks(8, lambda r, port, q: (r, 8, port, 4_000_000, 32768, q, exchange))
    assert res == [(r, True) for r in range(8)]


def test_sharded_direct_exchange_eight_ranks_one_lists():
    """The `pending` protocol at eight ranks: rank 5 hands nearly every leaf to the list kernels, the other seven none; every rank
    reads it from the exchanged records, rank 5 finishes its leaves, all exchange once more.  Same table on all eight."""
    res = _run_ranks(8, lambda r, port, q: (r, 8, port, 4_000_000, 32768, q, "direct", 5))
    assert res == [(r, True) for r in range(8)]


def test_bench_eight_ranks_gloo_direct(tmp_path):
    """bench.py as the driver launches it for N = 8 (torch.distributed.run, one rank per process) -- on one GPU, gloo for
    torch.distributed, the direct exchange for the rows: one JSON line with eight per_rank entries."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "2", "--backend", "gloo", "--exchange", "direct",
           "--keys", "8000000", "--leaves", "65536", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=dict(os.environ, RMI_HIP_PEER_TIMEOUT_S="400"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 8 and len(d["per_rank"]) == 8 and d["value"] > 0
    assert sum(p["leaves"] for p in d["per_rank"]) == 65536 and sum(p["keys"] for p in d["per_rank"]) >= 8000000
