#!/usr/bin/env python3
"""bench.py -- keys/s of the two-layer leaf hot path on MI355X, one JSON line on rank 0.

A "step" is one pass of the hot path (rmi_hip_train_two_layer: bucketing scan, per-leaf fits,
per-leaf max-error pass, lower-bound widening, row packing) over one batch of synthetic sorted
keys that is already resident in HBM, with the root-model parameters given.

N=1 workload = the configuration BASELINE.json's metric is quoted on:
    200M uniform uint64 keys, linear,linear, 2^20 leaves.
N>1: weak scaling -- every rank holds `--keys` keys (a contiguous range of the global sorted
array) and its own 2^20-leaf shard of the model; rows are exchanged with one RCCL all-gather.

Launch (N>1): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
              --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
KERNEL_NAMES = ["k_fit_stream", "k_fill", "k_fit_long", "k_err_range", "k_finalize+stats"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (a step is ~1 ms: 200 steps let the clocks settle, the first ~20 run 5-10 %% slower)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--keys", type=int, default=200_000_000, help="keys per GPU")
    ap.add_argument("--leaves", type=int, default=1 << 20, help="leaves per GPU")
    ap.add_argument("--spec", default="linear,linear")
    ap.add_argument("--dataset", default="uniform", choices=["uniform", "dups", "books"],
                    help="uniform / dups are generated in HBM; books (heavy-tailed, books_200M-shaped) on the host")
    ap.add_argument("--dtype", default="uint64", choices=["uint64", "uint32"])
    ap.add_argument("--cpu-sample", type=int, default=200_000_000,
                    help="keys of the workload the CPU baseline is timed on (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N>1: finish the row exchange of a step before the next step starts (default: it overlaps the next step's kernels)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo for functional tests)")
    return ap.parse_args()


def cpu_baseline(keys_np, spec, leaves_total, n_total):
    """The oracle (C restatement of the reference's CPU path, 2 threads like rayon::join) timed on
    a bounded prefix of the same workload.  Reported baseline, not the optimisation target."""
    from oracle import binding as oracle
    oracle.build()
    n = len(keys_np)
    L = max(2, int(round(leaves_total * (n / n_total))))   # same keys-per-leaf as the GPU workload
    root, leaf = spec.split(",")
    t0 = time.perf_counter()
    oracle.train_two_layer(root, leaf, keys_np, L, threads=2)
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "keys/s", "cores": 2, "kind": "port",
        "sample": f"first {n} keys of the workload, {spec}, {L} leaves (same keys/leaf); "
                  f"C restatement of the reference CPU path incl. root fit, 2 threads (rayon::join), "
                  f"{dt:.2f} s wall; host has {os.cpu_count()} cores",
    }


def main():
    args = parse_args()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    local_rank = local_rank % max(1, torch.cuda.device_count())   # (functional tests may oversubscribe one GPU with gloo)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    from rmi_amd import train as T

    n_local = args.keys
    n_global = n_local * world
    L_local = args.leaves
    L_global = L_local * world
    np_dtype = np.uint64 if args.dtype == "uint64" else np.uint32
    key_bytes = np.dtype(np_dtype).itemsize

    tr = T.Trainer(device=local_rank)
    sh = None
    if world == 1:
        if args.dataset == "books":
            from rmi_amd import datagen
            tr.set_keys(datagen.books_u64(n_local))
        else:
            tr.generate_keys(args.dataset, np_dtype, n_global, 0, n_local)
        root_kind, leaf_kind = T.parse_spec(args.spec)
        t0 = time.perf_counter()
        keys_np = tr.download_keys()
        root = tr.fit_root(root_kind, L_global)        # exact (reference-order) root fit on the host
        root_s = time.perf_counter() - t0
        run_step = lambda: tr.train_leaves(root, leaf_kind, L_local)
    else:
        from rmi_amd import sharded
        sh = sharded.ShardedTrainer(tr, dist, rank, world, args.dataset, np_dtype, n_global, L_global, args.spec,
                                    pipeline=not args.no_pipeline)
        root_s = sh.root_seconds
        run_step = sh.step
        keys_np = None

    def sync():
        if sh is not None:
            sh.finish()                 # the exchange of the last step (pipelined mode) belongs to the timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # hipEvents on the library's stream.  An event between two kernels costs ~5.5 us of idle device
    # time, so the timed steps bracket only the dominant (first) kernel of the path -- the one the
    # roofline is quoted for -- and the full per-kernel breakdown is taken in the warm-up steps.
    tr.set_profile_level(2)
    warm_ns = np.zeros(8, dtype=np.float64)
    for _ in range(args.warmup):
        res = run_step()
        warm_ns += np.array(res.kernel_ns, dtype=np.float64)
    sync()
    tr.set_profile_level(1)
    dom_ns = 0.0
    device_ns = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = run_step()
        dom_ns += res.kernel_ns[0]
        device_ns += res.device_ns
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ns = warm_ns / max(args.warmup, 1) * args.steps      # breakdown from the warm-up steps ...
    kernel_ns[0] = dom_ns                                       # ... the dominant kernel live over the timed region
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_global / (elapsed / args.steps)
        kernel_us = (kernel_ns / args.steps / 1e3)[:5]
        dom = int(np.argmax(kernel_us))
        # algorithmic bytes per launch: one read of the keys + one write of the rows
        b_leaf = n_local * key_bytes + 24 * L_local
        dom_s = kernel_us[dom] * 1e-6
        achieved = b_leaf / dom_s / 1e9 if dom_s > 0 else 0.0
        dev_s = device_ns / args.steps * 1e-9
        pipeline_gbs = b_leaf / dev_s / 1e9 if dev_s > 0 else 0.0
        out = {
            "metric": "keys/s trained (200M uint64, linear,linear 2^20)",
            "value": value, "unit": "keys/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.spec} {L_global} leaves on {n_global} synthetic sorted {args.dtype} keys "
                            f"({args.dataset}), {n_local} keys + {L_local} leaves per GPU",
                "keys_per_gpu": n_local, "leaves_per_gpu": L_local, "mode": "exact (reference-order SLR)",
                "exchange": (None if world == 1 else ("all-gather of rows, overlapped with the next step's kernels (double-buffered)"
                                                      if (sh is not None and sh.pipeline) else "all-gather of rows at the end of every step")),
                "root_fit_seconds_untimed": root_s,
            },
            "roofline": {
                "bound": "hbm", "kernel": KERNEL_NAMES[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes": b_leaf, "kernel_us": {k: float(v) for k, v in zip(KERNEL_NAMES, kernel_us)},
                "kernel_us_note": "the dominant (first) kernel: hipEvents over the timed steps; the others: hipEvents over the warm-up steps "
                                  "(an event between two kernels idles the device ~5.5 us, so the timed steps carry only the two that bracket the dominant kernel)",

                "pipeline_device_us": dev_s * 1e6, "pipeline_achieved": pipeline_gbs,
                "pipeline_frac": pipeline_gbs / HBM_PEAK_GBS,
            },
        }
        tpath = os.path.join(ROOT, "profiles", "traffic_calibrated.json")
        if os.path.exists(tpath) and world == 1 and args.keys == 200_000_000 and args.leaves == (1 << 20) and args.spec == "linear,linear":
            try:
                tj = json.load(open(tpath))
                out["roofline"]["traffic"] = tj.get(KERNEL_NAMES[dom], {}).get("hbm_bytes_per_launch")
                out["roofline"]["traffic_note"] = tj.get("note")
            except Exception:
                pass
        if world == 1 and keys_np is not None:
            # The boundary also takes host buffers (rmi_hip_upload_keys): the PCIe-inclusive rate of
            # "pageable host keys -> HBM -> one pass of the hot path".  Reported beside, never as, `value`.
            t0 = time.perf_counter()
            tr.set_keys(keys_np)
            t1 = time.perf_counter()
            run_step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out["pcie_inclusive"] = {"value": n_local / (t2 - t0), "unit": "keys/s", "upload_ms": (t1 - t0) * 1e3,
                                     "upload_GBps": n_local * key_bytes / (t1 - t0) / 1e9, "step_ms": (t2 - t1) * 1e3,
                                     "note": "pageable host buffer, hipMemcpy, then one step; not the headline value"}
        if world == 1 and root_kind in (0, 4):
            # B_leaf + B_root over t_root + t_leaf when the root is fitted on the GPU -- the
            # opt-in fast root (parallel sums, not bit-identical to the reference's sequential fit)
            tr.fit_root(root_kind, L_global, mode="fast")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fr = tr.fit_root(root_kind, L_global, mode="fast")
            root_fast_s = (time.perf_counter() - t0) / 10
            t0 = time.perf_counter()
            for _ in range(10):
                tr.train_leaves(fr, leaf_kind, L_local)
            leaf_s = (time.perf_counter() - t0) / 10
            out["fast_root_inclusive"] = {"value": n_local / (root_fast_s + leaf_s), "unit": "keys/s", "root_ms": root_fast_s * 1e3,
                                          "leaf_ms": leaf_s * 1e3, "bytes": 2 * n_local * key_bytes + 24 * L_local,
                                          "note": "root fitted on the GPU from parallel sums (opt-in, coefficients within ~1e-12 of the "
                                                  "exact fit, not bit-identical) + the leaf path of THAT root; not the headline value"}
        if not args.no_cpu_baseline and args.cpu_sample > 0 and world == 1:
            sample = keys_np[: min(args.cpu_sample, len(keys_np))]
            out["cpu_baseline"] = cpu_baseline(sample, args.spec, L_global, n_global)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
