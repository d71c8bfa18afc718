#!/usr/bin/env python3
"""bench.py -- keys/s of the two-layer leaf hot path on MI355X, one JSON line on rank 0.

A "step" is one pass of the hot path (rmi_hip_train_two_layer: bucketing scan, per-leaf fits, per-leaf
max-error pass, lower-bound widening, row packing -- two_layer.rs:126-287) over one batch of synthetic
sorted keys that is already resident in HBM, with the root-model parameters given (SURVEY.md section 8d;
the root fit, two_layer.rs:109-110, is timed separately and reported beside).  For N > 1 a step ends
when every rank holds the full row table (rmi_hip_train_sharded: kernels + one RCCL all-gather).

Workloads (`--config`, BASELINE.json):
    M   (default)  200M uniform uint64, linear,linear, 2^20 leaves   -- the configuration of `metric`
    C2             200M books-shaped uint64, linear,linear, 262144 leaves
    C3             200M uniform uint64, cubic,linear, 2^20 leaves
    C4             800M uniform uint64, linear,linear, 2^21 leaves   (8 GPUs x 100M keys; strong scaling)
    C5             400M uint32 (uniform | --dataset dups), radix,linear_spline, 2^22 leaves  (strong scaling)
`--scaling strong` (default; BASELINE.json quotes its metric on ONE 200M-key problem "at 1/2/4/8 GPUs"): the configuration is the
GLOBAL problem, cut into N leaf-aligned shards.  `--scaling weak`: every rank holds `--keys` keys and `--leaves` leaves of its own,
the global model is N times larger.
`--mode`: how linear leaves are fitted (include/rmi_hip.h, rmi_hip_set_fit_mode): exact | onepass_guarded
(default: one HBM pass, per-leaf error integers still bit-identical to the reference) | onepass.

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
KERNELS_EXACT = ["k_fit_stream", "k_fill", "k_fit_long", "k_err_range", "k_finalize+stats"]
KERNELS_ONEPASS = ["k_sigma2", "k_fill", "k_list", "k_list_tail", "k_finalize+stats"]
MODES = {"exact": 0, "onepass_guarded": 1, "onepass": 2}
CONFIGS = {
    # name: (keys, leaves, spec, dataset, dtype, scaling)
    "M": (200_000_000, 1 << 20, "linear,linear", "uniform", "uint64", None),
    "C2": (200_000_000, 262_144, "linear,linear", "books", "uint64", None),
    "C3": (200_000_000, 1 << 20, "cubic,linear", "uniform", "uint64", None),
    "C4": (800_000_000, 1 << 21, "linear,linear", "uniform", "uint64", "strong"),
    "C5": (400_000_000, 1 << 22, "radix,linear_spline", "uniform", "uint32", "strong"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (a step is under a millisecond: 200 steps let the clocks settle)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="a BASELINE.json configuration (sets keys, leaves, spec, dataset, dtype)")
    ap.add_argument("--keys", type=int, default=None, help="keys (per GPU with --scaling weak, in total with strong)")
    ap.add_argument("--leaves", type=int, default=None, help="leaves (per GPU with --scaling weak, in total with strong)")
    ap.add_argument("--spec", default=None)
    ap.add_argument("--dataset", default=None, choices=["uniform", "dups", "books"],
                    help="uniform / dups are generated in HBM; books (heavy-tailed, books_200M-shaped) on the host (1 GPU)")
    ap.add_argument("--dtype", default=None, choices=["uint64", "uint32"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--mode", default="onepass_guarded", choices=sorted(MODES))
    ap.add_argument("--cpu-sample", type=int, default=200_000_000,
                    help="keys of the workload the CPU baseline is timed on (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side figures (exact mode, PCIe-inclusive, fast root)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo for functional tests)")
    a = ap.parse_args()
    keys, leaves, spec, dataset, dtype, scaling = CONFIGS[a.config or "M"]
    a.keys = a.keys or keys
    a.leaves = a.leaves or leaves
    a.spec = a.spec or spec
    a.dataset = a.dataset or dataset
    a.dtype = a.dtype or dtype
    a.scaling = a.scaling or scaling or "strong"
    a.config = a.config or "M"
    return a


def cpu_baseline(keys_np, spec, leaves_total, n_total):
    """The oracle (C restatement of the reference's CPU path, 2 threads like rayon::join) timed on a bounded prefix of
    the same workload: once as the reference runs it (root fit included, two_layer.rs:109-287), once over the scope of
    `value` only (root given: :126-287).  Reported baseline, not the optimisation target."""
    from oracle import binding as oracle
    oracle.build()
    n = len(keys_np)
    L = max(2, int(round(leaves_total * (n / n_total))))   # same keys-per-leaf as the GPU workload
    root, leaf = spec.split(",")
    t0 = time.perf_counter()
    o = oracle.train_two_layer(root, leaf, keys_np, L, threads=2)
    dt_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.train_two_layer(root, leaf, keys_np, L, root=o.root, threads=2)
    dt_leaf = time.perf_counter() - t0
    return {
        "value": n / dt_leaf, "unit": "keys/s", "cores": 2, "kind": "port",
        "sample": f"first {n} keys of the workload, {spec}, {L} leaves (same keys/leaf); C restatement of the reference CPU path, "
                  f"LEAF PATH ONLY like `value` (root parameters given; two_layer.rs:126-287), 2 threads (rayon::join), "
                  f"{dt_leaf:.2f} s wall; host has {os.cpu_count()} cores",
        "with_root_fit": {"value": n / dt_full, "seconds": dt_full,
                          "note": "the same with the reference's sequential root fit included (two_layer.rs:109-110), as rmi_lib::train runs it"},
    }


def main():
    args = parse_args()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    local_rank = local_rank % max(1, torch.cuda.device_count())   # (functional tests may oversubscribe one GPU with gloo)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    from rmi_amd import train as T

    strong = args.scaling == "strong"
    if strong:
        n_global, L_global = args.keys, args.leaves
        if L_global % world:
            raise SystemExit("--leaves must be a multiple of the number of GPUs")
    else:
        n_global, L_global = args.keys * world, args.leaves * world
    np_dtype = np.uint64 if args.dtype == "uint64" else np.uint32
    key_bytes = np.dtype(np_dtype).itemsize
    mode = MODES[args.mode]
    root_kind, leaf_kind = T.parse_spec(args.spec)

    tr = T.Trainer(device=local_rank)
    sh = None
    keys_np = None
    if world == 1:
        if args.dataset == "books":
            from rmi_amd import datagen
            tr.set_keys(datagen.books_u64(n_global))
        else:
            tr.generate_keys(args.dataset, np_dtype, n_global, 0, n_global)
        t0 = time.perf_counter()
        keys_np = tr.download_keys()
        root = tr.fit_root(root_kind, L_global)        # exact (reference-order) root fit
        root_s = time.perf_counter() - t0
        tr.set_fit_mode(mode)
        run_step = lambda: tr.train_leaves(root, leaf_kind, L_global)
        n_local, L_local = n_global, L_global
    else:
        from rmi_amd import sharded
        if args.dataset == "books":
            raise SystemExit("books-shaped keys are generated on the host: single GPU only")
        sh = sharded.ShardedTrainer(tr, dist, rank, world, args.dataset, np_dtype, n_global, L_global, args.spec, fit_mode=mode)
        root_s = sh.root_seconds
        run_step = sh.step
        n_local, L_local = sh.plan.key_hi - sh.plan.key_lo, sh.plan.leaf_hi - sh.plan.leaf_lo

    def sync():
        if sh is not None:
            sh.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # hipEvents on the library's stream.  An event between two kernels costs ~5.5 us of idle device time, so the
    # timed steps bracket only the first (dominant) kernel of the path, and the per-kernel breakdown is taken in the
    # warm-up steps.
    tr.set_profile_level(2)
    warm_ns = np.zeros(8, dtype=np.float64)
    res = None
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
    kernel_ns[0] = dom_ns                                       # ... the first kernel live over the timed region
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        used = int(getattr(res, "fit_mode_used", 0))
        names = KERNELS_ONEPASS if used else KERNELS_EXACT
        ms_per_step = elapsed / args.steps * 1e3
        value = n_global / (elapsed / args.steps)
        kernel_us = (kernel_ns / args.steps / 1e3)[:5]
        dom = int(np.argmax(kernel_us))
        # algorithmic bytes per launch: one read of the keys + one write of the rows, per GPU
        row_bytes = 40 if leaf_kind == 2 else 24
        b_leaf = n_local * key_bytes + row_bytes * L_local
        dev_s = device_ns / args.steps * 1e-9               # all device work of a step on this rank (N>1: incl. the all-gather)
        path_gbs = b_leaf / dev_s / 1e9 if dev_s > 0 else 0.0
        dom_s = kernel_us[dom] * 1e-6
        dom_gbs = b_leaf / dom_s / 1e9 if dom_s > 0 else 0.0
        out = {
            "metric": f"keys/s trained ({n_global // 1_000_000}M {args.dtype}, {args.spec} {L_global} leaves)",
            "value": value, "unit": "keys/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {args.spec} {L_global} leaves on {n_global} synthetic sorted {args.dtype} keys ({args.dataset})"
                            + (f", cut into {world} leaf-aligned shards" if (world > 1 and strong) else
                               (f", {n_local} keys + {L_local} leaves per GPU" if world > 1 else "")),
                "keys_per_gpu": int(n_local), "leaves_per_gpu": int(L_local),
                "mode": {0: "exact: reference-order recurrence per leaf, two passes over the keys; coefficients bit-identical",
                         1: "one pass (sufficient statistics from LDS); error integers bit-identical through the guard, "
                            "flagged leaves re-fitted by the exact kernels; coefficients to the reference's rounding noise",
                         3: "one pass, bit-identical: linear_spline leaves (the line through a container's end points) from the "
                            "LDS ring, error pass from LDS",
                         2: "one pass, the least-squares line of the sums everywhere they are defined: guard-flagged leaves only "
                            "counted, long leaves from merged per-wave partial sums; a valid index, integers not certified"}[used],
                "mode_requested": args.mode,
                "exact_refit_leaves": int(getattr(res, "exact_leaves", 0)), "guard_flagged_leaves": int(getattr(res, "guard_leaves", 0)),
                "merged_long_leaves": int(getattr(res, "merged_leaves", 0)),
                "exchange": None if world == 1 else getattr(sh, "exchange", "") + "; a step ends when every rank holds the table",
                "root_fit_seconds_untimed": root_s,
            },
            "roofline": {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # B_leaf over ALL device work of the leaf path
                "achieved": path_gbs, "frac": path_gbs / HBM_PEAK_GBS, "device_us_per_step": dev_s * 1e6,
                "algorithmic_bytes": int(b_leaf),
                "kernel": names[dom], "kernel_achieved": dom_gbs, "kernel_frac": dom_gbs / HBM_PEAK_GBS,
                "kernel_us": {k: float(v) for k, v in zip(names, kernel_us)},
                "kernel_us_note": "the first kernel: hipEvents over the timed steps; the others: hipEvents over the warm-up steps "
                                  "(an event between two kernels idles the device ~5.5 us, so the timed steps carry only the two "
                                  "that bracket the first kernel)",
                "traffic": None,
            },
        }
        tpath = os.path.join(ROOT, "profiles", "traffic_r02.json")
        if os.path.exists(tpath) and world == 1 and args.config == "M":
            try:
                tj = json.load(open(tpath))
                ent = tj.get(names[dom], {})
                out["roofline"]["traffic"] = ent.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_note"] = "NOT measured in this run: rocprofv3 FETCH_SIZE/WRITE_SIZE of the same command, from " \
                                                  "profiles/traffic_r02.json (" + str(tj.get("note", "")) + ")"
            except Exception:
                pass
        if world == 1 and not args.no_extras:
            if used:                                         # the exact mode beside it
                tr.set_fit_mode(0)
                tr.set_profile_level(0)
                for _ in range(5):
                    run_step()
                t0 = time.perf_counter(); dv = 0
                for _ in range(50):
                    dv += run_step().device_ns
                torch.cuda.synchronize()
                ex_s = (time.perf_counter() - t0) / 50
                out["exact_mode"] = {"value": n_global / ex_s, "unit": "keys/s", "ms_per_step": ex_s * 1e3,
                                     "frac": b_leaf / (dv / 50 * 1e-9) / 1e9 / HBM_PEAK_GBS,
                                     "note": "the same step with rmi_hip_set_fit_mode(RMI_FIT_EXACT): two passes, coefficients bit-identical"}
                tr.set_fit_mode(mode)
            # The boundary also takes host buffers: the PCIe-inclusive rate of "pageable host keys -> HBM -> the leaf path",
            # (a) plain: rmi_hip_upload_keys (one hipMemcpy) then one step; (b) rmi_hip_train_streamed: chunked upload
            # through pinned staging buffers, every leaf-aligned shard trained behind the upload of the following ones.
            # Reported beside, never as, `value`.
            t0 = time.perf_counter()
            tr.set_keys(keys_np)
            t1 = time.perf_counter()
            run_step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            plain = {"value": n_global / (t2 - t0), "upload_ms": (t1 - t0) * 1e3, "upload_GBps": n_global * key_bytes / (t1 - t0) / 1e9,
                     "step_ms": (t2 - t1) * 1e3}
            chunks = 16
            while L_global % chunks:
                chunks //= 2
            tr.train_streamed(keys_np, root, leaf_kind, L_global, chunks=chunks)      # (first call allocates the staging buffers)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                tr.train_streamed(keys_np, root, leaf_kind, L_global, chunks=chunks)
                ts.append(time.perf_counter() - t0)
            st_s = min(ts)
            out["pcie_inclusive"] = {"value": n_global / st_s, "unit": "keys/s", "ms": st_s * 1e3, "GBps": n_global * key_bytes / st_s / 1e9,
                                     "chunks": chunks, "plain_upload_then_step": plain,
                                     "note": "pageable host keys -> rmi_hip_train_streamed (64 MB pinned staging buffers, host copy of chunk "
                                             "c+1 beside the DMA of chunk c, shards trained behind the upload; best of 3) ; the 63 GB/s link "
                                             "bounds it at ~2.5e-2 s for these keys; not the headline value"}
            if root_kind in (0, 4):
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
                    tr.train_leaves(fr, leaf_kind, L_global)
                leaf_s = (time.perf_counter() - t0) / 10
                out["fast_root_inclusive"] = {"value": n_global / (root_fast_s + leaf_s), "unit": "keys/s", "root_ms": root_fast_s * 1e3,
                                              "leaf_ms": leaf_s * 1e3, "bytes": 2 * n_global * key_bytes + row_bytes * L_global,
                                              "note": "root fitted on the GPU from parallel sums (opt-in, coefficients within ~1e-12 of the "
                                                      "exact fit, not bit-identical) + the leaf path of THAT root; not the headline value"}
        if not args.no_cpu_baseline and args.cpu_sample > 0 and world == 1 and keys_np is not None:
            sample = keys_np[: min(args.cpu_sample, len(keys_np))]
            out["cpu_baseline"] = cpu_baseline(sample, args.spec, L_global, n_global)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
