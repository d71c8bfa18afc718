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
`--mode`: how linear leaves are fitted (include/rmi_hip.h, rmi_hip_set_fit_mode): exact (default: the reference's
recurrence per leaf in reference order -- bucket ids, error integers AND coefficients bit-identical; `value` is quoted in
this mode, and `parity_check` in the same JSON line says what the oracle found) | onepass_guarded | onepass (the
sufficient-statistics modes: side figure `fast_mode`, coefficients only to the reference's own rounding noise).

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
KERNELS_EXACT = ["k_fit_stream", "k_fill", "k_fit_long", "k_err_range", "k_finalize+stats"]            # RMI_HIP_PIPELINE=2
KERNELS_REGS = ["k_leaf_regs", "k_leaf_lanes_listed+k_regs_finalize", "k_lane_reduce", "-", "-"]      # the default exact path (pipeline 4)
KERNELS_SCAN = ["k_spline_scan", "k_scan_gaps+k_lane_reduce", "-", "-", "-"]                         # pipeline 5: linear_spline leaves
KERNELS_LANES = ["k_leaf_lanes", "k_lane_reduce", "-", "-", "-"]                                    # pipeline 3 (RMI_HIP_REGS=0, and where 4 does not apply)
KERNELS_LANES_INSTREAM = ["k_leaf_lanes", "k_list", "k_list_tail", "k_finalize_listed+stats", "-"]    # RMI_HIP_OPT_TAIL=0
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
    ap.add_argument("--mode", default="exact", choices=sorted(MODES))
    ap.add_argument("--cpu-sample", type=int, default=200_000_000,
                    help="keys of the workload the CPU baseline is timed on (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side figures (fast mode, other configurations, PCIe-inclusive, fast root)")
    ap.add_argument("--no-configs", action="store_true", help="skip the side figures of the other BASELINE configurations")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo for functional tests)")
    ap.add_argument("--exchange", default=None, choices=["rccl", "direct", "auto"],
                    help="N>1: how the rows reach every rank -- ncclAllGather, direct peer stores over xGMI (rmi_hip_peer_*), or (default "
                         "with the nccl backend) an A/B at start-up that takes the direct form only if its table equals RCCL's and it is "
                         "faster, and prints both timings (exchange_ab)")
    a = ap.parse_args()
    keys, leaves, spec, dataset, dtype, scaling = CONFIGS[a.config or "M"]
    a.keys = a.keys or keys
    a.leaves = a.leaves or leaves
    a.spec = a.spec or spec
    a.dataset = a.dataset or dataset
    a.dtype = a.dtype or dtype
    a.scaling = a.scaling or scaling or "strong"
    a.config = a.config or "M"
    if a.exchange is None:                # an unflagged N > 1 run measures both exchanges (xGMI is point-to-point: a ring all-gather alone would miss the target)
        a.exchange = "auto" if a.backend == "nccl" else "rccl"
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
    return o, L, {
        "value": n / dt_leaf, "unit": "keys/s", "cores": 2, "kind": "port",
        "sample": f"first {n} keys of the workload, {spec}, {L} leaves (same keys/leaf); C restatement of the reference CPU path, "
                  f"LEAF PATH ONLY like `value` (root parameters given; two_layer.rs:126-287), 2 threads (rayon::join), "
                  f"{dt_leaf:.2f} s wall; host has {os.cpu_count()} cores",
        "with_root_fit": {"value": n / dt_full, "seconds": dt_full,
                          "note": "the same with the reference's sequential root fit included (two_layer.rs:109-110), as rmi_lib::train runs it"},
    }


def sources_sha256():
    """sha256 over the kernel sources the library is built from: ties a counter file under profiles/ to the code it measured."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rmi_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def traffic_file(tag):
    """The counter file of configuration `tag` under profiles/, the newest round's first."""
    for rnd in ("r06", "r05"):
        p = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (rnd, tag))
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "traffic_r06_%s.json" % tag)


def env_info(torch):
    """Versions and clocks of the box the line was measured on (best effort: never fails the bench)."""
    import subprocess
    info = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    try:
        info["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info["rccl"] = None
    try:
        p = torch.cuda.get_device_properties(0)
        info["device"] = p.name
        info["cus"] = p.multi_processor_count
        info["gcn_arch"] = getattr(p, "gcnArchName", None)
    except Exception:
        pass
    try:
        info["rocm_version"] = open("/opt/rocm/.info/version").read().strip()
    except Exception:
        info["rocm_version"] = None
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        c0 = j.get("card0", {})
        info["clocks"] = {k: v for k, v in c0.items() if "sclk" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower()}
    except Exception as ex:
        info["clocks"] = "unavailable: " + str(ex)[:80]
    return info


def parity_against(o, g, what):
    """The three output clauses of north_star, measured: the GPU result `g` of the timed mode against the oracle's `o`
    on the same keys (bucket table and integers bit for bit, coefficients relative)."""
    import numpy as np
    gp, op = g.leaf_params, o.leaf_params
    with np.errstate(all="ignore"):
        rel = np.abs(gp - op) / np.maximum(np.abs(op), 1e-300)
    rel[gp == op] = 0.0
    slope = rel[:, -1] if rel.ndim == 2 and rel.shape[1] == 2 else rel.max(axis=1)
    return {
        "against": "oracle (C restatement of the reference's trainer; parity UNPINNED: the Rust reference cannot be built here)",
        "what": what,
        "buckets_equal": bool(np.array_equal(g.leaf_starts, o.leaf_start)),
        "ints_equal": bool(np.array_equal(g.last_layer_max_l1s, o.leaf_err)),
        "counts_equal": bool(np.array_equal(g.leaf_counts, o.leaf_count)),
        "coef_bit_identical": bool(np.array_equal(gp.view(np.uint64), op.view(np.uint64))),
        "coef_max_rel": float(rel.max()) if rel.size else 0.0,                      # elementwise over every coefficient (alpha AND beta)
        "coef_frac_within_1e-9": float((rel <= 1e-9).all(axis=1).mean()) if rel.size else 1.0,
        "slope_max_rel": float(slope.max()) if slope.size else 0.0,
        "slope_frac_within_1e-9": float((slope <= 1e-9).mean()) if slope.size else 1.0,
        "intercept_max_abs_diff": float(np.abs(gp[:, 0] - op[:, 0]).max()) if gp.size else 0.0,   # in positions (what a prediction moves by)
        "leaves": int(len(o.leaf_err)),
        # the model-level aggregates (two_layer.rs:267-287): maximum and its leaf, average error exactly; the two f64 sums to 1e-9 (north_star's tolerance for floating point: they run in another order than the reference's, their differences are printed below)
        "aggregates_equal": bool(int(g.model_max_error) == int(o.model_max_error) and int(g.model_max_error_idx) == int(o.model_max_error_idx)
                                 and float(g.model_avg_error) == float(o.model_avg_error)
                                 and abs(float(g.model_avg_log2_error) - float(o.model_avg_log2_error)) <= 1e-9 * max(1.0, abs(float(o.model_avg_log2_error)))
                                 and abs(float(g.model_avg_l2_error) - float(o.model_avg_l2_error)) <= 1e-9 * max(1.0, abs(float(o.model_avg_l2_error)))),
        # (the two f64 sums are the reference's in leaf order, here per group of 64 leaves and then over the groups: not the same roundings)
        "avg_l2_rel_diff": abs(float(g.model_avg_l2_error) - float(o.model_avg_l2_error)) / max(1e-300, abs(float(o.model_avg_l2_error))),
        "avg_log2_rel_diff": abs(float(g.model_avg_log2_error) - float(o.model_avg_log2_error)) / max(1e-300, abs(float(o.model_avg_log2_error))),
    }


def time_steps(step, n, warm=3):
    """(seconds per step by the wall clock, device seconds per step) of `step`, a callable returning a result record."""
    import torch
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); dv = 0
    r = None
    for _ in range(n):
        r = step(); dv += r.device_ns
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, dv / n * 1e-9, r


def side_configs(T, tr_m, device, with_oracle):
    """C2 / C3 / C5 on one GPU: ms per step, keys/s, fraction of the HBM roofline, the mode that ran, and -- where the
    oracle finishes in seconds -- the parity flags.  Few steps each: side figures, not `value`."""
    import numpy as np
    import torch
    from rmi_amd import datagen
    res = {}

    def run(tr, root, leaf_kind, L, mode, steps, nkeys, kbytes, rowb=24):
        """One configuration: wall and device time per step, the dominant kernel (first kernel group of the call) with its own bracket,
        the algorithmic bytes and the fractions of the 8 TB/s roofline they give."""
        tr.set_fit_mode(mode)
        tr.train_leaves(root, leaf_kind, L)                         # (the first training of a configuration allocates)
        tr.set_profile_level(1)
        k0 = 0.0
        for _ in range(4):
            k0 += tr.train_leaves(root, leaf_kind, L).kernel_ns[0]
        tr.set_profile_level(0)
        w, d, r = time_steps(lambda: tr.train_leaves(root, leaf_kind, L), steps, warm=2)
        pl = int(getattr(r, "pipeline", 0))
        b = nkeys * kbytes + rowb * L
        return {"ms_per_step": w * 1e3, "device_ms": d * 1e3, "value": nkeys / w, "unit": "keys/s", "algorithmic_bytes": int(b),
                "frac": b / d / 1e9 / HBM_PEAK_GBS, "frac_wall": b / w / 1e9 / HBM_PEAK_GBS,
                "kernel": {5: "k_spline_scan", 4: "k_leaf_regs", 3: "k_leaf_lanes"}.get(pl, "k_fit_stream" if not int(r.fit_mode_used) else "k_sigma2"),
                "kernel_us": k0 / 4 / 1e3, "kernel_frac": b / (k0 / 4 * 1e-9) / 1e9 / HBM_PEAK_GBS if k0 else None, "pipeline": pl,
                "mode_used": int(r.fit_mode_used), "listed_long_leaves": int(r.long_leaves), "exact_refit_leaves": int(r.exact_leaves)}, r

    def with_traffic(e, tag):
        """HBM bytes of the step from profiles/traffic_r05_<tag>.json (rocprofv3 counters of the same workload), quoted only if that
        file was taken on a build of these kernel sources."""
        try:
            tpath = traffic_file(tag)
            tj = json.load(open(tpath))
            if tj.get("sources_sha256") == sources_sha256():
                e["traffic"] = tj.get("step_hbm_bytes")
                e["traffic_ratio"] = tj.get("traffic_ratio")
                e["traffic_note"] = "not measured in this run: %s (tools/profile_cfg.sh), same kernel sources" % os.path.relpath(tpath, ROOT)
        except Exception:
            pass
        return e

    # the shard shapes of the 8-GPU configurations on this one GPU: what a rank's kernels cost at N = 8 (DESIGN section 6 quotes these)
    for name, nk, L in (("M shard 1/8: linear,linear 131072 leaves on 25M u64", 25_000_000, 1 << 17),
                        ("C4 shard 1/8: linear,linear 262144 leaves on 100M u64 (381 keys a leaf)", 100_000_000, 1 << 18)):
        try:
            ts = T.Trainer(device=device)
            ts.generate_keys("uniform", np.uint64, nk)
            root = ts.fit_root("linear", L, mode="fast")
            e, r = run(ts, root, 0, L, 0, 20, nk, 8)
            ts.set_profile_level(2)
            acc = np.zeros(8)
            for _ in range(4):
                acc += np.array(ts.train_leaves(root, 0, L).kernel_ns, dtype=float)
            e["kernel_groups_us"] = [float(x) / 4e3 for x in acc[:5]]
            e["note"] = "root from the parallel sums (its coefficients do not matter to the leaf path's time); single GPU, exact mode, no exchange"
            res[name] = {"exact": with_traffic(e, "ms" if nk == 25_000_000 else "c4s")}
            ts.close()
        except Exception as ex:
            res[name] = {"error": str(ex)}

    # C3: cubic root over the metric configuration's keys (already resident)
    try:
        n, L = tr_m.n, 1 << 20
        root = tr_m.fit_root("cubic", L)
        e, r = run(tr_m, root, 0, L, 0, 20, n, 8)
        if with_oracle:
            from oracle import binding as oracle
            t0 = time.perf_counter()
            o = oracle.train_two_layer("cubic", "linear", tr_m.download_keys(), L, threads=2)
            e["parity_check"] = parity_against(o, r.materialize(), "C3 at full size, exact mode (root included: the oracle fits its own cubic)")
            e["parity_check"]["root_equal"] = bool(tuple(root.p) == tuple(o.root.p))
            e["parity_check"]["oracle_seconds"] = time.perf_counter() - t0
            del o
        res["C3 cubic,linear 2^20 on 200M u64"] = {"exact": with_traffic(e, "c3")}
        tr_m.set_fit_mode(0)
    except Exception as ex:                                   # a side figure must not take the headline down
        res["C3"] = {"error": str(ex)}
    # S64: linear_spline leaves over the metric configuration's keys (191 keys a leaf against 64 keys of look-ahead: k_spline_scan's short form
    # looks for the open leaf's end in the key array)
    try:
        n, L = tr_m.n, 1 << 20
        root = tr_m.fit_root("linear", L)
        e, r = run(tr_m, root, 1, L, 0, 20, n, 8)
        if with_oracle:
            from oracle import binding as oracle
            t0 = time.perf_counter()
            o = oracle.train_two_layer("linear", "linear_spline", tr_m.download_keys(), L, threads=2)
            e["parity_check"] = parity_against(o, r.materialize(), "200M u64 linear,linear_spline 2^20 at full size: every leaf")
            e["parity_check"]["root_equal"] = bool(tuple(root.p) == tuple(o.root.p))
            e["parity_check"]["oracle_seconds"] = time.perf_counter() - t0
            del o
        res["S64 linear,linear_spline 2^20 on 200M u64"] = {"exact": with_traffic(e, "s64")}
        tr_m.set_fit_mode(0)
    except Exception as ex:
        res["S64"] = {"error": str(ex)}
    # C5: 400M u32, radix root, linear_spline leaves (one pass, bit-identical in every mode), uniform and duplicate-heavy
    for ds in ("uniform", "dups"):
        try:
            t5 = T.Trainer(device=device)
            t5.generate_keys(ds, np.uint32, 400_000_000)
            root = t5.fit_root("radix", 1 << 22)
            e, r = run(t5, root, 1, 1 << 22, 0, 20, 400_000_000, 4)
            if with_oracle:
                from oracle import binding as oracle
                t0 = time.perf_counter()
                o = oracle.train_two_layer("radix", "linear_spline", t5.download_keys(), 1 << 22, threads=2)
                e["parity_check"] = parity_against(o, r.materialize(), f"C5 ({ds}) at full size: every leaf of the 2^22")
                e["parity_check"]["root_equal"] = bool(tuple(root.p) == tuple(o.root.p) and tuple(root.ip) == tuple(o.root.ip))
                e["parity_check"]["oracle_seconds"] = time.perf_counter() - t0
                del o
            res[f"C5 radix,linear_spline 2^22 on 400M u32 ({ds})"] = {"exact": with_traffic(e, "c5" if ds == "uniform" else "c5_dups")}
            t5.close()
        except Exception as ex:
            res[f"C5 {ds}"] = {"error": str(ex)}
    # 4-byte keys with `linear` leaves (src/load.rs:47-69 `*_uint32` files): the register kernel at TWO waves per SIMD (k_leaf_regs<K, 2>: raw keys stashed)
    try:
        t4 = T.Trainer(device=device)
        t4.generate_keys("uniform", np.uint32, 400_000_000)
        L = 1 << 21
        root = t4.fit_root("linear", L)
        e, r = run(t4, root, 0, L, 0, 20, 400_000_000, 4)
        if with_oracle:
            from oracle import binding as oracle
            t0 = time.perf_counter()
            o = oracle.train_two_layer("linear", "linear", t4.download_keys(), L, threads=2)
            e["parity_check"] = parity_against(o, r.materialize(), "400M u32 linear,linear 2^21 at full size: every leaf")
            e["parity_check"]["root_equal"] = bool(tuple(root.p) == tuple(o.root.p))
            e["parity_check"]["oracle_seconds"] = time.perf_counter() - t0
            del o
        res["U32 linear,linear 2^21 on 400M u32 (uniform)"] = {"exact": with_traffic(e, "u32")}
        t4.close()
    except Exception as ex:
        res["U32"] = {"error": str(ex)}
    # C2: books-shaped 200M (heavy-tailed: one leaf of ~2.5 M keys), 262144 leaves, every mode
    try:
        t2 = T.Trainer(device=device)
        kt = datagen.books_u64_torch(200_000_000, device=f"cuda:{device}")
        torch.cuda.synchronize()
        t2.set_keys(kt)
        L = 262_144
        root = t2.fit_root("linear", L)
        ent = {}
        e, r_exact = run(t2, root, 0, L, 0, 3, 200_000_000, 8)
        ent["exact"] = with_traffic(e, "c2")
        g_exact = r_exact.materialize()
        # what bounds the exact mode here: the recurrence of a leaf is ONE sequential chain by the definition of bit-identical
        # coefficients; the longest container sets the floor -- on a host core (~3.2 ns a point, containers of more than
        # RMI_HIP_HOST_MIN = 262 144 points) or on one wave of the device (~28 ns a point)
        try:
            ls = np.asarray(g_exact.leaf_starts, dtype=np.int64)
            longest = int(np.diff(np.append(ls, 200_000_000)).max()) + 2
            host_min = int(os.environ.get("RMI_HIP_HOST_MIN", "262144"))
            ns_pt = 3.2 if (host_min > 0 and longest > host_min) else 28.0
            ent["exact"]["longest_container"] = longest
            ent["exact"]["chain_floor_ms"] = longest * ns_pt * 1e-6
            ent["exact"]["chain_floor_note"] = (f"{longest} points x {ns_pt} ns a point ({'host core' if ns_pt < 10 else 'one wave'}): the sequential chain of the longest "
                                                "leaf; lowering RMI_HIP_HOST_MIN was measured (tools/c2_sweep.py): 131 072 -> 47 ms, 32 768 -> 76 ms against 11.7 ms")
        except Exception:
            pass
        for mname, m, steps in (("onepass_guarded", 1, 3), ("onepass", 2, 10)):
            e, r = run(t2, root, 0, L, m, steps, 200_000_000, 8)
            e["ints_equal_to_exact"] = bool(np.array_equal(r.last_layer_max_l1s, g_exact.last_layer_max_l1s))
            ent[mname] = e
        if with_oracle:
            from oracle import binding as oracle
            keys_h = t2.download_keys()
            t0 = time.perf_counter()
            o = oracle.train_two_layer("linear", "linear", keys_h, L, threads=2)
            ent["exact"]["parity_check"] = parity_against(o, g_exact, "C2, exact mode")
            ent["exact"]["parity_check"]["oracle_seconds"] = time.perf_counter() - t0
        res["C2 linear,linear 262144 on 200M books-shaped u64"] = ent
        t2.close()
    except Exception as ex:
        res["C2"] = {"error": str(ex)}
    return res


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment (the way the driver starts N = 1): start the N ranks
    ourselves -- the same command under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 at a free port.  Launched by
    torchrun already (WORLD_SIZE set), nothing happens here."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse_args()
    respawn_under_torchrun(args)
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
        # a step = one call of the C entry point (launches + synchronisation + result record); the Python wrapper's result
        # object (~5 us per call) is built once, behind the timed region
        import ctypes as C
        from rmi_amd import _lib as L_
        root_c, res_c = root._c(), L_.Result()

        def run_step():
            rc = tr._lib.rmi_hip_train_two_layer(tr._h, C.byref(root_c), leaf_kind, L_global, C.byref(res_c))
            if rc:
                T._check(rc, tr._h)
            return res_c
        n_local, L_local = n_global, L_global
    else:
        from rmi_amd import sharded
        if args.dataset == "books":
            raise SystemExit("books-shaped keys are generated on the host: single GPU only")
        sh = sharded.ShardedTrainer(tr, dist, rank, world, args.dataset, np_dtype, n_global, L_global, args.spec, fit_mode=mode, exchange=args.exchange)
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
    # the box's own streaming read rate, measured in this run on the resident key array (a read-only kernel,
    # non-temporal 16-byte loads): what "HBM-bound" can mean on this machine, beside the 8 TB/s of the data sheet.  It runs
    # HERE, in front of the warm-up: the exact root fit before it keeps the host busy for ~1 s with the GPU idle, and a device
    # that has dropped to its low-power state needs more than W = 5 steps of 0.5 ms to be back at its clocks.
    measured_bw, measured_bw_stride, measured_bw_err = None, None, None
    if world == 1:
        try:
            # the best read-only pattern found on this machine (contiguous 8 KB pieces per wave, non-temporal loads: what the one-read
            # kernels issue), and the grid-stride pattern of rounds 1-4 beside it
            measured_bw = float(max(tr.measure_read_bandwidth(10, 1) for _ in range(3)))
            measured_bw_stride = float(max(tr.measure_read_bandwidth(10, 0) for _ in range(3)))
        except Exception as ex:                                         # (never the reason a bench line is lost)
            measured_bw_err = str(ex)[:120]
    tr.set_profile_level(2)
    warm_ns = np.zeros(8, dtype=np.float64)
    res = None
    for _ in range(args.warmup):
        res = run_step()
        warm_ns += np.array(res.kernel_ns, dtype=np.float64)
    sync()
    # Events are not free (measured, tools/host_overhead.py: the two around the first kernel cost a step ~8 us of device idle
    # time and ~8 us of host time; the two around the whole call ~4 us): of every 4 timed steps one carries the bracket of
    # the first kernel (profile level 1), one only the bracket of the whole call (level 0 -> device_ns), two none (-1, as a
    # caller who wants the model and no timings runs it).  Each figure is the average over the steps that measure it.
    dom_ns, dom_steps = 0.0, 0
    device_ns, dev_steps = 0.0, 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        lvl = (1, -1, 0, -1)[i % 4] if args.steps >= 4 else 1
        tr.set_profile_level(lvl)
        res = run_step()
        if lvl == 1:
            dom_ns += res.kernel_ns[0]
            dom_steps += 1
        if lvl == 0 or args.steps < 4:
            device_ns += res.device_ns
            dev_steps += 1
    sync()
    elapsed = time.perf_counter() - t0
    tr.set_profile_level(1)
    device_ns = device_ns / max(dev_steps, 1) * args.steps      # (scaled to the timed region: the code below divides by the steps)
    kernel_ns = warm_ns / max(args.warmup, 1) * args.steps      # breakdown from the warm-up steps ...
    kernel_ns[0] = dom_ns / max(dom_steps, 1) * args.steps      # ... the first kernel live over the timed region
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # per-rank split of a step for N > 1 (the first scaling curve should explain itself): kernels, tail, exchange
    per_rank = None
    if dist is not None:
        kus = warm_ns / max(args.warmup, 1) / 1e3
        mine = torch.tensor([device_ns / args.steps / 1e3, kus[0], float(kus[1:5].sum()), kus[7], float(n_local), float(L_local)],
                            dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "device_us": float(v[0]), "dominant_kernel_us": float(v[1]), "other_kernels_us": float(v[2]),
                     "exchange_us": float(v[3]), "keys": int(v[4]), "leaves": int(v[5])} for r, v in enumerate(allr)]

    if rank == 0:
        used = int(getattr(res, "fit_mode_used", 0))
        lanes_path = used == 0 and leaf_kind in (0, 1) and os.environ.get("RMI_HIP_PIPELINE", "3") not in ("2",) and n_local >= 1024
        last_pl = int(tr._lib.rmi_hip_last_pipeline(tr._h))
        regs_path = lanes_path and last_pl == 4
        scan_path = lanes_path and last_pl == 5
        names = KERNELS_ONEPASS if used else ((KERNELS_SCAN if scan_path else KERNELS_REGS if regs_path else (KERNELS_LANES_INSTREAM if os.environ.get("RMI_HIP_OPT_TAIL", "1") == "0" else KERNELS_LANES)) if lanes_path else KERNELS_EXACT)
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
        # (the timed trainings' arrays, through the wrapper: one more training of the same configuration, before any other one)
        g_head = tr.train_leaves(root, leaf_kind, L_global).materialize() if world == 1 else None
        mode_text = {
            0: ("exact, key-parallel one-read kernel for linear_spline leaves (k_spline_scan): bucketing scan, the containers' end points, the error "
                "pass and the leaf ends in one pass over coalesced 16-byte loads; no recurrence, coefficients bit-identical") if scan_path else
               ("exact, register-resident leaf kernel: leaf boundaries by search (k_leaf_search), then 64 leaves per wave in lockstep, ONE wave "
                "per SIMD with 512 registers -- the reference's recurrence per leaf in reference order (coefficients bit-identical); the keys "
                "arrive by LDS-DMA and stay in the lane's registers for the error pass behind the fit (k_leaf_regs): the keys are read ONCE; "
                "the leaf's widening, row and aggregates in k_regs_finalize") if regs_path else
               ("exact, leaf-lane kernels: leaf boundaries by search (k_leaf_search), then 64 leaves per wave in lockstep -- the reference's "
                "recurrence per leaf in reference order (coefficients bit-identical), the error pass and the leaf's finalize behind it in the "
                "same kernel (k_leaf_lanes); the keys are read twice, the second time through the Infinity Cache") if lanes_path else
               "exact: reference-order recurrence per leaf, two streaming passes over the keys; coefficients bit-identical",
            1: "one pass (sufficient statistics from LDS); error integers bit-identical through the guard, "
               "flagged leaves re-fitted by the exact kernels; coefficients to the reference's rounding noise (NOT within 1e-9 everywhere)",
            3: "one pass, bit-identical: linear_spline leaves (the line through a container's end points) from the "
               "LDS ring, error pass from LDS",
            2: "one pass, the least-squares line of the sums everywhere they are defined: guard-flagged leaves only "
               "counted, long leaves from merged per-wave partial sums; a valid index, integers not certified"}[used]
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
                "mode": mode_text, "mode_requested": args.mode,
                "exact_refit_leaves": int(getattr(res, "exact_leaves", 0)), "guard_flagged_leaves": int(getattr(res, "guard_leaves", 0)),
                "merged_long_leaves": int(getattr(res, "merged_leaves", 0)), "listed_long_leaves": int(getattr(res, "long_leaves", 0)),
                "exchange": None if world == 1 else getattr(sh, "exchange", "") + "; a step ends when every rank holds the table",
                "root_fit_seconds_untimed": root_s,
            },
            "roofline": {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # B_leaf over ALL device work of the leaf path
                "achieved": path_gbs, "frac": path_gbs / HBM_PEAK_GBS, "device_us_per_step": dev_s * 1e6,
                "algorithmic_bytes": int(b_leaf),
                "kernel": names[dom], "kernel_achieved": dom_gbs, "kernel_frac": dom_gbs / HBM_PEAK_GBS,
                "kernel_us": {k: float(v) for k, v in zip(names, kernel_us) if k != "-"},
                "unbracketed_us": float(dev_s * 1e6 - kernel_us[0]),
                "kernel_us_note": "hipEvents on the library's stream, inside the timed region: of every 4 timed steps one brackets the first "
                                  "kernel (kernel_us[0]), one only the whole call (device_us_per_step), two carry no event -- an event "
                                  "between two kernels idles the device ~5.5 us and costs the host ~4 us; the other kernels: events over "
                                  "the warm-up steps; unbracketed_us = device time of a step outside the first kernel (k_leaf_samples "
                                  "with the init, k_leaf_search, k_lane_reduce; pipeline 4: the listed groups and k_regs_finalize too)",
                "traffic": None,
            },
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            if getattr(sh, "auto_report", None):
                out["exchange_ab"] = sh.auto_report
        tag = args.config.lower() + ("_dups" if args.dataset == "dups" and args.config == "C5" else "")
        tpath = traffic_file(tag)
        if os.path.exists(tpath) and world == 1 and not used:
            try:
                tj = json.load(open(tpath))
                # the counters were taken on a build of these very sources, or the figure is not quoted
                if tj.get("sources_sha256") == sources_sha256():
                    out["roofline"]["traffic"] = tj.get("step_hbm_bytes")
                    out["roofline"]["traffic_ratio"] = tj.get("traffic_ratio")
                    out["roofline"]["traffic_kernels"] = {k: v.get("hbm_bytes_per_launch") for k, v in tj.get("kernels", {}).items()}
                    out["roofline"]["traffic_note"] = "the STEP's sum over its kernels; NOT measured in this run: rocprofv3 FETCH_SIZE / WRITE_SIZE of the same " \
                                                      "workload (tools/profile_cfg.sh) on a build of the same kernel sources (sha256 " + sources_sha256()[:12] + \
                                                      "), from " + os.path.relpath(tpath, ROOT) + " (" + str(tj.get("note", ""))[:160] + " ...)"
                else:
                    out["roofline"]["traffic_note"] = "profiles/ holds counters of OTHER kernel sources (" + str(tj.get("sources_sha256"))[:12] + \
                                                      " against " + sources_sha256()[:12] + "): not quoted; tools/profile_cfg.sh takes them again"
            except Exception:
                pass

        if world == 1:
            bw = measured_bw
            if bw:
                out["roofline"]["measured_peak"] = float(bw)
                out["roofline"]["frac_of_measured"] = float(path_gbs / bw)
                out["roofline"]["kernel_frac_of_measured"] = float(dom_gbs / bw)
                out["roofline"]["measured_peak_grid_stride"] = measured_bw_stride
                out["roofline"]["measured_peak_note"] = ("rmi_hip_measure_read_bandwidth_ex, pattern 1: best of 3 x 10 passes of a read-only kernel over the same key "
                                                         "array (every wave reads contiguous 8 KB pieces with non-temporal 16-byte loads -- the best read-only pattern "
                                                         "found on this machine), this run, right in front of the warm-up steps; measured_peak_grid_stride: the "
                                                         "grid-stride pattern rounds 1-4 quoted")
            else:
                out["roofline"]["measured_peak"] = None
                out["roofline"]["measured_peak_note"] = "failed: " + str(measured_bw_err)
        out["env"] = env_info(torch)
        # the driver's flags give a 12 ms timed region; the same steps once more over 200 (the protocol of profiles/)
        if world == 1 and args.steps < 200 and not args.no_extras:
            tr.set_profile_level(-1)
            for _ in range(20):
                run_step()
            sync()
            t1 = time.perf_counter()
            for _ in range(200):
                run_step()
            sync()
            ms200 = (time.perf_counter() - t1) / 200 * 1e3
            out["long_protocol"] = {"steps": 200, "warmup": 20 + args.steps + args.warmup, "ms_per_step": ms200, "value": n_global / (ms200 * 1e-3),
                                    "note": "the same call, no events at all: 200 back-to-back steps behind the timed region"}
            tr.set_profile_level(1)

        def frac_of(nkeys, kbytes, L, rowb, dsec):
            return (nkeys * kbytes + rowb * L) / dsec / 1e9 / HBM_PEAK_GBS if dsec > 0 else 0.0

        if world == 1 and not args.no_extras:
            tr.set_profile_level(0)
            if not used and leaf_kind == 0:
                # the sufficient-statistics mode beside it (SURVEY H2's fast mode): one read of the keys, integers through
                # the guard, coefficients to the reference's own rounding noise -- compared here with the exact result
                tr.set_fit_mode(1)
                w_s, d_s, _ = time_steps(run_step, 50, warm=5)
                r1 = tr.train_leaves(root, leaf_kind, L_global)
                fm = {"value": n_global / w_s, "unit": "keys/s", "ms_per_step": w_s * 1e3, "frac": frac_of(n_global, key_bytes, L_global, row_bytes, d_s),
                      "mode_used": int(r1.fit_mode_used), "exact_refit_leaves": int(r1.exact_leaves),
                      "note": "rmi_hip_set_fit_mode(RMI_FIT_ONEPASS_GUARDED): NOT the headline -- its coefficients miss north_star's 1e-9 on some leaves"}
                if g_head is not None:
                    gp, ep = r1.leaf_params, g_head.leaf_params
                    with np.errstate(all="ignore"):
                        rel = np.abs(gp[:, 1] - ep[:, 1]) / np.maximum(np.abs(ep[:, 1]), 1e-300)
                    rel[gp[:, 1] == ep[:, 1]] = 0.0
                    fm["vs_exact"] = {"ints_equal": bool(np.array_equal(r1.last_layer_max_l1s, g_head.last_layer_max_l1s)),
                                      "buckets_equal": bool(np.array_equal(r1.leaf_starts, g_head.leaf_starts)),
                                      "slope_max_rel": float(rel.max()), "slope_frac_within_1e-9": float((rel <= 1e-9).mean()),
                                      "intercept_max_abs_diff": float(np.abs(gp[:, 0] - ep[:, 0]).max()),
                                      "note": "slopes relative; intercepts in positions (alpha = mean_y - beta mean_x cancels to ~0 on uniform keys: "
                                              "its RELATIVE difference says nothing)"}
                out["fast_mode"] = fm
                tr.set_fit_mode(mode)
            elif used:                                       # the exact mode beside it
                tr.set_fit_mode(0)
                w_s, d_s, _ = time_steps(run_step, 50, warm=5)
                out["exact_mode"] = {"value": n_global / w_s, "unit": "keys/s", "ms_per_step": w_s * 1e3,
                                     "frac": frac_of(n_global, key_bytes, L_global, row_bytes, d_s),
                                     "note": "the same step with rmi_hip_set_fit_mode(RMI_FIT_EXACT): coefficients bit-identical"}
                tr.set_fit_mode(mode)
            # two trainings in flight on the one resident key set (two contexts / streams, one host thread each -- how the
            # optimizer issues its configurations, optimizer.rs:220-231): the host's turn-around between trainings and the
            # short kernels at both ends of one training hide behind the other's
            try:
                import threading
                views = [tr.view(), tr.view()]
                for v in views:
                    v.set_fit_mode(mode)
                    v.train_leaves(root, leaf_kind, L_global)
                torch.cuda.synchronize()
                per = 50

                def work(v):
                    for _ in range(per):
                        v.train_leaves(root, leaf_kind, L_global)
                th = [threading.Thread(target=work, args=(v,)) for v in views]
                t0 = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (2 * per)
                out["two_in_flight"] = {"value": n_global / dt, "unit": "keys/s", "ms_per_training": dt * 1e3,
                                        "note": "throughput with two independent trainings in flight on the same resident keys; a side figure, "
                                                "not `value` (a step of `value` is one training at a time)"}
                for v in views:
                    v.close()
            except Exception as ex:
                out["two_in_flight"] = {"error": str(ex)}
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
        o_head = None
        if not args.no_cpu_baseline and args.cpu_sample > 0 and world == 1 and keys_np is not None:
            sample = keys_np[: min(args.cpu_sample, len(keys_np))]
            o_head, L_cpu, out["cpu_baseline"] = cpu_baseline(sample, args.spec, L_global, n_global)
            # ---- the three output clauses, measured for the mode `value` is quoted in
            if len(sample) == n_global and g_head is not None:
                out["parity_check"] = parity_against(o_head, g_head, f"the timed configuration itself, mode {args.mode}")
            else:
                t2 = T.Trainer(sample, device=local_rank)
                t2.set_fit_mode(mode)
                g2 = t2.train_leaves(t2.fit_root(root_kind, L_cpu), leaf_kind, L_cpu).materialize()
                out["parity_check"] = parity_against(o_head, g2, f"the CPU sample ({len(sample)} keys, {L_cpu} leaves), mode {args.mode}")
                t2.close()
        else:
            out["cpu_baseline"] = None
        # ---- the other BASELINE configurations on this GPU, as side figures of the default run
        if world == 1 and args.config == "M" and not args.no_extras and not args.no_configs:
            out["configs"] = side_configs(T, tr, local_rank, not args.no_cpu_baseline)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
