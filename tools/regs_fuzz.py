"""Development aid for the GPU box: random configurations through pipeline 4 against the oracle, bit for bit (leaf boundaries,
coefficients, error integers, counts, the integer aggregates) -- key sets of every generator, 2 000 .. 3 000 000 keys, 6 .. 300 keys
per leaf on average (the written-out blocks, the rolled loop, the far lanes, the listed groups), a few seconds each.  With `long` as third
argument: 150 .. 1 000 keys per leaf (k_leaf_regs<K, LONG>: the walk behind the stash, the second trip through the ring, the tail that is
still in the ring, containers beyond 1 008 points listed), half of the configurations with LONG forced on every shape.  With `u32`: 4-byte keys,
6 .. 800 keys per leaf, linear and radix roots, the one-wave and the two-wave kernel (RMI_HIP_REGS_U32 = 1 / 2).  With `spline`: linear_spline leaves of
4-byte keys through k_spline_scan with few persistent waves (a wave takes many tiles: the batched leaf ends), 8 .. 3 000 keys per leaf.
`spline64`: the same on 8-byte keys.
`splinelong` / `splinelong64`: 300 .. 280 000 keys per spline leaf (k_spline_scan<.., FAR = 2>).
usage: python tools/regs_fuzz.py [seconds [seed [long|u32|spline|spline64|splinelong|splinelong64]]]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rmi_amd import datagen as dg, train  # noqa: E402

sys.path.insert(0, "tools")
from cfg_run import mk  # noqa: E402


def main():
    from oracle import binding as orc
    orc.build()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260927)
    gens = [g for g in dg.GENERATORS if g.endswith("u64") or g.endswith("f64")]
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    long_mode = mode == "long"
    if mode in ("u32", "spline"):
        gens = [g for g in dg.GENERATORS if g.endswith("u32")]
    spline_long = mode in ("splinelong", "splinelong64")     # leaves of 300 .. 280 000 keys: the FAR = 2 instance (one gather for the open leaf's end, eight blocks a trip)
    if mode == "splinelong":
        gens = [g for g in dg.GENERATORS if g.endswith("u32")]
    if mode in ("spline64", "splinelong", "splinelong64"):
        mode = "spline"                                      # (8-byte keys: the pending slots' end keys are read from the key array again)
    t0, done, bad, p4 = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        gen = gens[int(rng.integers(len(gens)))]
        n = int(10 ** rng.uniform(3.3, 6.48))
        per = float(10 ** (rng.uniform(2.18, 3.0) if long_mode else rng.uniform(0.8, 2.48)))
        if mode == "u32":
            per = float(10 ** rng.uniform(0.8, 2.9))
        if mode == "spline":
            per = float(10 ** rng.uniform(0.9, 3.48))
        if spline_long:
            n = int(10 ** rng.uniform(4.5, 6.6))
            per = float(10 ** rng.uniform(2.5, 5.45))
        L = max(2, int(n / per))
        root_kind = "linear" if mode not in ("u32", "spline") or gen.endswith("f64") or rng.random() < 0.5 else "radix"
        leaf_kind = "linear_spline" if mode == "spline" else "linear"
        want = 5 if mode == "spline" else 4
        env = {"RMI_HIP_REGS": "1"}
        if mode == "u32":
            env["RMI_HIP_REGS_U32"] = "1" if rng.random() < 0.35 else "2"
        if mode == "spline":
            env = {"RMI_HIP_SCAN_WAVES": str(int(rng.integers(1, 64)))} if rng.random() < 0.8 else {}
        if rng.random() < 0.3 and mode != "spline":
            env["RMI_HIP_REGS_GRID"] = str(int(rng.integers(1, 40)))
        if rng.random() < 0.3 and not long_mode and mode != "spline":
            env["RMI_HIP_REGS_MAX_AVG"] = "100000"
        if long_mode and rng.random() < 0.5:
            env["RMI_HIP_REGS_MAX_AVG"] = "0"
            env["RMI_HIP_REGS_LONG_MAX_AVG"] = "100000"
        keys = dg.GENERATORS[gen](n)
        tr = mk(env)
        tr.set_keys(keys)
        try:
            root = tr.fit_root(root_kind, L)
            o = orc.train_two_layer(root_kind, leaf_kind, keys, L)
        except orc.OracleError:
            tr.close()
            continue
        try:
            g = tr.train_leaves(root, leaf_kind, L)
        except train.RMIError as e:
            print(f"BAD {gen} n={n} L={L} {env}: GPU error {e}", flush=True)
            bad += 1
            tr.close()
            continue
        ok = (np.array_equal(g.leaf_starts, o.leaf_start) and np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64))
              and np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
              and g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx and g.model_avg_error == o.model_avg_error)
        done += 1
        p4 += int(g.pipeline == want)
        if not ok:
            bad += 1
            print(f"BAD {gen} n={n} L={L} {root_kind},{leaf_kind} {env}: pipeline {g.pipeline}", flush=True)
        tr.close()
    print(f"FUZZ {mode or 'u64'} {done} configurations ({p4} through pipeline {4 if mode != 'spline' else 5}), {bad} bad, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
