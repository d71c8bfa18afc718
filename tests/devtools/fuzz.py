"""Randomised differential test of the one-pass modes against the CPU oracle (development aid for the GPU box):
python tests/devtools/fuzz.py [cases seed].  Mode 1: bucket table, error integers, counts and aggregates must be the oracle's.
Mode 2: bucket table and counts; error integers valid for the emitted lines; mismatches against the oracle only counted."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import datagen as dg, train
from oracle import binding as orc

BIG_LOG2N = 0.0        # python tests/devtools/fuzz.py cases seed big: keys up to 2^24.5
GENS = ["uniform_u64", "books_u64", "uniform_u32", "uniform_f64", "dups_u64", "dups_u32", "clustered_u64"]
ROOTS = ["linear", "linear_spline", "radix", "cubic"]


def run_case(rng, c=0):
    """One random configuration; returns (ok or None when the reference itself panics on it, description)."""
    import os
    gen = GENS[rng.integers(len(GENS))]
    root = ROOTS[rng.integers(len(ROOTS))]
    n = int(2 ** rng.uniform(12.1, BIG_LOG2N if BIG_LOG2N else 21.5))
    L = int(2 ** rng.uniform(3, np.log2(n / 32)))
    if root == "radix":
        L = 1 << max(3, int(np.log2(L)))
    mode = int(rng.integers(0, 3))
    streamed = bool(rng.integers(0, 2))
    leaf = "linear_spline" if rng.integers(0, 4) == 0 else "linear"      # (linear_spline: one-pass in every mode, bit for bit)
    waves = [None, "64", "1000", "100000"][rng.integers(4)]
    # (RMI_HIP_SIGMA_WAVES, the one-pass kernel's wave count, was an environment switch until round 6)
    # the leaf-lane pipeline's own switches (exact mode): leaves handed to the list kernels / to the host, the tail in-stream
    knobs = {"RMI_HIP_LONG_MIN": [None, "64", "512"][rng.integers(3)], "RMI_HIP_HOST_MIN": [None, "2000"][rng.integers(2)],
             "RMI_HIP_OPT_TAIL": [None, "0"][rng.integers(2)], "RMI_HIP_LANES_SEARCH": [None, "0"][rng.integers(2)]}
    for k, v in knobs.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    keys = dg.GENERATORS[gen](n, seed=int(rng.integers(1, 1 << 30)))
    try:
        o = orc.train_two_layer(root, leaf, keys, L, threads=2)
    except orc.OracleError:
        return None, "reference panics"
    chunks = 1
    if streamed:
        while chunks < 8 and L % (chunks * 2) == 0 and rng.integers(0, 3) > 0:
            chunks *= 2
        tr = train.Trainer()
        tr.set_fit_mode(mode)
        try:
            g_root = tr.fit_root_host(keys, root, L)
        except train.RMIError:
            tr.close()
            return None, "root not fitted on the host"
        g = tr.train_streamed(keys, g_root, leaf, L, chunks=chunks).materialize()
    else:
        tr = train.Trainer(keys)
        tr.set_fit_mode(mode)
        g_root = tr.fit_root(root, L)
        g = tr.train_leaves(g_root, leaf, L).materialize()
    tr.close()
    ok = np.array_equal(g.leaf_starts, o.leaf_start) and np.array_equal(g.leaf_counts, o.leaf_count)
    ge, oe = g.last_layer_max_l1s.astype(np.int64), o.leaf_err.astype(np.int64)
    nd = int(np.count_nonzero(ge != oe))
    if mode == 0 or leaf != "linear":
        ok = ok and nd == 0 and np.array_equal(g.leaf_params, o.leaf_params) and g.model_max_error == o.model_max_error and g.model_avg_error == o.model_avg_error
    elif mode == 1:
        ok = ok and nd == 0 and g.model_max_error == o.model_max_error and g.model_avg_error == o.model_avg_error
    else:
        ok = ok and nd <= g.guard_leaves + g.merged_leaves
        # validity of the bounds for the emitted lines
        cnt = np.diff(g.leaf_starts.astype(np.int64))[:L]
        leaf_of = np.repeat(np.arange(L), cnt)
        x = keys.astype(np.float64).astype(np.longdouble)
        f = g.leaf_params[leaf_of, 1].astype(np.longdouble) * x + g.leaf_params[leaf_of, 0].astype(np.longdouble)
        pred = np.clip(np.floor(f), 0, n).astype(np.int64)
        first = np.arange(n, dtype=np.int64); dup = np.zeros(n, bool); dup[1:] = keys[1:] == keys[:-1]; first[dup] = 0
        first = np.maximum.accumulate(first)
        mx = np.zeros(L, np.int64); np.maximum.at(mx, leaf_of, np.abs(pred - first))
        over = mx - ge
        ok = ok and int(np.count_nonzero(over > 0)) <= 2 and (over.max() <= 1)
    for k in knobs: os.environ.pop(k, None)
    kn = ",".join(f"{k[8:].lower()}={v}" for k, v in knobs.items() if v is not None)
    return ok, (f"[{kn}] " + f"{c:3d} {gen:14s} {root:13s} n={n:8d} L={L:7d} leaf={leaf} mode={mode} streamed={chunks if streamed else 0} waves={waves} used={g.fit_mode_used} exact={g.exact_leaves} "
                f"merged={g.merged_leaves} guard={g.guard_leaves} diff={nd}")


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if len(sys.argv) > 3 and sys.argv[3] == "big":
        BIG_LOG2N = 24.5
    bad = 0
    t_start = time.time()
    for c in range(cases):
        ok, desc = run_case(rng, c)
        if ok is None:
            continue
        bad += 0 if ok else 1
        print("ok " if ok else "BAD", desc, flush=True)
    print(f"{cases} cases, {bad} bad, {time.time() - t_start:.0f} s")
