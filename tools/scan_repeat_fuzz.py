"""Development aid (GPU box): pipeline 5 trained THREE times on one context per random configuration -- a first training whose short form lists hundreds of
tiles (a small RMI_HIP_LONG_MIN, or skewed keys) marks the configuration, the next ones take the long-leaf instance of k_spline_scan (rmi_scan.hip,
`long_leaves`) -- every training against the oracle.  usage: python tools/scan_repeat_fuzz.py [seconds [seed]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rmi_amd import datagen as dg, train  # noqa: E402
from oracle import binding as orc  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc.build()
    gens = ["uniform_u64", "dups_u64", "books_u64", "clustered_u64", "uniform_u32", "dups_u32", "uniform_f64"]
    t0, done, bad, marked = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        gen = gens[int(rng.integers(len(gens)))]
        n = int(10 ** rng.uniform(5.8, 6.7))
        per = float(10 ** rng.uniform(1.5, 2.6))
        L = max(2, int(n / per))
        root_kind = "radix" if gen.endswith("u32") and rng.random() < 0.5 else "linear"
        os.environ["RMI_HIP_LONG_MIN"] = str(int(rng.integers(16, 200))) if rng.random() < 0.7 else "4096"
        if rng.random() < 0.5:
            os.environ["RMI_HIP_SCAN_WAVES"] = str(int(rng.integers(4, 200)))
        else:
            os.environ.pop("RMI_HIP_SCAN_WAVES", None)
        keys = dg.GENERATORS[gen](n)
        tr = train.Trainer(keys)
        try:
            root = tr.fit_root(root_kind, L)
            o = orc.train_two_layer(root_kind, "linear_spline", keys, L)
        except orc.OracleError:
            tr.close()
            continue
        first_listed = None
        for rep in range(3):
            g = tr.train_leaves(root, "linear_spline", L).materialize()
            if rep == 0:
                first_listed = int(g.long_leaves)
            ok = (g.pipeline == 5 and np.array_equal(g.leaf_starts, o.leaf_start) and np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64))
                  and np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
                  and g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx and g.model_avg_error == o.model_avg_error)
            if not ok:
                bad += 1
                print(f"BAD {gen} n={n} L={L} {root_kind} long_min={os.environ['RMI_HIP_LONG_MIN']} waves={os.environ.get('RMI_HIP_SCAN_WAVES')} training {rep}", flush=True)
        done += 1
        marked += int(first_listed is not None and first_listed > 0)
        tr.close()
    print(f"REPEAT FUZZ {done} configurations x 3 trainings ({marked} with listed leaves in the first), {bad} bad, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
