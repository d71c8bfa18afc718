"""Development aid for the GPU box: pipeline 5 (k_spline_scan) against the oracle on small key sets, every case in a process of
its own (a GPU fault ends only that case), with a dump of what differs.
usage: python tools/scan_check.py            all cases
       python tools/scan_check.py one <gen> <n> <L> <root>     (the child)"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, ".")

CASES = []
for gen in ["uniform_u64", "dups_u64", "books_u64", "uniform_u32", "dups_u32", "uniform_f64", "clustered_u64"]:
    for root in ["linear", "radix", "cubic", "radix18", "bradix", "normal"]:
        if gen == "uniform_f64" and root in ("radix", "radix18", "bradix"):
            continue
        CASES.append((gen, 300_000, 4096, root))
for gen in ["uniform_u64", "dups_u64", "books_u64", "uniform_u32", "dups_u32"]:
    for n, L in [(300_000, 64), (300_001, 1000), (299_999, 3333), (300_000, 40_000), (100_000, 99_999), (50_000, 200_000), (2_000_000, 2048),
                 (1500, 7), (1, 1), (2, 2), (3, 4), (65, 1), (64, 2), (2048, 16), (2049, 16), (1024, 1024), (1025, 3)]:
        CASES.append((gen, n, L, "linear"))


def one(gen, n, L, root):
    from rmi_amd import datagen as dg, train
    from oracle import binding as orc
    orc.build()
    keys = dg.GENERATORS[gen](n)
    tr = train.Trainer(keys)
    g_root = tr.fit_root(root, L)
    try:
        o = orc.train_two_layer(root, "linear_spline", keys, L)
    except orc.OracleError as oe:
        try:
            tr.train_leaves(g_root, "linear_spline", L)
            print(f"BAD oracle error {oe.code}, GPU none")
        except train.RMIError as ge:
            print("ok (error %d)" % ge.code if ge.code == oe.code else f"BAD error codes {ge.code} vs oracle {oe.code}")
        return
    try:
        g = tr.train_leaves(g_root, "linear_spline", L).materialize()
    except train.RMIError as ge:
        print(f"BAD GPU error {ge.code} ({ge}), oracle none")
        return
    msgs = []
    if g.pipeline != 5:
        msgs.append(f"pipeline {g.pipeline}")
    for name, a, b in [("starts", g.leaf_starts, o.leaf_start), ("alpha", g.leaf_params[:, 0].view(np.uint64), o.leaf_params[:, 0].view(np.uint64)),
                       ("beta", g.leaf_params[:, 1].view(np.uint64), o.leaf_params[:, 1].view(np.uint64)), ("err", g.last_layer_max_l1s, o.leaf_err),
                       ("count", g.leaf_counts, o.leaf_count)]:
        bad = np.flatnonzero(a != b)
        if bad.size:
            j = int(bad[0])
            jj = min(j, L - 1)
            msgs.append(f"{name}: {bad.size} differ, first leaf {j}: gpu {a[j]} oracle {b[j]} (leaf keys [{o.leaf_start[jj]}, {o.leaf_start[jj + 1]}), last {int(bad[-1])})")
    rows = g.rows.view(np.uint64).reshape(L, 3)
    if not (np.array_equal(rows[:, :2], g.leaf_params.view(np.uint64)) and np.array_equal(rows[:, 2], g.last_layer_max_l1s)):
        msgs.append("rows != params/err")
    if (g.model_max_error, g.model_max_error_idx, g.model_avg_error) != (o.model_max_error, o.model_max_error_idx, o.model_avg_error):
        msgs.append(f"aggregates: max {g.model_max_error}@{g.model_max_error_idx} vs {o.model_max_error}@{o.model_max_error_idx}, avg {g.model_avg_error} vs {o.model_avg_error}")
    for nm in ("model_avg_l2_error", "model_avg_log2_error"):
        a, b = getattr(g, nm), getattr(o, nm)
        if abs(a - b) > 1e-9 * max(1.0, abs(b)):
            msgs.append(f"{nm}: {a} vs {b}")
    print("ok" if not msgs else "BAD " + "; ".join(msgs))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    nbad = 0
    for gen, n, L, root in CASES:
        tag = f"{gen} n={n} L={L} {root}"
        if flt and flt not in tag:
            continue
        r = subprocess.run([sys.executable, sys.argv[0], "one", gen, str(n), str(L), root], capture_output=True, text=True, timeout=300)
        out = r.stdout.strip().splitlines()
        line = out[-1] if out else ""
        if r.returncode != 0:
            err = [l for l in r.stderr.splitlines() if "fault" in l.lower() or "error" in l.lower()]
            line = f"BAD crashed rc={r.returncode} " + (err[0][:200] if err else r.stderr[-200:].replace("\n", " | "))
        if not line.startswith("ok"):
            nbad += 1
        print(f"{tag:55s} {line}", flush=True)
    print("bad cases:", nbad)
