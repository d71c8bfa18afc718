"""Development aid for the GPU box: the register-resident leaf kernel (pipeline 4, rmi_regs.hip.h) against the oracle on
small key sets, with a dump of what differs, then warm timings of the metric configuration.
usage: python tools/regs_check.py [check] [time [n L steps]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rmi_amd import datagen as dg, train  # noqa: E402

sys.path.insert(0, "tools")
from cfg_run import mk  # noqa: E402

VARIANTS = {
    "p4 regs": {"RMI_HIP_REGS": "1"},
    "p4 regs, LONG everywhere": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_MAX_AVG": "0", "RMI_HIP_REGS_LONG_MAX_AVG": "100000"},
    "p4 small grid": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_GRID": "7"},
}


def check(variants=VARIANTS):
    from oracle import binding as orc
    orc.build()
    bad_total = 0
    gens = dict(dg.GENERATORS)
    gens.update(dg.ADVERSARIAL)
    cases = [("uniform_u64", 300_000, 4096, "linear"), ("uniform_u64", 300_000, 16384, "linear"), ("uniform_u64", 1_000_000, 8192, "linear"),
             ("uniform_u64", 2_000_000, 10_000, "linear"), ("uniform_u64", 2_000_000, 10_500, "linear"), ("uniform_u64", 1_000_000, 5300, "linear"),
             ("books_u64", 300_000, 4096, "linear"), ("dups_u64", 300_000, 4096, "linear"), ("clustered_u64", 300_000, 2048, "linear"),
             ("uniform_f64", 300_000, 4096, "linear"), ("uniform_f64", 1_000_000, 6000, "linear"), ("dups_u64", 200_000, 40_000, "linear"),
             ("uniform_u64", 300_000, 4096, "radix"), ("uniform_u64", 300_000, 4096, "cubic"), ("uniform_u64", 70_000, 1000, "linear"),
             ("uniform_u64", 5_000, 64, "linear"), ("uniform_u64", 300_000, 100_000, "linear"), ("books_u64", 1_000_000, 20_000, "linear"),
             ("progression_u64", 300_000, 4096, "linear"), ("around_2_53", 300_000, 4096, "linear"), ("around_2_63", 300_000, 4096, "linear"),
             ("progression_outlier_u64", 300_000, 4096, "linear"), ("progression_f64", 300_000, 4096, "linear")]
    for name, env in variants.items():
        for gen, n, L, root in cases:
            if gen not in gens:
                print("no generator", gen)
                continue
            keys = gens[gen](n)
            tr = mk(env)
            tr.set_keys(keys)
            g_root = tr.fit_root(root, L)
            try:
                o = orc.train_two_layer(root, "linear", keys, L)
            except orc.OracleError as oe:
                try:
                    tr.train_leaves(g_root, "linear", L)
                    print(f"[{name}] {gen} n={n} L={L} {root}: oracle error {oe.code}, GPU none  BAD")
                    bad_total += 1
                except train.RMIError as ge:
                    print(f"[{name}] {gen} n={n} L={L} {root}: both error {oe.code}/{ge.code}", "ok" if oe.code == ge.code else "BAD")
                tr.close()
                continue
            try:
                g = tr.train_leaves(g_root, "linear", L)
            except train.RMIError as ge:
                print(f"[{name}] {gen} n={n} L={L} {root}: GPU error {ge.code} {ge}  BAD")
                bad_total += 1
                tr.close()
                continue
            ls = np.nonzero(g.leaf_starts != o.leaf_start)[0]
            pp = np.nonzero((g.leaf_params != o.leaf_params).any(axis=1))[0]
            ee = np.nonzero(g.last_layer_max_l1s != o.leaf_err)[0]
            cc = np.nonzero(g.leaf_counts != o.leaf_count)[0]
            agg = (g.model_max_error == o.model_max_error) and (g.model_avg_error == o.model_avg_error)
            ok = not (len(ls) or len(pp) or len(ee) or len(cc)) and agg
            bad_total += 0 if ok else 1
            print(f"[{name}] {gen} n={n} L={L} {root}: starts {len(ls)} params {len(pp)} errs {len(ee)} counts {len(cc)} agg {agg} long {g.long_leaves}",
                  "ok" if ok else "BAD", flush=True)
            if len(pp):
                for j in pp[:6]:
                    print(f"    leaf {int(j)} (group {int(j) // 64} lane {int(j) % 64}) [{int(o.leaf_start[j])},{int(o.leaf_start[j+1])}) gpu {tuple(g.leaf_params[j])} oracle {tuple(o.leaf_params[j])}")
            if len(ee):
                print("    errs:", [(int(j), int(g.last_layer_max_l1s[j]), int(o.leaf_err[j]), int(o.leaf_start[j + 1] - o.leaf_start[j])) for j in ee[:8]])
            tr.close()
    print("CHECK", "ALL OK" if bad_total == 0 else f"{bad_total} BAD", flush=True)


def timing(n, L, steps, variants):
    base = None
    for name, env in variants.items():
        tr = mk(env)
        tr.generate_keys("uniform", np.uint64, n)
        root = tr.fit_root(0, L, mode="fast")
        tr.set_profile_level(2)
        acc = np.zeros(8)
        for _ in range(8):
            r = tr.train_leaves(root, 0, L)
            acc += np.array(r.kernel_ns, dtype=float)
        acc /= 8
        tr.set_profile_level(0)
        t0 = time.perf_counter()
        dev = 0
        for _ in range(steps):
            r = tr.train_leaves(root, 0, L)
            dev += r.device_ns
        wall = (time.perf_counter() - t0) / steps
        b = n * 8 + 24 * L
        sig = (tuple(r.last_layer_max_l1s[:64]), r.model_max_error, float(r.leaf_params[:, 1].sum()))
        if base is None:
            base = sig
        print(f"{name:24s}: device {dev/steps/1e6:.4f} ms  wall {wall*1e3:.4f} ms  frac(8TB/s) {b/(dev/steps*1e-9)/8e12:.3f}  "
              f"kernels(us) {[round(k/1e3,1) for k in acc[:5]]} long {r.long_leaves} same_as_first {sig == base}", flush=True)
        tr.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or "check" in args:
        check()
    if "time" in args:
        i = args.index("time")
        rest = args[i + 1:]
        n = int(rest[0]) if len(rest) > 0 else 200_000_000
        L = int(rest[1]) if len(rest) > 1 else 1 << 20
        steps = int(rest[2]) if len(rest) > 2 else 30
        tv = {"p3 lanes": {"RMI_HIP_REGS": "0"}, "p4 regs": {"RMI_HIP_REGS": "1"}, "p4 regs, LONG everywhere": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_MAX_AVG": "0", "RMI_HIP_REGS_LONG_MAX_AVG": "100000"},
              "p4 regs grid 512": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_GRID": "512"}, "p4 regs grid 2048": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_GRID": "2048"}}
        timing(n, L, steps, tv)
