"""Development aid for the GPU box: the leaf-lane pipeline (3) against the oracle on small key sets, with a dump of
what differs, then warm timings of the metric configuration per pipeline variant.
usage: python tools/lanes_check.py [check] [time [n L steps]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rmi_amd import datagen as dg, train  # noqa: E402


def mk(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    tr = train.Trainer()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return tr


VARIANTS = {
    "p3 fused search": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_LANES_FUSE": "1", "RMI_HIP_LANES_SEARCH": "1"},
    "p3 unfused search": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_LANES_FUSE": "0", "RMI_HIP_LANES_SEARCH": "1"},
    "p3 fused scan": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_LANES_FUSE": "1", "RMI_HIP_LANES_SEARCH": "0"},
    "p3 giants": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_HOST_MIN": "2000", "RMI_HIP_LONG_MIN": "512"},
}


def check():
    from oracle import binding as orc
    orc.build()
    bad_total = 0
    cases = [("uniform_u64", 300_000, 1024, "linear"), ("uniform_u64", 300_000, 16384, "linear"), ("books_u64", 300_000, 4096, "linear"),
             ("dups_u64", 300_000, 4096, "linear"), ("clustered_u64", 300_000, 1024, "linear"), ("uniform_u32", 300_000, 4096, "linear"),
             ("dups_u32", 300_000, 1024, "linear"), ("uniform_u64", 5_000, 8, "linear"), ("uniform_u64", 300_000, 4096, "cubic"),
             ("dups_u64", 200_000, 40_000, "linear"), ("uniform_u64", 1_000_000, 64, "linear"), ("uniform_u64", 300_000, 4096, "radix"),
             ("dups_u64", 300_000, 64, "linear"), ("books_u64", 300_000, 100, "linear"), ("uniform_u64", 2_000_000, 8, "linear"),
             ("dups_u32", 300_000, 32, "linear"), ("clustered_u64", 300_000, 16, "linear")]
    spl = [(g, n, L, r, "linear_spline") for g, n, L, r in
           [("uniform_u32", 300_000, 4096, "radix"), ("dups_u32", 300_000, 4096, "radix"), ("dups_u64", 300_000, 1024, "linear"),
            ("books_u64", 300_000, 256, "radix"), ("uniform_u64", 1_000_000, 16, "radix"), ("uniform_f64", 200_000, 512, "radix"),
            ("clustered_u64", 300_000, 1024, "radix"), ("dups_u32", 200_000, 60_000, "radix")]]
    cases = [c + ("linear",) for c in cases] + spl
    for name, env in VARIANTS.items():
        for gen, n, L, root, leaf in cases:
            keys = dg.GENERATORS[gen](n)
            tr = mk(env)
            tr.set_keys(keys)
            g_root = tr.fit_root(root, L)
            try:
                o = orc.train_two_layer(root, leaf, keys, L)
            except orc.OracleError as oe:
                try:
                    tr.train_leaves(g_root, leaf, L)
                    print(f"[{name}] {gen} n={n} L={L} {root},{leaf}: oracle error {oe.code}, GPU none  BAD")
                    bad_total += 1
                except train.RMIError as ge:
                    print(f"[{name}] {gen} n={n} L={L} {root}: both error {oe.code}/{ge.code}", "ok" if oe.code == ge.code else "BAD")
                tr.close()
                continue
            try:
                g = tr.train_leaves(g_root, leaf, L)
            except train.RMIError as ge:
                print(f"[{name}] {gen} n={n} L={L} {root}: GPU error {ge.code} {ge}  BAD")
                bad_total += 1
                tr.close()
                continue
            ls = np.nonzero(g.leaf_starts != o.leaf_start)[0]
            pp = np.nonzero((g.leaf_params != o.leaf_params).any(axis=1))[0]
            ee = np.nonzero(g.last_layer_max_l1s != o.leaf_err)[0]
            cc = np.nonzero(g.leaf_counts != o.leaf_count)[0]
            ok = not (len(ls) or len(pp) or len(ee) or len(cc))
            bad_total += 0 if ok else 1
            print(f"[{name}] {gen} n={n} L={L} {root},{leaf}: starts {len(ls)} params {len(pp)} errs {len(ee)} counts {len(cc)} long {g.long_leaves}", "ok" if ok else "BAD")
            if len(ls):
                print("    starts:", [(int(j), int(g.leaf_starts[j]), int(o.leaf_start[j])) for j in ls[:6]])
            if len(pp):
                for j in pp[:6]:
                    print(f"    leaf {int(j)} [{int(o.leaf_start[j])},{int(o.leaf_start[j+1])}) gpu {tuple(g.leaf_params[j])} oracle {tuple(o.leaf_params[j])}")
            if len(ee):
                print("    errs:", [(int(j), int(g.last_layer_max_l1s[j]), int(o.leaf_err[j]), int(o.leaf_start[j + 1] - o.leaf_start[j])) for j in ee[:8]])
            tr.close()
    print("CHECK", "ALL OK" if bad_total == 0 else f"{bad_total} BAD")


def timing(n, L, steps, gen="uniform", only=""):
    variants = dict(VARIANTS)
    variants["p2 exact (pass A/B)"] = {"RMI_HIP_PIPELINE": "2"}
    variants["p2 onepass guarded"] = {"RMI_HIP_PIPELINE": "2", "RMI_HIP_FIT_MODE": "1"}
    if only:
        variants = {k: v for k, v in variants.items() if only in k}
    base = None
    for name, env in variants.items():
        tr = mk(env)
        if gen == "books":
            tr.set_keys(dg.books_u64(n))
        else:
            tr.generate_keys(gen, np.uint64, n)
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
        print(f"{name:22s}: device {dev/steps/1e6:.4f} ms  wall {wall*1e3:.4f} ms  frac(8TB/s) {b/(dev/steps*1e-9)/8e12:.3f}  "
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
        gen = rest[3] if len(rest) > 3 else "uniform"
        only = rest[4] if len(rest) > 4 else ""
        timing(n, L, steps, gen, only)
