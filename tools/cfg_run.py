"""One BASELINE configuration through whatever the environment selects (GPU box): per-kernel-group times, device and wall time per step.
usage: [TAG=name] [RMI_HIP_...=..] python tools/cfg_run.py <M|C2|C3|C4s|C5|Ms|D> [dataset] [steps]
   M   200 M uniform u64, linear,linear 2^20          C2  200 M books u64, linear,linear 262144
   C3  200 M uniform u64, cubic,linear 2^20           C5  400 M u32 (uniform | dups), radix,linear_spline 2^22
   Ms  M's 1/8 shard shape: 25 M u64, 131072 leaves   C4s C4's 1/8 shard shape: 100 M u64, 262144 leaves (381 keys a leaf)
   D   200 M dups u64, linear,linear 2^20             U32 / U32r  400 M uniform u32, linear,linear / radix,linear 2^21
   S64 200 M uniform u64, linear,linear_spline 2^20
With RMI_CFG_TRACE=1 only the trainings run (no read-bandwidth kernel, few steps): the workload of a counter pass."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
train = None


def _gpu():
    """torch first: its HIP runtime must see the device before the library's does (the other way round: "No HIP GPUs are available")"""
    global train
    import torch
    torch.cuda.init()
    from rmi_amd import train as t
    train = t


CFG = {
    "M": (200_000_000, 1 << 20, "linear", "linear", "uniform", np.uint64),
    "C2": (200_000_000, 262_144, "linear", "linear", "books", np.uint64),
    "C3": (200_000_000, 1 << 20, "cubic", "linear", "uniform", np.uint64),
    "C5": (400_000_000, 1 << 22, "radix", "linear_spline", "uniform", np.uint32),
    "Ms": (25_000_000, 1 << 17, "linear", "linear", "uniform", np.uint64),
    "C4s": (100_000_000, 1 << 18, "linear", "linear", "uniform", np.uint64),
    "D": (200_000_000, 1 << 20, "linear", "linear", "dups", np.uint64),
    "U32": (400_000_000, 1 << 21, "linear", "linear", "uniform", np.uint32),      # 4-byte keys, linear leaves (190 keys a leaf)
    "U32r": (400_000_000, 1 << 21, "radix", "linear", "uniform", np.uint32),
    "S32": (400_000_000, 1 << 20, "radix", "linear_spline", "uniform", np.uint32),    # C5's keys in a quarter of its leaves (381 keys a leaf)
    "S64": (200_000_000, 1 << 20, "linear", "linear_spline", "uniform", np.uint64),   # linear_spline leaves on M's keys (k_spline_scan, 8-byte keys)
}


def mk(env):
    """a Trainer created under the environment switches `env` (they are read when the context is created)"""
    _gpu()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    tr = train.Trainer()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return tr


def setup(cfg, dataset=None):
    _gpu()
    if cfg.startswith("X"):                                  # X<keys>x<leaves>: uniform u64, linear,linear (sweeps over the keys per leaf)
        a, b = cfg[1:].split("x")
        CFG[cfg] = (int(a), int(b), "linear", "linear", "uniform", np.uint64)
    if cfg.startswith("Y:"):                                 # Y:<keys>:<leaves>:<root>:<leaf kind>:<u64|u32|f64>: any shape of the sweep (tools/sweep_shapes.py)
        _, a, b, rk, lk, dtn = cfg.split(":")
        CFG[cfg] = (int(a), int(b), rk, lk, "uniform", {"u64": np.uint64, "u32": np.uint32, "f64": np.float64}[dtn])
    n, L, root_kind, leaf, ds, dt = CFG[cfg]
    ds = dataset or ds
    tr = train.Trainer()
    if ds == "books":
        import torch
        from rmi_amd import datagen
        kt = datagen.books_u64_torch(n, device="cuda:0")
        torch.cuda.synchronize()
        tr.set_keys(kt)
    elif ds == "iid":                                        # independent uniform draws, sorted: Poisson-filled leaves (`uniform` is a jittered grid)
        import torch
        if np.dtype(dt).itemsize == 8:
            kt = torch.sort(torch.randint(0, (1 << 63) - 1, (n,), dtype=torch.int64, device="cuda:0")).values
        else:
            kt = torch.sort(torch.randint(0, (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda:0")).values
        torch.cuda.synchronize()
        tr.set_keys(kt)
    else:
        tr.generate_keys(ds, dt, n)
    # (the root's coefficients do not matter to the leaf path's speed: linear roots by the parallel sums)
    root = tr.fit_root(root_kind, L, mode="fast" if root_kind == "linear" else "exact")
    return tr, root, leaf, n, L, np.dtype(dt).itemsize


def main():
    cfg = sys.argv[1]
    dataset = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    tr, root, leaf, n, L, kb = setup(cfg, dataset)
    trace = os.environ.get("RMI_CFG_TRACE") == "1"
    if trace:
        if os.environ.get("RMI_CFG_BW", "1") == "1":
            tr.measure_read_bandwidth(3)
        for _ in range(steps if len(sys.argv) > 3 else 3):
            tr.train_leaves(root, leaf, L)
        tr.close()
        return
    first = time.perf_counter()
    r = tr.train_leaves(root, leaf, L)
    first = time.perf_counter() - first
    for _ in range(5):
        r = tr.train_leaves(root, leaf, L)
    tr.set_profile_level(2)
    acc = np.zeros(8)
    for _ in range(5):
        r = tr.train_leaves(root, leaf, L)
        acc += np.array(r.kernel_ns, dtype=float)
    acc /= 5
    tr.set_profile_level(0)
    dev = 0
    for _ in range(10):
        r = tr.train_leaves(root, leaf, L)
        dev += r.device_ns
    dev /= 10
    tr.set_profile_level(-1)
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = tr.train_leaves(root, leaf, L)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    b = n * kb + 24 * L
    print("%s %s %s: device %.4f ms  wall %.4f ms  first call %.2f ms  pipeline %s  long %d  frac(dev) %.3f  frac(wall) %.3f  groups us %s" % (
        os.environ.get("TAG", "-"), cfg, dataset or "", dev / 1e6, wall * 1e3, first * 1e3, getattr(r, "pipeline", None), int(r.long_leaves),
        b / (dev * 1e-9) / 8e12, b / wall / 8e12, [round(k / 1e3, 1) for k in acc[:5]]), flush=True)
    tr.close()


if __name__ == "__main__":
    main()
