"""Where does the leaf path fall off its roofline?  One GPU, resident keys, every (root, leaf kind) the dispatch table has a fast pipeline for,
over the numbers of leaves the reference's optimizer walks (optimizer.rs:43-58: 2^6 .. 2^25): ms per training (wall, synchronisation included),
the pipeline that ran, B_leaf over the wall time as a fraction of 8 TB/s.
usage: python tools/sweep_shapes.py [u64|u32|f64|dups64|dups32|books|iid64|iid32]... [L<lo>-<hi>] [steps]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch  # noqa: E402  (torch first: its HIP runtime must see the device before the library's does)

torch.cuda.init()
from rmi_amd import train  # noqa: E402


def timed(tr, root, leaf, L, steps):
    r = tr.train_leaves(root, leaf, L)
    r = tr.train_leaves(root, leaf, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = tr.train_leaves(root, leaf, L)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r


SETS = {
    "u64": ("uniform", np.uint64, 200_000_000, ("linear", "radix", "cubic")),
    "u32": ("uniform", np.uint32, 400_000_000, ("linear", "radix")),
    "f64": ("f64", np.float64, 200_000_000, ("linear",)),
    "dups64": ("dups", np.uint64, 200_000_000, ("linear",)),
    "dups32": ("dups", np.uint32, 400_000_000, ("radix",)),
    "books": ("books", np.uint64, 200_000_000, ("linear",)),
    # independent uniform draws, sorted: Poisson-filled leaves (the `uniform` generator is a jittered grid: evenly filled ones)
    "iid64": ("iid", np.uint64, 200_000_000, ("linear",)),
    "iid32": ("iid", np.uint32, 400_000_000, ("radix",)),
}


def main():
    args = [a for a in sys.argv[1:] if not a.isdigit()] or ["u64", "u32"]
    steps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 10)
    lo, hi = 10, 25
    for a in sys.argv[1:]:
        if a.startswith("L"):                                 # L<lo>-<hi>: the range of log2(leaves)
            lo, hi = (int(x) for x in a[1:].split("-"))
    args = [a for a in args if not a.startswith("L")] or ["u64", "u32"]
    for name in args:
        ds, dt, n, roots = SETS[name]
        tr = train.Trainer()
        if ds == "books":
            from rmi_amd import datagen
            kt = datagen.books_u64_torch(n, device="cuda:0")
            torch.cuda.synchronize()
            tr.set_keys(kt)
        elif ds == "iid":
            if dt == np.uint64:
                kt = torch.sort(torch.randint(0, (1 << 63) - 1, (n,), dtype=torch.int64, device="cuda:0")).values
            else:
                kt = torch.sort(torch.randint(0, (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda:0")).values
            torch.cuda.synchronize()
            tr.set_keys(kt)
        elif ds == "f64":
            kt = torch.sort(torch.rand(n, dtype=torch.float64, device="cuda:0") * 1e12).values
            torch.cuda.synchronize()
            tr.set_keys(kt)
        else:
            tr.generate_keys(ds, dt, n)
        kb = np.dtype(dt).itemsize
        for root_kind in roots:
            for leaf, leaf_name in ((0, "linear"), (1, "linear_spline")):
                for lg in range(lo, hi + 1):
                    L = 1 << lg
                    try:
                        root = tr.fit_root(root_kind, L, mode="fast" if root_kind == "linear" else "exact")
                        w, r = timed(tr, root, leaf, L, steps if lg > 12 else 3)
                        b = n * kb + 24 * L
                        print("%-7s %-6s %-13s 2^%-2d %8.1f keys/leaf  %8.3f ms  pipeline %s  listed %d  frac(wall) %.3f" % (
                            name, root_kind, leaf_name, lg, n / L, w * 1e3, getattr(r, "pipeline", None), int(r.long_leaves), b / w / 8e12), flush=True)
                    except Exception as ex:
                        print("%-7s %-6s %-13s 2^%-2d error: %s" % (name, root_kind, leaf_name, lg, str(ex)[:120]), flush=True)
        tr.close()


if __name__ == "__main__":
    main()
