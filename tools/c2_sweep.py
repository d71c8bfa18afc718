"""Development aid (GPU box): C2 (books-shaped 200 M keys, 262 144 leaves, exact) by RMI_HIP_HOST_MIN: where the host takes over the long chains."""
import os
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
from cfg_run import mk
from rmi_amd import datagen as dg

n, L = 200_000_000, 262_144
keys = dg.books_u64(n)
for hm in sys.argv[1:] or ["262144", "131072", "65536", "32768", "16384"]:
    tr = mk({"RMI_HIP_HOST_MIN": hm})
    tr.set_keys(keys)
    root = tr.fit_root(0, L, mode="fast")
    for _ in range(2):
        r = tr.train_leaves(root, 0, L)
    t0 = time.perf_counter()
    for _ in range(5):
        r = tr.train_leaves(root, 0, L)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    ls = r.leaf_starts
    longest = int(np.diff(np.append(ls, n)).max())
    print(f"host_min {hm:>7}: {ms:7.2f} ms per training, listed {r.long_leaves}, longest leaf {longest} keys, pipeline {r.pipeline}", flush=True)
    tr.close()
