#!/usr/bin/env python3
"""Throughput of the metric configuration with J independent trainings in flight on ONE resident key
set (J contexts / streams, one host thread each -- how the optimizer could issue its ~100
configurations).  Not the headline bench: kernels of different trainings overlap, so per-kernel
durations are not comparable with the sequential numbers."""
import sys, time, threading
import numpy as np
import torch
sys.path.insert(0, ".")
from rmi_amd import train as T

n, L, steps = 200_000_000, 1 << 20, 24
gen = T.Trainer()
gen.generate_keys("uniform", np.uint64, n)
host = gen.download_keys()
root = gen.fit_root("linear", L)
gen.close()
dev = torch.from_numpy(host.view(np.int64)).cuda()
for J in (1, 2, 3, 1, 2, 3):
    trs = [T.Trainer() for _ in range(J)]
    for t in trs:
        t.set_keys(dev)
        t.train_leaves(root, "linear", L)          # warm-up (allocations)
    torch.cuda.synchronize()
    def work(t, k):
        for _ in range(k):
            t.train_leaves(root, "linear", L)
    th = [threading.Thread(target=work, args=(t, steps // J)) for t in trs]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {J}: {dt / (steps // J * J) * 1e3:.4f} ms per training, {n * (steps // J * J) / dt:.3e} keys/s")
    for t in trs: t.close()
