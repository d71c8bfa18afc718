"""Workload for the traffic measurement: the read-only streaming kernel (k_read_bw reads every byte of the key array exactly once:
the calibration of FETCH_SIZE for this access width), then the leaf path in the exact and in the guarded one-pass mode."""
import sys
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import train
n, L = 200_000_000, 1 << 20
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
root = tr.fit_root("linear", L, mode="fast")
tr.measure_read_bandwidth(3)
for mode in (0, 1):
    tr.set_fit_mode(mode)
    for _ in range(3):
        tr.train_leaves(root, "linear", L)
tr.close()
