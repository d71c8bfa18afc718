"""Workload of the traffic measurement (round 4): the read-only streaming kernel (k_read_bw reads every byte of the key array
exactly once: the calibration of FETCH_SIZE for 16-byte loads per lane), then the leaf path in ONE fit mode (argv[1]: 0 exact, 1 guarded)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
from rmi_amd import train
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, L = 200_000_000, 1 << 20
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
root = tr.fit_root("linear", L, mode="fast")
tr.measure_read_bandwidth(3)
tr.set_fit_mode(mode)
for _ in range(3):
    tr.train_leaves(root, "linear", L)
tr.close()
