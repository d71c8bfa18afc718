"""Development aid (GPU box): duplicate-heavy 8-byte keys (every group meets a duplicate: pipeline 4 lists them all) against pipeline 3;
usage: TAG=name [RMI_HIP_REGS=0|1] python tools/dups_variants.py [n L]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from rmi_amd import train

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_048_576
tr = train.Trainer()
tr.generate_keys("dups", np.uint64, n)
root = tr.fit_root("linear", L, mode="fast")
for _ in range(3):
    r = tr.train_leaves(root, "linear", L)
dev = 0
for _ in range(10):
    r = tr.train_leaves(root, "linear", L)
    dev += r.device_ns
print(os.environ.get("TAG"), "device %.4f ms" % (dev / 10 / 1e6), "pipeline", r.pipeline, "long", r.long_leaves, flush=True)
tr.close()
