"""Development aid (GPU box): the library's two read-only streaming patterns over 400 M 4-byte keys."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
from rmi_amd import train
tr = train.Trainer(); tr.generate_keys("uniform", np.uint32, 400_000_000)
for p in (0, 1, 0, 1):
    print("pattern", p, "%.0f GB/s" % max(tr.measure_read_bandwidth(10, p) for _ in range(3)), flush=True)
