import sys, time
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import train
n, L = int(sys.argv[1]), int(sys.argv[2]); mode = int(sys.argv[3]); chunks = int(sys.argv[4])
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
keys = tr.download_keys()
root = tr.fit_root("linear", L)
tr.set_fit_mode(mode)
r = tr.train_leaves(root, "linear", L).materialize()
print("resident ok", flush=True)
for i in range(3):
    t0 = time.perf_counter()
    s = tr.train_streamed(keys, root, "linear", L, chunks=chunks)
    dt = time.perf_counter() - t0
    print(f"streamed {i}: {dt*1e3:.1f} ms  {n*8/dt/1e9:.1f} GB/s used={s.fit_mode_used}", flush=True)
s.materialize()
print("equal:", np.array_equal(s.leaf_starts, r.leaf_starts), np.array_equal(s.last_layer_max_l1s, r.last_layer_max_l1s), np.array_equal(s.leaf_params, r.leaf_params) if mode == 0 else "-")
