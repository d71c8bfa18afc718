"""Where the wall time of a step goes beyond the device time: raw C-ABI call vs the Python wrapper.
python tools/walltime.py [n L steps]"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import train, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
root = tr.fit_root(0, L, mode="fast")
tr.set_fit_mode(1)
lib = tr._lib
res = _lib.Result()
rc_ = root._c()
for _ in range(20):
    lib.rmi_hip_train_two_layer(tr._h, C.byref(rc_), 0, L, C.byref(res))
t0 = time.perf_counter(); dev = 0
for _ in range(steps):
    lib.rmi_hip_train_two_layer(tr._h, C.byref(rc_), 0, L, C.byref(res)); dev += res.device_ns
raw = (time.perf_counter() - t0) / steps
t0 = time.perf_counter(); dev2 = 0
for _ in range(steps):
    r = tr.train_leaves(root, 0, L); dev2 += r.device_ns
wrap = (time.perf_counter() - t0) / steps
print(f"device {dev/steps/1e3:.1f} us | raw C-ABI call {raw*1e6:.1f} us (+{raw*1e6 - dev/steps/1e3:.1f}) | Python wrapper {wrap*1e6:.1f} us (+{wrap*1e6 - dev2/steps/1e3:.1f})")
tr.close()
