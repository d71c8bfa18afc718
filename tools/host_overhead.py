"""Development aid (GPU box): the host's share of a training -- wall time per call of rmi_hip_train_two_layer against the
device time between its first and last event, on a problem small enough that the device work is ~20 us.
usage: python tools/host_overhead.py [n L calls]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from rmi_amd import _lib, train  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
root = tr.fit_root(0, L, mode="fast")
rc_, res = root._c(), _lib.Result()
for pl in (-1, 0, 1):
    tr.set_profile_level(pl)
    for _ in range(50):
        tr._lib.rmi_hip_train_two_layer(tr._h, C.byref(rc_), 0, L, C.byref(res))
    t0 = time.perf_counter()
    dev = 0
    for _ in range(calls):
        tr._lib.rmi_hip_train_two_layer(tr._h, C.byref(rc_), 0, L, C.byref(res))
        dev += res.device_ns
    w = (time.perf_counter() - t0) / calls
    print(f"profile level {pl}: wall {w*1e6:.1f} us per call, device {dev/calls/1e3:.1f} us, host share {w*1e6 - dev/calls/1e3:.1f} us", flush=True)
tr.close()
