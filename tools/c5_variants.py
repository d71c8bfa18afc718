"""Development aid (GPU box): the C5 configuration (400 M u32 keys, radix root, linear_spline leaves, 2^22 leaves) through whatever
the environment selects; usage: TAG=name [RMI_HIP_...=..] python tools/c5_variants.py"""
import sys, time, os, numpy as np
sys.path.insert(0, ".")
from rmi_amd import train
n, L = 400_000_000, 4_194_304
tr = train.Trainer()
tr.generate_keys("uniform", np.uint32, n)
root = tr.fit_root("radix", L)
for _ in range(3): r = tr.train_leaves(root, "linear_spline", L)
tr.set_profile_level(2)
acc = np.zeros(8)
for _ in range(5):
    r = tr.train_leaves(root, "linear_spline", L); acc += np.array(r.kernel_ns, dtype=float)
acc /= 5
tr.set_profile_level(0)
dev = 0
for _ in range(10):
    r = tr.train_leaves(root, "linear_spline", L); dev += r.device_ns
print(os.environ.get("TAG"), "device %.4f ms" % (dev / 10 / 1e6), "pipeline", getattr(r, "pipeline", None), [round(k/1e3,1) for k in acc[:5]])
