"""Warm timing of the leaf path per fit mode (no torch): python tools/sigma/time_modes.py [n L steps modes gen]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import train

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
modes = [int(m) for m in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2").split(",")]
gen = sys.argv[5] if len(sys.argv) > 5 else "uniform"
spec = sys.argv[6] if len(sys.argv) > 6 else "linear,linear"
dtype = np.uint32 if (len(sys.argv) > 7 and sys.argv[7] == "u32") else np.uint64
tr = train.Trainer()
if gen == "books":
    from rmi_amd import datagen
    tr.set_keys(datagen.books_u64(n))
else:
    tr.generate_keys(gen, dtype, n)
rk, lk = train.parse_spec(spec)
root = tr.fit_root(rk, L, mode="fast" if rk in (0, 4) else "exact")
for m in modes:
    tr.set_fit_mode(m)
    tr.set_profile_level(2)
    acc = np.zeros(8)
    for _ in range(max(3, steps // 3)):
        r = tr.train_leaves(root, lk, L)
        acc += np.array(r.kernel_ns, dtype=float)
    acc /= max(3, steps // 3)
    tr.set_profile_level(0)
    t0 = time.perf_counter(); dev = 0
    for _ in range(steps):
        r = tr.train_leaves(root, lk, L); dev += r.device_ns
    wall = (time.perf_counter() - t0) / steps
    b = n * np.dtype(dtype).itemsize + 24 * L
    print(f"mode {m} (used {r.fit_mode_used}): device {dev/steps/1e6:.4f} ms  wall {wall*1e3:.4f} ms  frac(8TB/s) {b/(dev/steps*1e-9)/8e12:.3f}  "
          f"kernels(us) {[round(k/1e3,1) for k in acc[:5]]}  exact_leaves {r.exact_leaves} guard {r.guard_leaves} long {r.long_leaves}", flush=True)
tr.close()
