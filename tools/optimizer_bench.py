"""Wall time of the Pareto search (rmi_amd/optimizer.py) over a resident synthetic key set.
usage: python tools/optimizer_bench.py [n_keys] [threads] [profile] [exact|fast] [leaf passes in flight]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prof = sys.argv[3] if len(sys.argv) > 3 else "fast"
if prof in ("", "default"):
    os.environ.pop("RMI_OPTIMIZER_PROFILE", None)          # the reference's default lists
else:
    os.environ["RMI_OPTIMIZER_PROFILE"] = prof
from rmi_amd import optimizer, train

tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
tr.download_keys()
dev_ns, cfgs = [], []
t0 = time.perf_counter()
root_mode = sys.argv[4] if len(sys.argv) > 4 else "exact"
in_flight = int(sys.argv[5]) if len(sys.argv) > 5 else 4
front = optimizer.find_pareto_efficient_configs(tr, 10, threads=threads, root_mode=root_mode, in_flight=in_flight,
                                                progress=lambda s, r: (dev_ns.append(r.device_ns), cfgs.append((s.models, s.branching_factor))))
wall = time.perf_counter() - t0
optimizer.display_table(front)
roots = len({(m.split(",")[0], bf) for m, bf in cfgs})
print(f"keys {n}  profile {os.environ.get('RMI_OPTIMIZER_PROFILE', 'default')}  configurations trained {len(cfgs)}  distinct root fits {roots}  host threads {threads}")
print(f"wall {wall:.2f} s   device time of all leaf passes {sum(dev_ns) / 1e9:.3f} s   "
      f"(mean {sum(dev_ns) / len(dev_ns) / 1e6:.2f} ms, max {max(dev_ns) / 1e6:.2f} ms per configuration)")
slow = sorted(zip(dev_ns, cfgs), reverse=True)[:8]
print("slowest leaf passes: " + ", ".join(f"{m} {bf}: {t / 1e6:.0f} ms" for t, (m, bf) in slow))
