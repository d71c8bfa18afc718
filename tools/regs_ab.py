"""Timing of k_leaf_regs builds on the same box: the in-tree library and every build_ab/var/*.so (experiment macros).
usage: python tools/regs_ab.py [n L steps]   (RMI_AB_ONLY=name: that library only)"""
import glob
import os
import subprocess
import sys

import os
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "."))
n = sys.argv[1] if len(sys.argv) > 1 else "200000000"
L = sys.argv[2] if len(sys.argv) > 2 else "1048576"
steps = sys.argv[3] if len(sys.argv) > 3 else "20"
CODE = r'''
import sys, time, numpy as np
sys.path.insert(0, ".")
from rmi_amd import train
n, L, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, n)
root = tr.fit_root(0, L, mode="fast")
tr.set_profile_level(2)
acc = np.zeros(8)
for _ in range(6):
    r = tr.train_leaves(root, 0, L)
    acc += np.array(r.kernel_ns, dtype=float)
acc /= 6
tr.set_profile_level(0)
dev = 0
for _ in range(steps):
    r = tr.train_leaves(root, 0, L)
    dev += r.device_ns
print("device %.4f ms  kernels(us) %s" % (dev / steps / 1e6, [round(k / 1e3, 1) for k in acc[:4]]))
'''
libs = [("in-tree", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in sorted(glob.glob("build_ab/var/*.so"))]
only = os.environ.get("RMI_AB_ONLY")               # one library only (tools/regs_prof.sh: a profiler follows the LAST child it sees)
if only:
    libs = [(n, l) for n, l in libs if n == only]
for name, lib in libs:
    env = dict(os.environ)
    if lib:
        env["RMI_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, "-c", CODE, n, L, steps], env=env, capture_output=True, text=True, timeout=300)
    print(f"{name:16s}", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ("ERR " + out.stderr[-300:]), flush=True)
    for line in out.stderr.splitlines():
        if "cycles" in line:                       # (RG_PROF builds: the phase clocks, printed when the context is destroyed)
            print("   ", line.strip(), flush=True)
