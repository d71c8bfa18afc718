"""In-tree build of the gfx950 HIP library (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librmi_hip.so")
SOURCES = ["rmi_hip.hip"]
HEADERS = ["rmi_kernels.hip.h", "rmi_stream.hip.h", "rmi_sigma.hip.h", "rmi_lanes.hip.h", "rmi_regs.hip.h", "rmi_regs_block.inc.h", "rmi_multi.inc.h", "rmi_device.hip.h", "rmi_root_host.h", "../../include/rmi_hip.h"]

# -ffp-contract=off: HIP's default (fast-honor-pragmas) would fuse `c += dx*(y-mean_y)` into an
# FMA and break bit parity with the reference's unfused Rust arithmetic.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-ldl"]


def _stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    for f in SOURCES + HEADERS:
        p = os.path.normpath(os.path.join(CSRC, f))
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", SO] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
