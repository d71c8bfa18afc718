"""In-tree build of the gfx950 HIP library (hipcc cross-compiles without a GPU).

Two translation units, compiled to objects under rmi_amd/build/ (git-ignored) and linked into rmi_amd/librmi_hip.so:
rmi_hip.hip (the C ABI, the host orchestration and the kernels of pipelines 1-4) and rmi_scan.hip (pipeline 5)."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
SO = os.path.join(HERE, "librmi_hip.so")
COMMON = ["rmi_kernels.hip.h", "rmi_stream.hip.h", "rmi_sigma.hip.h", "rmi_lanes.hip.h", "rmi_device.hip.h", "rmi_scan_launch.h"]
# source -> the headers it includes (besides COMMON)
UNITS = {
    "rmi_hip.hip": ["rmi_regs.hip.h", "rmi_regs_block.inc.h", "rmi_regs_replay.inc.h", "rmi_multi.inc.h", "rmi_root_host.h", "../../include/rmi_hip.h"],
    "rmi_scan.hip": ["rmi_scan.hip.h", "rmi_scan_ends.inc.h"],
}
SOURCES = list(UNITS)
HEADERS = sorted(set(COMMON + [h for hs in UNITS.values() for h in hs]))

# -ffp-contract=off: HIP's default (fast-honor-pragmas) would fuse `c += dx*(y-mean_y)` into an
# FMA and break bit parity with the reference's unfused Rust arithmetic.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in deps)


def _deps(src: str) -> list[str]:
    return [os.path.normpath(os.path.join(CSRC, f)) for f in [src] + COMMON + UNITS[src]]


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale() -> bool:
    return any(_newer(_obj(s), _deps(s)) for s in SOURCES) or _newer(SO, [_obj(s) for s in SOURCES])


def _hipcc() -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale() and os.path.exists(SO):
        return SO
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SOURCES if force or _newer(_obj(s), _deps(s))]

    def compile_one(src: str) -> None:
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [_obj(s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
