"""End to end on the GPU, the way the reference's tests drive its binary (tests/*/Makefile:8-12):
train from a key file with the `rmi` command line, compile the emitted C++, check every key."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen,spec,L", [("books_u64", "linear,linear", 4096), ("dups_u32", "radix,linear_spline", 1024),
                                        ("uniform_u64", "cubic,linear", 2048), ("dups_u64", "linear,cubic", 1024)])
def test_cli_end_to_end(tmp_path, oracle, gen, spec, L):
    from tests.test_codegen import MAIN_CPP
    keys = dg.GENERATORS[gen](150_000)
    suffix = "uint64" if keys.dtype == np.uint64 else "uint32"
    kfile = str(tmp_path / f"synthetic_{suffix}")
    dg.write_keys(kfile, keys)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "rmi_amd.cli", kfile, "rmi", spec, str(L), "-d", "rmi_data", "--zero-build-time"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Max model error on model" in r.stdout
    # parameter file == oracle rows, byte for byte
    root, leaf = spec.split(",")
    o = oracle.train_two_layer(root, leaf, keys, L)
    ppl = o.leaf_params.shape[1]
    raw = np.fromfile(str(tmp_path / "rmi_data" / "rmi_L1_PARAMETERS"), dtype="<u8").reshape(L, ppl + 1)
    assert np.array_equal(raw[:, :ppl], o.leaf_params.view(np.uint64))
    assert np.array_equal(raw[:, ppl], o.leaf_err)
    assert "const uint64_t BUILD_TIME_NS = 0;" in (tmp_path / "rmi.h").read_text()
    if shutil.which("g++"):
        kt = "uint64_t" if suffix == "uint64" else "uint32_t"
        (tmp_path / "main.cpp").write_text(MAIN_CPP.replace("NS", "rmi").replace("LKEYT", "uint64_t").replace("KEYT", kt))
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-ffast-math", "-march=native", "-o", "a.out", "main.cpp", "rmi.cpp"],
                              cwd=str(tmp_path))
        out = subprocess.run(["./a.out", kfile, "rmi_data"], cwd=str(tmp_path), capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr


def test_cli_param_grid(tmp_path):
    import json
    keys = dg.uniform_u64(100_000)
    kfile = str(tmp_path / "k_uint64")
    dg.write_keys(kfile, keys)
    grid = {"configs": [{"layers": "linear,linear", "branching factor": 256},
                        {"layers": "radix,linear_spline", "branching factor": 1024, "namespace": "g2"}]}
    (tmp_path / "grid.json").write_text(json.dumps(grid))
    r = subprocess.run([sys.executable, "-m", "rmi_amd.cli", kfile, "--param-grid", "grid.json"], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "grid.json_results").read_text())
    assert len(res) == 2 and res[0]["layers"] == "linear,linear" and res[1]["namespace"] == "g2"
    assert (tmp_path / "g2.cpp").exists() and (tmp_path / "rmi_data" / "g2_L1_PARAMETERS").exists()
