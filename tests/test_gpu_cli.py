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
                                        ("uniform_u64", "cubic,linear", 2048), ("dups_u64", "linear,cubic", 1024),
                                        ("books_u64", "radix18,linear", 8192)])
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
    res = json.loads((tmp_path / "grid.json_results").read_text())["results"]     # src/main.rs:254-257
    assert len(res) == 2 and res[0]["layers"] == "linear,linear" and res[1]["namespace"] == "g2"
    assert "namespace" in res[0] and res[0]["namespace"] is None                  # src/main.rs:207-221
    assert set(res[0]) == {"layers", "branching factor", "average error", "average error %", "average l2 error",
                           "average log2 error", "max error", "max error %", "max log2 error", "size binary search", "namespace"}
    assert (tmp_path / "g2.cpp").exists() and (tmp_path / "rmi_data" / "g2_L1_PARAMETERS").exists()


def test_optimizer_fast_profile(tmp_path, oracle, monkeypatch):
    """--optimize (src/main.rs:134-163) with the `fast` profile, whose model lists are entirely on the
    device path: every configuration on the returned front carries the oracle's statistics."""
    import json
    from rmi_amd import codegen, optimizer, train
    monkeypatch.setenv("RMI_OPTIMIZER_PROFILE", "fast")
    keys = dg.books_u64(60_000)
    tr = train.Trainer(keys)
    seen = []
    front = optimizer.find_pareto_efficient_configs(tr, 10, threads=4, progress=lambda s, r: seen.append(s))
    tr.close()
    assert 2 <= len(front) <= 10 and len(seen) > len(optimizer.first_phase_configs())
    assert front == sorted(front, key=lambda r: r.average_log2_error)
    assert not any(a.dominated_by(b) for a in front for b in front)
    for st in front[:4]:
        root, leaf = st.models.split(",")
        o = oracle.train_two_layer(root, leaf, keys, st.branching_factor)
        assert abs(o.model_avg_log2_error - st.average_log2_error) <= 1e-12 * max(1.0, abs(st.average_log2_error))
        assert st.size == codegen.rmi_size(train.parse_spec(st.models)[0], train.parse_spec(st.models)[1], st.branching_factor, True)
    # the command line writes the grid spec the reference's --param-grid mode reads back
    kfile = str(tmp_path / "opt_uint64")
    dg.write_keys(kfile, keys)
    r = subprocess.run([sys.executable, "-m", "rmi_amd.cli", kfile, "--optimize", "front.json"], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=ROOT, RMI_OPTIMIZER_PROFILE="fast"), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Models" in r.stdout and "AvgLg2" in r.stdout
    cfgs = json.loads((tmp_path / "front.json").read_text())["configs"]
    assert [c["layers"] for c in cfgs] == [s.models for s in front]
    assert [c["branching factor"] for c in cfgs] == [s.branching_factor for s in front]
    assert cfgs[0]["namespace"] == "opt_uint64_0" and cfgs[0]["binary"] is True
    r = subprocess.run([sys.executable, "-m", "rmi_amd.cli", kfile, "sized", "--max-size", "200000", "--no-code"], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=ROOT, RMI_OPTIMIZER_PROFILE="fast"), capture_output=True, text=True)
    assert r.returncode == 0 and "Found RMI config" in r.stdout, r.stdout + r.stderr


def test_views_and_leaf_passes_in_flight(oracle):
    """Trainer.view: further contexts on the same resident keys.  Leaf passes issued through them
    concurrently (optimizer.measure_rmis, in_flight > 1) return exactly what one-at-a-time returns,
    and a view trains what its parent trains, bit for bit."""
    from rmi_amd import optimizer, train
    keys = dg.dups_u64(250_000)
    tr = train.Trainer(keys)
    configs = [(f"{r},{l}", bf) for r in ("linear", "radix", "radix18", "cubic") for l in ("linear", "cubic", "linear_spline")
               for bf in (64, 4096)]
    one = optimizer.measure_rmis(tr, configs, threads=4, in_flight=1)
    four = optimizer.measure_rmis(tr, configs, threads=4, in_flight=4)
    # (the integers exactly; the average of the log2 terms to 1e-12: k_spline_scan's general form takes the tiles its short form listed in the
    #  order the list was filled -- per-wave float sums in another order are the same number to the last bit or two)
    assert len(one) == len(configs) == len(four)
    for a, b in zip(one, four):
        assert (a.models, a.branching_factor, a.size, a.max_log2_error) == (b.models, b.branching_factor, b.size, b.max_log2_error)
        assert abs(a.average_log2_error - b.average_log2_error) <= 1e-12 * max(1.0, abs(a.average_log2_error))
    v = tr.view()
    root = tr.fit_root("linear", 2048)
    a, b = tr.train_leaves(root, "linear", 2048), v.train_leaves(root, "linear", 2048)
    assert np.array_equal(a.rows, b.rows) and a.model_max_error == b.model_max_error
    o = oracle.train_two_layer("linear", "linear", keys, 2048)
    assert np.array_equal(b.leaf_params, o.leaf_params) and np.array_equal(b.last_layer_max_l1s, o.leaf_err)
    v.close()
    tr.close()


def test_bounded_cli_end_to_end(tmp_path, oracle):
    """`rmi <keys> rmi linear_spline,linear 4096 --bounded 8` (tests/cache_fix_wiki/Makefile:8): spline and
    rows equal the oracle's, the emitted lookup stays within the line for every key."""
    from tests.test_codegen import MAIN_CPP
    from rmi_amd import train
    keys = dg.books_u64(120_000)
    tr = train.Trainer(keys)
    sp = tr.cache_fix(8)
    o, osp = oracle.train_bounded("linear_spline", "linear", keys, 4096, 8)
    assert np.array_equal(sp, osp)
    g = tr.train_bounded("linear_spline,linear", 4096, 8)
    assert g.num_rmi_rows == len(osp) and g.num_data_rows == len(keys)
    assert np.array_equal(g.leaf_params, o.leaf_params) and np.array_equal(g.last_layer_max_l1s, o.leaf_err)
    tr.close()
    kfile = str(tmp_path / "b_uint64")
    dg.write_keys(kfile, keys)
    r = subprocess.run([sys.executable, "-m", "rmi_amd.cli", kfile, "rmi", "linear_spline,linear", "4096", "--bounded", "8"],
                       cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(str(tmp_path / "rmi_data" / "rmi_L2_PARAMETERS"), dtype="<u8").reshape(-1, 2), osp)
    raw = np.fromfile(str(tmp_path / "rmi_data" / "rmi_L1_PARAMETERS"), dtype="<u8").reshape(4096, 3)
    assert np.array_equal(raw[:, :2], o.leaf_params.view(np.uint64)) and np.array_equal(raw[:, 2], o.leaf_err)
    if shutil.which("g++"):
        (tmp_path / "main.cpp").write_text(MAIN_CPP.replace("NS", "rmi").replace("LKEYT", "uint64_t").replace("KEYT", "uint64_t"))
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-ffast-math", "-march=native", "-o", "a.out", "main.cpp", "rmi.cpp"],
                              cwd=str(tmp_path))
        out = subprocess.run(["./a.out", kfile, "rmi_data"], cwd=str(tmp_path), capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
    # u32 data is refused like the reference (src/main.rs:281-282)
    k32 = str(tmp_path / "c_uint32")
    dg.write_keys(k32, dg.uniform_u32(10_000))
    r = subprocess.run([sys.executable, "-m", "rmi_amd.cli", k32, "rmi", "linear,linear", "64", "--bounded", "8"],
                       cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    assert r.returncode != 0 and "u64" in (r.stdout + r.stderr)


def test_train_many_equals_single_trainings(oracle):
    """rmi_hip_train_many (optimizer.rs:220-231's par_iter as ONE call of the library): the aggregates of every configuration equal
    those of a training of its own, bit for bit, whatever runs beside it; a configuration the reference panics on reports its
    code and the others still train."""
    from rmi_amd import train
    keys = dg.books_u64(400_000)
    tr = train.Trainer(keys)
    cfgs = [("linear", "linear", 4096), ("cubic", "linear", 1024), ("radix", "linear_spline", 2048), ("linear", "cubic", 512),
            ("radix18", "linear", 8192), ("linear", "linear", 64), ("linear", "linear", 100_000), ("robust_linear", "linear", 256)]
    roots = [tr.fit_root(r, L) for r, _l, L in cfgs]
    singles = [tr.train_leaves(root, leaf, L) for root, (_r, leaf, L) in zip(roots, cfgs)]
    for in_flight in (1, 3, 8):
        many = tr.train_many([(root, leaf, L) for root, (_r, leaf, L) in zip(roots, cfgs)], in_flight=in_flight)
        assert len(many) == len(cfgs)
        for (rc, m), s_ in zip(many, singles):
            assert rc == 0
            for f in ("model_avg_error", "model_avg_l2_error", "model_avg_log2_error", "model_max_error", "model_max_error_idx",
                      "model_max_log2_error", "num_rmi_rows", "branching_factor", "models"):
                assert getattr(m, f) == getattr(s_, f), f
    # robust_linear leaves of three points or fewer are an assertion of the reference (linear.rs:248); the same root with linear
    # leaves trains: one configuration of the call reports the code, the other its result
    L = 200_000
    root = tr.fit_root("linear", L)
    res = tr.train_many([(root, "linear", L), (root, "robust_linear", L), (root, "linear_spline", L)], in_flight=3)
    t2 = tr
    for (rc, m), leaf in zip(res, ("linear", "robust_linear", "linear_spline")):
        try:
            o = oracle.train_two_layer("linear", leaf, keys, L)
            assert rc == 0 and m.model_max_error == o.model_max_error and m.model_avg_error == o.model_avg_error
        except oracle.OracleError as oe:
            assert rc == oe.code and m is None
    assert {rc for rc, _ in res} != {0}, "the test wants a configuration the reference panics on"
    tr.close()
