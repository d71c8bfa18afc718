"""Emission: the writer produces the reference's artefact set; the emitted
C++ compiles with the reference tests' g++ line (tests/simple_model_wiki/Makefile:10-12) and
passes the reference's acceptance loop (tests/simple_model_wiki/main.cpp:26-41) on synthetic data."""
import os
import shutil
import subprocess
import types

import numpy as np
import pytest

from rmi_amd import codegen, datagen as dg

MAIN_CPP = r'''
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <vector>
#include "NS.h"
int main(int argc, char** argv) {
  std::ifstream in(argv[1], std::ios::binary);
  uint64_t n = 0; in.read((char*)&n, 8);
  std::vector<KEYT> keys(n); in.read((char*)keys.data(), n * sizeof(KEYT));
  if (!NS::load(argv[2])) { std::printf("load failed\n"); return 2; }
  uint64_t bad = 0;
  for (uint64_t i = 0; i < n; i++) {
    size_t err = 0;
    uint64_t guess = NS::lookup((LKEYT)keys[i], &err);
    uint64_t actual = std::lower_bound(keys.begin(), keys.end(), keys[i]) - keys.begin();
    uint64_t diff = guess > actual ? guess - actual : actual - guess;
    if (diff > err) bad++;
  }
  NS::cleanup();
  std::printf("checked %llu keys, %llu outside the bound, RMI_SIZE %zu\n", (unsigned long long)n, (unsigned long long)bad, (size_t)NS::RMI_SIZE);
  return bad ? 1 : 0;
}
'''


def _as_rmi(o, n):
    """Adapter: oracle result -> the attributes codegen.output_rmi reads (a TrainedRMI look-alike)."""
    return types.SimpleNamespace(
        branching_factor=o.num_leaves, num_rmi_rows=n,
        root=types.SimpleNamespace(kind=o.root.kind, p=o.root.p, ip=o.root.ip, table=o.root.table),
        leaf_kind=o.leaf_kind, params_per_leaf=o.params_per_leaf,
        leaf_params=o.leaf_params, last_layer_max_l1s=o.leaf_err, build_time=123)


@pytest.mark.parametrize("gen,root,leaf,L,with_err", [
    ("books_u64", "linear", "linear", 1024, True),
    ("dups_u64", "cubic", "linear", 4096, True),          # tests/simple_model_wiki
    ("uniform_u32", "radix", "linear_spline", 1024, True),
    ("uniform_u64", "radix", "linear", 1024, True),       # tests/radix_model_wiki
    ("uniform_f64", "linear", "linear", 512, True),
    ("books_u64", "radix18", "linear", 2048, True),       # hint table in <ns>_L0_PARAMETERS (radix.rs:83-170)
    ("dups_u32", "radix8", "linear_spline", 256, True),   # 1 KiB table: literal array in <ns>_data.h
    ("uniform_u64", "radix22", "cubic", 512, True),
    ("uniform_u64", "bradix", "linear", 1024, True),      # bradix_clamp_high, three integer literals (balanced_radix.rs:124-152)
    ("dups_u32", "bradix", "linear_spline", 300, True),
    ("uniform_u64", "normal", "linear", 1024, True),      # ncdf + phi + exp1 (normal.rs:94-117)
    ("books_u64", "loglinear", "linear", 512, True),      # loglinear + exp1 (linear.rs:193-209)
])
def test_emitted_code_compiles_and_is_sound(oracle, tmp_path, gen, root, leaf, L, with_err):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    keys = dg.GENERATORS[gen](60_000)
    o = oracle.train_two_layer(root, leaf, keys, L)
    ns = "rmi"
    suffix = {np.dtype(np.uint64): "uint64", np.dtype(np.uint32): "uint32", np.dtype(np.float64): "f64"}[keys.dtype]
    kfile = str(tmp_path / f"keys_{suffix}")
    dg.write_keys(kfile, keys)
    key_c = "double" if keys.dtype == np.float64 else "uint64_t"
    paths = codegen.output_rmi(ns, _as_rmi(o, len(keys)), str(tmp_path / "rmi_data"), key_type=key_c,
                               include_errors=with_err, out_dir=str(tmp_path))
    # binary parameter file: L rows of (params..., err) little endian (codegen.rs:288-315; mod.rs:613-651)
    raw = np.fromfile(paths["L1_PARAMETERS"], dtype="<u8").reshape(L, o.params_per_leaf + 1)
    assert np.array_equal(raw[:, :-1], o.leaf_params.view(np.uint64))
    assert np.array_equal(raw[:, -1], o.leaf_err)
    hdr = open(paths[f"{ns}.h"]).read()
    tlen = 0 if o.root.table is None else len(o.root.table)
    assert f"const size_t RMI_SIZE = {codegen.rmi_size(o.root.kind, o.leaf_kind, L, True, tlen)};" in hdr
    if tlen * 4 > 4096:
        assert np.array_equal(np.fromfile(paths["L0_PARAMETERS"], dtype="<u4"), o.root.table)
    assert "bool load(char const* dataPath);" in hdr and "void cleanup();" in hdr and 'const char NAME[] = "rmi";' in hdr
    kt = {"uint64": "uint64_t", "uint32": "uint32_t", "f64": "double"}[suffix]
    main = MAIN_CPP.replace("NS", ns).replace("LKEYT", key_c).replace("KEYT", kt)
    (tmp_path / "main.cpp").write_text(main)
    exe = str(tmp_path / "a.out")
    subprocess.check_call(["g++", "-std=c++17", "-O3", "-ffast-math", "-march=native", "-o", exe,
                           str(tmp_path / "main.cpp"), paths[f"{ns}.cpp"]], cwd=str(tmp_path))
    out = subprocess.run([exe, kfile, str(tmp_path / "rmi_data")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("gen,root,leaf,L,line", [("books_u64", "linear_spline", "linear", 4096, 8),   # tests/cache_fix_wiki
                                                  ("dups_u64", "cubic", "linear", 768, 8),            # tests/cache_fix_osm
                                                  ("uniform_u64", "linear", "linear", 64, 32)])
def test_bounded_rmi_compiles_and_is_within_the_line(oracle, tmp_path, gen, root, leaf, L, line):
    """`--bounded line_size` (cache_fix.rs, train_bounded, codegen.rs:396-447): the emitted lookup is
    within line_size of lower_bound for every key -- the loop of tests/cache_fix_wiki/main.cpp:26-44."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    keys = dg.GENERATORS[gen](80_000)
    o, spline = oracle.train_bounded(root, leaf, keys, L, line)
    assert oracle.check_bounded_property(o, spline, line, keys) == (0, len(keys))
    rmi = _as_rmi(o, len(spline))
    rmi.cache_fix = (line, spline)
    rmi.num_data_rows = len(keys)
    kfile = str(tmp_path / "keys_uint64")
    dg.write_keys(kfile, keys)
    paths = codegen.output_rmi("rmi", rmi, str(tmp_path / "rmi_data"), key_type="uint64_t", out_dir=str(tmp_path))
    assert np.array_equal(np.fromfile(paths["L2_PARAMETERS"], dtype="<u8").reshape(-1, 2), spline)
    hdr = open(paths["rmi.h"]).read()
    assert f"const size_t RMI_SIZE = {codegen.rmi_size(o.root.kind, o.leaf_kind, L, True, 0, len(spline))};" in hdr
    assert "uint64_t lookup(uint64_t key, size_t* err);" in hdr
    (tmp_path / "main.cpp").write_text(MAIN_CPP.replace("NS", "rmi").replace("LKEYT", "uint64_t").replace("KEYT", "uint64_t"))
    exe = str(tmp_path / "a.out")
    subprocess.check_call(["g++", "-std=c++17", "-O3", "-ffast-math", "-march=native", "-o", exe,
                           str(tmp_path / "main.cpp"), paths["rmi.cpp"]], cwd=str(tmp_path))
    out = subprocess.run([exe, kfile, str(tmp_path / "rmi_data")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cache_fix_against_direct_restatement(oracle):
    """cache_fix.rs has no unit test in the reference: pin the oracle against an independent
    transcription of SplineFit / cache_fix on small inputs."""
    rng = np.random.default_rng(11)

    for trial in range(8):
        n = int(rng.integers(40, 600))
        keys = np.sort(rng.integers(2, 1 << int(rng.integers(12, 40)), size=n, dtype=np.uint64))
        if trial % 2:
            keys[5:9] = keys[5]
            keys = np.sort(keys)
        line = int(rng.choice([2, 4, 8]))
        got = oracle.cache_fix(keys, line)
        # --- direct restatement (Python floats are IEEE f64; fma emulated exactly with fractions) ---
        from fractions import Fraction

        def pred(sp, x):
            fx, fy, tx, ty = sp
            t = float(x - fx) / float(tx - fx)
            one_minus_t = 1.0 - t
            tv1 = t * float(ty)
            exact = Fraction(one_minus_t) * Fraction(float(fy)) + Fraction(tv1)     # fma: one rounding of the exact sum
            v = float(exact)
            return int(v) if v > 0 else 0

        spline, cur, pts, out = None, None, [], []
        uniq = [(int(k), i) for i, k in enumerate(keys) if i == 0 or k != keys[i - 1]]
        last_key = 0

        def add(pt):
            nonlocal spline, pts
            if spline is None:
                spline = (pt[0], pt[1], pt[0], pt[1])
                return pt
            last = spline
            prop = (last[0], last[1], pt[0], pt[1])
            pts.append((last[2], last[3]))
            if all(pred(prop, x) // line == y // line for x, y in pts):
                spline = prop
                return None
            spline = (last[2], last[3], pt[0], pt[1])
            pts = [pt]
            return (last[2], last[3])

        for key, off in uniq:
            if key - 1 != last_key:
                r = add((key - 1, off))
                if r:
                    out.append(r)
            r = add((key, off))
            if r:
                out.append(r)
            last_key = key
        out.append((spline[2], spline[3]))
        assert got.tolist() == [list(p) for p in out]


def test_c_float_formatting():
    # models/mod.rs:568-574: Rust Display (no exponent) + ".0" when there is no '.'
    assert codegen.c_float(1.0) == "1.0"
    assert codegen.c_float(-0.4999918888028674) == "-0.4999918888028674"
    assert codegen.c_float(5.5511098244514394e-17) == "0.000000000000000055511098244514394"
    assert codegen.c_float(1.8e19) == "18000000000000000000.0"
    assert float(codegen.c_float(2.2250738585072014e-308)) == 2.2250738585072014e-308
    for v in np.random.default_rng(1).standard_normal(200) * 10.0 ** np.random.default_rng(2).integers(-30, 30, 200):
        assert float(codegen.c_float(float(v))) == float(v)          # exact round trip is what matters


def test_rmi_size_readme_sample():
    # README.md:51: cubic root + 2^21 linear leaves with errors = 50331680
    assert codegen.rmi_size(codegen.CUBIC, codegen.LINEAR, 1 << 21, True) == 50331680
