"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/rmi_hip.h declares, and the registry / spec validation (host logic) behaves like the
reference's train_model()/validate() (train/mod.rs:35-85)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rmi_amd import build, _lib
    build.build_hip()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "rmi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rmi_hip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from rmi_amd import _lib
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, f"header/binding mismatch: {declared ^ bound}"
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"


def test_abi_version(lib):
    assert lib.rmi_hip_abi_version() == 6


def test_registry_names(lib):
    # train/mod.rs:37-54
    names = ["linear", "robust_linear", "linear_spline", "cubic", "loglinear", "normal", "lognormal",
             "radix", "radix8", "radix18", "radix22", "radix26", "radix28", "bradix", "histogram"]
    ids = set()
    for n in names:
        k = lib.rmi_hip_model_from_name(n.encode())
        assert k >= 0
        assert lib.rmi_hip_model_name(k).decode() == n
        ids.add(k)
    assert len(ids) == len(names)
    assert lib.rmi_hip_model_from_name(b"nope") == -1


def test_parse_spec(lib):
    from rmi_amd import train
    assert train.parse_spec("linear,linear") == (0, 0)
    assert train.parse_spec("cubic,linear") == (2, 0)
    assert train.parse_spec("radix,linear_spline") == (3, 1)
    for spec, code in [("linear,radix", -2), ("linear,bradix", -2), ("foo,linear", -1),
                       ("linear", -12), ("linear,linear,linear", -12)]:
        with pytest.raises(train.RMIError) as e:
            train.parse_spec(spec)
        assert e.value.code == code, spec


def test_no_device_fails_loudly(lib):
    """Without a GPU the compute entry points refuse to run (no CPU fallback)."""
    if lib.rmi_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.rmi_hip_create(0, C.byref(h)) == -15
    from rmi_amd import train
    with pytest.raises(train.RMIError):
        train.Trainer()


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rmi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "rmi_oracle" not in src, f


def test_host_root_recurrence_is_the_reference_recurrence(lib, oracle):
    """The host root fit runs the recurrence with fma(a, r, a*rl) in place of a / n (one division off
    the dependent chain): bit for bit the same quotients (self-test on random and near-midpoint
    operands), and the streamed `linear` root equals the oracle's on every key set -- no GPU needed."""
    import ctypes as C
    import numpy as np
    from rmi_amd import _lib, datagen as dg
    bad = C.c_uint64(99)
    assert lib.rmi_hip_selftest_host_div(20_000_000, 5, C.byref(bad)) == 0 and bad.value == 0
    for gen, dt in [("uniform_u64", 0), ("books_u64", 0), ("dups_u64", 0), ("clustered_u64", 0), ("dups_u32", 1), ("uniform_f64", 2)]:
        keys = dg.GENERATORS[gen](300_000)
        for L in (64, 1 << 16, 300_000):
            rs = C.c_void_p()
            assert lib.rmi_hip_root_stream_begin(0, dt, len(keys), L, C.byref(rs)) == 0
            for lo in range(0, len(keys), 70_001):                       # uneven chunks
                part = np.ascontiguousarray(keys[lo:lo + 70_001])
                assert lib.rmi_hip_root_stream_push(rs, part.ctypes.data, len(part)) == 0
            m = _lib.ModelParams()
            assert lib.rmi_hip_root_stream_finish(rs, C.byref(m)) == 0
            o = oracle.fit_root("linear", keys, L)
            assert (m.p[0], m.p[1]) == (o.p[0], o.p[1]), (gen, L)


def test_host_root_target_of_radix_family(lib, oracle):
    """rmi_hip_root_target (the bucketing a multi-GPU caller plans its cuts with) for radix and both
    bradix functions (balanced_radix.rs:104-116) equals min(L-1, predict_to_int) of the oracle."""
    import ctypes as C
    import numpy as np
    from rmi_amd import _lib
    rng = np.random.default_rng(9)
    keys = rng.integers(0, 1 << 63, size=300, dtype=np.uint64) * 2 + 1
    L = 1000
    for kind, ip in [(3, (0, 10, 0, 0)), (13, (0, 10, 998, 1)), (13, (3, 12, 500, 1)), (13, (0, 10, 300, 0)),
                     (13, (0, 10, (1 << 64) - 1048, 0))]:
        m = _lib.ModelParams()
        m.kind = kind
        for i in range(4):
            m.ip[i] = ip[i]
        om = oracle.Model(kind, (0.0,) * 4, ip)
        for k in keys:
            out = C.c_uint64()
            assert lib.rmi_hip_root_target(C.byref(m), 0, int(k), L, C.byref(out)) == 0
            assert out.value == min(L - 1, om.predict_to_int(int(k))), (kind, ip, int(k))


def test_host_root_target_of_float_roots(lib, oracle):
    """rmi_hip_root_target for loglinear (linear.rs:177-180) and normal (normal.rs:81-84) roots equals
    min(L-1, predict_to_int) of the oracle: exp1 / phi are plain IEEE arithmetic on both sides."""
    import ctypes as C
    import numpy as np
    from rmi_amd import _lib, datagen as dg
    keys = dg.uniform_u64(50_000)
    L = 4096
    for name, kind in [("loglinear", 5), ("normal", 6)]:
        om = oracle.fit_root(name, keys, L)
        m = _lib.ModelParams()
        m.kind = kind
        for i in range(4):
            m.p[i] = om.p[i]
        for k in keys[::97]:
            out = C.c_uint64()
            assert lib.rmi_hip_root_target(C.byref(m), 0, int(k), L, C.byref(out)) == 0
            assert out.value == min(L - 1, om.predict_to_int(int(k))), (name, int(k))
    for kind in (7, 14):                                             # lognormal, histogram: not on the device path
        m = _lib.ModelParams()
        m.kind = kind
        assert lib.rmi_hip_root_target(C.byref(m), 0, 5, L, C.byref(C.c_uint64())) < 0


def test_struct_layout_matches_a_c_consumer(tmp_path):
    """tests/abi_check.c includes include/rmi_hip.h and prints sizeof / offsetof of every structure that crosses the boundary;
    the ctypes mirrors of rmi_amd/_lib.py must say the same (a header edit that moves a field fails here, not in a caller)."""
    import ctypes as C
    import subprocess
    from rmi_amd import _lib
    exe = str(tmp_path / "abi_check")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_check.c"), "-o", exe], check=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    mirrors = {"rmi_hip_model_params": _lib.ModelParams, "rmi_hip_shard": _lib.Shard, "rmi_hip_result": _lib.Result,
               "rmi_hip_train_config": _lib.TrainConfig}
    seen = {k: set() for k in mirrors}
    for ln in lines:
        w = ln.split()
        if not w:
            continue
        if w[0] == "abi":
            assert int(w[1]) == 6
        elif w[1] == "size":
            assert C.sizeof(mirrors[w[0]]) == int(w[2]), ln
        else:
            sname, fname = w[0].split(".")
            f = getattr(mirrors[sname], fname)
            assert (f.offset, f.size) == (int(w[1]), int(w[2])), ln
            seen[sname].add(fname)
    for sname, cls in mirrors.items():                                   # ... and the C program names every field of every mirror
        assert seen[sname] == {n for n, _ in cls._fields_ if not n.startswith("_")}, sname
