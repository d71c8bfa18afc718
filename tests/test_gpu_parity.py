"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bucket assignments and per-leaf max-error integers bit-identical,
leaf coefficients within 1e-9 relative.  The device path is the exact (reference-order) mode, so
coefficients are additionally expected to be bit-identical and are checked as such.
"""
import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu

REL_TOL = 1e-9   # tolerance stated by north_star for leaf coefficients


@pytest.fixture(scope="module")
def trainer_mod():
    from rmi_amd import train
    lib = __import__("rmi_amd._lib", fromlist=["load"]).load()
    assert lib.rmi_hip_device_count() >= 1, "no HIP device visible"
    return train


# The whole matrix runs on the default pipelines (3 = leaf-lane kernels, with the register-resident kernel of pipeline 4 taking the
# configurations it applies to); the round-2 pipeline (RMI_HIP_PIPELINE=2: streaming passes; round 1's one-kernel-per-pass
# pipeline is gone) serves as the fall-back for tiny key sets and special leaves and keep a smoke set of their own.
LEGACY_SMOKE = {"test_parity_config1", "test_parity_tiny", "test_parity_degenerate_inputs", "test_parity_long_leaves",
                "test_parity_many_empty_leaves", "test_parity_f64_keys", "test_parity_chunk_geometry", "test_error_codes",
                # (the leaf kinds and roots only these pipelines serve on their own kernels: cubic and robust_linear leaves, a radix table)
                "test_parity_cubic_leaves_f64", "test_parity_robust_linear_leaves", "test_radix_table_wider_than_the_key_range"}


def pytest_generate_tests(metafunc):
    if "pipeline" in metafunc.fixturenames:
        params = ["3", "2"] if metafunc.function.__name__ in LEGACY_SMOKE else ["3"]
        metafunc.parametrize("pipeline", params, ids=[f"pipeline{p}" for p in params], indirect=True)


@pytest.fixture(autouse=True)
def pipeline(request, monkeypatch):
    """Every test of this module under RMI_HIP_PIPELINE = its parameter (see LEGACY_SMOKE)."""
    monkeypatch.setenv("RMI_HIP_PIPELINE", request.param)
    return request.param


def _compare(trainer_mod, oracle, keys, root, leaf, L, exact_root=True):
    tr = trainer_mod.Trainer(keys)
    o_root = oracle.fit_root(root, keys, L)
    g_root = tr.fit_root(root, L)
    assert g_root.p == o_root.p and g_root.ip == o_root.ip, f"root params differ: {g_root} vs {o_root}"
    assert (g_root.table is None) == (o_root.table is None)
    if o_root.table is not None:
        assert np.array_equal(g_root.table, o_root.table), "hint table differs"
    try:
        o = oracle.train_two_layer(root, leaf, keys, L)
    except oracle.OracleError as oe:
        # where the reference panics, the ABI must report the matching code
        with pytest.raises(trainer_mod.RMIError) as ge:
            tr.train_leaves(g_root, leaf, L)
        assert ge.value.code == oe.code
        tr.close()
        return None, None
    g = tr.train_leaves(g_root, leaf, L)
    assert np.array_equal(g.leaf_starts, o.leaf_start), "bucket assignment differs"
    gp, op = g.leaf_params, o.leaf_params
    denom = np.maximum(np.abs(op), 1e-300)
    rel = np.abs(gp - op) / denom
    rel[op == gp] = 0.0
    assert float(rel.max()) <= REL_TOL, f"leaf coefficient rel err {rel.max()}"
    assert np.array_equal(gp, op), f"exact mode: {np.count_nonzero(gp != op)} coefficient words differ"
    assert np.array_equal(g.last_layer_max_l1s, o.leaf_err), \
        f"{np.count_nonzero(g.last_layer_max_l1s != o.leaf_err)} max-error ints differ"
    assert np.array_equal(g.leaf_counts, o.leaf_count)
    assert g.model_max_error == o.model_max_error
    assert g.model_max_error_idx == o.model_max_error_idx
    assert g.model_avg_error == o.model_avg_error
    assert abs(g.model_avg_l2_error - o.model_avg_l2_error) <= 1e-9 * max(1.0, abs(o.model_avg_l2_error))
    assert abs(g.model_avg_log2_error - o.model_avg_log2_error) <= 1e-9 * max(1.0, abs(o.model_avg_log2_error))
    # rows == the reference's L1_PARAMETERS image: (alpha, beta, err) per leaf, little endian
    rows = g.rows.view(np.uint64).reshape(L, g.params_per_leaf + 1)
    assert np.array_equal(rows[:, :-1], gp.view(np.uint64))
    assert np.array_equal(rows[:, -1], g.last_layer_max_l1s)
    tr.close()
    return g, o


GENS = ["uniform_u64", "books_u64", "dups_u64", "clustered_u64", "uniform_u32", "dups_u32"]


@pytest.mark.parametrize("gen", GENS)
@pytest.mark.parametrize("root,leaf,L", [
    ("linear", "linear", 1024),
    ("linear", "linear", 16384),
    ("cubic", "linear", 4096),
    ("radix", "linear_spline", 8192),
    ("radix", "linear", 1024),
    ("linear_spline", "linear", 4096),
    ("robust_linear", "linear", 4096),
    ("linear", "linear_spline", 4096),
    ("radix18", "linear", 4096),
    ("radix8", "linear_spline", 200),
    ("radix22", "cubic", 32768),
    ("bradix", "linear", 4096),
    ("bradix", "linear_spline", 1000),
    ("normal", "linear", 4096),
    ("normal", "linear_spline", 512),
    ("loglinear", "linear", 2048),
])
def test_parity_small(trainer_mod, oracle, gen, root, leaf, L):
    keys = dg.GENERATORS[gen](300_000)
    _compare(trainer_mod, oracle, keys, root, leaf, L)


@pytest.mark.parametrize("root,leaf,L", [("linear", "linear", 2048), ("linear", "linear_spline", 2048),
                                         ("cubic", "linear", 1024), ("linear_spline", "linear", 512),
                                         ("normal", "linear", 1024), ("loglinear", "linear", 256), ("bradix", "linear", 512)])
def test_parity_f64_keys(trainer_mod, oracle, root, leaf, L):
    """f64 key files (src/load.rs:71-95): as_float is the identity, +-epsilon widening (models/mod.rs:101-111)."""
    keys = dg.uniform_f64(200_000)
    assert keys.dtype == np.float64 and (keys[1:] > keys[:-1]).all()
    _compare(trainer_mod, oracle, keys, root, leaf, L)


@pytest.mark.parametrize("gen", ["uniform_u64", "books_u64", "dups_u64", "clustered_u64", "dups_u32"])
@pytest.mark.parametrize("root,L", [("linear", 4096), ("cubic", 1024), ("radix", 8192), ("linear", 40_000), ("linear", 8),
                                    ("linear", 150)])
def test_parity_cubic_leaves(trainer_mod, oracle, gen, root, L):
    """leaf = cubic (cubic_spline.rs:108-136 on every container; SURVEY 8a row a8''): 4 coefficients per
    leaf, rows of 40 bytes, the cube of the key range from the host's libm like the reference's."""
    keys = dg.GENERATORS[gen](200_000)
    g, o = _compare(trainer_mod, oracle, keys, root, "cubic", L)
    if g is not None:
        assert g.params_per_leaf == 4 and g.rows.size == L * 40


def test_parity_cubic_leaves_f64(trainer_mod, oracle):
    keys = dg.uniform_f64(100_000)
    _compare(trainer_mod, oracle, keys, "linear", "cubic", 2048)
    _compare(trainer_mod, oracle, keys, "cubic", "cubic", 256)


@pytest.mark.parametrize("gen,root,L", [("uniform_u64", "linear", 512), ("dups_u64", "robust_linear", 2048),
                                        ("books_u64", "linear", 64), ("dups_u32", "radix", 1024),
                                        ("books_u64", "linear", 20_000)])
def test_parity_robust_linear_leaves(trainer_mod, oracle, gen, root, L):
    """leaf = robust_linear (linear.rs:239-260 on every container): 0.01 % tails trimmed; containers of
    fewer than 4 points trip the reference's assert -> RMI_ERR_ROBUST_TOO_SMALL (last case)."""
    keys = dg.GENERATORS[gen](250_000)
    _compare(trainer_mod, oracle, keys, root, "robust_linear", L)


def test_parity_config1(trainer_mod, oracle):
    """BASELINE config 1: linear,linear 1024 on 1M synthetic sorted uint64."""
    keys = dg.uniform_u64(1_000_000)
    g, o = _compare(trainer_mod, oracle, keys, "linear", "linear", 1024)
    bad, _ = oracle.check_lookup_property(o, keys)
    assert bad == 0


def test_parity_many_empty_leaves(trainer_mod, oracle):
    """More leaves than keys: long runs of empty leaves (suffix fill + constant models)."""
    keys = dg.books_u64(20_000)
    _compare(trainer_mod, oracle, keys, "linear", "linear", 65536)
    _compare(trainer_mod, oracle, keys, "radix", "linear_spline", 1 << 18)


@pytest.mark.parametrize("threads,min_chunk", [(1 << 20, 16), (4096, 64), (64, 64)])
def test_parity_chunk_geometry(trainer_mod, oracle, monkeypatch, threads, min_chunk):
    """Pass A must give identical results for any chunking (leaves far smaller / larger than a chunk)."""
    monkeypatch.setenv("RMI_HIP_FIT_THREADS", str(threads))
    monkeypatch.setenv("RMI_HIP_FIT_MIN_CHUNK", str(min_chunk))
    for gen in ("books_u64", "dups_u64"):
        keys = dg.GENERATORS[gen](150_000)
        _compare(trainer_mod, oracle, keys, "linear", "linear", 512)
        _compare(trainer_mod, oracle, keys, "linear", "linear", 50_000)


@pytest.mark.parametrize("gen", ["uniform_u64", "books_u64", "dups_u64", "dups_u32"])
def test_parity_long_leaves(trainer_mod, oracle, gen):
    """Leaves far longer than pass A's reciprocal table (1024 counts): handed over to k_fit_long."""
    keys = dg.GENERATORS[gen](400_000)
    for L in (2, 16, 128, 300):
        _compare(trainer_mod, oracle, keys, "linear", "linear", L)
    _compare(trainer_mod, oracle, keys, "linear", "linear_spline", 16)
    _compare(trainer_mod, oracle, keys, "cubic", "linear", 64)


@pytest.mark.parametrize("name", ["all_equal", "two_keys", "one_key", "two_values", "max_key", "run_over_split"])
def test_parity_degenerate_inputs(trainer_mod, oracle, name):
    """Inputs on which the reference mostly panics (one leaf swallows everything, the 2-way split
    degenerates, num_bits asserts): the same results or the matching error code."""
    keys = {
        "all_equal": np.full(1000, 77, dtype=np.uint64),
        "two_keys": np.array([5, 9], dtype=np.uint64),
        "one_key": np.array([5], dtype=np.uint64),
        "two_values": np.array([3] * 500 + [900] * 500, dtype=np.uint64),
        "max_key": np.array([1, 2, 3, (1 << 64) - 2], dtype=np.uint64),
        "run_over_split": np.sort(np.concatenate([np.arange(1, 400, dtype=np.uint64), np.full(300, 400, dtype=np.uint64),
                                                  np.arange(401, 800, dtype=np.uint64)])),
    }[name]
    for root, leaf in [("linear", "linear"), ("radix", "linear"), ("cubic", "linear_spline"), ("linear", "cubic"),
                       ("radix8", "linear"), ("bradix", "linear")]:
        for L in (2, 8, 64):
            try:
                oracle.fit_root(root, keys, L)
            except oracle.OracleError:
                continue                                   # the root itself is undefined here (e.g. num_bits); covered by test_error_codes
            _compare(trainer_mod, oracle, keys, root, leaf, L)


def test_parity_tiny(trainer_mod, oracle):
    keys = np.array([10, 11, 12, 20, 21, 30, 30, 31, 40, 41, 42, 50], dtype=np.uint64)
    for L in (2, 3, 4, 7):
        _compare(trainer_mod, oracle, keys, "linear", "linear", L)
        _compare(trainer_mod, oracle, keys, "linear", "linear_spline", L)


def test_division_by_count_is_ieee(trainer_mod):
    """The table-reciprocal division of the SLR step == IEEE division, bit for bit (random and
    near-midpoint numerators, every count of the table)."""
    import ctypes as C
    tr = trainer_mod.Trainer()
    bad = C.c_uint64(123)
    rc = tr._lib.rmi_hip_selftest_div(tr._h, 4_000_000_000, 7, C.byref(bad))
    assert rc == 0 and bad.value == 0, f"{bad.value} mismatching quotients"
    tr.close()


def test_computed_reciprocal_is_ieee(trainer_mod):
    """Counts beyond the table: the computed reciprocal == 1.0 / n bit for bit, for every n below
    2^32 and for ranges up to the 2^40 the division proof covers."""
    import ctypes as C
    tr = trainer_mod.Trainer()
    for lo, hi in [(1, 1 << 32), ((1 << 36) - (1 << 28), (1 << 36) + (1 << 28)), ((1 << 40) - (1 << 30), 1 << 40)]:
        bad = C.c_uint64(123)
        rc = tr._lib.rmi_hip_selftest_recip(tr._h, lo, hi, C.byref(bad))
        assert rc == 0 and bad.value == 0, f"{bad.value} mismatching reciprocals in [{lo}, {hi})"
    tr.close()


def test_radix_table_wider_than_the_key_range(trainer_mod, oracle):
    """prefix + table_bits > 64: the slot shift of radix.rs:98-99 bottoms out at 0 and the slot is the
    key's low bits; and a table fitted for more leaves than keys (scale > 1)."""
    rng = np.random.default_rng(3)
    keys = np.unique(rng.integers(1, 1 << 20, size=60_000, dtype=np.uint64))          # 44 common leading bits
    _compare(trainer_mod, oracle, keys, "radix22", "linear", 512)
    _compare(trainer_mod, oracle, keys, "radix18", "linear_spline", 1 << 17)


@pytest.mark.parametrize("gen", ["uniform_u64", "books_u64", "dups_u64", "clustered_u64", "dups_u32", "uniform_f64"])
def test_fast_root_mode(trainer_mod, oracle, gen):
    """Opt-in fast root (parallel sums): the same line as the exact fit to 1e-10, and the
    leaf path trains exactly the RMI of THAT root -- equal to the oracle given the same root, sound
    for every key."""
    keys = dg.GENERATORS[gen](300_000)
    tr = trainer_mod.Trainer(keys)
    for root_name in ("linear", "robust_linear"):
        L = 2048
        exact = tr.fit_root(root_name, L)
        fast = tr.fit_root(root_name, L, mode="fast")
        # compare predictions over the key range (the intercept of a line far from the origin is ill-conditioned by itself)
        x = np.array([float(keys[0]), float(keys[len(keys) // 2]), float(keys[-1])])
        pe = exact.p[0] + exact.p[1] * x
        pf = fast.p[0] + fast.p[1] * x
        if gen != "clustered_u64":
            assert np.all(np.abs(pe - pf) <= 1e-6 * L), (gen, root_name, exact, fast)
            assert abs(fast.p[1] - exact.p[1]) <= 1e-9 * abs(exact.p[1])
        else:
            # keys near 2^62 with gaps < 2^12: f64(key) has a resolution of 1024 there and the
            # reference's running mean is quantised to it -- its own line is off by ~0.2 %; the
            # fast fit works on x - x0 and is the better conditioned of the two.  Same shape only.
            assert np.all(np.abs(pe - pf) <= 0.01 * L) and abs(fast.p[1] - exact.p[1]) <= 0.01 * abs(exact.p[1])
        g = tr.train_leaves(fast, "linear", L)
        o_root = oracle.Model(fast.kind, fast.p, fast.ip)
        o = oracle.train_two_layer(root_name, "linear", keys, L, root=o_root)
        assert np.array_equal(g.leaf_params, o.leaf_params) and np.array_equal(g.last_layer_max_l1s, o.leaf_err)
        assert oracle.check_lookup_property(o, keys) == (0, len(keys))
    tr.close()


def test_root_fit_from_device_resident_keys(trainer_mod, oracle):
    """Keys that exist in HBM only: `radix` and `linear_spline` roots are fitted from the handful of
    keys they depend on (no download), the others after one download; all equal the oracle's."""
    for gen, dt in [("uniform", np.uint64), ("dups", np.uint64), ("dups", np.uint32)]:
        tr = trainer_mod.Trainer()
        tr.generate_keys(gen, dt, 300_000)
        assert tr._host_keys is None
        roots = {name: tr.fit_root(name, 4096) for name in ("radix", "linear_spline")}
        assert tr._host_keys is None                       # still no host copy
        keys = tr.download_keys().copy()
        tr._host_keys = None
        roots["linear"] = tr.fit_root("linear", 4096)
        for name, g in roots.items():
            o = oracle.fit_root(name, keys, 4096)
            assert g.p == o.p and g.ip == o.ip, (gen, name, g, o)
        tr.close()


def test_device_generators_match_numpy(trainer_mod):
    for gen, dt, ref in [("uniform", np.uint64, dg.uniform_u64), ("dups", np.uint64, dg.dups_u64),
                         ("uniform", np.uint32, dg.uniform_u32), ("dups", np.uint32, dg.dups_u32)]:
        tr = trainer_mod.Trainer()
        tr.generate_keys(gen, dt, 1_000_003)
        assert np.array_equal(tr.download_keys(), ref(1_000_003))
        tr.close()
    tr = trainer_mod.Trainer()
    tr.generate_keys("uniform", np.uint64, 10_000_000, start=777_777, count=1234)
    assert np.array_equal(tr.download_keys(), dg.uniform_u64(10_000_000, start=777_777, count=1234))
    tr.close()


def test_error_codes(trainer_mod, oracle):
    keys = dg.uniform_u64(10_000)
    tr = trainer_mod.Trainer(keys)
    bad_root = trainer_mod.Model(0, (15.0, -1e-18, 0.0, 0.0), (0, 0))     # decreasing -> non monotone
    with pytest.raises(trainer_mod.RMIError) as e:
        tr.train_leaves(bad_root, "linear", 16)
    assert e.value.code in (-3, -4)
    with pytest.raises(trainer_mod.RMIError) as e:
        tr.train("linear,linear", 1)          # L=1: split_idx == 0 (two_layer.rs:27)
    assert e.value.code == -4
    with pytest.raises(trainer_mod.RMIError) as e:
        tr.train("linear,radix", 16)
    assert e.value.code == -2
    with pytest.raises(trainer_mod.RMIError) as e:
        tr.train("linear,linear,linear", 16)
    assert e.value.code == -12
    tr.close()
