"""Pipeline 5 (rmi_amd/csrc/rmi_scan.hip.h: k_spline_scan -- bucketing scan, linear_spline end points, error pass and leaf
ends in one key-parallel read) against the oracle through the C ABI.  Bar: bucket table, error integers, counts AND
coefficients bit-identical (linear_spline.rs:13-35 has no order-dependent arithmetic), the reference's panics as the same
error codes.  Every case asserts that pipeline 5 ran."""
import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu


def _check(monkeypatch, oracle, keys, root, L, env=None, expect_pipeline=5):
    from rmi_amd import train
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    tr = train.Trainer(keys)
    g_root = tr.fit_root(root, L)
    try:
        o = oracle.train_two_layer(root, "linear_spline", keys, L)
    except oracle.OracleError as oe:
        with pytest.raises(train.RMIError) as ge:
            tr.train_leaves(g_root, "linear_spline", L)
        assert ge.value.code == oe.code
        tr.close()
        return None
    g = tr.train_leaves(g_root, "linear_spline", L)
    assert g.pipeline == expect_pipeline
    assert np.array_equal(g.leaf_starts, o.leaf_start), "bucket assignment differs"
    bad = np.flatnonzero((g.leaf_params.view(np.uint64) != o.leaf_params.view(np.uint64)).any(axis=1))
    assert bad.size == 0, f"{bad.size} coefficient rows differ, first at leaf {bad[:5]}"
    bad = np.flatnonzero(g.last_layer_max_l1s != o.leaf_err)
    assert bad.size == 0, f"{bad.size} error integers differ, first at leaf {bad[:5]}: {g.last_layer_max_l1s[bad[:5]]} vs {o.leaf_err[bad[:5]]}"
    assert np.array_equal(g.leaf_counts, o.leaf_count)
    assert g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx
    assert g.model_avg_error == o.model_avg_error
    assert abs(g.model_avg_l2_error - o.model_avg_l2_error) <= 1e-9 * max(1.0, abs(o.model_avg_l2_error))
    assert abs(g.model_avg_log2_error - o.model_avg_log2_error) <= 1e-9 * max(1.0, abs(o.model_avg_log2_error))
    rows = g.rows.view(np.uint64).reshape(L, 3)
    assert np.array_equal(rows[:, :2], g.leaf_params.view(np.uint64)) and np.array_equal(rows[:, 2], g.last_layer_max_l1s)
    tr.close()
    return g


GENS = ["uniform_u64", "books_u64", "dups_u64", "clustered_u64", "uniform_u32", "dups_u32", "uniform_f64"]
ROOTS = ["linear", "radix", "cubic", "linear_spline", "robust_linear", "radix18", "bradix", "normal", "loglinear"]


@pytest.mark.parametrize("gen", GENS)
@pytest.mark.parametrize("root", ROOTS)
def test_scan_roots_and_key_types(monkeypatch, oracle, gen, root):
    if gen == "uniform_f64" and root.startswith("radix"):
        pytest.skip("radix roots take integer keys")
    _check(monkeypatch, oracle, dg.GENERATORS[gen](300_000), root, 4096)


@pytest.mark.parametrize("gen", ["uniform_u64", "dups_u64", "books_u64", "uniform_u32", "dups_u32"])
@pytest.mark.parametrize("n,L", [
    (300_000, 64),            # leaves far longer than a tile: every leaf runs over tiles, most go to the list kernels
    (300_001, 1000), (299_999, 3333),
    (300_000, 40_000),        # 7.5 keys per leaf: more leaf starts in a tile than a batch of slots holds
    (100_000, 99_999),        # a key per leaf
    (50_000, 200_000),        # more leaves than keys: long gaps of empty leaves
    (2_000_000, 2048), (2_000_000, 1 << 17),
    (1500, 7), (1, 1), (2, 2), (3, 4), (65, 1), (64, 2), (2048, 16), (2049, 16), (1024, 1024), (1025, 3),
])
def test_scan_geometry(monkeypatch, oracle, gen, n, L):
    """Tile, slot-batch and look-ahead geometry: sizes around the tile (1 024 keys of 8 bytes, 2 048 of 4), leaves per tile
    below / above the 64 slots of a batch, leaves longer than the look-ahead and than `long_min`, gaps of empty leaves."""
    _check(monkeypatch, oracle, dg.GENERATORS[gen](n), "linear", L)


@pytest.mark.parametrize("gen", ["uniform_u64", "dups_u64", "uniform_u32"])
@pytest.mark.parametrize("env", [{"RMI_HIP_LONG_MIN": "64"}, {"RMI_HIP_LONG_MIN": "256", "RMI_HIP_OPT_TAIL": "0"}, {"RMI_HIP_SCAN_WAVES": "8"},
                                 {"RMI_HIP_SCAN_WAVES": "40"}])
def test_scan_lists_and_grids(monkeypatch, oracle, gen, env):
    """Leaves that run on behind their tile for more than long_min keys go to the list kernels (behind the synchronisation, or in the
    stream); few persistent waves: many tiles per wave."""
    _check(monkeypatch, oracle, dg.GENERATORS[gen](400_000), "linear", 700, env=env)
    _check(monkeypatch, oracle, dg.GENERATORS[gen](400_000), "radix" if gen != "uniform_f64" else "linear", 4096, env=env)


@pytest.mark.parametrize("gen,root", [("uniform_u64", "linear"), ("dups_u64", "linear"), ("books_u64", "linear"), ("uniform_f64", "linear"),
                                      ("uniform_u32", "radix"), ("dups_u32", "radix"), ("dups_u32", "linear")])
@pytest.mark.parametrize("n,L,waves", [
    (1_500_000, 3000, None), (1_500_000, 3000, "8"),     # 500 keys a leaf: FAR = 2, an open leaf ends a block or two behind the look-ahead
    (2_000_000, 301, "24"),                              # 6 600 keys a leaf: the end found by the gather's first round, eight blocks a trip with a ragged last trip
    (3_000_000, 37, None), (3_000_000, 37, "8"),         # 81 000 keys a leaf: twenty rounds of the gather; few waves: a wave meets several such leaves
    (3_000_000, 9, "16"),                                # 333 000 keys a leaf: beyond the far limit (262 144) -- the general form lists them
])
def test_scan_long_leaves(monkeypatch, oracle, gen, root, n, L, waves):
    """Leaves of hundreds to hundreds of thousands of keys (k_spline_scan<.., FAR = 2>, rmi_scan.hip.h): the open leaf's end by one gather of the
    blocks' last keys, its far keys eight blocks a trip from the key array, their duplicates found on the way (the search does not look), a prime
    number of waves per XCD in the launch -- same bits as the oracle."""
    _check(monkeypatch, oracle, dg.GENERATORS[gen](n), root, L, env={"RMI_HIP_SCAN_WAVES": waves} if waves else None)


@pytest.mark.parametrize("gen,root", [("uniform_u64", "linear"), ("dups_u64", "linear"), ("uniform_u32", "radix"), ("dups_u32", "radix"), ("uniform_f64", "linear")])
@pytest.mark.parametrize("n,L", [(400_000, 40_000), (400_000, 25_000), (300_000, 60_000)])
def test_scan_leaves_shorter_than_a_row(monkeypatch, oracle, gen, root, n, L):
    """Fewer than 1.25 rows of a lane per leaf (8-byte keys: 20 keys, 4-byte keys: 40): the launcher skips the short form -- nearly every tile has a
    lane with two leaf starts -- and runs the general form over all tiles; around that limit (10 / 16 / 5 keys a leaf) both routes give the oracle's bits."""
    _check(monkeypatch, oracle, dg.GENERATORS[gen](n), root, L)


@pytest.mark.parametrize("gen,root,n,L,env", [("uniform_u64", "linear", 2_000_000, 20_000, {"RMI_HIP_LONG_MIN": "32"}), ("dups_u64", "linear", 2_000_000, 20_000, {"RMI_HIP_LONG_MIN": "32"}),
                                              ("dups_u32", "radix", 4_000_000, 1 << 15, {"RMI_HIP_LONG_MIN": "40"}), ("books_u64", "linear", 3_000_000, 30_000, {})])
def test_scan_second_training_of_a_key_set_that_listed_many_tiles(monkeypatch, oracle, gen, root, n, L, env):
    """A training whose short form leaves hundreds of tiles to the general form (here: a tiny long_min; in the field: a skewed key set, long leaves among
    short ones) marks the configuration: its next trainings on the same context take the long-leaf instance of the short form (rmi_scan.hip, `long_leaves`),
    which keeps such leaves.  Both trainings give the oracle's bits."""
    from rmi_amd import train
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    keys = dg.GENERATORS[gen](n)
    tr = train.Trainer(keys)
    g_root = tr.fit_root(root, L)
    o = oracle.train_two_layer(root, "linear_spline", keys, L)
    for rep in range(3):
        g = tr.train_leaves(g_root, "linear_spline", L).materialize()
        assert g.pipeline == 5
        assert np.array_equal(g.leaf_starts, o.leaf_start), f"bucket assignment differs (training {rep})"
        assert np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64)), f"coefficients differ (training {rep})"
        assert np.array_equal(g.last_layer_max_l1s, o.leaf_err), f"error integers differ (training {rep})"
        assert np.array_equal(g.leaf_counts, o.leaf_count)
        assert g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx and g.model_avg_error == o.model_avg_error
    tr.close()


@pytest.mark.parametrize("dups", [False, True])
@pytest.mark.parametrize("waves", ["8", "3"])
def test_scan_batched_leaf_ends_uneven_density(monkeypatch, oracle, dups, waves):
    """4-byte keys: the short form keeps the leaf ends of several tiles pending and runs them together (rmi_scan.hip.h, `FB`).  Few persistent waves, so
    that a wave takes many tiles; keys whose density changes fivefold and back, so that tiles with ~10 leaf starts are followed by tiles with ~50: the
    pending slots are run early, and a tile whose starts do not fit behind them after all goes to the general form.  Same bits as the oracle."""
    rng = np.random.default_rng(5)
    parts = []
    for lo, hi, cnt in ((0, 1 << 30, 1_500_000), (1 << 30, 1 << 32, 900_000)):
        parts.append(rng.choice(np.arange(lo, hi, 64, dtype=np.uint64), size=cnt, replace=False) + rng.integers(0, 64, cnt).astype(np.uint64))
    keys = np.sort(np.concatenate(parts)).astype(np.uint32)
    # (dense, sparse, dense, sparse: interleave the two regions in four stripes by folding the key space)
    keys = np.sort(np.where((keys >> 29) & 1, keys ^ np.uint32(1 << 31), keys)).astype(np.uint32)
    if dups:
        keys[100_000:1_000_000:7] = keys[99_999:999_999:7]
        keys = np.sort(keys)
    g = _check(monkeypatch, oracle, keys, "linear", 30_000, env={"RMI_HIP_SCAN_WAVES": waves})
    assert g is not None


@pytest.mark.parametrize("root,params", [("cubic", (0.0, 0.0, -1e-15, 3000.0)), ("cubic", (0.0, 0.0, 1e-13, 0.0)), ("cubic", (1e-50, -3e-32, 2e-14, 5.0)),
                                         ("linear", (5000.0, -2e-16, 0.0, 0.0)), ("linear", (0.0, 1e-12, 0.0, 0.0)), ("linear", (-3.0, 2.3e-16, 0.0, 0.0))])
def test_scan_reports_the_reference_panics(monkeypatch, oracle, root, params):
    """Caller-provided roots that decrease, leave [0, L) or wiggle: the scan checks every adjacent pair of targets like
    two_layer.rs:45-50 -- the oracle's error code, or the oracle's result.  (0, 1e-12) puts one key into leaf L / 2 -- the split key,
    Q2 -- and the next key 44 leaves further on: Q4 on the borrowed point, then a leaf without a prev point.)"""
    from rmi_amd import train
    keys = dg.uniform_u64(300_000)
    L = 4096
    kind = {"linear": 0, "cubic": 2}[root]
    model = train.Model(kind, params, (0, 0, 0, 0))
    tr = train.Trainer(keys)
    try:
        o = oracle.train_two_layer(root, "linear_spline", keys, L, root=oracle.Model(kind, params, (0, 0, 0, 0)))
    except oracle.OracleError as oe:
        with pytest.raises(train.RMIError) as ge:
            tr.train_leaves(model, "linear_spline", L)
        tr.close()
        if ge.value.code != oe.code:
            # a root that is out of bounds / decreasing / splits degenerately at once: the reference panics at whichever it meets first in
            # its order of passes and keys; the kernels raise the flags in parallel and report one of the codes -- the one the scan-based
            # pipeline 2 reports
            assert {ge.value.code, oe.code} <= {-3, -4, -5}
            monkeypatch.setenv("RMI_HIP_PIPELINE", "2")
            t2 = train.Trainer(keys)
            with pytest.raises(train.RMIError) as g2:
                t2.train_leaves(model, "linear_spline", L)
            assert g2.value.code == ge.value.code
            t2.close()
        return
    g = tr.train_leaves(model, "linear_spline", L).materialize()
    assert g.pipeline == 5
    assert np.array_equal(g.leaf_starts, o.leaf_start)
    assert np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64))
    assert np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
    tr.close()


@pytest.mark.parametrize("gen,root,n,L", [("uniform_u32", "radix", 3_000_000, 1 << 15), ("dups_u32", "radix", 3_000_000, 1 << 15),
                                          ("uniform_u64", "linear", 2_000_000, 1 << 14), ("dups_u64", "linear", 2_000_000, 1 << 14),
                                          ("uniform_u64", "radix", 2_000_000, 1 << 14), ("uniform_f64", "linear", 1_000_000, 1 << 13),
                                          ("uniform_u32", "linear", 3_000_000, 1 << 16), ("dups_u32", "linear", 1_500_000, 1 << 13)])
def test_scan_ordinary_tiles(monkeypatch, oracle, gen, root, n, L):
    """The shapes of the BASELINE configurations (about 95 / 120 keys a leaf, a lane holds at most one leaf start): nearly every tile takes
    the short form; a few dozen keys a leaf more or less move tiles between the two forms."""
    _check(monkeypatch, oracle, dg.GENERATORS[gen](n), root, L)
    _check(monkeypatch, oracle, dg.GENERATORS[gen](n // 3), root, L)


@pytest.mark.parametrize("gen,root,n,L,chunks", [("uniform_u32", "radix", 150_000, 1024, 4), ("dups_u32", "radix", 150_000, 1024, 16), ("uniform_u64", "linear", 150_000, 1024, 8),
                                                 ("dups_u64", "linear", 1_500_000, 4096, 16), ("books_u64", "linear", 400_000, 512, 8), ("uniform_u32", "radix", 5_000_000, 1 << 15, 2)])
def test_scan_shards_of_a_streamed_training(oracle, gen, root, n, L, chunks):
    """rmi_hip_train_streamed cuts the key set into leaf-aligned shards, one launch each: the point in front of a shard's first leaf and the
    point behind its last are keys OUTSIDE the launch (its halo) -- the key array's values, not the filled-in positions of the LDS image."""
    from rmi_amd import train
    keys = dg.GENERATORS[gen](n)
    tr = train.Trainer()
    g = tr.train_streamed(keys, tr.fit_root_host(keys, root, L), "linear_spline", L, chunks=chunks).materialize()
    o = oracle.train_two_layer(root, "linear_spline", keys, L)
    assert g.pipeline == 5
    assert np.array_equal(g.leaf_starts, o.leaf_start)
    assert np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64))
    assert np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
    assert g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx and g.model_avg_error == o.model_avg_error
    tr.close()


@pytest.mark.parametrize("gen,leaf", [("uniform_u64", "linear_spline"), ("uniform_u32", "linear_spline"), ("dups_u32", "linear_spline"), ("uniform_u64", "linear")])
def test_shard_cut_on_a_tile_end(oracle, gen, leaf):
    """A shard whose last key is the last position of a tile of k_spline_scan (1 024 8-byte / 2 048 4-byte keys from the line the shard's
    first key lies in): the tile's last target is then the target of a key the shard OWNS, and only the launch that holds the data's last
    key may record it as the owner of the extra count (Q7, two_layer.rs:226-232) -- ADVICE round 5.  The cut is put there by dropping keys
    from the first shard under a caller-provided root (the targets of the other keys do not move)."""
    from rmi_amd import train, sharded
    keys0 = dg.GENERATORS[gen](300_000)
    L = 2048
    tile = 1024 if keys0.dtype.itemsize == 8 else 2048
    tr = train.Trainer(keys0)
    root = tr.fit_root("linear", L)
    tr.close()
    cut = sharded.Planner(lambda i: keys0[i], len(keys0), keys0.dtype, root, L).plan(2)[0].key_hi
    drop = cut % tile
    keys = np.ascontiguousarray(np.concatenate([keys0[:100], keys0[100 + drop:]]))
    plans = sharded.Planner(lambda i: keys[i], len(keys), keys.dtype, root, L).plan(2)
    assert plans[0].key_hi % tile == 0 and plans[0].read_lo == 0 and plans[0].key_hi < len(keys)
    o = oracle.train_two_layer("linear", leaf, keys, L, root=oracle.Model(root.kind, root.p, root.ip))
    starts, params, errs, counts = [], [], [], []
    for pl in plans:
        t = train.Trainer(np.ascontiguousarray(keys[pl.read_lo:pl.read_hi]))
        res = sharded.run_shard(t, pl, root, leaf)
        starts.append(res.leaf_starts[:-1].copy()); params.append(res.leaf_params.copy())
        errs.append(res.last_layer_max_l1s.copy()); counts.append(res.leaf_counts.copy())
        if leaf == "linear_spline":
            assert res.pipeline == 5
        t.close()
    assert np.array_equal(np.concatenate(starts), o.leaf_start[:-1])
    assert np.array_equal(np.concatenate(counts), o.leaf_count), "a shard that does not hold the last key claimed the extra count"
    assert np.array_equal(np.concatenate(params).view(np.uint64), o.leaf_params.view(np.uint64))
    assert np.array_equal(np.concatenate(errs), o.leaf_err)


def test_scan_repeated_trainings_lean_and_full(monkeypatch, oracle):
    """The same configuration three times on one context (the second and third launch size the general form's kernel by the first one's list),
    then with RMI_HIP_LEAN=0 (the kernel writes the coefficient / error / count arrays itself instead of leaving them to be filled from the
    rows on download): every array the same bytes, and the oracle's."""
    from rmi_amd import train
    keys = dg.dups_u32(3_000_000)
    L = 1 << 15
    o = oracle.train_two_layer("radix", "linear_spline", keys, L)
    outs = []
    for lean in ("1", "0"):
        monkeypatch.setenv("RMI_HIP_LEAN", lean)
        tr = train.Trainer(keys)
        root = tr.fit_root("radix", L)
        for _ in range(3):
            g = tr.train_leaves(root, "linear_spline", L).materialize()
            assert g.pipeline == 5
            outs.append((g.leaf_starts.copy(), g.leaf_params.copy(), g.last_layer_max_l1s.copy(), g.leaf_counts.copy(), g.rows.copy(),
                         g.model_max_error, g.model_max_error_idx, g.model_avg_error))
        tr.close()
    for t in outs:
        assert np.array_equal(t[0], o.leaf_start) and np.array_equal(t[1].view(np.uint64), o.leaf_params.view(np.uint64))
        assert np.array_equal(t[2], o.leaf_err) and np.array_equal(t[3], o.leaf_count)
        assert np.array_equal(t[4], outs[0][4]) and t[5:] == outs[0][5:]
        assert (t[5], t[6], t[7]) == (o.model_max_error, o.model_max_error_idx, o.model_avg_error)


def test_read_bandwidth_patterns_and_release_views():
    """ABI v6: the two read-only patterns both report a positive rate; the worker contexts of train_many can be released and come back."""
    from rmi_amd import train
    tr = train.Trainer(dg.uniform_u64(2_000_000))
    assert tr.measure_read_bandwidth(2, 0) > 0 and tr.measure_read_bandwidth(2, 1) > 0
    root = tr.fit_root("linear", 1024)
    cfgs = [(root, "linear", 1024), (root, "linear_spline", 1024), (root, "linear", 1024)]
    a = tr.train_many(cfgs, in_flight=3)
    tr.release_views()
    b = tr.train_many(cfgs, in_flight=3)
    assert [x[0] for x in a] == [0, 0, 0] and [x[0] for x in b] == [0, 0, 0]
    assert [x[1].model_max_error for x in a] == [x[1].model_max_error for x in b]
    tr.close()
