"""GPU parity of the ONE-PASS leaf path (rmi_hip_set_fit_mode: sufficient statistics from LDS, one HBM
read of the keys; rmi_amd/csrc/rmi_sigma.hip.h) against the CPU oracle.

Bar: bucket assignments, per-leaf error integers, counts and aggregates bit-identical in the guarded
mode (1); coefficients are those of the same least-squares line, not the reference's bits: both are
roundings of it, and the reference's own recurrence (linear.rs:24-34) carries a noise of about
n u X / sigma_x relative in the slope (2.3e-9 at most on 200 M uniform u64 keys, where 97.5 % of the
leaves agree to 1e-9) -- asserted here as: slope within 1e-8 relative, intercept within 1e-8 of the size
of its two terms, and at least 95 % of the leaves within the north_star's 1e-9.  Mode 2 (no exact re-fit
of guard-flagged leaves): error integers may differ by one, in at most `guard_leaves` leaves.
"""
import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    from rmi_amd import train
    lib = __import__("rmi_amd._lib", fromlist=["load"]).load()
    assert lib.rmi_hip_device_count() >= 1, "no HIP device visible"
    return train


def _run(T, oracle, keys, root, L, mode, leaf="linear"):
    tr = T.Trainer(keys)
    tr.set_fit_mode(mode)
    g_root = tr.fit_root(root, L)
    o_root = oracle.fit_root(root, keys, L)
    assert g_root.p == o_root.p and g_root.ip == o_root.ip
    o = oracle.train_two_layer(root, leaf, keys, L, threads=2)
    g = tr.train_leaves(g_root, leaf, L).materialize()
    tr.close()
    return g, o


def _check(g, o, keys, mode, expect_used=True):
    L = o.num_leaves
    assert g.fit_mode_used == (mode if expect_used else 0)
    assert np.array_equal(g.leaf_starts, o.leaf_start), "bucket assignment differs"
    assert np.array_equal(g.leaf_counts, o.leaf_count)
    ge, oe = g.last_layer_max_l1s, o.leaf_err
    gp, op = g.leaf_params, o.leaf_params
    if mode == 1 or not expect_used:
        assert np.array_equal(ge, oe), f"{np.count_nonzero(ge != oe)} max-error ints differ"
        assert g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx
        assert g.model_avg_error == o.model_avg_error
        assert abs(g.model_avg_l2_error - o.model_avg_l2_error) <= 1e-9 * max(1.0, abs(o.model_avg_l2_error))
    else:
        diff = np.abs(ge.astype(np.int64) - oe.astype(np.int64))
        assert diff.max() <= 1 and np.count_nonzero(diff) <= g.guard_leaves, (diff.max(), np.count_nonzero(diff), g.guard_leaves)
    # the same line: slope, and intercept relative to the size of its two terms mean_y and beta * mean_x
    nonempty = o.leaf_start[1:] > o.leaf_start[:-1]
    xs = keys.astype(np.float64)
    xend = xs[np.minimum(o.leaf_start[1:], len(keys) - 1).astype(np.int64)]
    with np.errstate(all="ignore"):
        relb = np.abs(gp[:, 1] - op[:, 1]) / np.abs(op[:, 1])
        scale = np.abs(op[:, 1]) * np.abs(xend) + o.leaf_start[1:].astype(np.float64) + 1.0
        rela = np.abs(gp[:, 0] - op[:, 0]) / scale
    relb[(gp[:, 1] == op[:, 1]) | ~nonempty] = 0.0
    rela[(gp[:, 0] == op[:, 0]) | ~nonempty] = 0.0
    assert np.nanmax(relb) <= 1e-8 and np.nanmax(rela) <= 1e-8, (np.nanmax(relb), np.nanmax(rela))
    assert np.mean(relb <= 1e-9) >= 0.95
    # rows == the reference's L1_PARAMETERS image of THESE coefficients and errors
    rows = g.rows.view(np.uint64).reshape(L, 3)
    assert np.array_equal(rows[:, :2], gp.view(np.uint64)) and np.array_equal(rows[:, 2], ge)
    return float(np.nanmax(relb))


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("gen,n,L", [
    ("uniform_u64", 2_000_000, 8192), ("uniform_u64", 600_000, 4096), ("books_u64", 2_000_000, 8192),
    ("uniform_u32", 2_000_000, 8192), ("uniform_f64", 600_000, 2048),
    ("dups_u64", 600_000, 2048), ("dups_u32", 600_000, 2048), ("clustered_u64", 600_000, 2048),
])
@pytest.mark.parametrize("root", ["linear", "linear_spline", "radix", "cubic"])
def test_onepass_parity(T, oracle, gen, n, L, root, mode):
    if root == "radix" and gen == "uniform_f64":
        pytest.skip("radix roots take integer keys")
    keys = dg.GENERATORS[gen](n)
    try:
        g, o = _run(T, oracle, keys, root, L, mode)
    except oracle.OracleError as oe:
        pytest.skip(f"the reference panics on this combination ({oe})")
    worst = _check(g, o, keys, mode)
    nonempty = int(np.count_nonzero(o.leaf_start[1:] > o.leaf_start[:-1]))
    print(f"\n{gen} {root} L={L} mode={mode}: exact re-fits {g.exact_leaves} of {nonempty} non-empty leaves, guard {g.guard_leaves}, "
          f"worst slope difference {worst:.2e}")
    if gen.startswith("dups") or gen.startswith("clustered"):
        assert g.exact_leaves >= 0.9 * nonempty          # duplicates / collapsed f64 keys: the sums do not apply, (nearly) every leaf is exact
    elif gen.startswith("uniform"):
        assert g.exact_leaves <= 0.05 * nonempty + 8


def test_onepass_falls_back_for_short_leaves(T, oracle):
    """Fewer than ~32 keys per leaf on average: the exact kernels are used whatever the mode says."""
    keys = dg.uniform_u64(300_000)
    g, o = _run(T, oracle, keys, "linear", 65536, 1)
    _check(g, o, keys, 1, expect_used=False)
    assert np.array_equal(g.leaf_params, o.leaf_params)


@pytest.mark.parametrize("n,L", [(1_000_003, 4099), (524_288, 2048), (70_001, 1024), (4_096 * 3 + 1, 97)])
def test_onepass_odd_sizes(T, oracle, n, L):
    """Chunk, batch and ring edges: sizes around the kernel's geometry (512-key batches, 2048-key ring)."""
    keys = dg.uniform_u64(n)
    g, o = _run(T, oracle, keys, "linear", L, 1)
    _check(g, o, keys, 1)


def test_onepass_long_leaves(T, oracle):
    """Leaves far longer than the LDS ring (and than one wave's share of the keys): irregular, exact kernels."""
    keys = dg.uniform_u64(2_000_000)
    g, o = _run(T, oracle, keys, "linear", 64, 1)
    _check(g, o, keys, 1)
    assert g.exact_leaves == 64 and np.array_equal(g.leaf_params, o.leaf_params)


def test_onepass_results_do_not_depend_on_the_wave_count(T, oracle, monkeypatch):
    """The cut of the key array into per-wave chunks must not show in any integer output."""
    keys = dg.books_u64(1_500_000)
    outs = []
    for waves in ("64", "4096", "100000"):
        monkeypatch.setenv("RMI_HIP_SIGMA_WAVES", waves)
        g, o = _run(T, oracle, keys, "linear", 4096, 1)
        _check(g, o, keys, 1)
        outs.append(g.last_layer_max_l1s.copy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])


def test_onepass_shards_match_the_oracle(T, oracle):
    """2 and 4 leaf-aligned shards in guarded one-pass mode, run one after the other on one GPU:
    every shard's error integers and counts are the oracle's."""
    from rmi_amd import sharded
    n, L = 1_200_000, 4096
    keys = dg.uniform_u64(n)
    o = oracle.train_two_layer("linear", "linear", keys, L, threads=2)
    root = T.Model(0, tuple(o.root.p), tuple(o.root.ip))
    for world in (2, 4):
        plans = sharded.Planner(lambda i: keys[i], n, keys.dtype, root, L).plan(world)
        for pl in plans:
            tr = T.Trainer(np.ascontiguousarray(keys[pl.read_lo:pl.read_hi]))
            tr.set_fit_mode(1)
            res = sharded.run_shard(tr, pl, root, "linear").materialize()
            assert res.fit_mode_used == 1
            assert np.array_equal(res.last_layer_max_l1s, o.leaf_err[pl.leaf_lo:pl.leaf_hi])
            assert np.array_equal(res.leaf_counts, o.leaf_count[pl.leaf_lo:pl.leaf_hi])
            assert np.array_equal(res.leaf_starts[:-1], o.leaf_start[pl.leaf_lo:pl.leaf_hi])
            tr.close()
