"""Upload + train overlapped (rmi_hip_train_streamed: pinned chunked staging, leaf-aligned shards trained behind the
upload; replaces src/load.rs:132-157 + train/mod.rs:100-126 for keys in host memory): the results are those of the
resident path -- and of the oracle."""
import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu


def _same(a, b, exact_params=True):
    assert np.array_equal(a.leaf_starts, b.leaf_starts)
    assert np.array_equal(a.leaf_counts, b.leaf_counts)
    assert np.array_equal(a.last_layer_max_l1s, b.last_layer_max_l1s)
    if exact_params:
        assert np.array_equal(a.leaf_params, b.leaf_params)
        assert np.array_equal(a.rows, b.rows)
    assert a.model_max_error == b.model_max_error and a.model_max_error_idx == b.model_max_error_idx
    assert a.model_avg_error == b.model_avg_error
    assert abs(a.model_avg_l2_error - b.model_avg_l2_error) <= 1e-12 * max(1.0, abs(b.model_avg_l2_error))
    assert abs(a.model_avg_log2_error - b.model_avg_log2_error) <= 1e-12 * max(1.0, abs(b.model_avg_log2_error))


@pytest.mark.parametrize("gen,n,L,spec,chunks", [
    ("uniform_u64", 3_000_000, 4096, "linear,linear", 8),
    ("books_u64", 2_000_000, 2048, "linear,linear", 16),
    ("dups_u64", 1_500_000, 2048, "linear_spline,linear", 4),
    ("uniform_u32", 3_000_000, 4096, "radix,linear_spline", 8),
    ("uniform_f64", 1_000_000, 1024, "linear,cubic", 2),
    ("uniform_u64", 1_000_003, 4096, "linear,linear", 64),
    ("clustered_u64", 1_200_000, 1024, "linear,linear", 1),
    ("uniform_u64", 1_000_000, 1024, "linear,robust_linear", 4),
    ("books_u64", 1_500_000, 2048, "cubic,linear", 8),
])
def test_streamed_equals_resident_and_oracle(oracle, gen, n, L, spec, chunks):
    from rmi_amd import train
    keys = dg.GENERATORS[gen](n)
    root_name, leaf_name = spec.split(",")
    o = oracle.train_two_layer(root_name, leaf_name, keys, L, threads=2)
    tr = train.Trainer()
    root = tr.fit_root_host(keys, root_name, L)
    assert root.p == o.root.p and root.ip == o.root.ip
    s = tr.train_streamed(keys, root, leaf_name, L, chunks=chunks).materialize()
    r = tr.train_leaves(root, leaf_name, L).materialize()          # the keys are resident now
    _same(s, r)
    assert np.array_equal(s.leaf_starts, o.leaf_start) and np.array_equal(s.leaf_params, o.leaf_params)
    assert np.array_equal(s.last_layer_max_l1s, o.leaf_err) and np.array_equal(s.leaf_counts, o.leaf_count)
    assert s.model_max_error == o.model_max_error and s.model_avg_error == o.model_avg_error
    tr.close()


def test_streamed_one_pass_modes(oracle):
    from rmi_amd import train
    keys = dg.uniform_u64(4_000_000)
    L = 16384
    o = oracle.train_two_layer("linear", "linear", keys, L, threads=2)
    for mode in (1, 2):
        tr = train.Trainer()
        tr.set_fit_mode(mode)
        root = tr.fit_root_host(keys, "linear", L)
        s = tr.train_streamed(keys, root, "linear", L, chunks=8).materialize()
        assert s.fit_mode_used == mode
        assert np.array_equal(s.leaf_starts, o.leaf_start) and np.array_equal(s.leaf_counts, o.leaf_count)
        d = np.abs(s.last_layer_max_l1s.astype(np.int64) - o.leaf_err.astype(np.int64))
        if mode == 1:
            assert d.max() == 0 and s.model_max_error == o.model_max_error and s.model_avg_error == o.model_avg_error
        else:
            assert d.max() <= 1 and np.count_nonzero(d) <= s.guard_leaves + s.merged_leaves
        rel = np.abs(s.leaf_params[:, 1] - o.leaf_params[:, 1]) / np.abs(o.leaf_params[:, 1])
        assert np.nanmax(rel) <= 1e-8
        tr.close()


def test_streamed_bad_arguments():
    from rmi_amd import train
    keys = dg.uniform_u64(100_000)
    tr = train.Trainer()
    root = tr.fit_root_host(keys, "linear", 1000)
    with pytest.raises(train.RMIError):
        tr.train_streamed(keys, root, "linear", 1000, chunks=7)      # 1000 leaves are not a multiple of 7
    with pytest.raises(train.RMIError):
        tr.train_streamed(keys, root, "linear", 1000, chunks=65)
    g = tr.train_streamed(keys, root, "linear", 1000, chunks=8)
    assert g.branching_factor == 1000
    tr.close()


@pytest.mark.parametrize("spec,L", [("radix18,linear", 4096), ("bradix,linear", 2048), ("linear,linear", 1000), ("radix,linear_spline", 4096)])
def test_train_from_host_matches_the_oracle(oracle, spec, L):
    """train.train(keys, ...) for keys in host memory: host root fit + streamed training where the root allows it
    (a leaf count that no power of two divides trains in one shard), the plain path for radix tables and bradix."""
    from rmi_amd import train
    keys = dg.uniform_u64(1_500_000)
    root_name, leaf_name = spec.split(",")
    o = oracle.train_two_layer(root_name, leaf_name, keys, L, threads=2)
    g = train.train(keys, spec, L)
    assert np.array_equal(g.leaf_starts, o.leaf_start) and np.array_equal(g.leaf_params, o.leaf_params)
    assert np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
    assert g.model_max_error == o.model_max_error and g.model_avg_error == o.model_avg_error
