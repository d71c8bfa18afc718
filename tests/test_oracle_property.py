"""The reference's own acceptance test (tests/simple_model_wiki/main.cpp:26-41), applied to the
oracle's output on synthetic data: for every key |lookup(key) - lower_bound(key)| <= err.
Model pairs are those of the reference's tests/*/Makefile plus the BASELINE configs."""
import numpy as np
import pytest

from rmi_amd import datagen as dg

SPECS = [
    ("linear", "linear", 1024),          # BASELINE config 1
    ("cubic", "linear", 4096),           # tests/simple_model_wiki
    ("robust_linear", "linear", 4096),   # tests/simple_model_osm
    ("radix", "linear", 1024),           # tests/radix_model_wiki
    ("radix", "linear_spline", 4096),    # BASELINE config 5 shape
    ("linear_spline", "linear", 4096),   # tests/cache_fix_wiki model pair
    ("linear", "cubic", 512),
]


@pytest.mark.parametrize("gen", ["uniform_u64", "books_u64", "dups_u64", "clustered_u64", "uniform_u32", "dups_u32", "uniform_f64"])
@pytest.mark.parametrize("root,leaf,L", SPECS)
def test_lookup_property(oracle, gen, root, leaf, L):
    keys = dg.GENERATORS[gen](200_000)
    r = oracle.train_two_layer(root, leaf, keys, L)
    bad, first = oracle.check_lookup_property(r, keys)
    assert bad == 0, f"{bad} keys violate the bound, first at {first}"
    assert int(r.leaf_count.sum()) == len(keys) + 1      # Q7: tail duplicate counted once more
    assert r.leaf_start[0] == 0 and r.leaf_start[-1] == len(keys)


def test_two_threads_equals_one(oracle):
    keys = dg.books_u64(300_000)
    a = oracle.train_two_layer("linear", "linear", keys, 2048, threads=1)
    b = oracle.train_two_layer("linear", "linear", keys, 2048, threads=2)
    assert np.array_equal(a.leaf_params, b.leaf_params)
    assert np.array_equal(a.leaf_err, b.leaf_err)


def test_error_codes(oracle):
    keys = dg.uniform_u64(1000)
    with pytest.raises(oracle.OracleError) as e:
        oracle.train_two_layer("linear", "radix", keys, 16)     # radix must be top: radix.rs:75-80
    assert e.value.code == -2
    with pytest.raises(oracle.OracleError) as e:
        oracle.train_two_layer("nope", "linear", keys, 16)       # train/mod.rs:53
    assert e.value.code == -1
    # decreasing root -> targets decrease -> two_layer.rs:50 / :144
    root = oracle.Model(oracle.MODEL_LINEAR, (15.0, -1e-18, 0.0, 0.0), (0, 0))
    with pytest.raises(oracle.OracleError):
        oracle.train_two_layer("linear", "linear", keys, 16, root=root)
