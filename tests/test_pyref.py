"""The C oracle against a second, independent restatement of the reference (tests/pyref.py: pure
Python, written from the reference source in the reference's own terms -- vectors of pairs,
iterators -- where the oracle uses index arithmetic).  Small random key sets with duplicate runs,
skew and empty leaves; every output must agree exactly, and where one panics the other must too."""
import numpy as np
import pytest

from tests import pyref

CASES = [("linear", "linear"), ("linear", "linear_spline"), ("linear", "cubic"), ("cubic", "linear"), ("radix", "linear"),
         ("linear_spline", "linear"), ("robust_linear", "linear"), ("radix", "linear_spline"), ("cubic", "cubic"),
         ("normal", "linear"), ("loglinear", "linear")]


def _keys(rng, trial):
    n = int(rng.integers(12, 400))
    kind = trial % 4
    if kind == 0:                                             # spread over the whole u64 range
        keys = rng.integers(1, (1 << 63) - 1, size=n, dtype=np.uint64) * 2
    elif kind == 1:                                           # small values, many duplicates
        keys = rng.integers(1, max(4, n // 2), size=n, dtype=np.uint64)
    elif kind == 2:                                           # clustered: empty leaves, long leaves
        keys = np.concatenate([rng.integers(1, 1000, size=n - n // 4, dtype=np.uint64),
                               rng.integers(1 << 40, (1 << 40) + 1000, size=n // 4, dtype=np.uint64)])
    else:                                                     # dense with runs
        keys = np.arange(1000, 1000 + n, dtype=np.uint64)
        keys[n // 3:n // 3 + n // 6] = keys[n // 3]
    return np.sort(keys)


@pytest.mark.parametrize("root,leaf", CASES)
def test_oracle_agrees_with_python_restatement(oracle, root, leaf):
    rng = np.random.default_rng(abs(hash((root, leaf))) % (1 << 32) if False else len(root) * 131 + len(leaf))
    agreed = panics = 0
    for trial in range(24):
        keys = _keys(rng, trial)
        L = int(rng.integers(2, 48))
        try:
            ref = pyref.train_two_layer([int(k) for k in keys], root, leaf, L)
        except pyref.ReferencePanic:
            with pytest.raises(oracle.OracleError):
                oracle.train_two_layer(root, leaf, keys, L)
            panics += 1
            continue
        o = oracle.train_two_layer(root, leaf, keys, L)
        if root == "radix":
            assert tuple(o.root.ip[:2]) == tuple(ref["root"].ip), (trial, keys, L)
        else:
            assert list(o.root.p[:len(ref["root"].params())]) == ref["root"].params(), (trial, L)
        ppl = o.params_per_leaf
        got = [[float(v) for v in o.leaf_params[j, :ppl]] for j in range(L)]
        want = [m.params() for m in ref["leaves"]]
        assert got == want, (trial, L, [j for j in range(L) if got[j] != want[j]][:5])
        assert [int(v) for v in o.leaf_err] == ref["errs"], (trial, L)
        assert [int(v) for v in o.leaf_count] == ref["counts"], (trial, L)
        assert o.model_max_error == ref["max_error"] and o.model_max_error_idx == ref["max_error_idx"]
        assert o.model_avg_error == ref["avg_error"]
        agreed += 1
    assert agreed >= 8, (agreed, panics)


@pytest.mark.parametrize("root,leaf,keys,L", [
    ("robust_linear", "linear", [5, 9, 12], 2),                    # linear.rs:248: assert!(bnd*2+1 < data.len())
    ("radix", "linear", [7] * 40, 4),                              # utils.rs:18: num_bits of 0
    ("linear", "linear", [7] * 40, 4),                             # every key in one leaf: two_layer.rs:27 / :144
    ("linear", "linear", [3] * 20 + [900] * 20, 2),                # split at index 20 is fine ...
    ("cubic", "linear", [1, 2, 3, (1 << 64) - 2], 8),
    ("linear", "cubic", list(range(10, 400, 3)), 2),
])
def test_panics_and_edge_cases_agree(oracle, root, leaf, keys, L):
    arr = np.array(keys, dtype=np.uint64)
    try:
        ref = pyref.train_two_layer(keys, root, leaf, L)
    except pyref.ReferencePanic:
        with pytest.raises(oracle.OracleError):
            oracle.train_two_layer(root, leaf, arr, L)
        return
    o = oracle.train_two_layer(root, leaf, arr, L)
    ppl = o.params_per_leaf
    assert [[float(v) for v in o.leaf_params[j, :ppl]] for j in range(L)] == [m.params() for m in ref["leaves"]]
    assert [int(v) for v in o.leaf_err] == ref["errs"] and [int(v) for v in o.leaf_count] == ref["counts"]


@pytest.mark.parametrize("kind", ["u32", "f64"])
@pytest.mark.parametrize("root,leaf", [("linear", "linear"), ("radix", "linear_spline"), ("cubic", "linear"), ("linear", "cubic")])
def test_other_key_types_agree(oracle, kind, root, leaf):
    """u32 keys (T::MAX = 2^32-1 in the widening of the last leaf, widened to u64 for radix) and f64 keys
    (as_float is the identity, +-EPSILON, `as u64` saturating for radix): models/mod.rs:89-111."""
    rng = np.random.default_rng(7 + len(root) + 3 * len(leaf) + (0 if kind == "u32" else 100))
    for trial in range(16):
        n = int(rng.integers(12, 300))
        if kind == "u32":
            hi = [1 << 32, 5000, 1 << 20, 300][trial % 4]
            keys = np.sort(rng.integers(1, hi - 1, size=n, dtype=np.uint64)).astype(np.uint32)
        else:
            keys = np.sort(rng.random(n) * [1e6, 3.0, 1e15, 50.0][trial % 4] + 1.0)
            if trial % 3 == 0:
                keys[n // 2:n // 2 + 4] = keys[n // 2]
                keys = np.sort(keys)
        L = int(rng.integers(2, 40))
        try:
            ref = pyref.train_two_layer(list(keys.tolist()), root, leaf, L, kind=kind)
        except pyref.ReferencePanic:
            with pytest.raises(oracle.OracleError):
                oracle.train_two_layer(root, leaf, keys, L)
            continue
        o = oracle.train_two_layer(root, leaf, keys, L)
        ppl = o.params_per_leaf
        got = [[float(v) for v in o.leaf_params[j, :ppl]] for j in range(L)]
        assert got == [m.params() for m in ref["leaves"]], (kind, trial, L)
        assert [int(v) for v in o.leaf_err] == ref["errs"], (kind, trial, L)
        assert [int(v) for v in o.leaf_count] == ref["counts"]
        assert o.model_max_error == ref["max_error"] and o.model_max_error_idx == ref["max_error_idx"]
