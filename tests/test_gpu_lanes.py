"""The leaf-lane pipeline (rmi_amd/csrc/rmi_lanes.hip.h: k_leaf_search, k_leaf_lanes, the giant-leaf epilogue) against
the oracle through the C ABI: every variant of the path (boundaries by search / by the
bucketing scan, leaves handed to the list kernels and to the host), on the seeded generators and on adversarial key
sets with exact linear structure.  Bar: bucket table, error integers, counts AND coefficients bit-identical."""
import numpy as np
import pytest

from rmi_amd import datagen as dg

pytestmark = pytest.mark.gpu

VARIANTS = {
    "fused_search": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_LANES_SEARCH": "1"},
    "fused_scan": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_LANES_SEARCH": "0"},
    # leaves of more than 512 points go to the list kernels, of more than 2000 to the host
    "lists_and_host": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_HOST_MIN": "2000", "RMI_HIP_LONG_MIN": "512"},
    # the list kernels and k_finalize_listed inside the stream of every training (default: k_lane_reduce publishes, the list
    # kernels run behind the synchronisation, and only when a leaf was handed over)
    "in_stream_tail": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_OPT_TAIL": "0"},
    "in_stream_lists": {"RMI_HIP_PIPELINE": "3", "RMI_HIP_OPT_TAIL": "0", "RMI_HIP_HOST_MIN": "2000", "RMI_HIP_LONG_MIN": "512"},
}


def _train(monkeypatch, env, keys, root, leaf, L, mode=0):
    from rmi_amd import train
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tr = train.Trainer(keys)
    tr.set_fit_mode(mode)
    g_root = tr.fit_root(root, L)
    return tr, g_root, train


def _check(monkeypatch, oracle, env, keys, root, L, leaf="linear", mode=0, coef_exact=True):
    tr, g_root, train = _train(monkeypatch, env, keys, root, leaf, L, mode)
    try:
        o = oracle.train_two_layer(root, leaf, keys, L)
    except oracle.OracleError as oe:
        with pytest.raises(train.RMIError) as ge:
            tr.train_leaves(g_root, leaf, L)
        assert ge.value.code == oe.code
        tr.close()
        return None
    g = tr.train_leaves(g_root, leaf, L)
    assert np.array_equal(g.leaf_starts, o.leaf_start), "bucket assignment differs"
    assert np.array_equal(g.last_layer_max_l1s, o.leaf_err), f"{np.count_nonzero(g.last_layer_max_l1s != o.leaf_err)} error integers differ"
    assert np.array_equal(g.leaf_counts, o.leaf_count)
    if coef_exact:
        assert np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64)), \
            f"{np.count_nonzero((g.leaf_params != o.leaf_params).any(axis=1))} coefficient rows differ"
    assert g.model_max_error == o.model_max_error and g.model_max_error_idx == o.model_max_error_idx
    assert g.model_avg_error == o.model_avg_error
    assert abs(g.model_avg_l2_error - o.model_avg_l2_error) <= 1e-9 * max(1.0, abs(o.model_avg_l2_error))
    assert abs(g.model_avg_log2_error - o.model_avg_log2_error) <= 1e-9 * max(1.0, abs(o.model_avg_log2_error))
    rows = g.rows.view(np.uint64).reshape(L, 3)
    assert np.array_equal(rows[:, :2], g.leaf_params.view(np.uint64)) and np.array_equal(rows[:, 2], g.last_layer_max_l1s)
    tr.close()
    return g


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("gen,n,L,root", [
    ("uniform_u64", 300_000, 1024, "linear"), ("uniform_u64", 300_000, 16384, "linear"), ("books_u64", 300_000, 4096, "linear"),
    ("dups_u64", 300_000, 4096, "linear"), ("clustered_u64", 300_000, 1024, "linear"), ("uniform_u32", 300_000, 4096, "linear"),
    ("dups_u32", 300_000, 1024, "linear"), ("uniform_u64", 300_000, 4096, "cubic"), ("uniform_u64", 300_000, 4096, "radix"),
    ("dups_u64", 200_000, 40_000, "linear"), ("uniform_u64", 1_000_000, 64, "linear"), ("dups_u64", 300_000, 64, "linear"),
    ("books_u64", 300_000, 100, "linear"), ("uniform_u64", 2_000_000, 8, "linear"), ("dups_u32", 300_000, 32, "linear"),
    ("clustered_u64", 300_000, 16, "linear"), ("uniform_f64", 200_000, 2048, "linear"), ("uniform_u64", 300_001, 1000, "linear_spline"),
    ("uniform_u64", 1500, 7, "linear"), ("dups_u64", 100_000, 99_999, "linear"),
])
def test_lanes_variants(monkeypatch, oracle, variant, gen, n, L, root):
    _check(monkeypatch, oracle, VARIANTS[variant], dg.GENERATORS[gen](n), root, L)


@pytest.mark.parametrize("gen,n,L", [("uniform_u64", 300_000, 1024), ("dups_u64", 300_000, 1024), ("books_u64", 400_000, 512)])
def test_more_giant_leaves_than_the_early_list_holds(monkeypatch, oracle, gen, n, L):
    """Every leaf a 'giant' (host threshold 150 points): more of them than the early list carries to the host beside k_list
    (256) -- the host then fits them all behind the list kernels, from the complete list.  Same bits."""
    env = {"RMI_HIP_PIPELINE": "3", "RMI_HIP_HOST_MIN": "150", "RMI_HIP_LONG_MIN": "64"}
    g = _check(monkeypatch, oracle, env, dg.GENERATORS[gen](n), "linear", L)
    assert g is not None and g.long_leaves > 256


@pytest.mark.parametrize("gen,n,L", [("books_u64", 600_000, 64), ("clustered_u64", 600_000, 32), ("uniform_u64", 1_000_000, 16)])
def test_guarded_mode_hands_its_giant_leaves_to_the_host(monkeypatch, oracle, gen, n, L):
    """The guarded one-pass mode re-fits its long leaves exactly; those beyond the host threshold (2 000 points here) are
    chains for host cores, like on the exact path, with every leaf finalized once more behind them: error integers, counts
    and aggregates the oracle's."""
    env = {"RMI_HIP_HOST_MIN": "2000"}
    g = _check(monkeypatch, oracle, env, dg.GENERATORS[gen](n), "linear", L, mode=1, coef_exact=False)
    assert g is not None and g.fit_mode_used == 1


@pytest.mark.parametrize("gen", sorted(dg.ADVERSARIAL))
@pytest.mark.parametrize("n,L", [(200_000, 1024), (199_999, 1000), (65_536, 4096)])
def test_adversarial_exact(monkeypatch, oracle, gen, n, L):
    """Exact linear structure (every prediction on an integer +- rounding), keys around 2^53 / 2^63, an outlier, L not
    dividing n: the exact mode has nothing to guard -- same operations in the same order -- and must still agree."""
    keys = dg.ADVERSARIAL[gen](n)
    _check(monkeypatch, oracle, VARIANTS["fused_search"], keys, "linear", L)


@pytest.mark.parametrize("gen", sorted(dg.ADVERSARIAL))
def test_adversarial_guard(monkeypatch, oracle, gen, capsys):
    """The same sets in the guarded one-pass mode (rmi_hip_set_fit_mode 1): its first-order guard must keep every error
    integer equal to the reference's -- a silent +-1 would be an unsound index (tests/simple_model_wiki/main.cpp:26-41).
    Prints the share of leaves the guard sent to the exact kernels."""
    n, L = 200_000, 1024
    keys = dg.ADVERSARIAL[gen](n)
    g = _check(monkeypatch, oracle, {"RMI_HIP_PIPELINE": "3"}, keys, "linear", L, mode=1, coef_exact=False)
    if g is not None:
        with capsys.disabled():
            print(f"\n[guard] {gen}: mode used {g.fit_mode_used}, exact re-fits {g.exact_leaves} of {L} leaves, guard-flagged {g.guard_leaves}")


def test_books_torch_generator_matches_numpy():
    """(In a process of its own: torch brings its own HIP runtime, which must be the first one the process initialises.)"""
    import subprocess, sys, os
    code = ("import torch, numpy as np, sys; sys.path.insert(0, %r); from rmi_amd import datagen as dg\n"
            "for n in (1000, 300001):\n"
            "    a = dg.books_u64(n); b = dg.books_u64_torch(n, device='cuda').cpu().numpy().view(np.uint64)\n"
            "    assert np.array_equal(a, b), n\n"
            "print('same')") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "same" in r.stdout, r.stderr[-2000:]


def test_one_pass_request_beyond_32_bit_indices_reports_exact(monkeypatch):
    """n >= 2^32 - 2^16: the one-pass kernels (32-bit indices) are not used; fit_mode_used says so.
    Checked on the decision itself with a generated key set just above the limit would need 34 GB; the full-size test
    (test_gpu_fullsize.py::test_more_than_2_pow_32_keys) asserts it on 4.3 G keys."""
    from rmi_amd import train
    tr = train.Trainer()
    tr.generate_keys("uniform", np.uint64, 2_000_000)
    tr.set_fit_mode(1)
    root = tr.fit_root("linear", 4096, mode="fast")
    g = tr.train_leaves(root, "linear", 4096)
    assert g.fit_mode_used == 1                  # below the limit the request is honoured
    tr.close()


@pytest.mark.parametrize("params,what", [((0.0, 0.0, -1e-15, 3000.0), "decreasing"), ((0.0, 0.0, 1e-13, 0.0), "out of bounds"),
                                         ((1e-50, -3e-32, 2e-14, 5.0), "wiggle")])
def test_cubic_root_verified_by_the_error_pass(monkeypatch, oracle, params, what):
    """A cubic root is not monotone by arithmetic: the search assumes it and k_leaf_lanes verifies every key's target during
    its error pass (two_layer.rs:45-50).  Caller-provided roots that are decreasing / out of bounds / wiggling must give the
    reference's panic as the same error code as the scan-based pipelines and the oracle -- or the same result."""
    from rmi_amd import train
    keys = dg.uniform_u64(300_000)
    L = 4096
    root = train.Model(2, params, (0, 0, 0, 0))
    outs = []
    for pl in ("3", "2"):
        monkeypatch.setenv("RMI_HIP_PIPELINE", pl)
        tr = train.Trainer(keys)
        try:
            g = tr.train_leaves(root, "linear", L).materialize()
            outs.append(("ok", g.last_layer_max_l1s.copy(), g.leaf_params.copy()))
        except train.RMIError as e:
            outs.append(("err", e.code, None))
        tr.close()
    assert outs[0][0] == outs[1][0], (what, outs[0][:2], outs[1][:2])
    if outs[0][0] == "err":
        assert outs[0][1] == outs[1][1], what
    else:
        assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2].view(np.uint64), outs[1][2].view(np.uint64))
