"""The register-resident leaf kernel (rmi_amd/csrc/rmi_regs.hip.h: k_leaf_regs, k_regs_finalize, k_leaf_lanes_listed) against the
oracle through the C ABI, and against the leaf-lane pipeline it replaces: every variant of the path (non-temporal / plain
loads (gone: measured equal), the LONG variant forced on every shape, a handful of persistent waves, ), on the seeded generators and on
key sets that exercise its special cases -- containers of more than 240 points (the lanes that go on from the key array),
of more than 1 008 (the group is listed), duplicate keys found while walking (listed), the leaf behind the split, empty
leaves, f64 keys (IEEE division), shards.  Bar: bucket table, error integers, counts AND coefficients bit-identical."""
import numpy as np
import pytest

from rmi_amd import datagen as dg

from tests.test_gpu_lanes import _check

pytestmark = pytest.mark.gpu

VARIANTS = {
    "default": {"RMI_HIP_REGS": "1"},
    "long": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_MAX_AVG": "0", "RMI_HIP_REGS_LONG_MAX_AVG": "100000"},   # k_leaf_regs<K, LONG> for every shape
    "seven_waves": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_GRID": "7"},          # every wave takes many groups, the last ones uneven
    "sixteen_waves": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_GRID": "16"},
    "any_average": {"RMI_HIP_REGS": "1", "RMI_HIP_REGS_MAX_AVG": "100000"},  # also where most groups hold long containers
}
CASES = [
    ("uniform_u64", 300_000, 4096, "linear"), ("uniform_u64", 300_000, 16384, "linear"), ("uniform_u64", 1_000_000, 8192, "linear"),
    # ~200 keys per leaf: containers on both sides of 240 points in most groups
    ("uniform_u64", 2_000_000, 10_000, "linear"), ("uniform_u64", 1_000_000, 5300, "linear"),
    # skewed: containers of more than 1 008 points beside short ones, empty leaves
    ("books_u64", 300_000, 4096, "linear"), ("books_u64", 1_000_000, 20_000, "linear"), ("clustered_u64", 300_000, 2048, "linear"),
    # duplicates: found while walking, the group goes on the list
    ("dups_u64", 300_000, 4096, "linear"), ("dups_u64", 200_000, 40_000, "linear"),
    ("uniform_f64", 300_000, 4096, "linear"), ("uniform_f64", 1_000_000, 6000, "linear"),
    ("uniform_u64", 300_000, 4096, "radix"), ("uniform_u64", 70_000, 1000, "linear"), ("uniform_u64", 5_000, 64, "linear"),
    ("uniform_u64", 300_000, 100_000, "linear"), ("uniform_u64", 300_001, 1000, "linear"),
    # long leaves (k_leaf_regs<K, LONG> by default between 208 and 640 keys a leaf): C4's shard shape (381), walks of two dozen blocks
    # whose error steps behind the stash come through the ring a second time; 600; skewed; f64 keys; C1's shape (977: most groups listed)
    ("uniform_u64", 1_500_000, 3937, "linear"), ("uniform_u64", 2_000_000, 3333, "linear"), ("books_u64", 2_000_000, 5000, "linear"),
    ("uniform_f64", 1_200_000, 3000, "linear"), ("uniform_u64", 1_000_000, 1024, "linear"), ("uniform_u64", 1_500_000, 4096, "radix"),
    # 4-byte keys (src/load.rs:47-69): one stash register a key, half-line panels; the C5-like shape (95 keys a leaf), ~200 keys a leaf,
    # duplicates (the groups that meet one are listed), long leaves, a radix root
    ("uniform_u32", 400_000, 4096, "linear"), ("uniform_u32", 1_000_000, 5300, "linear"), ("uniform_u32", 300_000, 4096, "radix"),
    ("dups_u32", 300_000, 4096, "linear"), ("uniform_u32", 1_500_000, 3937, "linear"), ("uniform_u32", 70_000, 1000, "linear"),
]


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("gen,n,L,root", CASES)
def test_regs_variants(monkeypatch, oracle, variant, gen, n, L, root):
    g = _check(monkeypatch, oracle, VARIANTS[variant], dg.GENERATORS[gen](n), root, L)
    # (skewed long leaves -- books, 400 keys a leaf -- : most groups hold a container of more than 1 008 points and are listed: reported as 3)
    if g is not None and n >= 1024 and n <= (100000 if variant in ("any_average", "long") else 640) * L and not (gen == "books_u64" and n > 208 * L):
        # (every group listed -- by the switch, because the boundary search met duplicates, or because every group met keys whose f64 images
        #  collapse: k_leaf_lanes_listed did the work, reported as 3)
        assert g.pipeline in ((3, 4) if (gen.startswith("dups") or gen == "clustered_u64") else (4,))


@pytest.mark.parametrize("u32", ["0", "1", "2"])
@pytest.mark.parametrize("gen,n,L,root", [c for c in CASES if c[0].endswith("u32")] + [("uniform_u32", 2_000_000, 8192, "cubic"), ("dups_u32", 1_000_000, 4000, "radix")])
def test_regs_four_byte_keys_every_route(monkeypatch, oracle, u32, gen, n, L, root):
    """4-byte keys with linear leaves on their three routes (RMI_HIP_REGS_U32): 2 = k_leaf_regs<u32, 2>, two waves per SIMD with the raw keys stashed (the
    default), 1 = the one-wave kernel in half-line panels (short and LONG variants by the average), 0 = k_leaf_lanes.  Same bits as the oracle every way."""
    g = _check(monkeypatch, oracle, {"RMI_HIP_REGS": "1" if u32 != "0" else "", "RMI_HIP_REGS_U32": u32}, dg.GENERATORS[gen](n), root, L)
    if g is not None and u32 == "0":
        assert g.pipeline == 3
    if g is not None and u32 != "0" and gen == "uniform_u32" and root != "cubic":
        assert g.pipeline == 4


@pytest.mark.parametrize("name", sorted(dg.ADVERSARIAL))
def test_regs_adversarial(monkeypatch, oracle, name):
    """Key sets with exact linear structure (progressions, keys around 2^53 and 2^63, an outlier): the closed form of the y
    half and the 32-bit duplicate test on keys whose low words repeat."""
    keys = dg.ADVERSARIAL[name](300_000)
    for L in (2048, 4096):
        _check(monkeypatch, oracle, {"RMI_HIP_REGS": "1"}, keys, "linear", L)


def test_regs_equals_lanes_rows(monkeypatch):
    """The two pipelines leave the same bytes in every output array (rows, parameters, error integers, counts)."""
    from rmi_amd import train
    keys = dg.uniform_u64(1_500_000)
    out = {}
    for name, v in (("lanes", "0"), ("regs", "1")):
        monkeypatch.setenv("RMI_HIP_REGS", v)
        tr = train.Trainer(keys)
        g = tr.train("linear,linear", 8192)
        assert g.pipeline == (4 if v == "1" else 3)
        out[name] = (g.rows.copy(), g.leaf_params.copy(), g.last_layer_max_l1s.copy(), g.leaf_counts.copy(), g.leaf_starts.copy(),
                     g.model_max_error, g.model_max_error_idx, g.model_avg_error, g.model_avg_l2_error, g.model_avg_log2_error)
        tr.close()
    for a, b in zip(out["lanes"], out["regs"]):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


def test_regs_repeated_trainings_one_context(monkeypatch, oracle):
    """The persistent kernel's counters and lists are reset by every training: different L on one context, back and forth."""
    from rmi_amd import train
    monkeypatch.setenv("RMI_HIP_REGS", "1")
    keys = dg.books_u64(400_000)
    tr = train.Trainer(keys)
    for L in (4096, 2048, 16384, 4096, 100_000, 2048):
        g = tr.train("linear,linear", L)
        o = oracle.train_two_layer("linear", "linear", keys, L)
        assert np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64))
        assert np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
    tr.close()


def test_regs_backs_off_on_duplicate_heavy_keys(monkeypatch, oracle):
    """Every group of a duplicate-heavy key set meets a duplicate and goes on the list.  The boundary search sees the duplicates in its probes
    (DevState::regs_dups) and k_leaf_regs then lists every group at once, without walking any: pipeline 3 already on the FIRST training; the
    context remembers (key set, leaves) and does not launch k_leaf_regs from the second training on; same bits either way; new keys start
    afresh; RMI_HIP_REGS_BACKOFF=0 keeps pipeline 4."""
    from rmi_amd import train
    monkeypatch.setenv("RMI_HIP_REGS", "1")
    keys = dg.dups_u64(400_000)
    o = oracle.train_two_layer("linear", "linear", keys, 4096)
    tr = train.Trainer(keys)
    seen = []
    for _ in range(3):
        g = tr.train("linear,linear", 4096)
        seen.append(g.pipeline)
        assert np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64))
        assert np.array_equal(g.last_layer_max_l1s, o.leaf_err) and np.array_equal(g.leaf_counts, o.leaf_count)
    assert seen == [3, 3, 3]
    assert tr.train("linear,linear", 2048).pipeline == 3              # (another leaf count: routed by its own probes)
    tr.set_keys(dg.uniform_u64(400_000))
    assert [tr.train("linear,linear", 4096).pipeline for _ in range(2)] == [4, 4]
    tr.close()
    monkeypatch.setenv("RMI_HIP_REGS_BACKOFF", "0")
    tr = train.Trainer(keys)
    assert [tr.train("linear,linear", 4096).pipeline for _ in range(2)] == [4, 4]
    tr.close()


@pytest.mark.parametrize("gen,n,L", [("uniform_u64", 1_000_000, 8192), ("books_u64", 1_000_000, 20_000), ("uniform_f64", 300_000, 4096), ("dups_u64", 300_000, 4096),
                                     ("clustered_u64", 300_000, 2048), ("uniform_u64", 2_000_000, 10_000)])
@pytest.mark.parametrize("env,expect", [({}, (3, 4)), ({"RMI_HIP_CUBIC_MARGIN": "0"}, (3,)), ({"RMI_HIP_CUBIC_MARGIN_SCALE": "1e13"}, (3, 4))])
def test_regs_cubic_root_by_margin(monkeypatch, oracle, gen, n, L, env, expect):
    """A cubic root on pipeline 4: the host proves the exact polynomial increasing over the keys' range, k_regs_finalize<K, K_CUBIC> that every
    leaf's end keys clear their leaf's interval by the rounding bound of the three fmas -- O(L) instead of a root evaluation per key.
    RMI_HIP_CUBIC_MARGIN=0: the per-key verification in k_leaf_lanes (pipeline 3); a margin widened by 1e13 leaves every leaf undecided:
    each is then verified key by key by its wave in k_regs_finalize.  Same bits every way, and the oracle's."""
    g = _check(monkeypatch, oracle, dict({"RMI_HIP_REGS": "1"}, **env), dg.GENERATORS[gen](n), "cubic", L)
    if g is not None:
        assert g.pipeline in expect
