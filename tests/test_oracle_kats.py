"""Pins the CPU oracle against every known-answer test the reference holds for this path.

The reference's in-file unit tests are stale (written against a removed ModelData API and never
run by its CI -- SURVEY.md section 4) but their inputs/expected values are still meaningful for
the current arithmetic; they are the only numeric pins that exist ("parity unpinned").
Citations: /root/reference/rmi_lib/src/models/*.rs.
"""
import numpy as np
import pytest


def u64(*v):
    return np.array(v, dtype=np.uint64)


def test_linear1(oracle):
    # linear.rs:127-134
    m = oracle.fit_pairs("linear", u64(1, 2, 3), [2, 3, 4])
    assert m.predict_to_int(1) == 2
    assert m.predict_to_int(6) == 7


def test_linear_single(oracle):
    # linear.rs:137-143
    m = oracle.fit_pairs("linear", u64(1), [2])
    assert m.predict_to_int(1) == 2


def test_linear_spline1(oracle):
    # linear_spline.rs:90-97
    m = oracle.fit_pairs("linear_spline", u64(1, 2, 3), [2, 3, 8])
    assert m.predict_to_int(1) == 2
    assert m.predict_to_int(3) == 8


def test_linear_spline_single(oracle):
    # linear_spline.rs:100-106
    m = oracle.fit_pairs("linear_spline", u64(1), [2])
    assert m.predict_to_int(1) == 2


@pytest.mark.parametrize("keys,ys,lo,hi", [
    ((1, 2, 3, 4), (2, 3, 8, 20), (1, 2.0), (4, 20.0)),            # cubic_spline.rs:200-207
    ((1, 2, 3, 4, 5), (2, 3, 8, 20, 80), (1, 2.0), (5, 80.0)),      # :210-217
    ((1, 1, 3, 4, 5), (2, 2, 8, 20, 80), (1, 2.0), (5, 80.0)),      # :220-227 (dup)
])
def test_cubic_endpoints(oracle, keys, ys, lo, hi):
    m = oracle.fit_pairs("cubic", u64(*keys), list(ys))
    assert abs(m.predict_to_float(lo[0]) - lo[1]) <= 0.5
    assert abs(m.predict_to_float(hi[0]) - hi[1]) <= 0.5


def test_cubic_all_dup_and_single(oracle):
    # cubic_spline.rs:230-245
    m = oracle.fit_pairs("cubic", u64(1, 1, 1), [2, 2, 2])
    assert abs(m.predict_to_float(1) - 2.0) <= 0.5
    m = oracle.fit_pairs("cubic", u64(1), [2])
    assert m.predict_to_int(1) == 2


@pytest.mark.parametrize("kind,expect", [
    ("linear", (0.0, 0.0, 0.0, 0.0)),           # linear.rs:37-39
    ("linear_spline", (0.0, 0.0, 0.0, 0.0)),    # linear_spline.rs:14-16
    ("cubic", (0.0, 0.0, 1.0, 0.0)),            # cubic_spline.rs:19-21
    ("robust_linear", (0.0, 0.0, 0.0, 0.0)),    # linear.rs:241-245
])
def test_empty(oracle, kind, expect):
    # test_empty in every model file: models must accept empty data
    m = oracle.fit_pairs(kind, u64(), [])
    assert m.p == expect


def test_radix_empty(oracle):
    m = oracle.fit_pairs("radix", u64(), [])
    assert m.ip[:2] == (0, 0)
    m = oracle.fit_pairs("bradix", u64(), [])                 # balanced_radix.rs:91-96, :176-179 (test_empty)
    assert m.ip == (0, 0, 0, 1)


def test_common_prefix_size(oracle):
    # utils.rs:110-126
    assert oracle.common_prefix_size(u64(1, 4, 8)) == 60
    assert oracle.common_prefix_size(u64(1, 8, 9, 12)) == 60


def test_num_bits(oracle):
    # utils.rs:13-21: floor(log2(t+1)); asserts >= 1
    assert oracle.num_bits(0) == -1
    assert oracle.num_bits(1) == 1
    assert oracle.num_bits(2) == 1
    assert oracle.num_bits(3) == 2
    assert oracle.num_bits(1023) == 10
    assert oracle.num_bits(1024) == 10
    assert oracle.num_bits((1 << 20) - 1) == 20


def test_fixdups_tail_duplicate_q1(oracle):
    """Q1 (models/mod.rs:170-181): iter() yields len+1 items, last one twice.  slr over
    {(0,0),(1,1),(2,5)} therefore sees (2,5) twice; check against a direct Welford run."""
    pts = [(0.0, 0.0), (1.0, 1.0), (2.0, 5.0), (2.0, 5.0)]
    mx = my = c = m2 = 0.0
    n = 0
    for x, y in pts:
        n += 1
        dx = x - mx
        mx += dx / n
        my += (y - my) / n
        c += dx * (y - my)
        m2 += dx * (x - mx)
    beta = (c / (n - 1)) / (m2 / (n - 1))
    alpha = my - beta * mx
    m = oracle.fit_pairs("linear", u64(0, 1, 2), [0, 1, 5])
    assert m.p[0] == alpha and m.p[1] == beta


def test_rmi_size_readme_sample():
    # README.md:51 : RMI_SIZE 50331680 = 32 (cubic root) + 24 * 2^21 (codegen.rs:375-394)
    assert 4 * 8 + (2 * 8 + 8) * (1 << 21) == 50331680


def test_worked_example_survey(oracle):
    """SURVEY.md section 8a worked example (Q1-Q3) with a toy monotone linear root."""
    keys = u64(10, 11, 12, 20, 21, 30, 30, 31, 40, 41, 42, 50)
    # root t(k) = floor((k-10)/11): as linear model alpha=-10/11, beta=1/11 is inexact, so use a
    # radix-free exact form: beta = 1/16, alpha = -0.625 -> targets floor((k-10)/16)
    root = oracle.Model(oracle.MODEL_LINEAR, (-0.625, 0.0625, 0.0, 0.0), (0, 0))
    ids = oracle.bucket_ids(root, keys, 4)
    assert ids.tolist() == [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2]
    r = oracle.train_two_layer("linear", "linear_spline", keys, 4, root=root)
    assert r.leaf_start.tolist() == [0, 5, 10, 12, 12]
    # leaf 0 (first half, ends before split_idx=10): container = own [0..4] + next-first (30, 5)
    # linear_spline uses get(0)/get(len-1): (10,0) and (30,5) -> slope 5/20
    assert r.leaf_params[0, 1] == (0.0 - 5.0) / (10.0 - 30.0)
    # leaf 1: prev-last (21,4) + own (30,5),(30,5),(31,7),(40,8),(41,9); no next (half boundary, Q3)
    assert r.leaf_params[1, 1] == (4.0 - 9.0) / (21.0 - 41.0)
    # leaf 2 = st: key at split_idx (42,10) dropped (Q2); own = (50,11) only, no prev (Q3) -> single point
    assert r.leaf_params[2].tolist() == [11.0, 0.0]
    # leaf 3: empty and last (Q6) -> stays (0,0)
    assert r.leaf_params[3].tolist() == [0.0, 0.0]
    assert oracle.check_lookup_property(r, keys)[0] == 0


def test_radix_table_against_direct_restatement(oracle):
    """RadixTable::new (radix.rs:90-121) has no unit test in the reference; pin the oracle's table
    against a direct, independent transcription of those lines on small inputs."""
    import numpy as np
    rng = np.random.default_rng(5)
    for trial in range(6):
        n = int(rng.integers(2, 400))
        keys = np.sort(rng.integers(1, 1 << int(rng.integers(10, 64)), size=n, dtype=np.uint64))
        if trial % 2:
            keys[n // 2:n // 2 + 3] = keys[n // 2]           # duplicate run
            keys = np.sort(keys)
        L = int(rng.integers(2, 300))
        m = oracle.fit_root("radix8", keys, L)
        ks = [int(k) for k in keys]
        any_ones, no_ones = 0, (1 << 64) - 1
        for k in ks:
            any_ones |= k
            no_ones &= k
        agree = ((~no_ones) ^ any_ones) & ((1 << 64) - 1)     # bits on which all keys agree
        prefix = 64
        for b in range(64):
            if not (agree >> (63 - b)) & 1:
                prefix = b
                break
        bits, scale = 8, L / n
        table = [0] * (1 << bits)
        nb = 0 if prefix + bits > 64 else 64 - (prefix + bits)
        last = 0
        items, first = [], 0
        for i, k in enumerate(ks):                            # FixDups: first-occurrence offsets, tail duplicate
            if i == 0 or k != ks[i - 1]:
                first = i
            items.append((k, first))
        items.append(items[-1])
        for k, y in items:
            ys = int(float(y) * scale) if abs(scale - 1.0) > 2.220446049250313e-16 else y
            cur = (((k << (prefix & 63)) & ((1 << 64) - 1)) >> (prefix & 63)) >> (nb & 63)
            if cur == last:
                continue
            table[cur] = ys
            for j in range(last + 1, cur):
                table[j] = ys
            last = cur
        for j in range(last + 1, len(table)):
            table[j] = len(table)
        assert m.ip[:2] == (prefix, bits)
        assert [int(v) for v in m.table] == table


def test_bradix_against_direct_restatement(oracle):
    """BalancedRadixModel (balanced_radix.rs:20-101) has no numeric unit test in the reference; pin the
    oracle against an independent transcription on small inputs (release-build arithmetic: the
    `max_output - bits_max` of :63 wraps)."""
    import numpy as np
    M64 = (1 << 64) - 1

    def predict(prefix, bits, clamp, high, x):                # :104-116
        res = ((x << (prefix & 63)) & M64) >> ((64 - bits) & 63)
        if high:
            return min(res, clamp)
        return 0 if res < clamp else res - clamp

    def fit(ks, L):
        n = len(ks)
        scale = L / n
        items = []
        for i, k in enumerate(ks):                            # FixDups + Q1
            if i == 0 or k != ks[i - 1]:
                first = i
            items.append((k, int(float(first) * scale) if abs(scale - 1.0) > 2.220446049250313e-16 else first))
        items.append(items[-1])
        max_output = max(y for _, y in items)
        bits = 0
        while ((1 << (bits + 1)) - 1) <= max_output:
            bits += 1
        assert bits >= 1
        any_ones, no_ones = 0, M64
        for k in ks:
            any_ones |= k
            no_ones &= k
        agree = ((~no_ones) ^ any_ones) & M64
        prefix = 64
        for b in range(64):
            if not (agree >> (63 - b)) & 1:
                prefix = b
                break
        best, best_score = None, float("inf")
        for tb in range(bits, min(bits + 2, 64)):
            bits_max = (1 << (tb + 1)) - 1
            for high, clamp in ((1, max_output - 1), (0, (max_output - bits_max) & M64)):
                counts = [0] * max_output
                for k, _ in items:
                    counts[predict(prefix, tb, clamp, high, k)] += 1
                expected = n / max_output
                score = 0.0
                for c in counts:
                    score += ((c - expected) * (c - expected)) / expected
                if score < best_score:
                    best_score, best = score, (prefix, tb, clamp, high)
        return best

    # worked example: 16 keys i << 4, L = 8 -> max_output 7, bits 3, prefix 56; high/3 bits wins
    keys = np.array([i << 4 for i in range(16)], dtype=np.uint64)
    assert oracle.fit_root("bradix", keys, 8).ip == (56, 3, 6, 1) == fit([int(k) for k in keys], 8)
    rng = np.random.default_rng(11)
    for trial in range(12):
        n = int(rng.integers(20, 600))
        keys = np.sort(rng.integers(1, 1 << int(rng.integers(10, 64)), size=n, dtype=np.uint64))
        if trial % 3 == 1:
            keys[n // 2:n // 2 + 5] = keys[n // 2]
            keys = np.sort(keys)
        if trial % 3 == 2:                                    # skewed: most keys share their leading bits
            keys[: n - 3] = np.sort(rng.integers(1, 1 << 12, size=n - 3, dtype=np.uint64))
            keys = np.sort(keys)
        L = int(rng.integers(4, 200))
        m = oracle.fit_root("bradix", keys, L)
        assert m.ip == fit([int(k) for k in keys], L), (trial, m)
        for k in (int(keys[0]), int(keys[n // 3]), int(keys[-1])):
            assert m.predict_to_int(k) == predict(*m.ip, k)


def test_normal_and_loglinear_reference_kats(oracle):
    """normal.rs:131-139 (test_ncdf1) and linear.rs:216-224 (loglin test).  Both tests predate FixDupsIter's
    tail duplicate (Q1): their expected values hold for parameters fitted WITHOUT the repeated last
    item, which pins exp1 / phi / predict_to_int; the fits themselves include the repeated item, like
    every model fitted through data.iter() today (models/mod.rs:180)."""
    import math
    # ncdf over (1,1),(2,3),(3,5) without Q1: mean 2, stdev sqrt(2/3), scale 5
    m = oracle.Model(6, (2.0, math.sqrt(2.0 / 3.0), 5.0, 0.0), (0, 0))
    assert m.predict_to_int(2) == 2 and m.predict_to_int(1) == 0
    # loglinear over (2,2),(3,4),(4,16) without Q1: slr of (x, ln y)
    xs, ys = [2.0, 3.0, 4.0], [math.log(2.0), math.log(4.0), math.log(16.0)]
    mx, my = sum(xs) / 3, sum(ys) / 3
    beta = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    m = oracle.Model(5, (my - beta * mx, beta, 0.0, 0.0), (0, 0))
    assert m.predict_to_int(2) == 1 and m.predict_to_int(4) == 13
    # the fits with Q1: four items (the last twice), n = len = 3 in ncdf (normal.rs:34)
    f = oracle.fit_pairs("normal", u64(1, 2, 3), [1, 3, 5])
    assert f.p[0] == 1 / 3 + 2 / 3 + 3 / 3 + 3 / 3 and f.p[2] == 5.0
    assert f.p[1] == math.sqrt(((1 - f.p[0]) ** 2 + (2 - f.p[0]) ** 2 + (3 - f.p[0]) ** 2 + (3 - f.p[0]) ** 2) / 3)
    g = oracle.fit_pairs("loglinear", u64(2, 3, 4), [0, 4, 16])           # ln 0 = -inf: the item is dropped (linear.rs:65)
    h = oracle.fit_pairs("loglinear", u64(3, 4), [4, 16])
    assert g.p == h.p
