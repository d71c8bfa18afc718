#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small key sets with the expected outputs of the hot path.

Provenance, stated plainly: the reference is Rust and cannot be built or run in this image (no
cargo / rustc), and its own tests hold no numeric vectors for this path.  These
vectors are therefore produced by the CPU restatement in oracle/ (which is pinned by the reference's
in-file KATs, by independent transcriptions and by the reference's acceptance property, see
tests/test_oracle_kats.py and tests/test_oracle_property.py).  They freeze that behaviour: the oracle
must keep reproducing them (tests/test_golden.py, CPU) and the HIP path must match them bit for bit
(same file, GPU) -- a fixed target that does not move when the oracle is edited.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as oracle          # noqa: E402
from rmi_amd import datagen as dg             # noqa: E402

CASES = [
    # name, generator, n, root, leaf, L
    ("uniform_u64_linear_linear", "uniform_u64", 6000, "linear", "linear", 64),
    ("dups_u64_cubic_linear", "dups_u64", 5000, "cubic", "linear", 128),
    ("uniform_u32_radix_spline", "uniform_u32", 6000, "radix", "linear_spline", 256),
    ("books_u64_robust_linear", "books_u64", 8000, "robust_linear", "linear", 32),
    ("clustered_u64_spline_linear", "clustered_u64", 4000, "linear_spline", "linear", 50),
    ("dups_u32_linear_cubic", "dups_u32", 5000, "linear", "cubic", 40),
    ("books_u64_radix18_linear", "books_u64", 6000, "radix18", "linear", 100),
    ("uniform_u64_bradix_linear", "uniform_u64", 5000, "bradix", "linear", 128),
    ("uniform_u64_normal_linear", "uniform_u64", 5000, "normal", "linear", 64),
    ("uniform_u64_loglinear_linear", "uniform_u64", 5000, "loglinear", "linear", 64),
    ("uniform_f64_linear_linear", "uniform_f64", 4000, "linear", "linear", 48),
    ("dups_u64_linear_robust", "dups_u64", 6000, "linear", "robust_linear", 24),
]


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, gen, n, root, leaf, L in CASES:
        keys = dg.GENERATORS[gen](n)
        o = oracle.train_two_layer(root, leaf, keys, L)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            keys=keys, root=np.array(root), leaf=np.array(leaf), num_leaves=np.int64(L),
            root_p=np.array(o.root.p, dtype=np.float64).view(np.uint64), root_ip=np.array(o.root.ip, dtype=np.uint64),
            root_table=(o.root.table if o.root.table is not None else np.zeros(0, dtype=np.uint32)),
            leaf_start=o.leaf_start, leaf_params=o.leaf_params.view(np.uint64), leaf_err=o.leaf_err, leaf_count=o.leaf_count,
            model_max_error=np.uint64(o.model_max_error), model_max_error_idx=np.uint64(o.model_max_error_idx),
            model_avg_error=np.float64(o.model_avg_error))
        print(name, "ok: max error", o.model_max_error)
    # the worked example of SURVEY.md section 8a (Q1-Q3), linear root
    keys = np.array([10, 11, 12, 20, 21, 30, 30, 31, 40, 41, 42, 50], dtype=np.uint64)
    o = oracle.train_two_layer("linear", "linear", keys, 4)
    np.savez_compressed(os.path.join(out_dir, "survey_example_linear_linear.npz"), keys=keys, root=np.array("linear"),
                        leaf=np.array("linear"), num_leaves=np.int64(4),
                        root_p=np.array(o.root.p, dtype=np.float64).view(np.uint64), root_ip=np.array(o.root.ip, dtype=np.uint64),
                        root_table=np.zeros(0, dtype=np.uint32),
                        leaf_start=o.leaf_start, leaf_params=o.leaf_params.view(np.uint64), leaf_err=o.leaf_err, leaf_count=o.leaf_count,
                        model_max_error=np.uint64(o.model_max_error), model_max_error_idx=np.uint64(o.model_max_error_idx),
                        model_avg_error=np.float64(o.model_avg_error))


if __name__ == "__main__":
    main()
