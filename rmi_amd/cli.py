"""`rmi`-compatible command line (reference: src/main.rs:33-340).

    python -m rmi_amd.cli <input> [namespace] [models] [branching factor] [flags]

Same positional arguments and the flags that concern the single-train and param-grid modes:
--no-code, --no-errors, -d/--data-path, -t/--threads (accepted, unused: the work runs on the GPU),
--zero-build-time, --param-grid, --disable-parallel-training, --optimize <file> (Pareto search,
src/main.rs:134-163) and --max-size <bytes> (train_for_size, :276-294).  The data type comes from a
substring of the input path (uint64 / uint32 / f64, src/main.rs:122-132).  --bounded <line_size>
(cache-fix, u64 data only, src/main.rs:277-285) trains over the spline points.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

from . import codegen, datagen, optimizer, train


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="rmi", description="MI355X-native RMI trainer (two-layer)")
    ap.add_argument("input", help="Path to input file containing data")
    ap.add_argument("namespace", nargs="?", help="Namespace to use in generated code")
    ap.add_argument("models", nargs="?", help='Comma-separated list of model layers, e.g. linear,linear')
    ap.add_argument("branching_factor", nargs="?", type=int, help="Branching factor between each model level")
    ap.add_argument("--no-code", action="store_true", help="Skip code generation")
    ap.add_argument("--no-errors", action="store_true", help="Do not save last-level errors, and modify the RMI function signature")
    ap.add_argument("-d", "--data-path", default="rmi_data", help="exports parameters to files stored in this directory")
    ap.add_argument("-t", "--threads", type=int, default=4, help="host threads for the optimizer's root fits, default = 4")
    ap.add_argument("--zero-build-time", action="store_true", help="zero out the model build time field")
    ap.add_argument("--param-grid", help="train the RMIs specified in the JSON file and report their errors")
    ap.add_argument("--disable-parallel-training", action="store_true", help="accepted for compatibility")
    ap.add_argument("--optimize", metavar="file", help="Search for Pareto efficient RMI configurations. Specify the name of the output file.")
    ap.add_argument("--max-size", metavar="BYTES", type=int, help="uses the optimizer to find an RMI with a size less than specified")
    ap.add_argument("--bounded", metavar="line_size", type=int,
                    help="construct an error-bounded RMI using the cachefix method for the given line size")
    # declared by the reference's CLI and never read there either (src/main.rs:55-66): accepted, no effect
    ap.add_argument("--dump-ll-model-data", metavar="model_index", help="accepted for compatibility (unused in the reference too)")
    ap.add_argument("--dump-ll-errors", action="store_true", help="accepted for compatibility (unused in the reference too)")
    ap.add_argument("-s", "--stats-file", metavar="file", help="accepted for compatibility (unused in the reference too)")
    ap.add_argument("--fast-root", action="store_true",
                    help="(extension) fit linear / robust_linear roots from parallel sums on the GPU: not bit-identical to the reference")
    ap.add_argument("--device", type=int, default=0)
    return ap


def _stats(rmi: train.TrainedRMI, n: int) -> dict:
    # src/main.rs:207-221, key for key ("average error %" is the MAX error over the row count there, :211-212)
    return {"layers": rmi.models, "branching factor": rmi.branching_factor,
            "average error": rmi.model_avg_error, "average error %": rmi.model_max_error / n * 100.0,
            "average l2 error": rmi.model_avg_l2_error, "average log2 error": rmi.model_avg_log2_error,
            "max error": rmi.model_max_error, "max error %": rmi.model_max_error / n * 100.0,
            "max log2 error": rmi.model_max_log2_error,
            "size binary search": codegen.rmi_size(rmi.root.kind, rmi.leaf_kind, rmi.branching_factor, True,
                                                      0 if rmi.root.table is None else len(rmi.root.table))}


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    if args.namespace and args.param_grid:
        print("Can only specify one of namespace or param-grid", file=sys.stderr)      # src/main.rs:116-118
        return 2
    keys = datagen.read_keys(args.input)                                              # src/load.rs:132-157
    key_c = "double" if keys.dtype == np.float64 else "uint64_t"                       # src/main.rs:122-132
    keys = np.ascontiguousarray(keys)
    # a single exact training takes the keys from host memory with the upload overlapped (Trainer.train_from_host);
    # the optimizer, a parameter grid and the bounded mode train many times on resident keys
    single = not args.optimize and not args.param_grid and args.bounded is None and args.max_size is None and not args.fast_root
    tr = train.Trainer(device=args.device) if single else train.Trainer(keys, device=args.device)
    n = len(keys)
    try:
        if args.optimize:                                                              # src/main.rs:134-163
            if optimizer.skipped_models():
                print("not on the device path, left out of the search: " + ", ".join(optimizer.skipped_models()), file=sys.stderr)
            results = optimizer.find_pareto_efficient_configs(tr, 10, threads=args.threads,
                                                              root_mode="fast" if args.fast_root else "exact")
            optimizer.display_table(results)
            prefix = args.namespace or os.path.basename(args.input) or "rmi"
            specs = [r.to_grid_spec(f"{prefix}_{i}") for i, r in enumerate(results)]
            with open(args.optimize, "w") as f:
                json.dump({"configs": specs}, f)
            return 0
        if args.param_grid:                                                            # src/main.rs:171-261
            grid = json.load(open(args.param_grid))
            results = []
            for cfg in grid["configs"]:
                rmi = tr.train(cfg["layers"], int(cfg["branching factor"]))
                res = _stats(rmi, n)
                res["namespace"] = cfg.get("namespace")                                # src/main.rs:207-221: null when absent
                if "namespace" in cfg:
                    # (grid mode always emits with errors, src/main.rs:232-238)
                    codegen.output_rmi(cfg["namespace"], rmi, args.data_path, key_type=key_c, include_errors=True,
                                       build_time_ns=0 if args.zero_build_time else None)
                results.append(res)
            with open(f"{args.param_grid}_results", "w") as f:
                json.dump({"results": results}, f)                                     # src/main.rs:254-257
            return 0
        if not args.namespace:
            print("Must specify either a name space or a parameter grid.", file=sys.stderr)
            return 2
        if args.max_size is not None:                                                  # src/main.rs:286-292
            print(f"Constructing RMI with size less than {args.max_size}")
            rmi = optimizer.train_for_size(tr, args.max_size, threads=args.threads, root_mode="fast" if args.fast_root else "exact")
            print(f"Found RMI config {rmi.models} {rmi.branching_factor}")
        elif not args.models or args.branching_factor is None:
            print("models and branching factor are required", file=sys.stderr)
            return 2
        elif args.bounded is not None:                                                 # src/main.rs:277-285
            if keys.dtype != np.uint64:
                print("Can only construct a bounded RMI on u64 data.", file=sys.stderr)
                return 1
            rmi = tr.train_bounded(args.models, args.branching_factor, args.bounded)
        else:
            if single:
                rmi = tr.train_from_host(keys, args.models, args.branching_factor)
            else:
                rmi = tr.train(args.models, args.branching_factor, root_mode="fast" if args.fast_root else "exact")
        print(f"Model build time: {rmi.build_time // 1_000_000} ms (device {rmi.device_ns / 1e6:.3f} ms)")
        print(f"Average model error: {rmi.model_avg_error} ({rmi.model_avg_error / n * 100.0}%)")
        print(f"Average model L2 error: {rmi.model_avg_l2_error}")
        print(f"Average model log2 error: {rmi.model_avg_log2_error}")
        print(f"Max model log2 error: {rmi.model_max_log2_error}")
        print(f"Max model error on model {rmi.model_max_error_idx}: {rmi.model_max_error} ({rmi.model_max_error / n * 100.0}%)")
        if not args.no_code:
            codegen.output_rmi(args.namespace, rmi, args.data_path, key_type=key_c, include_errors=not args.no_errors,
                               build_time_ns=0 if args.zero_build_time else None)
        return 0
    except train.RMIError as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    finally:
        tr.close()


if __name__ == "__main__":
    sys.exit(main())
