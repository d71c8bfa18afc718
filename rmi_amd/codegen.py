"""Emission of the trained RMI: `<ns>.cpp`, `<ns>.h`, `<ns>_data.h` and the binary parameter files
`<data_dir>/<ns>_L{i}_PARAMETERS` -- the reference's artefact format, so the output is a drop-in
(SURVEY.md section 8b "wire format", section 8f-1).

Reference: rmi_lib/src/codegen.rs (LayerParams :24-333, rmi_size :375-394, generate_code :450-754,
output_rmi :757-788) and rmi_lib/src/models/mod.rs:510-674 (ModelParam: c_val, write_to) plus the
`code()` / `function_name()` of the plugins (linear.rs:103-114, cubic_spline.rs:169-183,
radix.rs:64-74).  The reference iterates HashSets when it emits declarations, so its text order
varies run to run; this writer emits them in a fixed order.  What must match is the
binary parameter file (byte for byte: it is the `rows` buffer the kernels produce) and the
behaviour of `load()` / `lookup()` / `cleanup()`.
"""
from __future__ import annotations

import os
from decimal import Decimal

import numpy as np

LINEAR, LINEAR_SPLINE, CUBIC, RADIX, ROBUST_LINEAR = 0, 1, 2, 3, 4
RADIX_TABLES = (8, 9, 10, 11, 12)            # radix8/18/22/26/28 (RadixTable, radix.rs:83-170)
BRADIX = 13                                  # BalancedRadixModel (balanced_radix.rs)
LOGLINEAR, NORMAL = 5, 6                     # linear.rs:152-210, normal.rs:70-126 (roots)

# StdFunctions (models/stdlib.rs:29-45), emitted before the model functions that use them
_STD_CODE = {
    "exp1": """
inline double exp1(double x) {
  x = 1.0 + x / 64.0;
  x *= x; x *= x; x *= x; x *= x;
  x *= x; x *= x;
  return x;
}""",
    "phi": """
inline double phi(double x) {
  return 1.0 / (1.0 + exp1(- 1.65451 * x));
}""",
}
_STD_NEEDED = {LOGLINEAR: ("exp1",), NORMAL: ("exp1", "phi")}     # linear.rs:205-209, normal.rs:112-117

_MODEL_CODE = {
    "linear": """
inline double linear(double alpha, double beta, double inp) {
    return std::fma(beta, inp, alpha);
}""",
    "cubic": """
inline double cubic(double a, double b, double c, double d, double x) {
    auto v1 = std::fma(a, x, b);
    auto v2 = std::fma(v1, x, c);
    auto v3 = std::fma(v2, x, d);
    return v3;
}""",
    "radix": """
inline uint64_t radix(uint64_t prefix_length, uint64_t bits, uint64_t inp) {
    return (inp << prefix_length) >> (64 - bits);
}""",
    "loglinear": """
inline double loglinear(double alpha, double beta, double inp) {
    return exp1(std::fma(beta, inp, alpha));
}""",
    "ncdf": """
inline double ncdf(double mean, double stdev, double scale, double inp) {
    return phi((inp - mean) / stdev) * scale;
}""",
    # balanced_radix.rs:132-152
    "bradix_clamp_high": """
inline uint64_t bradix_clamp_high(uint64_t prefix_length,
                                  uint64_t bits, uint64_t clamp, uint64_t inp) {
    uint64_t tmp = (inp << prefix_length) >> (64 - bits);
    return (tmp > clamp ? clamp : tmp);
}""",
    "bradix_clamp_low": """
inline uint64_t bradix_clamp_low(uint64_t prefix_length,
                                 uint64_t bits, uint64_t clamp, uint64_t inp) {
    uint64_t tmp = (inp << prefix_length) >> (64 - bits);
    return (tmp < clamp ? 0 : tmp - clamp);
}""",
}


def _fn_name(kind: int, model=None) -> str:
    if kind in RADIX_TABLES:
        return "radix_table"
    if kind == BRADIX:                                                   # balanced_radix.rs:155-161
        return "bradix_clamp_high" if int(model.ip[3]) else "bradix_clamp_low"
    return {LINEAR: "linear", LINEAR_SPLINE: "linear", ROBUST_LINEAR: "linear", CUBIC: "cubic", RADIX: "radix",
            LOGLINEAR: "loglinear", NORMAL: "ncdf"}[kind]


def _output_is_float(kind: int) -> bool:
    return kind not in (RADIX, BRADIX) and kind not in RADIX_TABLES


def _needs_bounds_check(kind: int) -> bool:
    # cubic_spline.rs:184-186, radix.rs:72-74, :160-162, balanced_radix.rs:164-166
    return kind not in (CUBIC, RADIX, BRADIX) and kind not in RADIX_TABLES


def _radix_table_code(prefix: int, table_bits: int) -> str:              # radix.rs:140-153
    num_bits = 0 if prefix + table_bits > 64 else 64 - (prefix + table_bits)
    return f"""
inline uint64_t radix_table(const uint32_t* table, const uint64_t inp) {{
    return table[((inp << {prefix}) >> {prefix}) >> {num_bits}];
}}"""


def c_float(v: float) -> str:
    """ModelParam::Float::c_val (models/mod.rs:568-574): Rust's Display for f64 -- shortest digits
    that round-trip, positional notation, never an exponent -- with ".0" appended if there is no '.'."""
    v = float(v)
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    s = format(Decimal(repr(v)), "f")
    if s.startswith("-") and float(s) == 0.0:
        s = "-0"
    if "." not in s:
        s += ".0"
    return s


def rmi_size(root_kind: int, leaf_kind: int, num_leaves: int, with_errors: bool, root_table_entries: int = 0,
             spline_points: int = 0) -> int:
    """codegen.rs:375-394 (two layers).  A radix-table root is its hint table (4 B per entry); a
    bounded RMI adds 16 B per spline point (:389-391)."""
    root_bytes = 4 * root_table_entries if root_kind in RADIX_TABLES else {CUBIC: 32, RADIX: 16, BRADIX: 24, NORMAL: 24}.get(root_kind, 16)
    leaf_bytes = 32 if leaf_kind == CUBIC else 16
    return root_bytes + leaf_bytes * num_leaves + (8 * num_leaves if with_errors else 0) + 16 * spline_points


def output_rmi(namespace: str, rmi, data_dir: str, key_type: str = "uint64_t", include_errors: bool = True,
               out_dir: str = ".", build_time_ns: int | None = None) -> dict:
    """`rmi`: an object with .root (kind, p, ip), .leaf_kind, .params_per_leaf, .branching_factor,
    .num_rmi_rows, .leaf_params [L, ppl] f64, .last_layer_max_l1s [L] u64 and (optionally) .rows,
    .build_time -- i.e. rmi_amd.train.TrainedRMI.  Returns the paths written."""
    L = int(rmi.branching_factor)
    n = int(rmi.num_rmi_rows)
    root = rmi.root
    ppl = int(rmi.params_per_leaf)
    leaf_kind = int(rmi.leaf_kind)
    os.makedirs(data_dir, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    paths = {}

    data_h = [f"namespace {namespace} {{"]
    read_code = ["bool load(char const* dataPath) {"]
    free_code = ["void cleanup() {"]

    # ---- layer 0: a single model; all-same-typed, <= 4096 bytes -> Constant (codegen.rs:45-63) ----
    root_table_args = None
    if root.kind in RADIX_TABLES:
        table = np.ascontiguousarray(root.table, dtype="<u4")
        if table.size * 4 <= 4096:                           # Constant: literal array (codegen.rs:68-77, mod.rs:594-597)
            lits = ", ".join(f"{int(v)}UL" for v in table)
            data_h.append(f"const uint32_t L0_PARAMETER0[] = {{ {lits} }};")
            root_table_args = "L0_PARAMETER0"
        else:                                                # Array: binary file, malloc'd (codegen.rs:79-93, 104-113)
            f0 = f"{namespace}_L0_PARAMETERS"
            with open(os.path.join(data_dir, f0), "wb") as f:
                f.write(table.tobytes())
            paths["L0_PARAMETERS"] = os.path.join(data_dir, f0)
            data_h.append("uint32_t* L0_PARAMETERS;")
            read_code += ["  {",
                          f"    std::ifstream infile(std::filesystem::path(dataPath) / \"{f0}\", std::ios::in | std::ios::binary);",
                          "    if (!infile.good()) return false;",
                          f"    L0_PARAMETERS = (uint32_t*) malloc({table.size * 4});",
                          "    if (L0_PARAMETERS == NULL) return false;",
                          f"    infile.read((char*)L0_PARAMETERS, {table.size * 4});",
                          "    if (!infile.good()) return false;", "  }"]
            free_code.append("    free(L0_PARAMETERS);")
            root_table_args = "L0_PARAMETERS"
        root_vals, root_ctype = [], "uint32_t"
    elif root.kind == RADIX:
        root_vals = [f"{int(root.ip[0])}UL", f"{int(root.ip[1])}UL"]
        root_ctype = "uint64_t"
    elif root.kind == BRADIX:                                # params(): (prefix, bits, clamp), balanced_radix.rs:124-130
        root_vals = [f"{int(root.ip[i])}UL" for i in range(3)]
        root_ctype = "uint64_t"
    else:
        nroot = {CUBIC: 4, NORMAL: 3}.get(root.kind, 2)
        root_vals = [c_float(root.p[i]) for i in range(nroot)]
        root_ctype = "double"
    for i, v in enumerate(root_vals):
        data_h.append(f"const {root_ctype} L0_PARAMETER{i} = {v};")

    # ---- layer 1 ----
    errs = np.ascontiguousarray(rmi.last_layer_max_l1s, dtype=np.uint64)
    params = np.ascontiguousarray(rmi.leaf_params, dtype=np.float64).reshape(L, ppl)
    fname = f"{namespace}_L1_PARAMETERS"
    fpath = os.path.join(data_dir, fname)
    if include_errors and L > 1:
        # with_zipped_errors (codegen.rs:288-315): rows (params..., Int err) -> mixed -> MixedArray
        rows = getattr(rmi, "rows", None)
        if rows is None:
            buf = np.empty((L, ppl + 1), dtype="<u8")
            buf[:, :ppl] = params.view(np.uint64)
            buf[:, ppl] = errs
            rows = buf.view(np.uint8).reshape(-1)
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        row_bytes = ppl * 8 + 8
        assert rows.size == L * row_bytes
        with open(fpath, "wb") as f:
            f.write(rows.tobytes())
        layer1_size = L * row_bytes
        data_h.append("char* L1_PARAMETERS;")
        ptr_type = "char"
        malloc = True

        def acc(p_idx: int) -> str:                         # access_by_ref, MixedArray (codegen.rs:259-282)
            ctype = "uint64_t" if p_idx == ppl else "double"
            return f"*(({ctype}*) (L1_PARAMETERS + (modelIndex * {row_bytes}) + {p_idx * 8}))"
        err_line = f"  *err = {acc(ppl)};"
    else:
        with open(fpath, "wb") as f:
            f.write(params.astype("<f8").tobytes())
        layer1_size = L * ppl * 8
        malloc = layer1_size >= 4 * 1024                    # requires_malloc (codegen.rs:104-113)
        ptr_type = "double"
        data_h.append("double* L1_PARAMETERS;" if malloc else f"double L1_PARAMETERS[{L * ppl}];")

        def acc(p_idx: int) -> str:                         # access_by_ref, Array (codegen.rs:250-257)
            return f"L1_PARAMETERS[{ppl}*modelIndex + {p_idx}]"
        err_line = f"  *err = {int(errs[0])};" if (include_errors and L == 1) else ""
    paths["L1_PARAMETERS"] = fpath
    read_code += [
        "  {",
        f"    std::ifstream infile(std::filesystem::path(dataPath) / \"{fname}\", std::ios::in | std::ios::binary);",
        "    if (!infile.good()) return false;",
    ]
    if malloc:
        read_code += [f"    L1_PARAMETERS = ({ptr_type}*) malloc({layer1_size});",
                      "    if (L1_PARAMETERS == NULL) return false;"]
        free_code.append("    free(L1_PARAMETERS);")
    read_code += [f"    infile.read((char*)L1_PARAMETERS, {layer1_size});",
                  "    if (!infile.good()) return false;", "  }"]
    # ---- cache-fix spline: one more layer of (key, offset) pairs, always an Array (codegen.rs:487-496) ----
    cache_fix = getattr(rmi, "cache_fix", None)
    if cache_fix is not None:
        line_size, spline = int(cache_fix[0]), np.ascontiguousarray(cache_fix[1], dtype="<u8").reshape(-1, 2)
        f2 = f"{namespace}_L2_PARAMETERS"
        with open(os.path.join(data_dir, f2), "wb") as f:
            f.write(spline.tobytes())
        paths["L2_PARAMETERS"] = os.path.join(data_dir, f2)
        nbytes = spline.size * 8
        cf_malloc = nbytes >= 4 * 1024
        data_h.append("uint64_t* L2_PARAMETERS;" if cf_malloc else f"uint64_t L2_PARAMETERS[{spline.size}];")
        read_code += ["  {", f"    std::ifstream infile(std::filesystem::path(dataPath) / \"{f2}\", std::ios::in | std::ios::binary);",
                      "    if (!infile.good()) return false;"]
        if cf_malloc:
            read_code += [f"    L2_PARAMETERS = (uint64_t*) malloc({nbytes});", "    if (L2_PARAMETERS == NULL) return false;"]
            free_code.append("    free(L2_PARAMETERS);")
        read_code += [f"    infile.read((char*)L2_PARAMETERS, {nbytes});", "    if (!infile.good()) return false;", "  }"]
    read_code += ["  return true;", "}"]
    free_code.append("}")
    data_h.append("} // namespace")

    # ---- code ----
    report_errors = include_errors
    lookup_name = "lookup" if cache_fix is None else "_rmi_lookup_pre_cachefix"          # codegen.rs:621-625
    sig = f"uint64_t {lookup_name}({key_type} key, size_t* err)" if report_errors else f"uint64_t {lookup_name}({key_type} key)"
    code = [f'#include "{namespace}.h"', f'#include "{namespace}_data.h"', "#include <math.h>", "#include <cmath>",
            "#include <fstream>", "#include <filesystem>", "#include <iostream>"]
    if cache_fix is not None:
        code.append("#include <algorithm>")
    code.append(f"namespace {namespace} {{")
    code += read_code + free_code
    fns = [_STD_CODE[f] for f in _STD_NEEDED.get(root.kind, ())]
    for k in (root.kind, leaf_kind):
        c = _radix_table_code(int(root.ip[0]), int(root.ip[1])) if k in RADIX_TABLES else _MODEL_CODE[_fn_name(k, root)]
        if c not in fns:
            fns.append(c)
    code += fns
    code.append("""
inline size_t FCLAMP(double inp, double bound) {
  if (inp < 0.0) return 0;
  return (inp > bound ? bound : (size_t)inp);
}
""")
    code.append(f"{sig} {{")
    code.append("  size_t modelIndex;")
    if _output_is_float(root.kind) or _output_is_float(leaf_kind):
        code.append("  double fpred;")
    if not _output_is_float(root.kind):
        code.append("  uint64_t ipred;")
    root_in = "double" if _output_is_float(root.kind) else "uint64_t"
    root_var = "fpred" if _output_is_float(root.kind) else "ipred"
    args = root_table_args or ", ".join(f"L0_PARAMETER{i}" for i in range(len(root_vals)))
    code.append(f"  {root_var} = {_fn_name(root.kind, root)}({args}, ({root_in})key);")
    # model_index_from_output! (codegen.rs:346-373)
    if _output_is_float(root.kind):
        mi = f"FCLAMP(fpred, {L}.0 - 1.0)" if _needs_bounds_check(root.kind) else "(uint64_t) fpred"
    else:
        mi = f"(ipred > {L} - 1 ? {L} - 1 : ipred)" if _needs_bounds_check(root.kind) else "ipred"
    code.append(f"  modelIndex = {mi};")
    largs = ", ".join(acc(i) for i in range(ppl))
    code.append(f"  fpred = {_fn_name(leaf_kind)}({largs}, (double)key);")
    code.append(err_line)
    code.append(f"  return FCLAMP(fpred, {n}.0 - 1.0);")       # always bounds-checked (codegen.rs:713-717)
    code.append("}")
    if cache_fix is not None:                                   # generate_cache_fix_code, codegen.rs:396-447
        code.append(f"""
struct __attribute__((packed)) SplinePoint {{
  uint64_t key;
  uint64_t value;
}};

uint64_t lookup(uint64_t key, size_t* err) {{
  const uint64_t num_spline_pts = {len(spline)};
  const uint64_t total_keys = {int(rmi.num_data_rows)};
  size_t error_on_spline_search;

  struct SplinePoint* begin = (struct SplinePoint*) L2_PARAMETERS;

  *err = {line_size};
  uint64_t start = _rmi_lookup_pre_cachefix(key, &error_on_spline_search);

  size_t upper = (start + error_on_spline_search > num_spline_pts
                  ? num_spline_pts : start + error_on_spline_search);
  size_t lower = (error_on_spline_search > start
                  ? 0 : start - error_on_spline_search);

  struct SplinePoint* res = std::lower_bound(begin + lower,
                                             begin + upper,
                                             key,
                                             [](const auto& lhs, const auto rhs) {{ return lhs.key < rhs; }});

  if (res == begin + num_spline_pts)
    // we've searched for something past the last point
    return total_keys - 1;

  auto pt1 = *(res - 1);
  auto pt2 = *res;

  auto v0 = (double)pt1.value;
  auto v1 = (double)pt2.value;
  auto t = ((double)(key - pt1.key)) / (double)(pt2.key - pt1.key);
  return (((uint64_t) std::fma(1.0 - t, v0, t * v1)) / {line_size}) * {line_size};
}}""")
    code.append("} // namespace")

    bt = int(getattr(rmi, "build_time", 0) if build_time_ns is None else build_time_ns)
    header = ["#include <cstddef>", "#include <cstdint>", f"namespace {namespace} {{",
              "bool load(char const* dataPath);", "void cleanup();",
              f"const size_t RMI_SIZE = {rmi_size(root.kind, leaf_kind, L, include_errors, 0 if root.table is None else len(root.table), 0 if cache_fix is None else len(spline))};",
              f"const uint64_t BUILD_TIME_NS = {bt};", f'const char NAME[] = "{namespace}";',
              (f"{sig};" if cache_fix is None else "uint64_t lookup(uint64_t key, size_t* err);"), "}"]

    for name, lines in ((f"{namespace}.cpp", code), (f"{namespace}_data.h", data_h), (f"{namespace}.h", header)):
        pth = os.path.join(out_dir, name)
        with open(pth, "w") as f:
            f.write("\n".join(lines) + "\n")
        paths[name] = pth
    return paths
