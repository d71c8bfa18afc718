// rmi_device.hip.h -- device-side helpers shared by the gfx950 kernels.
//
// Arithmetic rules: IEEE f64, no contraction (built with
// -ffp-contract=off) except the explicit fma() that mirrors f64::mul_add in the reference
// (linear.rs:89, linear_spline.rs:52, cubic_spline.rs:146-148); `f64 as u64` saturates
// (models/mod.rs:736); `u64 as f64` is round-to-nearest-even.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rmi {

enum : int { K_LINEAR = 0, K_LINEAR_SPLINE = 1, K_CUBIC = 2, K_RADIX = 3, K_ROBUST_LINEAR = 4,
             K_RADIX_TABLE = 5,      // RadixTable (radix8/18/22/26/28): roots only
             K_LOGLINEAR = 6,        // exp1(fma(beta, x, alpha)), linear.rs:176-180: roots only
             K_NORMAL = 7 };         // phi((x - mean) / stdev) * scale, normal.rs:80-84: roots only

// error bits raised by kernels (host maps them to rmi_hip_error codes)
enum : uint32_t {
  EF_NON_MONOTONE = 1u << 0,
  EF_DEGENERATE_SPLIT = 1u << 1,
  EF_ROOT_OOB = 1u << 2,
  EF_NEG_VARIANCE = 1u << 3,
  EF_ROBUST_TOO_SMALL = 1u << 5,   // linear.rs:248: assert!(bnd*2+1 < data.len()) on a leaf container
  EF_CUBIC_DEGENERATE = 1u << 4,   // cubic_spline.rs:46-65: `.unwrap()` on an empty search (distinct keys, one f64)
  EF_LIST_OVERFLOW = 1u << 6,      // internal: a hand-over list (leaves for the list kernels, their stretches) was full
  EF_PEER_TIMEOUT = 1u << 7,       // direct exchange: a peer's flag of this epoch did not arrive in time
};

// Root model parameters + branching factor, passed by value to kernels.
struct RootP {
  double p0, p1, p2, p3;   // linear-like, loglinear: (alpha, beta); cubic: (a, b, c, d); normal: (mean, stdev, scale)
  uint32_t prefix, bits;   // radix: (prefix, bits); radix table: (prefix, shift = 64 - prefix - table_bits or 0)
  uint64_t L;              // number of leaves
  const uint32_t* table;   // radix table: hint_table in HBM (2^table_bits entries), else null
  // radix family: largest leaf id a key can get, and the raw prediction above which the root is
  // out of bounds (two_layer.rs:45-48).  radix: (L-1, L-1).  bradix (balanced_radix.rs:104-116) is
  // the radix function with its own clamp: (min(clamp, L-1), never).
  uint64_t cap, oob_cap;
};

// largest leaf id / out-of-bounds threshold of the raw root prediction
template <int ROOT> __device__ __forceinline__ uint64_t root_cap(const RootP& r) {
  if constexpr (ROOT == K_RADIX) return r.cap; else return r.L - 1;
}
template <int ROOT> __device__ __forceinline__ uint64_t root_oob_above(const RootP& r) {
  if constexpr (ROOT == K_RADIX) return r.oob_cap; else return r.L - 1;
}

// radix.rs:125-131 (release-mode masked shifts)
__device__ __forceinline__ uint64_t radix_table_slot(const RootP& r, uint64_t v) {
  return ((v << (r.prefix & 63u)) >> (r.prefix & 63u)) >> (r.bits & 63u);
}

// Index space of one launch.  All key indices and leaf ids inside the kernels are GLOBAL; the key
// and per-leaf array pointers handed to the kernels are pre-offset so that ptr[global_index] is
// the right element.  One GPU: it = rd = [0, n), leaves [0, L).  A shard of a multi-GPU run owns
// the leaves [leaf_lo, leaf_hi) and the keys [it_lo, it_hi) that map to them, and can read a
// small halo around them.
struct Span {
  uint64_t it_lo, it_hi;      // keys this launch is responsible for
  uint64_t rd_lo, rd_hi;      // keys that may be read (rd_lo <= it_lo, it_hi <= rd_hi)
  uint64_t n;                 // number of keys of the whole data set
  uint64_t leaf_lo, leaf_hi;  // leaves this launch writes
};

// Small device-resident state shared between the kernels of one train call.
struct DevState {
  unsigned long long split_idx;     // two_layer.rs:132-136 ; == n when no key reaches L/2
  unsigned long long split_target;  // two_layer.rs:152-156
  unsigned long long last_target;   // leaf of key[n-1] (owner of the Q7 extra count)
  unsigned int err_flags;
  unsigned int regs_listed;         // pipeline 4: groups k_leaf_regs put on the list (the host stops taking it for key sets where that is most of them)
  // aggregate statistics (two_layer.rs:267-287), filled by the stats kernel
  unsigned long long max_err;
  unsigned long long max_err_idx;
  unsigned long long sum_n_err;     // sum(n*err) wrapping u64
  double sum_l2;                    // sum((n*err)^2 / N)
  double sum_log2;                  // sum(n * log2(2 err + 2))
  // leaves k_leaf_lanes handed to the list kernels while the record was published WITHOUT them (k_lane_reduce): the host
  // runs them behind its synchronisation (listed_epilogue); part of the record the ranks of a sharded training exchange
  unsigned long long pending;
  // leaves too long for the lockstep pass (see k_fit_long): number of entries in the long list
  unsigned long long long_count;
  unsigned long long long_cap;
  // one-pass mode (rmi_sigma.hip.h): leaves handed to the exact kernels, and how many of them by the guard
  unsigned long long flag_count;
  unsigned long long flag_cap;
  unsigned long long guard_count;
  unsigned long long merged_count;  // long leaves fitted from merged partial sums (one-pass mode 2)
  unsigned long long seg_count;     // stretches of SG_SEG keys the error pass of the long listed leaves is cut into (k_list_tail)
  unsigned long long seg_cap;
  unsigned long long giant_count;   // listed leaves whose exact fit is left to the host (GiantLeaf entries), see rmi_hip.hip
  unsigned long long giant_cap;
  // k_leaf_search: leaves (of every 16th block) whose boundary probes (pairs of neighbouring keys) met two equal keys.  Pipeline 4 takes a group of 64 leaves
  // only if none of its ~12 000 keys repeats: when the probes -- some twenty keys a leaf -- see a repeat at more than one leaf per group,
  // k_leaf_regs lists every group at once, without walking any (duplicate-heavy keys on the FIRST training of a key set)
  unsigned long long regs_dups;
  // k_spline_scan: tiles the short form's kernel left to the general form's (the host sizes that kernel's launch by the last count)
  unsigned long long scan_listed;
};

// A leaf whose container is so long that its sequential recurrence is faster on a host core (~4 ns per point against
// ~28 on a wave): k_list records it here instead of fitting it.
struct GiantLeaf { unsigned long long j, lo, hi, y0; };    // leaf, container [lo, hi], FixDups offset of its first point

template <typename K> struct KeyTraits;
template <> struct KeyTraits<uint64_t> {
  // `key as f64` (mod.rs:83): round-to-nearest-even of the integer.  hi * 2^32 + lo is the integer
  // exactly and the FMA rounds it once, so this is the same value as the compiler's
  // cvt/ldexp/cvt/add sequence in one operation fewer (it is on the per-key path of every pass).
  static __device__ __forceinline__ double as_float(uint64_t k) {
    return __builtin_fma((double)(uint32_t)(k >> 32), 4294967296.0, (double)(uint32_t)k);
  }
  static __device__ __forceinline__ uint64_t as_uint(uint64_t k) { return k; }
  static __device__ __forceinline__ uint64_t minus_eps(uint64_t k) { return k - 1ull; }  // mod.rs:78
  static __device__ __forceinline__ uint64_t plus_eps(uint64_t k) { return k + 1ull; }   // mod.rs:80
  static __device__ __forceinline__ uint64_t max_value() { return ~0ull; }
  static __device__ __forceinline__ uint64_t zero_value() { return 0ull; }
};
template <> struct KeyTraits<uint32_t> {
  static __device__ __forceinline__ double as_float(uint32_t k) { return (double)k; }   // mod.rs:95
  static __device__ __forceinline__ uint64_t as_uint(uint32_t k) { return (uint64_t)k; } // mod.rs:474-478
  static __device__ __forceinline__ uint32_t minus_eps(uint32_t k) { return k - 1u; }
  static __device__ __forceinline__ uint32_t plus_eps(uint32_t k) { return k + 1u; }
  static __device__ __forceinline__ uint32_t max_value() { return 0xFFFFFFFFu; }
  static __device__ __forceinline__ uint32_t zero_value() { return 0u; }
};

// Rust `f64 as u64` (saturating, NaN -> 0)
__device__ __forceinline__ uint64_t sat_f64_to_u64(double v) {
  if (!(v > 0.0)) return 0ull;
  if (v >= 18446744073709551616.0) return ~0ull;
  return (uint64_t)v;
}
template <> struct KeyTraits<double> {
  static __device__ __forceinline__ double as_float(double k) { return k; }              // mod.rs:107
  static __device__ __forceinline__ uint64_t as_uint(double k) { return sat_f64_to_u64(k); } // mod.rs:108
  static __device__ __forceinline__ double minus_eps(double k) { return k - 2.220446049250313e-16; }
  static __device__ __forceinline__ double plus_eps(double k) { return k + 2.220446049250313e-16; }
  static __device__ __forceinline__ double max_value() { return 1.7976931348623157e308; }
  static __device__ __forceinline__ double zero_value() { return 0.0; }
};

// Model::predict_to_int default (models/mod.rs:735-737) applied to a float prediction
__device__ __forceinline__ uint64_t float_pred_to_int(double f) {
  return sat_f64_to_u64(fmax(0.0, floor(f)));
}

// exp1 / phi of the reference's standard functions (normal.rs:12-27, linear.rs:156-166,
// stdlib.rs:29-45): plain IEEE operations, so the device value is the host value.
__device__ __forceinline__ double exp1_ref(double x) {
  x = 1.0 + x / 64.0;
  x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
  return x;
}
__device__ __forceinline__ double phi_ref(double x) { return 1.0 / (1.0 + exp1_ref(-1.65451 * x)); }

// predict_to_float of a float root at x = key as f64
template <int ROOT>
__device__ __forceinline__ double root_eval_f(const RootP& r, double x) {
  if constexpr (ROOT == K_CUBIC) {
    return __builtin_fma(__builtin_fma(__builtin_fma(r.p0, x, r.p1), x, r.p2), x, r.p3);   // cubic_spline.rs:146-148
  } else if constexpr (ROOT == K_LOGLINEAR) {
    return exp1_ref(__builtin_fma(r.p1, x, r.p0));           // linear.rs:177-180
  } else if constexpr (ROOT == K_NORMAL) {
    return phi_ref((x - r.p0) / r.p1) * r.p2;                // normal.rs:81-84
  } else {
    return __builtin_fma(r.p1, x, r.p0);                     // linear.rs:87-90
  }
}

// Root prediction, NOT yet clamped to L-1.
template <int ROOT, typename K>
__device__ __forceinline__ uint64_t root_predict(const RootP& r, K k) {
  if constexpr (ROOT == K_RADIX) {
    // radix.rs:43-50 (release-mode masked shifts)
    uint64_t v = KeyTraits<K>::as_uint(k);
    return (v << (r.prefix & 63u)) >> ((64u - r.bits) & 63u);
  } else if constexpr (ROOT == K_RADIX_TABLE) {
    return (uint64_t)r.table[radix_table_slot(r, KeyTraits<K>::as_uint(k))];   // radix.rs:124-134
  } else {
    return float_pred_to_int(root_eval_f<ROOT>(r, KeyTraits<K>::as_float(k)));
  }
}

template <int ROOT>
__device__ __forceinline__ constexpr bool root_needs_bounds_check() {
  return !(ROOT == K_CUBIC || ROOT == K_RADIX || ROOT == K_RADIX_TABLE);   // cubic_spline.rs:184-186, radix.rs:72-74, :160-162
}

// Leaf prediction (predict_to_int) from a row of parameters.
template <int LEAF, typename K>
__device__ __forceinline__ uint64_t leaf_predict(const double* __restrict__ p, K k) {
  double x = KeyTraits<K>::as_float(k);
  if constexpr (LEAF == K_CUBIC) {
    double v1 = __builtin_fma(p[0], x, p[1]);
    double v2 = __builtin_fma(v1, x, p[2]);
    double v3 = __builtin_fma(v2, x, p[3]);
    return float_pred_to_int(v3);
  } else {
    return float_pred_to_int(__builtin_fma(p[1], x, p[0]));
  }
}

// error_between: two_layer.rs:14-18
__device__ __forceinline__ uint64_t error_between(uint64_t v1, uint64_t v2, uint64_t max_pred) {
  uint64_t p1 = v1 < max_pred ? v1 : max_pred;
  uint64_t p2 = v2 < max_pred ? v2 : max_pred;
  return p1 > p2 ? p1 - p2 : p2 - p1;
}

// Offset of the first occurrence of keys[i] (FixDupsIter semantics, models/mod.rs:154-185):
// lower_bound of keys[i] in [0, i].  O(1) when keys[i-1] != keys[i].
template <typename K>
__device__ __forceinline__ uint64_t first_occurrence(const K* __restrict__ keys, uint64_t i, uint64_t rd_lo = 0) {
  K v = keys[i];
  if (i <= rd_lo || keys[i - 1] != v) return i;
  // keys[hi] == v.  Gallop down first: runs of equal keys are short as a rule, and every step of a plain bisection of
  // [rd_lo, i) is a dependent load from the key array (28 of them for 200 M keys, ~2 us each, with a whole wave waiting)
  uint64_t lo = rd_lo, hi = i - 1;
  for (uint64_t step = 1; hi - rd_lo > step; step <<= 1) {
    const uint64_t q = hi - step;
    if (keys[q] != v) { lo = q + 1; break; }
    hi = q;
  }
  while (lo < hi) {
    uint64_t mid = lo + ((hi - lo) >> 1);
    if (keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------------
// Root target in the f64 domain: min(L-1, predict_to_int(key)) as a double.  Exact because every
// intermediate is an integer < 2^53 (L <= 2^31 is enforced by the host): identical buckets to
// the integer formulation, without the 64-bit float->int conversions.
// ---------------------------------------------------------------------------------------------
template <int ROOT, typename K>
__device__ __forceinline__ double root_target_f(const RootP& r, double Lm1f, K k, bool& oob) {
  if constexpr (ROOT == K_RADIX) {
    uint64_t v = KeyTraits<K>::as_uint(k);
    uint64_t p = (v << (r.prefix & 63u)) >> ((64u - r.bits) & 63u);
    oob = p > r.oob_cap;
    p = p < r.cap ? p : r.cap;
    return (double)p;
  } else if constexpr (ROOT == K_RADIX_TABLE) {
    uint64_t p = (uint64_t)r.table[radix_table_slot(r, KeyTraits<K>::as_uint(k))];
    oob = p > (uint64_t)Lm1f;
    p = p < r.L - 1 ? p : r.L - 1;
    return (double)p;
  } else {
    double f = root_eval_f<ROOT>(r, KeyTraits<K>::as_float(k));
    f = fmax(0.0, floor(f));          // f64::max(0.0, NaN) == 0.0, as in models/mod.rs:736
    oob = f > Lm1f;
    return fmin(f, Lm1f);
  }
}


}  // namespace rmi
