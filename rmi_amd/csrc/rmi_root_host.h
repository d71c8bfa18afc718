// rmi_root_host.h -- host-side root-model fits in reference order.
//
// The root of a two-layer RMI is `train_model(layer1, data)` with scale = L/N
// (two_layer.rs:109-110).  For `linear` / `robust_linear` this is a *sequential* streaming SLR
// recurrence over N+1 points (linear.rs:12-59): floating-point non-associativity means a
// parallel formulation changes low bits of (alpha, beta) and therefore a handful of bucket
// assignments.  Bit-identical buckets need the identical recurrence, so
// the exact root fit runs here on the host, once per (data, root, L); its result is an *input*
// of the device hot path.  `radix`, `linear_spline` and the `cubic` coefficients are O(1).
//
// This is product code, independent of oracle/ (which restates the same reference lines for
// the tests).  Build with -ffp-contract=off.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rmi_hip.h"

namespace rmi_host {

inline uint64_t sat_u64(double v) {           // Rust `f64 as u64`
  if (!(v > 0.0)) return 0;
  if (v >= 18446744073709551616.0) return UINT64_MAX;
  return (uint64_t)v;
}

template <typename K> inline double as_float(K k) { return (double)k; }
template <typename K> inline uint64_t as_uint(K k) { return (uint64_t)k; }
template <> inline uint64_t as_uint<double>(double k) { return sat_u64(k); }

// View of RMITrainingData<T> over a raw slice with `scale` (models/mod.rs:233-317).
template <typename K>
struct Data {
  const K* keys;
  uint64_t n;
  double scale;
  inline uint64_t scale_y(uint64_t y) const {        // map_scale!: mod.rs:238-250
    if (std::fabs(scale - 1.0) > DBL_EPSILON) return sat_u64((double)y * scale);
    return y;
  }
};

// Calls f(key, scaled_y) for every item of data.iter() -- FixDupsIter (mod.rs:154-185): y is the
// offset of the first occurrence of the key, and the last item is yielded twice (Q1).
// Items [skip, skip+take) of that sequence are visited.
template <typename K, typename F>
inline void for_each_fixdups(const Data<K>& d, uint64_t skip, uint64_t take, F&& f) {
  if (d.n == 0) return;
  uint64_t first = 0;
  uint64_t emitted = 0, visited = 0;
  const uint64_t total = d.n + 1;
  for (uint64_t i = 0; i < total && visited < take; i++) {
    const uint64_t src = i < d.n ? i : d.n - 1;      // tail duplicate
    if (i < d.n && (i == 0 || !(d.keys[i] == d.keys[i - 1]))) first = i;
    if (emitted++ < skip) continue;
    f(d.keys[src], d.scale_y(first));
    visited++;
  }
}

// common_prefix_size (utils.rs:23-36) is an OR / AND fold over all keys: the number of leading bits
// on which every key agrees.  The keys are sorted (as unsigned integers; `as_uint` of an f64 key is
// monotone), so every key lies between the first and the last one and shares their common leading
// bits, and the first bit on which those two differ is a bit on which not all keys agree: the fold
// equals the number of leading zeros of first XOR last.  O(1) instead of a pass over the data.
template <typename K>
inline int common_prefix_sorted(const Data<K>& d) {
  if (d.n == 0) return 64;
  const uint64_t x = as_uint(d.keys[0]) ^ as_uint(d.keys[d.n - 1]);
  return x == 0 ? 64 : __builtin_clzll(x);
}

struct Slr {                                          // linear.rs:12-59
  double mean_x = 0.0, mean_y = 0.0, c = 0.0, m2 = 0.0;
  uint64_t n = 0;
  inline void push(double x, double y) {
    n += 1;
    const double dx = x - mean_x;
    mean_x += dx / (double)n;
    mean_y += (y - mean_y) / (double)n;
    c += dx * (y - mean_y);
    const double dx2 = x - mean_x;
    m2 += dx * dx2;
  }
  inline int finish(double* alpha, double* beta) const {
    if (n == 0) { *alpha = 0.0; *beta = 0.0; return RMI_OK; }
    if (n == 1) { *alpha = mean_y; *beta = 0.0; return RMI_OK; }
    const double cov = c / (double)(n - 1);
    const double var = m2 / (double)(n - 1);
    if (!(var >= 0.0)) return RMI_ERR_NEGATIVE_VARIANCE;
    if (var == 0.0) { *alpha = mean_y; *beta = 0.0; return RMI_OK; }
    const double b = cov / var;
    *alpha = mean_y - b * mean_x;
    *beta = b;
    return RMI_OK;
  }
};

// The same step with the two quotients by the count n formed as fma(a, r, a*rl), r = RN(1/n) and
// rl = RN((1 - n r) r) its tail: the correctly rounded a/n for integer n < 2^40 and the operands
// integer keys produce (argument at div_by_count2 in rmi_stream.hip.h; checked against `/` by
// rmi_hip_selftest_host_div), i.e. the same bits as push() -- but the one real division, 1/n, does
// not depend on the running means, so it leaves the loop-carried chain: sub, mul, fma, add instead
// of sub, div, add.  ~25 % less time per key on the sequential root fit.  Needs hardware FMA
// (checked at run time); f64 keys keep the plain form.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define RMI_HOST_FMA __attribute__((target("fma")))
inline bool host_has_fma() { static const bool v = __builtin_cpu_supports("fma"); return v; }
#else
#define RMI_HOST_FMA
inline bool host_has_fma() { return false; }
#endif

RMI_HOST_FMA inline void slr_push_recip(Slr& s, double x, double y) {
  s.n += 1;
  const double nf = (double)s.n;
  const double r = 1.0 / nf;
  const double rl = __builtin_fma(-nf, r, 1.0) * r;
  const double dx = x - s.mean_x;
  s.mean_x += __builtin_fma(dx, r, dx * rl);
  const double dy = y - s.mean_y;
  s.mean_y += __builtin_fma(dy, r, dy * rl);
  s.c += dx * (y - s.mean_y);
  const double dx2 = x - s.mean_x;
  s.m2 += dx * dx2;
}

// slr over items [skip, skip+take) of data.iter() (FixDups offsets, Q1 tail duplicate), reciprocal form
template <typename K>
RMI_HOST_FMA inline void slr_run_recip(const Data<K>& d, uint64_t skip, uint64_t take, Slr& s) {
  if (d.n == 0) return;
  uint64_t first = 0, emitted = 0, visited = 0;
  const uint64_t total = d.n + 1;
  for (uint64_t i = 0; i < total && visited < take; i++) {
    const uint64_t src = i < d.n ? i : d.n - 1;      // tail duplicate
    if (i < d.n && (i == 0 || !(d.keys[i] == d.keys[i - 1]))) first = i;
    if (emitted++ < skip) continue;
    slr_push_recip(s, as_float(d.keys[src]), (double)d.scale_y(first));
    visited++;
  }
}
template <typename K> constexpr bool integer_keys() { return true; }
template <> constexpr bool integer_keys<double>() { return false; }

// slr over one leaf's container given as a host copy ck[0 .. npts) of the keys [lo, lo + npts) (linear.rs:12-59 over
// C_j.iter(), two_layer.rs:52-90): y is the FixDups offset -- y0 for the first point (its first occurrence may lie in
// front of the container), the index of a run's first key afterwards -- and the last item counts twice (Q1).
template <typename K, bool RECIP>
RMI_HOST_FMA inline int leaf_slr_impl(const K* ck, uint64_t npts, uint64_t lo, uint64_t y0, double* alpha, double* beta) {
  Slr s;
  double y = (double)y0, x = 0.0;
  for (uint64_t i = 0; i < npts; i++) {
    if (i > 0 && !(ck[i] == ck[i - 1])) y = (double)(lo + i);
    x = as_float(ck[i]);
    if (RECIP) slr_push_recip(s, x, y); else s.push(x, y);
  }
  if (npts > 0) { if (RECIP) slr_push_recip(s, x, y); else s.push(x, y); }      // models/mod.rs:180
  return s.finish(alpha, beta);
}
template <typename K>
inline int leaf_slr(const K* ck, uint64_t npts, uint64_t lo, uint64_t y0, double* alpha, double* beta) {
  if (integer_keys<K>() && host_has_fma() && npts < (1ull << 40)) return leaf_slr_impl<K, true>(ck, npts, lo, y0, alpha, beta);
  return leaf_slr_impl<K, false>(ck, npts, lo, y0, alpha, beta);
}

template <typename K>
inline void slr_run(const Data<K>& d, uint64_t skip, uint64_t take, Slr& s) {
  if (integer_keys<K>() && host_has_fma() && d.n < (1ull << 40)) slr_run_recip(d, skip, take, s);
  else for_each_fixdups(d, skip, take, [&](K k, uint64_t y) { s.push(as_float(k), (double)y); });
}

template <typename K>
inline int fit_linear(const Data<K>& d, rmi_hip_model_params* m) {        // linear.rs:79-83
  Slr s;
  slr_run(d, 0, UINT64_MAX, s);
  return s.finish(&m->p[0], &m->p[1]);
}

// loglinear_slr (linear.rs:60-72): slr over (x, ln(y)) of the items whose ln(y) is finite (y >= 1).
// `ln` is the platform libm's, as in the reference.
template <typename K>
inline int fit_loglinear(const Data<K>& d, rmi_hip_model_params* m) {
  Slr s;
  for_each_fixdups(d, 0, UINT64_MAX, [&](K k, uint64_t y) {
    const double ly = std::log((double)y);
    if (std::isfinite(ly)) s.push(as_float(k), ly);
  });
  return s.finish(&m->p[0], &m->p[1]);
}

// ncdf (normal.rs:29-50): params (mean, stdev, scale); sequential sums over the N+1 items of iter().
template <typename K>
inline int fit_normal(const Data<K>& d, rmi_hip_model_params* m) {
  double scale = -INFINITY, mean = 0.0, stdev = 0.0;
  const double n = (double)d.n;
  for_each_fixdups(d, 0, UINT64_MAX, [&](K k, uint64_t y) {
    mean += as_float(k) / n;
    scale = std::fmax(scale, (double)y);
  });
  for_each_fixdups(d, 0, UINT64_MAX, [&](K k, uint64_t) {
    const double dx = as_float(k) - mean;
    stdev += dx * dx;                                                        // powf(2.0)
  });
  stdev /= n;
  stdev = std::sqrt(stdev);
  m->p[0] = mean; m->p[1] = stdev; m->p[2] = scale;
  return RMI_OK;
}

// exp1 / phi: normal.rs:12-27, linear.rs:156-166
inline double exp1_ref(double x) {
  x = 1.0 + x / 64.0;
  x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
  return x;
}
inline double phi_ref(double x) { return 1.0 / (1.0 + exp1_ref(-1.65451 * x)); }

template <typename K>
inline int fit_robust_linear(const Data<K>& d, rmi_hip_model_params* m) { // linear.rs:239-260
  if (d.n == 0) { m->p[0] = 0.0; m->p[1] = 0.0; return RMI_OK; }
  uint64_t bnd = sat_u64((double)d.n * 0.0001);
  if (bnd < 1) bnd = 1;
  if (!(bnd * 2 + 1 < d.n)) return RMI_ERR_ROBUST_TOO_SMALL;
  Slr s;
  slr_run(d, bnd, d.n - 2 * bnd, s);
  return s.finish(&m->p[0], &m->p[1]);
}

template <typename K>
inline void linear_splines(const Data<K>& d, double* alpha, double* beta) { // linear_spline.rs:13-35
  if (d.n == 0) { *alpha = 0.0; *beta = 0.0; return; }
  const double y0 = (double)d.scale_y(0);
  if (d.n == 1) { *alpha = y0; *beta = 0.0; return; }
  const K k0 = d.keys[0], k1 = d.keys[d.n - 1];
  if (k0 == k1) { *alpha = y0; *beta = 0.0; return; }
  const double y1 = (double)d.scale_y(d.n - 1);       // get(): raw index, no FixDups
  const double slope = (y0 - y1) / (as_float(k0) - as_float(k1));
  *alpha = y0 - slope * as_float(k0);
  *beta = slope;
}

inline double cubic_eval(const double p[4], double x) {                     // cubic_spline.rs:140-151
  return std::fma(std::fma(std::fma(p[0], x, p[1]), x, p[2]), x, p[3]);
}

// cubic() (cubic_spline.rs:18-101) through a key accessor.  The two searches of the reference --
// the first item of iter() whose scaled x is > 0 and the last index whose scaled x is < 1 -- are
// monotone predicates over sorted keys, so they are binary searches here (the same predicate, the
// same answer, O(log n) keys instead of a scan that is long exactly on degenerate data).
template <typename K, typename Get>
inline int cubic_coeffs_get(Get get, const Data<K>& d, double out[4]) {
  if (d.n == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; out[3] = 0.0; return RMI_OK; }
  const double y_first = (double)d.scale_y(0);
  if (d.n == 1) { out[0] = out[1] = out[2] = 0.0; out[3] = y_first; return RMI_OK; }
  const K k0 = get(0), kl = get(d.n - 1);
  if (k0 == kl) { out[0] = out[1] = out[2] = 0.0; out[3] = y_first; return RMI_OK; }  // sorted: all equal
  const double xmin = as_float(k0), ymin = y_first;
  const double xmax = as_float(kl), ymax = (double)d.scale_y(d.n - 1);
  auto sc = [](double v, double mn, double mx) { return (v - mn) / (mx - mn); };
  double m1;
  {  // :46-54.  The found key differs from its predecessor, so its FixDups offset is its own index.
    uint64_t lo = 0, hi = d.n;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (sc(as_float(get(mid)), xmin, xmax) > 0.0) hi = mid; else lo = mid + 1; }
    if (lo >= d.n) return RMI_ERR_CUBIC_DEGENERATE;                           // .unwrap() on None
    const double sxn = sc(as_float(get(lo)), xmin, xmax), syn = sc((double)d.scale_y(lo), ymin, ymax);
    m1 = (syn - 0.0) / (sxn - 0.0);
  }
  double m2;
  {  // :56-65 (get(): raw index)
    uint64_t lo = 0, hi = d.n;                                                // first index whose scaled x is NOT < 1
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (sc(as_float(get(mid)), xmin, xmax) < 1.0) lo = mid + 1; else hi = mid; }
    if (lo == 0) return RMI_ERR_CUBIC_DEGENERATE;
    const uint64_t ip = lo - 1;
    const double sxp = sc(as_float(get(ip)), xmin, xmax), syp = sc((double)d.scale_y(ip), ymin, ymax);
    m2 = (1.0 - syp) / (1.0 - sxp);
  }
  if (m1 * m1 + m2 * m2 > 9.0) {
    const double tau = 3.0 / std::sqrt(m1 * m1 + m2 * m2);
    m1 *= tau; m2 *= tau;
  }
  const double den = std::pow(xmax - xmin, 3.0);
  double a = (m1 + m2 - 2.0) / den;
  double b = -(xmax * (2.0 * m1 + m2 - 3.0) + xmin * (m1 + 2.0 * m2 - 3.0)) / den;
  double c = (m1 * (xmax * xmax) + m2 * (xmin * xmin) + xmax * xmin * (2.0 * m1 + 2.0 * m2 - 6.0)) / den;
  double dd = -xmin * (m1 * (xmax * xmax) + xmax * xmin * (m2 - 3.0) + (xmin * xmin)) / den;
  a *= ymax - ymin; b *= ymax - ymin; c *= ymax - ymin; dd *= ymax - ymin; dd += ymin;
  out[0] = a; out[1] = b; out[2] = c; out[3] = dd;
  return RMI_OK;
}
template <typename K>
inline int cubic_coeffs(const Data<K>& d, double out[4]) {
  return cubic_coeffs_get<K>([&](uint64_t i) { return d.keys[i]; }, d, out);
}

template <typename K>
inline int fit_cubic(const Data<K>& d, rmi_hip_model_params* m) {          // cubic_spline.rs:108-136
  double cp[4];
  int rc = cubic_coeffs(d, cp);
  if (rc) return rc;
  double la, lb; linear_splines(d, &la, &lb);
  double our_error = 0.0, lin_error = 0.0;
  for_each_fixdups(d, 0, UINT64_MAX, [&](K k, uint64_t y) {
    const double x = as_float(k);
    our_error += std::fabs(cubic_eval(cp, x) - (double)y);
    lin_error += std::fabs(std::fma(lb, x, la) - (double)y);
  });
  if (lin_error < our_error) { m->p[0] = 0.0; m->p[1] = 0.0; m->p[2] = lb; m->p[3] = la; }
  else std::memcpy(m->p, cp, sizeof cp);
  return RMI_OK;
}

inline int num_bits(uint64_t largest) {                                    // utils.rs:13-21
  int nbits = 0;
  while (nbits + 1 < 64 && ((1ull << (nbits + 1)) - 1) <= largest) nbits++;
  return nbits >= 1 ? nbits : -1;
}

template <typename K>
inline int fit_radix(const Data<K>& d, rmi_hip_model_params* m) {          // radix.rs:18-39
  m->ip[0] = 0; m->ip[1] = 0;
  if (d.n == 0) return RMI_OK;
  // max scaled y over iter(): y is monotone, so it is the first-occurrence offset of the last key
  uint64_t first = d.n - 1;
  while (first > 0 && d.keys[first - 1] == d.keys[d.n - 1]) first--;
  const int bits = num_bits(d.scale_y(first));
  if (bits < 0) return RMI_ERR_NUM_BITS;
  const int prefix = common_prefix_sorted(d);
  m->ip[0] = (uint64_t)(uint8_t)prefix;
  m->ip[1] = (uint64_t)(uint8_t)bits;
  return RMI_OK;
}

// The roots that need O(1) keys (`radix`: first key, last key, start of the last run; `linear_spline`:
// first and last key), fitted through an accessor -- for key sets that live in HBM only.
template <typename K, typename Get>
inline int fit_root_sparse(int kind, Get get, uint64_t n, uint64_t num_leaves, rmi_hip_model_params* m) {
  std::memset(m, 0, sizeof *m);
  m->kind = kind;
  if (n == 0) return RMI_OK;
  const Data<K> d{nullptr, n, (double)num_leaves / (double)n};
  const K k0 = get(0), kl = get(n - 1);
  if (kind == RMI_MODEL_LINEAR_SPLINE) {                                     // linear_spline.rs:13-35
    const double y0 = (double)d.scale_y(0);
    if (n == 1 || k0 == kl) { m->p[0] = y0; m->p[1] = 0.0; return RMI_OK; }
    const double y1 = (double)d.scale_y(n - 1);
    const double slope = (y0 - y1) / (as_float(k0) - as_float(kl));
    m->p[0] = y0 - slope * as_float(k0);
    m->p[1] = slope;
    return RMI_OK;
  }
  if (kind == RMI_MODEL_RADIX) {                                             // radix.rs:18-39
    uint64_t lo = 0, hi = n - 1;                                             // first occurrence of the last key
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (get(mid) < kl) lo = mid + 1; else hi = mid; }
    const int bits = num_bits(d.scale_y(lo));
    if (bits < 0) return RMI_ERR_NUM_BITS;
    const uint64_t x = as_uint(k0) ^ as_uint(kl);
    m->ip[0] = (uint64_t)(uint8_t)(x == 0 ? 64 : __builtin_clzll(x));
    m->ip[1] = (uint64_t)(uint8_t)bits;
    return RMI_OK;
  }
  return RMI_ERR_UNSUPPORTED_MODEL;
}

// BalancedRadixModel (balanced_radix.rs).  predict_to_int (:104-116) is the radix function followed
// by a clamp: high = min(res, clamp); low = res < clamp ? 0 : res - clamp.
inline uint64_t bradix_predict(uint64_t prefix, uint64_t bits, uint64_t clamp, bool high, uint64_t x) {
  const uint64_t res = (x << (prefix & 63)) >> ((64 - bits) & 63);             // release-mode masked shifts
  if (high) return res < clamp ? res : clamp;
  return res < clamp ? 0 : res - clamp;
}
// chi2 (:20-38): sum over the bins, in bin order, of (count - expected)^2 / expected, where the
// counts come from the N+1 items of iter_model_input() (Q1).  `count_of(bin)`: exact bin counts;
// the reference counts in the default integer type (i32), so the value converted is the wrapped one.
template <typename CountOf>
inline double bradix_chi2(uint64_t n, uint64_t max_bin, CountOf count_of) {
  const double expected = (double)n / (double)max_bin;
  double sum = 0.0;                                                            // Iterator::sum::<f64>() starts from 0.0
  for (uint64_t b = 0; b < max_bin; b++) {
    const double c = (double)(int32_t)(uint32_t)count_of(b);
    const double d = c - expected;
    sum += (d * d) / expected;                                                 // powf(2.0) == x * x
  }
  return sum;
}
// bradix (:40-87): candidates in the reference's order -- for test_bits in {bits, bits + 1}: high
// (clamp = max_output - 1), then low (clamp = max_output - bits_max) -- the first strictly smaller
// score wins.  bits_max = 2^(test_bits+1) - 1 always exceeds max_output (num_bits: 2^(bits+1) - 1 >
// max_output), so the low clamp wraps around (release build: no overflow checks) to 2^64 - d and
// the low function sends every key to bin 0.  `high_counts(test_bits)` returns an accessor for the
// bin counts of the high candidate; the low candidates' counts are (n + 1, 0, 0, ...).
template <typename HighCounts>
inline int bradix_choose(uint64_t n, uint64_t max_output, int prefix, HighCounts high_counts, rmi_hip_model_params* m) {
  const int bits = num_bits(max_output);
  if (bits < 0) return RMI_ERR_NUM_BITS;
  double best = INFINITY;
  bool have = false;
  const int hi_bits = bits + 2 < 64 ? bits + 2 : 64;
  for (int tb = bits; tb < hi_bits; tb++) {
    const uint64_t bits_max = (1ull << ((tb + 1) & 63)) - 1;
    {
      auto cnt = high_counts(tb);
      const double score = bradix_chi2(n, max_output, cnt);
      if (score < best) { best = score; have = true; m->ip[0] = (uint64_t)(uint8_t)prefix; m->ip[1] = (uint64_t)tb; m->ip[2] = max_output - 1; m->ip[3] = 1; }
    }
    {
      const uint64_t clamp = max_output - bits_max;                            // wraps
      if (!(clamp >> 63)) return RMI_ERR_UNSUPPORTED_MODEL;                    // (cannot happen, see above)
      const double score = bradix_chi2(n, max_output, [&](uint64_t b) { return b == 0 ? n + 1 : 0ull; });
      if (score < best) { best = score; have = true; m->ip[0] = (uint64_t)(uint8_t)prefix; m->ip[1] = (uint64_t)tb; m->ip[2] = clamp; m->ip[3] = 0; }
    }
  }
  return have ? RMI_OK : RMI_ERR_BAD_ARG;                                      // best_result.unwrap(): every score NaN
}

// RadixTable (radix.rs:83-170): table size by registry name (train/mod.rs:46-50) and the slot function.
inline int radix_table_bits(int kind) {
  switch (kind) {
    case RMI_MODEL_RADIX8: return 8;
    case RMI_MODEL_RADIX18: return 18;
    case RMI_MODEL_RADIX22: return 22;
    case RMI_MODEL_RADIX26: return 26;
    case RMI_MODEL_RADIX28: return 28;
    default: return -1;
  }
}
inline uint64_t radix_table_slot(uint64_t prefix, uint64_t bits, uint64_t x) {   // radix.rs:98-99, :125-131
  const uint64_t num_bits = (prefix + bits > 64) ? 0 : 64 - (prefix + bits);
  return ((x << (prefix & 63)) >> (prefix & 63)) >> (num_bits & 63);           // release-mode masked shifts
}
template <typename K>
inline int fit_root(int kind, const K* keys, uint64_t n, uint64_t num_leaves, rmi_hip_model_params* m) {
  std::memset(m, 0, sizeof *m);
  m->kind = kind;
  Data<K> d{keys, n, (double)num_leaves / (double)n};                       // two_layer.rs:109
  switch (kind) {
    case RMI_MODEL_LINEAR: return fit_linear(d, m);
    case RMI_MODEL_ROBUST_LINEAR: return fit_robust_linear(d, m);
    case RMI_MODEL_LINEAR_SPLINE: linear_splines(d, &m->p[0], &m->p[1]); return RMI_OK;
    case RMI_MODEL_CUBIC: return fit_cubic(d, m);
    case RMI_MODEL_RADIX: return fit_radix(d, m);
    case RMI_MODEL_LOGLINEAR: return fit_loglinear(d, m);
    case RMI_MODEL_NORMAL: return fit_normal(d, m);
    default: return RMI_ERR_UNSUPPORTED_MODEL;      // (the radix tables and bradix are fitted on the device: rmi_hip.hip)
  }
}

// Streaming form of the `linear` root fit: the same recurrence, fed with consecutive chunks of
// the global key array (for data sets that are produced / held shard by shard).
template <typename K>
struct LinearRootStream {
  Slr slr;
  double scale = 1.0;
  uint64_t n_global = 0, seen = 0, first = 0;
  K last_key{};
  bool have_last = false;
  void begin(uint64_t n, uint64_t num_leaves) { *this = LinearRootStream(); n_global = n; scale = (double)num_leaves / (double)n; }
  inline uint64_t scale_y(uint64_t y) const {
    if (std::fabs(scale - 1.0) > DBL_EPSILON) return sat_u64((double)y * scale);
    return y;
  }
  template <bool RECIP>
  RMI_HOST_FMA void push_impl(const K* keys, uint64_t count) {
    for (uint64_t q = 0; q < count; q++) {
      const K k = keys[q];
      if (!have_last || !(k == last_key)) first = seen;       // FixDups first-occurrence offset
      if (RECIP) slr_push_recip(slr, as_float(k), (double)scale_y(first));
      else slr.push(as_float(k), (double)scale_y(first));
      last_key = k; have_last = true; seen++;
    }
  }
  void push(const K* keys, uint64_t count) {
    if (integer_keys<K>() && host_has_fma() && n_global < (1ull << 40)) push_impl<true>(keys, count);
    else push_impl<false>(keys, count);
  }
  int finish(rmi_hip_model_params* m) {
    if (seen != n_global) return RMI_ERR_BAD_ARG;
    if (have_last) slr.push(as_float(last_key), (double)scale_y(first));   // Q1 tail duplicate
    return slr.finish(&m->p[0], &m->p[1]);
  }
};

// min(L-1, root.predict_to_int(key)) on the host (two_layer.rs:49; used to plan shard cuts)
template <typename K>
inline uint64_t root_target(const rmi_hip_model_params& m, K k, uint64_t L, const uint32_t* table = nullptr) {
  uint64_t p;
  switch (m.kind) {
    case RMI_MODEL_RADIX8: case RMI_MODEL_RADIX18: case RMI_MODEL_RADIX22: case RMI_MODEL_RADIX26: case RMI_MODEL_RADIX28:
      p = table[radix_table_slot(m.ip[0], m.ip[1], as_uint(k))]; break;
    case RMI_MODEL_RADIX: p = (as_uint(k) << (m.ip[0] & 63)) >> ((64 - m.ip[1]) & 63); break;
    case RMI_MODEL_BRADIX: p = bradix_predict(m.ip[0], m.ip[1], m.ip[2], m.ip[3] != 0, as_uint(k)); break;
    case RMI_MODEL_CUBIC: p = sat_u64(std::fmax(0.0, std::floor(cubic_eval(m.p, as_float(k))))); break;
    case RMI_MODEL_LOGLINEAR: p = sat_u64(std::fmax(0.0, std::floor(exp1_ref(std::fma(m.p[1], as_float(k), m.p[0]))))); break;
    case RMI_MODEL_NORMAL: p = sat_u64(std::fmax(0.0, std::floor(phi_ref((as_float(k) - m.p[0]) / m.p[1]) * m.p[2]))); break;
    case RMI_MODEL_LOGNORMAL: case RMI_MODEL_HISTOGRAM: p = 0; break;          // (not on the device path; callers reject them first)
    default: p = sat_u64(std::fmax(0.0, std::floor(std::fma(m.p[1], as_float(k), m.p[0])))); break;
  }
  return p < L - 1 ? p : L - 1;
}

// ---------------------------------------------------------------------------------------------
// cache_fix (cache_fix.rs:109-150), the host half of `--bounded`: a greedy spline over the unique
// keys and their predecessors key-1 whose linear interpolation lands in the right line_size block
// for every point it skips.  Sequential by construction (each decision depends on the last accepted
// spline).  pairs = (key, offset) flattened.
// ---------------------------------------------------------------------------------------------
struct CfSpline { uint64_t from_x, from_y, to_x, to_y; };
inline uint64_t cf_predict(const CfSpline& s, uint64_t inp) {                 // cache_fix.rs:37-43
  const double v0 = (double)s.from_y, v1 = (double)s.to_y;
  const double t = ((double)(inp - s.from_x)) / (double)(s.to_x - s.from_x);
  return sat_u64(std::fma(1.0 - t, v0, t * v1));
}
inline int cache_fix(const uint64_t* keys, uint64_t n, uint64_t line_size, std::vector<uint64_t>& pairs) {
  pairs.clear();
  if (!keys || line_size == 0 || !(n > line_size)) return RMI_ERR_BAD_ARG;   // assert, cache_fix.rs:110
  bool has = false;
  CfSpline sp{0, 0, 0, 0};
  std::vector<uint64_t> px, py;                                               // curr_pts
  int rc = RMI_OK;
  auto add_point = [&](uint64_t x, uint64_t y) {                              // SplineFit::add_point :60-86
    if (!has) { has = true; sp = CfSpline{x, y, x, y}; pairs.push_back(x); pairs.push_back(y); return; }
    const CfSpline last = sp;
    if (!(x >= last.from_x) || !(y >= last.from_y)) { rc = RMI_ERR_BAD_ARG; return; }
    const CfSpline prop{last.from_x, last.from_y, x, y};
    px.push_back(last.to_x); py.push_back(last.to_y);
    bool ok = true;
    for (size_t i = 0; i < px.size(); i++)
      if (cf_predict(prop, px[i]) / line_size != py[i] / line_size) { ok = false; break; }
    if (ok) { sp = prop; return; }
    if (!(x > last.to_x) || !(last.to_y <= y)) { rc = RMI_ERR_BAD_ARG; return; }
    sp = CfSpline{last.to_x, last.to_y, x, y};
    px.clear(); py.clear(); px.push_back(x); py.push_back(y);
    pairs.push_back(last.to_x); pairs.push_back(last.to_y);
  };
  uint64_t last_key = 0;
  for (uint64_t i = 0; i < n && rc == RMI_OK; i++) {                          // iter_unique(): first occurrences
    if (i > 0 && keys[i] == keys[i - 1]) continue;
    const uint64_t key = keys[i];
    if (key == 0 || !(key - 1 >= last_key)) { rc = RMI_ERR_BAD_ARG; break; }   // minus_epsilon / assert :122
    if (key - 1 != last_key) add_point(key - 1, i);
    if (rc == RMI_OK) add_point(key, i);
    last_key = key;
  }
  if (rc != RMI_OK) { pairs.clear(); return rc; }
  if (has) { pairs.push_back(sp.to_x); pairs.push_back(sp.to_y); }            // finish()
  return RMI_OK;
}

}  // namespace rmi_host
