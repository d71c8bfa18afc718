/*
 * rmi_oracle.c -- CPU restatement of learnedsystems/RMI's two-layer training path.
 *
 * TEST INFRASTRUCTURE ONLY (see rmi_oracle.h).  PARITY UNPINNED (no reference-generated
 * golden vectors exist; Rust toolchain absent).  Deliberately follows the reference's
 * control flow literally -- explicit per-leaf containers, FixDups iterators with the
 * tail-duplicate behaviour, boxed-model style dispatch -- rather than the closed forms
 * the GPU code uses, so that agreement between the two checks the derivation.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  FMA is used
 * only where the reference calls f64::mul_add.
 *
 * All citations are relative to /root/reference.
 */
#include "rmi_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

const char* orc_version(void) { return "rmi_oracle 0.1 (restatement of learnedsystems/RMI @2025-10-17)"; }

/* ------------------------------------------------------------------------------------------
 * TrainingKey (rmi_lib/src/models/mod.rs:65-111).  A key is carried as raw bits + dtype.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t bits; } okey;

static inline double bits_to_f64(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t f64_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* Rust `f64 as u64`: saturating, NaN -> 0 */
static inline uint64_t sat_f64_to_u64(double v) {
  if (!(v == v)) return 0;
  if (v <= 0.0) return 0;
  if (v >= 18446744073709551616.0) return UINT64_MAX;
  return (uint64_t)v;
}

/* as_float: mod.rs:83,95,107 */
static inline double key_as_float(int dtype, okey k) {
  if (dtype == ORC_KEY_F64) return bits_to_f64(k.bits);
  return (double)k.bits; /* u64/u32 as f64 : round-to-nearest-even */
}
/* as_uint / ModelInput::as_int: mod.rs:84,96,108,428-433 */
static inline uint64_t key_as_uint(int dtype, okey k) {
  if (dtype == ORC_KEY_F64) return sat_f64_to_u64(bits_to_f64(k.bits));
  return k.bits;
}
/* PartialEq on the key type */
static inline int key_eq(int dtype, okey a, okey b) {
  if (dtype == ORC_KEY_F64) return bits_to_f64(a.bits) == bits_to_f64(b.bits);
  return a.bits == b.bits;
}
/* minus_epsilon / plus_epsilon: mod.rs:78-80, 90-92, 102-104 (release build: wrapping) */
static inline okey key_minus_eps(int dtype, okey k) {
  okey r;
  if (dtype == ORC_KEY_F64) r.bits = f64_to_bits(bits_to_f64(k.bits) - DBL_EPSILON);
  else if (dtype == ORC_KEY_U32) r.bits = (uint32_t)((uint32_t)k.bits - 1u);
  else r.bits = k.bits - 1u;
  return r;
}
static inline okey key_plus_eps(int dtype, okey k) {
  okey r;
  if (dtype == ORC_KEY_F64) r.bits = f64_to_bits(bits_to_f64(k.bits) + DBL_EPSILON);
  else if (dtype == ORC_KEY_U32) r.bits = (uint32_t)((uint32_t)k.bits + 1u);
  else r.bits = k.bits + 1u;
  return r;
}
static inline okey key_zero(int dtype) { okey r; r.bits = (dtype == ORC_KEY_F64) ? f64_to_bits(0.0) : 0; return r; }
static inline okey key_max(int dtype) {
  okey r;
  if (dtype == ORC_KEY_F64) r.bits = f64_to_bits(DBL_MAX);
  else if (dtype == ORC_KEY_U32) r.bits = 0xFFFFFFFFull;
  else r.bits = UINT64_MAX;
  return r;
}

/* ------------------------------------------------------------------------------------------
 * RMITrainingData<T> (mod.rs:233-317) over two providers:
 *   - a raw key slice with y = index     (src/load.rs:21-95, SliceAdapter*)
 *   - a Vec<(K, usize)> of pairs          (mod.rs:124-140)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int dtype;
  const void* raw;      /* raw slice provider (pairs_k == NULL) */
  const okey* pairs_k;  /* pair provider */
  const size_t* pairs_y;
  size_t len;
  double scale;
} otd;

static inline okey raw_key_at(int dtype, const void* raw, size_t i) {
  okey k;
  if (dtype == ORC_KEY_U32) k.bits = ((const uint32_t*)raw)[i];
  else k.bits = ((const uint64_t*)raw)[i];
  return k;
}

/* provider.get(idx): load.rs:32-37 / mod.rs:137-139 */
static inline void otd_provider_get(const otd* d, size_t i, okey* k, size_t* y) {
  if (d->pairs_k) { *k = d->pairs_k[i]; *y = d->pairs_y[i]; }
  else { *k = raw_key_at(d->dtype, d->raw, i); *y = i; }
}

/* map_scale!: mod.rs:238-250 */
static inline size_t otd_scale_y(const otd* d, size_t y) {
  double sf = d->scale;
  int use_sf = fabs(sf - 1.0) > DBL_EPSILON;
  if (use_sf) return (size_t)sat_f64_to_u64((double)y * sf);
  return y;
}

/* RMITrainingData::get / get_key: mod.rs:268-274 (scale applied, NO FixDups) */
static inline void otd_get(const otd* d, size_t i, okey* k, size_t* y) {
  size_t yy; otd_provider_get(d, i, k, &yy); *y = otd_scale_y(d, yy);
}

/* FixDupsIter: mod.rs:143-185; RMITrainingData::iter: mod.rs:276-278.
 * NB (Q1): when the inner iterator is exhausted, next() returns last_item.take(), i.e.
 * the last distinct (key, first_offset) is yielded a second time -> len+1 items. */
typedef struct {
  const otd* d;
  size_t pos;       /* inner cdf_iter position */
  int has_last;
  okey last_k;
  size_t last_y;    /* unscaled */
} fixdups_it;

static inline void fd_init(fixdups_it* it, const otd* d) { it->d = d; it->pos = 0; it->has_last = 0; }

/* advance the inner iterator n items (Iterator::skip on the *outer* iterator is emulated
 * by calling fd_next n times; see callers) */
static inline int fd_next(fixdups_it* it, okey* k, size_t* y) {
  const otd* d = it->d;
  if (!it->has_last) {                               /* mod.rs:161-169 */
    if (it->pos >= d->len) return 0;
    okey kk; size_t yy; otd_provider_get(d, it->pos++, &kk, &yy);
    it->has_last = 1; it->last_k = kk; it->last_y = yy;
    *k = kk; *y = otd_scale_y(d, yy);
    return 1;
  }
  if (it->pos < d->len) {                            /* mod.rs:171-179 */
    okey kk; size_t yy; otd_provider_get(d, it->pos++, &kk, &yy);
    if (key_eq(d->dtype, kk, it->last_k)) {
      *k = kk; *y = otd_scale_y(d, it->last_y);
      return 1;
    }
    it->last_k = kk; it->last_y = yy;
    *k = kk; *y = otd_scale_y(d, yy);
    return 1;
  }
  /* mod.rs:180 : None => self.last_item.take() */
  it->has_last = 0;
  *k = it->last_k; *y = otd_scale_y(d, it->last_y);
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * Model plugins
 * ---------------------------------------------------------------------------------------- */

/* slr: linear.rs:12-59.  `it` yields (x, y); skip/take are applied by the caller through
 * skip_n / take_n (RobustLinearModel, linear.rs:250-252); take_n == SIZE_MAX => unlimited. */
static int slr_over_iter(const otd* d, size_t skip_n, size_t take_n, double* alpha, double* beta) {
  double mean_x = 0.0, mean_y = 0.0, c = 0.0, m2 = 0.0;
  uint64_t n = 0;
  size_t data_size = 0;
  fixdups_it it; fd_init(&it, d);
  okey k; size_t y;
  for (size_t s = 0; s < skip_n; s++) { if (!fd_next(&it, &k, &y)) break; }
  size_t taken = 0;
  while (taken < take_n && fd_next(&it, &k, &y)) {
    taken++;
    double x = key_as_float(d->dtype, k);
    double yf = (double)y;
    n += 1;
    double dx = x - mean_x;
    mean_x += dx / (double)n;
    mean_y += (yf - mean_y) / (double)n;
    c += dx * (yf - mean_y);
    double dx2 = x - mean_x;
    m2 += dx * dx2;
    data_size += 1;
  }
  if (data_size == 0) { *alpha = 0.0; *beta = 0.0; return ORC_OK; }       /* linear.rs:37-39 */
  if (data_size == 1) { *alpha = mean_y; *beta = 0.0; return ORC_OK; }    /* linear.rs:41-43 */
  double cov = c / (double)(n - 1);
  double var = m2 / (double)(n - 1);
  if (!(var >= 0.0)) return ORC_ERR_NEGATIVE_VARIANCE;                    /* linear.rs:48 */
  if (var == 0.0) { *alpha = mean_y; *beta = 0.0; return ORC_OK; }        /* linear.rs:50-53 */
  double b = cov / var;
  double a = mean_y - b * mean_x;
  *alpha = a; *beta = b;
  return ORC_OK;
}

/* LinearModel::new: linear.rs:79-83 */
static int fit_linear(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_LINEAR;
  return slr_over_iter(d, 0, SIZE_MAX, &m->p[0], &m->p[1]);
}

/* loglinear_slr + LogLinearModel::new: linear.rs:60-72, :168-174.  The pairs (x, ln y) with a finite
 * ln y are collected and handed to slr; ln is libm's. */
static int fit_loglinear(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_LOGLINEAR;
  size_t cap = d->len + 1, cnt = 0;
  okey* pk = (okey*)malloc(cap * sizeof(okey));
  double* ly = (double*)malloc(cap * sizeof(double));
  if (!pk || !ly) { free(pk); free(ly); return ORC_ERR_BAD_ARG; }
  fixdups_it it; fd_init(&it, d); okey k; size_t y;
  while (fd_next(&it, &k, &y)) {
    double v = log((double)y);
    if (isfinite(v)) { pk[cnt] = k; ly[cnt] = v; cnt++; }
  }
  double mean_x = 0.0, mean_y = 0.0, c = 0.0, m2 = 0.0;
  uint64_t n = 0;
  for (size_t i = 0; i < cnt; i++) {                                       /* slr: linear.rs:24-34 */
    double x = key_as_float(d->dtype, pk[i]);
    n += 1;
    double dx = x - mean_x;
    mean_x += dx / (double)n;
    mean_y += (ly[i] - mean_y) / (double)n;
    c += dx * (ly[i] - mean_y);
    double dx2 = x - mean_x;
    m2 += dx * dx2;
  }
  free(pk); free(ly);
  if (cnt == 0) { m->p[0] = 0.0; m->p[1] = 0.0; return ORC_OK; }
  if (cnt == 1) { m->p[0] = mean_y; m->p[1] = 0.0; return ORC_OK; }
  double cov = c / (double)(n - 1), var = m2 / (double)(n - 1);
  if (!(var >= 0.0)) return ORC_ERR_NEGATIVE_VARIANCE;
  if (var == 0.0) { m->p[0] = mean_y; m->p[1] = 0.0; return ORC_OK; }
  double b = cov / var;
  m->p[0] = mean_y - b * mean_x; m->p[1] = b;
  return ORC_OK;
}

/* exp1, phi: normal.rs:12-27 (and linear.rs:156-166, stdlib.rs:29-45) */
static inline double exp1(double inp) {
  double x = inp;
  x = 1.0 + x / 64.0;
  x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
  return x;
}
static inline double phi(double x) { return 1.0 / (1.0 + exp1(-1.65451 * x)); }

/* ncdf + NormalModel::new: normal.rs:29-50, :74-78 */
static int fit_normal(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_NORMAL;
  double scale = -INFINITY, mean = 0.0, stdev = 0.0;
  const double n = (double)d->len;
  fixdups_it it; okey k; size_t y;
  fd_init(&it, d);
  while (fd_next(&it, &k, &y)) {
    double x = key_as_float(d->dtype, k);
    mean += x / n;
    scale = fmax(scale, (double)y);
  }
  fd_init(&it, d);
  while (fd_next(&it, &k, &y)) {
    double x = key_as_float(d->dtype, k);
    stdev += (x - mean) * (x - mean);                                      /* powf(2.0) */
  }
  stdev /= n;
  stdev = sqrt(stdev);
  m->p[0] = mean; m->p[1] = stdev; m->p[2] = scale;
  return ORC_OK;
}

/* RobustLinearModel::new: linear.rs:239-260 */
static int fit_robust_linear(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_ROBUST_LINEAR;
  size_t total = d->len;
  if (total == 0) { m->p[0] = 0.0; m->p[1] = 0.0; return ORC_OK; }
  size_t bnd = (size_t)sat_f64_to_u64((double)total * 0.0001);
  if (bnd < 1) bnd = 1;
  if (!(bnd * 2 + 1 < total)) return ORC_ERR_ROBUST_TOO_SMALL;
  return slr_over_iter(d, bnd, total - 2 * bnd, &m->p[0], &m->p[1]);
}

/* linear_splines: linear_spline.rs:13-35 */
static void linear_splines(const otd* d, double* alpha, double* beta) {
  if (d->len == 0) { *alpha = 0.0; *beta = 0.0; return; }
  okey k0, k1; size_t y0, y1;
  if (d->len == 1) { otd_get(d, 0, &k0, &y0); *alpha = (double)y0; *beta = 0.0; return; }
  otd_get(d, 0, &k0, &y0);
  otd_get(d, d->len - 1, &k1, &y1);
  if (key_eq(d->dtype, k0, k1)) { *alpha = (double)y0; *beta = 0.0; return; }
  double slope = ((double)y0 - (double)y1) / (key_as_float(d->dtype, k0) - key_as_float(d->dtype, k1));
  double intercept = (double)y0 - slope * key_as_float(d->dtype, k0);
  *alpha = intercept; *beta = slope;
}
static int fit_linear_spline(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_LINEAR_SPLINE;
  linear_splines(d, &m->p[0], &m->p[1]);
  return ORC_OK;
}

/* scale!: cubic_spline.rs:11-15 */
#define CS_SCALE(v, mn, mx) (((v) - (mn)) / ((mx) - (mn)))

/* cubic(): cubic_spline.rs:18-101.  powf(2.0) is lowered to x*x by LLVM; powf(3.0) is a
 * libm pow call. */
static int cubic_fn(const otd* d, double out[4]) {
  if (d->len == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; out[3] = 0.0; return ORC_OK; }
  okey k0; size_t y0;
  if (d->len == 1) { otd_get(d, 0, &k0, &y0); out[0] = out[1] = out[2] = 0.0; out[3] = (double)y0; return ORC_OK; }

  otd_get(d, 0, &k0, &y0);
  { /* cubic_spline.rs:28-36 : data.iter().any(|(x,_)| x != candidate) */
    fixdups_it it; fd_init(&it, d); okey k; size_t y; int uniq = 0;
    while (fd_next(&it, &k, &y)) { if (!key_eq(d->dtype, k, k0)) { uniq = 1; break; } }
    if (!uniq) { out[0] = out[1] = out[2] = 0.0; out[3] = (double)y0; return ORC_OK; }
  }
  okey kl; size_t yl;
  otd_get(d, d->len - 1, &kl, &yl);
  double xmin = key_as_float(d->dtype, k0), ymin = (double)y0;
  double xmax = key_as_float(d->dtype, kl), ymax = (double)yl;
  double x1 = 0.0, y1 = 0.0, x2 = 1.0, y2 = 1.0;

  double m1;
  { /* cubic_spline.rs:46-54 : first item of iter() with scaled x > 0; unwrap() */
    fixdups_it it; fd_init(&it, d); okey k; size_t y; int found = 0; okey xn = k0; size_t yn = y0;
    while (fd_next(&it, &k, &y)) {
      if (CS_SCALE(key_as_float(d->dtype, k), xmin, xmax) > 0.0) { xn = k; yn = y; found = 1; break; }
    }
    if (!found) return ORC_ERR_CUBIC_DEGENERATE;  /* .unwrap() on None: distinct keys collapsing to one f64 */
    double sxn = CS_SCALE(key_as_float(d->dtype, xn), xmin, xmax);
    double syn = CS_SCALE((double)yn, ymin, ymax);
    m1 = (syn - y1) / (sxn - x1);
  }
  double m2;
  { /* cubic_spline.rs:56-65 : last idx (via get) with scaled x < 1 */
    okey xp = k0; size_t yp = y0; int found = 0;
    for (size_t idx = d->len; idx-- > 0;) {
      okey k; size_t y; otd_get(d, idx, &k, &y);
      if (CS_SCALE(key_as_float(d->dtype, k), xmin, xmax) < 1.0) { xp = k; yp = y; found = 1; break; }
    }
    if (!found) return ORC_ERR_CUBIC_DEGENERATE;
    double sxp = CS_SCALE(key_as_float(d->dtype, xp), xmin, xmax);
    double syp = CS_SCALE((double)yp, ymin, ymax);
    m2 = (y2 - syp) / (x2 - sxp);
  }
  /* cubic_spline.rs:68-72 */
  if (m1 * m1 + m2 * m2 > 9.0) {
    double tau = 3.0 / sqrt(m1 * m1 + m2 * m2);
    m1 *= tau; m2 *= tau;
  }
  double den = pow(xmax - xmin, 3.0);
  double a = (m1 + m2 - 2.0) / den;                                                     /* :76 */
  double b = -(xmax * (2.0 * m1 + m2 - 3.0) + xmin * (m1 + 2.0 * m2 - 3.0)) / den;      /* :80-81 */
  double c = (m1 * (xmax * xmax) + m2 * (xmin * xmin) + xmax * xmin * (2.0 * m1 + 2.0 * m2 - 6.0)) / den; /* :86-88 */
  double dd = -xmin * (m1 * (xmax * xmax) + xmax * xmin * (m2 - 3.0) + (xmin * xmin)) / den; /* :92-93 */
  a *= ymax - ymin; b *= ymax - ymin; c *= ymax - ymin; dd *= ymax - ymin; dd += ymin;  /* :95-99 */
  out[0] = a; out[1] = b; out[2] = c; out[3] = dd;
  return ORC_OK;
}

static inline double cubic_predict(const double p[4], double val) {
  /* cubic_spline.rs:140-151 */
  double v1 = fma(p[0], val, p[1]);
  double v2 = fma(v1, val, p[2]);
  double v3 = fma(v2, val, p[3]);
  return v3;
}

/* CubicSplineModel::new: cubic_spline.rs:108-136 */
static int fit_cubic(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_CUBIC;
  double cp[4]; int rc = cubic_fn(d, cp);
  if (rc) return rc;
  double la, lb; linear_splines(d, &la, &lb);
  double our_error = 0.0, lin_error = 0.0;
  fixdups_it it; fd_init(&it, d); okey k; size_t y;
  while (fd_next(&it, &k, &y)) {
    /* iter_model_input(): Int keys -> ModelInput::Int -> as_float() = (u64 as f64) */
    double xf = key_as_float(d->dtype, k);
    double c_pred = cubic_predict(cp, xf);
    double l_pred = fma(lb, xf, la);            /* linear_spline.rs:50-53 */
    our_error += fabs(c_pred - (double)y);
    lin_error += fabs(l_pred - (double)y);
  }
  if (lin_error < our_error) { m->p[0] = 0.0; m->p[1] = 0.0; m->p[2] = lb; m->p[3] = la; }
  else memcpy(m->p, cp, sizeof cp);
  return ORC_OK;
}

/* num_bits: utils.rs:13-21 (returns -1 where the reference asserts) */
int orc_num_bits(uint64_t largest_target) {
  int nbits = 0;
  /* (1 << (nbits+1)) - 1 <= largest_target ; the literal is u64 there */
  while (nbits + 1 < 64 && ((1ull << (nbits + 1)) - 1) <= largest_target) nbits += 1;
  if (nbits < 1) return -1;
  return nbits;
}

/* common_prefix_size: utils.rs:23-36 -- iterates iter_model_input() (FixDups + Q1, no effect
 * on an OR/AND fold) */
static int common_prefix_size_td(const otd* d) {
  uint64_t any_ones = 0, no_ones = ~0ull;
  fixdups_it it; fd_init(&it, d); okey k; size_t y;
  while (fd_next(&it, &k, &y)) {
    uint64_t v = key_as_uint(d->dtype, k);
    any_ones |= v; no_ones &= v;
  }
  uint64_t any_zeros = ~no_ones;
  uint64_t prefix_bits = any_zeros ^ any_ones;
  uint64_t inv = ~prefix_bits;
  if (inv == 0) return 64;
  return __builtin_clzll(inv);
}

int orc_common_prefix_size(int dtype, const void* keys, size_t len) {
  otd d = { dtype, keys, NULL, NULL, len, 1.0 };
  return common_prefix_size_td(&d);
}

/* RadixModel::new: radix.rs:18-39 */
static int fit_radix(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_RADIX;
  m->ip[0] = 0; m->ip[1] = 0;
  if (d->len == 0) return ORC_OK;
  uint64_t largest = 0;
  fixdups_it it; fd_init(&it, d); okey k; size_t y;
  while (fd_next(&it, &k, &y)) { if ((uint64_t)y > largest) largest = y; }
  int bits = orc_num_bits(largest);
  if (bits < 0) return ORC_ERR_NUM_BITS;
  int prefix = common_prefix_size_td(d);
  m->ip[0] = (uint64_t)(uint8_t)prefix;
  m->ip[1] = (uint64_t)(uint8_t)bits;
  return ORC_OK;
}

/* BalancedRadixModel: balanced_radix.rs.  predict_to_int :104-116 */
static inline uint64_t bradix_predict(const orc_model* m, uint64_t as_int) {
  const uint64_t res = (as_int << (m->ip[0] & 63)) >> ((64 - m->ip[1]) & 63);   /* release-mode masked shifts */
  const uint64_t clamp = m->ip[2];
  if (m->ip[3]) return res < clamp ? res : clamp;                               /* u64::min(res, clamp) */
  return res < clamp ? 0 : res - clamp;
}
/* chi2: :20-38.  counts are i32 there (integer literal default); the sum starts from 0.0 in bin order */
static int bradix_chi2(const otd* d, uint64_t max_bin, const orc_model* m, double* score) {
  int32_t* counts = (int32_t*)calloc(max_bin ? max_bin : 1, sizeof(int32_t));
  if (!counts) return ORC_ERR_BAD_ARG;
  fixdups_it it; fd_init(&it, d); okey k; size_t y;
  while (fd_next(&it, &k, &y)) {                                                /* iter_model_input(): N+1 items */
    const uint64_t b = bradix_predict(m, key_as_uint(d->dtype, k));
    if (b >= max_bin) { free(counts); return ORC_ERR_BAD_ARG; }                 /* index out of bounds: panic */
    counts[b] = (int32_t)((uint32_t)counts[b] + 1u);
  }
  const double expected = (double)d->len / (double)max_bin;
  double sum = 0.0;
  for (uint64_t b = 0; b < max_bin; b++) {
    const double df = (double)counts[b] - expected;
    sum += (df * df) / expected;                                                /* powf(2.0) */
  }
  free(counts);
  *score = sum;
  return ORC_OK;
}
/* bradix: :40-87; BalancedRadixModel::new: :90-101 */
static int fit_bradix(const otd* d, orc_model* m) {
  m->kind = ORC_MODEL_BRADIX;
  m->ip[0] = 0; m->ip[1] = 0; m->ip[2] = 0; m->ip[3] = 1;
  if (d->len == 0) return ORC_OK;
  uint64_t max_output = 0;
  { fixdups_it it; fd_init(&it, d); okey k; size_t y;
    while (fd_next(&it, &k, &y)) { if ((uint64_t)y > max_output) max_output = y; } }
  const int bits = orc_num_bits(max_output);
  if (bits < 0) return ORC_ERR_NUM_BITS;
  const int prefix = common_prefix_size_td(d);
  double best = INFINITY;
  int have = 0;
  const int end = bits + 2 < 64 ? bits + 2 : 64;
  for (int tb = bits; tb < end; tb++) {
    const uint64_t bits_max = (1ull << ((tb + 1) & 63)) - 1;
    orc_model cand; memset(&cand, 0, sizeof cand);
    cand.kind = ORC_MODEL_BRADIX;
    cand.ip[0] = (uint64_t)(uint8_t)prefix; cand.ip[1] = (uint64_t)tb;
    double score;
    cand.ip[2] = max_output - 1; cand.ip[3] = 1;                                /* high */
    int rc = bradix_chi2(d, max_output, &cand, &score);
    if (rc) return rc;
    if (score < best) { best = score; *m = cand; have = 1; }
    cand.ip[2] = max_output - bits_max; cand.ip[3] = 0;                         /* low; wrapping sub (release build) */
    rc = bradix_chi2(d, max_output, &cand, &score);
    if (rc) return rc;
    if (score < best) { best = score; *m = cand; have = 1; }
  }
  return have ? ORC_OK : ORC_ERR_BAD_ARG;                                       /* best_result.unwrap() */
}

/* RadixTable::new: radix.rs:90-121 */
static inline int is_radix_table(int kind) { return kind >= ORC_MODEL_RADIX8 && kind <= ORC_MODEL_RADIX28; }
static inline int radix_table_bits(int kind) {
  static const int bits[5] = { 8, 18, 22, 26, 28 };
  return bits[kind - ORC_MODEL_RADIX8];
}
static inline uint64_t radix_table_slot(uint64_t prefix, uint64_t bits, uint64_t x) {
  /* radix.rs:98-99 / :125-131; release-mode masked shifts */
  uint64_t num_bits = (prefix + bits > 64) ? 0 : 64 - (prefix + bits);
  return (((x << (prefix & 63)) >> (prefix & 63)) >> (num_bits & 63));
}
static int fit_radix_table(const otd* d, int kind, orc_model* m) {
  const uint64_t bits = (uint64_t)radix_table_bits(kind);
  const uint64_t prefix = (uint64_t)(uint8_t)common_prefix_size_td(d);
  const uint64_t len = 1ull << bits;
  uint32_t* t = (uint32_t*)calloc(len, sizeof(uint32_t));
  if (!t) return ORC_ERR_BAD_ARG;
  uint64_t last_radix = 0;
  fixdups_it it; fd_init(&it, d); okey k; size_t y;
  while (fd_next(&it, &k, &y)) {
    uint64_t x = key_as_uint(d->dtype, k);
    uint64_t cur = radix_table_slot(prefix, bits, x);
    if (cur == last_radix) continue;
    if (cur >= len) { free(t); return ORC_ERR_BAD_ARG; }     /* assert!, radix.rs:101 (cannot fire) */
    t[cur] = (uint32_t)y;
    for (uint64_t i = last_radix + 1; i < cur; i++) t[i] = (uint32_t)y;
    last_radix = cur;
  }
  for (uint64_t i = last_radix + 1; i < len; i++) t[i] = (uint32_t)len;
  m->kind = kind; m->ip[0] = prefix; m->ip[1] = bits; m->table = t; m->table_len = len;
  return ORC_OK;
}
void orc_model_free(orc_model* m) { if (m && m->table) { free(m->table); m->table = NULL; m->table_len = 0; } }

/* train_model: train/mod.rs:35-57 */
static int train_model(int kind, const otd* d, orc_model* m) {
  memset(m, 0, sizeof *m);
  switch (kind) {
    case ORC_MODEL_LINEAR: return fit_linear(d, m);
    case ORC_MODEL_ROBUST_LINEAR: return fit_robust_linear(d, m);
    case ORC_MODEL_LOGLINEAR: return fit_loglinear(d, m);
    case ORC_MODEL_NORMAL: return fit_normal(d, m);
    case ORC_MODEL_LINEAR_SPLINE: return fit_linear_spline(d, m);
    case ORC_MODEL_CUBIC: return fit_cubic(d, m);
    case ORC_MODEL_RADIX: return fit_radix(d, m);
    case ORC_MODEL_BRADIX: return fit_bradix(d, m);
    case ORC_MODEL_RADIX8: case ORC_MODEL_RADIX18: case ORC_MODEL_RADIX22: case ORC_MODEL_RADIX26:
    case ORC_MODEL_RADIX28: return fit_radix_table(d, kind, m);
    default: return ORC_ERR_UNKNOWN_MODEL;
  }
}

static inline int model_needs_bounds_check(int kind) {
  /* default true (mod.rs:751-753); cubic false (cubic_spline.rs:184-186); radix false (radix.rs:72-74) */
  return !(kind == ORC_MODEL_CUBIC || kind == ORC_MODEL_RADIX || kind == ORC_MODEL_BRADIX || is_radix_table(kind));   /* radix.rs:160-162, balanced_radix.rs:164-166 */
}
static inline int model_params_per(int kind) { return kind == ORC_MODEL_CUBIC ? 4 : 2; }

/* predict_to_float */
static inline double model_predict_float_k(const orc_model* m, int dtype, okey k) {
  switch (m->kind) {
    case ORC_MODEL_LINEAR: case ORC_MODEL_ROBUST_LINEAR: case ORC_MODEL_LINEAR_SPLINE:
      return fma(m->p[1], key_as_float(dtype, k), m->p[0]);       /* linear.rs:87-90 */
    case ORC_MODEL_CUBIC:
      return cubic_predict(m->p, key_as_float(dtype, k));
    case ORC_MODEL_LOGLINEAR:
      return exp1(fma(m->p[1], key_as_float(dtype, k), m->p[0]));  /* linear.rs:177-180 */
    case ORC_MODEL_NORMAL:
      return phi((key_as_float(dtype, k) - m->p[0]) / m->p[1]) * m->p[2];   /* normal.rs:81-84 */
    case ORC_MODEL_RADIX: {
      /* default predict_to_float = predict_to_int as f64 (mod.rs:731-733) */
      uint64_t v = key_as_uint(dtype, k);
      uint64_t r = (v << (m->ip[0] & 63)) >> ((64 - m->ip[1]) & 63);
      return (double)r;
    }
    case ORC_MODEL_BRADIX: return (double)bradix_predict(m, key_as_uint(dtype, k));
    case ORC_MODEL_RADIX8: case ORC_MODEL_RADIX18: case ORC_MODEL_RADIX22: case ORC_MODEL_RADIX26:
    case ORC_MODEL_RADIX28:
      return (double)m->table[radix_table_slot(m->ip[0], m->ip[1], key_as_uint(dtype, k))];
  }
  return 0.0;
}
/* predict_to_int: mod.rs:735-737; radix.rs:43-50 */
static inline uint64_t model_predict_int_k(const orc_model* m, int dtype, okey k) {
  if (m->kind == ORC_MODEL_RADIX) {
    uint64_t v = key_as_uint(dtype, k);
    return (v << (m->ip[0] & 63)) >> ((64 - m->ip[1]) & 63);  /* release-mode masked shifts */
  }
  if (m->kind == ORC_MODEL_BRADIX) return bradix_predict(m, key_as_uint(dtype, k));
  if (is_radix_table(m->kind))                                /* radix.rs:124-134 */
    return (uint64_t)m->table[radix_table_slot(m->ip[0], m->ip[1], key_as_uint(dtype, k))];
  double f = floor(model_predict_float_k(m, dtype, k));
  return sat_f64_to_u64(fmax(0.0, f));
}
/* set_to_constant_model: linear.rs:116-119, linear_spline.rs:79-82, cubic_spline.rs:188-191 */
static inline int model_set_constant(orc_model* m, uint64_t c) {
  switch (m->kind) {
    case ORC_MODEL_LINEAR: case ORC_MODEL_ROBUST_LINEAR: case ORC_MODEL_LINEAR_SPLINE:
      m->p[0] = (double)c; m->p[1] = 0.0; return 1;
    case ORC_MODEL_CUBIC:
      m->p[0] = 0.0; m->p[1] = 0.0; m->p[2] = 0.0; m->p[3] = (double)c; return 1;
  }
  return 0;
}

double orc_predict_to_float(const orc_model* m, int dtype, uint64_t key_bits) {
  okey k; k.bits = key_bits; return model_predict_float_k(m, dtype, k);
}
uint64_t orc_predict_to_int(const orc_model* m, int dtype, uint64_t key_bits) {
  okey k; k.bits = key_bits; return model_predict_int_k(m, dtype, k);
}

int orc_fit_pairs(int kind, int dtype, const void* keys, const uint64_t* ys, size_t len,
                  double scale, orc_model* out) {
  okey* pk = (okey*)malloc((len ? len : 1) * sizeof(okey));
  size_t* py = (size_t*)malloc((len ? len : 1) * sizeof(size_t));
  for (size_t i = 0; i < len; i++) { pk[i] = raw_key_at(dtype, keys, i); py[i] = (size_t)ys[i]; }
  otd d = { dtype, NULL, pk, py, len, scale };
  int rc = train_model(kind, &d, out);
  free(pk); free(py);
  return rc;
}

/* validate: train/mod.rs:59-85.  radix is MustBeTop (radix.rs:75-80). */
static int validate(int root_kind, int leaf_kind) {
  int kinds[2] = { root_kind, leaf_kind };
  for (int idx = 0; idx < 2; idx++) {
    switch (kinds[idx]) {
      case ORC_MODEL_LINEAR: case ORC_MODEL_ROBUST_LINEAR: case ORC_MODEL_LINEAR_SPLINE:
      case ORC_MODEL_CUBIC: break;
      case ORC_MODEL_LOGLINEAR: case ORC_MODEL_NORMAL:
        /* no restriction in the reference; as leaves their fits are not restated here */
        if (idx != 0) return ORC_ERR_BAD_ARG;
        break;
      case ORC_MODEL_RADIX: case ORC_MODEL_BRADIX:             /* MustBeTop: radix.rs:75-80, balanced_radix.rs:167-169 */
        if (idx != 0) return ORC_ERR_RESTRICTION;
        break;
      case ORC_MODEL_RADIX8: case ORC_MODEL_RADIX18: case ORC_MODEL_RADIX22: case ORC_MODEL_RADIX26:
      case ORC_MODEL_RADIX28:
        /* ModelRestriction::None (radix.rs:163-165); as a leaf its parameters are a table per leaf,
         * which this restatement does not carry */
        if (idx != 0) return ORC_ERR_BAD_ARG;
        break;
      default: return ORC_ERR_UNKNOWN_MODEL;
    }
  }
  return ORC_OK;
}

int orc_fit_root(int root_kind, int dtype, const void* keys, uint64_t n, uint64_t num_leaves,
                 orc_model* out) {
  otd d = { dtype, keys, NULL, NULL, (size_t)n, 1.0 };
  d.scale = (double)num_leaves / (double)n;      /* two_layer.rs:109 */
  return train_model(root_kind, &d, out);
}

int orc_bucket_ids(const orc_model* root, int dtype, const void* keys, uint64_t n,
                   uint64_t num_leaves, uint64_t* out_ids) {
  for (uint64_t i = 0; i < n; i++) {
    uint64_t p = model_predict_int_k(root, dtype, raw_key_at(dtype, keys, i));
    out_ids[i] = p < num_leaves - 1 ? p : num_leaves - 1;
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * build_models_from: two_layer.rs:20-99
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const otd* data; const orc_model* top; int leaf_kind;
  size_t start_idx, end_idx, first_model_idx, num_models;
  orc_model* out;       /* [num_models] */
  int rc;
} bmf_args;

typedef struct { okey* k; size_t* y; size_t len, cap; } pairvec;
static void pv_push(pairvec* v, okey k, size_t y) {
  if (v->len == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->k = (okey*)realloc(v->k, v->cap * sizeof(okey));
    v->y = (size_t*)realloc(v->y, v->cap * sizeof(size_t));
  }
  v->k[v->len] = k; v->y[v->len] = y; v->len++;
}

static void* build_models_from(void* argp) {
  bmf_args* a = (bmf_args*)argp;
  const otd* data = a->data;
  a->rc = ORC_OK;
  if (!(a->end_idx > a->start_idx) || a->end_idx > data->len || a->start_idx > data->len) {
    a->rc = ORC_ERR_DEGENERATE_SPLIT; return NULL;                       /* two_layer.rs:27-31 */
  }
  otd dummy = { data->dtype, NULL, NULL, NULL, 0, 1.0 };   /* RMITrainingData::empty(): two_layer.rs:33 */
  size_t n_out = 0;
  pairvec sld = { NULL, NULL, 0, 0 };
  size_t last_target = a->first_model_idx;

  /* data.iter().skip(start_idx).take(end_idx - start_idx): two_layer.rs:39-41 */
  fixdups_it it; fd_init(&it, data);
  okey x; size_t y;
  for (size_t s = 0; s < a->start_idx; s++) fd_next(&it, &x, &y);
  size_t remaining = a->end_idx - a->start_idx;

  while (remaining-- > 0 && fd_next(&it, &x, &y)) {
    size_t model_pred = (size_t)model_predict_int_k(a->top, data->dtype, x);
    if (!(model_needs_bounds_check(a->top->kind) || model_pred < a->first_model_idx + a->num_models)) {
      a->rc = ORC_ERR_ROOT_OUT_OF_BOUNDS; goto done;                       /* two_layer.rs:45-48 */
    }
    size_t target = a->first_model_idx + a->num_models - 1;
    if (model_pred < target) target = model_pred;                          /* :49 */
    if (!(target >= last_target)) { a->rc = ORC_ERR_NON_MONOTONE; goto done; } /* :50 */

    if (target > last_target) {                                            /* :52 */
      int had_last = sld.len > 0; okey lk = x; size_t ly = 0;
      if (had_last) { lk = sld.k[sld.len - 1]; ly = sld.y[sld.len - 1]; }  /* :58 */
      pv_push(&sld, x, y);                                                 /* :59 */
      otd cont = { data->dtype, NULL, sld.k, sld.y, sld.len, 1.0 };        /* :61 */
      int rc = train_model(a->leaf_kind, &cont, &a->out[n_out++]);         /* :62-63 */
      if (rc) { a->rc = rc; goto done; }
      for (size_t sk = last_target + 1; sk < target; sk++) {               /* :67-69 */
        rc = train_model(a->leaf_kind, &dummy, &a->out[n_out++]);
        if (rc) { a->rc = rc; goto done; }
      }
      sld.len = 0;                                                         /* :72 */
      if (had_last) pv_push(&sld, lk, ly);                                 /* :76-78 */
    }
    pv_push(&sld, x, y);                                                   /* :82 */
    last_target = target;
  }
  { /* :86-90 */
    otd cont = { data->dtype, NULL, sld.k, sld.y, sld.len, 1.0 };
    int rc = train_model(a->leaf_kind, &cont, &a->out[n_out++]);
    if (rc) { a->rc = rc; goto done; }
  }
  for (size_t sk = last_target + 1; sk < a->first_model_idx + a->num_models; sk++) { /* :94-96 */
    int rc = train_model(a->leaf_kind, &dummy, &a->out[n_out++]);
    if (rc) { a->rc = rc; goto done; }
  }
  if (n_out != a->num_models) a->rc = ORC_ERR_BAD_ARG;                     /* :97 */
done:
  free(sld.k); free(sld.y);
  return NULL;
}

/* lower_bound_by: mod.rs:294-309 with the comparator of two_layer.rs:132-136 */
static size_t lower_bound_by_target(const otd* d, const orc_model* top, uint64_t L, uint64_t midpoint) {
  size_t size = d->len;
  if (size == 0) return 0;
  size_t base = 0;
  while (size > 1) {
    size_t half = size / 2;
    size_t mid = base + half;
    okey k; size_t y; otd_get(d, mid, &k, &y);
    uint64_t mi = model_predict_int_k(top, d->dtype, k);
    uint64_t mt = mi < L - 1 ? mi : L - 1;
    if (mt < midpoint) base = mid;
    size -= half;
  }
  okey k; size_t y; otd_get(d, base, &k, &y);
  uint64_t mi = model_predict_int_k(top, d->dtype, k);
  uint64_t mt = mi < L - 1 ? mi : L - 1;
  return base + (mt < midpoint ? 1 : 0);
}

static inline uint64_t error_between(uint64_t v1, uint64_t v2, uint64_t max_pred) { /* two_layer.rs:14-18 */
  uint64_t p1 = v1 < max_pred ? v1 : max_pred;
  uint64_t p2 = v2 < max_pred ? v2 : max_pred;
  return (p1 > p2 ? p1 : p2) - (p1 < p2 ? p1 : p2);
}

/* ------------------------------------------------------------------------------------------
 * train_two_layer: two_layer.rs:101-306 (+ LowerBoundCorrection::new, lower_bound_correction.rs)
 * ---------------------------------------------------------------------------------------- */
int orc_train_two_layer(int root_kind, int leaf_kind, int dtype, const void* keys, uint64_t n,
                        uint64_t num_leaves, const orc_model* root_override, int threads,
                        orc_trained_rmi* out) {
  int rc = validate(root_kind, leaf_kind);                                 /* :104 */
  if (rc) return rc;
  if (n == 0 || num_leaves == 0) return ORC_ERR_BAD_ARG;
  const uint64_t L = num_leaves;
  const size_t num_rows = (size_t)n;
  otd md = { dtype, keys, NULL, NULL, num_rows, 1.0 };

  orc_model top;
  if (root_override) top = *root_override;
  else {
    md.scale = (double)L / (double)num_rows;                               /* :109 */
    rc = train_model(root_kind, &md, &top);                                /* :110 */
    if (rc) return rc;
  }
  md.scale = 1.0;                                                          /* :128 */

  uint64_t midpoint_model = L / 2;                                         /* :131 */
  size_t split_idx = lower_bound_by_target(&md, &top, L, midpoint_model);  /* :132-136 */
  if (split_idx > 0 && split_idx < md.len) {                               /* :139-145 */
    okey ka, kp; size_t yy;
    otd_get(&md, split_idx, &ka, &yy); otd_get(&md, split_idx - 1, &kp, &yy);
    uint64_t key_at = model_predict_int_k(&top, dtype, ka);
    uint64_t key_pr = model_predict_int_k(&top, dtype, kp);
    if (!(key_at > key_pr)) return ORC_ERR_DEGENERATE_SPLIT;
  }

  orc_model* leaf_models = (orc_model*)calloc((size_t)L, sizeof(orc_model));
  if (!leaf_models) return ORC_ERR_BAD_ARG;

  if (split_idx >= md.len) {                                               /* :147-150 */
    bmf_args a = { &md, &top, leaf_kind, 0, md.len, 0, (size_t)L, leaf_models, 0 };
    build_models_from(&a);
    if (a.rc) { free(leaf_models); return a.rc; }
  } else {
    okey ks; size_t yy; otd_get(&md, split_idx, &ks, &yy);
    uint64_t pi = model_predict_int_k(&top, dtype, ks);
    size_t st = (size_t)(pi < L - 1 ? pi : L - 1);                          /* :152-156 */
    bmf_args a1 = { &md, &top, leaf_kind, 0, split_idx, 0, st, leaf_models, 0 };
    bmf_args a2 = { &md, &top, leaf_kind, split_idx + 1, md.len, st, (size_t)L - st, leaf_models + st, 0 };
    if (threads >= 2) {                                                    /* rayon::join :161-169 */
      pthread_t t;
      pthread_create(&t, NULL, build_models_from, &a2);
      build_models_from(&a1);
      pthread_join(t, NULL);
    } else {
      build_models_from(&a1);
      build_models_from(&a2);
    }
    if (a1.rc) { free(leaf_models); return a1.rc; }
    if (a2.rc) { free(leaf_models); return a2.rc; }
  }

  /* ---- LowerBoundCorrection::new: lower_bound_correction.rs:92-137 ---- */
  typedef struct { int some; size_t idx; okey key; } optpair;
  optpair* first_key = (optpair*)calloc((size_t)L, sizeof(optpair));
  optpair* last_key = (optpair*)calloc((size_t)L, sizeof(optpair));
  uint64_t* max_run = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
  size_t* next_idx = (size_t*)calloc((size_t)L, sizeof(size_t));
  okey* next_key = (okey*)calloc((size_t)L, sizeof(okey));
  size_t* prev_idx = (size_t*)calloc((size_t)L, sizeof(size_t));
  okey* prev_key = (okey*)calloc((size_t)L, sizeof(okey));
  uint64_t* l1_n = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
  uint64_t* l1_e = (uint64_t*)calloc((size_t)L, sizeof(uint64_t));
  {
    size_t last_target = 0;
    uint64_t current_run_length = 0;
    okey current_run_key; size_t yy0; otd_get(&md, 0, &current_run_key, &yy0);   /* :103 */
    fixdups_it it; fd_init(&it, &md); okey x; size_t y;
    uint64_t last_seen_target = 0; int have_seen = 0;
    while (fd_next(&it, &x, &y)) {                                         /* :104 */
      uint64_t leaf_idx = model_predict_int_k(&top, dtype, x);
      size_t target = (size_t)(leaf_idx < L - 1 ? leaf_idx : L - 1);       /* :106 */
      if (out->leaf_start) {          /* diagnostic only: bucket boundaries */
        if (!have_seen) { for (uint64_t j = 0; j <= target; j++) out->leaf_start[j] = 0; have_seen = 1; }
        else if (target > last_seen_target) {
          /* y of a non-dup first key of a leaf == its index */
          for (uint64_t j = last_seen_target + 1; j <= target; j++) out->leaf_start[j] = y;
        }
        last_seen_target = target;
      }
      if (target == last_target && key_eq(dtype, x, current_run_key)) {    /* :108-109 */
        current_run_length += 1;
      } else if (target != last_target || !key_eq(dtype, x, current_run_key)) { /* :110-119 */
        if (current_run_length > max_run[last_target]) max_run[last_target] = current_run_length;
        current_run_length = 1;
        current_run_key = x;
        last_target = target;
      }
      if (!first_key[target].some) { first_key[target].some = 1; first_key[target].idx = y; first_key[target].key = x; }
      last_key[target].some = 1; last_key[target].idx = y; last_key[target].key = x;
    }
    if (out->leaf_start) for (uint64_t j = last_seen_target + 1; j <= L; j++) out->leaf_start[j] = n;

    /* compute_next_for_leaf: lower_bound_correction.rs:30-56 */
    {
      size_t idx = 0;
      while (idx < (size_t)L) {
        size_t found = (size_t)L;  /* find_first_above: :16-26 */
        if (idx != (size_t)L - 1) {
          for (size_t i = idx + 1; i < (size_t)L; i++) { if (first_key[i].some) { found = i; break; } }
        }
        if (found < (size_t)L) {
          for (size_t i = idx; i < found; i++) { next_idx[i] = first_key[found].idx; next_key[i] = first_key[found].key; }
          idx = found;
        } else {
          for (size_t i = idx; i < (size_t)L; i++) { next_idx[i] = md.len; next_key[i] = key_max(dtype); }
          break;
        }
      }
    }
    /* compute_prev_for_leaf: lower_bound_correction.rs:58-80 */
    {
      for (size_t i = 0; i < (size_t)L; i++) { prev_idx[i] = 0; prev_key[i] = key_zero(dtype); }
      size_t idx = (size_t)L - 1;
      while (idx > 0) {
        size_t found = (size_t)L; /* find_first_below: :4-14 */
        for (size_t i = idx; i-- > 0;) { if (last_key[i].some) { found = i; break; } }
        if (found < (size_t)L) {
          for (size_t i = found + 1; i < idx + 1; i++) { prev_idx[i] = last_key[found].idx; prev_key[i] = last_key[found].key; }
          idx = found;
        } else break;
      }
    }
  }

  /* ---- empty-leaf fix: two_layer.rs:185-197 ---- */
  for (size_t idx = 0; idx + 1 < (size_t)L; idx++) {
    if (!last_key[idx].some) model_set_constant(&leaf_models[idx], (uint64_t)next_idx[idx]);
  }

  /* ---- last-level errors: two_layer.rs:207-217 ---- */
  {
    fixdups_it it; fd_init(&it, &md); okey x; size_t y;
    while (fd_next(&it, &x, &y)) {
      uint64_t leaf_idx = model_predict_int_k(&top, dtype, x);
      size_t target = (size_t)(leaf_idx < L - 1 ? leaf_idx : L - 1);
      uint64_t pred = model_predict_int_k(&leaf_models[target], dtype, x);
      uint64_t err = error_between(pred, (uint64_t)y, (uint64_t)md.len);
      l1_n[target] += 1;
      if (err > l1_e[target]) l1_e[target] = err;
    }
  }

  /* ---- lower-bound widening: two_layer.rs:226-259 ---- */
  for (size_t leaf_idx = 0; leaf_idx < (size_t)L; leaf_idx++) {
    uint64_t curr_err = l1_e[leaf_idx];
    uint64_t upper_error, lower_error;
    {
      uint64_t pred = model_predict_int_k(&leaf_models[leaf_idx], dtype, key_minus_eps(dtype, next_key[leaf_idx]));
      upper_error = error_between(pred, (uint64_t)next_idx[leaf_idx] + 1, (uint64_t)md.len);
    }
    {
      size_t pidx = leaf_idx == 0 ? 0 : leaf_idx - 1;
      size_t first_idx = next_idx[pidx];
      uint64_t pred = model_predict_int_k(&leaf_models[leaf_idx], dtype, key_plus_eps(dtype, prev_key[leaf_idx]));
      lower_error = error_between(pred, (uint64_t)first_idx, (uint64_t)md.len);
    }
    uint64_t mx = curr_err;
    if (upper_error > mx) mx = upper_error;
    if (lower_error > mx) mx = lower_error;
    l1_e[leaf_idx] = mx + max_run[leaf_idx];
  }

  /* ---- stats: two_layer.rs:267-287 ---- */
  {
    size_t m_idx = 0; uint64_t m_err = l1_e[0];
    for (size_t i = 0; i < (size_t)L; i++) if (l1_e[i] >= m_err) { m_err = l1_e[i]; m_idx = i; } /* max_by_key: last max */
    out->model_max_error = m_err; out->model_max_error_idx = m_idx;
    uint64_t s = 0;
    for (size_t i = 0; i < (size_t)L; i++) s += l1_n[i] * l1_e[i];
    out->model_avg_error = (double)s / (double)num_rows;
    double l2 = 0.0;
    for (size_t i = 0; i < (size_t)L; i++) { double v = (double)(l1_n[i] * l1_e[i]); l2 += (v * v) / (double)num_rows; }
    out->model_avg_l2_error = l2;
    double lg = 0.0;
    for (size_t i = 0; i < (size_t)L; i++) lg += (double)l1_n[i] * log2((double)(2 * l1_e[i] + 2));
    out->model_avg_log2_error = lg / (double)num_rows;
    out->model_max_log2_error = log2((double)m_err);
  }

  out->n = n; out->num_leaves = L; out->root = top; out->leaf_kind = leaf_kind;
  out->params_per_leaf = model_params_per(leaf_kind);
  for (size_t i = 0; i < (size_t)L; i++) {
    for (int p = 0; p < out->params_per_leaf; p++)
      out->leaf_params[i * (size_t)out->params_per_leaf + (size_t)p] = leaf_models[i].p[p];
    out->leaf_err[i] = l1_e[i];
    if (out->leaf_count) out->leaf_count[i] = l1_n[i];
  }

  free(leaf_models); free(first_key); free(last_key); free(max_run);
  free(next_idx); free(next_key); free(prev_idx); free(prev_key); free(l1_n); free(l1_e);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * The emitted lookup() (codegen.rs:612-718) and the reference tests' property
 * (tests/simple_model_wiki/main.cpp:26-41)
 * ---------------------------------------------------------------------------------------- */
static inline size_t fclamp(double inp, double bound) {  /* codegen.rs:615-618 */
  if (inp < 0.0) return 0;
  return (inp > bound ? (size_t)bound : (size_t)inp);
}

static uint64_t emitted_lookup(const orc_trained_rmi* r, int dtype, okey key, uint64_t* err) {
  size_t model_index;
  const orc_model* top = &r->root;
  /* key C type: uint64_t for u64/u32 files, double for f64 (main.rs:122-132) */
  if (top->kind == ORC_MODEL_RADIX || top->kind == ORC_MODEL_BRADIX || is_radix_table(top->kind)) {
    uint64_t ipred = model_predict_int_k(top, dtype, key);
    model_index = (size_t)ipred;                                   /* no bounds check: radix.rs:72-74 */
  } else {
    double fpred = model_predict_float_k(top, dtype, key);
    if (model_needs_bounds_check(top->kind)) model_index = fclamp(fpred, (double)r->num_leaves - 1.0);
    else model_index = (size_t)(uint64_t)fpred;                    /* "(uint64_t) fpred" */
  }
  orc_model leaf; leaf.kind = r->leaf_kind;
  for (int p = 0; p < r->params_per_leaf; p++) leaf.p[p] = r->leaf_params[model_index * (size_t)r->params_per_leaf + (size_t)p];
  double fpred = model_predict_float_k(&leaf, dtype, key);
  *err = r->leaf_err[model_index];
  return (uint64_t)fclamp(fpred, (double)r->n - 1.0);               /* always bounds-checked: codegen.rs:713-717 */
}

uint64_t orc_check_lookup_property(const orc_trained_rmi* rmi, int dtype, const void* keys,
                                   uint64_t n, uint64_t* first_bad) {
  uint64_t bad = 0; if (first_bad) *first_bad = n;
  uint64_t lb = 0;
  for (uint64_t i = 0; i < n; i++) {
    okey k = raw_key_at(dtype, keys, i);
    if (i > 0 && !key_eq(dtype, k, raw_key_at(dtype, keys, i - 1))) lb = i;   /* std::lower_bound index */
    uint64_t err; uint64_t guess = emitted_lookup(rmi, dtype, k, &err);
    uint64_t diff = guess > lb ? guess - lb : lb - guess;
    if (diff > err) { if (!bad && first_bad) *first_bad = i; bad++; }
  }
  return bad;
}

/* ------------------------------------------------------------------------------------------
 * cache_fix.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t from_x, from_y, to_x, to_y; } cf_spline;

static inline uint64_t cf_predict(const cf_spline* s, uint64_t inp) {      /* cache_fix.rs:37-43 */
  double v0 = (double)s->from_y, v1 = (double)s->to_y;
  double t = ((double)(inp - s->from_x)) / (double)(s->to_x - s->from_x);
  return sat_f64_to_u64(fma(1.0 - t, v0, t * v1));                          /* `as usize` */
}

typedef struct {
  int has; cf_spline sp;
  uint64_t* px; uint64_t* py; size_t np, cap;      /* curr_pts */
  uint64_t line;
} cf_fit;

static int cf_push_pt(cf_fit* f, uint64_t x, uint64_t y) {
  if (f->np == f->cap) {
    size_t nc = f->cap ? f->cap * 2 : 64;
    uint64_t* nx = (uint64_t*)realloc(f->px, nc * 8); uint64_t* ny = (uint64_t*)realloc(f->py, nc * 8);
    if (!nx || !ny) return -1;
    f->px = nx; f->py = ny; f->cap = nc;
  }
  f->px[f->np] = x; f->py[f->np] = y; f->np++;
  return 0;
}

/* SplineFit::add_point (cache_fix.rs:60-86): 1 = emitted a point into (*ox,*oy), 0 = none, <0 = assert */
static int cf_add_point(cf_fit* f, uint64_t x, uint64_t y, uint64_t* ox, uint64_t* oy) {
  if (!f->has) { f->has = 1; f->sp.from_x = f->sp.to_x = x; f->sp.from_y = f->sp.to_y = y; *ox = x; *oy = y; return 1; }
  cf_spline last = f->sp;
  if (!(x >= last.from_x) || !(y >= last.from_y)) return ORC_ERR_BAD_ARG;   /* with_new_dest asserts */
  cf_spline prop = last; prop.to_x = x; prop.to_y = y;
  if (cf_push_pt(f, last.to_x, last.to_y)) return ORC_ERR_BAD_ARG;
  int ok = 1;
  for (size_t i = 0; i < f->np; i++) {                                      /* check_spline :95-102 */
    if (cf_predict(&prop, f->px[i]) / f->line != f->py[i] / f->line) { ok = 0; break; }
  }
  if (ok) { f->sp = prop; return 0; }
  if (!(x > last.to_x)) return ORC_ERR_BAD_ARG;                              /* assert :77 */
  if (!(last.to_x <= x) || !(last.to_y <= y)) return ORC_ERR_BAD_ARG;        /* Spline::from asserts */
  f->sp.from_x = last.to_x; f->sp.from_y = last.to_y; f->sp.to_x = x; f->sp.to_y = y;
  f->np = 0;
  if (cf_push_pt(f, x, y)) return ORC_ERR_BAD_ARG;
  *ox = last.to_x; *oy = last.to_y;
  return 1;
}

int orc_cache_fix(const uint64_t* keys, uint64_t n, uint64_t line_size, uint64_t** pairs_out, uint64_t* count_out) {
  if (!keys || !pairs_out || !count_out || line_size == 0) return ORC_ERR_BAD_ARG;
  if (!(n > line_size)) return ORC_ERR_BAD_ARG;                              /* assert :110 */
  cf_fit f; memset(&f, 0, sizeof f); f.line = line_size;
  size_t cap = 1024, cnt = 0;
  uint64_t* out = (uint64_t*)malloc(cap * 16);
  if (!out) return ORC_ERR_BAD_ARG;
#define CF_EMIT(X, Y) do { if (cnt == cap) { cap *= 2; uint64_t* t_ = (uint64_t*)realloc(out, cap * 16); if (!t_) { free(out); free(f.px); free(f.py); return ORC_ERR_BAD_ARG; } out = t_; } \
                           out[2 * cnt] = (X); out[2 * cnt + 1] = (Y); cnt++; } while (0)
  uint64_t last_key = 0;
  int rc = 0;
  for (uint64_t i = 0; i < n; i++) {                                         /* iter_unique(): first occurrences */
    if (i > 0 && keys[i] == keys[i - 1]) continue;
    const uint64_t key = keys[i], off = i;
    if (key == 0 || !(key - 1 >= last_key)) { rc = ORC_ERR_BAD_ARG; break; } /* minus_epsilon / assert :122 */
    uint64_t ox, oy;
    if (key - 1 != last_key) {
      rc = cf_add_point(&f, key - 1, off, &ox, &oy);
      if (rc < 0) break;
      if (rc == 1) CF_EMIT(ox, oy);
    }
    rc = cf_add_point(&f, key, off, &ox, &oy);
    if (rc < 0) break;
    if (rc == 1) CF_EMIT(ox, oy);
    rc = 0;
    last_key = key;
  }
  if (rc == 0 && f.has) CF_EMIT(f.sp.to_x, f.sp.to_y);                       /* finish() :89-91 */
#undef CF_EMIT
  free(f.px); free(f.py);
  if (rc < 0) { free(out); return rc; }
  *pairs_out = out; *count_out = cnt;
  return ORC_OK;
}
void orc_free(void* p) { free(p); }

/* generate_cache_fix_code: codegen.rs:396-447 */
static uint64_t emitted_bounded_lookup(const orc_trained_rmi* r, const uint64_t* sp, uint64_t num_spline_pts,
                                       uint64_t line_size, uint64_t total_keys, uint64_t key, int* ub) {
  uint64_t esearch; okey k; k.bits = key;
  uint64_t start = emitted_lookup(r, ORC_KEY_U64, k, &esearch);
  uint64_t upper = (start + esearch > num_spline_pts) ? num_spline_pts : start + esearch;
  uint64_t lower = (esearch > start) ? 0 : start - esearch;
  uint64_t lo = lower, hi = upper;                                           /* std::lower_bound on .key */
  while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if (sp[2 * mid] < key) lo = mid + 1; else hi = mid; }
  uint64_t res = lo;
  if (res == num_spline_pts) return total_keys - 1;
  if (res == 0) { *ub = 1; return 0; }                                       /* *(res - 1) before begin: UB in the emitted code */
  uint64_t k1 = sp[2 * (res - 1)], v1i = sp[2 * (res - 1) + 1], k2 = sp[2 * res], v2i = sp[2 * res + 1];
  double v0 = (double)v1i, v1 = (double)v2i;
  double t = ((double)(key - k1)) / (double)(k2 - k1);
  return (sat_f64_to_u64(fma(1.0 - t, v0, t * v1)) / line_size) * line_size;
}

uint64_t orc_check_bounded_property(const orc_trained_rmi* rmi, const uint64_t* spline_pairs, uint64_t num_spline,
                                    uint64_t line_size, const uint64_t* keys, uint64_t n, uint64_t* first_bad) {
  uint64_t bad = 0; if (first_bad) *first_bad = n;
  uint64_t lb = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (i > 0 && keys[i] != keys[i - 1]) lb = i;
    int ub = 0;
    uint64_t guess = emitted_bounded_lookup(rmi, spline_pairs, num_spline, line_size, n, keys[i], &ub);
    uint64_t diff = guess > lb ? guess - lb : lb - guess;
    if (ub || diff > line_size) { if (!bad && first_bad) *first_bad = i; bad++; }
  }
  return bad;
}

