// sigma_model.c -- CPU model of the arithmetic of the one-pass sufficient-statistics leaf fit
// (rmi_sigma.hip.h), used to calibrate the guard bound against the oracle's exact coefficients.
// Development tool, not part of the product.  gcc -O2 -ffp-contract=off -shared -fPIC.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double u64_to_f64(uint64_t k) { return (double)k; }

// Per-leaf fit from shifted sums over the container [lo, hi] plus the tail duplicate of hi
// (regular leaves: lo = s-1, hi = e), pivot = x of the first key of the tile that holds e.
// Sums are accumulated in rows of 16 keys relative to the pivot, rows added in order.
// out: params[j*2+{0,1}], info[j*5+{0..4}] = (n, X=max|x|, W=range, sigma_x, sxx/m2)
int sigma_fit(const uint64_t* keys, uint64_t n, const uint64_t* ls, uint64_t L, uint64_t tile,
              uint64_t split_idx, double* params, double* info, uint8_t* regular) {
  for (uint64_t j = 0; j < L; j++) {
    uint64_t s = ls[j], e = ls[j + 1];
    regular[j] = 0;
    if (s >= e) continue;
    if (s == 0 || e >= n) continue;
    if (s == split_idx || e == split_idx) continue;
    uint64_t lo = s - 1, hi = e;
    // duplicates inside the container -> not regular (y != index)
    int dup = 0;
    for (uint64_t i = lo + 1; i <= hi; i++) if (keys[i] == keys[i - 1]) { dup = 1; break; }
    if (lo > 0 && keys[lo] == keys[lo - 1]) dup = 1;
    if (dup) continue;
    regular[j] = 1;
    uint64_t tb = (e / tile) * tile;
    double p = u64_to_f64(keys[tb]);
    double q = (double)tb;
    double sx = 0, sxx = 0, sxy = 0, sy = 0, cnt = 0;
    // rows of 16 aligned to global index
    uint64_t i = lo;
    while (i <= hi) {
      uint64_t rend = ((i / 16) + 1) * 16;
      if (rend > hi + 1) rend = hi + 1;
      double rx = 0, rxx = 0, rxy = 0;
      for (uint64_t k = i; k < rend; k++) {
        double dx = u64_to_f64(keys[k]) - p;
        double dy = (double)k - q;
        rx += dx; rxx = fma(dx, dx, rxx); rxy = fma(dx, dy, rxy);
        sy += dy; cnt += 1;
      }
      sx += rx; sxx += rxx; sxy += rxy;
      i = rend;
    }
    { double dx = u64_to_f64(keys[hi]) - p, dy = (double)hi - q; sx += dx; sxx = fma(dx, dx, sxx); sxy = fma(dx, dy, sxy); sy += dy; cnt += 1; }
    double mx = sx / cnt, my = sy / cnt;
    double m2 = sxx - sx * mx;
    double cxy = sxy - sx * my;
    double beta = cxy / m2;
    double alpha = (q + my) - beta * (p + mx);
    params[j * 2] = alpha; params[j * 2 + 1] = beta;
    double X = fabs(u64_to_f64(keys[hi]));
    double W = u64_to_f64(keys[hi]) - u64_to_f64(keys[lo]);
    info[j * 5 + 0] = cnt; info[j * 5 + 1] = X; info[j * 5 + 2] = W; info[j * 5 + 3] = sqrt(m2 / cnt); info[j * 5 + 4] = sxx / m2;
  }
  return 0;
}

static inline uint64_t pred_int(double a, double b, double x, uint64_t n) {
  double f = fma(b, x, a);
  f = floor(f); if (!(f > 0)) f = 0;
  uint64_t p = f >= 18446744073709551616.0 ? ~0ull : (uint64_t)f;
  return p < n ? p : n;
}

// For every regular leaf: maximum |f_ref - f_sig| over own keys, the closest distance of f_sig to an
// integer (own keys), and the own-key max error with both parameter sets.
int sigma_eval(const uint64_t* keys, uint64_t n, const uint64_t* ls, uint64_t L, const double* pref, const double* psig,
               const uint8_t* regular, double* disc, double* closest, uint64_t* err_ref, uint64_t* err_sig) {
  for (uint64_t j = 0; j < L; j++) {
    disc[j] = 0; closest[j] = 1; err_ref[j] = 0; err_sig[j] = 0;
    if (!regular[j]) continue;
    uint64_t s = ls[j], e = ls[j + 1];
    double d = 0, c = 1;
    uint64_t er = 0, es = 0;
    for (uint64_t i = s; i < e; i++) {
      double x = u64_to_f64(keys[i]);
      double fr = fma(pref[j * 2 + 1], x, pref[j * 2]);
      double fs = fma(psig[j * 2 + 1], x, psig[j * 2]);
      double dd = fabs(fr - fs); if (dd > d) d = dd;
      double fl = fs - floor(fs); double cc = fl < 0.5 ? fl : 1 - fl; if (cc < c) c = cc;
      uint64_t pr = pred_int(pref[j * 2], pref[j * 2 + 1], x, n), ps = pred_int(psig[j * 2], psig[j * 2 + 1], x, n);
      uint64_t a = pr > i ? pr - i : i - pr, b = ps > i ? ps - i : i - ps;
      if (a > er) er = a; if (b > es) es = b;
    }
    // widening keys
    {
      double xs[2] = { u64_to_f64(keys[e] - 1), u64_to_f64(keys[s - 1] + 1) };
      for (int k = 0; k < 2; k++) {
        double fr = fma(pref[j * 2 + 1], xs[k], pref[j * 2]);
        double fs = fma(psig[j * 2 + 1], xs[k], psig[j * 2]);
        double dd = fabs(fr - fs); if (dd > d) d = dd;
        double fl = fs - floor(fs); double cc = fl < 0.5 ? fl : 1 - fl; if (cc < c) c = cc;
      }
    }
    disc[j] = d; closest[j] = c; err_ref[j] = er; err_sig[j] = es;
  }
  return 0;
}
