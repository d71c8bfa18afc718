// rmi_kernels.hip.h -- gfx950 kernels of the two-layer leaf path (pipeline "v1": one kernel per
// reference pass; see DESIGN.md for the roofline of each and for the fused successors).
//
//   k_bounds_vec   bucketing scan: target(key) for every key, leaf boundary table, split point,
//                  monotonicity / bounds checks            (two_layer.rs:43-50, 130-156)
//   k_fill_*       suffix-min fill of leaf_start for empty leaves
//   k_fit_leaf     per-leaf fit on the reference's container C_j (two_layer.rs:52-90) in
//                  reference order: Welford SLR (linear.rs:12-59) / endpoints
//                  (linear_spline.rs:13-35)
//   (round 6: k_boundaries and k_err, the one-thread-per-key kernels of round 1 -- RMI_HIP_PIPELINE=1 -- are gone; k_bounds_vec and
//    k_err_range, rmi_stream.hip.h, do their work)
//   k_finalize     empty-leaf fix, lower-bound widening, row packing
//                  (two_layer.rs:185-197, 226-259; codegen.rs:288-315)
//   k_stats_reduce aggregates from k_finalize's block records (two_layer.rs:267-287)
#pragma once
#include "rmi_device.hip.h"

namespace rmi {

constexpr int WAVE = 64;
constexpr unsigned long long NO_START = ~0ull;

// ---------------------------------------------------------------------------------------------
// FAST (not bit-identical) root fit of `linear` / `robust_linear`: the same least-squares line over
// the same points (x = f64(key), y = scaled FixDups offset, items [lo, hi) of iter() plus, for the
// full range, the Q1 tail duplicate), from parallel sums instead of the reference's sequential
// recurrence -- the coefficients agree to ~1e-12 relative, a handful of keys next to leaf
// boundaries may change bucket.  Opt-in (rmi_hip_fit_root_fast); the exact host fit
// stays the default.  x is taken relative to the first key to keep the squares well conditioned.
// ---------------------------------------------------------------------------------------------
constexpr int RF_BLOCKS = 1024;
struct RootPartial { double n, sx, sy, sxx, sxy; };

template <typename K>
__global__ void __launch_bounds__(256) k_root_sums(const K* __restrict__ keys, uint64_t n, uint64_t lo, uint64_t hi,
                                                   double scale, int scaled, int tail_dup, RootPartial* __restrict__ partials) {
  __shared__ double sm[5][256];
  const double x0 = KeyTraits<K>::as_float(keys[0]);
  double c = 0.0, sx = 0.0, sy = 0.0, sxx = 0.0, sxy = 0.0;
  auto add = [&](uint64_t i) {
    const uint64_t f = first_occurrence(keys, i);
    const double y = (double)(scaled ? sat_f64_to_u64((double)f * scale) : f);
    const double x = KeyTraits<K>::as_float(keys[i]) - x0;
    c += 1.0; sx += x; sy += y; sxx += x * x; sxy += x * y;
  };
  for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * blockDim.x) add(i);
  if (tail_dup && blockIdx.x == 0 && threadIdx.x == 0) add(n - 1);          // models/mod.rs:180
  sm[0][threadIdx.x] = c; sm[1][threadIdx.x] = sx; sm[2][threadIdx.x] = sy; sm[3][threadIdx.x] = sxx; sm[4][threadIdx.x] = sxy;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int q = 0; q < 5; q++) sm[q][threadIdx.x] += sm[q][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = RootPartial{sm[0][0], sm[1][0], sm[2][0], sm[3][0], sm[4][0]};
}

// Cubic root: the comparison pass of CubicSplineModel::new (cubic_spline.rs:117-131), sum |cubic - y|
// and sum |line - y| over iter(), as a parallel reduction.  The sums only DECIDE between the two
// candidate models (whose coefficients come from O(1) keys, exactly); the host accepts the decision
// when the sums differ by far more than any summation order can move them, and otherwise repeats the
// reference's sequential pass -- so the fitted root is always the reference's, bit for bit.
struct CubicPartial { double our_err, lin_err; };
template <typename K>
__global__ void __launch_bounds__(256) k_cubic_root_sums(const K* __restrict__ keys, uint64_t n, double scale, int scaled,
                                                         double a, double b, double c, double d, double la, double lb,
                                                         CubicPartial* __restrict__ partials) {
  __shared__ double sm[2][256];
  double eo = 0.0, el = 0.0;
  auto add = [&](uint64_t i) {
    const uint64_t f = first_occurrence(keys, i);
    const double y = (double)(scaled ? sat_f64_to_u64((double)f * scale) : f);
    const double x = KeyTraits<K>::as_float(keys[i]);
    eo += fabs(__builtin_fma(__builtin_fma(__builtin_fma(a, x, b), x, c), x, d) - y);
    el += fabs(__builtin_fma(lb, x, la) - y);
  };
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) add(i);
  if (blockIdx.x == 0 && threadIdx.x == 0) add(n - 1);                       // Q1 tail duplicate
  sm[0][threadIdx.x] = eo; sm[1][threadIdx.x] = el;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = CubicPartial{sm[0][0], sm[1][0]};
}

// ---------------------------------------------------------------------------------------------
// Radix-table root on the device (RadixTable::new, radix.rs:90-121), integer work, exact: the table
// slot of a key is the `radix` bucketing function with bits = table_bits, so "first key index of
// every slot" is the bucketing scan over 2^bits slots followed by the suffix-min fill (a gap takes
// the next present slot's value, radix.rs:104-107); the hint is that index scaled like every y
// (map_scale!, models/mod.rs:238-250); slots after the last present one hold table.len() (:112-114).
// The first key of a slot differs from its predecessor, so its FixDups offset is its own index.
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_table_init(unsigned long long* __restrict__ first_idx, uint64_t slots) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= slots; j += stride) first_idx[j] = NO_START;
}
static __global__ void __launch_bounds__(256) k_table_from_starts(const unsigned long long* __restrict__ first_idx, uint64_t slots,
                                                           double scale, int scaled, unsigned int* __restrict__ table) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < slots; j += stride) {
    const unsigned long long i = first_idx[j];
    unsigned long long y = (unsigned long long)slots;                      // tail: hint_table.len()
    if (i != NO_START) y = scaled ? sat_f64_to_u64((double)i * scale) : i;
    table[j] = (unsigned int)y;                                            // `y as u32`
  }
}

// ---------------------------------------------------------------------------------------------
// k_init: one launch instead of four memsets and two small copies: leaf_start := "no start" with
// the sentinel entry leaf_start[L_own] = it_hi, maxerr := run := 0, device state := initial state.
// ---------------------------------------------------------------------------------------------
// `arrays` = false (the leaf-lane pipeline with its search and its fused error pass: every entry of the three arrays is
// written by a plain store later on): only the sentinel, the list counters and the state -- a launch of one block.
static __global__ void __launch_bounds__(256) k_init(unsigned long long* __restrict__ leaf_start,
                                              unsigned long long* __restrict__ maxerr,
                                              unsigned long long* __restrict__ run, uint64_t L_own,
                                              unsigned long long sentinel, DevState* __restrict__ st, DevState init,
                                              unsigned long long* __restrict__ list_cnt, int n_list_cnt, bool arrays,
                                              unsigned int* __restrict__ tickets = nullptr) {   // (3 arrival / list counters, where the caller has no memset for them)
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t lim = (arrays && L_own + 1 > (uint64_t)n_list_cnt) ? L_own + 1 : (uint64_t)n_list_cnt;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < lim; j += stride) {
    if (arrays && j <= L_own) leaf_start[j] = (j == L_own) ? sentinel : NO_START;
    if (arrays && j < L_own) { maxerr[j] = 0; run[j] = 0; }
    if (j < (uint64_t)n_list_cnt) list_cnt[j] = 0ull;      // the one-pass mode's list counters (a memset of their own costs ~4 us)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { *st = init; if (!arrays) leaf_start[L_own] = sentinel; if (tickets) { tickets[0] = 0u; tickets[1] = 0u; tickets[2] = 0u; } }
}

// ---------------------------------------------------------------------------------------------
// k_bounds_vec: the same bucketing scan with 16 bytes per lane and load (2 x 8-byte or 4 x 4-byte
// keys), fully coalesced, BV_UNROLL independent loads in flight per lane.  The target of the key
// before a lane's first key comes from the previous lane (DPP-free wave shuffle of its last
// target) or, for lane 0 of a wave, from one extra load.  One root evaluation per key.
// ---------------------------------------------------------------------------------------------
constexpr int BV_UNROLL = 4;

template <int ROOT, typename K>
__global__ void __launch_bounds__(256) k_bounds_vec(const K* __restrict__ keys, Span sp, RootP r,
                                                    unsigned long long* __restrict__ leaf_start,
                                                    DevState* __restrict__ st) {
  constexpr int V = 16 / sizeof(K);
  const uint64_t n = sp.n;
  const uint64_t Lm1 = root_cap<ROOT>(r);
  const uint64_t mid = r.L / 2;                                   // two_layer.rs:131
  const int lane = threadIdx.x & 63;
  const uint64_t first = sp.it_lo + ((uint64_t)blockIdx.x * BV_UNROLL * blockDim.x + threadIdx.x) * V;
  const bool aligned = (((uintptr_t)(keys + sp.it_lo)) & 15) == 0;
  K kk[BV_UNROLL][V];
#pragma unroll
  for (int u = 0; u < BV_UNROLL; u++) {                           // all loads first
    const uint64_t base = first + (uint64_t)u * blockDim.x * V;
    if (base + V <= sp.it_hi && aligned) {
      const uint4 raw = *reinterpret_cast<const uint4*>(keys + base);
      __builtin_memcpy(kk[u], &raw, 16);
    } else {
#pragma unroll
      for (int q = 0; q < V; q++) kk[u][q] = (base < sp.it_hi) ? keys[(base + q < sp.it_hi) ? base + q : sp.it_hi - 1] : K();
    }
  }
  unsigned int flags = 0;
#pragma unroll
  for (int u = 0; u < BV_UNROLL; u++) {
    const uint64_t base = first + (uint64_t)u * blockDim.x * V;
    const bool live = base < sp.it_hi;                            // (wave-uniform except in the last wave)
    uint64_t t[V];
#pragma unroll
    for (int q = 0; q < V; q++) {
      const uint64_t p = root_predict<ROOT, K>(r, kk[u][q]);
      if constexpr (!root_needs_bounds_check<ROOT>()) { if (live && base + q < sp.it_hi && p > root_oob_above<ROOT>(r)) flags |= EF_ROOT_OOB; }   // two_layer.rs:45-48
      t[q] = p < Lm1 ? p : Lm1;                                   // two_layer.rs:49
    }
    // target of the key before this lane's first key: the previous lane's last target; lane 0 loads it
    uint64_t tp = __shfl_up(t[V - 1], 1);
    if (lane == 0 || !live) {
      tp = ~0ull;
      if (live && base > sp.rd_lo) {
        const uint64_t pp = root_predict<ROOT, K>(r, keys[base - 1]);
        tp = pp < Lm1 ? pp : Lm1;
      }
    }
    if (!live) continue;
#pragma unroll
    for (int q = 0; q < V; q++) {
      const uint64_t i = base + q;
      if (i < sp.it_hi) {
        const uint64_t tq = t[q];
        const bool mine = tq >= sp.leaf_lo && tq < sp.leaf_hi;
        if (i == 0) {
          if (mine) leaf_start[tq] = 0;
          if (tq >= mid) flags |= EF_DEGENERATE_SPLIT;            // split_idx == 0 -> :27
        } else if (tp != ~0ull) {
          if (tq < tp) flags |= EF_NON_MONOTONE;                  // two_layer.rs:50 / :144
          else if (tq > tp) {
            if (mine) leaf_start[tq] = i;
            if (tp < mid && tq >= mid) {                          // two_layer.rs:132-136,152-156
              st->split_idx = i;
              st->split_target = tq;
              if (i + 1 >= n) flags |= EF_DEGENERATE_SPLIT;       // second half empty -> :27
            }
          }
        }
        if (i == n - 1) st->last_target = tq;
        tp = tq;
      }
    }
  }
  if (flags) atomicOr(&st->err_flags, flags);
}

// ---------------------------------------------------------------------------------------------
// Suffix-min fill: leaf_start[j] = min_{j' >= j} leaf_start[j'] with leaf_start[L] = n.
// Three small kernels over L+1 entries (tile = 2048 entries per block).
// ---------------------------------------------------------------------------------------------
constexpr int FILL_TILE = 2048;

static __global__ void __launch_bounds__(256) k_fill_tilemin(const unsigned long long* __restrict__ ls, uint64_t count,
                                                      unsigned long long* __restrict__ tile_min) {
  __shared__ unsigned long long sm[256];
  const uint64_t base = (uint64_t)blockIdx.x * FILL_TILE;
  unsigned long long m = NO_START;
  for (int k = threadIdx.x; k < FILL_TILE; k += 256) {
    uint64_t idx = base + k;
    if (idx < count) { unsigned long long v = ls[idx]; m = v < m ? v : m; }
  }
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { unsigned long long o = sm[threadIdx.x + s]; if (o < sm[threadIdx.x]) sm[threadIdx.x] = o; }
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_min[blockIdx.x] = sm[0];
}

// single block: exclusive suffix-min over the tile minima (carry-in for each tile)
static __global__ void __launch_bounds__(1024) k_fill_scan_tiles(unsigned long long* __restrict__ tile_min, uint64_t ntiles) {
  __shared__ unsigned long long sm[1024];
  // each thread owns a contiguous span of tiles, processed right-to-left
  const uint64_t per = (ntiles + 1023) / 1024;
  const uint64_t lo = (uint64_t)threadIdx.x * per;
  const uint64_t hi = lo + per < ntiles ? lo + per : ntiles;
  unsigned long long m = NO_START;
  for (uint64_t k = hi; k-- > lo;) { unsigned long long v = tile_min[k]; m = v < m ? v : m; }
  sm[threadIdx.x] = m;
  __syncthreads();
  // carry for thread t = min over threads > t: inclusive suffix-min by doubling, then shift by one
  for (int d = 1; d < 1024; d <<= 1) {
    unsigned long long o = ((int)threadIdx.x + d < 1024) ? sm[threadIdx.x + d] : NO_START;
    __syncthreads();
    if (o < sm[threadIdx.x]) sm[threadIdx.x] = o;
    __syncthreads();
  }
  unsigned long long carry = (threadIdx.x + 1 < 1024) ? sm[threadIdx.x + 1] : NO_START;
  for (uint64_t k = hi; k-- > lo;) {
    unsigned long long v = tile_min[k];
    tile_min[k] = carry;              // exclusive: min of all tiles to the right
    carry = v < carry ? v : carry;
  }
}

static __global__ void __launch_bounds__(256) k_fill_apply(unsigned long long* __restrict__ ls, uint64_t count,
                                                    const unsigned long long* __restrict__ tile_carry) {
  // tile of 2048 entries; thread t owns 8 consecutive entries, block-level exclusive suffix-min
  // across threads via LDS.
  __shared__ unsigned long long sm[256];
  const uint64_t base = (uint64_t)blockIdx.x * FILL_TILE + (uint64_t)threadIdx.x * 8;
  unsigned long long v[8];
  unsigned long long m = NO_START;
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    uint64_t idx = base + k;
    v[k] = idx < count ? ls[idx] : NO_START;
    m = v[k] < m ? v[k] : m;
  }
  sm[threadIdx.x] = m;
  __syncthreads();
  // inclusive suffix-min over the 256 thread minima by doubling, then shifted by one thread and
  // combined with the carry of the tiles to the right
  for (int d = 1; d < 256; d <<= 1) {
    const unsigned long long o = ((int)threadIdx.x + d < 256) ? sm[threadIdx.x + d] : NO_START;
    __syncthreads();
    if (o < sm[threadIdx.x]) sm[threadIdx.x] = o;
    __syncthreads();
  }
  unsigned long long carry = tile_carry[blockIdx.x];
  if (threadIdx.x + 1 < 256) { const unsigned long long o = sm[threadIdx.x + 1]; carry = o < carry ? o : carry; }
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    uint64_t idx = base + k;
    carry = v[k] < carry ? v[k] : carry;
    if (idx < count) ls[idx] = carry;
  }
}

// ---------------------------------------------------------------------------------------------
// Container bounds of leaf j (closed form of build_models_from, two_layer.rs:20-99, including
// the quirks Q2-Q4 of SURVEY.md section 8a).  Returns kind of container:
//   0 = empty model, 1 = single borrowed point at index `lo` (Q4), 2 = range [lo, hi] inclusive.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int leaf_container(uint64_t j, uint64_t s, uint64_t e, uint64_t n,
                                              uint64_t split_idx, uint64_t split_target,
                                              uint64_t& lo, uint64_t& hi) {
  uint64_t a, b, f;
  if (split_idx < n) {
    if (j < split_target) { a = 0; b = split_idx; f = 0; }          // first half  (:162-165)
    else { a = split_idx + 1; b = n; f = split_target; }            // second half (:166-169), Q2
  } else { a = 0; b = n; f = 0; }                                   // :147-150
  uint64_t own_lo = s > a ? s : a;
  uint64_t own_hi = e < b ? e : b;
  if (own_lo >= own_hi) {
    if (j == f) { lo = hi = a; return 1; }                          // Q4 (:52-63 on empty data)
    return 0;                                                       // :67-69 / :94-96
  }
  lo = own_lo > a ? own_lo - 1 : own_lo;                            // prev-last  (:74-78), Q3
  hi = own_hi < b ? own_hi : own_hi - 1;                            // next-first (:58-59), Q3
  return 2;
}

// ---------------------------------------------------------------------------------------------
// Cubic leaves (CubicSplineModel::new, cubic_spline.rs:108-136, on the container [lo, hi]).  The
// points a container stores carry FixDups first-occurrence offsets, for get() as well as for
// iter().  cube = pow(xmax - xmin, 3.0) comes from the host: the reference's value IS the
// platform libm's (`powf(3.0)`, cubic_spline.rs:76-93), see k_cubic_span.
// ---------------------------------------------------------------------------------------------
constexpr int CUBIC_LONG = 1024;      // containers longer than this: one wave per leaf (k_fit_cubic_long)

struct CubicFit {
  double a, b, c, d;                  // cubic()
  double slope, icept;                // the line through the end points (linear_spline.rs:13-35)
  double ymin;
  __device__ __forceinline__ double eval(double x) const {
    return __builtin_fma(__builtin_fma(__builtin_fma(a, x, b), x, c), x, d);
  }
  __device__ __forceinline__ void store(double* out, bool line_wins) const {      // :133-135
    if (line_wins) { out[0] = 0.0; out[1] = 0.0; out[2] = slope; out[3] = icept; }
    else { out[0] = a; out[1] = b; out[2] = c; out[3] = d; }
  }
};

// Everything before the comparison pass.  Returns false when the model is already decided (written
// to `out`): all keys equal, or the reference's `.unwrap()` on an empty search (error flag).
template <typename K>
__device__ __forceinline__ bool cubic_prolog(const K* __restrict__ keys, uint64_t lo, uint64_t hi, const Span& sp,
                                             DevState* __restrict__ st, double den, double* __restrict__ out, CubicFit& cf) {
  const K k0 = keys[lo], kl = keys[hi];
  const double y0 = (double)first_occurrence(keys, lo, sp.rd_lo);
  if (k0 == kl) { out[0] = 0.0; out[1] = 0.0; out[2] = 0.0; out[3] = y0; return false; }   // :28-36 (sorted: all equal)
  const double xmin = KeyTraits<K>::as_float(k0), ymin = y0;
  const double xmax = KeyTraits<K>::as_float(kl), ymax = (double)first_occurrence(keys, hi, sp.rd_lo);
  const double xr = xmax - xmin, yr = ymax - ymin;
  double m1, m2;
  {  // :46-54 first item of iter() whose scaled x is > 0
    bool found = false;
    uint64_t yn = 0; double xn = 0.0;
    uint64_t first = (uint64_t)y0;
    for (uint64_t i = lo; i <= hi; i++) {
      const K k = keys[i];
      if (i > lo && !(k == keys[i - 1])) first = i;
      const double x = KeyTraits<K>::as_float(k);
      if ((x - xmin) / xr > 0.0) { xn = x; yn = first; found = true; break; }
    }
    if (!found) { atomicOr(&st->err_flags, EF_CUBIC_DEGENERATE); out[0] = out[1] = out[2] = out[3] = 0.0; return false; }
    const double sxn = (xn - xmin) / xr, syn = ((double)yn - ymin) / yr;
    m1 = (syn - 0.0) / (sxn - 0.0);
  }
  {  // :56-65 last index (get) whose scaled x is < 1
    bool found = false;
    uint64_t ip = 0; double xp = 0.0;
    for (uint64_t i = hi + 1; i-- > lo;) {
      const double x = KeyTraits<K>::as_float(keys[i]);
      if ((x - xmin) / xr < 1.0) { xp = x; ip = i; found = true; break; }
    }
    if (!found) { atomicOr(&st->err_flags, EF_CUBIC_DEGENERATE); out[0] = out[1] = out[2] = out[3] = 0.0; return false; }
    const double yp = (double)first_occurrence(keys, ip, sp.rd_lo);
    const double sxp = (xp - xmin) / xr, syp = (yp - ymin) / yr;
    m2 = (1.0 - syp) / (1.0 - sxp);
  }
  if (m1 * m1 + m2 * m2 > 9.0) {                                     // :68-72
    const double tau = 3.0 / sqrt(m1 * m1 + m2 * m2);
    m1 *= tau; m2 *= tau;
  }
  double a = (m1 + m2 - 2.0) / den;                                                        // :76
  double b = -(xmax * (2.0 * m1 + m2 - 3.0) + xmin * (m1 + 2.0 * m2 - 3.0)) / den;         // :80-81
  double c = (m1 * (xmax * xmax) + m2 * (xmin * xmin) + xmax * xmin * (2.0 * m1 + 2.0 * m2 - 6.0)) / den;   // :86-88
  double d = -xmin * (m1 * (xmax * xmax) + xmax * xmin * (m2 - 3.0) + (xmin * xmin)) / den;   // :92-93
  a *= yr; b *= yr; c *= yr; d *= yr; d += ymin;                                            // :95-99
  cf.a = a; cf.b = b; cf.c = c; cf.d = d;
  cf.slope = (ymin - ymax) / (xmin - xmax);
  cf.icept = ymin - cf.slope * xmin;
  cf.ymin = ymin;
  return true;
}

// ---------------------------------------------------------------------------------------------
// k_fit_leaf: one lane per leaf, reference order.  (Exact mode: the SLR recurrence is order
// dependent, so each leaf is a sequential chain; parallelism is across leaves.)
// ---------------------------------------------------------------------------------------------
template <int LEAF, typename K>
__device__ __forceinline__ void fit_one_leaf(uint64_t j, const K* __restrict__ keys, const Span& sp,
                                             const unsigned long long* __restrict__ leaf_start,
                                             DevState* __restrict__ st, double* __restrict__ params,
                                             const double* __restrict__ cube = nullptr, bool robust = false) {
  const uint64_t n = sp.n;
  constexpr int PPL = (LEAF == K_CUBIC) ? 4 : 2;
  const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
  uint64_t lo, hi;
  const int ck = leaf_container(j, s, e, n, st->split_idx, st->split_target, lo, hi);
  double* out = params + j * PPL;
  if (ck == 0) {
    if constexpr (LEAF == K_CUBIC) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; out[3] = 0.0; }  // cubic_spline.rs:19-21
    else { out[0] = 0.0; out[1] = 0.0; }                                                         // linear.rs:37-39
    return;
  }
  if constexpr (LEAF == K_LINEAR) {
    if (robust) {
      // RobustLinearModel::new (linear.rs:239-260) on the container: drop max(1, 0.01 %) items at each
      // end of iter() -- the Q1 duplicate is never reached -- and run slr over the rest.
      const uint64_t len = hi - lo + 1;
      uint64_t bnd = (uint64_t)((double)len * 0.0001);
      if (bnd < 1) bnd = 1;
      if (!(bnd * 2 + 1 < len)) {                                      // assert!, linear.rs:248
        atomicOr(&st->err_flags, EF_ROBUST_TOO_SMALL);
        out[0] = 0.0; out[1] = 0.0;
        return;
      }
      const uint64_t a = lo + bnd, b = hi - bnd;                       // items [bnd, len - bnd) of the container
      double mean_x = 0.0, mean_y = 0.0, c = 0.0, m2 = 0.0;
      uint64_t cnt = 0;
      uint64_t y = first_occurrence(keys, a, sp.rd_lo);
      for (uint64_t i = a; i <= b; i++) {
        const K k = keys[i];
        if (i > a && !(k == keys[i - 1])) y = i;
        const double x = KeyTraits<K>::as_float(k), yf = (double)y;
        cnt += 1;
        const double nf = (double)cnt;
        const double dx = x - mean_x;
        mean_x += dx / nf;
        mean_y += (yf - mean_y) / nf;
        c += dx * (yf - mean_y);
        m2 += dx * (x - mean_x);
      }
      if (cnt == 1) { out[0] = mean_y; out[1] = 0.0; return; }         // linear.rs:41-43
      const double cov = c / (double)(cnt - 1);
      const double var = m2 / (double)(cnt - 1);
      if (!(var >= 0.0)) atomicOr(&st->err_flags, EF_NEG_VARIANCE);
      if (var == 0.0) { out[0] = mean_y; out[1] = 0.0; return; }
      const double beta = cov / var;
      out[0] = mean_y - beta * mean_x; out[1] = beta;
      return;
    }
  }
  if (ck == 1) {
    // single point (key[lo], y = lo): its key differs from key[lo-1] (different leaf) so y == lo
    const double y = (double)lo;
    if constexpr (LEAF == K_CUBIC) { out[0] = 0.0; out[1] = 0.0; out[2] = 0.0; out[3] = y; }    // cubic_spline.rs:23-25
    else { out[0] = y; out[1] = 0.0; }   // linear.rs:50-53 (two identical items) / linear_spline.rs:18-20
    return;
  }
  if constexpr (LEAF == K_LINEAR) {
    // slr: linear.rs:12-59 over C_j.iter(): FixDups offsets + the tail duplicate (Q1)
    double mean_x = 0.0, mean_y = 0.0, c = 0.0, m2 = 0.0;
    uint64_t cnt = 0;
    uint64_t y = first_occurrence(keys, lo, sp.rd_lo);
    K prev = keys[lo];
    double x = 0.0, yf = 0.0;
    for (uint64_t i = lo; i <= hi; i++) {
      const K k = keys[i];
      if (i > lo && !(k == prev)) y = i;
      prev = k;
      x = KeyTraits<K>::as_float(k);
      yf = (double)y;
      cnt += 1;
      const double nf = (double)cnt;
      const double dx = x - mean_x;
      mean_x += dx / nf;
      mean_y += (yf - mean_y) / nf;
      c += dx * (yf - mean_y);
      const double dx2 = x - mean_x;
      m2 += dx * dx2;
    }
    {  // Q1: last item once more (models/mod.rs:180)
      cnt += 1;
      const double nf = (double)cnt;
      const double dx = x - mean_x;
      mean_x += dx / nf;
      mean_y += (yf - mean_y) / nf;
      c += dx * (yf - mean_y);
      const double dx2 = x - mean_x;
      m2 += dx * dx2;
    }
    const double cov = c / (double)(cnt - 1);
    const double var = m2 / (double)(cnt - 1);
    if (!(var >= 0.0)) atomicOr(&st->err_flags, EF_NEG_VARIANCE);   // linear.rs:48
    if (var == 0.0) { out[0] = mean_y; out[1] = 0.0; return; }       // linear.rs:50-53
    const double beta = cov / var;
    const double alpha = mean_y - beta * mean_x;                     // no fma: linear.rs:56
    out[0] = alpha; out[1] = beta;
  } else if constexpr (LEAF == K_LINEAR_SPLINE) {
    // linear_splines: linear_spline.rs:13-35 on get(0), get(len-1) of the container
    const K k0 = keys[lo], k1 = keys[hi];
    const double y0 = (double)first_occurrence(keys, lo, sp.rd_lo);
    if (lo == hi || k0 == k1) { out[0] = y0; out[1] = 0.0; return; }
    const double y1 = (double)first_occurrence(keys, hi, sp.rd_lo);
    const double x0 = KeyTraits<K>::as_float(k0), x1 = KeyTraits<K>::as_float(k1);
    const double slope = (y0 - y1) / (x0 - x1);
    const double intercept = y0 - slope * x0;                        // plain mul+sub
    out[0] = intercept; out[1] = slope;
  } else if constexpr (LEAF == K_CUBIC) {
    if (hi - lo + 1 > (uint64_t)CUBIC_LONG) return;          // k_fit_cubic_long's
    CubicFit cf;
    if (!cubic_prolog<K>(keys, lo, hi, sp, st, cube[j], out, cf)) return;
    // the comparison pass over iter() (cubic_spline.rs:117-131), in iteration order
    double our_error = 0.0, lin_error = 0.0;
    uint64_t y = (uint64_t)cf.ymin;
    double x = 0.0, yf = 0.0;
    for (uint64_t i = lo; i <= hi; i++) {
      const K k = keys[i];
      if (i > lo && !(k == keys[i - 1])) y = i;
      x = KeyTraits<K>::as_float(k);
      yf = (double)y;
      our_error += fabs(cf.eval(x) - yf);
      lin_error += fabs(__builtin_fma(cf.slope, x, cf.icept) - yf);
    }
    our_error += fabs(cf.eval(x) - yf);                        // Q1: tail duplicate
    lin_error += fabs(__builtin_fma(cf.slope, x, cf.icept) - yf);
    cf.store(out, lin_error < our_error);
  }
}

// Cubic leaves: xmax - xmin of every container, for the host's pow(., 3.0) (0 where no cube is needed).
template <typename K>
__global__ void __launch_bounds__(256) k_cubic_span(const K* __restrict__ keys, Span sp,
                                                    const unsigned long long* __restrict__ leaf_start,
                                                    const DevState* __restrict__ st, double* __restrict__ span) {
  const uint64_t j = sp.leaf_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= sp.leaf_hi) return;
  uint64_t lo, hi;
  const int ck = leaf_container(j, leaf_start[j], leaf_start[j + 1], sp.n, st->split_idx, st->split_target, lo, hi);
  span[j] = (ck == 2) ? KeyTraits<K>::as_float(keys[hi]) - KeyTraits<K>::as_float(keys[lo]) : 0.0;
}

// Long cubic leaves, one wave per leaf: the two error sums of the comparison pass are sequential
// (floating-point sums in iteration order), but their terms are not -- 64 keys at a time the lanes
// compute |cubic(x) - y| and |line(x) - y| (y: first-occurrence index by ballot) into LDS, then
// every lane adds the 64 pairs in order (broadcast reads).
template <typename K>
__global__ void __launch_bounds__(64) k_fit_cubic_long(const K* __restrict__ keys, Span sp,
                                                       const unsigned long long* __restrict__ leaf_start,
                                                       DevState* __restrict__ st, double* __restrict__ params,
                                                       const double* __restrict__ cube) {
  __shared__ double s_e[2][64][2];
  const int lane = threadIdx.x;
  for (uint64_t j = sp.leaf_lo + blockIdx.x; j < sp.leaf_hi; j += gridDim.x) {
    uint64_t lo, hi;
    const int ck = leaf_container(j, leaf_start[j], leaf_start[j + 1], sp.n, st->split_idx, st->split_target, lo, hi);
    if (ck != 2 || hi - lo + 1 <= (uint64_t)CUBIC_LONG) continue;
    double* out = params + j * 4;
    CubicFit cf;
    double scratch[4];
    if (!cubic_prolog<K>(keys, lo, hi, sp, st, cube[j], lane == 0 ? out : scratch, cf)) continue;
    double our_error = 0.0, lin_error = 0.0;
    double carry_y = cf.ymin, last_c = 0.0, last_l = 0.0;
    int b = 0;
    for (uint64_t base = lo; base <= hi; base += 64, b ^= 1) {
      uint64_t i = base + lane;
      const bool valid = i <= hi;
      i = valid ? i : hi;
      const K k = keys[i];
      const K kp = keys[i > lo ? i - 1 : i];
      const bool newrun = valid && i > lo && !(k == kp);
      const unsigned long long mk = __ballot(newrun);
      const unsigned long long below = mk & ((2ull << lane) - 1ull);
      const double y = below ? (double)(base + (uint64_t)(63 - __builtin_clzll(below))) : carry_y;
      if (mk) carry_y = (double)(base + (uint64_t)(63 - __builtin_clzll(mk)));
      const double x = KeyTraits<K>::as_float(k);
      s_e[b][lane][0] = fabs(cf.eval(x) - y);
      s_e[b][lane][1] = fabs(__builtin_fma(cf.slope, x, cf.icept) - y);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int steps = (hi - base + 1 < 64ull) ? (int)(hi - base + 1) : 64;
      if (steps == 64) {
#pragma unroll 16
        for (int u = 0; u < 64; u++) { our_error += s_e[b][u][0]; lin_error += s_e[b][u][1]; }
      } else {
        for (int u = 0; u < steps; u++) { our_error += s_e[b][u][0]; lin_error += s_e[b][u][1]; }
      }
      last_c = s_e[b][steps - 1][0];
      last_l = s_e[b][steps - 1][1];
    }
    our_error += last_c;                                       // Q1: tail duplicate
    lin_error += last_l;
    if (lane == 0) cf.store(out, lin_error < our_error);
  }
}

template <int LEAF, typename K>
__global__ void __launch_bounds__(256) k_fit_leaf(const K* __restrict__ keys, Span sp,
                                                  const unsigned long long* __restrict__ leaf_start,
                                                  DevState* __restrict__ st,
                                                  double* __restrict__ params, const double* __restrict__ cube,
                                                  bool robust) {
  const uint64_t j = sp.leaf_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= sp.leaf_hi) return;
  fit_one_leaf<LEAF, K>(j, keys, sp, leaf_start, st, params, cube, robust);
}


// ---------------------------------------------------------------------------------------------
// k_err: one thread per key: err = |min(pred,N) - min(y,N)| with the leaf's model
// (two_layer.rs:207-217) and run lengths of equal keys (lower_bound_correction.rs:104-119,
// incl. Q5: the globally last run is never recorded).  Keys of one leaf are contiguous, so a
// wave first reduces per leaf segment with shuffles and issues one atomic per segment.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d) {
  unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
  lo = __shfl_up(lo, d, WAVE); hi = __shfl_up(hi, d, WAVE);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_down_u64(unsigned long long v, int d) {
  unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
  lo = __shfl_down(lo, d, WAVE); hi = __shfl_down(hi, d, WAVE);
  return ((unsigned long long)hi << 32) | lo;
}


// ---------------------------------------------------------------------------------------------
// Aggregates of two_layer.rs:267-287.  Every block of k_finalize reduces its leaves to one partial
// record; k_stats_reduce (a single block) then combines the records in a fixed order (deterministic sums,
// no same-address atomics: those serialise at ~26 ns each).  The f64 sums are a tree reduction, not
// the reference's sequential sum: equal within 1e-12 relative, which is what the tests ask.
// (max_error, max_error_idx) is the lexicographic maximum: `max_by_key` keeps the LAST maximum.
// ---------------------------------------------------------------------------------------------
struct StatsPartial { unsigned long long mx, mi, sum; double l2, lg; };

template <int THREADS = 256>
__device__ __forceinline__ void stats_block_reduce(unsigned long long& mx, unsigned long long& mi, unsigned long long& sm,
                                                   double& l2, double& lg) {
  __shared__ unsigned long long s_max[THREADS], s_idx[THREADS], s_sum[THREADS];
  __shared__ double s_l2[THREADS], s_lg[THREADS];
  s_max[threadIdx.x] = mx; s_idx[threadIdx.x] = mi; s_sum[threadIdx.x] = sm; s_l2[threadIdx.x] = l2; s_lg[threadIdx.x] = lg;
  __syncthreads();
  for (int s = THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const int o = threadIdx.x + s;
      if (s_max[o] > s_max[threadIdx.x] || (s_max[o] == s_max[threadIdx.x] && s_idx[o] > s_idx[threadIdx.x])) {
        s_max[threadIdx.x] = s_max[o]; s_idx[threadIdx.x] = s_idx[o];
      }
      s_sum[threadIdx.x] += s_sum[o]; s_l2[threadIdx.x] += s_l2[o]; s_lg[threadIdx.x] += s_lg[o];
    }
    __syncthreads();
  }
  mx = s_max[0]; mi = s_idx[0]; sm = s_sum[0]; l2 = s_l2[0]; lg = s_lg[0];
}

// ---------------------------------------------------------------------------------------------
// One leaf's share of two_layer.rs:185-197 (empty-leaf model), :226-259 (widening by the two boundary keys, longest run)
// and the Q7 count, from its fitted parameters `p`, its raw maximum `curr` and longest run `run` (runs of 1 reported
// as 0, see k_err_range): what k_finalize does per thread, for the kernels that finish a leaf where they fit it.
// ---------------------------------------------------------------------------------------------
// (finalize_one_pre: with the two boundary keys already in registers -- key_next = keys[e] or the type's maximum behind the
//  last key, key_prev = keys[s - 1] or zero in front of the first: k_leaf_lanes fetches them when the wave starts, so that
//  their round trip does not stand alone at the wave's end)
template <int LEAF, typename K>
__device__ __forceinline__ void finalize_one_pre(uint64_t j, uint64_t s, uint64_t e, const Span& sp, uint64_t L, const K* __restrict__ keys,
                                                 double* p, uint64_t curr, uint64_t run, uint64_t last_target, K key_next, K key_prev,
                                                 uint64_t& final_err, uint64_t& cnt_j) {
  const uint64_t n = sp.n;
  if (!(s < e)) {
    if constexpr (LEAF == K_CUBIC) { p[0] = 0.0; p[1] = 0.0; p[2] = (j + 1 < L) ? 0.0 : 1.0; p[3] = (j + 1 < L) ? (double)e : 0.0; }
    else { p[0] = (j + 1 < L) ? (double)e : 0.0; p[1] = 0.0; }
  }
  const uint64_t up_pred = leaf_predict<LEAF, K>(p, KeyTraits<K>::minus_eps(key_next));   // lower_bound_correction.rs:47-49
  const uint64_t upper = error_between(up_pred, e + 1, n);                   // two_layer.rs:229-235
  const uint64_t first_idx = (j == 0) ? e : s;                               // next_index(max(j-1,0))
  const uint64_t lo_pred = leaf_predict<LEAF, K>(p, KeyTraits<K>::plus_eps(key_prev));    // lower_bound_correction.rs:62-63
  const uint64_t lower = error_between(lo_pred, first_idx, n);               // two_layer.rs:237-247
  uint64_t m = curr;
  m = upper > m ? upper : m;
  m = lower > m ? lower : m;
  if (run == 0 && s < e && !(e == n && keys[s] == keys[n - 1])) run = 1;    // (Q5: the globally last run is never recorded)
  final_err = m + run;                                                       // two_layer.rs:250-251
  cnt_j = (e - s) + (last_target == j ? 1ull : 0ull);                        // Q7: tail duplicate
}
template <int LEAF, typename K>
__device__ __forceinline__ void finalize_one(uint64_t j, uint64_t s, uint64_t e, const Span& sp, uint64_t L, const K* __restrict__ keys,
                                             double* p, uint64_t curr, uint64_t run, uint64_t last_target,
                                             uint64_t& final_err, uint64_t& cnt_j) {
  const K key_next = e < sp.n ? keys[e] : KeyTraits<K>::max_value();
  const K key_prev = s > 0 ? keys[s - 1] : KeyTraits<K>::zero_value();
  finalize_one_pre<LEAF, K>(j, s, e, sp, L, keys, p, curr, run, last_target, key_next, key_prev, final_err, cnt_j);
}

// ---------------------------------------------------------------------------------------------
// k_finalize: one thread per leaf (O(L)).
// ---------------------------------------------------------------------------------------------
template <int LEAF, typename K>
__global__ void __launch_bounds__(256) k_finalize(const K* __restrict__ keys, Span sp, uint64_t L,
                                                  const unsigned long long* __restrict__ leaf_start,
                                                  const DevState* __restrict__ st,
                                                  double* __restrict__ params,
                                                  const unsigned long long* __restrict__ leaf_maxerr,
                                                  const unsigned long long* __restrict__ leaf_run,
                                                  unsigned long long* __restrict__ leaf_err,
                                                  unsigned long long* __restrict__ leaf_count,
                                                  unsigned char* __restrict__ rows,
                                                  StatsPartial* __restrict__ partials,
                                                  const K* __restrict__ bnext = nullptr, const K* __restrict__ bprev = nullptr) {
  constexpr int PPL = (LEAF == K_CUBIC) ? 4 : 2;
  constexpr int ROWB = PPL * 8 + 8;
  const uint64_t j = sp.leaf_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = j < sp.leaf_hi;
  const uint64_t n = sp.n;
  unsigned long long st_mx = 0, st_mi = 0, st_sum = 0;      // this leaf's terms of the aggregates
  double st_l2 = 0.0, st_lg = 0.0;
  if (in_range) {
  const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
  double p[PPL];
  if (s < e) {
#pragma unroll
    for (int q = 0; q < PPL; q++) p[q] = params[j * PPL + q];
  } else {
    // a leaf without keys: the fit passes may not have written it.  Empty model
    // (linear.rs:37-39 / cubic_spline.rs:19-21), replaced by the constant next_index(j) == e
    // for every leaf but the last (two_layer.rs:185-197; linear.rs:116-119, cubic_spline.rs:188-191).
    if constexpr (LEAF == K_CUBIC) { p[0] = 0.0; p[1] = 0.0; p[2] = (j + 1 < L) ? 0.0 : 1.0; p[3] = (j + 1 < L) ? (double)e : 0.0; }
    else { p[0] = (j + 1 < L) ? (double)e : 0.0; p[1] = 0.0; }
#pragma unroll
    for (int q = 0; q < PPL; q++) params[j * PPL + q] = p[q];
  }
  const uint64_t curr = leaf_maxerr[j];
  // upper error: two_layer.rs:229-235 ; next(j) = first (idx,key) of the next non-empty leaf
  // (one-pass mode: the boundary keys of a non-empty leaf inside the launch's key range were recorded by
  // k_sigma2 -- a coalesced read instead of two gathers from the key array; an empty leaf's constant model
  // predicts the same for every key, so it needs none)
  const bool rec = bnext != nullptr && s < e && s > sp.it_lo && e < sp.it_hi;
  const K key_next = rec ? bnext[j] : (s == e && bnext != nullptr ? K() : (e < n ? keys[e] : KeyTraits<K>::max_value()));   // lower_bound_correction.rs:47-49
  const uint64_t up_pred = leaf_predict<LEAF, K>(p, KeyTraits<K>::minus_eps(key_next));
  const uint64_t upper = error_between(up_pred, e + 1, n);
  // lower error: two_layer.rs:237-247 ; prev_key(j) = last key of the nearest non-empty leaf below
  const K key_prev = rec ? bprev[j] : (s == e && bnext != nullptr ? K() : (s > 0 ? keys[s - 1] : KeyTraits<K>::zero_value()));   // lower_bound_correction.rs:62-63
  const uint64_t first_idx = (j == 0) ? e : s;                             // next_index(max(j-1,0))
  const uint64_t lo_pred = leaf_predict<LEAF, K>(p, KeyTraits<K>::plus_eps(key_prev));
  const uint64_t lower = error_between(lo_pred, first_idx, n);
  uint64_t m = curr;
  m = upper > m ? upper : m;
  m = lower > m ? lower : m;
  // longest_run: k_err_tile reports only runs longer than 1; every leaf that owns a *recorded* run
  // (any run except the globally last one, Q5) has longest_run >= 1.
  uint64_t run = leaf_run[j];
  if (run == 0 && s < e && !(e == n && keys[s] == keys[n - 1])) run = 1;
  const uint64_t final_err = m + run;                                      // two_layer.rs:250-251
  leaf_err[j] = final_err;
  const uint64_t cnt_j = (e - s) + (st->last_target == j ? 1ull : 0ull);   // Q7: tail duplicate
  leaf_count[j] = cnt_j;
  // packed row = the reference's L1_PARAMETERS record (codegen.rs:288-315)
  double* rp = reinterpret_cast<double*>(rows + j * ROWB);
#pragma unroll
  for (int q = 0; q < PPL; q++) rp[q] = p[q];
  *reinterpret_cast<unsigned long long*>(rows + j * ROWB + PPL * 8) = final_err;
  // two_layer.rs:267-287
  st_mx = final_err; st_mi = j;
  st_sum = cnt_j * final_err;                                              // wrapping u64, like the reference's sum
  const double v = (double)st_sum;
  st_l2 = (v * v) / (double)n;
  st_lg = (double)cnt_j * log2((double)(2 * final_err + 2));
  }
  stats_block_reduce(st_mx, st_mi, st_sum, st_l2, st_lg);
  if (threadIdx.x == 0) partials[blockIdx.x] = StatsPartial{st_mx, st_mi, st_sum, st_l2, st_lg};
}

static __global__ void __launch_bounds__(1024) k_stats_reduce(const StatsPartial* __restrict__ partials, int count, DevState* __restrict__ st,
                                                       DevState* __restrict__ host_copy) {
  unsigned long long mx = 0, mi = 0, sm = 0;
  double l2 = 0.0, lg = 0.0;
  for (int q = threadIdx.x; q < count; q += 1024) {           // (lexicographic maximum and sums: any order combines)
    const StatsPartial p = partials[q];
    if (p.mx > mx || (p.mx == mx && p.mi > mi)) { mx = p.mx; mi = p.mi; }
    sm += p.sum; l2 += p.l2; lg += p.lg;
  }
  stats_block_reduce<1024>(mx, mi, sm, l2, lg);
  if (threadIdx.x == 0) {
    st->max_err = mx; st->max_err_idx = mi; st->sum_n_err = sm; st->sum_l2 = l2; st->sum_log2 = lg;
    if (host_copy) *host_copy = *st;                       // pinned host memory: visible to the host once the stream is synchronised
  }
}

// ---------------------------------------------------------------------------------------------
// Synthetic key generators (bit-identical to rmi_amd/datagen.py).  Integer only,
// sorted by construction, so a shard of the global array can be produced in place on any rank.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  unsigned long long z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// gen 0: uniform  key[i] = 1 + i*stride + (h(i + seed) mod stride)
// gen 1: dups     key[i] = uniform[i - (i mod r_i)], r_i in {1,1,1,2,8} by h(i/8 + dup_seed) mod 5
template <typename K>
__global__ void __launch_bounds__(256) k_generate(K* __restrict__ out, uint64_t start, uint64_t count,
                                                  unsigned long long stride, unsigned long long seed,
                                                  int gen, unsigned long long dup_seed) {
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long i = start + t;
    if (gen == 1) {
      const unsigned long long c = splitmix64((i >> 3) + dup_seed) % 5ull;
      const unsigned long long r = c < 3 ? 1ull : (c == 3 ? 2ull : 8ull);
      i = i - (i % r);
    }
    const unsigned long long k = 1ull + i * stride + (splitmix64(i + seed) % stride);
    out[t] = (K)k;
  }
}

}  // namespace rmi
