// rmi_sigma.hip.h -- the ONE-PASS leaf path for linear leaves (gfx950, wave64): bucketing scan, per-leaf
// least-squares fit from shifted sufficient statistics and the last-level error pass with a single
// coalesced HBM read of the keys (two_layer.rs:43-90, linear.rs:12-59, two_layer.rs:207-217).
//
// Why a second formulation next to the exact streaming kernels (rmi_stream.hip.h): the reference's
// fit is a Welford recurrence, order dependent, so bit-identical coefficients make every leaf a
// sequential chain and the keys have to be read twice (fit, then errors with the finished
// coefficients).  Here a leaf's line comes from sums (n, S dx, S dx^2, S dx dy) that any number of
// lanes can add up, so a wave finishes a leaf while its keys are still in LDS and runs the error
// pass from LDS.  The coefficients then differ from the reference's in the last digits (both are
// roundings of the same least-squares line); the per-leaf error INTEGERS stay bit-identical through a
// guard: every prediction of a leaf (its own keys and the two widening keys of two_layer.rs:226-259)
// must stay further from an integer than a bound `delta` on the distance between the two lines; a
// leaf that fails the test is handed to the exact kernels (list in HBM).
//
// Guard distance: a first-order bound on |reference line - this line| over the keys of a leaf with n
// points, key magnitude X, key range W, spread sigma_x, slope beta, positions up to Y (u = 2^-53):
//   reference (Welford, linear.rs:24-34): the running mean of x is rounded to ulp(X) at each of the n steps
//     and every step's dx inherits it -> n u |beta| X (offset) and n u |beta| W X / sigma_x (slope over W);
//   both lines: one rounding of alpha at the size of the prediction, u (|beta| X + Y);
//   these sums: cancellation of S dx^2 against (S dx)^2 / n, 4 u (S dx^2 / m2) |beta| W;
//   delta = guard_k (default 2) times the sum.  tests/devtools/calibrate.py measures the actual distance against the
//   oracle: at most 0.46 of the bound with guard_k = 1 over uniform / heavy-tailed / clustered key sets (200 M keys).
// Leaves the ring cannot finish are "irregular": duplicate keys (y is a first-occurrence offset), the leaves next to
// the split of the 2-way join (Q2/Q3), the first and the last leaf, leaves that do not fit the LDS ring, variance 0.
// linear_spline leaves (template parameter LEAFK) take the same kernel without sums or guard: two end points, the
// reference's two operations, the same bits -- in every fit mode.
// Mode 1 hands them to the exact kernels.  Mode 2 sums the LONG ones piecewise (one record per wave and stretch,
// merged by k_list with the container's rules, error pass by k_list_tail) and keeps every line the sums define.
#pragma once
#include <type_traits>

#include "rmi_device.hip.h"
#include "rmi_kernels.hip.h"
#include "rmi_stream.hip.h"

namespace rmi {

// The list of leaves for the exact kernels is kept in SG_REGIONS regions, region = leaf id mod SG_REGIONS, each with
// a counter of its own: appends from thousands of waves to ONE counter serialise at ~26 ns each (0.1 ms for the
// 4 000 waves of a 200 M-key run); spread by leaf id no region can overflow its share of the capacity.
constexpr int SG_REGIONS = 64;
constexpr int SG_SEG = 2048;                // error pass of a long listed leaf: stretches of this many keys, one wave each (k_list_tail)
constexpr int SG_ERR_LONG = 16384;          // listed leaves with more keys: error pass in stretches (k_list_tail)
struct SgList {
  unsigned int* ids;                       // [SG_REGIONS][cap]
  unsigned long long* cnt;                 // [SG_REGIONS]
  unsigned long long cap;                  // entries per region (>= ceil(leaves / SG_REGIONS))
  __device__ __forceinline__ void push(unsigned int leaf) const {
    const unsigned int rg = leaf % SG_REGIONS;
    const unsigned long long pos = atomicAdd(&cnt[rg], 1ull);
    if (pos < cap) ids[(unsigned long long)rg * cap + pos] = leaf;
  }
};

// A long leaf (one that does not fit a wave's ring, or that runs across chunks) is summed piecewise: every wave that
// holds a stretch of it leaves one record, shifted sums about the stretch's first key; k_list merges them.
constexpr unsigned int SG_TAG = 0x80000000u;   // list entry: "fitted from merged records" (leaf ids are below 2^31)
struct SgRec {
  unsigned int leaf, first, n, pad;        // keys [first, first + n) of the leaf
  double piv, a0, a1, a2;                  // x[first]; S dx, S dx^2, S dx dy with dx = x - piv, dy = index - first
};

struct SgParams {
  uint64_t chunk;                          // keys per block, a multiple of the batch
  double guard_k;                          // safety factor of the guard bound
  int mode;                                // 1: guard-flagged leaves are re-fitted exactly; 2: only counted
  int dbg;                                 // timing experiments only (results wrong): 1 no leaf rounds, 2 no sums loop, 4 no error loop, 8 loads only
  SgList flist;                            // leaf ids handed to the exact kernels
  SgRec* recs;                             // [blocks][rpw] (mode 2)
  unsigned int* rec_cnt;                   // [blocks]
  unsigned int rpw;                        // records a block can leave
  unsigned long long* segs;                // (leaf << 32 | stretch) of the merged leaves, DevState::seg_count entries
};

// 16 bytes of keys, streamed once (non-temporal); the address is only key-aligned in a shard
template <typename K>
__device__ __forceinline__ uint4 sg_load16(const K* __restrict__ p) {
  typedef unsigned int raw_t __attribute__((ext_vector_type(4 * sizeof(K) / sizeof(K)), aligned(sizeof(K))));
  const raw_t rw = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(p));
  return make_uint4(rw.x, rw.y, rw.z, rw.w);
}

// v_cvt_u32_f64: truncation with saturation to [0, 2^32-1], NaN -> 0: max(0, floor(f)) for f < 2^32
__device__ __forceinline__ unsigned int sg_cvt_u32(double f) { unsigned int r; asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(f)); return r; }
__device__ __forceinline__ unsigned int sg_absdiff(unsigned int a, unsigned int b) { unsigned int r; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }

// ---------------------------------------------------------------------------------------------
// The kernels for the leaves k_sigma2 hands over (k_list, k_list_tail below).
// ---------------------------------------------------------------------------------------------
// Moments of a set of points about a common origin, merged pairwise (Chan et al.): exact algebra, and the
// cancellation stays at the scale of one stretch.
struct SgMom {
  double n, mx, my, m2, c;
  __device__ __forceinline__ void add(const SgMom& b) {
    const double nn = n + b.n, dx = b.mx - mx, dy = b.my - my, w = n * b.n / nn;
    m2 += b.m2 + dx * dx * w;
    c += b.c + dx * dy * w;
    mx += dx * (b.n / nn);
    my += dy * (b.n / nn);
    n = nn;
  }
};
// The line of leaf j = keys [s, e) from the records the waves of k_sigma2 left for it.  The records hold the sums
// over the leaf's own keys; the container (leaf_container: [lo, hi], Q2/Q3 at the split and at the two ends of the
// data) adds the borrowed points, drops the key at the split, and its last item counts twice (Q1).  x relative to
// key s, y relative to s.  false: a duplicate or a missing record spoils the sums -> the exact path.
template <typename K>
__device__ bool sg_merge_long(uint64_t j, uint64_t s, uint64_t e, uint64_t lo, uint64_t hi, const K* __restrict__ keys, const Span& sp,
                              const SgParams& sg, double* __restrict__ params) {
  if (!(e > s) || lo < sp.rd_lo || hi >= sp.rd_hi) return false;
  uint64_t w = (s - sp.it_lo) / sg.chunk;
  const uint64_t nw = (sp.it_hi - sp.it_lo + sg.chunk - 1) / sg.chunk;
  const SgRec* rec = nullptr;
  {
    const unsigned int cnt = sg.rec_cnt[w] < sg.rpw ? sg.rec_cnt[w] : sg.rpw;
    const SgRec* base = sg.recs + w * sg.rpw;
    for (unsigned int q = 0; q < cnt; q++) if (base[q].leaf == (unsigned int)j && base[q].first == (unsigned int)s) { rec = base + q; break; }
  }
  if (!rec) return false;
  const double P = rec->piv;
  SgMom m; m.n = 0.0; m.mx = 0.0; m.my = 0.0; m.m2 = 0.0; m.c = 0.0;
  uint64_t next = s;
  for (;;) {
    if (rec->n == 0u) return false;
    const double nr = (double)rec->n, h = (nr - 1.0) * 0.5;
    SgMom b;
    b.n = nr;
    b.mx = (rec->piv - P) + rec->a0 / nr;
    b.my = (double)(next - s) + h;
    b.m2 = rec->a1 - rec->a0 * (rec->a0 / nr);
    b.c = rec->a2 - rec->a0 * h;
    if (m.n == 0.0) m = b; else m.add(b);
    next += rec->n;
    if (next >= e) break;
    if (++w >= nw || sg.rec_cnt[w] == 0u) return false;
    rec = sg.recs + w * sg.rpw;                                                // an inherited stretch is a wave's first record
    if (rec->leaf != (unsigned int)j || rec->first != (unsigned int)next) return false;
  }
  if (next != e || !(m.m2 == m.m2)) return false;                              // (NaN: duplicate keys inside the leaf)
  SgMom pt; pt.n = 1.0; pt.m2 = 0.0; pt.c = 0.0;
  if (lo > s) {                                                                // Q2: the key at the split belongs to no container
    if (lo != s + 1 || m.n < 2.0) return false;
    const double x = KeyTraits<K>::as_float(keys[s]) - P, y = 0.0;
    const double n1 = m.n - 1.0;
    const double mxo = m.mx - (x - m.mx) / n1, myo = m.my - (y - m.my) / n1, dx = x - mxo;
    m.c -= dx * (y - m.my); m.m2 -= dx * (x - m.mx);
    m.mx = mxo; m.my = myo; m.n = n1;
  } else if (lo < s) {                                                         // prev-last (two_layer.rs:74-78)
    pt.mx = KeyTraits<K>::as_float(keys[lo]) - P;
    pt.my = (double)first_occurrence(keys, lo, sp.rd_lo) - (double)s;
    m.add(pt);
  }
  // the last item of the container, twice (models/mod.rs:180): next-first (key e, its own first occurrence) or the last own key
  pt.mx = KeyTraits<K>::as_float(keys[hi]) - P; pt.my = (double)(hi - s);
  if (hi >= e) m.add(pt);
  m.add(pt);
  if (!(m.m2 > 0.0) || !(m.m2 < 1.7e308)) return false;
  const double beta = m.c / m.m2;
  const double alpha = ((double)s + m.my) - beta * (P + m.mx);
  if (!(fabs(beta) < 1.7e308) || !(fabs(alpha) < 1.7e308)) return false;
  params[2ull * j] = alpha; params[2ull * j + 1] = beta;
  return true;
}

// k_list: ONE WAVE per listed leaf does everything the leaf needs -- a tagged leaf is merged from its records (lane 0),
// any other one is fitted exactly (fit_long_leaf: the wave prepares 64 keys at a time and walks the recurrence at
// 28 ns per point; the O(1) containers -- empty, single borrowed point: Q4 -- by fit_one_leaf), and then the same wave
// runs the leaf's error pass + run lengths (two_layer.rs:207-217, lower_bound_correction.rs:104-119, exactly what
// k_err computes per key) while its keys are still in L2.  Leaves of more than SG_ERR_LONG keys, and the merged ones,
// leave their error pass to k_list_tail as stretches of SG_SEG keys: (leaf << 32 | stretch), bit 63 = "no duplicates
// inside" (a merged leaf: y is the key's index, runs of 1).  Block b takes region b % SG_REGIONS of the list.
// (Separate kernels for list fit, long fit, list errors and long errors cost four launches of ~4 us each; the
// counters are per region because same-address atomics serialise.)
constexpr unsigned long long SG_SEG_PLAIN = 1ull << 63;
template <typename K, int LEAFK = K_LINEAR>
__global__ void __launch_bounds__(64) k_list(const K* __restrict__ keys, Span sp,
                                             const unsigned long long* __restrict__ leaf_start, DevState* __restrict__ st,
                                             double* __restrict__ params, SgList fl, SgParams sg,
                                             unsigned long long* __restrict__ leaf_maxerr, unsigned long long* __restrict__ leaf_run,
                                             GiantLeaf* __restrict__ giant = nullptr, unsigned long long host_min = ~0ull,
                                             bool record_giants = true) {
  __shared__ FitLongLds lds;
  const unsigned int rg = blockIdx.x % SG_REGIONS;
  const unsigned long long cnt = fl.cnt[rg] < fl.cap ? fl.cnt[rg] : fl.cap;
  const unsigned int* ids = fl.ids + (unsigned long long)rg * fl.cap;
  const int lane = threadIdx.x;
  unsigned int merged_here = 0u;
  for (unsigned long long t = blockIdx.x / SG_REGIONS; t < cnt; t += gridDim.x / SG_REGIONS) {
    const bool tagged = (ids[t] & SG_TAG) != 0u;
    const uint64_t j = ids[t] & ~SG_TAG;
    const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
    uint64_t lo, hi;
    const int ck = leaf_container(j, s, e, sp.n, st->split_idx, st->split_target, lo, hi);
    // a container of more than host_min points: its fit is a sequential chain of that length, ~28 ns per point on a wave
    // and ~4 on a host core -- recorded for the host (rmi_hip.hip: the giant-leaf epilogue), nothing else done here
    if (LEAFK == K_LINEAR && giant != nullptr && !tagged && ck == 2 && hi - lo + 1 > host_min) {
      if (lane == 0 && record_giants) {                              // (else k_giant_scan has recorded it before this launch)
        const unsigned long long pos = atomicAdd(&st->giant_count, 1ull);
        if (pos < st->giant_cap) giant[pos] = GiantLeaf{j, lo, hi, first_occurrence(keys, lo, sp.rd_lo)};
      }
      continue;
    }
    bool merged = false;
    if (tagged && ck == 2) {
      int ok = 0;
      if (lane == 0) ok = sg_merge_long<K>(j, s, e, lo, hi, keys, sp, sg, params) ? 1 : 0;
      merged = __shfl(ok, 0) != 0;
    }
    if (merged) merged_here++;
    else if (LEAFK == K_LINEAR && ck == 2) {
      fit_long_leaf<K>(keys, sp, lo, hi, st, params + j * 2, lds);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // (the LDS buffers are reused by the next leaf)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if (lane == 0) fit_one_leaf<LEAFK, K>(j, keys, sp, leaf_start, st, params);
    if (merged || e - s > (uint64_t)SG_ERR_LONG) {
      if (lane == 0) {
        leaf_maxerr[j] = 0ull; leaf_run[j] = 0ull;                   // (k_list_tail raises them with atomicMax)
        const uint64_t nseg = (e - s + SG_SEG - 1) / SG_SEG;
        const unsigned long long pos = atomicAdd(&st->seg_count, (unsigned long long)nseg);
        const unsigned long long tagb = merged ? SG_SEG_PLAIN : 0ull;
        for (uint64_t q = 0; q < nseg && pos + q < st->seg_cap; q++) sg.segs[pos + q] = tagb | ((unsigned long long)j << 32) | q;
      }
      continue;
    }
    // the error pass of this leaf with the coefficients lane 0 has just written
    double pp[2];
    {
      double a = 0.0, b = 0.0;
      if (lane == 0) { a = params[j * 2]; b = params[j * 2 + 1]; }
      pp[0] = __shfl(a, 0); pp[1] = __shfl(b, 0);
    }
    unsigned long long err = 0, run = 0;
    for (uint64_t i = s + lane; i < e; i += 64) {
      const K k = keys[i];
      const uint64_t y = first_occurrence(keys, i, sp.rd_lo);
      const uint64_t er = error_between(leaf_predict<K_LINEAR, K>(pp, k), y, sp.n);
      err = er > err ? er : err;
      if (i + 1 < sp.n && !(keys[i + 1] == k)) { const uint64_t rl = i - y + 1; run = rl > run ? rl : run; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long oe = shfl_down_u64(err, d), orn = shfl_down_u64(run, d);
      err = oe > err ? oe : err; run = orn > run ? orn : run;
    }
    if (lane == 0) { leaf_maxerr[j] = err; leaf_run[j] = run; }
  }
  if (lane == 0 && merged_here) atomicAdd(&fl.cnt[SG_REGIONS + rg], (unsigned long long)merged_here);
}

// The giant leaves of the list, recorded BEFORE k_list runs (a launch of SG_REGIONS waves): the host reads the short list
// through pinned memory and walks their chains while k_list is still fitting the other listed leaves (rmi_hip.hip).
// Same condition as in k_list, which then only skips them.
template <typename K>
__global__ void __launch_bounds__(64) k_giant_scan(const K* __restrict__ keys, Span sp, const unsigned long long* __restrict__ leaf_start,
                                                   DevState* __restrict__ st, SgList fl, GiantLeaf* __restrict__ giant, unsigned long long host_min) {
  const unsigned int rg = blockIdx.x;
  const unsigned long long cnt = fl.cnt[rg] < fl.cap ? fl.cnt[rg] : fl.cap;
  const unsigned int* ids = fl.ids + (unsigned long long)rg * fl.cap;
  for (unsigned long long t = threadIdx.x; t < cnt; t += 64) {
    if (ids[t] & SG_TAG) continue;
    const uint64_t j = ids[t];
    uint64_t lo, hi;
    const int ck = leaf_container(j, leaf_start[j], leaf_start[j + 1], sp.n, st->split_idx, st->split_target, lo, hi);
    if (ck == 2 && hi - lo + 1 > host_min) {
      const unsigned long long pos = atomicAdd(&st->giant_count, 1ull);
      if (pos < st->giant_cap) giant[pos] = GiantLeaf{j, lo, hi, first_occurrence(keys, lo, sp.rd_lo)};
    }
  }
}

// k_list_tail: the error pass of the long listed leaves, one wave per stretch of SG_SEG keys, eight independent loads
// per lane in flight; block 0 also adds up the region counters for the caller.
template <typename K>
__global__ void __launch_bounds__(64) k_list_tail(const K* __restrict__ keys, Span sp,
                                                  const unsigned long long* __restrict__ leaf_start, DevState* __restrict__ st,
                                                  const double* __restrict__ params, SgList fl, const unsigned long long* __restrict__ segs,
                                                  unsigned long long* __restrict__ leaf_maxerr, unsigned long long* __restrict__ leaf_run,
                                                  const StatsPartial* __restrict__ wave_partials = nullptr, unsigned int nwave = 0,
                                                  StatsPartial* __restrict__ slice_out = nullptr) {
  constexpr int U = 8;
  const int lane = threadIdx.x;
  // (leaf-lane pipeline: this launch exists anyway -- its first SG_REGIONS blocks also combine the aggregate records of
  //  k_leaf_lanes' waves, a slice each, so that the single block of k_finalize_listed reads 64 records instead of L / 64)
  if (wave_partials != nullptr && blockIdx.x < (unsigned)SG_REGIONS) {
    const unsigned int per = (nwave + SG_REGIONS - 1) / SG_REGIONS;
    const unsigned int a = blockIdx.x * per, b = a + per < nwave ? a + per : nwave;
    unsigned long long mx = 0, mi = 0, sm = 0;
    double l2 = 0.0, lg = 0.0;
    for (unsigned int q = a + (unsigned int)lane; q < b; q += 64u) {
      const StatsPartial p = wave_partials[q];
      if (p.mx > mx || (p.mx == mx && p.mi > mi)) { mx = p.mx; mi = p.mi; }
      sm += p.sum; l2 += p.l2; lg += p.lg;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long omx = shfl_down_u64(mx, d), omi = shfl_down_u64(mi, d);
      if (omx > mx || (omx == mx && omi > mi)) { mx = omx; mi = omi; }
      sm += shfl_down_u64(sm, d);
      l2 += __shfl_down(l2, d);
      lg += __shfl_down(lg, d);
    }
    if (lane == 0) slice_out[blockIdx.x] = StatsPartial{mx, mi, sm, l2, lg};
  }
  if (blockIdx.x == 0) {
    unsigned long long a = fl.cnt[lane] < fl.cap ? fl.cnt[lane] : fl.cap, m = fl.cnt[SG_REGIONS + lane];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += shfl_down_u64(a, d); m += shfl_down_u64(m, d); }
    // (an entry dropped by a full region or a full stretch list would leave its leaf unfitted: say so instead)
    const bool over = fl.cnt[lane] > fl.cap || (lane == 0 && st->seg_count > st->seg_cap);
    if (__any(over)) { if (lane == 0) atomicOr(&st->err_flags, EF_LIST_OVERFLOW); }
    if (lane == 0) { st->flag_count = a; st->merged_count = m; }
  }
  const unsigned long long cnt = st->seg_count < st->seg_cap ? st->seg_count : st->seg_cap;
  for (unsigned long long t = blockIdx.x; t < cnt; t += gridDim.x) {
    const unsigned long long sv = segs[t];
    const bool plain = (sv & SG_SEG_PLAIN) != 0ull;
    const uint64_t j = (sv & ~SG_SEG_PLAIN) >> 32, q = sv & 0xffffffffull;
    const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
    const uint64_t a = s + q * SG_SEG, b = a + SG_SEG < e ? a + SG_SEG : e;
    const double pp[2] = {params[j * 2], params[j * 2 + 1]};
    unsigned long long err = 0, run = 0;
    if (plain) {
      for (uint64_t i0 = a + lane; i0 < b; i0 += 64 * U) {
        K kv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t i = i0 + (uint64_t)u * 64; kv[u] = keys[i < b ? i : b - 1]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint64_t i = i0 + (uint64_t)u * 64;
          if (i < b) { const uint64_t er = error_between(leaf_predict<K_LINEAR, K>(pp, kv[u]), i, sp.n); err = er > err ? er : err; }
        }
      }
      if (q == 0 && s + 1 < sp.n) run = 1;                         // (lower_bound_correction.rs:104-119: runs of equal keys)
    } else {
      for (uint64_t i = a + lane; i < b; i += 64) {
        const K k = keys[i];
        const uint64_t y = first_occurrence(keys, i, sp.rd_lo);
        const uint64_t er = error_between(leaf_predict<K_LINEAR, K>(pp, k), y, sp.n);
        err = er > err ? er : err;
        if (i + 1 < sp.n && !(keys[i + 1] == k)) { const uint64_t rl = i - y + 1; run = rl > run ? rl : run; }
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long oe = shfl_down_u64(err, d), orn = shfl_down_u64(run, d);
      err = oe > err ? oe : err; run = orn > run ? orn : run;
    }
    if (lane == 0) {
      // (same-address atomics serialise at ~26 ns: most stretches of a long leaf do not raise its maximum)
      if (err > __hip_atomic_load(&leaf_maxerr[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&leaf_maxerr[j], err);
      if (run > __hip_atomic_load(&leaf_run[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&leaf_run[j], run);
    }
  }
}

// =============================================================================================
// k_sigma2: AUTONOMOUS WAVES.  (A first version with block-wide tiles, rows of 16 keys per thread and seven
// barriers per tile was correct but spent half of its wave time in those barriers: 1.24 ms for 200 M keys.)
//
// A wave (= a block of 64 threads) owns a contiguous chunk of the keys and never synchronises
// with another wave.  Per batch of BATCH keys:
//   phase 1  classification straight from the registers the coalesced 16-byte loads landed in (the
//            keys of the next batch are in flight meanwhile): x = f64(key) of every key goes to a ring
//            in LDS (a duplicate key as NaN: its leaf then fails the checks below and goes to the
//            exact kernels); the root target is evaluated for the LAST key of each lane only and
//            compared with the previous lane's (DPP) -- targets are monotone, so a lane whose last
//            target equals its predecessor's holds no boundary; the lanes that do are noted per load and
//            ONE pass behind the batch's loads evaluates their other keys and appends (index, leaf id,
//            flags) to a list (edge batches: per load).
//   phase 2  every leaf that is complete in the ring (both boundaries seen) is handled by a GROUP of
//            GL lanes, 64/GL leaves per round: the lanes stride over the leaf's container
//            [s-1, e] (+ the Q1 duplicate of e) adding up (S dx, S dx^2, S dx dy) relative to the
//            leaf's first key, butterfly all-reduce inside the group (DPP), every lane solves
//            (alpha, beta, delta), then the lanes stride over the leaf's own keys for
//            max |pred - y| and the closest approach of a prediction to an integer (+ the widening keys
//            on six lanes), butterfly max, lane 0 of the group decides: leaf_maxerr, or the exact list.
// A leaf that does not fit the ring (RING keys) is irregular ("long"); so are the leaves at the split
// of the 2-way join, the first and the last leaf, and leaves with duplicate keys.
// =============================================================================================
constexpr unsigned S2_SPLIT = 1u, S2_START = 2u, S2_END = 4u, S2_LONG = 8u, S2_INHERIT = 16u;
constexpr int S2_PAD = 128;                          // the first S2_PAD ring entries are mirrored behind its end
constexpr int S2_U = 4;                              // keys per lane and unrolled step of the two leaf loops

// butterfly all-reduce inside groups of GL lanes (GL = 4, 8, 16): every lane ends up with the total
template <int CTRL>
__device__ __forceinline__ double s2_dpp_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
template <int CTRL>
__device__ __forceinline__ unsigned int s2_dpp_u32(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// lane ^ 16 inside groups of 32 lanes (ds_swizzle, bit mode: and 0x1F, or 0, xor 0x10)
__device__ __forceinline__ double s2_swz16_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_ds_swizzle((int)(unsigned int)b, 0x401F);
  const int hi = __builtin_amdgcn_ds_swizzle((int)(unsigned int)(b >> 32), 0x401F);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
template <int GL> __device__ __forceinline__ double s2_group_sum(double v) {
  v += s2_dpp_f64<0xB1>(v);                        // quad_perm [1,0,3,2]
  v += s2_dpp_f64<0x4E>(v);                        // quad_perm [2,3,0,1]
  if constexpr (GL >= 8) v += s2_dpp_f64<0x141>(v);    // row_half_mirror
  if constexpr (GL >= 16) v += s2_dpp_f64<0x140>(v);   // row_mirror
  if constexpr (GL >= 32) v += s2_swz16_f64(v);
  return v;
}
template <int GL> __device__ __forceinline__ double s2_group_max(double v) {
  v = fmax_raw(v, s2_dpp_f64<0xB1>(v));
  v = fmax_raw(v, s2_dpp_f64<0x4E>(v));
  if constexpr (GL >= 8) v = fmax_raw(v, s2_dpp_f64<0x141>(v));
  if constexpr (GL >= 16) v = fmax_raw(v, s2_dpp_f64<0x140>(v));
  if constexpr (GL >= 32) v = fmax_raw(v, s2_swz16_f64(v));
  return v;
}
template <int GL> __device__ __forceinline__ unsigned int s2_group_max(unsigned int v) {
  v = max(v, s2_dpp_u32<0xB1>(v));
  v = max(v, s2_dpp_u32<0x4E>(v));
  if constexpr (GL >= 8) v = max(v, s2_dpp_u32<0x141>(v));
  if constexpr (GL >= 16) v = max(v, s2_dpp_u32<0x140>(v));
  if constexpr (GL >= 32) v = max(v, (unsigned int)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));
  return v;
}

// 1 / v to a few ulp (v_rcp_f64 + two Newton steps): these quotients only have to be accurate, not IEEE
__device__ __forceinline__ double s2_rcp(double v) {
  double y = __builtin_amdgcn_rcp(v);
  y = __builtin_fma(__builtin_fma(-v, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-v, y, 1.0), y, y);
  return y;
}

// min(L-1, predict_to_int(key)) as u32 (L <= 2^31): float roots through v_cvt_u32_f64 (truncation,
// saturation, NaN -> 0: exactly max(0, floor(f)) clamped), the others through the generic form.
template <int ROOT, typename K>
__device__ __forceinline__ unsigned int s2_target(const RootP& r, double Lm1f, unsigned int Lm1, K k, double x, bool& oob) {
  if constexpr (ROOT == K_RADIX || ROOT == K_RADIX_TABLE) {
    return (unsigned int)root_target_f<ROOT, K>(r, Lm1f, k, oob);
  } else {
    const unsigned int u = sg_cvt_u32(root_eval_f<ROOT>(r, x));
    oob = u > Lm1;
    return u < Lm1 ? u : Lm1;
  }
}
// roots whose targets are monotone in the key BY ARITHMETIC (fma and the shifts round monotonically), so that
// equal targets at the two ends of a key range prove that there is no boundary inside; the others
// (cubic, loglinear, normal; a radix table may be caller-provided) are evaluated at every key.
template <int ROOT> __device__ __forceinline__ constexpr bool s2_root_monotone() { return ROOT == K_LINEAR || ROOT == K_RADIX; }

template <int ROOT, typename K, int RING, int BATCH, bool LSUM, int LEAFK = K_LINEAR>
__global__ void __launch_bounds__(64) k_sigma2(const K* __restrict__ keys, Span sp, RootP r, SgParams sg,
                                               unsigned long long* __restrict__ leaf_start, double* __restrict__ params,
                                               unsigned long long* __restrict__ leaf_maxerr, DevState* __restrict__ st,
                                               K* __restrict__ bnext, K* __restrict__ bprev) {
  constexpr int KPL = 16 / (int)sizeof(K);          // keys per lane and load
  constexpr int NLOAD = BATCH / (64 * KPL);         // loads per batch
  constexpr int BCAP = 192;                         // boundary list (more boundaries in a batch: see `dense`)
  constexpr int LBUF = 64;
  constexpr int U = S2_U;
  constexpr unsigned MASK = RING - 1;
  constexpr bool SPARSE = s2_root_monotone<ROOT>();
  static_assert((RING & (RING - 1)) == 0 && BATCH % (64 * KPL) == 0 && RING >= 2 * BATCH && (U - 1) * 32 + 1 <= S2_PAD, "geometry");
  // 4-byte keys stay RAW in the ring (converted on read: one instruction): twice the keys in the same LDS, so the batch
  // is four loads like an 8-byte key's.  A raw key cannot carry the NaN that marks a duplicate: the loads that hold
  // one are noted in a bit mask over the ring's load slots (`dupw`, a wave-uniform register) instead.
  constexpr bool RAW = std::is_same<K, uint32_t>::value;
  using RingT = std::conditional_t<RAW, unsigned int, double>;
  constexpr int SLOT = 64 * KPL;                    // keys per load = ring positions per slot
  static_assert(!RAW || RING / SLOT <= 32, "dupw");
  __shared__ RingT xring[RING + S2_PAD];
  unsigned int dupw = 0u;
  auto rx = [&](unsigned int pos) -> double { return (double)xring[pos]; };
  __shared__ unsigned int b_idx[BCAP], b_t[BCAP];
  __shared__ unsigned char b_fl[BCAP];
  __shared__ unsigned int l_buf[LBUF];

  const int lane = threadIdx.x;
  const uint64_t c0 = sp.it_lo + (uint64_t)blockIdx.x * sg.chunk;
  if (c0 >= sp.it_hi) return;
  const uint64_t c1 = (c0 + sg.chunk < sp.it_hi) ? c0 + sg.chunk : sp.it_hi;
  const unsigned int c0u = (unsigned int)c0, c1u = (unsigned int)c1;
  // ring position of key i: (i - c0) mod RING -- relative to the chunk start, so that a lane's 16 bytes never
  // straddle the end of the ring and a load covers an aligned stretch of it whatever the shard's first index is
  auto rpos = [&](unsigned int i) -> unsigned int { return (i - c0u) & MASK; };
  const double Lm1f = (double)(r.L - 1);
  const unsigned int Lm1 = (unsigned int)(r.L - 1);
  const unsigned int mid = (unsigned int)(r.L / 2);                 // two_layer.rs:131
  const unsigned int n32 = (unsigned int)sp.n;
  // every index of this kernel is below 2^32 - 2^16 (the launch checks it): 32-bit index arithmetic throughout
  const unsigned int rdlo = (unsigned int)sp.rd_lo, rdhi = (unsigned int)sp.rd_hi, itlo = (unsigned int)sp.it_lo;
  const unsigned int leaf_lo32 = (unsigned int)sp.leaf_lo;
  const unsigned long long below = (1ull << lane) - 1ull;

  uint4 cur[NLOAD], nxt[NLOAD];
  // One 16-byte load per lane and piece, UNCONDITIONALLY (a second, scalar path for the end of the data would make
  // the number of loads in flight unknown to the compiler, which then waits for all of them at once: no prefetch).
  // A piece that would reach past the readable keys is fetched from the last full piece instead and realigned in
  // phase 1 (edge batches only).
  const unsigned int lim = rdhi - (unsigned int)KPL;
  auto load_batch = [&](uint4 (&dst)[NLOAD], unsigned int A) {
#pragma unroll
    for (int k = 0; k < NLOAD; k++) {
      const unsigned int gi = A + (unsigned int)((k * 64 + lane) * KPL);
      dst[k] = sg_load16<K>(keys + (gi < lim ? gi : lim));
    }
  };
  auto ring_store = [&](unsigned int pos, RingT v) {                 // one entry (rare paths)
    xring[pos] = v;
    if (pos < (unsigned)S2_PAD) xring[pos + RING] = v;
  };
  // duplicates among the ring positions [a, b] (RAW): any noted load slot the stretch touches
  auto dup_in = [&](unsigned int pa, unsigned int pb) -> bool {
    if constexpr (!RAW) return false;
    else {
      const unsigned int sa = pa / (unsigned)SLOT, sb = pb / (unsigned)SLOT;
      const unsigned int upto_b = (2u << sb) - 1u, from_a = ~((1u << sa) - 1u);
      const unsigned int m = sa <= sb ? (upto_b & from_a) : (upto_b | from_a);      // (wrapped: the slots from sa up, and up to sb)
      return (dupw & m & ((RING / SLOT >= 32) ? 0xFFFFFFFFu : ((1u << (RING / SLOT)) - 1u))) != 0u;
    }
  };

  if (lane == 0 && sp.n - 1 >= c0 && sp.n - 1 < c1 && sp.n - 1 < sp.rd_hi) {
    bool oob_;
    st->last_target = (unsigned long long)root_target_f<ROOT, K>(r, Lm1f, keys[sp.n - 1], oob_);
  }
  // the key before the chunk
  K carry_key = K();
  unsigned int carry_t = 0u;
  if (c0 > sp.rd_lo) {
    bool oob_;
    carry_key = keys[c0 - 1];
    carry_t = s2_target<ROOT, K>(r, Lm1f, Lm1, carry_key, KeyTraits<K>::as_float(carry_key), oob_);
    // it is the prev-last point of a leaf that starts exactly at c0 (NaN if it is a duplicate: y is not its index)
    const bool dupc = (c0 - 1 > sp.rd_lo) && (keys[c0 - 2] == carry_key);
    if constexpr (RAW) {
      if (lane == 0) ring_store(rpos((unsigned int)(c0 - 1)), (RingT)key_to_bits<K>(carry_key));
      if (dupc) dupw |= 1u << (rpos((unsigned int)(c0 - 1)) / (unsigned)SLOT);
    } else {
      if (lane == 0) ring_store(rpos((unsigned int)(c0 - 1)), (RingT)(dupc ? __builtin_nan("") : KeyTraits<K>::as_float(carry_key)));
    }
  }
  int bcnt = 0;                                                     // entries of the boundary list (wave-uniform)
  // Leaves for the exact kernels are collected per wave and appended to the global list in one piece:
  // same-address atomics serialise at ~26 ns each.
  int lcnt = 0;
  unsigned int guard_cnt = 0;
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto flush_exact = [&]() {
    for (int q = lane; q < lcnt; q += 64) sg.flist.push(l_buf[q]);
    lcnt = 0;
  };
  bool stop = false;                                                // a boundary at or behind c1 is in the list
  unsigned int eflags = 0;

  // ---- long leaves.  A leaf that outgrows the ring, or that is still open XWIN keys behind the end of the chunk,
  // cannot be finished from LDS.  Chunk rule (both sides evaluate it on the same keys): a leaf open at a chunk
  // border c with no boundary in [c, c + XWIN) is cut at c -- the wave before stops there, the wave behind takes
  // [c, ...) as an "inherited" stretch.  In mode 2 every stretch is summed while it passes through the ring (all 64
  // lanes, shifted sums about the stretch's first key) and left as one record; k_list merges the records and
  // the error kernels of the list do the rest.  In mode 1 such a leaf cannot be certified: exact kernels.
  constexpr unsigned int XWIN = RING / 2;
  static_assert(XWIN % BATCH == 0, "the chunk rule is evaluated at batch ends");
  // (the rarely touched scalars of this path live in LDS: as wave-uniform registers they pushed the hot loop's
  //  scalars into spills -- 25 us on the 200 M-key run)
  __shared__ unsigned int acc_u[4];                                 // first and next index of the stretch being summed, records left
  __shared__ double acc_pxs;                                        // x of the stretch's first key
  if (lane == 0) acc_u[2] = 0u;
  unsigned int lst = 0u;                                            // LS_*: state of this path
  constexpr unsigned int LS_OPEN = 1u, LS_ACC = 2u, LS_SEEN = 4u;   // entry 0 is an open long leaf; it is being summed; a boundary was seen
  double P0 = 0.0, P1 = 0.0, P2 = 0.0;
  auto acc_start = [&](unsigned int s0) {
    lst |= LS_ACC;
    const double px = rx(rpos(s0));
    if (lane == 0) { acc_u[0] = s0; acc_u[1] = s0; acc_pxs = px; }
    P0 = 0.0; P1 = 0.0; P2 = 0.0;
    wave_sync();
  };
  auto acc_run = [&](unsigned int upto) {                           // adds the keys [next, upto) (still in the ring)
    const unsigned int first = acc_u[0], next = acc_u[1];
    if (upto <= next) return;
    const double px = acc_pxs;
    for (unsigned int k = next + (unsigned int)lane; k < upto; k += 64u) {
      const double dx = rx(rpos(k)) - px, dy = (double)(k - first);
      P0 += dx; P1 = __builtin_fma(dx, dx, P1); P2 = __builtin_fma(dx, dy, P2);
    }
    if (dup_in(rpos(next), rpos(upto - 1u))) P0 = __builtin_nan("");      // (RAW: a duplicate in the stretch spoils the sums like a NaN x would)
    wave_sync();
    if (lane == 0) acc_u[1] = upto;
    wave_sync();
  };
  auto acc_emit = [&](unsigned int leaf) {
    double t0 = s2_group_sum<32>(P0), t1 = s2_group_sum<32>(P1), t2 = s2_group_sum<32>(P2);
    t0 += __shfl_xor(t0, 32); t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
    const unsigned int rcnt = acc_u[2];
    wave_sync();
    if (lane == 0) {
      if (rcnt < sg.rpw) {
        SgRec rc; rc.leaf = leaf; rc.first = acc_u[0]; rc.n = acc_u[1] - acc_u[0]; rc.pad = 0u;
        rc.piv = acc_pxs; rc.a0 = t0; rc.a1 = t1; rc.a2 = t2;
        sg.recs[(unsigned long long)blockIdx.x * sg.rpw + rcnt] = rc;
      }
      acc_u[2] = rcnt + 1u;
    }
    lst &= ~LS_ACC;
    wave_sync();
  };

  // ---- phase 2: the complete leaves [entry i, entry i+1) of the list ----
  // One round: the leaves [entry i0 + g, entry i0 + g + 1), g < 64 / GL, one group of GL lanes each.
  auto round = [&](auto gl_tag, int i0) {
    constexpr int GL = decltype(gl_tag)::value;
    constexpr int NG = 64 / GL;
    const int g = lane / GL, l = lane % GL;
    {
      const int i = i0 + g;
      const bool act = i + 1 < bcnt;
      unsigned int s = 0, e = 0, leaf = 0, fl = 0;
      if (act) {
        s = b_idx[i]; e = b_idx[i + 1]; leaf = b_t[i];
        fl = (b_fl[i] & (S2_SPLIT | S2_START | S2_LONG)) | (b_fl[i + 1] & (S2_SPLIT | S2_END));
      }
      const bool owned = act && s >= c0u && s < c1u;
      bool irregular = fl != 0u;
      const bool work = owned && !irregular;
      double alpha = 0.0, beta = 0.0, delta = 1.0;
      double xpl = 0.0, xe = 0.0;
      if constexpr (LEAFK == K_LINEAR_SPLINE) {
        // linear_spline.rs:13-35: the line through the first and the last point of the container [s-1, e] -- the same two
        // points, the same operations, so the same bits as the reference; nothing to sum and nothing to guard.
        if (work) { xpl = rx(rpos(s - 1u)); xe = rx(rpos(e)); }
        if (!(xpl < xe) || (work && dup_in(rpos(s - 1u), rpos(e)))) irregular = true;                            // NaN (a duplicate: y is not the index), or two keys on one f64
        else {
          const double y0 = (double)(s - 1u), y1 = (double)e;
          beta = (y0 - y1) / (xpl - xe);
          alpha = y0 - beta * xpl;                                    // plain multiply and subtract
        }
        delta = 0.0;                                                  // (no rounding differences to guard against)
      } else {
      // ---- sums over the container [s-1, e] relative to (x[s], s); a lane takes the keys s-1+l, s-1+l+GL, ...
      double p = 0.0;
      double R0 = 0.0, R1 = 0.0, Q = 0.0;                            // Q = sum of the running R0: S (index in the lane) dx = T R0 - Q
      unsigned int T = 0;
      if (work) {
        p = rx(rpos(s)); xpl = rx(rpos(s - 1u)); xe = rx(rpos(e));
        if (dup_in(rpos(s - 1u), rpos(e))) R0 = __builtin_nan("");    // (RAW: the leaf then fails m2 > 0 below like one with a NaN x)
        unsigned int k = s - 1u + (unsigned int)l;
        if (sg.dbg & 2) k = e + 1u;
        // U keys per step while the whole group has them: all loads first, no wrap inside a step (mirror)
        for (; k + (unsigned)((U - 1) * GL) <= e; k += U * GL) {
          const RingT* xp = &xring[rpos(k)];
          double xv[U];
#pragma unroll
          for (int u = 0; u < U; u++) xv[u] = (double)xp[u * GL];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const double dx = xv[u] - p;
            R0 += dx; R1 = __builtin_fma(dx, dx, R1); Q += R0;
          }
          T += U;
        }
        for (; k <= e; k += GL) {
          const double dx = rx(rpos(k)) - p;
          R0 += dx; R1 = __builtin_fma(dx, dx, R1); Q += R0;
          T += 1;
        }
      }
      // S dx dy with dy = (l - 1) + GL * j for the lane's j-th key:  (l - 1) R0 + GL (T R0 - Q);  the Q1 duplicate of e on lane 0
      double R2 = __builtin_fma((double)(l - 1), R0, (double)GL * __builtin_fma((double)T, R0, -Q));
      if (work && l == 0) {
        const double dx = xe - p, dyy = (double)(e - s);
        R0 += dx; R1 = __builtin_fma(dx, dx, R1); R2 = __builtin_fma(dx, dyy, R2);
      }
      R0 = s2_group_sum<GL>(R0); R1 = s2_group_sum<GL>(R1); R2 = s2_group_sum<GL>(R2);
      // ---- every lane of the group solves
      const double len = (double)(e - s);
      const double cnt = len + 3.0;
      const double sy = (len + 2.0) * (len - 1.0) * 0.5 + len;      // S (i - s) over i = s-1 .. e, plus (e - s) once more
      const double rn = s2_rcp(cnt);
      const double mx = R0 * rn, my = sy * rn;
      const double m2 = __builtin_fma(-R0, mx, R1);
      const double cxy = __builtin_fma(-R0, my, R2);
      if (!(m2 > 0.0) || !(R1 < 1.7e308)) irregular = true;         // all keys on one f64 (linear.rs:50-53), NaN (duplicates), overflow
      else {
        const double rm2 = s2_rcp(m2);
        beta = cxy * rm2;
        beta = __builtin_fma(__builtin_fma(-beta, m2, cxy), rm2, beta);
        alpha = ((double)s + my) - beta * (p + mx);
        if (!(fabs(beta) < 1.7e308) || !(fabs(alpha) < 1.7e308)) irregular = true;
        // guard distance (see the head of this file)
        const double X = fmax(fabs(xpl), fabs(xe)), W = xe - xpl, ab = fabs(beta);
        const double wos = W * __builtin_amdgcn_rsq(m2 * rn);         // W / sigma_x
        delta = sg.guard_k * 1.1102230246251565e-16 * 1.0001 * (cnt * ab * X * (1.0 + wos) + ab * X + (double)e + 4.0 * (R1 * rm2) * ab * W);
        if (!(delta < 0.5) && sg.mode == 1) irregular = true;         // (mode 2: counted with the guard-flagged leaves)
      }
      }
      // ---- error pass over the own keys [s, e)
      unsigned int emax = 0u;
      double hmax = 0.0;
      unsigned int nanf = 0u;                                        // (linear_spline: a duplicate key inside the leaf shows as NaN here)
      if (work && !irregular) {
        unsigned int k = s + (unsigned int)l;
        if (sg.dbg & 4) k = e;
        for (; k + (unsigned)((U - 1) * GL) < e; k += U * GL) {
          const RingT* xp = &xring[rpos(k)];
          double xv[U];
#pragma unroll
          for (int u = 0; u < U; u++) xv[u] = (double)xp[u * GL];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const double f = __builtin_fma(beta, xv[u], alpha);            // linear.rs:87-90
            const unsigned int pr = min(sg_cvt_u32(f), n32);              // models/mod.rs:735-737, two_layer.rs:14-18
            emax = max(emax, sg_absdiff(pr, k + (unsigned)(u * GL)));
            if constexpr (LEAFK == K_LINEAR) hmax = fmax_abs_raw(hmax, __builtin_amdgcn_fract(f) - 0.5);
            else nanf |= (f != f) ? 1u : 0u;
          }
        }
        for (; k < e; k += GL) {
          const double f = __builtin_fma(beta, rx(rpos(k)), alpha);
          const unsigned int pr = min(sg_cvt_u32(f), n32);
          emax = max(emax, sg_absdiff(pr, k));
          if constexpr (LEAFK == K_LINEAR) hmax = fmax_abs_raw(hmax, __builtin_amdgcn_fract(f) - 0.5);
          else nanf |= (f != f) ? 1u : 0u;
        }
        // widening keys (two_layer.rs:229-247), one candidate per lane: RN(key[e] - 1), RN(key[s-1] + 1).  For u64
        // keys only x = RN(key) is at hand: RN(key -+ 1) is x -+ 1 below 2^53 and x or its neighbour above.
        if (LEAFK == K_LINEAR && l < 6) {
          double xc;
          const bool hiside = l < 3;
          const double xb = hiside ? xe : xpl;
          if constexpr (std::is_same<K, double>::value) xc = hiside ? KeyTraits<K>::minus_eps(xb) : KeyTraits<K>::plus_eps(xb);
          else {
            const int v = hiside ? l : l - 3;                       // 0: x -+ 1, 1: x, 2: the neighbouring double
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, xb);
            xc = v == 0 ? (hiside ? xb - 1.0 : xb + 1.0) : (v == 1 ? xb : __builtin_bit_cast(double, hiside ? bits - 1ull : bits + 1ull));
          }
          const double f = __builtin_fma(beta, xc, alpha);
          hmax = fmax_abs_raw(hmax, __builtin_amdgcn_fract(f) - 0.5);
        }
      }
      emax = s2_group_max<GL>(emax);
      if constexpr (LEAFK == K_LINEAR) hmax = s2_group_max<GL>(hmax);
      else if (s2_group_max<GL>(nanf)) irregular = true;              // y of a duplicate is not its index: exact kernels
      const bool head = owned && l == 0;                             // the lane that speaks for the leaf
      const bool guarded = head && !irregular && !(0.5 - hmax >= delta);
      guard_cnt += (unsigned int)__popcll(__ballot(guarded));
      const bool exact = head && (irregular || (guarded && sg.mode == 1));
      if (head) {
        if (!irregular) { params[2ull * leaf] = alpha; params[2ull * leaf + 1] = beta; }
        if (!exact) leaf_maxerr[leaf] = (unsigned long long)emax;
      }
      const unsigned long long em = __ballot(exact);
      if (em) {
        if (exact) l_buf[lcnt + __popcll(em & below)] = leaf;
        lcnt += __popcll(em);
        if (lcnt > LBUF - 16) { wave_sync(); flush_exact(); }
      }
    }
    (void)NG;
  };
  // The complete leaves of the list, 64 / GL per round.  The group width follows the leaves' length (about 24
  // keys per lane); a round waits until the list holds enough leaves to occupy every group, unless `force`
  // (the ring is needed for the next batch, or the chunk ends).  The entries left over move to the front.
  auto process = [&](bool force, bool drop_open) {
    wave_sync();
    int head = 0;
    if (LSUM && (lst & LS_OPEN) && bcnt >= 2) {                     // the long leaf at entry 0 ends at entry 1
      const unsigned int leaf = b_t[0], f0 = b_fl[0];
      bool tag = false;
      if (lst & LS_ACC) { acc_run(b_idx[1]); acc_emit(leaf); tag = true; }
      if (!(f0 & S2_INHERIT)) {                                     // (the owner lists it; the others only leave records)
        if (lane == 0) l_buf[lcnt] = leaf | (tag ? SG_TAG : 0u);
        lcnt++;
        if (lcnt > LBUF - 16) { wave_sync(); flush_exact(); }
      }
      lst &= ~LS_OPEN;
      head = 1;
    }
    while (bcnt - 1 - head > 0) {
      const int pend = bcnt - 1 - head;
      const unsigned int span = b_idx[head + (pend < 8 ? pend : 8)] - b_idx[head];
      const unsigned int avg = span / (unsigned int)(pend < 8 ? pend : 8);
      const int want = (avg <= 320u && RING >= 2048) ? 8 : (avg <= 640u ? 4 : 2);
      if (pend >= want) {
        if (want == 8) round(std::integral_constant<int, 8>{}, head);
        else if (want == 4) round(std::integral_constant<int, 16>{}, head);
        else round(std::integral_constant<int, 32>{}, head);
        head += want;
      } else if (force) {
        if (pend >= 5 && want == 8) { round(std::integral_constant<int, 8>{}, head); head += pend; }
        else if (pend >= 4) { round(std::integral_constant<int, 16>{}, head); head += 4; }
        else { round(std::integral_constant<int, 32>{}, head); head += pend < 2 ? pend : 2; }
      } else break;
    }
    // compact: the entries from `head` on (pending leaves and the start of the open one) move to the front
    if (drop_open) { bcnt = 0; lst &= ~(LS_OPEN | LS_ACC); }
    else if (head > 0) {
      const int keep = bcnt - head;
      wave_sync();
      for (int base = 0; base < keep; base += 64) {
        const int q = base + lane;
        unsigned int vi = 0, vt = 0; unsigned char vf = 0;
        if (q < keep) { vi = b_idx[head + q]; vt = b_t[head + q]; vf = b_fl[head + q]; }
        wave_sync();
        if (q < keep) { b_idx[q] = vi; b_t[q] = vt; b_fl[q] = vf; }
      }
      bcnt = keep;
    }
    wave_sync();
  };

  // batch counters of the chunk rule: the batch that ends at c0 + XWIN, and the first one that ends at or behind c1 + XWIN
  const int bi_inherit = c0 > sp.rd_lo ? (int)(XWIN / BATCH) - 1 : -1;
  const int bi_giveup = (int)((c1 - c0 + XWIN + BATCH - 1) / BATCH) - 1;
  int bi = -1;
  load_batch(nxt, c0u);
  for (unsigned int A = c0u; !stop; A += BATCH) {
    bi++;
#pragma unroll
    for (int k = 0; k < NLOAD; k++) cur[k] = nxt[k];
    load_batch(nxt, A + BATCH);
    const bool interior = (A > rdlo) && (A + BATCH <= rdhi);
    bool dense = false;
    if (sg.dbg & 8) {                                               // timing experiment: the loads alone
      unsigned int acc = 0;
#pragma unroll
      for (int k = 0; k < NLOAD; k++) acc ^= cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
      if (acc == 0x12345678u) eflags |= 64u;
      if (A + BATCH >= c1u) stop = true;
      continue;
    }
    // Interior batches: the lanes that hold a boundary are only noted per load (LB), with the key / target in front of the
    // load; ONE pass behind the batch's loads then handles the boundaries of all of them, each lane with the load it has
    // one in.  The boundary code costs a wave ~90 vector instructions whether 1 or 64 lanes are in it, and 2.7 loads of
    // a 512-key batch held a boundary at ~190 keys per leaf: per load it was 41 % of the kernel's vector instructions.
    unsigned long long LB[NLOAD];
    K ck[NLOAD];
    auto phase1 = [&](auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      constexpr bool FULL = EDGE || !SPARSE;                        // every key's target is evaluated up front
#pragma unroll
      for (int k = 0; k < NLOAD; k++) {
        K kk[KPL];
        __builtin_memcpy(kk, &cur[k], 16);
        const unsigned int g0 = A + (unsigned int)((k * 64 + lane) * KPL);
        if constexpr (EDGE) {
          if (g0 > lim) {                                           // fetched from `lim`: realign, pad with the last key
            const unsigned int sh = (unsigned int)(g0 - lim);
            K t[KPL];
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              t[q] = kk[KPL - 1];
#pragma unroll
              for (int w = 1; w < KPL; w++) if (sh == (unsigned)w && q + w < KPL) t[q] = kk[q + w];
            }
#pragma unroll
            for (int q = 0; q < KPL; q++) kk[q] = t[q];
          }
        }
        double xs[KPL];
        unsigned int ts[KPL];
        bool oobany = false;
#pragma unroll
        for (int q = 0; q < KPL; q++) {
          xs[q] = KeyTraits<K>::as_float(kk[q]);
          if (FULL || q == KPL - 1) {
            bool oob;
            ts[q] = s2_target<ROOT, K>(r, Lm1f, Lm1, kk[q], xs[q], oob);
            if constexpr (!root_needs_bounds_check<ROOT>()) {
              if constexpr (EDGE) oobany |= oob && (g0 + (unsigned int)q < rdhi); else oobany |= oob;
            }
          } else ts[q] = 0u;
        }
        // the key / target before this lane's first key: the previous lane's last one (lane 0: the carry)
        K kp0;
        unsigned int tp0;
        {
          const unsigned long long lastb = key_to_bits<K>(kk[KPL - 1]), cb = key_to_bits<K>(carry_key);
          const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)cb, (int)(unsigned int)lastb, 0x138, 0xF, 0xF, false);   // wave_shr:1
          unsigned int hi = 0u;
          if constexpr (sizeof(K) == 8) hi = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)(cb >> 32), (int)(unsigned int)(lastb >> 32), 0x138, 0xF, 0xF, false);
          kp0 = bits_to_key<K>(((unsigned long long)hi << 32) | lo);
          tp0 = (unsigned int)__builtin_amdgcn_update_dpp((int)carry_t, (int)ts[KPL - 1], 0x138, 0xF, 0xF, false);
        }
        // duplicates (every key against its predecessor)
        bool dq[KPL];
        bool anyd = false;
        {
          K kp = kp0;
#pragma unroll
          for (int q = 0; q < KPL; q++) {
            bool ok = true;
            if constexpr (EDGE) { const unsigned int idx = g0 + (unsigned int)q; ok = idx > rdlo && idx < rdhi; }
            dq[q] = ok && (kk[q] == kp);
            anyd |= dq[q];
            kp = kk[q];
          }
        }
        {
          const bool dl = __ballot(anyd) != 0ull;                    // y of a duplicate is not its index: its leaf goes to the exact kernels
          if constexpr (RAW) {
            const unsigned int slot = rpos((unsigned int)A + (unsigned int)(k * SLOT)) / (unsigned)SLOT;
            dupw = dl ? (dupw | (1u << slot)) : (dupw & ~(1u << slot));
          } else if (dl) {
            asm volatile("" ::: "memory");                          // (keep this a branch: as selects it costs every load 8 instructions)
#pragma unroll
            for (int q = 0; q < KPL; q++) if (dq[q]) xs[q] = __builtin_nan("");
          }
        }
        // the keys into the ring (16 bytes per store; the first entries once more behind the end)
        {
          const unsigned int off = rpos((unsigned int)g0);
          // (a load's 64 * KPL keys land on a 64 * KPL-aligned stretch of the ring: the mirror is hit by whole loads)
          const bool mirror = rpos((unsigned int)A + (unsigned int)(k * SLOT)) < (unsigned)S2_PAD && off < (unsigned)S2_PAD;   // (a load of 4-byte keys covers 256 positions, the mirror 128)
          if constexpr (RAW) {
            uint4 raw4;
            __builtin_memcpy(&raw4, kk, 16);
            *reinterpret_cast<uint4*>(&xring[off]) = raw4;
            if (mirror) *reinterpret_cast<uint4*>(&xring[off + RING]) = raw4;
          } else {
#pragma unroll
            for (int q = 0; q < KPL; q += 2) *reinterpret_cast<double2*>(&xring[off + q]) = make_double2(xs[q], xs[q + 1]);
            if (mirror) {
#pragma unroll
              for (int q = 0; q < KPL; q += 2) *reinterpret_cast<double2*>(&xring[off + RING + q]) = make_double2(xs[q], xs[q + 1]);
            }
          }
        }
        // does this lane hold a boundary?
        bool lane_b;
        if constexpr (EDGE) {
          lane_b = false;
#pragma unroll
          for (int q = 0; q < KPL; q++) {
            const unsigned int idx = g0 + (unsigned int)q;
            const unsigned int tprev = q == 0 ? tp0 : ts[q - 1];
            lane_b |= (idx > rdlo && idx < rdhi && ts[q] != tprev) || (idx == rdlo && idx == itlo) || (idx == rdhi);
          }
        } else if constexpr (FULL) {
          lane_b = ts[0] != tp0;
#pragma unroll
          for (int q = 1; q < KPL; q++) lane_b |= ts[q] != ts[q - 1];
        } else lane_b = ts[KPL - 1] != tp0;
        const unsigned long long bm = __ballot(lane_b);
        if constexpr (!EDGE) { LB[k] = bm; ck[k] = carry_key; }
        if (EDGE && bm) {
          if constexpr (LSUM) lst |= LS_SEEN;
          // ---- the boundary block (a load in three has one): per-key flags of the lanes that hold a boundary
          bool bq[KPL];
          unsigned int fq[KPL], told[KPL];
          int mine = 0;
          bool nonmono = false;
#pragma unroll
          for (int q = 0; q < KPL; q++) { bq[q] = false; fq[q] = 0u; }
          if (lane_b) {
            if constexpr (!FULL) {
#pragma unroll
              for (int q = 0; q < KPL - 1; q++) { bool oob; ts[q] = s2_target<ROOT, K>(r, Lm1f, Lm1, kk[q], KeyTraits<K>::as_float(kk[q]), oob); }
            }
            unsigned int tp = tp0;
            told[0] = tp0;
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              if (q > 0) told[q] = ts[q - 1];
              bool cmp_ok = true, fstart = false, fend = false;
              if constexpr (EDGE) {
                const unsigned int idx = g0 + (unsigned int)q;
                cmp_ok = idx > rdlo && idx < rdhi;
                fstart = (idx == rdlo && idx == itlo);              // the first key of the data: no previous key
                fend = (idx == rdhi);                               // the end of the (readable) data: no next key
              }
              nonmono |= cmp_ok && ts[q] < tp;
              bq[q] = (cmp_ok && ts[q] != tp) || fstart || fend;
              fq[q] = (((cmp_ok && tp < mid && ts[q] >= mid) || (fstart && ts[q] >= mid)) ? S2_SPLIT : 0u) | (fstart ? S2_START : 0u) | (fend ? S2_END : 0u);
              mine += bq[q] ? 1 : 0;
              tp = ts[q];
            }
          }
          if (nonmono) eflags |= EF_NON_MONOTONE;                   // two_layer.rs:50 / :144
          // ranks in index order: the boundaries of the lanes below, then this lane's in order
          int add = 0, rank = bcnt;
          {
            // (wave prefix sum of `mine` over the lanes of bm: boundaries per lane are almost always 0 or 1)
            int pre = 0;
            unsigned long long m1 = __ballot(mine >= 1);
            pre += __popcll(m1 & below); add += __popcll(m1);
#pragma unroll
            for (int c = 2; c <= KPL; c++) {
              const unsigned long long mc = __ballot(mine >= c);
              if (mc) { pre += __popcll(mc & below); add += __popcll(mc); }
            }
            rank += pre;
          }
          if (!dense && bcnt + add > BCAP) {
            // More boundaries than the list holds (leaves of a few keys): for the rest of this batch every leaf
            // goes straight to the exact kernels, the open one included; the list restarts with the next batch.
            dense = true;
            if (lane == 0 && bcnt > 0 && b_idx[bcnt - 1] >= c0u && b_idx[bcnt - 1] < c1u && !(b_fl[bcnt - 1] & S2_INHERIT)) sg.flist.push(b_t[bcnt - 1]);
          }
          if (lane_b) {
#pragma unroll
            for (int q = 0; q < KPL; q++) {
              if (bq[q]) {
                const unsigned int idx = g0 + (unsigned int)q;
                const bool own = idx >= c0u && idx < c1u && !(fq[q] & S2_END);     // this wave owns the leaf that starts here
                if (!dense) { b_idx[rank] = idx; b_t[rank] = ts[q]; b_fl[rank] = (unsigned char)fq[q]; }
                else if (own) sg.flist.push(ts[q]);
                rank++;
                if (own) {
                  // the keys on both sides of the boundary, for the widening of the two leaves (k_finalize reads
                  // them from here instead of gathering key[e] and key[s-1] from the key array)
                  // (the leaf that ends at a shard's first key belongs to the shard before: outside this launch's arrays)
                  if (!(fq[q] & S2_START)) { if (told[q] >= leaf_lo32) bnext[told[q]] = kk[q]; bprev[ts[q]] = q == 0 ? kp0 : kk[q > 0 ? q - 1 : 0]; }
                  leaf_start[ts[q]] = (unsigned long long)idx;
                  if (fq[q] & S2_SPLIT) {
                    st->split_idx = (unsigned long long)idx; st->split_target = ts[q];
                    if (idx == 0u || idx + 1u >= n32) eflags |= EF_DEGENERATE_SPLIT;   // two_layer.rs:27
                  }
                }
                if (idx >= c1u) stop = true;
              }
            }
          }
          if (!dense) bcnt += add;
          stop = __any(stop);
        }
        if (oobany) eflags |= EF_ROOT_OOB;                          // two_layer.rs:45-48
        // carries for the next load
        {
          const unsigned long long lastb = key_to_bits<K>(kk[KPL - 1]);
          const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)lastb, 63);
          unsigned int hi = 0u;
          if constexpr (sizeof(K) == 8) hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(lastb >> 32), 63);
          carry_key = bits_to_key<K>(((unsigned long long)hi << 32) | lo);
          carry_t = (unsigned int)__builtin_amdgcn_readlane((int)ts[KPL - 1], 63);
        }
      }
      if constexpr (!EDGE) {
        unsigned long long anyb = 0ull;
#pragma unroll
        for (int k = 0; k < NLOAD; k++) anyb |= LB[k];
        if (anyb) {
          if constexpr (LSUM) lst |= LS_SEEN;
          unsigned int pmask = 0u;                                  // the loads this lane holds a boundary in
#pragma unroll
          for (int k = 0; k < NLOAD; k++) pmask |= (unsigned int)((LB[k] >> lane) & 1ull) << k;
          // One pass handles a RANGE of consecutive loads in which no lane holds boundaries of two loads (ranks in index
          // order need every lane's count before its successors'): nearly always the whole batch; a lane with two such
          // loads (leaves of a few dozen keys) splits the batch into two ranges.
          K kprev[NLOAD];                                           // the key in front of the lane's first key, per load
#pragma unroll
          for (int k = 0; k < NLOAD; k++) {
            K kl[KPL];
            __builtin_memcpy(kl, &cur[k], 16);
            const unsigned long long lastb = key_to_bits<K>(kl[KPL - 1]), cb = key_to_bits<K>(ck[k]);
            const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)cb, (int)(unsigned int)lastb, 0x138, 0xF, 0xF, false);
            unsigned int hi = 0u;
            if constexpr (sizeof(K) == 8) hi = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)(cb >> 32), (int)(unsigned int)(lastb >> 32), 0x138, 0xF, 0xF, false);
            kprev[k] = bits_to_key<K>(((unsigned long long)hi << 32) | lo);
          }
          // (8-byte keys, ~190 keys per leaf: 4 % of the batches have such a lane -- one pass per load then, which keeps
          //  the common path free of the range bookkeeping; 4-byte keys: 16 keys per lane and batch, half of the batches)
          const bool multi = !RAW && __ballot((pmask & (pmask - 1u)) != 0u) != 0ull;
          for (int gs = 0; gs < NLOAD;) {
            int ge = gs;
            if constexpr (RAW) {
              unsigned long long gm = 0ull;
#pragma unroll
              for (int k = 0; k < NLOAD; k++) if (k == ge && k >= gs && (k == gs || !(LB[k] & gm))) { gm |= LB[k]; ge = k + 1; }
            } else ge = multi ? gs + 1 : NLOAD;
            const unsigned int pm = pmask & ((1u << ge) - 1u) & ~((1u << gs) - 1u);
            gs = ge;
            const bool active = pm != 0u;
            if (__ballot(active) == 0ull) continue;
            const unsigned int ksel = (unsigned int)__builtin_ctz(pm | (1u << (NLOAD - 1)));
            uint4 v = cur[0];
            K kp0 = kprev[0];
#pragma unroll
            for (int j = 1; j < NLOAD; j++) {
              // (opaque conditions: the compiler otherwise turns the select chain into a per-lane indexed read of `cur`
              //  from scratch memory, and the whole batch's registers go through scratch: 0.65 -> 0.84 ms)
              unsigned int kj = ksel ^ (unsigned int)j;
              asm volatile("" : "+v"(kj));
              if (kj == 0u) { v = cur[j]; kp0 = kprev[j]; }
            }
            K kk[KPL];
            __builtin_memcpy(kk, &v, 16);
            const unsigned int g0 = A + (ksel * 64u + (unsigned int)lane) * (unsigned int)KPL;
            unsigned int ts[KPL];
            bool oobp;
#pragma unroll
            for (int q = 0; q < KPL; q++) ts[q] = s2_target<ROOT, K>(r, Lm1f, Lm1, kk[q], KeyTraits<K>::as_float(kk[q]), oobp);
            const unsigned int tp0 = s2_target<ROOT, K>(r, Lm1f, Lm1, kp0, KeyTraits<K>::as_float(kp0), oobp);
            bool bq[KPL];
            unsigned int fq[KPL], told[KPL];
            int mine = 0;
            bool nonmono = false;
            {
              unsigned int tp = tp0;
#pragma unroll
              for (int q = 0; q < KPL; q++) {
                told[q] = tp;
                nonmono |= active && ts[q] < tp;
                bq[q] = active && ts[q] != tp;
                fq[q] = (tp < mid && ts[q] >= mid) ? S2_SPLIT : 0u;
                mine += bq[q] ? 1 : 0;
                tp = ts[q];
              }
            }
            if (nonmono) eflags |= EF_NON_MONOTONE;                 // two_layer.rs:50 / :144
            // ranks in index order: the loads before the lane's, the lanes below in its load, then its own in order
            int add = 0;
            unsigned int rank = (unsigned int)bcnt;
            {
              unsigned long long Kj[NLOAD], own = 0ull;
#pragma unroll
              for (int j = 0; j < NLOAD; j++) { Kj[j] = __ballot(active && ksel == (unsigned int)j); if (ksel == (unsigned int)j) own = Kj[j]; }
#pragma unroll
              for (int c = 1; c <= KPL; c++) {
                const unsigned long long mc = __ballot(mine >= c);
                if (c == 1 || mc) {
#pragma unroll
                  for (int j = 0; j < NLOAD; j++) {
                    const int w = __popcll(mc & Kj[j]);
                    add += w;
                    if (ksel > (unsigned int)j) rank += (unsigned int)w;
                  }
                  rank += (unsigned int)__popcll(mc & own & below);
                }
              }
            }
            if (!dense && bcnt + add > BCAP) {
              dense = true;
              if (lane == 0 && bcnt > 0 && b_idx[bcnt - 1] >= c0u && b_idx[bcnt - 1] < c1u && !(b_fl[bcnt - 1] & S2_INHERIT)) sg.flist.push(b_t[bcnt - 1]);
            }
            if (mine) {
#pragma unroll
              for (int q = 0; q < KPL; q++) {
                if (bq[q]) {
                  const unsigned int idx = g0 + (unsigned int)q;
                  const bool own_leaf = idx >= c0u && idx < c1u;    // this wave owns the leaf that starts here
                  if (!dense) { b_idx[rank] = idx; b_t[rank] = ts[q]; b_fl[rank] = (unsigned char)fq[q]; }
                  else if (own_leaf) sg.flist.push(ts[q]);
                  rank++;
                  if (own_leaf) {
                    if (told[q] >= leaf_lo32) bnext[told[q]] = kk[q];     // (the leaf before a shard's first key: not this launch's)
                    bprev[ts[q]] = q == 0 ? kp0 : kk[q > 0 ? q - 1 : 0];
                    leaf_start[ts[q]] = (unsigned long long)idx;
                    if (fq[q] & S2_SPLIT) {
                      st->split_idx = (unsigned long long)idx; st->split_target = ts[q];
                      if (idx == 0u || idx + 1u >= n32) eflags |= EF_DEGENERATE_SPLIT;   // two_layer.rs:27
                    }
                  }
                  if (idx >= c1u) stop = true;
                }
              }
            }
            if (!dense) bcnt += add;
            stop = __any(stop);
          }
        }
      }
    };
    if (interior) phase1(std::false_type{}); else phase1(std::true_type{});
    // The next batch overwrites the ring from (A + 2 BATCH - RING) down: whatever the oldest listed leaf still needs
    // (its container starts at s - 1) has to be processed first; an OPEN leaf that does not fit is long.
    const unsigned int bend = A + BATCH;
    const bool fits = bcnt == 0 || (bend + BATCH - (b_idx[0] - 1u) <= (unsigned int)RING);
    const bool giveup = LSUM && !stop && bi >= bi_giveup;           // (bend >= c1 + XWIN) the chunk rule: the leaf open at c1 is cut there
    if (sg.dbg & 1) { bcnt = bcnt > 0 ? 1 : 0; } else
    process(stop || giveup || dense || !fits, dense);
    if constexpr (!LSUM) {
      // (modes without partial sums: a long leaf is irregular, its owner walks to its end)
      if (bcnt > 0 && !(bend + BATCH - (b_idx[0] - 1u) <= (unsigned int)RING)) {
        if (lane == 0) b_fl[0] |= S2_LONG;
        wave_sync();
      }
    } else {
      if (bi == bi_inherit && !(lst & LS_SEEN)) {                   // (bend == c0 + XWIN) the chunk rule, seen from behind: an inherited stretch
        if (lane == 0) { b_idx[0] = c0u; b_t[0] = carry_t; b_fl[0] = (unsigned char)(S2_LONG | S2_INHERIT); }
        bcnt = 1; lst |= LS_OPEN | LS_SEEN;
        wave_sync();
        acc_start(c0u);
      }
      if (bcnt > 0 && !(lst & LS_OPEN) && !(bend + BATCH - (b_idx[0] - 1u) <= (unsigned int)RING)) {
        if (lane == 0) b_fl[0] |= S2_LONG;
        lst |= LS_OPEN;
        wave_sync();
        acc_start(b_idx[0]);
      }
      if (lst & LS_ACC) acc_run(bend < c1u ? bend : c1u);
      if (giveup) {
        if (bcnt > 0) {                                             // entry 0: the leaf open at c1 (it starts before c1)
          const unsigned int leaf = b_t[0], f0 = b_fl[0];
          if (!(lst & LS_ACC)) acc_start(b_idx[0]);
          acc_run(c1u); acc_emit(leaf);
          if (!(f0 & S2_INHERIT)) {
            if (lane == 0) l_buf[lcnt] = leaf | SG_TAG;
            lcnt++;
          }
        }
        stop = true;
      }
    }
  }
  wave_sync();
  if (LSUM && lane == 0) sg.rec_cnt[blockIdx.x] = acc_u[2] < sg.rpw ? acc_u[2] : sg.rpw;
  wave_sync();
  flush_exact();
  if (lane == 0 && guard_cnt) atomicAdd(&st->guard_count, (unsigned long long)guard_cnt);
  if (eflags) atomicOr(&st->err_flags, eflags);
}

}  // namespace rmi
