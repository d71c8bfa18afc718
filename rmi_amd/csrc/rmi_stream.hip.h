// rmi_stream.hip.h -- the streaming kernels of the leaf hot path (gfx950, wave64).
//
//   k_fit_stream  ("pass A")  bucketing scan + exact per-leaf SLR (two_layer.rs:43-90, linear.rs:12-59)
//   k_err_range   ("pass B")  last-level error pass + run lengths (two_layer.rs:207-217,
//                             lower_bound_correction.rs:104-119), boundaries from leaf_start
//   k_fit_long                leaves too long for the lockstep pass, one wave each
//
// Passes A and B share one skeleton.  A wave owns 64 consecutive chunks of C keys, one chunk per
// lane, and all lanes advance one key per step in lockstep.  Per panel of 64 rows x 16 keys:
//   load     coalesced, non-temporal HBM reads (16 bytes per lane for 8-byte keys: an instruction
//            covers 8 rows = 8 full 128-B lines), transposed through a padded LDS image (row stride
//            17 slots: lane-per-row accesses are bank-conflict free); the next panel is prefetched
//            into registers meanwhile.
//   phase 1  every lane classifies the 16 keys of its own row (root target, leaf-boundary and
//            duplicate-key bit masks, split point), converts them to f64 in place and leaves the
//            leaf ids in a second LDS panel.  This is the bucketing scan.
//   phase 2  16 lockstep steps at raised issue priority (s_setprio).  The common step is
//            straight-line code; steps in which some lane crosses a leaf boundary take a
//            wave-uniform slow path.
//
// Why lane-per-chunk: the SLR recurrence (linear.rs:24-34) is order dependent, so bit-identical
// coefficients need the reference order inside a leaf; the parallelism is across leaves, and
// chunks (not leaves) per lane keep all 64 lanes busy whatever the leaf sizes are.
#pragma once
#include <type_traits>

#include "rmi_device.hip.h"

namespace rmi {

constexpr int FS_ROW = 16;        // keys per panel row
constexpr int FS_STRIDE = 17;     // padded row stride (slots)
constexpr int FS_TMAX = 1024;     // reciprocal table size (beyond it the reciprocal is computed, see recip_exact)
constexpr unsigned long long FS_NO_NEXT = 1ull << 63;

struct SlrState { double mx, my, c, m2, nf; };

__device__ __forceinline__ void slr_push(SlrState& s, double x, double y) {   // linear.rs:25-32
  s.nf += 1.0;
  const double dx = x - s.mx;
  s.mx += dx / s.nf;
  s.my += (y - s.my) / s.nf;
  s.c += dx * (y - s.my);
  const double dx2 = x - s.mx;
  s.m2 += dx * dx2;
}

// RN(a / nf) for an integer-valued nf < 2^40, given r == RN(1 / nf)  [one Markstein round].
//   q0 = RN(a r) = t (1 + e1)(1 + e2), t = a / nf, |e1|, |e2| <= 2^-53
//   rem = a - q0 nf  is exact: it is a multiple of ulp(q0) and below 2 ulp(a) ~ 2 nf ulp(q0)
//   v = q0 + rem r = t + (t - q0) e1, so |v - t| <= |t| 2^-105, and the FMA returns RN(v).
// RN(v) == RN(t) unless a rounding boundary (a midpoint m of two doubles) lies between them.  But
// a - nf m is a non-zero multiple of half an ulp of t (a tie a == nf m would need more than 53
// significand bits), so |t - m| >= |t| 2^-53 / nf >> |t| 2^-105.  Hence the result is the
// correctly rounded quotient -- bit for bit what IEEE division returns.  No over/underflow can
// occur for the operands of the recurrence on integer keys (|a| in [2^-84, 2^65] or 0).
// rmi_hip_selftest_div checks it against `/` on random and near-midpoint operands.
__device__ __forceinline__ double div_by_count(double a, double nf, double r) {
  const double q = a * r;
  const double e = __builtin_fma(-q, nf, a);
  return __builtin_fma(e, r, q);
}

// RN(1 / nf) for an integer-valued nf in [1, 2^40): hardware estimate, Newton steps, and a last
// Markstein step  y + y (1 - nf y)  which rounds correctly once y is within an ulp (the only
// exception, a significand of all ones, cannot occur for nf < 2^53).  rmi_hip_selftest_recip
// checks it against 1.0 / nf exhaustively over ranges of nf.
__device__ __forceinline__ double recip_exact(double nf) {
  double y = __builtin_amdgcn_rcp(nf);
  double e = __builtin_fma(-nf, y, 1.0);
  y = __builtin_fma(e, y, y);
  e = __builtin_fma(-nf, y, 1.0);
  y = __builtin_fma(e, y, y);
  e = __builtin_fma(-nf, y, 1.0);
  return __builtin_fma(e, y, y);
}

// RN(a / nf) with a dependent chain of two instead of three operations, given the reciprocal of
// the integer nf < 2^40 as an unevaluated sum r + rl:  r = RN(1/nf),  rl = RN((1 - nf r) r)
// (1 - nf r is exact in an FMA), so 1/nf = (r + rl)(1 + O(2^-105)).  Then
//   a r + RN(a rl) = (a / nf)(1 + eta),  |eta| < 2^-103,
// and the FMA rounds that sum once.  As in div_by_count, a / nf is at least 2^-53 / nf > 2^-93
// (relative) away from every rounding boundary, so the result is the correctly rounded quotient.
__device__ __forceinline__ double recip_tail(double nf, double r) { return __builtin_fma(-nf, r, 1.0) * r; }
__device__ __forceinline__ double div_by_count2(double a, double r, double rl) { return __builtin_fma(a, r, a * rl); }

__device__ __forceinline__ void slr_push_r(SlrState& s, double x, double y, double r) {
  s.nf += 1.0;
  const double dx = x - s.mx;
  s.mx += div_by_count(dx, s.nf, r);
  s.my += div_by_count(y - s.my, s.nf, r);
  s.c += dx * (y - s.my);
  const double dx2 = x - s.mx;
  s.m2 += dx * dx2;
}

// OR of a 32-bit value over the 64 lanes of the wave, returned wave-uniform (an SGPR): DPP inside
// rows of 16, then across rows.
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int v) {
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8 -> lane 15 of each row holds the row's OR
  const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)v, 15), r1 = (unsigned int)__builtin_amdgcn_readlane((int)v, 31);
  const unsigned int r2 = (unsigned int)__builtin_amdgcn_readlane((int)v, 47), r3 = (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
  return r0 | r1 | r2 | r3;
}

// v_max_f64 / v_min_f64 as such.  Through fmax() / fmin() the compiler first canonicalises every
// operand it cannot prove quiet (`v_max_f64 x, x, x`: loop-carried maxima, kernel arguments) -- two
// extra f64 operations in a nine-operation step of pass B.  No signalling NaN can reach these
// operands; for quiet NaNs the instructions already are maxNum / minNum.
__device__ __forceinline__ double fmin_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmax_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmax_abs_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <typename K> struct UseRecipTable { static constexpr bool value = true; };
template <> struct UseRecipTable<double> { static constexpr bool value = false; };   // f64 keys: plain IEEE division

template <typename K>
__device__ __forceinline__ unsigned long long key_to_bits(K k) {
  if constexpr (sizeof(K) == 8) return __builtin_bit_cast(unsigned long long, k);
  else return (unsigned long long)__builtin_bit_cast(unsigned int, k);
}
template <typename K>
__device__ __forceinline__ K bits_to_key(unsigned long long b) {
  if constexpr (sizeof(K) == 8) return __builtin_bit_cast(K, b);
  else return __builtin_bit_cast(K, (unsigned int)b);
}

// Coalesced load of panel P of this wave (64 rows x 16 keys) into registers: a lane fetches V = 2
// consecutive keys of one row (16 bytes for 8-byte keys), LPR = 8 lanes cover a row (one full 128-B
// line for 8-byte keys), an instruction covers 8 rows, 8 instructions cover the panel.  Near the end of the readable keys the panel is fetched key by key instead, indices past
// the end clamped to n-1 (no branches around the loads); the duplicated key is masked out by the
// validity masks downstream.
template <typename K> struct PanelGeom {
  static constexpr int V = 2;                         // keys per lane per load: 16-byte loads for 8-byte keys, 8-byte
                                                      // loads for u32 (4 keys per lane spread an instruction over 16
                                                      // rows and cost the u32 configuration 30 % in pass B)
  static constexpr int LPR = FS_ROW / V;              // lanes per row == loads per panel
  static constexpr int RPI = 64 / LPR;                // rows per load instruction
};
template <typename K>
__device__ __forceinline__ void load_panel(K (&stage)[FS_ROW], const K* __restrict__ keys, uint64_t n,
                                           uint64_t wave_base, uint64_t C, uint64_t P, int lane) {
  using G = PanelGeom<K>;
  struct alignas(sizeof(K)) Vec { K v[G::V]; };         // the address is only key-aligned in a shard
  // wave-uniform base (SGPRs) + one 32-bit lane offset: no per-load 64-bit VGPR address math
  const uint64_t ubase = wave_base + P * FS_ROW;
  const unsigned int loff = (unsigned int)(lane / G::LPR) * (unsigned int)C + (unsigned int)(lane % G::LPR) * G::V;
  const uint64_t step = (uint64_t)G::RPI * C;
  if (wave_base + 63 * C + (P + 1) * FS_ROW <= n) {          // wave-uniform: whole panel in range
#pragma unroll
    for (int k = 0; k < G::LPR; k++) {
      const K* __restrict__ pk = keys + (ubase + (uint64_t)k * step);
      // streamed once per kernel: non-temporal (`nt`) loads -- the keys do not displace leaf_start /
      // params in the L2 (pass B -3 %, pass A -1 %, measured)
      typedef unsigned int raw_t __attribute__((ext_vector_type(sizeof(Vec) / 4), aligned(sizeof(K))));
      const raw_t rw = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(pk + loff));
      const Vec t = __builtin_bit_cast(Vec, rw);
#pragma unroll
      for (int q = 0; q < G::V; q++) stage[k * G::V + q] = t.v[q];
    }
  } else {
    const uint64_t last = n - 1;
#pragma unroll
    for (int k = 0; k < G::LPR; k++) {
#pragma unroll
      for (int q = 0; q < G::V; q++) {
        const uint64_t gi = ubase + (uint64_t)k * step + loff + q;
        stage[k * G::V + q] = keys[gi < last ? gi : last];
      }
    }
  }
}
template <typename K>
__device__ __forceinline__ void stage_to_lds(const K (&stage)[FS_ROW], unsigned long long* panel, int lane) {
  using G = PanelGeom<K>;
  const int base = (lane / G::LPR) * FS_STRIDE + (lane % G::LPR) * G::V;
#pragma unroll
  for (int k = 0; k < G::LPR; k++)
#pragma unroll
    for (int q = 0; q < G::V; q++) panel[base + k * G::RPI * FS_STRIDE + q] = key_to_bits<K>(stage[k * G::V + q]);
}

// Phase 1: classify the row of this lane -- straight-line code, no branches: all 16 raw keys are
// read first (one LDS wait), classified, and written back as f64 x plus leaf ids.  Keys past the
// end of the data are clamped copies of key[n-1]; `vmask` (valid positions) masks their bits.
// WRITE: pass A additionally publishes leaf_start / the split point / error flags.
// Leaf id of the key at row position s: from the leaf-id panel (radix roots: the id needs the raw
// key bits) or recomputed from the f64 x that phase 1 left in the key panel (float roots).
template <int ROOT, bool LEAFP>
__device__ __forceinline__ unsigned int leaf_id_at(const unsigned long long* __restrict__ panel,
                                                   const unsigned int* __restrict__ leafp, int lane, int s,
                                                   const RootP& r, double Lm1f) {
  if constexpr (LEAFP) return leafp[lane * FS_STRIDE + s];
  else {
    const double x = __builtin_bit_cast(double, panel[lane * FS_STRIDE + s]);
    return (unsigned int)fmin(fmax(0.0, floor(root_eval_f<ROOT>(r, x))), Lm1f);
  }
}

template <int ROOT, typename K, bool WRITE, bool LEAFP>
__device__ __forceinline__ void classify_row(unsigned long long* __restrict__ panel, unsigned int* __restrict__ leafp,
                                             int lane, const RootP& r, double Lm1f, double midf,
                                             uint64_t row_i, uint64_t n, unsigned int vmask, unsigned int ownmask,
                                             K& kprev, double& tprev, unsigned int& bmask, unsigned int& dmask,
                                             int& split_pos, unsigned int& flags,
                                             unsigned long long* __restrict__ leaf_start, DevState* __restrict__ st) {
  // Per key: convert, evaluate the root, two compares.  Everything that only matters at a leaf
  // boundary (split point, monotonicity, leaf_start) is derived afterwards from the boundary bits.
  // Positions past the readable end hold clamped copies of the last key, so the carries below are
  // right without per-key selects.
  unsigned int bm = 0, dm = 0, oobm = 0;
  const double tin = tprev;
  double tp = tprev;
  K kp = kprev;
#pragma unroll
  for (int h = 0; h < FS_ROW; h += 8) {                    // two halves of 8 keys: bounded register pressure
    K kk[8];
#pragma unroll
    for (int q = 0; q < 8; q++) kk[q] = bits_to_key<K>(panel[lane * FS_STRIDE + h + q]);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int s = h + q;
      const K k = kk[q];
      bool oob;
      const double t = root_target_f<ROOT, K>(r, Lm1f, k, oob);
      bm |= (t != tp) ? (1u << s) : 0u;
      dm |= (k == kp) ? (1u << s) : 0u;
      if constexpr (WRITE && !root_needs_bounds_check<ROOT>()) oobm |= oob ? (1u << s) : 0u;   // two_layer.rs:45-48
      if constexpr (LEAFP) leafp[lane * FS_STRIDE + s] = (unsigned int)t;
      panel[lane * FS_STRIDE + s] = __builtin_bit_cast(unsigned long long, KeyTraits<K>::as_float(k));
      tp = t; kp = k;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (row_i == 0) dm &= ~1u;                               // key 0 has no predecessor
  bm &= vmask; dm &= vmask;
  bmask = bm; dmask = dm;
  kprev = kp; tprev = tp;
  split_pos = -1;
  if constexpr (WRITE) {
    if (oobm & vmask) flags |= EF_ROOT_OOB;
    // walk the boundaries of this row in order (targets only change there)
    const unsigned int midu = (unsigned int)midf;
    long long tc = (tin < 0.0) ? -1ll : (long long)(unsigned int)tin;   // target before the row (-1: none)
    unsigned int m = bm;
    while (m) {
      const int s = __ffs(m) - 1;
      m &= m - 1;
      const uint64_t idx = row_i + s;
      const unsigned int t = leaf_id_at<ROOT, LEAFP>(panel, leafp, lane, s, r, Lm1f);
      if ((long long)t < tc) flags |= EF_NON_MONOTONE;     // two_layer.rs:50 / :144
      const bool is_split = (tc < (long long)midu && t >= midu);       // idx == split_idx (two_layer.rs:132-136)
      if (is_split) split_pos = s;
      if ((ownmask >> s) & 1u) {                           // this lane owns the leaf starting here
        leaf_start[t] = idx;
        if (is_split) {
          if (idx == 0 || idx + 1 >= n) flags |= EF_DEGENERATE_SPLIT;   // two_layer.rs:27
          st->split_idx = idx;
          st->split_target = t;
        }
      }
      tc = (long long)t;
    }
    if (n - 1 >= row_i && n - 1 - row_i < (uint64_t)FS_ROW && ((vmask >> (int)(n - 1 - row_i)) & 1u))
      st->last_target = leaf_id_at<ROOT, LEAFP>(panel, leafp, lane, (int)(n - 1 - row_i), r, Lm1f);
  }
}

// =============================================================================================
// Pass A.  Block = 4 independent waves (they only share the reciprocal table).
// =============================================================================================
constexpr int FA_WAVES = 4;
constexpr int FS_QDRAIN = 64;     // drain the close queue in full batches of 64 leaves (every lane busy)
constexpr int FS_QCAP2 = FS_QDRAIN + 64;

template <int ROOT, typename K>
__global__ void __launch_bounds__(64 * FA_WAVES) k_fit_stream(const K* __restrict__ keys, Span sp, RootP r, uint64_t C,
                                                             unsigned long long* __restrict__ leaf_start,
                                                             double* __restrict__ params,
                                                             DevState* __restrict__ st,
                                                             unsigned long long* __restrict__ long_idx,
                                                             unsigned int long_min) {
  __shared__ unsigned long long s_panel[FA_WAVES][64 * FS_STRIDE];   // raw key bits, then f64 x
  constexpr bool LEAFP = (ROOT == K_RADIX || ROOT == K_RADIX_TABLE);
  __shared__ unsigned int s_leafp[FA_WAVES][LEAFP ? 64 * FS_STRIDE : 1];   // leaf ids (radix roots only)
  __shared__ double rtab[FS_TMAX];
  __shared__ double s_q[FA_WAVES][5][FS_QCAP2];
  __shared__ unsigned long long s_qidx[FA_WAVES][FS_QCAP2];

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  unsigned long long* panel = s_panel[wv];
  unsigned int* leafp = s_leafp[wv];
  double* q_mx = s_q[wv][0]; double* q_my = s_q[wv][1]; double* q_c = s_q[wv][2];
  double* q_m2 = s_q[wv][3]; double* q_nf = s_q[wv][4];
  unsigned long long* q_idx = s_qidx[wv];

  const uint64_t n = sp.n;                                   // global key count
  const uint64_t rd_hi = sp.rd_hi;                           // one past the last readable key
  const uint64_t wave_base = sp.it_lo + ((uint64_t)blockIdx.x * FA_WAVES + wv) * 64 * C;
  const uint64_t p0 = wave_base + (uint64_t)lane * C;         // first key of this lane's chunk
  const uint64_t chunk_end = (p0 + C < sp.it_hi) ? p0 + C : sp.it_hi;
  const double Lm1f = (double)(r.L - 1);
  const double midf = (double)(r.L / 2);                     // two_layer.rs:131

  for (int q = threadIdx.x; q < FS_TMAX; q += 64 * FA_WAVES) rtab[q] = 1.0 / (double)(q > 0 ? q : 1);
  __syncthreads();

  // carries of the classification (phase 1)
  K kprev = K();
  double tprev = -1.0;
  bool carry_split = false;                                  // the key just before the row was the split key
  // carries of the recurrence (phase 2)
  double xprev = 0.0, yprev = 0.0;
  // Which lanes have a leaf open: kept as an explicit lane mask in an SGPR pair and updated with
  // scalar instructions at the (wave-uniform) boundary steps; predicates are formed from it with
  // inverse_ballot, which costs nothing.  (A per-lane bool would be carried in a VGPR through the
  // divergent boundary code and re-tested with vector compares at every step.)
  unsigned long long active_m = 0;
  unsigned int roff = 8;                                     // byte offset of 1/(count+1) in the reciprocal table
  SlrState sl = {0.0, 0.0, 0.0, 0.0, 0.0};
  unsigned int flags = 0;
  if (p0 < sp.it_hi && p0 > sp.rd_lo) {
    bool oob;
    kprev = keys[p0 - 1];
    xprev = KeyTraits<K>::as_float(kprev);
    yprev = (double)first_occurrence(keys, p0 - 1, sp.rd_lo);
    tprev = root_target_f<ROOT, K>(r, Lm1f, kprev, oob);
    if (p0 > sp.rd_lo + 1) {
      const double tpp = root_target_f<ROOT, K>(r, Lm1f, keys[p0 - 2], oob);
      carry_split = (tpp < midf && tprev >= midf);
    }
  }
  int pending = 0;                                           // wave-uniform

  // drain: every lane finishes one queued leaf (extra points + final divisions)
  // (full batches of 64 only, so that every lane has a leaf to finish; `all`: the last call)
  auto drain = [&](bool all = false) {
    const int todo = all ? pending : (pending & ~63);
    for (int b = 0; b < todo; b += 64) {
      const int slot = b + lane;
      if (slot < todo) {
        SlrState s2 = {q_mx[slot], q_my[slot], q_c[slot], q_m2[slot], q_nf[slot]};
        const unsigned long long qi = q_idx[slot];
        const uint64_t bi = qi & ~FS_NO_NEXT;                // the boundary index; the leaf is that of key[bi-1]
        bool oob_;
        const uint64_t lj = (uint64_t)root_target_f<ROOT, K>(r, Lm1f, keys[bi - 1], oob_);
        double a = 0.0, be = 0.0;
        bool have = true;
        if (!(qi & FS_NO_NEXT)) {
          // next-first point (two_layer.rs:58-59): first key of the next non-empty leaf, y == its index
          const double x = KeyTraits<K>::as_float(keys[bi]);
          const double y = (double)bi;
          slr_push(s2, x, y);
          slr_push(s2, x, y);                                // Q1 tail duplicate (models/mod.rs:180)
        } else if (s2.nf > 0.0) {
          // container ends with the leaf's own last key: duplicate that one
          const double x = KeyTraits<K>::as_float(keys[bi - 1]);
          const double y = (double)first_occurrence(keys, bi - 1, sp.rd_lo);
          slr_push(s2, x, y);
        } else have = false;                                 // only reachable together with a degenerate split
        if (have) {
          const double cov = s2.c / (s2.nf - 1.0);           // linear.rs:46-47
          const double var = s2.m2 / (s2.nf - 1.0);
          if (!(var >= 0.0)) flags |= EF_NEG_VARIANCE;
          if (var == 0.0) { a = s2.my; be = 0.0; }
          else { be = cov / var; a = s2.my - be * s2.mx; }
        }
        params[lj * 2 + 0] = a;
        params[lj * 2 + 1] = be;
      }
    }
    // the incomplete batch stays queued, moved to the front
    const int rem = pending - todo;
    if (rem > 0 && todo > 0) {
      const bool mv = lane < rem;
      double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0; unsigned long long ti = 0;
      if (mv) { t0 = q_mx[todo + lane]; t1 = q_my[todo + lane]; t2 = q_c[todo + lane]; t3 = q_m2[todo + lane]; t4 = q_nf[todo + lane]; ti = q_idx[todo + lane]; }
      if (mv) { q_mx[lane] = t0; q_my[lane] = t1; q_c[lane] = t2; q_m2[lane] = t3; q_nf[lane] = t4; q_idx[lane] = ti; }
    }
    pending = rem;
  };

  K stage[FS_ROW];
  uint64_t row_i = p0;                                       // index of the first key of the current row
  double row_if = (double)p0;
  bool lane_done = !(p0 < sp.it_hi);
  uint64_t P = 0;
  load_panel<K>(stage, keys, rd_hi, wave_base, C, 0, lane);
  while (__any(!lane_done)) {
    stage_to_lds<K>(stage, panel, lane);
    load_panel<K>(stage, keys, rd_hi, wave_base, C, P + 1, lane);   // prefetch (consumed next iteration)

    // ---------------- phase 1 ----------------
    unsigned int bmask = 0, dmask = 0;
    int split_pos = -1;
    const int end_pos = (row_i >= rd_hi) ? 0 : ((rd_hi - row_i < (uint64_t)FS_ROW) ? (int)(rd_hi - row_i) : FS_ROW);
    const int own_cnt = (row_i >= chunk_end) ? 0 : ((chunk_end - row_i < (uint64_t)FS_ROW) ? (int)(chunk_end - row_i) : FS_ROW);
    const bool prev_split_in = carry_split;
    {
      const unsigned int vmask = lane_done ? 0u : ((1u << end_pos) - 1u);
      const unsigned int ownmask = (1u << own_cnt) - 1u;
      classify_row<ROOT, K, true, LEAFP>(panel, leafp, lane, r, Lm1f, midf, row_i, n, vmask, ownmask,
                                  kprev, tprev, bmask, dmask, split_pos, flags, leaf_start, st);

      if (!lane_done && end_pos < FS_ROW) bmask |= 1u << end_pos;   // end of data acts as a final boundary
      carry_split = (split_pos == FS_ROW - 1);
    }

    // ---------------- phase 2: 16 lockstep steps of the recurrence ----------------
    // Two instances of the step loop.  FAST (the usual one): divisions through the reciprocal
    // table, and at most one leaf closes per lane in this row, so the queue slots are assigned
    // once per panel and the loop body has no cross-lane operation besides the boundary vote.
    // GENERAL: plain IEEE division (running counts beyond the table) and/or several closes per lane
    // (leaves shorter than a row); slots by ballot, drain inside the loop.  Same results.
    // Counts that may leave the reciprocal table during this row (wave-uniform, rare): the row's
    // reciprocals are computed instead.  And a leaf of more than `long_min` points leaves the
    // lockstep pass altogether (its lane would walk far beyond its chunk, alone, at the cost of a
    // whole wave): the lane hands it over to k_fit_long (by the index of its previous key) and goes
    // on looking for the next leaf start.
    bool active = __builtin_amdgcn_inverse_ballot_w64(active_m);
    bool beyond = __any(active && (roff >> 3) + FS_ROW + 2 >= (unsigned)FS_TMAX);
    if (beyond) {
      const bool hand_over = active && (roff >> 3) >= long_min;
      active_m &= ~__ballot(hand_over);
      if (hand_over) {
        const unsigned long long pos = atomicAdd(&st->long_count, 1ull);
        if (pos < st->long_cap) long_idx[pos] = row_i - 1;
      }
      active = __builtin_amdgcn_inverse_ballot_w64(active_m);
      beyond = __any(active && (roff >> 3) + FS_ROW + 2 >= (unsigned)FS_TMAX);
    }
    // steps at which some lane crosses a leaf boundary, wave-uniform: the step loop tests a scalar bit
    const unsigned int any_mask = wave_or_u32(bmask);
    int my_closes = 0;
    {
      bool act = active;
      unsigned int m = bmask;
      while (m) {
        const int s = __ffs(m) - 1;
        m &= m - 1;
        if (act) my_closes++;
        act = (s != end_pos) && (s < own_cnt);
      }
    }
    const bool general = !UseRecipTable<K>::value || __any(my_closes > 1);
    auto steps = [&](auto fast_tag, auto nodup_tag, auto tab_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      constexpr bool NODUP = decltype(nodup_tag)::value;   // no duplicate key in this panel: y == index
      constexpr bool TAB = decltype(tab_tag)::value;       // 1/(count+1) from the LDS table
      int my_slot = 0;
      if constexpr (FAST) {
        const unsigned long long cmask = __ballot(my_closes == 1);
        my_slot = pending + __popcll(cmask & ((1ull << lane) - 1ull));
        pending += __popcll(cmask);
      }
      double xn = __builtin_bit_cast(double, panel[lane * FS_STRIDE]);
      auto one = [&](int s) {
        const double x = xn;
        xn = __builtin_bit_cast(double, panel[lane * FS_STRIDE + ((s + 1) & (FS_ROW - 1))]);   // next step's x
        double rr = 0.0;
        // (TAB: the counts of this row stay inside the table, see `beyond`: no wrap-around mask needed)
        if constexpr (FAST && TAB) rr = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(rtab) + roff);
        if constexpr (FAST && !TAB) rr = recip_exact(sl.nf + 1.0);
        const bool bit = (bmask >> s) & 1u;
        const double idxf = row_if + (double)s;
        double y = idxf;                                     // FixDups first-occurrence offset
        if constexpr (!NODUP) y = ((dmask >> s) & 1u) ? yprev : idxf;
        unsigned long long push_m = active_m;               // lanes that consume this key
        if ((any_mask >> s) & 1u) {
          // close first (queue the running state of the leaf that ends here), then open.  All the
          // conditions are lane masks built from one vector compare each and combined with scalar
          // instructions; `sb` pins the index arithmetic inside this (rare) block.
          int sb = s;
          asm volatile("" : "+s"(sb));
          const uint64_t idx = row_i + (uint64_t)sb;
          const unsigned long long bit_m = __ballot(bit);
          const unsigned long long end_m = __ballot(sb == end_pos);
          const unsigned long long split_m = __ballot(sb == split_pos);
          const unsigned long long close_m = bit_m & active_m;
          const bool do_close = __builtin_amdgcn_inverse_ballot_w64(close_m);
          int slot = my_slot;
          if constexpr (!FAST) {
            slot = pending + __popcll(close_m & ((1ull << lane) - 1ull));
            pending += __popcll(close_m);
          }
          if (do_close) {
            q_mx[slot] = sl.mx; q_my[slot] = sl.my; q_c[slot] = sl.c; q_m2[slot] = sl.m2; q_nf[slot] = sl.nf;
            // Q3: no next-first across the halves / at the end
            q_idx[slot] = __builtin_amdgcn_inverse_ballot_w64(end_m | split_m) ? (idx | FS_NO_NEXT) : idx;
          }
          // open the leaf that starts here: prev-last point (two_layer.rs:74-78) unless at the start
          // of a half / after Q4 (Q3/Q4); Q2: the key at split_idx is in neither half.
          const unsigned long long open_m = bit_m & ~end_m & __ballot(sb < own_cnt);
          const unsigned long long psplit_m = (sb == 0) ? __ballot(prev_split_in) : __ballot(split_pos == sb - 1);
          const unsigned long long wp_m = ~(split_m | __ballot(idx == 0) | psplit_m);
          double yp = yprev;                                   // y of the previous key
          if constexpr (NODUP) { if (sb != 0) yp = idxf - 1.0; }
          if (__builtin_amdgcn_inverse_ballot_w64(open_m)) {
            // Re-initialise the running state in place, under the execution mask of the opening
            // lanes (the state of the others is not touched, so nothing is selected or copied):
            // w = 1 with a prev-last point (count 1, means = that point), else 0 (all zero).
            // The asm operands are tied ("+v") so that the values stay in their registers.
            const bool with_prev = __builtin_amdgcn_inverse_ballot_w64(wp_m);
            const double w = with_prev ? 1.0 : 0.0;
            const unsigned int ro = with_prev ? 16u : 8u;
            asm("v_mul_f64 %0, %1, %2" : "+v"(sl.mx) : "v"(xprev), "v"(w));          // (keys are >= 0 or w == 1 / +-0 alike)
            asm("v_mul_f64 %0, %1, %2" : "+v"(sl.my) : "v"(yp), "v"(w));
            asm("v_mov_b64 %0, 0" : "+v"(sl.c));
            asm("v_mov_b64 %0, 0" : "+v"(sl.m2));
            asm("v_mov_b64 %0, %1" : "+v"(sl.nf) : "v"(w));
            asm("v_mov_b32 %0, %1" : "+v"(roff) : "v"(ro));
            asm("v_fma_f64 %0, %1, -0.5, 1.0" : "+v"(rr) : "v"(w));               // 1/(cnt+1): 0.5 or 1.0
          }
          // end of data / the next lane takes over: inactive.  Q2: the key at split_idx is not consumed.
          active_m = (active_m & ~bit_m) | open_m;
          push_m = (push_m & ~bit_m) | (open_m & ~split_m);
        }
        if (__builtin_amdgcn_inverse_ballot_w64(push_m)) {
          roff += 8;
          if constexpr (FAST) slr_push_r(sl, x, y, rr);
          else slr_push(sl, x, y);
        }
        xprev = x;
        if constexpr (!NODUP) yprev = y;
        if constexpr (!FAST) { if (pending >= FS_QDRAIN) drain(); }
      };
      if constexpr (FAST && TAB) {                           // the usual instances: fully unrolled (constant LDS offsets, no loop state)
#pragma unroll
        for (int s = 0; s < FS_ROW; s++) one(s);
      } else {
#pragma unroll 2
        for (int s = 0; s < FS_ROW; s++) one(s);
      }
      if constexpr (NODUP) yprev = row_if + (double)(FS_ROW - 1);
    };
    // The step loop is the dependent chain of the recurrence; the other wave of the SIMD is usually
    // in another phase (loads, staging, classification: independent instructions).  Raised issue
    // priority for the chain shortens it and lets the other wave fill the gaps: pass A -4.5 %.
    __builtin_amdgcn_s_setprio(2);
    if (general) steps(std::false_type{}, std::false_type{}, std::false_type{});
    else if (beyond) steps(std::true_type{}, std::false_type{}, std::false_type{});
    else if (!__any(dmask != 0u)) steps(std::true_type{}, std::true_type{}, std::true_type{});
    else steps(std::true_type{}, std::false_type{}, std::true_type{});
    __builtin_amdgcn_s_setprio(0);
    if (pending >= FS_QDRAIN) drain();
    row_i += FS_ROW;
    row_if += (double)FS_ROW;
    if (!__builtin_amdgcn_inverse_ballot_w64(active_m) && (row_i >= chunk_end || row_i >= rd_hi)) lane_done = true;
    P += 1;
  }
  if (pending) drain(true);
  if (flags) atomicOr(&st->err_flags, flags);
}

// =============================================================================================
// Pass B driven by leaf_start ("k_err_range").  After pass A and the fill, every leaf boundary is
// known, so pass B needs no root evaluation per key: a lane walks its chunk, compares the running
// index with the end of its current leaf, and at a boundary switches to the next leaf whose
// parameters and end index were prefetched when the previous leaf was entered.  Phase 1 shrinks to
// "convert 16 keys to f64 + duplicate mask".  One atomicMax per (lane, leaf) segment.
// =============================================================================================
template <typename K>
__device__ __forceinline__ void convert_row(unsigned long long* __restrict__ panel, int lane, uint64_t row_i,
                                            K& kprev, unsigned int& dmask) {
  unsigned int dm = 0;
  K kp = kprev;
#pragma unroll
  for (int h = 0; h < FS_ROW; h += 8) {
    K kk[8];
#pragma unroll
    for (int q = 0; q < 8; q++) kk[q] = bits_to_key<K>(panel[lane * FS_STRIDE + h + q]);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const K k = kk[q];
      dm |= (k == kp) ? (1u << (h + q)) : 0u;
      panel[lane * FS_STRIDE + h + q] = __builtin_bit_cast(unsigned long long, KeyTraits<K>::as_float(k));
      kp = k;
    }
  }
  if (row_i == 0) dm &= ~1u;                               // key 0 has no predecessor
  dmask = dm;
  kprev = kp;
}

// Result queue of a wave: (leaf, max error, longest run) per (lane, leaf) segment.  4 waves/SIMD
// leave 1536 B of LDS per wave for it.  Below 2^32 keys the two maxima fit 32 bits: 128 entries,
// drained in bursts of >= 64; otherwise 72 entries of f64 maxima (bursts of >= 8).  At most 64
// entries arrive per panel in the fast loop.
constexpr int ER_QWORDS = 384;
      // result queue per wave (drained in bursts of >= 64)

template <int ROOT, int LEAF, typename K>
__global__ void __launch_bounds__(64, 4) k_err_range(const K* __restrict__ keys, Span sp, RootP r, uint64_t C,
                                                  const unsigned long long* __restrict__ leaf_start,
                                                  const double* __restrict__ params,
                                                  unsigned long long* __restrict__ leaf_maxerr,
                                                  unsigned long long* __restrict__ leaf_run) {
  constexpr int PPL = (LEAF == K_CUBIC) ? 4 : 2;
  __shared__ unsigned long long panel[64 * FS_STRIDE];
  __shared__ alignas(8) unsigned int qraw[ER_QWORDS];
  const bool q32 = sp.n < (1ull << 32);
  const int q_cap = q32 ? 128 : 72, q_drain = q_cap - 64;
  unsigned int* q_leaf = qraw;
  unsigned int* q_err32 = qraw + 128; unsigned int* q_run32 = qraw + 256;
  double* q_err64 = reinterpret_cast<double*>(qraw + 72); double* q_run64 = reinterpret_cast<double*>(qraw + 72 + 144);

  const int lane = threadIdx.x;
  const uint64_t n = sp.n;
  const uint64_t wave_base = sp.it_lo + (uint64_t)blockIdx.x * 64 * C;
  const uint64_t p0 = wave_base + (uint64_t)lane * C;
  const uint64_t chunk_end = (p0 + C < sp.it_hi) ? p0 + C : sp.it_hi;
  const double Lm1f = (double)(r.L - 1);
  const double nf = (double)n;
  const unsigned int leaf_last = (unsigned int)(sp.leaf_hi - 1);

  K kprev = K();
  double yprev = 0.0;
  unsigned int cur_leaf = 0;
  bool have_leaf = false;                                    // cur_leaf is a leaf of this launch
  double e_cur = (double)p0;                                 // index at which the current leaf ends
  double e_next = 0.0;                                       // end of leaf cur_leaf + 1 (prefetched)
  // Prefetch of (params, end) of leaf cur_leaf+1 is issued at a panel top, right before the key
  // prefetch, so that the one vmcnt wait per panel covers it: need -> inflight -> ok.
  bool pn_need = false, pn_inflight = false, pn_ok = false;
  double pa[PPL], pn[PPL], pn_l[PPL];
  unsigned long long e_next_l = 0;
#pragma unroll
  for (int q = 0; q < PPL; q++) { pa[q] = 0.0; pn[q] = 0.0; pn_l[q] = 0.0; }
  double maxerr = 0.0, maxrun = 0.0;
  if (p0 < sp.it_hi && p0 > sp.rd_lo) {
    bool oob;
    kprev = keys[p0 - 1];
    yprev = (double)first_occurrence(keys, p0 - 1, sp.rd_lo);
    const unsigned int tl = (unsigned int)root_target_f<ROOT, K>(r, Lm1f, kprev, oob);   // owner of the run ending at p0-1
    if (tl >= sp.leaf_lo && tl < sp.leaf_hi) {               // (the halo key before a shard belongs to another rank)
      cur_leaf = tl; have_leaf = true;
      e_cur = (double)leaf_start[tl + 1];
#pragma unroll
      for (int q = 0; q < PPL; q++) pa[q] = params[(uint64_t)tl * PPL + q];
      pn_need = tl < leaf_last;
    }
  }
  int pending = 0;                                           // wave-uniform

  auto drain = [&]() {
    for (int b = 0; b < pending; b += 64) {
      const int slot = b + lane;
      if (slot < pending) {
        const unsigned int lj = q_leaf[slot];
        const unsigned long long e = q32 ? (unsigned long long)q_err32[slot] : (unsigned long long)q_err64[slot];
        const unsigned long long rn = q32 ? (unsigned long long)q_run32[slot] : (unsigned long long)q_run64[slot];
        if (e > 0) atomicMax(&leaf_maxerr[lj], e);
        if (rn > 1) atomicMax(&leaf_run[lj], rn);
      }
    }
    pending = 0;
  };
  // queue the maxima of the leaf this lane leaves (wave-uniform bookkeeping)
  auto flush = [&](bool leaving) {
    const bool push = leaving && have_leaf && (maxerr > 0.0 || maxrun > 1.0);
    const unsigned long long pm = __ballot(push);
    if (pm) {
      if (push) {
        const int slot = pending + __popcll(pm & ((1ull << lane) - 1ull));
        q_leaf[slot] = cur_leaf;
        if (q32) { q_err32[slot] = (unsigned int)maxerr; q_run32[slot] = (unsigned int)maxrun; }
        else { q_err64[slot] = maxerr; q_run64[slot] = maxrun; }
      }
      pending += __popcll(pm);
    }
  };

  K stage[FS_ROW];
  uint64_t row_i = p0;
  double row_if = (double)p0;
  bool lane_done = !(p0 < sp.it_hi);
  uint64_t P = 0;
  load_panel<K>(stage, keys, sp.rd_hi, wave_base, C, 0, lane);
  while (__any(!lane_done)) {
    stage_to_lds<K>(stage, panel, lane);                     // (waits for every outstanding load)
    // the prefetch issued one panel ago has landed (same wait): move it to plain registers so that
    // the step loop never waits on the vector-memory counter
    if (pn_inflight) {
#pragma unroll
      for (int q = 0; q < PPL; q++) pn[q] = pn_l[q];
      e_next = (double)e_next_l;
      pn_ok = true; pn_inflight = false;
    }
    if (pn_need) {
#pragma unroll
      for (int q = 0; q < PPL; q++) pn_l[q] = params[(uint64_t)(cur_leaf + 1) * PPL + q];
      e_next_l = leaf_start[cur_leaf + 2];
      pn_need = false; pn_inflight = true;
    }
    load_panel<K>(stage, keys, sp.rd_hi, wave_base, C, P + 1, lane);

    unsigned int dmask = 0;
    convert_row<K>(panel, lane, row_i, kprev, dmask);
    const int end_pos = (row_i >= chunk_end) ? 0 : ((chunk_end - row_i < (uint64_t)FS_ROW) ? (int)(chunk_end - row_i) : FS_ROW);
    const unsigned int vmask = lane_done ? 0u : ((1u << end_pos) - 1u);   // keys of this lane's chunk in the row

    // Will some lane cross a boundary in this row for which the prefetched next leaf cannot be used
    // (not landed yet, next leaf empty, or a second boundary inside the row)?  Then the whole wave
    // takes the general step loop for this panel; otherwise the fast one, which contains no
    // vector-memory instruction at all (so the compiler places no vmcnt wait in it).
    const double row_endf = row_if + (double)end_pos;
    const bool crosses = !lane_done && end_pos > 0 && e_cur < row_endf;
    const bool general = crosses && !(have_leaf && pn_ok && e_next > e_cur && e_next >= row_endf);
    // PLAIN: every lane has a full row, no duplicate keys in the panel and no open duplicate run:
    // y is the running index and run lengths are all 1 (never reported), so the step is just
    // "predict, compare, max".
    const bool plain_ok = !__any(lane_done || end_pos < FS_ROW || dmask != 0u || (row_i > sp.rd_lo && yprev != row_if - 1.0));
    // fast loop: a lane crosses at most once in this row, at a position known now -> the steps at
    // which some lane crosses are a wave-uniform mask and the loop tests a scalar bit
    const int cross_pos = crosses ? (int)(e_cur - row_if) : -1;           // e_cur in (row_if, row_endf): exact
    const unsigned int cross_any = wave_or_u32((crosses && cross_pos >= 0 && cross_pos < FS_ROW) ? (1u << cross_pos) : 0u);
    auto steps = [&](auto fast_tag, auto plain_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      constexpr bool PLAIN = decltype(plain_tag)::value;
      double xn = __builtin_bit_cast(double, panel[lane * FS_STRIDE]);
      auto one = [&](int s) {
        const double x = xn;
        xn = __builtin_bit_cast(double, panel[lane * FS_STRIDE + ((s + 1) & (FS_ROW - 1))]);
        const double idxf = row_if + (double)s;
        bool valid = true, dup = false;
        if constexpr (!PLAIN) {
          valid = (vmask >> s) & 1u;
          dup = (dmask >> s) & 1u;
          // a new key value ends the previous run: record its length for the leaf of the previous key
          if (valid && !dup && have_leaf) maxrun = fmax_raw(maxrun, idxf - yprev);
        }
        const double y = dup ? yprev : idxf;
        bool bit, some;
        if constexpr (FAST) { bit = (s == cross_pos); some = (cross_any >> s) & 1u; }
        else { bit = valid && (idxf >= e_cur); some = __any(bit); }           // first key of another leaf
        if (some) {
          flush(bit);
          if (bit) {
            bool use_prefetch = true;
            if constexpr (!FAST) use_prefetch = have_leaf && pn_ok && e_next > idxf;
            if (use_prefetch) {
              // the next leaf is not empty: the prefetched parameters and end index are the ones
              cur_leaf += 1;
#pragma unroll
              for (int q = 0; q < PPL; q++) pa[q] = pn[q];
              e_cur = e_next;
            } else {
              if constexpr (!FAST) {
                // empty leaves in between, first leaf of this lane, or leaves shorter than two rows
                bool oob;
                cur_leaf = (unsigned int)root_target_f<ROOT, K>(r, Lm1f, keys[row_i + s], oob);
#pragma unroll
                for (int q = 0; q < PPL; q++) pa[q] = params[(uint64_t)cur_leaf * PPL + q];
                e_cur = (double)leaf_start[cur_leaf + 1];
              }
            }
            have_leaf = true; maxerr = 0.0; maxrun = 0.0;
            pn_ok = false; pn_inflight = false;
            pn_need = cur_leaf < leaf_last;
          }
          if constexpr (!FAST) { if (pending >= q_drain) drain(); }
        }
        if (valid) {
          double f;
          if constexpr (LEAF == K_CUBIC) f = __builtin_fma(__builtin_fma(__builtin_fma(pa[0], x, pa[1]), x, pa[2]), x, pa[3]);
          else f = __builtin_fma(pa[1], x, pa[0]);
          // err in the f64 domain (all integers < 2^53): |min(pred, N) - y|, y < N
          maxerr = fmax_abs_raw(maxerr, fmin_raw(fmax(0.0, floor(f)), nf) - y);
          if constexpr (!PLAIN) yprev = y;
        }
      };
      if constexpr (PLAIN) {
#pragma unroll 8
        for (int s = 0; s < FS_ROW; s++) one(s);
      } else {
#pragma unroll 2
        for (int s = 0; s < FS_ROW; s++) one(s);
      }
      if constexpr (PLAIN) yprev = row_if + (double)(FS_ROW - 1);
    };
    __builtin_amdgcn_s_setprio(2);                           // as in pass A: -2.5 % (with the nt loads -5.6 %)
    if (__any(general)) steps(std::false_type{}, std::false_type{});
    else if (plain_ok) steps(std::true_type{}, std::true_type{});
    else steps(std::true_type{}, std::false_type{});
    __builtin_amdgcn_s_setprio(0);
    if (pending >= q_drain) drain();                         // (at most 64 more can arrive per panel in the fast loop)
    row_i += FS_ROW;
    row_if += (double)FS_ROW;
    if (row_i >= chunk_end) lane_done = true;
    P += 1;
  }
  // The key right after a shard starts another leaf (hence another key value): it ends the run of
  // this shard's last key, but it is processed by the next rank, so account for it here.
  if (p0 < sp.it_hi && chunk_end == sp.it_hi && sp.it_hi < sp.n && have_leaf) maxrun = fmax(maxrun, (double)sp.it_hi - yprev);
  flush(true);
  if (pending) drain();
}



// Read-only streaming kernel: the box's achievable HBM read bandwidth in this harness (the
// denominator SURVEY.md section 8d asks to report next to the 8 TB/s spec peak).
static __global__ void __launch_bounds__(256) k_read_bw(const uint4* __restrict__ src, uint64_t n16, unsigned int* __restrict__ sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned int acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) { const uint4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345678u) sink[0] = acc;   // practically never; keeps the loads alive
}
// the same bytes with the access pattern of the one-read kernels: a wave owns contiguous pieces of 8 KB (8 non-temporal 16-byte
// loads per lane back to back), pieces dealt round-robin to the waves -- the best read-only pattern of tools/probe/bw_probe.hip
static __global__ void __launch_bounds__(256) k_read_bw_chunk(const uint4* __restrict__ src, uint64_t n16, unsigned int* __restrict__ sink) {
  typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
  constexpr int U = 8;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x % 64;
  const uint64_t nwaves = (uint64_t)gridDim.x * blockDim.x / 64;
  const raw_t* const s4 = reinterpret_cast<const raw_t*>(src);
  unsigned int acc = 0;
  uint64_t c = wave;
  for (; (c + 1) * U * 64 <= n16; c += nwaves) {
    const raw_t* p = s4 + c * U * 64 + lane;
    raw_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (c * U * 64 < n16 && c * U * 64 + U * 64 > n16)                     // the last, partial piece
    for (uint64_t i = c * U * 64 + lane; i < n16; i += 64) { const raw_t v = s4[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}


// Self-test of div_by_count against IEEE division: every count 1..FS_TMAX-1 with pseudo-random
// numerators and with numerators constructed next to rounding midpoints of the quotient.
static __global__ void __launch_bounds__(256) k_selftest_div(unsigned long long trials_per_thread, unsigned long long seed,
                                                      unsigned long long* __restrict__ mismatches) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad = 0;
  unsigned long long state = seed + tid * 0x9E3779B97F4A7C15ull;
  for (unsigned long long it = 0; it < trials_per_thread; it++) {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long z = state ^ (state >> 29);
    z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    // counts of the table (reciprocal as the table holds it), or any count below 2^40 (computed)
    const bool big = (z >> 6) & 1ull;
    const unsigned long long n = big ? 1ull + ((state >> 11) % ((1ull << 40) - 1ull))
                                     : 1ull + (z % (unsigned long long)(FS_TMAX - 1));
    const double nf = (double)n;
    const double r = big ? recip_exact(nf) : 1.0 / nf;
    double a;
    const unsigned int mode = (unsigned int)(z >> 60) & 3u;
    // exponent range of the recurrence's operands (and some margin): 2^-100 .. 2^+100
    const int ex = (int)((z >> 40) % 201ull) - 100;
    const double frac = 1.0 + (double)((z >> 8) & 0xFFFFFFFFFFFFFull) * 0x1p-52;
    if (mode == 0) {
      a = ldexp(frac, ex);
    } else {
      // q a random double; a = RN(nf * (q +- half an ulp +- tiny)): quotients next to a midpoint
      const double q = ldexp(frac, ex);
      const double half = ldexp(1.0, ex - 53);
      const double qm = (mode & 1u) ? q + half : q - half;          // exact: q has 53 bits, half is 2^-53 below
      a = qm * nf;                                                  // rounded: lands within an ulp of nf * midpoint
      if (mode == 3) a = nextafter(a, (z & 1ull) ? 1e300 : -1e300);
    }
    if ((z >> 7) & 1ull) a = -a;
    const double want = a / nf;
    const double got = div_by_count(a, nf, r);
    const double got2 = div_by_count2(a, r, recip_tail(nf, r));
    if (__builtin_bit_cast(unsigned long long, want) != __builtin_bit_cast(unsigned long long, got)) bad++;
    if (__builtin_bit_cast(unsigned long long, want) != __builtin_bit_cast(unsigned long long, got2)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}


// =============================================================================================
// k_fit_long: the leaves pass A hands over (more than `long_min` points).  The recurrence of one
// leaf is a sequential chain of its length, so what matters is the latency of a step: ONE WAVE per
// leaf.  64 keys at a time the lanes prepare, in parallel, everything that does not depend on the
// chain -- x = f64(key), y = FixDups first-occurrence index, count and RN(1/count) -- into LDS;
// then the wave walks the 64 steps with nothing but the recurrence itself in the loop.  An f64
// instruction costs a wave the same 4 cycles whatever it does in its lanes, so the two mean
// updates of a step share their instructions: even lanes carry (x, mean_x, m2), odd lanes
// (y, mean_y, c); the one cross term, c += dx * (y - mean_y'), gets dx from the even neighbour by
// DPP.  9 vector instructions per step, next to a dependent chain of 4 (sub, mul, fma, add: the
// quotient by the count comes from a two-term reciprocal, div_by_count2).  Same operations in
// the same order as linear.rs:12-59 on the container of the leaf: bit-identical coefficients.
// (f64 keys divide with `/`: every lane runs the whole recurrence.)
// =============================================================================================
constexpr int FL_TILE = 64;

__device__ __forceinline__ double dpp_from_even_lane(double v) {           // quad_perm [0,0,2,2]
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned int)b, 0xA0, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned int)(b >> 32), 0xA0, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}

struct FitLongLds {
  double s_v[2][2][FL_TILE];                               // [buffer][x | y][step]
  double s_rn[2][FL_TILE][2];                              // [buffer][step][1/count: head, tail]
};

// One wave fits the leaf whose container is [lo, hi] (see above); lane 0 writes (alpha, beta) to out.
template <typename K>
__device__ __forceinline__ void fit_long_leaf(const K* __restrict__ keys, const Span& sp, uint64_t lo, uint64_t hi,
                                              DevState* __restrict__ st, double* __restrict__ out, FitLongLds& L) {
  constexpr bool SPLIT = UseRecipTable<K>::value;          // two-lane form with reciprocal division
  auto& s_v = L.s_v;
  auto& s_rn = L.s_rn;
  const int lane = threadIdx.x;
  const int half = lane & 1;
  {
    double carry_y = (double)first_occurrence(keys, lo, sp.rd_lo);
    SlrState sl = {0.0, 0.0, 0.0, 0.0, 0.0};                 // !SPLIT: the whole state in every lane
    double m = 0.0, acc = 0.0;                               // SPLIT: mean_x | mean_y,  m2 | c
    auto prepare = [&](uint64_t base, int b, K k, K kp) {
      const uint64_t i = base + lane;
      const bool newrun = i <= hi && i > lo && !(k == kp);
      const unsigned long long mk = __ballot(newrun);
      const unsigned long long below = mk & ((2ull << lane) - 1ull);      // lane 63: 2<<63 wraps to 0, minus 1 = all ones
      const double y = below ? (double)(base + (uint64_t)(63 - __builtin_clzll(below))) : carry_y;
      s_v[b][0][lane] = KeyTraits<K>::as_float(k);
      s_v[b][1][lane] = y;
      if constexpr (SPLIT) {
        const double nn = (double)(i - lo + 1);
        const double rr = recip_exact(nn);
        s_rn[b][lane][0] = rr;
        s_rn[b][lane][1] = recip_tail(nn, rr);
      }
      // the first-occurrence index carried into the next tile: that of this tile's last key
      if (mk) carry_y = (double)(base + (uint64_t)(63 - __builtin_clzll(mk)));
    };
    auto load = [&](uint64_t base, K& k, K& kp) {
      uint64_t i = base + lane;
      i = i <= hi ? i : hi;
      k = keys[i];
      kp = keys[i > lo ? i - 1 : i];
    };
    auto step = [&](double v, double rr, double rl) {      // v: x | y of this step
      const double d = v - m;
      m += div_by_count2(d, rr, rl);
      const double d2 = v - m;                              // x - mean_x' | y - mean_y'
      acc += dpp_from_even_lane(d) * d2;                    // m2 += dx dx2 | c += dx dy2
    };
    K k, kp;
    load(lo, k, kp);
    int b = 0;
    double last_v = 0.0, last_x = 0.0, last_y = 0.0;
    for (uint64_t base = lo; base <= hi; base += FL_TILE, b ^= 1) {
      prepare(base, b, k, kp);
      if (base + FL_TILE <= hi) load(base + FL_TILE, k, kp);             // next tile's keys: in flight during the chain
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int steps = (hi - base + 1 < (uint64_t)FL_TILE) ? (int)(hi - base + 1) : FL_TILE;
      if constexpr (SPLIT) {
        const double* __restrict__ pv = s_v[b][half];
        if (steps == FL_TILE) {
#pragma unroll 8
          for (int u = 0; u < FL_TILE; u++) step(pv[u], s_rn[b][u][0], s_rn[b][u][1]);
        } else {
          for (int u = 0; u < steps; u++) step(pv[u], s_rn[b][u][0], s_rn[b][u][1]);
        }
        last_v = pv[steps - 1];
      } else {
        for (int u = 0; u < steps; u++) slr_push(sl, s_v[b][0][u], s_v[b][1][u]);
        last_x = s_v[b][0][steps - 1];
        last_y = s_v[b][1][steps - 1];
      }
    }
    // Q1: the tail duplicate of the FixDups iterator (models/mod.rs:180)
    if constexpr (SPLIT) {
      const double nn = (double)(hi - lo + 2);
      const double rr = recip_exact(nn);
      step(last_v, rr, recip_tail(nn, rr));
      sl.mx = __shfl(m, 0); sl.my = __shfl(m, 1);
      sl.m2 = __shfl(acc, 0); sl.c = __shfl(acc, 1);
      sl.nf = nn;
    } else {
      slr_push(sl, last_x, last_y);
    }
    if (lane == 0) {
      const double cov = sl.c / (sl.nf - 1.0);
      const double var = sl.m2 / (sl.nf - 1.0);
      if (!(var >= 0.0)) atomicOr(&st->err_flags, EF_NEG_VARIANCE);       // linear.rs:48
      if (var == 0.0) { out[0] = sl.my; out[1] = 0.0; }                     // linear.rs:50-53
      else {
        const double beta = cov / var;
        out[0] = sl.my - beta * sl.mx;                                      // no fma: linear.rs:56
        out[1] = beta;
      }
    }
  }
}

template <int ROOT, typename K>
__global__ void __launch_bounds__(64) k_fit_long(const K* __restrict__ keys, Span sp, RootP r,
                                                 const unsigned long long* __restrict__ leaf_start,
                                                 DevState* __restrict__ st, double* __restrict__ params,
                                                 const unsigned long long* __restrict__ long_idx) {
  __shared__ FitLongLds lds;
  const uint64_t cnt = st->long_count < st->long_cap ? st->long_count : st->long_cap;
  for (uint64_t t = blockIdx.x; t < cnt; t += gridDim.x) {
    bool oob;
    const uint64_t j = (uint64_t)root_target_f<ROOT, K>(r, (double)(r.L - 1), keys[long_idx[t]], oob);
    uint64_t lo, hi;
    const int ck = leaf_container(j, leaf_start[j], leaf_start[j + 1], sp.n, st->split_idx, st->split_target, lo, hi);
    if (ck != 2) continue;                                   // cannot happen for a handed-over leaf
    fit_long_leaf<K>(keys, sp, lo, hi, st, params + j * 2, lds);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the buffers are reused by the next leaf)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// recip_exact(n) == 1.0 / n for every integer n in [n_lo, n_hi)
static __global__ void __launch_bounds__(256) k_selftest_recip(unsigned long long n_lo, unsigned long long n_hi,
                                                        unsigned long long* __restrict__ mismatches) {
  unsigned long long bad = 0;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long v = n_lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < n_hi; v += stride) {
    const double nf = (double)v;
    if (__builtin_bit_cast(unsigned long long, 1.0 / nf) != __builtin_bit_cast(unsigned long long, recip_exact(nf))) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace rmi
