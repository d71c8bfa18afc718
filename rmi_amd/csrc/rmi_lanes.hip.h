// rmi_lanes.hip.h -- the LEAF-LANE kernels of the exact leaf path (gfx950, wave64): pipeline 3.
//
//   k_leaf_search  leaf boundaries WITHOUT a pass over the keys: for a root whose targets are monotone by
//                  arithmetic (linear.rs:87-90 with a slope >= 0), leaf_start[j] = first index with
//                  target >= j is a lower bound in the sorted key array (exactly lower_bound_by of
//                  two_layer.rs:132-136, once per leaf): a block brackets its 256 leaves with a cooperative
//                  256-ary search, then every thread interpolates and gallops to its own boundary.  ~3 probes
//                  per leaf touch lines the fit reads anyway.  Replaces the bucketing scan AND the suffix-min
//                  fill (an empty leaf's lower bound is the next leaf's start by definition).
//   k_leaf_lanes   exact per-leaf SLR (linear.rs:12-59 on the container of two_layer.rs:52-90) and, fused behind
//                  it, the last-level error pass (two_layer.rs:207-217, lower_bound_correction.rs:104-119):
//                  a wave owns 64 CONSECUTIVE LEAVES, one per lane, and all lanes walk their containers in
//                  lockstep from their first point.  Because every lane is at the same step k, everything
//                  that depends on the running count only -- RN(1/k), k, (k-1)/2 -- is wave-uniform and comes
//                  from a table through the scalar cache (SGPR operands, no vector instruction), and because a
//                  lane never changes leaf mid-walk there is no boundary machinery in the step at all.
//
// The step.  The reference's recurrence per point (x, y) is
//     n += 1; dx = x - mx; mx += dx / n; my += (y - my) / n; c += dx * (y - my); m2 += dx * (x - mx).
// With y_k = y_1 + (k - 1) -- consecutive positions, i.e. no duplicate key in the container so far -- the
// y half is EXACT in closed form: y_k - my_{k-1} = k/2, the quotient is 0.5, my_k = y_1 + (k-1)/2 and
// y_k - my_k = (k-1)/2, all representable, so every rounding of the reference's my-chain is the identity and
//     c += dx * ((k-1)/2)
// is bit for bit the reference's update (checked against the oracle on every generator).
// The step shrinks to: convert (3), dx (1), quotient by the count (3: div_by_count, provably the IEEE
// quotient), mx (1), c (2), m2 (3) = 13 vector instructions per key, no LDS read besides the key itself.
// A lane that meets a duplicate key (y = first-occurrence offset, models/mod.rs:154-185) switches to the
// explicit my-chain for the rest of its leaf (the wave takes the general variant of the panel).
//
// Data movement: the containers of a wave's 64 leaves are one contiguous stretch of the key array, but lane-per-leaf
// reads are strided by a leaf; so per panel of 16 steps the wave fetches 64 rows x 16 keys with 8 instructions of
// 8 rows x 128 B (one row = 8 lanes x 16 B: full lines), stages them through a padded LDS image (row stride 17:
// conflict-free lane-per-row reads) and prefetches the next panel into registers meanwhile.  The error pass then
// re-reads the same stretch (98 KB per wave at 191 keys per leaf) -- the second read is what the 256 MiB
// Infinity Cache is for: the footprint in flight (waves x 98 KB) is kept below it by the launch geometry.
#pragma once
#include <type_traits>

#include "rmi_device.hip.h"
#include "rmi_kernels.hip.h"
#include "rmi_stream.hip.h"
#include "rmi_sigma.hip.h"

#ifndef RMI_LN_NT_FIT
#define RMI_LN_NT_FIT 0          // fit-phase loads: default policy (they are read again by the error phase)
#endif
#ifndef RMI_LN_NT_ERR
#define RMI_LN_NT_ERR 1          // error-phase loads: non-temporal (last use)
#endif
#ifndef RMI_LN_WPE
#define RMI_LN_WPE 2             // waves per SIMD the register allocation of k_leaf_lanes aims at
#endif
#ifndef RMI_LN_NBUF
#define RMI_LN_NBUF 2            // panels in flight per wave (8 KB each, landing in registers) beside the one being staged;
#endif                           // measured 2 / 3 / 4: 515 / 539 / 580 us for the fused kernel (the trips are NBUF panels long)

namespace rmi {

// Panel geometry: a load instruction is 64 lanes x 16 bytes = 8 rows x one aligned 128-byte line; a panel is 8 such loads
// (64 rows), i.e. LnGeom<K>::KPL keys per lane and load and ROW = 8 KPL keys per row: 16 keys of 8 bytes, 32 of 4.
// (Tried for 8-byte keys: half lines per row -- 8 keys, a ring of 11.8 KB, 3 waves per SIMD: 0.71-0.74 ms against 0.60.
//  4-byte keys in 16-key panels, i.e. half lines and 8-byte loads: C5 0.82 ms against 0.68 with whole lines.)
template <typename K> struct LnGeom {
  static constexpr int KPL = 16 / (int)sizeof(K);   // keys per lane and load
  static constexpr int ROW = 8 * KPL;               // keys per panel row = lockstep steps per panel
  static constexpr int RING = 2 * ROW;              // LDS slots per row: two aligned panels (a lane's steps straddle two of them)
  static constexpr int MIRROR = 7;                  // the first slots once more behind the ring: 8 consecutive reads never wrap
  static constexpr int STRIDE = RING + MIRROR;      // 39 / 71, odd: lane-per-row reads spread over the banks
};
constexpr int LN_LPR = 8;                           // lanes per row and load
constexpr int LN_RPI = 8;                           // rows per load instruction
constexpr int LN_NLD = 8;                           // loads per panel
constexpr int LN_LONG_MAX = 8192; // longest container the lockstep walk takes (longer: the list kernels, one wave per leaf)
constexpr int LN_TMAX = LN_LONG_MAX + 64;   // entries of the reciprocal table
#ifndef RMI_LS_BLOCK
#define RMI_LS_BLOCK 512          // (128: 59.8 us, 256: 50.3, 512: 48.0, 1024: 48.1 for 2^20 leaves: the table copy per block)
#endif
constexpr int LS_BLOCK = RMI_LS_BLOCK;   // threads per block of k_leaf_search
#ifndef RMI_LS_ILP
#define RMI_LS_ILP 1           // (2^20 leaves, block 512: 39.7 / 44.9 / 51.4 us with 1 / 2 / 4 -- the probes are bound by scattered line requests, not by their latency)
#endif
constexpr int LS_ILP = RMI_LS_ILP;       // leaves per thread of k_leaf_search: probe chains in flight per lane

// RN(1 / k) for the running count k of the lockstep walk: rtab[i] = 1 / (i + 1).  Wave-uniform, read through the
// scalar cache in two 64-byte loads per panel (SGPR operands of the quotient: no vector instruction, no VGPR).
// ... and next to it the count itself and (k - 1) / 2 as doubles: tab[i], tab[LN_TMAX + i], tab[2 LN_TMAX + i] for k = i + 1.
static __global__ void __launch_bounds__(256) k_lane_table(double* __restrict__ tab, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) { tab[i] = 1.0 / (double)(i + 1); tab[count + i] = (double)(i + 1); tab[2 * count + i] = (double)i * 0.5; }
}

template <typename K> struct LnBits { using type = unsigned long long; };   // the raw bits of a key
template <> struct LnBits<uint32_t> { using type = unsigned int; };

// ---------------------------------------------------------------------------------------------
// k_leaf_search (roots whose targets are monotone BY ARITHMETIC: linear-like with a slope >= 0, radix when every key
// shares the prefix -- the host checks both).  leaf_start[j] = lower bound of "target >= j" in the sorted keys, one
// thread per leaf: bracket from the sample table (k_leaf_samples; a binary search in a cache-resident table), interpolate
// on the unfloored root values (keys are locally uniform at that scale: within a few keys), then PAIR probes (one 16-byte
// load = two neighbouring keys) gallop and bisect: a chain of ~3 scattered loads.  What limits this kernel is that chain
// (~2-3 us per scattered load under load) and the number of scattered line requests.  History for 2^20 leaves: sector
// probes (four loads per lane) 171 us; bracket of +-0.5 leaf around j n / L (fails often: chains of 12) 212 us; anchors
// every 16th leaf 61-78 us; bracket of +-1.5 leaves 58 us -- and 3.4 ms on a radix root whose leaves are not evenly filled.
// ---------------------------------------------------------------------------------------------
// the root's value before flooring and clamping, as a double: only places the first probe of a search
template <int ROOT, typename K>
__device__ __forceinline__ double ls_root_value(const RootP& r, K k) {
  if constexpr (ROOT == K_RADIX) {
    // radix.rs:43-50 keeps the `bits` bits behind the common prefix; the bits behind those give the position inside the bin
    const uint64_t v = KeyTraits<K>::as_uint(k) << (r.prefix & 63u);
    return ldexp(KeyTraits<uint64_t>::as_float(v), -(int)((64u - r.bits) & 63u));
  } else return root_eval_f<ROOT>(r, KeyTraits<K>::as_float(k));
}

// Level 0 of the search: the root's target (exact: floored, clamped) and value (unfloored) at LS_SAMPLES + 1 evenly spaced
// keys of the launch.  A root need not spread the keys evenly over its leaves (a radix root over keys that do not fill
// their 2^k range, a linear root on skewed data), so "leaf j starts near j n / L" can be off by millions of keys; the
// sample table brackets every leaf to n / 1024 keys whatever the root does, and the values interpolate inside the bracket.
constexpr int LS_SAMPLES = 1024;
__device__ __forceinline__ uint64_t ls_sample_index(uint64_t A, uint64_t Bn, int t) {
  const uint64_t stride = (Bn - A - 1) / (uint64_t)LS_SAMPLES > 0 ? (Bn - A - 1) / (uint64_t)LS_SAMPLES : 1;
  const uint64_t i = A + (uint64_t)t * stride;
  return (t >= LS_SAMPLES || i >= Bn) ? Bn - 1 : i;
}
// What k_init does for this pipeline (state, sentinel, list counters), and the arrival counters of k_leaf_lanes' waves:
// carried by the launch of k_leaf_samples (one launch and one gap fewer in front of the search).
struct LaneInit {
  DevState* st; DevState init;
  unsigned long long* leaf_start; uint64_t L_own; unsigned long long sentinel;
  unsigned long long* list_cnt; int n_list_cnt;
  unsigned int* tickets; unsigned int n_tickets;
};
template <int ROOT, typename K>
__global__ void __launch_bounds__(256) k_leaf_samples(const K* __restrict__ keys, Span sp, RootP r, double* __restrict__ smp, LaneInit li) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (li.st != nullptr && blockIdx.x == gridDim.x - 1) {
    for (int q = threadIdx.x; q < li.n_list_cnt; q += 256) li.list_cnt[q] = 0ull;
    for (unsigned int q = threadIdx.x; q < li.n_tickets; q += 256u) li.tickets[q] = 0u;
    if (threadIdx.x == 0) { *li.st = li.init; li.leaf_start[li.L_own] = li.sentinel; }
  }
  if (t > LS_SAMPLES || !(sp.it_hi > sp.it_lo)) return;
  const K k = keys[ls_sample_index(sp.it_lo, sp.it_hi, t)];
  bool oob;
  smp[t] = root_target_f<ROOT, K>(r, (double)(r.L - 1), k, oob);
  smp[LS_SAMPLES + 1 + t] = ls_root_value<ROOT, K>(r, k);
}

// ILP leaves per thread (leaves j, j + LS_BLOCK, ... of the block's stretch), their probe chains interleaved: the kernel is bound by the
// latency of a chain of ~5 dependent scattered loads, so a thread keeps ILP of them in flight -- a search is a little state machine
// (`ph`: what the probe in flight is, `q`: where), all leaves of a thread issue their next probe back to back and then consume them.
template <int ROOT, typename K, int ILP>
__global__ void __launch_bounds__(LS_BLOCK) k_leaf_search(const K* __restrict__ keys, Span sp, RootP r,
                                                          unsigned long long* __restrict__ leaf_start,
                                                          DevState* __restrict__ st, const double* __restrict__ smp) {
  typedef typename LnBits<K>::type BT;
  typedef BT vec_t __attribute__((ext_vector_type(2), aligned(sizeof(K))));
  const double Lm1f = (double)(r.L - 1);
  const uint64_t A = sp.it_lo, Bn = sp.it_hi;
  auto tgt = [&](uint64_t i) -> double { bool oob; return root_target_f<ROOT, K>(r, Lm1f, keys[i], oob); };
  // the sample table into LDS (16 KB: the occupancy stays at 8 waves per SIMD): the bracket search is 11 dependent reads,
  // and the scattered probes of 8 192 resident waves keep evicting the table from the vector L1
  __shared__ double t_tgt[LS_SAMPLES + 1], t_val[LS_SAMPLES + 1];
  for (int t = threadIdx.x; t <= LS_SAMPLES; t += LS_BLOCK) { t_tgt[t] = smp[t]; t_val[t] = smp[LS_SAMPLES + 1 + t]; }
  __syncthreads();
  enum { DONE = 0, P1, P2, FIRST, UP, DOWN, BISECT };
  // one search: the answer lies in [lo, hi]; `q` = index of the pair probe in flight; `slope` = keys per unit of root value;
  // (v1, q1) = unfloored root value and place of the first probe; `d` = gallop step; `mark` = the bound the gallop moves, before the probe
  struct Search { uint64_t lo, hi, q, q1, mark; double jf, slope, v1; unsigned int d; int ph; };
  Search s[ILP];
  bool dup_seen[ILP];                                                // a probe of this leaf held two equal keys
  // the pair (g - 1, g) around a guess g of the answer, kept inside the bracket
  auto pair_at = [&](const Search& z, double gf) -> uint64_t {
    uint64_t g = gf >= 1.0 ? (gf < 1.8e19 ? (uint64_t)gf - 1 : ~0ull - 1) : 0;
    if (g < z.lo) g = z.lo;
    if (g >= z.hi) g = z.hi - 1;
    return g;
  };
  // the next probe of a gallop or of the bisection behind it (lo < hi)
  auto plan = [&](Search& z) {
    if (z.ph == UP) {
      const uint64_t q = z.lo + z.d - 1;
      if (q >= z.hi) z.ph = BISECT; else { z.q = q; z.mark = z.lo; }
    } else if (z.ph == DOWN) {
      if (z.hi - z.lo <= z.d) z.ph = BISECT; else { z.q = z.hi - z.d; z.mark = z.hi; }
    }
    if (z.ph == BISECT) z.q = z.lo + ((z.hi - z.lo) >> 1);
  };
#pragma unroll
  for (int u = 0; u < ILP; u++) {
    Search& z = s[u];
    const uint64_t j = sp.leaf_lo + ((uint64_t)blockIdx.x * ILP + (uint64_t)u) * LS_BLOCK + threadIdx.x;
    dup_seen[u] = false;
    z.jf = (double)j; z.lo = A; z.hi = Bn; z.ph = DONE; z.q = A; z.q1 = 0; z.mark = 0; z.slope = 0.0; z.v1 = 0.0; z.d = 2u;
    if (!(j < sp.leaf_hi)) { z.hi = z.lo; continue; }
    if (!(Bn > A)) { z.hi = z.lo; continue; }
    // first sample whose target is >= j
    int a = 0, b = LS_SAMPLES + 1;
    while (a < b) { const int m = (a + b) >> 1; if (t_tgt[m] < z.jf) a = m + 1; else b = m; }
    if (a == 0) z.hi = z.lo;                                         // the launch's first key is already in leaf j or behind it
    else if (a > LS_SAMPLES) z.lo = z.hi;                            // even the last key is below: no key reaches leaf j
    else {
      const uint64_t i1 = ls_sample_index(A, Bn, a - 1), i2 = ls_sample_index(A, Bn, a);
      z.lo = i1 + 1; z.hi = i2;                                      // key i1 is below, key i2 is not
      // Interpolation on the unfloored values: from the table (off by ~sqrt(bracket) / 2 keys where the keys are locally
      // uniform), then two secant steps from the values the probes see themselves (a few keys, then ~1), then gallop and
      // bisect from there.  Only the comparisons of exact targets narrow [lo, hi]: the guesses place the probes.
      const double f1 = t_val[a - 1], f2 = t_val[a];
      z.slope = (f2 > f1) ? (double)(i2 - i1) / (f2 - f1) : 0.0;
      const double g1 = (f2 > f1) ? (double)i1 + (z.jf - f1) * z.slope : (double)i1 + 0.5 * (double)(i2 - i1);
      if (z.lo < z.hi) { z.q = pair_at(z, g1); z.ph = P1; }
    }
  }
  for (;;) {
    bool any = false;
    K k0[ILP], k1[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      k0[u] = KeyTraits<K>::zero_value(); k1[u] = k0[u];
      if (s[u].ph != DONE) {
        const uint64_t i = s[u].q;
        if (i + 1 < sp.rd_hi) {
          const vec_t v = *reinterpret_cast<const vec_t*>(keys + i);
          k0[u] = bits_to_key<K>(v.x); k1[u] = bits_to_key<K>(v.y);
        } else { k0[u] = keys[i]; k1[u] = k0[u]; }
        any = true;
      }
    }
    if (!any) break;
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      Search& z = s[u];
      if (z.ph == DONE) continue;
      // the pair probe at i (lo <= i < hi) narrows [lo, hi] by the keys i and i + 1
      const uint64_t i = z.q;
      if (i + 1 < sp.rd_hi) dup_seen[u] = dup_seen[u] || (k0[u] == k1[u]);
      bool oob;
      const bool b0 = root_target_f<ROOT, K>(r, Lm1f, k0[u], oob) < z.jf;
      const bool b1 = root_target_f<ROOT, K>(r, Lm1f, k1[u], oob) < z.jf;
      if (!b0) z.hi = i;
      else if (!b1 || i + 1 >= z.hi) { z.lo = i + 1; if (!b1) z.hi = i + 1; }
      else z.lo = i + 2 < z.hi ? i + 2 : z.hi;
      if (!(z.lo < z.hi)) { z.ph = DONE; continue; }
      if (z.ph == P1) {
        if (z.slope > 0.0) {
          z.v1 = ls_root_value<ROOT, K>(r, k0[u]); z.q1 = i;
          z.q = pair_at(z, (double)i + (z.jf - z.v1) * z.slope + 1.0);
          z.ph = P2;
        } else { z.q = z.lo + ((z.hi - z.lo) >> 1); z.ph = FIRST; }
      } else if (z.ph == P2) {
        const double v2 = ls_root_value<ROOT, K>(r, k0[u]);
        const double s2 = (i != z.q1 && v2 != z.v1) ? ((double)i - (double)z.q1) / (v2 - z.v1) : z.slope;
        z.q = pair_at(z, (double)i + (z.jf - v2) * ((s2 > 0.0 && s2 < 64.0 * z.slope) ? s2 : z.slope) + 1.0);
        z.ph = FIRST;
      } else {
        if (z.ph == FIRST) z.ph = z.lo > i ? UP : DOWN;              // the keys at the guess are below: gallop upwards, else downwards
        else if (z.ph == UP) { if (z.lo > i && z.lo > z.mark) z.d <<= 1; else z.ph = BISECT; }       // "both below": on; else bracketed
        else if (z.ph == DOWN) { if (z.hi == i && z.hi < z.mark) z.d <<= 1; else z.ph = BISECT; }    // "none below": on
        plan(z);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < ILP; u++) {
    const uint64_t j = sp.leaf_lo + ((uint64_t)blockIdx.x * ILP + (uint64_t)u) * LS_BLOCK + threadIdx.x;
    if (!(j < sp.leaf_hi)) continue;
    const uint64_t lo = s[u].lo;
    leaf_start[j] = (unsigned long long)lo;
    if (j == r.L / 2 && lo < sp.it_hi) {                              // two_layer.rs:131-136, 152-156
      if (lo == 0) atomicOr(&st->err_flags, EF_DEGENERATE_SPLIT);     // split_idx == 0 -> :27
      else if (lo > sp.rd_lo) {
        st->split_idx = (unsigned long long)lo;
        st->split_target = (unsigned long long)tgt(lo);
        if (lo + 1 >= sp.n) atomicOr(&st->err_flags, EF_DEGENERATE_SPLIT);   // second half empty -> :27
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && sp.n - 1 >= sp.it_lo && sp.n - 1 < sp.it_hi) st->last_target = (unsigned long long)tgt(sp.n - 1);
  // DevState::regs_dups from every 16th stretch of LS_BLOCK leaves (a sample: thousands of additions to one counter serialise at ~26 ns each)
#pragma unroll
  for (int u = 0; u < ILP; u++) {
    if (((blockIdx.x * (unsigned int)ILP + (unsigned int)u) & 15u) != 0u) continue;
    const int nd = __syncthreads_count(dup_seen[u] ? 1 : 0);
    if (threadIdx.x == 0 && nd > 0) atomicAdd(&st->regs_dups, (unsigned long long)nd);
  }
}

// The direct exchange of a sharded training (rmi_multi.inc.h) inside k_leaf_lanes: a finished leaf's row goes not only to this
// rank's table but to every peer's (IPC-mapped, this epoch's half) -- SURVEY H8's "peer stores from the kernel epilogue":
// the exchange runs under the compute instead of behind it (22 MB per rank at 8 ranks: ~60 us as a step of its own against
// ~100 us of kernels).  tab[p] + 24 j is row j of peer p's table (j global); n == 0: no peers.
struct PeerRows { int n; unsigned char* tab[7]; };

// ---------------------------------------------------------------------------------------------
// k_leaf_lanes
// ---------------------------------------------------------------------------------------------
// VROOT >= 0: a root whose targets are NOT monotone by arithmetic (cubic: three nested fmas round independently).  The
// search then rests on an assumption, and this kernel verifies it key by key during the error pass: every key of the
// partition of leaf j must have target j -- which holds for all leaves exactly when the targets are non-decreasing
// (two_layer.rs:50, the reference's panic) -- and, for roots without a bounds check, stay below L (two_layer.rs:45-48).
// The body of k_leaf_lanes for the 64 leaves of wave `wid`, on `panel` (64 * LnGeom<K>::STRIDE key slots of LDS): a function,
// because k_leaf_regs (rmi_regs.hip.h) runs it for the groups of leaves its register-resident path does not take.
template <typename K, bool ERR, int LEAFK, int VROOT>
__device__ __forceinline__ void leaf_lanes_body(const unsigned int wid, typename LnBits<K>::type* const panel,
                                                const K* __restrict__ keys, const Span& sp,
                                                const unsigned long long* __restrict__ leaf_start,
                                                DevState* __restrict__ st, double* __restrict__ params,
                                                const double* __restrict__ rtab, const SgList& fl, unsigned int long_min,
                                                unsigned long long* __restrict__ leaf_maxerr,
                                                unsigned long long* __restrict__ leaf_run, uint64_t L,
                                                unsigned long long* __restrict__ leaf_err,
                                                unsigned long long* __restrict__ leaf_count,
                                                unsigned char* __restrict__ rows, StatsPartial* __restrict__ partials, const RootP& vr,
                                                const PeerRows& peers) {
  using B = typename LnBits<K>::type;
  constexpr bool DIVK = !UseRecipTable<K>::value;                     // f64 keys: plain IEEE division
  constexpr int LPR = LN_LPR, NLD = LN_NLD, RPI = LN_RPI;            // lanes per row, loads per panel, rows per load
  constexpr int KPL = LnGeom<K>::KPL, LN_STRIDE = LnGeom<K>::STRIDE, LN_RING = LnGeom<K>::RING, LN_ROW = LnGeom<K>::ROW;
  constexpr unsigned int ROWK = (unsigned int)LN_ROW;
  unsigned int* const s_off = reinterpret_cast<unsigned int*>(panel);   // (row descriptors of a phase: exchanged before its first panel is staged)
  unsigned int* const s_end = s_off + 64;

  const int lane = threadIdx.x;
  const uint64_t j0 = sp.leaf_lo + (uint64_t)wid * 64;
  const uint64_t j = j0 + (uint64_t)lane;
  const bool valid = j < sp.leaf_hi;
  uint64_t s = 0, e = 0;
  if (valid) { s = leaf_start[j]; e = leaf_start[j + 1]; }
  uint64_t lo = 0, hi = 0;
  int ck = 0;
  if (valid) ck = leaf_container(j, s, e, sp.n, st->split_idx, st->split_target, lo, hi);
  // wave base index: every container and every leaf of this wave starts at or behind it
  uint64_t wb;
  {
    const unsigned int s0l = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)s);
    const unsigned int s0h = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(s >> 32));
    const uint64_t s0 = ((uint64_t)s0h << 32) | s0l;
    wb = s0 > sp.rd_lo ? s0 - 1 : sp.rd_lo;
  }
  // ... moved down to the start of the 16-key line it lies in (by ADDRESS: whole aligned lines are fetched; the keys
  // of a line in front of the first / behind the last readable key share its page and are never used)
  wb -= (uint64_t)((reinterpret_cast<uintptr_t>(keys + wb) / sizeof(K)) % (uintptr_t)LnGeom<K>::ROW);
  const K* __restrict__ kb = keys + wb;
  const unsigned int npts = ck == 2 ? (unsigned int)((hi - lo + 1 < 0xFFFFFFFFull) ? hi - lo + 1 : 0xFFFFFFFFull) : 0u;
  // leaves for the list kernels (one wave per leaf: exact fit + its error pass): containers longer than the lockstep
  // walk takes, and whatever lies beyond 32-bit offsets from the wave base
  // (the row descriptors are 32-bit BYTE offsets from the wave base: 2^32 / sizeof(K) keys, less a margin of panels)
  constexpr uint64_t FAR = (1ull << 32) / sizeof(K) - (1ull << 16);
  // (linear_spline leaves: no walk for the fit, but a leaf far longer than its neighbours would hold the wave's error pass)
  const bool handed = valid && ((ck == 2 && (npts + 1u > long_min || hi - wb >= FAR)) || (e > s && e - wb >= FAR) ||
                                (LEAFK != K_LINEAR && e - s > (uint64_t)long_min));
  if (handed) fl.push((unsigned int)j);
  const bool act = ck == 2 && !handed;
  // Single keys this wave needs much later -- the container's last key (the Q1 point, models/mod.rs:180), the keys on either
  // side of the leaf (the widening of finalize_one) -- and the pair that decides the FixDups offset of the container's first
  // point: fetched NOW, in front of the first panels, so that none of their round trips stands alone later (a wave lives
  // ~60 us; an exposed round trip under load is 2-4 us, and the LDS ring allows only two waves per SIMD to cover it).
  const bool own = valid && !handed;
  K k_hi = KeyTraits<K>::zero_value(), k_lo = KeyTraits<K>::zero_value(), k_lom1 = KeyTraits<K>::zero_value();
  K k_him1 = KeyTraits<K>::zero_value();
  K k_next = KeyTraits<K>::max_value(), k_prev = KeyTraits<K>::zero_value();
  if (act) {
    k_hi = keys[hi]; k_lo = keys[lo];
    if (lo > sp.rd_lo) k_lom1 = keys[lo - 1];
    if constexpr (LEAFK == K_LINEAR_SPLINE) { if (hi > sp.rd_lo) k_him1 = keys[hi - 1]; }
  }
  if (own) { if (e < sp.n) k_next = keys[e]; if (s > 0) k_prev = keys[s - 1]; }
  // FixDups offset of point i given its key and the key in front of it (the search of first_occurrence only for a duplicate)
  auto first_occ = [&](uint64_t i, K ki, K kim1) -> uint64_t {
    return (i <= sp.rd_lo || kim1 != ki) ? i : first_occurrence(keys, i, sp.rd_lo);
  };

  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // ---- row descriptors.  A row is fetched in ALIGNED chunks of 16 keys (128-byte lines of 8-byte keys): lane l of load
  // instruction i fetches 2 keys of chunk p of row 8 i + l / 8, so an instruction asks for 8 whole lines and every line
  // of the key array is requested once per phase.  (Rows fetched from their own first key, i.e. at arbitrary 8-byte
  // offsets, straddle two lines per chunk: twice the requests, every line fetched twice, and the kernel stalled on the
  // issue of its loads -- 57 % of the wave time -- at 4.4 TB/s.)  A lane's key k then sits at ring slot (a + k) mod 32,
  // a = its row's offset inside its first chunk.  Every load is unconditional (the compiler counts them: NBUF panels
  // are in flight); a finished row keeps re-reading its last chunk.
  unsigned int roff[NLD], rlim[NLD];
  auto make_rows = [&](unsigned int my_off, unsigned int my_len) {   // my_off: relative to wb
    wave_sync();
    s_off[lane] = my_off & ~(ROWK - 1u);
    s_end[lane] = (my_len ? my_off + my_len - 1u : my_off) & ~(ROWK - 1u);
    wave_sync();
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const int row = i * RPI + lane / LPR;
      const unsigned int piece = (unsigned int)KPL * (unsigned int)(lane % LPR);
      roff[i] = (s_off[row] + piece) * (unsigned int)sizeof(K);       // byte offsets from kb: a scalar base + a 32-bit lane offset per load
      rlim[i] = (s_end[row] + piece) * (unsigned int)sizeof(K);
    }
    wave_sync();                                                      // (the descriptors alias the panel)
  };
  constexpr int NBUF = RMI_LN_NBUF;
  B bufs[NBUF][NLD][KPL];
  auto load_panel = [&](B (&buf)[NLD][KPL], unsigned int p16, bool nt) {
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      unsigned int off = roff[i] + p16 * (unsigned int)sizeof(K);
      off = off < rlim[i] ? off : rlim[i];                            // (tried: finished rows all read ONE shared line -- 0.582 against 0.577 ms)
      typedef B vec_t __attribute__((ext_vector_type(KPL)));
      const vec_t* pv = reinterpret_cast<const vec_t*>(reinterpret_cast<const char*>(kb) + off);   // (16-byte aligned: wb is line aligned, the piece a multiple of KPL)
      const vec_t v = nt ? __builtin_nontemporal_load(pv) : *pv;
#pragma unroll
      for (int q = 0; q < KPL; q++) buf[i][q] = v[q];
    }
  };
  // aligned panel p goes to the ring slots [16 (p & 1), +16); an even panel's first slots once more behind the ring
  auto stage = [&](B (&buf)[NLD][KPL], unsigned int p) {
    const unsigned int sb = (p & 1u) * ROWK;
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const unsigned int base = (unsigned int)(i * RPI + lane / LPR) * LN_STRIDE + sb + (unsigned int)KPL * (unsigned int)(lane % LPR);
#pragma unroll
      for (int q = 0; q < KPL; q++) panel[base + q] = buf[i][q];
    }
    if (sb == 0u) {
      const unsigned int slot0 = (unsigned int)KPL * (unsigned int)(lane % LPR);    // (the slots 0 .. 6 once more behind the ring)
      if (slot0 < (unsigned int)LnGeom<K>::MIRROR) {
#pragma unroll
        for (int i = 0; i < NLD; i++) {
          const unsigned int base = (unsigned int)(i * RPI + lane / LPR) * LN_STRIDE + LN_RING + slot0;
#pragma unroll
          for (int q = 0; q < KPL; q++)
            if (slot0 + (unsigned int)q < (unsigned int)LnGeom<K>::MIRROR) panel[base + q] = buf[i][q];
        }
      }
    }
  };

  // the rows of the error pass (the leaves' own keys) and its first panels on their way
  const bool eact = own && e > s;
  const unsigned int elen = eact ? (unsigned int)(e - s) : 0u;
  auto err_prime = [&]() {
    make_rows(eact ? (unsigned int)(s - wb) : 0u, elen);
#pragma unroll
    for (int u = 0; u < RMI_LN_NBUF; u++) load_panel(bufs[u], ROWK * (unsigned int)u, RMI_LN_NT_ERR != 0);
  };

  unsigned int flags = 0;
  double pa = 0.0, pb = 0.0;                                          // this lane's leaf: (alpha, beta)
  bool wave_dups = true;                                              // some container of this wave holds a duplicate key
  // =========================== the fit ===========================
  if constexpr (LEAFK == K_LINEAR_SPLINE) {
    // linear_spline.rs:13-35: the line through the first and the last point of the container -- two keys, their FixDups
    // offsets, the reference's two operations (IEEE division, plain multiply-subtract): the same bits, no walk
    if constexpr (ERR) err_prime();                                    // (the first panels of the error pass: on their way meanwhile)
    if (valid && !handed && ck == 2) {
      const K k0 = k_lo, k1 = k_hi;
      const double y0 = (double)first_occ(lo, k_lo, k_lom1);
      if (lo == hi || k0 == k1) { pa = y0; pb = 0.0; }
      else {
        const double y1 = (double)first_occ(hi, k_hi, k_him1);
        const double x0 = KeyTraits<K>::as_float(k0), x1 = KeyTraits<K>::as_float(k1);
        pb = (y0 - y1) / (x0 - x1);
        pa = y0 - pb * x0;
      }
    } else if (ck == 1) { pa = (double)lo; pb = 0.0; }
    if (valid && !handed) { params[2 * j] = pa; params[2 * j + 1] = pb; }
  } else {
    // ---- linear leaves: lockstep walk of the containers
    make_rows(act ? (unsigned int)(lo - wb) : 0u, act ? npts : 0u);
    const unsigned int a0 = act ? (unsigned int)(lo - wb) & (ROWK - 1u) : 0u;   // ring slot of the container's first point
#pragma unroll
    for (int u = 0; u < NBUF; u++) load_panel(bufs[u], ROWK * (unsigned int)u, RMI_LN_NT_FIT != 0);
    uint64_t y0 = lo;
    if (act) y0 = first_occ(lo, k_lo, k_lom1);                        // FixDups offset of the container's first point
    const double y0f = (double)y0, lof = (double)lo;
    double mx = 0.0, cc = 0.0, m2 = 0.0, my = 0.0, yprev = y0f;
    B kprev = 0;
    bool gen = false;                                                 // explicit my-chain (a duplicate key was met)
    // the 16 steps [p16, p16 + 16) of every lane: they read the aligned panels p16 / 16 (staged one trip ago) and
    // p16 / 16 + 1 (staged now, from `buf`, which is refilled with the panel NBUF further on)
    auto fit_panel = [&](B (&buf)[NLD][KPL], unsigned int p16) {
      wave_sync();
      stage(buf, p16 / ROWK + 1u);
      load_panel(buf, p16 + ROWK * (unsigned int)(NBUF + 1), RMI_LN_NT_FIT != 0);
      wave_sync();
#pragma unroll
      for (int hb = 0; hb < LN_ROW / 8; hb++) {
        const unsigned int b0 = p16 + 8u * (unsigned int)hb;
        // wave-uniform operands of the 8 steps, through the scalar cache: RN(1 / k), k, (k - 1) / 2
        double rr[8], kq[8], hq[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { rr[q] = rtab[b0 + (unsigned int)q]; kq[q] = rtab[LN_TMAX + b0 + (unsigned int)q]; hq[q] = rtab[2 * LN_TMAX + b0 + (unsigned int)q]; }
        B kk[8];
        {
          const unsigned int rb = (unsigned int)lane * LN_STRIDE + ((a0 + b0) & (2u * ROWK - 1u));
#pragma unroll
          for (int q = 0; q < 8; q++) kk[q] = panel[rb + (unsigned int)q];
        }
        const unsigned int rem = (act && npts > b0) ? npts - b0 : 0u;
        const unsigned int vmask = rem >= 8u ? 0xFFu : ((1u << rem) - 1u);
        // duplicates: first only WHETHER some lane may hold one (a compare into a scalar mask per key; positions behind a
        // row's end may raise it needlessly); the per-lane bit masks are formed only on the general path
        bool dq = (b0 == 0u) && (y0 != lo);       // the first point: y is y0, its own index unless the key before the container equals it
        {
          B kp = kprev;
#pragma unroll
          for (int q = 0; q < 8; q++) { if (q > 0 || b0 != 0u) dq = dq || (bits_to_key<K>(kk[q]) == bits_to_key<K>(kp)); kp = kk[q]; }
        }
        const bool full = __all(vmask == 0xFFu);
        const bool anygen = __any((gen || dq) && rem > 0u);
        unsigned int dmask = 0;
        if (anygen) {
          B kp = kprev;
#pragma unroll
          for (int q = 0; q < 8; q++) { dmask |= (bits_to_key<K>(kk[q]) == bits_to_key<K>(kp)) ? (1u << q) : 0u; kp = kk[q]; }
          if (b0 == 0) dmask = (dmask & ~1u) | ((y0 != lo) ? 1u : 0u);
          dmask &= vmask;
        }
        kprev = kk[7];
        auto steps = [&](auto full_tag, auto gen_tag) {
          constexpr bool FULL = decltype(full_tag)::value, GEN = decltype(gen_tag)::value;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const double r = rr[q], kf = kq[q], hh = hq[q];
            if (FULL || ((vmask >> q) & 1u)) {
              const double x = KeyTraits<K>::as_float(bits_to_key<K>(kk[q]));
              const double dx = x - mx;                                   // linear.rs:26
              if constexpr (DIVK) mx += dx / kf; else mx += div_by_count(dx, kf, r);   // :27
              if constexpr (GEN) {
                const double y = ((dmask >> q) & 1u) ? yprev : lof + (kf - 1.0);   // FixDups: a duplicate keeps its first occurrence's offset
                const double dy = y - my;
                if constexpr (DIVK) my += dy / kf; else my += div_by_count(dy, kf, r);    // :28
                cc += dx * (y - my);                                      // :29
                yprev = y;
              } else {
                cc += dx * hh;                                            // :28-29 in closed form (see the head of this file)
              }
              m2 += dx * (x - mx);                                        // :30-31
            }
          }
        };
        if (!anygen) {
          if (full) {
            __builtin_amdgcn_s_setprio(2);                             // (the dependent chain gets the issue slots it asks for)
            steps(std::true_type{}, std::false_type{});
            __builtin_amdgcn_s_setprio(0);
          } else steps(std::false_type{}, std::false_type{});
        } else {
          if (!gen) {                                                  // closed form after b0 points
            my = b0 ? y0f + (double)(b0 - 1u) * 0.5 : 0.0;
            yprev = b0 ? y0f + (double)(b0 - 1u) : y0f;
          }
          steps(std::false_type{}, std::true_type{});
          gen = gen || dmask != 0u;
        }
      }
    };
    // (NBUF panels per trip, no exit in between: with a branch between the panels the compiler loses count of the loads
    //  in flight and waits for ALL of them at every panel -- vmcnt(7..0) instead of vmcnt(23..16) -- i.e. no prefetch)
    wave_sync();
    stage(bufs[0], 0u);
    load_panel(bufs[0], ROWK * (unsigned int)NBUF, RMI_LN_NT_FIT != 0);
    for (unsigned int p16 = 0; __any(act && p16 < npts); p16 += ROWK * (unsigned int)NBUF) {
#pragma unroll
      for (int u = 0; u < NBUF; u++) fit_panel(bufs[(u + 1) % NBUF], p16 + ROWK * (unsigned int)u);
    }
    // (the first panels of the error pass are asked for before the arithmetic that ends the fit)
    if constexpr (ERR) err_prime();
    // ---- the container's last item once more (Q1, models/mod.rs:180), then linear.rs:36-58
    if (act) {
      const double x = KeyTraits<K>::as_float(k_hi);
      const double nn = (double)npts;
      if (!gen) { my = y0f + (nn - 1.0) * 0.5; yprev = y0f + (nn - 1.0); }
      const double nf = nn + 1.0;
      const double dx = x - mx;
      mx += dx / nf;
      my += (yprev - my) / nf;
      cc += dx * (yprev - my);
      m2 += dx * (x - mx);
      const double cov = cc / (nf - 1.0), var = m2 / (nf - 1.0);
      if (!(var >= 0.0)) flags |= EF_NEG_VARIANCE;                     // linear.rs:48
      if (var == 0.0) { pa = my; pb = 0.0; }                           // linear.rs:50-53
      else { pb = cov / var; pa = my - pb * mx; }                      // no fma: linear.rs:56
    } else if (ck == 1) { pa = (double)lo; pb = 0.0; }                 // Q4: one borrowed point (two identical items)
    wave_dups = __any(gen);
    if (valid && !handed) { params[2 * j] = pa; params[2 * j + 1] = pb; }
  }
  // =========================== the error pass over the leaves' own keys ===========================
  if constexpr (ERR) {
    const unsigned int len = elen;                                     // (err_prime() has run: the first panels are on their way)
    const unsigned int a0 = eact ? (unsigned int)(s - wb) & (ROWK - 1u) : 0u;
    const unsigned int n32 = (unsigned int)sp.n, s32 = (unsigned int)s;
    unsigned int emax = 0u, run = 0u, yprev = s32;
    bool tr = false;                                                   // yprev is being tracked (a run of equal keys is open)
    B kprev = 0;
    auto err_panel = [&](B (&buf)[NLD][KPL], unsigned int p16) {
      wave_sync();
      stage(buf, p16 / ROWK + 1u);
      load_panel(buf, p16 + ROWK * (unsigned int)(NBUF + 1), RMI_LN_NT_ERR != 0);
      wave_sync();
#pragma unroll
      for (int hb = 0; hb < LN_ROW / 8; hb++) {
        const unsigned int b0 = p16 + 8u * (unsigned int)hb;
        B kk[8];
        {
          const unsigned int rb = (unsigned int)lane * LN_STRIDE + ((a0 + b0) & (2u * ROWK - 1u));
#pragma unroll
          for (int q = 0; q < 8; q++) kk[q] = panel[rb + (unsigned int)q];
        }
        const unsigned int rem = len > b0 ? len - b0 : 0u;
        const unsigned int vmask = rem >= 8u ? 0xFFu : ((1u << rem) - 1u);
        // (a wave whose containers held no duplicate has none in its leaves either: no compares)
        bool dq = false;
        if (wave_dups) {
          B kp = kprev;
#pragma unroll
          for (int q = 0; q < 8; q++) { if (q > 0 || b0 != 0u) dq = dq || (bits_to_key<K>(kk[q]) == bits_to_key<K>(kp)); kp = kk[q]; }
        }
        const bool full = __all(vmask == 0xFFu);
        const bool anygen = wave_dups && __any((tr || dq) && rem > 0u);
        unsigned int dmask = 0;
        if (anygen) {
          B kp = kprev;
#pragma unroll
          for (int q = 0; q < 8; q++) { dmask |= (bits_to_key<K>(kk[q]) == bits_to_key<K>(kp)) ? (1u << q) : 0u; kp = kk[q]; }
          if (b0 == 0) dmask &= ~1u;                                   // a leaf's first key differs from the key before it
          dmask &= vmask;
        }
        kprev = kk[7];
        const unsigned int i0 = s32 + b0;
        auto steps = [&](auto full_tag, auto gen_tag) {
          constexpr bool FULL = decltype(full_tag)::value, GEN = decltype(gen_tag)::value;
#pragma unroll
          for (int q = 0; q < 8; q++) {
            if (FULL || ((vmask >> q) & 1u)) {
              const double x = KeyTraits<K>::as_float(bits_to_key<K>(kk[q]));
              const double f = __builtin_fma(pb, x, pa);                // linear.rs:87-90
              const unsigned int pr = min(sg_cvt_u32(f), n32);         // models/mod.rs:735-737, two_layer.rs:14-18
              if constexpr (VROOT >= 0) {
                const unsigned int rt = sg_cvt_u32(root_eval_f<VROOT>(vr, x));      // max(0, floor(.)), saturating
                const unsigned int Lm1 = (unsigned int)(vr.L - 1);
                if constexpr (!root_needs_bounds_check<VROOT>()) { if (rt > Lm1) flags |= EF_ROOT_OOB; }
                if ((rt < Lm1 ? rt : Lm1) != (unsigned int)j) flags |= EF_NON_MONOTONE;
              }
              const unsigned int idx = i0 + (unsigned int)q;
              if constexpr (GEN) {
                const bool dup = (dmask >> q) & 1u;
                if (!dup && (b0 | (unsigned int)q) != 0u) run = max(run, idx - yprev);   // a new key value ends the run before it (lower_bound_correction.rs:108-119)
                const unsigned int y = dup ? yprev : idx;
                emax = max(emax, sg_absdiff(pr, y));
                yprev = y;
              } else {
                emax = max(emax, sg_absdiff(pr, idx));
              }
            }
          }
        };
        if (!anygen) {
          if (full) steps(std::true_type{}, std::false_type{});
          else steps(std::false_type{}, std::false_type{});
        } else {
          if (rem > 0u && !tr) yprev = b0 ? i0 - 1u : s32;              // the key before this half is its own first occurrence
          steps(std::false_type{}, std::true_type{});
          if (rem > 0u) tr = rem <= 8u || ((dmask >> 7) & 1u) != 0u;    // (a row that ends here keeps the y of its last key)
        }
      }
    };
    wave_sync();
    stage(bufs[0], 0u);
    load_panel(bufs[0], ROWK * (unsigned int)NBUF, RMI_LN_NT_ERR != 0);
    for (unsigned int p16 = 0; __any(p16 < len); p16 += ROWK * (unsigned int)NBUF) {
#pragma unroll
      for (int u = 0; u < NBUF; u++) err_panel(bufs[(u + 1) % NBUF], p16 + ROWK * (unsigned int)u);
    }
    // the key behind the leaf is a different one: it ends the run of the leaf's last key (the globally last run is
    // never recorded, Q5)
    if (eact && e < sp.n) { const unsigned int yl = tr ? yprev : (unsigned int)e - 1u; run = max(run, (unsigned int)e - yl); }
    // ---- finish the leaf here (two_layer.rs:185-197, 226-259, the row of codegen.rs:288-315, the terms of :267-287):
    // the two boundary keys are the ends of the container this wave has just streamed
    unsigned long long st_mx = 0, st_mi = 0, st_sum = 0;
    double st_l2 = 0.0, st_lg = 0.0;
    if (valid && !handed) {
      double pp[2] = {pa, pb};
      uint64_t final_err, cnt_j;
      finalize_one_pre<K_LINEAR, K>(j, s, e, sp, L, keys, pp, (uint64_t)emax, run > 1u ? (uint64_t)run : 0ull, st->last_target, k_next, k_prev, final_err, cnt_j);
      if (!(s < e)) { params[2 * j] = pp[0]; params[2 * j + 1] = pp[1]; }
      leaf_err[j] = final_err;
      leaf_count[j] = cnt_j;
      double* rp = reinterpret_cast<double*>(rows + j * 24);
      rp[0] = pp[0]; rp[1] = pp[1];
      *reinterpret_cast<unsigned long long*>(rows + j * 24 + 16) = final_err;
      for (int p = 0; p < peers.n; p++) {                              // (wave-uniform trip count; 24-byte rows: three 8-byte stores)
        double* pr = reinterpret_cast<double*>(peers.tab[p] + j * 24);
        pr[0] = pp[0]; pr[1] = pp[1];
        *reinterpret_cast<unsigned long long*>(peers.tab[p] + j * 24 + 16) = final_err;
      }
      st_mx = final_err; st_mi = j;
      st_sum = cnt_j * final_err;                                      // wrapping u64, like the reference's sum
      const double v = (double)st_sum;
      st_l2 = (v * v) / (double)sp.n;
      st_lg = (double)cnt_j * log2((double)(2 * final_err + 2));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {                                 // (lexicographic maximum and sums: any order combines)
      const unsigned long long omx = shfl_down_u64(st_mx, d), omi = shfl_down_u64(st_mi, d);
      if (omx > st_mx || (omx == st_mx && omi > st_mi)) { st_mx = omx; st_mi = omi; }
      st_sum += shfl_down_u64(st_sum, d);
      st_l2 += __shfl_down(st_l2, d);
      st_lg += __shfl_down(st_lg, d);
    }
    if (lane == 0) partials[wid] = StatsPartial{st_mx, st_mi, st_sum, st_l2, st_lg};
  }
  if (flags) atomicOr(&st->err_flags, flags);
}

template <typename K, bool ERR, int LEAFK = K_LINEAR, int VROOT = -1>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RMI_LN_WPE, RMI_LN_WPE))) k_leaf_lanes(const K* __restrict__ keys, Span sp,
                                                   const unsigned long long* __restrict__ leaf_start,
                                                   DevState* __restrict__ st, double* __restrict__ params,
                                                   const double* __restrict__ rtab, SgList fl, unsigned int long_min,
                                                   unsigned long long* __restrict__ leaf_maxerr,
                                                   unsigned long long* __restrict__ leaf_run, uint64_t L,
                                                   unsigned long long* __restrict__ leaf_err,
                                                   unsigned long long* __restrict__ leaf_count,
                                                   unsigned char* __restrict__ rows, StatsPartial* __restrict__ partials, RootP vr,
                                                   PeerRows peers) {
  __shared__ typename LnBits<K>::type panel[64 * LnGeom<K>::STRIDE];   // 19 968 B for 8-byte keys, 18 176 for 4-byte keys: 8 waves per CU
  leaf_lanes_body<K, ERR, LEAFK, VROOT>(blockIdx.x, panel, keys, sp, leaf_start, st, params, rtab, fl, long_min, leaf_maxerr, leaf_run, L, leaf_err, leaf_count,
                                        rows, partials, vr, peers);
}

// ---------------------------------------------------------------------------------------------
// k_lane_reduce: the aggregates of a training whose leaves were all finished by k_leaf_lanes, in ONE launch behind it
// (instead of k_list + k_list_tail + k_finalize_listed: ~20 us + two gaps when nothing is listed).  Block b combines
// the records of the waves [LF_SLICE b, LF_SLICE (b + 1)); the last block to arrive combines the slices and publishes
// the result record -- FINAL when no leaf was handed to the list kernels (pending == 0; otherwise the host runs them
// behind its synchronisation: listed_epilogue in rmi_hip.hip).
// (Tried: every wave of k_leaf_lanes counts itself in and the last one reduces -- no launch at all, but a wave must
//  wait for the acknowledgement of its stores before it may count itself in, ~3-4 us with its LDS still allocated:
//  k_leaf_lanes 507 -> 540 us.)
// Coherence across the XCDs' L2s: the slice records are written and read with agent-scope accesses (sc1: write-through
// / miss-always), a block waits for its stores' acknowledgement (vmcnt) before its agent-scope ticket increment, and
// the last block reads only behind its own increment.  The order of the sums is fixed by the indices: same bits every run.
// ---------------------------------------------------------------------------------------------
constexpr unsigned int LF_SLICE = 128;
__device__ __forceinline__ void lf_store(StatsPartial* p, const StatsPartial& v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q + 0, v.mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, v.mi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 2, v.sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 3, (unsigned long long)__double_as_longlong(v.l2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 4, (unsigned long long)__double_as_longlong(v.lg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ StatsPartial lf_load(const StatsPartial* p) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<StatsPartial*>(p));
  StatsPartial v;
  v.mx = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.mi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.sum = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.l2 = __longlong_as_double((long long)__hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  v.lg = __longlong_as_double((long long)__hip_atomic_load(q + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return v;
}
__device__ __forceinline__ void lf_combine(StatsPartial& a, const StatsPartial& p) {
  if (p.mx > a.mx || (p.mx == a.mx && p.mi > a.mi)) { a.mx = p.mx; a.mi = p.mi; }
  a.sum += p.sum; a.l2 += p.l2; a.lg += p.lg;
}
// the LF_SLICE threads' records combined in thread order (two waves: lanes by shuffles, then wave 1 behind wave 0)
__device__ __forceinline__ void lf_block_reduce(StatsPartial& a, StatsPartial* s_w) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    StatsPartial o;
    o.mx = shfl_down_u64(a.mx, d); o.mi = shfl_down_u64(a.mi, d); o.sum = shfl_down_u64(a.sum, d);
    o.l2 = __shfl_down(a.l2, d); o.lg = __shfl_down(a.lg, d);
    lf_combine(a, o);
  }
  __syncthreads();
  if (threadIdx.x == 64) *s_w = a;
  __syncthreads();
  if (threadIdx.x == 0) lf_combine(a, *s_w);
}
static __global__ void __launch_bounds__(LF_SLICE) k_lane_reduce(const StatsPartial* __restrict__ partials, unsigned int nwaves,
                                                          StatsPartial* __restrict__ slices, unsigned int* __restrict__ ticket,
                                                          SgList fl, DevState* __restrict__ st, DevState* __restrict__ host_copy) {
  static_assert(LF_SLICE == 128 && SG_REGIONS == 64, "two waves per block; one hand-over counter per lane");
  __shared__ StatsPartial s_w;
  __shared__ bool s_last;
  const unsigned int q = blockIdx.x * LF_SLICE + threadIdx.x;
  StatsPartial a{0ull, 0ull, 0ull, 0.0, 0.0};
  if (q < nwaves) a = partials[q];
  lf_block_reduce(a, &s_w);
  if (threadIdx.x == 0) {
    lf_store(slices + blockIdx.x, a);
    s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  StatsPartial b{0ull, 0ull, 0ull, 0.0, 0.0};
  for (unsigned int t = threadIdx.x; t < gridDim.x; t += LF_SLICE) lf_combine(b, lf_load(slices + t));
  lf_block_reduce(b, &s_w);
  unsigned long long fc = threadIdx.x < (unsigned int)SG_REGIONS ? fl.cnt[threadIdx.x] : 0ull;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) fc += shfl_down_u64(fc, d);
  if (threadIdx.x == 0) {
    *ticket = 0u;                                                      // (for the next training; the init of k_leaf_samples zeroes it as well)
    st->max_err = b.mx; st->max_err_idx = b.mi; st->sum_n_err = b.sum; st->sum_l2 = b.l2; st->sum_log2 = b.lg;
    st->flag_count = fc; st->merged_count = 0; st->pending = fc;
    *host_copy = *st;                                                  // pinned host memory: visible to the host once the stream is synchronised
  }
}

// ---------------------------------------------------------------------------------------------
// k_finalize_listed: what is left of k_finalize when k_leaf_lanes finishes its own leaves -- the leaves of the list
// kernels (fitted and measured by k_list / k_list_tail) -- and the first level of the aggregates: block b also
// combines its slice of the waves' partial records, so that k_stats_reduce reads FL_BLOCKS records.
// ---------------------------------------------------------------------------------------------
constexpr int FL_BLOCKS = 1;        // one block of 1024 threads: no arrival counter, no device-scope fence (64 blocks + ticket: 15 us)
constexpr int FL_THREADS = 256;
template <typename K>
__global__ void __launch_bounds__(FL_THREADS) k_finalize_listed(const K* __restrict__ keys, Span sp, uint64_t L,
                                                         const unsigned long long* __restrict__ leaf_start,
                                                         const DevState* __restrict__ st, double* __restrict__ params,
                                                         const unsigned long long* __restrict__ leaf_maxerr,
                                                         const unsigned long long* __restrict__ leaf_run,
                                                         unsigned long long* __restrict__ leaf_err,
                                                         unsigned long long* __restrict__ leaf_count,
                                                         unsigned char* __restrict__ rows, SgList fl,
                                                         const StatsPartial* __restrict__ wave_partials, unsigned int nwave,
                                                         StatsPartial* __restrict__ out, unsigned long long* __restrict__ ticket,
                                                         DevState* __restrict__ stw, DevState* __restrict__ host_copy,
                                                         const GiantLeaf* __restrict__ flat, unsigned long long host_min) {
  unsigned long long mx = 0, mi = 0, sm = 0;
  double l2 = 0.0, lg = 0.0;
  const unsigned int gid = blockIdx.x * (unsigned int)FL_THREADS + threadIdx.x, gsz = gridDim.x * (unsigned int)FL_THREADS;
  // `flat` (the giant-leaf epilogue): the leaves of that array, fitted by the host meanwhile; else the regions of the
  // list, without the leaves k_list left to the host (the same test as there)
  const int nreg = flat ? 1 : SG_REGIONS;
  __shared__ unsigned long long s_cnt[SG_REGIONS];                   // (one load per region, not a chain of 64 dependent ones per thread)
  if (threadIdx.x < SG_REGIONS) s_cnt[threadIdx.x] = flat ? 0ull : (fl.cnt[threadIdx.x] < fl.cap ? fl.cnt[threadIdx.x] : fl.cap);
  __syncthreads();
  for (int rg = 0; rg < nreg; rg++) {
    const unsigned long long cnt = flat ? (st->giant_count < st->giant_cap ? st->giant_count : st->giant_cap) : s_cnt[rg];
    if (cnt == 0) continue;
    for (unsigned long long t = gid; t < cnt; t += gsz) {
      const uint64_t j = flat ? flat[t].j : (uint64_t)(fl.ids[(unsigned long long)rg * fl.cap + t] & ~SG_TAG);
      const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
      if (!flat && host_min != ~0ull) {
        uint64_t lo, hi;
        if (leaf_container(j, s, e, sp.n, st->split_idx, st->split_target, lo, hi) == 2 && hi - lo + 1 > host_min) continue;
      }
      double pp[2] = {params[2 * j], params[2 * j + 1]};
      uint64_t final_err, cnt_j;
      finalize_one<K_LINEAR, K>(j, s, e, sp, L, keys, pp, leaf_maxerr[j], leaf_run[j], st->last_target, final_err, cnt_j);
      if (!(s < e)) { params[2 * j] = pp[0]; params[2 * j + 1] = pp[1]; }
      leaf_err[j] = final_err;
      leaf_count[j] = cnt_j;
      double* rp = reinterpret_cast<double*>(rows + j * 24);
      rp[0] = pp[0]; rp[1] = pp[1];
      *reinterpret_cast<unsigned long long*>(rows + j * 24 + 16) = final_err;
      const unsigned long long ts = cnt_j * final_err;
      if (final_err > mx || (final_err == mx && j > mi)) { mx = final_err; mi = j; }
      sm += ts;
      const double v = (double)ts;
      l2 += (v * v) / (double)sp.n;
      lg += (double)cnt_j * log2((double)(2 * final_err + 2));
    }
  }
  for (unsigned int q = gid; q < nwave; q += gsz) {
    const StatsPartial p = wave_partials[q];
    if (p.mx > mx || (p.mx == mx && p.mi > mi)) { mx = p.mx; mi = p.mi; }
    sm += p.sum; l2 += p.l2; lg += p.lg;
  }
  stats_block_reduce<FL_THREADS>(mx, mi, sm, l2, lg);
  if (threadIdx.x == 0) out[blockIdx.x] = StatsPartial{mx, mi, sm, l2, lg};   // (read again by the giant-leaf epilogue)
  if (gridDim.x > 1) {
    // several blocks: the last one to arrive combines their records
    __shared__ bool last;
    if (threadIdx.x == 0) {
      __threadfence();
      last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    mx = 0; mi = 0; sm = 0; l2 = 0.0; lg = 0.0;
    if (threadIdx.x < gridDim.x) {
      const volatile StatsPartial* vo = out;
      mx = vo[threadIdx.x].mx; mi = vo[threadIdx.x].mi; sm = vo[threadIdx.x].sum; l2 = vo[threadIdx.x].l2; lg = vo[threadIdx.x].lg;
    }
    __syncthreads();                                                 // (stats_block_reduce reuses its LDS arrays)
    stats_block_reduce<FL_THREADS>(mx, mi, sm, l2, lg);
  }
  if (threadIdx.x == 0) {
    stw->max_err = mx; stw->max_err_idx = mi; stw->sum_n_err = sm; stw->sum_l2 = l2; stw->sum_log2 = lg; stw->pending = 0;
    if (host_copy) *host_copy = *stw;                                // pinned host memory: visible to the host once the stream is synchronised
  }
}

// The verification of k_leaf_lanes<.., VROOT> for the leaves it handed to the list kernels: every key of the partition
// of a listed leaf must have that leaf as its target (a block per listed leaf, its threads stride over the keys).
template <int ROOT, typename K>
__global__ void __launch_bounds__(256) k_verify_listed(const K* __restrict__ keys, Span sp, RootP r,
                                                       const unsigned long long* __restrict__ leaf_start, DevState* __restrict__ st, SgList fl) {
  unsigned int flags = 0;
  const unsigned int Lm1 = (unsigned int)(r.L - 1);
  for (int rg = 0; rg < SG_REGIONS; rg++) {
    const unsigned long long cnt = fl.cnt[rg] < fl.cap ? fl.cnt[rg] : fl.cap;
    for (unsigned long long t = blockIdx.x; t < cnt; t += gridDim.x) {
      const uint64_t j = fl.ids[(unsigned long long)rg * fl.cap + t] & ~SG_TAG;
      const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
      for (uint64_t i = s + threadIdx.x; i < e; i += 256) {
        const unsigned int rt = sg_cvt_u32(root_eval_f<ROOT>(r, KeyTraits<K>::as_float(keys[i])));
        if constexpr (!root_needs_bounds_check<ROOT>()) { if (rt > Lm1) flags |= EF_ROOT_OOB; }
        if ((rt < Lm1 ? rt : Lm1) != (unsigned int)j) flags |= EF_NON_MONOTONE;
      }
    }
  }
  if (flags) atomicOr(&st->err_flags, flags);
}

// the error pass of the giant leaves, once the host has written their coefficients: stretches for k_list_tail
static __global__ void __launch_bounds__(64) k_giant_segments(const GiantLeaf* __restrict__ giant, const unsigned long long* __restrict__ leaf_start,
                                                       DevState* __restrict__ st, unsigned long long* __restrict__ segs,
                                                       unsigned long long* __restrict__ leaf_maxerr, unsigned long long* __restrict__ leaf_run) {
  const unsigned long long cnt = st->giant_count < st->giant_cap ? st->giant_count : st->giant_cap;
  for (unsigned long long t = blockIdx.x * 64ull + threadIdx.x; t < cnt; t += (unsigned long long)gridDim.x * 64ull) {
    const uint64_t j = giant[t].j;
    const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
    leaf_maxerr[j] = 0ull; leaf_run[j] = 0ull;
    const uint64_t nseg = (e - s + SG_SEG - 1) / SG_SEG;
    const unsigned long long pos = atomicAdd(&st->seg_count, (unsigned long long)nseg);
    for (uint64_t q = 0; q < nseg && pos + q < st->seg_cap; q++) segs[pos + q] = ((unsigned long long)j << 32) | q;
  }
}

}  // namespace rmi
