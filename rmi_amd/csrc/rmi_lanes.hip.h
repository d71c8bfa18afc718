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

namespace rmi {

constexpr int LN_ROW = 16;        // keys per panel row = lockstep steps per panel
constexpr int LN_STRIDE = 17;     // padded row stride (slots)
constexpr int LN_LONG_MAX = 8192; // longest container the lockstep walk takes (longer: the list kernels, one wave per leaf)
constexpr int LN_TMAX = LN_LONG_MAX + 64;
constexpr int LS_BLOCK = 256;     // leaves per block of k_leaf_search

// per step k (= running count of the recurrence): wave-uniform operands
struct LnStep { double r, kf, h, km1; };   // RN(1/k), k, (k-1)/2, k-1

__global__ void __launch_bounds__(256) k_lane_table(LnStep* __restrict__ tab, int count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const double kf = (double)(k > 0 ? k : 1);
  LnStep t; t.r = 1.0 / kf; t.kf = kf; t.h = (kf - 1.0) * 0.5; t.km1 = kf - 1.0;
  tab[k] = t;
}

// ---------------------------------------------------------------------------------------------
// k_leaf_search (linear-like roots with slope >= 0; the host checks)
// ---------------------------------------------------------------------------------------------
template <typename K>
__global__ void __launch_bounds__(LS_BLOCK) k_leaf_search(const K* __restrict__ keys, Span sp, RootP r,
                                                          unsigned long long* __restrict__ leaf_start,
                                                          DevState* __restrict__ st) {
  const int t = threadIdx.x;
  const double Lm1f = (double)(r.L - 1);
  const uint64_t jb = sp.leaf_lo + (uint64_t)blockIdx.x * LS_BLOCK;
  if (jb >= sp.leaf_hi) return;
  const uint64_t je = jb + LS_BLOCK < sp.leaf_hi ? jb + LS_BLOCK : sp.leaf_hi;
  auto tgt = [&](uint64_t i) -> double { bool oob; return root_target_f<K_LINEAR, K>(r, Lm1f, keys[i], oob); };
  // ---- the block's bracket [A, B]: lower bounds of jb and je, both by the same rounds of 256 probes
  uint64_t loA = sp.it_lo, hiA = sp.it_hi, loB = sp.it_lo, hiB = sp.it_hi;
  const double jA = (double)jb, jB = (double)je;
  if (!(je < sp.leaf_hi)) { loB = sp.it_hi; hiB = sp.it_hi; }       // leaf_start[leaf_hi] = it_hi: the sentinel of k_init
  while (hiA > loA || hiB > loB) {                                   // (block-uniform)
    const uint64_t stA = (hiA - loA + LS_BLOCK - 1) / LS_BLOCK, stB = (hiB - loB + LS_BLOCK - 1) / LS_BLOCK;
    const uint64_t qA = loA + (uint64_t)t * stA, qB = loB + (uint64_t)t * stB;
    const int belowA = (hiA > loA && qA < hiA) ? (tgt(qA) < jA ? 1 : 0) : 0;
    const int belowB = (hiB > loB && qB < hiB) ? (tgt(qB) < jB ? 1 : 0) : 0;
    const int cA = __syncthreads_count(belowA);
    const int cB = __syncthreads_count(belowB);
    if (hiA > loA) {
      if (cA == 0) hiA = loA;
      else {
        const uint64_t nl = loA + (uint64_t)(cA - 1) * stA + 1, nh = loA + (uint64_t)cA * stA;
        hiA = (cA < LS_BLOCK && nh < hiA) ? nh : hiA;
        loA = nl;
      }
    }
    if (hiB > loB) {
      if (cB == 0) hiB = loB;
      else {
        const uint64_t nl = loB + (uint64_t)(cB - 1) * stB + 1, nh = loB + (uint64_t)cB * stB;
        hiB = (cB < LS_BLOCK && nh < hiB) ? nh : hiB;
        loB = nl;
      }
    }
  }
  const uint64_t A = loA, B = loB;
  // ---- every thread its own leaf: interpolate inside [A, B], gallop, bisect
  const uint64_t j = jb + (uint64_t)t;
  if (j < je) {
    uint64_t lo = A, hi = B;
    if (t > 0 && B > A) {
      const double jf = (double)j;
      const uint64_t R = B - A;
      uint64_t g = A + (uint64_t)t * R / (je - jb);
      if (g >= B) g = B - 1;
      uint64_t d = R >> 12;
      if (d < 1) d = 1;
      if (tgt(g) < jf) {
        lo = g + 1;
        while (lo < hi) {
          const uint64_t q = lo + d - 1;
          if (q >= hi) break;
          if (tgt(q) < jf) { lo = q + 1; d <<= 1; } else { hi = q; break; }
        }
      } else {
        hi = g;
        while (lo < hi) {
          if (hi - lo < d) break;
          const uint64_t q = hi - d;
          if (tgt(q) < jf) { lo = q + 1; break; } else { hi = q; d <<= 1; }
        }
      }
      while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (tgt(mid) < jf) lo = mid + 1; else hi = mid;
      }
    } else hi = lo;
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
  if (blockIdx.x == 0 && t == 0 && sp.n - 1 >= sp.it_lo && sp.n - 1 < sp.it_hi) st->last_target = (unsigned long long)tgt(sp.n - 1);
}

// ---------------------------------------------------------------------------------------------
// k_leaf_lanes
// ---------------------------------------------------------------------------------------------
template <typename K> struct LnBits { using type = unsigned long long; };
template <> struct LnBits<uint32_t> { using type = unsigned int; };

template <typename K, bool ERR>
__global__ void __launch_bounds__(64) k_leaf_lanes(const K* __restrict__ keys, Span sp,
                                                   const unsigned long long* __restrict__ leaf_start,
                                                   DevState* __restrict__ st, double* __restrict__ params,
                                                   const LnStep* __restrict__ tab, SgList fl, unsigned int long_min,
                                                   unsigned long long* __restrict__ leaf_maxerr,
                                                   unsigned long long* __restrict__ leaf_run) {
  using B = typename LnBits<K>::type;
  constexpr bool DIVK = !UseRecipTable<K>::value;                     // f64 keys: plain IEEE division
  constexpr int LPR = 8;                                              // lanes per row (2 keys each)
  __shared__ B panel[64 * LN_STRIDE];
  __shared__ unsigned int s_off[64], s_end[64];

  const int lane = threadIdx.x;
  const uint64_t j0 = sp.leaf_lo + (uint64_t)blockIdx.x * 64;
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
  const K* __restrict__ kb = keys + wb;
  const uint64_t rd_last = sp.rd_hi - 1 - wb;                         // relative index of the last readable key
  const unsigned int npts = ck == 2 ? (unsigned int)((hi - lo + 1 < 0xFFFFFFFFull) ? hi - lo + 1 : 0xFFFFFFFFull) : 0u;
  // leaves for the list kernels (one wave per leaf: exact fit + its error pass): containers longer than the lockstep
  // walk takes, and whatever lies beyond 32-bit offsets from the wave base
  constexpr uint64_t FAR = 1ull << 31;
  const bool handed = valid && ((ck == 2 && (npts + 1u > long_min || hi - wb >= FAR)) || (e > s && e - wb >= FAR));
  if (handed) fl.push((unsigned int)j);
  const bool act = ck == 2 && !handed;

  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // ---- row descriptors: lane l of load instruction i fetches 2 keys of row 8 i + l / 8
  unsigned int roff[8], rlim[8];
  auto make_rows = [&](unsigned int my_off, unsigned int my_len) {
    wave_sync();
    s_off[lane] = my_off;
    s_end[lane] = my_len ? my_off + my_len - 1u : my_off;
    wave_sync();
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int row = i * 8 + lane / LPR;
      roff[i] = s_off[row] + 2u * (unsigned int)(lane % LPR);
      rlim[i] = s_end[row];
    }
  };
  B nxt[8][2];
  auto load_panel = [&](unsigned int p16, bool edge, bool nt) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      unsigned int idx = roff[i] + p16;
      idx = idx < rlim[i] ? idx : rlim[i];                             // a finished row keeps re-reading its last line (cache hits)
      if (!edge) {
        typedef B vec_t __attribute__((ext_vector_type(2), aligned(sizeof(K))));
        const vec_t* pv = reinterpret_cast<const vec_t*>(kb + idx);
        const vec_t v = nt ? __builtin_nontemporal_load(pv) : *pv;
        nxt[i][0] = v.x; nxt[i][1] = v.y;
      } else {
        const uint64_t i0 = (uint64_t)idx < rd_last ? (uint64_t)idx : rd_last;
        const uint64_t i1 = (uint64_t)idx + 1 < rd_last ? (uint64_t)idx + 1 : rd_last;
        nxt[i][0] = key_to_bits<K>(kb[i0]); nxt[i][1] = key_to_bits<K>(kb[i1]);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int base = (i * 8 + lane / LPR) * LN_STRIDE + 2 * (lane % LPR);
      panel[base] = nxt[i][0]; panel[base + 1] = nxt[i][1];
    }
  };

  unsigned int flags = 0;
  double pa = 0.0, pb = 0.0;                                          // this lane's leaf: (alpha, beta)
  // =========================== the fit: lockstep walk of the containers ===========================
  {
    uint64_t y0 = lo;
    if (act) y0 = first_occurrence(keys, lo, sp.rd_lo);               // FixDups offset of the container's first point
    const double y0f = (double)y0, lof = (double)lo;
    make_rows(act ? (unsigned int)(lo - wb) : 0u, act ? npts : 0u);
    const bool edge = __any((act ? (hi - wb) + 1 : 1ull) >= rd_last + 1);   // a 16-byte load could reach past the readable keys
    double mx = 0.0, cc = 0.0, m2 = 0.0, my = 0.0, yprev = y0f;
    B kprev = 0;
    bool gen = false;                                                 // explicit my-chain (a duplicate key was met)
    load_panel(0u, edge, RMI_LN_NT_FIT != 0);
    for (unsigned int p16 = 0; __any(act && p16 < npts); p16 += 16) {
      wave_sync();
      stage();
      load_panel(p16 + 16u, edge, RMI_LN_NT_FIT != 0);                // prefetch: lands during the steps
      wave_sync();
      B kk[16];
#pragma unroll
      for (int q = 0; q < 16; q++) kk[q] = panel[lane * LN_STRIDE + q];
      const unsigned int rem = (act && npts > p16) ? npts - p16 : 0u;
      const unsigned int vmask = rem >= 16u ? 0xFFFFu : ((1u << rem) - 1u);
      unsigned int dmask = 0;
      {
        B kp = kprev;
#pragma unroll
        for (int q = 0; q < 16; q++) { dmask |= (bits_to_key<K>(kk[q]) == bits_to_key<K>(kp)) ? (1u << q) : 0u; kp = kk[q]; }
      }
      // the first point: y is y0, which is its own index unless the key before the container equals it
      if (p16 == 0) dmask = (dmask & ~1u) | ((y0 != lo) ? 1u : 0u);
      dmask &= vmask;
      kprev = kk[15];
      const bool full = __all(vmask == 0xFFFFu);
      const bool anygen = __any((gen && rem > 0u) || dmask != 0u);
      auto steps = [&](auto full_tag, auto gen_tag) {
        constexpr bool FULL = decltype(full_tag)::value, GEN = decltype(gen_tag)::value;
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const LnStep t = tab[p16 + (unsigned int)q + 1u];             // wave-uniform: scalar loads
          if (FULL || ((vmask >> q) & 1u)) {
            const double x = KeyTraits<K>::as_float(bits_to_key<K>(kk[q]));
            const double dx = x - mx;                                   // linear.rs:26
            if constexpr (DIVK) mx += dx / t.kf; else mx += div_by_count(dx, t.kf, t.r);   // :27
            if constexpr (GEN) {
              const double y = ((dmask >> q) & 1u) ? yprev : lof + t.km1;   // FixDups: a duplicate keeps its first occurrence's offset
              const double dy = y - my;
              if constexpr (DIVK) my += dy / t.kf; else my += div_by_count(dy, t.kf, t.r);  // :28
              cc += dx * (y - my);                                      // :29
              yprev = y;
            } else {
              cc += dx * t.h;                                           // :28-29 in closed form (see the head of this file)
            }
            m2 += dx * (x - mx);                                        // :30-31
          }
        }
      };
      if (!anygen) {
        if (full) steps(std::true_type{}, std::false_type{});
        else steps(std::false_type{}, std::false_type{});
      } else {
        if (!gen) {                                                    // closed form after p16 points
          my = p16 ? y0f + (double)(p16 - 1u) * 0.5 : 0.0;
          yprev = p16 ? y0f + (double)(p16 - 1u) : y0f;
        }
        steps(std::false_type{}, std::true_type{});
        gen = gen || dmask != 0u;
      }
    }
    // ---- the container's last item once more (Q1, models/mod.rs:180), then linear.rs:36-58
    if (act) {
      const double x = KeyTraits<K>::as_float(keys[hi]);
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
    if (valid && !handed) { params[2 * j] = pa; params[2 * j + 1] = pb; }
  }
  // =========================== the error pass over the leaves' own keys ===========================
  if constexpr (ERR) {
    const bool eact = valid && !handed && e > s;
    const unsigned int len = eact ? (unsigned int)(e - s) : 0u;
    make_rows(eact ? (unsigned int)(s - wb) : 0u, len);
    const bool edge = __any((eact ? (e - 1 - wb) + 1 : 1ull) >= rd_last + 1);
    const unsigned int n32 = (unsigned int)sp.n, s32 = (unsigned int)s;
    unsigned int emax = 0u, run = 0u, yprev = s32;
    bool tr = false;                                                   // yprev is being tracked (a run of equal keys is open)
    B kprev = 0;
    load_panel(0u, edge, RMI_LN_NT_ERR != 0);
    for (unsigned int p16 = 0; __any(p16 < len); p16 += 16) {
      wave_sync();
      stage();
      load_panel(p16 + 16u, edge, RMI_LN_NT_ERR != 0);
      wave_sync();
      B kk[16];
#pragma unroll
      for (int q = 0; q < 16; q++) kk[q] = panel[lane * LN_STRIDE + q];
      const unsigned int rem = len > p16 ? len - p16 : 0u;
      const unsigned int vmask = rem >= 16u ? 0xFFFFu : ((1u << rem) - 1u);
      unsigned int dmask = 0;
      {
        B kp = kprev;
#pragma unroll
        for (int q = 0; q < 16; q++) { dmask |= (bits_to_key<K>(kk[q]) == bits_to_key<K>(kp)) ? (1u << q) : 0u; kp = kk[q]; }
      }
      if (p16 == 0) dmask &= ~1u;                                      // a leaf's first key differs from the key before it
      dmask &= vmask;
      kprev = kk[15];
      const bool full = __all(vmask == 0xFFFFu);
      const bool anygen = __any((tr && rem > 0u) || dmask != 0u);
      const unsigned int i0 = s32 + p16;
      auto steps = [&](auto full_tag, auto gen_tag) {
        constexpr bool FULL = decltype(full_tag)::value, GEN = decltype(gen_tag)::value;
#pragma unroll
        for (int q = 0; q < 16; q++) {
          if (FULL || ((vmask >> q) & 1u)) {
            const double x = KeyTraits<K>::as_float(bits_to_key<K>(kk[q]));
            const double f = __builtin_fma(pb, x, pa);                  // linear.rs:87-90
            const unsigned int pr = min(sg_cvt_u32(f), n32);           // models/mod.rs:735-737, two_layer.rs:14-18
            const unsigned int idx = i0 + (unsigned int)q;
            if constexpr (GEN) {
              const bool dup = (dmask >> q) & 1u;
              if (!dup && (p16 | (unsigned int)q) != 0u) run = max(run, idx - yprev);   // a new key value ends the run before it (lower_bound_correction.rs:108-119)
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
        if (rem > 0u && !tr) yprev = p16 ? i0 - 1u : s32;              // the key before this panel is its own first occurrence
        steps(std::false_type{}, std::true_type{});
        if (rem > 0u) tr = rem <= 16u || ((dmask >> 15) & 1u) != 0u;   // (a row that ends here keeps the y of its last key)
      }
    }
    if (eact) {
      // the key behind the leaf is a different one: it ends the run of the leaf's last key (the globally last run is
      // never recorded, Q5)
      if (e < sp.n) { const unsigned int yl = tr ? yprev : (unsigned int)e - 1u; run = max(run, (unsigned int)e - yl); }
      leaf_maxerr[j] = (unsigned long long)emax;
      leaf_run[j] = run > 1u ? (unsigned long long)run : 0ull;          // (runs of 1: k_finalize's rule, like pass B)
    }
  }
  if (flags) atomicOr(&st->err_flags, flags);
}

}  // namespace rmi
