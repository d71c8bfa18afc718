// rmi_scan.hip.h -- pipeline 5: the KEY-PARALLEL, one-read kernel for `linear_spline` leaves (gfx950, wave64).
//
// A linear_spline leaf is the line through the first and the last point of its container (linear_spline.rs:13-35): no
// recurrence, so (alpha, beta) of leaf j are known as soon as the two leaf boundaries around it are, and the last-level
// error pass (two_layer.rs:207-217) is embarrassingly parallel over the keys.  k_spline_scan does the whole leaf path in
// ONE pass over the keys, with the bucketing done by the scan north_star names (models/mod.rs:735-737 per key,
// two_layer.rs:43-50): no search kernel, no fill kernels, no leaf table read back.
//
//   * A wave is autonomous (block = 64 threads, no barrier anywhere) and persistent: it takes tile after tile of 64 V
//     consecutive keys.  The tile comes in by coalesced 16-byte non-temporal loads into registers -- the NEXT tile's
//     loads are issued before the arithmetic of the current one --, is laid out in LDS with a padded row per lane
//     (row stride 4 (2 m + 1) dwords: conflict-free ds_read_b128) and read back BLOCKED: lane l holds the keys
//     [A + l V, A + (l + 1) V).  Everything per key is then sequential inside a lane, in registers.
//   * P1  per key: root target (exact: the reference's floor / clamp), a bit per key "starts a leaf" (target differs
//         from the previous key's) and a bit "new key value" (FixDups, models/mod.rs:154-185: y = first occurrence).
//         One wave scan (DPP) turns the per-lane counts into slot numbers (the tile's n-th non-empty leaf) and the
//         per-lane last heads into the y carried into each lane.
//   * P2  the lanes that hold a leaf start leave a record (start, leaf id, first empty leaf in front of it, y of the key
//         before) in LDS; monotonicity (two_layer.rs:50) is checked there: a decreasing target is a boundary.
//   * ext the leaf that is still open at the tile's end belongs to this wave (a leaf belongs to the tile it STARTS in):
//         its end is looked for in the 128 (64) keys behind the tile, which came with the tile's loads, then in steps of
//         64 keys from the key array; a leaf that runs on for more than `long_min` keys goes to the list kernels.
//   * P3  lane q = slot q: container by the closed form of two_layer.rs:20-99 (leaf_container, Q2-Q4), end points, the
//         reference's two operations -> (alpha, beta) into LDS.
//   * P4  per key again, from the registers: prediction (linear.rs:87-90, models/mod.rs:735-737), |pred - y|, the
//         running maximum of the lane's stretch of a leaf -> ds_max_u32 on the slot when the leaf changes; run lengths
//         of equal keys (lower_bound_correction.rs:104-119) likewise.  A tile without a duplicate key takes a variant
//         without y and runs.
//   * P5  lane q = slot q: finalize_one_pre (two_layer.rs:185-197, 226-259), row (codegen.rs:288-315), counts, the terms of
//         the aggregates (two_layer.rs:267-287) into per-lane accumulators that live as long as the wave; empty leaves in
//         front of a leaf start are finished by the lane that holds the start (long gaps: by the whole wave).
//   * The ORDINARY tile -- inside the launch, a root whose targets are monotone by arithmetic (linear with a slope >= 0, radix over a
//     common prefix), at most one leaf start per lane, the open leaf's end within the look-ahead, the split of the 2-way join not
//     nearby -- takes a short form of the same phases: the leaf start of a lane by bisection of its row (5 targets instead of 32),
//     slot numbers from one ballot, containers [s - 1, e] without the closed form, an error pass without a test per key (the model
//     switches where some lane's start lies: a scalar test), the widening in 32 bits.  Anything else falls back to the general form
//     BEFORE a byte is stored: both forms write the same values.
// No order-dependent reduction anywhere: integers and coefficients are the oracle's bit for bit by construction.
// Algorithmic bytes: N sizeof(key) + 24 L; this kernel reads every key once (+ 1/16 of look-ahead that the
// neighbouring wave of the same XCD has in L2) and writes rows, leaf_start and -- unless `lean` -- params, err, count.
#pragma once
#include <type_traits>

#include "rmi_lanes.hip.h"
#ifndef RMI_SC_DEBUG
#define RMI_SC_DEBUG 0                // debugging: printf of the boundary records (tiny inputs only)
#endif
#ifndef RMI_SC_STOP
#define RMI_SC_STOP 0                 // debugging: leave a tile after phase n (results wrong)
#endif
#ifndef RMI_SC_F5BATCH
#define RMI_SC_F5BATCH 1              // the short form's leaf ends for several tiles at once (4-byte keys): see `FB` in k_spline_scan
#endif
#ifndef RMI_SC_DIAG
#define RMI_SC_DIAG 0                 // timing experiments on the short form's leaf ends (results wrong): & 1 no aggregate terms, & 2 no stores
#endif
#ifndef RMI_SC_PROF
#define RMI_SC_PROF 0                 // development: cycles per phase of the short form, printed by a few waves (perturbs the kernel)
#endif
#if RMI_SC_PROF
#define SC_TICK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof[k] += t_ - tlast; tlast = t_; } while (0)
#define SC_GT(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); gprof[k] += t_ - glast; glast = t_; } while (0)
#else
#define SC_TICK(k) do { } while (0)
#define SC_GT(k) do { } while (0)
#endif
#ifndef RMI_SC_BRANCHFREE
#define RMI_SC_BRANCHFREE 0           // the error pass switches a lane's model without the scalar test: one basic block for all keys of a row
#endif
#ifndef RMI_SC_NSUB
#define RMI_SC_NSUB 1                 // tiles per big tile: 1 (2: the short form's lanes hold two rows -- fewer instructions a key, but 21.7 KB of LDS and spills: 0.59 against 0.52 ms)
#endif
#ifndef RMI_SC_FARB
#define RMI_SC_FARB 8                 // FAR = 2: blocks of 64 keys of an open leaf's far part read per trip
#endif
#ifndef RMI_SC_FAR_MAX
#define RMI_SC_FAR_MAX 262144         // FAR = 2: an open leaf's end is looked for this many keys behind the tile (beyond: the general form lists the leaf; a leaf of that many keys is ~1.2 ms on its wave, the list kernels cost a training ~2.3 ms)
#endif
#ifndef RMI_SC_WPE0
#define RMI_SC_WPE0 3                 // waves per SIMD of the short form's kernel (PHASE 0: 168 registers)
#endif
#ifndef RMI_SC_FAST
#define RMI_SC_FAST 1                 // 0: every tile through the general form
#endif
#ifndef RMI_SC_FAST_DUPS
#define RMI_SC_FAST_DUPS 1            // 0: tiles with duplicate keys through the general form
#endif
#ifndef RMI_SC_CLAMPFREE
#define RMI_SC_CLAMPFREE 0            // the short form's error pass without the clamp to n where no prediction of the tile can reach it
#endif
#ifndef RMI_SC_WPE
#define RMI_SC_WPE 2                  // waves per SIMD the kernel is compiled for (register budget 512 / that)
#endif
#include "rmi_scan_launch.h"

namespace rmi {

template <typename K, int V> struct ScGeom {
  static constexpr int DW = (int)sizeof(K) / 4;          // dwords per key
  static constexpr int KPC = 16 / (int)sizeof(K);        // keys per 16-byte chunk
  static constexpr int TILE = 64 * V;                    // keys per tile
  static constexpr int ROWD = V * DW;                    // dwords of a lane's keys
  static constexpr int S = ROWD + 4;                     // padded row stride in dwords: 4 x odd
  static constexpr int NCH = ROWD / 4;                   // 16-byte chunks per lane and tile
  static constexpr int EXTC = 32;                        // chunks of look-ahead behind the tile
  static constexpr int EXTN = EXTC * KPC;                // ... in keys: 128 (4-byte keys), 64 (8-byte keys)
  static constexpr int FHC = 4;                          // chunks in front of the tile (FixDups offsets of the keys around a tile's first key)
  static constexpr int FHN = FHC * KPC;                  // ... in keys: 16 / 8
  // ONE address rule for the front chunks, the tile and the look-ahead: the key with tile-relative index rel (in [-FHN, TILE + EXTN))
  // has its first dword at T0 + d + 4 floor(d / ROWD), d = rel DW -- rows of ROWD dwords, 4 dwords of padding behind each
  static constexpr int T0 = FHC * 4 + 4;
  // A wave takes NSUB consecutive tiles at a time (a "big tile": 128 rows): the short form of an ordinary big tile gives a lane NSUB
  // consecutive rows (VF keys) -- the phases around the error pass cost the same for twice the keys --, the general form takes the
  // big tile as NSUB tiles one after the other, out of the same LDS image.
  static constexpr int NSUB = RMI_SC_NSUB;
  static constexpr int BTILE = NSUB * TILE, NCHB = NSUB * NCH, VF = NSUB * V;
  static constexpr int LDS_DW = T0 + NSUB * 64 * S + (EXTC * 4 / ROWD) * S;
  static constexpr int LOGV = V == 32 ? 5 : (V == 16 ? 4 : (V == 8 ? 3 : -1));
  static constexpr int LOGVF = LOGV + 1;
  static_assert(VF <= 64, "a bit per key of a lane in a 64-bit mask");
  static_assert(V <= 32 && LOGV > 0, "a bit per key in a 32-bit mask");
  static_assert(ROWD == 32, "the address rule shifts by 5");
  static_assert(FHC + EXTC <= 64, "one aux chunk per lane");
};
// roots whose targets are monotone in the key by arithmetic (given a slope >= 0 / a common prefix: ScanArgs::mono)
template <int ROOT> struct ScMono { static constexpr bool value = ROOT == K_LINEAR || ROOT == K_RADIX; };
constexpr int SC_SLOTS = 64;                             // leaves per batch: one per lane in P3 / P5

// DPP steps of the wave scans (gfx9 row_shr / row_bcast; lanes without a source keep `old`)
template <int CTRL, int RM> __device__ __forceinline__ unsigned int sc_dpp(unsigned int old, unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, RM, 0xF, false);
}
__device__ __forceinline__ unsigned int sc_scan_add(unsigned int v) {          // inclusive prefix sum over the 64 lanes
  v += sc_dpp<0x111, 0xF>(0u, v); v += sc_dpp<0x112, 0xF>(0u, v); v += sc_dpp<0x114, 0xF>(0u, v); v += sc_dpp<0x118, 0xF>(0u, v);
  v += sc_dpp<0x142, 0xA>(0u, v);                                                // row_bcast:15 into rows 1, 3
  v += sc_dpp<0x143, 0xC>(0u, v);                                                // row_bcast:31 into rows 2, 3
  return v;
}
__device__ __forceinline__ unsigned int sc_scan_max(unsigned int v) {          // inclusive prefix maximum
  v = max(v, sc_dpp<0x111, 0xF>(0u, v)); v = max(v, sc_dpp<0x112, 0xF>(0u, v)); v = max(v, sc_dpp<0x114, 0xF>(0u, v)); v = max(v, sc_dpp<0x118, 0xF>(0u, v));
  v = max(v, sc_dpp<0x142, 0xA>(0u, v));
  v = max(v, sc_dpp<0x143, 0xC>(0u, v));
  return v;
}
__device__ __forceinline__ unsigned int sc_lane63(unsigned int v) { return (unsigned int)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ unsigned int sc_wave_or(unsigned int v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v |= (unsigned int)__shfl_xor((int)v, d);
  return (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ unsigned int sc_wave_max(unsigned int v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = max(v, (unsigned int)__shfl_xor((int)v, d));
  return (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ unsigned int sc_mask_below(unsigned int x) { return x >= 32u ? ~0u : ((1u << x) - 1u); }

// log2 of an integer 2 <= v <= 2^34 given as a double (the term log2(2 err + 2) of two_layer.rs:279-283), to the last bit or two: the
// library's log2 is 85 instructions of double-double arithmetic per leaf in a phase where a third of the lanes work -- 11 % of the short form's
// time.  Here: v = m 2^e with m in [sqrt(1/2), sqrt(2)), log(m) by fdlibm's e_log.c (s = f / (2 + f), a polynomial of degree 7 in s^2), the
// reciprocal by v_rcp_f64 and two Newton steps: 30 instructions, relative error <= 2.3e-16 against the library over every even v below
// 2 10^5 and 2 10^5 random ones below 2^33 (tools/log2_check.py).  The aggregates are compared to 1e-9 (the sums run in another order
// than the reference's anyway).
__device__ __forceinline__ double sc_log2_int(double v) {
  double m = __builtin_amdgcn_frexp_mant(v) * 2.0;                                // [1, 2)
  int e = __builtin_amdgcn_frexp_exp(v) - 1;
  const bool big = m > 1.4142135623730951;
  m = big ? m * 0.5 : m; e = big ? e + 1 : e;
  const double f = m - 1.0, d = 2.0 + f;
  double r = __builtin_amdgcn_rcp(d);
  r = r * __builtin_fma(-d, r, 2.0); r = r * __builtin_fma(-d, r, 2.0);
  const double s = f * r, z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double hfsq = 0.5 * f * f;
  const double lnm = f - (hfsq - s * (hfsq + (t1 + t2)));
  return __builtin_fma(lnm, 1.4426950408889634, (double)e);
}

// per-lane accumulators of the aggregates (two_layer.rs:267-287)
struct ScAgg {
  unsigned long long mx, mi, sum; double l2, lg;
  __device__ __forceinline__ void add(uint64_t j, uint64_t final_err, uint64_t cnt_j, double nf) {
    if (final_err > mx || (final_err == mx && j > mi)) { mx = final_err; mi = j; }   // max_by_key: the LAST maximum
    const unsigned long long ts = cnt_j * final_err;                            // wrapping u64, like the reference's sum
    sum += ts;
    if (cnt_j) {
      const double v = (double)ts;
      l2 += (v * v) / nf;
      lg += (double)cnt_j * log2((double)(2 * final_err + 2));
    }
  }
};

// The kernarg segment, in the CONSTANT address space: a load through this pointer is a scalar load (through a generic pointer it
// is a flat VECTOR load of host-visible memory -- microseconds, and a wait for every key load in flight).
// (the builtin exists in the device pass only; the host pass merely parses the kernel)
#define SC_AS4 __attribute__((address_space(4)))
typedef const SC_AS4 unsigned char* sc_kargp;
__device__ __forceinline__ sc_kargp sc_kernarg_ptr() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (sc_kargp)__builtin_amdgcn_kernarg_segment_ptr();
#else
  return nullptr;
#endif
}
// field `m` (type T) of the ScanArgs behind the kernarg pointer p
template <typename T> struct ScAs4 { typedef const T SC_AS4* ptr; };
#define SC_ARG(p, T, m) (*reinterpret_cast<typename ScAs4<T>::ptr>((p) + offsetof(ScanArgs, m)))

// The kernel's arguments: ONE plain struct.  The kernel never names its parameter: it reads the fields through the kernarg segment
// pointer, the hot ones once into registers, the cold ones (output pointers, peers' tables, the list) where they are used (SC_ARG),
// behind a compiler barrier on the pointer -- named parameters are all loaded at the kernel's entry and then live (= spilled: 100+
// SGPRs) for the whole kernel.
struct ScanArgs {
  const void* keys;               // pre-offset: keys[global index]
  long long tile0;                // global index of the first tile's first position (<= it_lo, a 128-byte line of the key array)
  unsigned int ntiles, tiles_per_xcd;
  unsigned int long_min;
  int host_split;
  int mono;                       // the root's targets are monotone in the key by arithmetic (the host has checked slope / prefix)
  DevState* st;
  Span sp;
  RootP r;
  // cold
  ScanOut out;
  SgList fl;
  PeerRows peers;
  GapRec* gaps;
  unsigned long long* gap_cnt;
  unsigned int* tile_list;        // PHASE 0: the tiles it leaves to the general form; PHASE 1: the tiles to take (null: all tiles of the launch)
  unsigned long long* tile_cnt;
};

// PHASE 0: the short form alone, at RMI_SC_WPE0 waves per SIMD -- the general form's registers are what kept the whole kernel at 2 --; a tile
//          it cannot take (an end of the launch, a lane with two leaf starts, a long gap or leaf, the split nearby) goes on a list.
// PHASE 1: the general form over the tiles of that list, or -- a root that is not monotone by arithmetic: no PHASE 0 -- over all tiles.
// FAR (PHASE 0): the end of the leaf that is open at a tile's end is looked for BEHIND the look-ahead too (in the key array); the launcher takes this variant
//          where the leaves are longer than the look-ahead on average -- the other keeps C5's code as it was (the far search costs it 2-3 %: registers).
//          FAR = 1: 64 keys a step, in the search and in the error pass (each step a memory round trip: fine while a leaf ends a block or two behind the
//          look-ahead).  FAR = 2, for leaves of several hundred keys and more: ONE gather finds the block that holds the end (lane l probes the last key of
//          the l-th block), the error pass reads eight blocks a trip -- 200 M u64 keys in 2^16 leaves 1.19 -> 0.64 ms.  An instance of its own because its
//          code left 5 spilled registers around the tile's error pass (FAR = 1 has none; M's keys in 2^20 spline leaves 0.43 -> 0.49 ms with them).
template <int ROOT, typename K, int V, int PHASE, int FAR = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PHASE == 0 ? RMI_SC_WPE0 : RMI_SC_WPE, PHASE == 0 ? RMI_SC_WPE0 : RMI_SC_WPE))) k_spline_scan(ScanArgs) {
  const sc_kargp kp = sc_kernarg_ptr();
  const ScanArgs* const ka = reinterpret_cast<const ScanArgs*>((const unsigned char*)kp);
  const K* const keys = (const K*)ka->keys;
  const long long tile0 = ka->tile0;
  const unsigned int ntiles = ka->ntiles, tiles_per_xcd = ka->tiles_per_xcd, long_min = ka->long_min;
  const int host_split = ka->host_split;
  const bool mono = ka->mono != 0;
  const unsigned int h_split = host_split ? (unsigned int)ka->st->split_idx : 0u;   // (a shard: the host has put the split there before the launch)
  DevState* const st = ka->st;
  const Span sp = ka->sp;
  const RootP r = ka->r;
  // the cold arguments, re-read where they are used
  auto cold = [&]() -> sc_kargp { sc_kargp p = kp; asm volatile("" : "+s"(p)); return p; };
  auto cold_out = [&](sc_kargp p) -> ScanOut {
    ScanOut o;
    o.leaf_start = SC_ARG(p, unsigned long long*, out.leaf_start); o.params = SC_ARG(p, double*, out.params);
    o.leaf_err = SC_ARG(p, unsigned long long*, out.leaf_err); o.leaf_count = SC_ARG(p, unsigned long long*, out.leaf_count);
    o.rows = SC_ARG(p, unsigned char*, out.rows); o.partials = SC_ARG(p, StatsPartial*, out.partials);
    return o;
  };
  auto cold_peer = [&](sc_kargp p, int i) -> unsigned char* { return reinterpret_cast<typename ScAs4<unsigned char*>::ptr>(p + offsetof(ScanArgs, peers.tab))[i]; };
  using G = ScGeom<K, V>;
  using B = typename LnBits<K>::type;
  constexpr int DW = G::DW, KPC = G::KPC, TILE = G::TILE, S = G::S, NCH = G::NCH, EXTN = G::EXTN, FHC = G::FHC, FHN = G::FHN, T0 = G::T0;
  constexpr int NSUB = G::NSUB, BTILE = G::BTILE, NCHB = G::NCHB, VF = G::VF;
  __shared__ __attribute__((aligned(16))) unsigned int lds[G::LDS_DW];
  // (FB, below: the gfx950 LDS is dealt out in granules of 1 280 bytes and 12 waves a CU get 10 of them -- 12 800 B; the kernel used 12 552.  So the batched
  //  form counts bytes: no r_g0 -- the number of empty leaves in front of a start, at most 4 in an ordinary tile, rides in the top bits of r_t --, no spare entries)
  constexpr bool FB = PHASE == 0 && RMI_SC_F5BATCH != 0 && ScMono<ROOT>::value;
  constexpr bool FBK = FB && sizeof(K) == 4;            // the pending slots' end keys in LDS (8-byte keys: read from the key array again when the ends are run)
  constexpr int NS1 = FB ? SC_SLOTS : SC_SLOTS + 1, NS2 = FB ? SC_SLOTS + 1 : SC_SLOTS + 2;
  __shared__ unsigned int r_s[NS1], r_t[NS1], r_g0[FB ? 1 : SC_SLOTS + 1], r_yp[NS1];                      // boundary records of a batch
  __shared__ __attribute__((aligned(16))) double m_ab[2 * NS2];                                              // (alpha, beta) per slot, one entry of padding in front (and behind)
  __shared__ unsigned int m_err[NS2], m_run[NS2];
  // FB: the short form's leaf ends (F5) are run for the slots of SEVERAL tiles at once.  An ordinary tile of C5 starts ~21 leaves: F3 and F5 -- lane =
  // slot, ~64 us of 400 by a knock-out build, a chain of LDS reads, conversions, a logarithm and four scattered stores -- ran with a third of the
  // lanes.  The slots of a tile are appended to the tables behind those still pending (`cnt`), with their ends and their containers' end keys
  // (the tile image is gone when the ends are computed); the ends run when the next tile's starts might not fit.  4-byte keys: the containers' first
  // keys take the place of r_yp, which a slot needs only until its model is computed (F3); 8-byte keys (their end keys would take another 768 B): the two
  // keys of a slot are read from the key array again when the ends are run -- two scattered loads a lane, once per batch of up to 64 leaves.
  // With 13 072 B a wave -- 11 waves a CU, the twelfth of the launch a second round -- the batched form ran 0.51 against 0.45 ms.
  __shared__ unsigned int r_e[FB ? SC_SLOTS : 1], r_khi[FBK ? SC_SLOTS : 1];
  unsigned int* const r_end = FB ? r_e : r_s + 1;                       // end of slot i (not batched: the start of slot i + 1)
  unsigned int* const r_klo = r_yp;                                     // (FB) written by the lane that has just read r_yp of the same slot

  unsigned int* const trow0 = lds + T0;               // row r of the big tile at trow0[r S ...]; the front chunks and the look-ahead by the same rule (ScGeom)

  const int lane = threadIdx.x;
  const unsigned int base32 = (unsigned int)(unsigned long long)tile0;          // global index of relative index 0 (mod 2^32)
  const K* const kb = keys + tile0;                                             // kb[relative index]
  const unsigned int rel_lo = (unsigned int)((long long)sp.it_lo - tile0), rel_hi = (unsigned int)((long long)sp.it_hi - tile0);
  const long long rd_lo_rel = (long long)sp.rd_lo - tile0, rd_hi_rel = (long long)sp.rd_hi - tile0;
  const unsigned int n32 = (unsigned int)sp.n;
  const unsigned int n_it = rel_hi - rel_lo;
  const unsigned int Lm1 = (unsigned int)root_cap<ROOT>(r);
  const double Lm1f = (double)(r.L - 1);
  const unsigned int mid = (unsigned int)(r.L / 2);                              // two_layer.rs:131
  const double nf = (double)sp.n;
  const double inv_nf = 1.0 / nf;
  unsigned int flags = 0;

  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto bits_at = [&](const unsigned int* p) -> B {
    if constexpr (DW == 1) return (B)p[0];
    else return (B)(*reinterpret_cast<const unsigned long long*>(p));
  };
  // min(L - 1, predict_to_int(key)) (models/mod.rs:735-737, two_layer.rs:49).  radix.rs:43-50 over 4-byte keys (widened to 64 bits,
  // models/mod.rs:474-478): with 32 <= prefix and 1 <= bits <= 32 the two 64-bit shifts are two 32-bit shifts of the key
  unsigned int rsh1 = 0u, rsh2 = 0u, oobcap32 = 0xFFFFFFFFu;
  bool r32 = false;
  if constexpr (ROOT == K_RADIX) {
    const unsigned int pfx = r.prefix & 63u, sh = (64u - r.bits) & 63u;
    if constexpr (std::is_same<K, uint32_t>::value) { r32 = pfx >= 32u && sh >= 32u; rsh1 = pfx - 32u; rsh2 = sh - 32u; }
    oobcap32 = r.oob_cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned int)r.oob_cap;
  }
  auto target_of = [&](K k, bool& oob) -> unsigned int {
    if constexpr (ROOT == K_RADIX) {
      if constexpr (std::is_same<K, uint32_t>::value) {
        if (r32) {                                                              // (wave-uniform)
          const unsigned int p32 = ((unsigned int)k << rsh1) >> rsh2;
          oob = p32 > oobcap32;
          return min(p32, Lm1);
        }
      }
      const uint64_t v = KeyTraits<K>::as_uint(k);
      const uint64_t pr = (v << (r.prefix & 63u)) >> ((64u - r.bits) & 63u);
      oob = pr > r.oob_cap;
      return (unsigned int)(pr < r.cap ? pr : r.cap);
    } else {
      return s2_target<ROOT, K>(r, Lm1f, Lm1, k, KeyTraits<K>::as_float(k), oob);
    }
  };

  // ---- big-tile loads.  A PLAIN big tile -- every chunk of it, of its front chunks and of its look-ahead readable, every position of tile
  //      and look-ahead a key of this launch -- comes in by NCHB coalesced 16-byte non-temporal loads per lane (chunk c 64 + lane) + one aux
  //      chunk (lanes 0 .. FHC - 1: the chunks in front; lanes FHC .. FHC + EXTC - 1: the look-ahead), issued a big tile ahead.  The others (the
  //      two ends of a launch) are staged chunk by chunk when their turn comes (stage_slow below): their tests and 64-bit index arithmetic stay out of
  //      the registers of the loop.
  uint4 pf[NCHB], pfx;
  auto plain = [&](unsigned int tile) -> bool {                                // (wave-uniform)
    const long long a = (long long)tile * BTILE;
    return a - FHN >= rd_lo_rel && a + BTILE + EXTN <= rd_hi_rel && a >= (long long)rel_lo + 1 && a + BTILE + EXTN + 1 <= (long long)rel_hi;
  };
  // (`ln`: the lane number behind a compiler barrier inside the loop -- the per-chunk addresses are then recomputed per big tile, two
  //  instructions each, instead of being kept as 40 loop invariants in registers the loop does not have)
  auto load_tile = [&](unsigned int tile, int ln) {
    typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
    const long long a = (long long)tile * BTILE;
    const raw_t* const p0 = reinterpret_cast<const raw_t*>(kb + a) + ln;
#pragma unroll
    for (int c = 0; c < NCHB; c++) { const raw_t rw = __builtin_nontemporal_load(p0 + c * 64); pf[c] = make_uint4(rw.x, rw.y, rw.z, rw.w); }
    pfx = make_uint4(0u, 0u, 0u, 0u);
    if (ln < FHC + G::EXTC) {
      const long long ax = ln < FHC ? a - (long long)(FHC - ln) * KPC : a + BTILE + (long long)(ln - FHC) * KPC;
      const raw_t rw = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(kb + ax));
      pfx = make_uint4(rw.x, rw.y, rw.z, rw.w);
    }
  };
  // persistent waves; block b runs on XCD b % 8 (observed, for speed only): each XCD streams a contiguous range of tiles, so
  // that the look-ahead of a tile is the neighbouring wave's tile in the same L2
  const unsigned int xcd = blockIdx.x & 7u, wix = blockIdx.x >> 3, wpx = (gridDim.x + 7u - xcd) >> 3;   // this wave's index among the wpx waves of its XCD
  const unsigned int t_lo = xcd * tiles_per_xcd, t_hi = min(ntiles, t_lo + tiles_per_xcd);
  ScAgg agg{0ull, 0ull, 0ull, 0.0, 0.0};
  unsigned int amx = 0u, ami = 0u;                    // PHASE 0's aggregates: maximum and its leaf (32 bits there), the integer sum; the float sums in LDS
  unsigned long long asum = 0ull;
  // ... in the 16 bytes of padding behind row `lane` of the tile image, which no staging store touches (in registers they were spilled: 12 waves
  // per CU leave 168; an array of their own took the LDS over what 12 waves can have)
  double* const aggp = reinterpret_cast<double*>(trow0 + lane * G::S + G::ROWD);
  if constexpr (PHASE == 0) { aggp[0] = 0.0; aggp[1] = 0.0; }
  unsigned int pend = 0u;                                                       // (FB) slots whose ends are pending
  unsigned int nbmax = 0u;                                                      // (FB) the most leaf starts a tile of this wave has had
  // the sequence of this wave's tiles: position k0, k0 + kstep, ... below kend; the tile at a position is the position itself, or the list's entry
  const unsigned int* const tlist = PHASE == 1 ? SC_ARG(kp, unsigned int*, tile_list) : (const unsigned int*)nullptr;
  unsigned int kpos = t_lo + wix, kstep = wpx, kend = t_hi;
  if (PHASE == 1 && tlist != nullptr) {
    const unsigned long long nl = *SC_ARG(kp, unsigned long long*, tile_cnt);
    kpos = blockIdx.x; kstep = gridDim.x; kend = (unsigned int)(nl < (unsigned long long)ntiles ? nl : (unsigned long long)ntiles);
  }
  auto tile_at = [&](unsigned int k) -> unsigned int { return (PHASE == 1 && tlist != nullptr) ? tlist[k] : k; };
  unsigned int tile = kpos < kend ? tile_at(kpos) : 0u;
  if (kpos < kend && plain(tile)) load_tile(tile, lane);

#if RMI_SC_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast;
  unsigned int ntile = 0, nfast = 0;
#endif
  for (; kpos < kend; kpos += kstep) {
    tile = tile_at(kpos);
    const bool more_t = kpos + kstep < kend;
    const unsigned int tile_nx = more_t ? tile_at(kpos + kstep) : 0u;
#if RMI_SC_PROF
    ntile++;
#endif
    SC_TICK(7);
    // (FB) the pending leaf ends, HERE -- nothing of a tile is live, only the loads of this one in flight -- when another tile's starts might not fit
    // behind them: as many as the fullest tile so far, and 4 more (evenly filled leaves: the count varies by one or two from tile to tile)
    if constexpr (FB) {
      if (pend && pend + nbmax + 4u > (unsigned int)SC_SLOTS) {
        const unsigned int ends_count = pend, ends_A2 = 0u;
#include "rmi_scan_ends.inc.h"
        pend = 0u;
      }
    }
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const unsigned int relA2 = tile * (unsigned int)BTILE;                      // relative index of the big tile's first key
    const unsigned int A2 = base32 + relA2;                                     // ... and its global index
    // ---- validity.  The big tiles at the two ends of the launch hold positions outside [it_lo, it_hi): those take the value of the
    //      nearest valid key on their way to LDS -- in front the launch's first key, behind its last key the key that follows it in the key
    //      set (the next shard's first) or, at the end of the data, the last key itself (no run is recorded there, Q5): no leaf start and no
    //      new key value arises among them; the launch's first key and the position behind its last key are leaf starts by decree (below).
    const bool edge2 = relA2 < rel_lo + 1u || relA2 + (unsigned int)BTILE + (unsigned int)EXTN + 1u > rel_hi;   // (wave-uniform)
    const bool plain_t = plain(tile);                                           // (implies !edge2)
    // PHASE 0 leaves a tile to the general form: on the list
    auto leave_tile = [&]() {
      if (lane == 0) {
        const unsigned long long pos = atomicAdd(SC_ARG(kp, unsigned long long*, tile_cnt), 1ull);
        SC_ARG(kp, unsigned int*, tile_list)[pos] = tile;
      }
    };
    // (ONE place that issues the next tile's loads, below: a second one makes the compiler carry the tile in two register sets)
    const bool take = PHASE == 1 || (mono && plain_t && relA2 >= rel_lo + (unsigned int)FHN + 1u && (!FB || r.L <= (1ull << 29)));
    // ---- stage the big tile (padded rows) and the aux chunks
    wave_sync();
    if (!take) {
    } else if (PHASE == 0 || plain_t) {
      // (dword offset of chunk c 64 + ln from the big tile's first key: d0 = 256 c + 4 ln, and d0 + 4 (d0 >> 5) = (4 ln + 4 (ln >> 3)) + 288 c
      //  for ln < 64 -- one address and immediate offsets; from the general rule the compiler forms every chunk's address by itself: 6
      //  instructions each)
      unsigned int* const srow = trow0 + 4 * ln + 4 * (ln >> 3);
#pragma unroll
      for (int c = 0; c < NCHB; c++) *reinterpret_cast<uint4*>(srow + 288 * c) = pf[c];
      if (ln < FHC + G::EXTC) {
        const int d0 = ln < FHC ? (ln - FHC) * 4 : NSUB * 64 * G::ROWD + (ln - FHC) * 4;
        *reinterpret_cast<uint4*>(trow0 + d0 + 4 * (d0 >> 5)) = pfx;
      }
    } else {
      // stage_slow: chunk by chunk from the key array; a chunk is loaded iff it overlaps the readable keys [rd_lo, rd_hi)
      const long long a = (long long)relA2;
      B k_first = 0, k_last = 0;
      // (behind the last key of a SHARD the next shard's first key follows: a different key -- leaf-aligned cuts --, so the run of equal keys
      //  that ends with the shard's last key IS recorded, lower_bound_correction.rs:108-119; behind the last key of all nothing follows, Q5)
      if (n_it > 0u) { k_first = (B)key_to_bits<K>(kb[rel_lo]); k_last = (B)key_to_bits<K>(sp.it_hi < sp.n ? keys[sp.it_hi] : kb[rel_hi - 1u]); }
      constexpr int NCHT = FHC + NCHB * 64 + G::EXTC;                            // chunks of the LDS image
#pragma unroll 1
      for (int ch = lane; ch < NCHT; ch += 64) {
        const int d0 = (ch - FHC) * 4;
        const long long rel_first = a + (long long)(ch - FHC) * KPC;
        unsigned int w[4] = {0u, 0u, 0u, 0u};
        if (rel_first + KPC > rd_lo_rel && rel_first < rd_hi_rel) {
          const uint4 q = *reinterpret_cast<const uint4*>(kb + rel_first);
          w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        }
        if (edge2 && n_it > 0u) {
#pragma unroll
          for (int k = 0; k < KPC; k++) {
            const long long rel = rel_first + k;
            if (rel < (long long)rel_lo || rel >= (long long)rel_hi) {
              const B kv = rel < (long long)rel_lo ? k_first : k_last;
              if constexpr (DW == 1) w[k] = (unsigned int)kv;
              else { w[2 * k] = (unsigned int)kv; w[2 * k + 1] = (unsigned int)((unsigned long long)kv >> 32); }
            }
          }
        }
        *reinterpret_cast<uint4*>(trow0 + d0 + 4 * (d0 >> 5)) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    wave_sync();
    // key (raw bits) with the relative index rel0 from the big tile's first key, in [-FHN, BTILE + EXTN)
    auto lds_bits0 = [&](int rel0) -> B { const int d = rel0 * DW; return bits_at(trow0 + d + 4 * (d >> 5)); };
    SC_TICK(0);
    // ---- the next big tile's loads: in flight during everything below
    if (more_t && plain(tile_nx)) load_tile(tile_nx, ln);
    SC_TICK(1);
    if constexpr (PHASE == 0) { if (!take) { leave_tile(); continue; } }

    // y (FixDups offset, models/mod.rs:154-185) of the key in front of the (sub-)tile that starts hoff keys into the big tile, at the global
    // index A_ (relative index relA_): its run of equal keys is walked in the keys in front of it; a run that reaches beyond the front
    // chunks is looked up in the key array
    auto front_y = [&](int hoff, unsigned int A_, unsigned int relA_) -> unsigned int {
      unsigned int y = A_ - 1u;
      if (lane == 0 && relA_ > rel_lo) {
        const unsigned int lim = min((unsigned int)FHN, relA_ - rel_lo);         // the keys A_ - lim .. A_ - 1 belong to this launch
        const K k1 = bits_to_key<K>(lds_bits0(hoff - 1));
        unsigned int back = 1u;
        while (back < lim && bits_to_key<K>(lds_bits0(hoff - 1 - (int)back)) == k1) back++;
        if (back < lim || relA_ - rel_lo == back) y = A_ - back;
        else y = (unsigned int)first_occurrence(keys, (uint64_t)(tile0 + (long long)relA_ - 1), sp.rd_lo);
      }
      return (unsigned int)__builtin_amdgcn_readfirstlane((int)y);
    };
    // ================= the ordinary big tile (see the head of the file): same values as the general form below, fewer instructions =================
    if constexpr (ScMono<ROOT>::value && PHASE == 0) {
      {
        bool done = false;
        do {
          constexpr int LOGV = G::LOGV;
          const unsigned int* const rowp = trow0 + (NSUB * ln) * S;             // the lane's NSUB consecutive rows: VF keys
          const unsigned int f = A2 + (unsigned int)(lane * VF);                // global index of the lane's first key
          // (a row at a time in registers: the next big tile's loads hold 68 of them for the whole pass)
          B kk[V];
          auto read_row = [&](int rr) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
              const uint4 q = *reinterpret_cast<const uint4*>(rowp + rr * S + 4 * c);
              if constexpr (DW == 1) { kk[4 * c] = q.x; kk[4 * c + 1] = q.y; kk[4 * c + 2] = q.z; kk[4 * c + 3] = q.w; }
              else { kk[2 * c] = ((B)q.y << 32) | q.x; kk[2 * c + 1] = ((B)q.w << 32) | q.z; }
            }
          };
          const B kprev_b = lds_bits0(lane * VF - 1), knext_b = lds_bits0((lane + 1) * VF), klast_b = lds_bits0((lane + 1) * VF - 1);
          bool oob_x = false, oob_l = false;
          const unsigned int tp = target_of(bits_to_key<K>(kprev_b), oob_x);    // the key in front of the lane
          const unsigned int tl = target_of(bits_to_key<K>(klast_b), oob_l);    // the lane's last key: the largest prediction of the lane
          if constexpr (!root_needs_bounds_check<ROOT>()) { if (oob_l) flags |= EF_ROOT_OOB; }   // two_layer.rs:45-48
          const bool has = tl != tp;                                            // a leaf starts in this lane (targets are monotone)
          const unsigned long long hm = __ballot(has);
          if (hm == 0ull) { done = true; break; }                               // the big tile lies inside one leaf that started earlier
          // (FB: this tile's starts go behind the pending slots.  The loop's head has made room for as many as any tile of this wave has had so far,
          //  and a few more; a tile with still more than that -- and slots pending -- goes to the general form)
          if constexpr (FB) { if (pend + (unsigned int)__builtin_popcountll(hm) > (unsigned int)SC_SLOTS) break; }
          // ---- the look-ahead's targets and the duplicate tests first: independent of the bisection below, their LDS round trips overlap
          unsigned int t_ext[EXTN / 64];
          bool dq = false;
          {
#pragma unroll
            for (int o = 0; o < EXTN; o += 64) {
              bool o2;
              const K kx = bits_to_key<K>(lds_bits0(BTILE + o + lane));
              t_ext[o / 64] = target_of(kx, o2);
              dq = dq || (kx == bits_to_key<K>(lds_bits0(BTILE + o + lane - 1)));
            }
            if (lane == 0) dq = dq || (bits_to_key<K>(lds_bits0(-1)) == bits_to_key<K>(lds_bits0(-2)));
            K kp = bits_to_key<K>(kprev_b);
#pragma unroll
            for (int rr = 0; rr < NSUB; rr++) {
              read_row(rr);
#pragma unroll
              for (int v = 0; v < V; v++) { const K kv = bits_to_key<K>(kk[v]); dq = dq || (kv == kp); kp = kv; }
            }
          }
          bool o2x;
          const unsigned int t2 = target_of(bits_to_key<K>(lds_bits0(-2)), o2x);
          // ---- F1: the lane's first leaf start by bisection of its rows
          unsigned int p = (unsigned int)VF, t_hi = tl;
          {
            unsigned int lo = 0u, hi = (unsigned int)VF - 1u;                   // t(hi) != tp; t(v) == tp for v < lo
#pragma unroll
            for (int it = 0; it < G::LOGVF; it++) {
              const unsigned int mid = (lo + hi) >> 1;
              bool o2;
              const unsigned int tm = target_of(bits_to_key<K>(bits_at(rowp + (mid >> LOGV) * (unsigned int)S + (mid & (unsigned int)(V - 1)) * (unsigned int)DW)), o2);
              const bool ne = tm != tp;
              hi = ne ? mid : hi; t_hi = ne ? tm : t_hi; lo = ne ? lo : mid + 1u;
            }
            if (has) p = hi;
          }
          if (__any(has && (t_hi != tl || t_hi - tp > 5u))) break;              // a lane with more than one start, or more than 4 empty leaves in front of it
          bool dups = __any(dq) != 0;                                           // duplicates among the keys [A2 - 2, A2 + BTILE + EXTN) (... and of the open leaf behind them: below)
          if (dups && !RMI_SC_FAST_DUPS) break;
          // ---- the end of the leaf that is open at the big tile's end: in the look-ahead; else looked for in the key array, 64 keys a step, up to long_min
          //      keys (8-byte keys have 64 keys of look-ahead, and a leaf of M's 191 keys that is open at a tile's end runs on for 95 on average: two tiles in
          //      three went to the general form -- and their listing, one counter for all, cost more than their work: 2.26 ms for 200 M u64 keys)
          const unsigned int t_tile_last = sc_lane63(tl);
          unsigned int term_rel = 0u, tt = 0u;
          {
            bool found = false;
#pragma unroll
            for (int o = 0; o < EXTN; o += 64) {
              if (!found) {
                const unsigned long long dm = __ballot(t_ext[o / 64] != t_tile_last);
                if (dm) {
                  const int src = __builtin_ctzll(dm);
                  term_rel = (unsigned int)(BTILE + o + src);
                  tt = (unsigned int)__builtin_amdgcn_readlane((int)t_ext[o / 64], src);
                  found = true;
                }
              }
            }
            if (!found) {
              if constexpr (FAR == 0 || (FAR == 2 && !RMI_SC_FAST_DUPS)) break;
              if constexpr (FAR == 1) {
                bool dq_far = false;
                const unsigned int lim = long_min > (unsigned int)EXTN ? long_min : (unsigned int)EXTN;
                for (unsigned int o = (unsigned int)EXTN; o < lim && !found; o += 64u) {
                  const unsigned int relx = relA2 + (unsigned int)BTILE + o;        // launch-relative index of this step's first key
                  if (relx + 64u + 1u > rel_hi) break;                              // the launch's end: the general form
                  const K kx = kb[(long long)relx + lane], kxp = kb[(long long)relx + lane - 1];
                  bool o2;
                  const unsigned int t = target_of(kx, o2);
                  dq_far = dq_far || (kx == kxp);
                  const unsigned long long dm = __ballot(t != t_tile_last);
                  if (dm) {
                    const int src = __builtin_ctzll(dm);
                    term_rel = (unsigned int)BTILE + o + (unsigned int)src;
                    tt = (unsigned int)__builtin_amdgcn_readlane((int)t, src);
                    found = true;
                  }
                }
                if (!found) break;                                                  // (a leaf that runs on for more than long_min keys: the list kernels, through the general form)
                if (__any(dq_far)) { dups = true; if (!RMI_SC_FAST_DUPS) break; }
              }
              if constexpr (FAR == 2) {
                // lane l probes the LAST key of the l-th block of 64 keys behind the look-ahead: the targets are monotone, so the first block whose last key
                // has left the open leaf holds its end -- one gather instead of a trip per block (a leaf of 3 000 keys: 48 dependent trips), then that block's keys.
                // (Whether the keys behind the look-ahead repeat is not looked at here: the error pass treats them as if they might.)  Leaves of up to
                // SC_FAR_MAX keys stay with this wave -- the way over the general form and the list kernels costs a training ~2 ms whatever it lists.
                const unsigned int lim = long_min > (unsigned int)RMI_SC_FAR_MAX ? long_min : (unsigned int)RMI_SC_FAR_MAX;
                for (unsigned int o0 = (unsigned int)EXTN; o0 < lim && !found; o0 += 64u * 64u) {
                  const unsigned int ob = o0 + 64u * (unsigned int)lane;              // this lane's block: the keys [ob, ob + 64) behind the tile
                  const unsigned int relb = relA2 + (unsigned int)BTILE + ob;
                  const bool vb = ob < lim && relb + 64u + 1u <= rel_hi;              // (beyond the limit, or the launch's end: the general form)
                  bool o2;
                  const unsigned int tb_ = vb ? target_of(kb[(long long)relb + 63], o2) : t_tile_last;
                  const unsigned long long nem = __ballot(vb && tb_ != t_tile_last), ivm = __ballot(!vb);
                  const int fne = nem ? __builtin_ctzll(nem) : 64, fiv = ivm ? __builtin_ctzll(ivm) : 64;
                  if (fne < fiv) {
                    const unsigned int o = o0 + 64u * (unsigned int)fne;
                    const unsigned int t = target_of(kb[(long long)(relA2 + (unsigned int)BTILE + o) + lane], o2);
                    const unsigned long long dm = __ballot(t != t_tile_last);       // (not empty: the block's last key has another target)
                    const int src = __builtin_ctzll(dm);
                    term_rel = (unsigned int)BTILE + o + (unsigned int)src;
                    tt = (unsigned int)__builtin_amdgcn_readlane((int)t, src);
                    found = true;
                  } else if (fiv < 64) break;
                }
                if (!found) break;                                                  // (a leaf that runs on for more than long_min keys: the list kernels, through the general form)
              }
            }
          }
          // key with the tile-relative index rr: in the image up to the look-ahead's end, behind it from the key array (the open leaf's far keys)
          auto key_rel = [&](int rr) -> K {
            if constexpr (FAR != 0) { if (rr >= BTILE + EXTN) return kb[(long long)relA2 + rr]; }
            return bits_to_key<K>(lds_bits0(rr));
          };
          // ---- the split of the 2-way join (two_layer.rs:130-175) anywhere near: the general form knows Q2-Q4
          if (host_split) {
            if ((unsigned int)(h_split - (A2 - 2u)) <= (term_rel > (unsigned int)(BTILE + EXTN) ? term_rel : (unsigned int)(BTILE + EXTN)) + 2u) break;
          } else {
            if (t2 < mid && tt >= mid) break;                                   // the targets of [A2 - 2, end of the open leaf] cross L / 2
          }
          if (RMI_SC_STOP == 1) { done = true; break; }
          SC_TICK(2);
          // ---- slots (one per lane with a start, in lane order), the y carried into each lane
          const unsigned int q = __builtin_amdgcn_mbcnt_hi((unsigned int)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)hm, 0u));
          const unsigned int nb = (unsigned int)__builtin_popcountll(hm);
          unsigned int y_in = f - 1u, yp = f + p - 1u;                           // (no duplicates: every key is its own first occurrence)
          if (dups) {
            unsigned long long hd = 0ull;
            K kp = bits_to_key<K>(kprev_b);
#pragma unroll
            for (int rr = 0; rr < NSUB; rr++) {
              read_row(rr);
#pragma unroll
              for (int v = 0; v < V; v++) { const K kv = bits_to_key<K>(kk[v]); hd |= !(kv == kp) ? (1ull << (rr * V + v)) : 0ull; kp = kv; }
            }
            const unsigned int lh = hd ? f + (63u - (unsigned int)__builtin_clzll(hd)) + 1u : 0u;   // (index + 1 of this lane's last head)
            const unsigned int lh_prev = sc_dpp<0x138, 0xF>(0u, lh);              // wave_shr:1
            const unsigned int lh_ex = sc_scan_max(lh_prev);
            const unsigned int y_tile = front_y(0, A2, relA2);
            y_in = lh_ex ? lh_ex - 1u : y_tile;
            const unsigned long long hb = hd & ((1ull << (p & 63u)) - 1ull);      // (p < VF <= 64 in the lanes that use it)
            yp = hb ? f + (63u - (unsigned int)__builtin_clzll(hb)) : y_in;
          }
          // ---- F2: the records of the starts; the open leaf's end closes them.  (FB: behind the pending slots)
          const unsigned int pq = pend + q;                                     // this lane's slot (entry pq + 1 of the model tables; entry pq: the leaf in front)
          wave_sync();
          if (has) {
            r_s[pq] = f + p; r_yp[pq] = yp;
            if constexpr (FB) r_t[pq] = t_hi | ((t_hi - tp - 1u) << 29);         // (leaf ids below 2^29: `take`; at most 4 empty leaves in front: above)
            else { r_t[pq] = t_hi; r_g0[pq] = tp + 1u; }
            if (q >= 1u) r_end[pq - 1u] = f + p;
          }
          if (lane == 0) r_end[pend + nb - 1u] = A2 + term_rel;
          wave_sync();
          // ---- F3: the models.  Lane l: slot l.  Container [s - 1, e]: both end points exist (no split, no end of the data nearby) and their keys differ
          // (nothing of a slot stays in registers across the error pass: F5 reads records, model and end keys from LDS again -- registers held
          //  there were spilled, and a spilled register's reload waits for the key loads in flight)
          // (... and whether the clamp of a prediction to n, two_layer.rs:14-18, can bind at all: a leaf's keys lie between its container's end
          //  keys, fma and the saturating conversion are monotone, so both ends predicting below n + 1 means every key does -- the error pass
          //  of a tile whose leaves all pass drops the min, one instruction of seven per key)
          bool cfree = true;
          if ((unsigned int)lane < nb) {
            const unsigned int sl = pend + (unsigned int)lane;
            const unsigned int q_s = r_s[sl], q_e = r_end[sl];
            const K k_lo = bits_to_key<K>(lds_bits0((int)(q_s - 1u - A2))), k_hi = key_rel((int)(q_e - A2));
            const double x0 = KeyTraits<K>::as_float(k_lo), x1 = KeyTraits<K>::as_float(k_hi);
            const double y0f = (double)r_yp[sl], y1f = (double)q_e;
            if constexpr (FBK) { r_klo[sl] = (unsigned int)key_to_bits<K>(k_lo); r_khi[sl] = (unsigned int)key_to_bits<K>(k_hi); }   // (behind the read of r_yp[sl]: the same words)
            const double mb = (y0f - y1f) / (x0 - x1);                           // linear_spline.rs:27
            const double ma = y0f - mb * x0;                                     // :28, plain multiply-subtract
            m_ab[2 * (sl + 1u)] = ma; m_ab[2 * (sl + 1u) + 1] = mb;
            m_err[sl + 1u] = 0u; m_run[sl + 1u] = 0u;
            const double np1 = (double)n32 + 1.0;
            cfree = (__builtin_fma(mb, x0, ma) < np1) && (__builtin_fma(mb, x1, ma) < np1);
          }
          const bool clampfree = RMI_SC_CLAMPFREE && __all(cfree);
          // which positions hold some lane's start (a scalar mask: the error pass tests it, not the lanes)
          unsigned long long am;
          {
            unsigned int a0 = (has && p < 32u) ? (1u << p) : 0u, a1 = (has && p >= 32u) ? (1u << (p - 32u)) : 0u;
            a0 |= sc_dpp<0x111, 0xF>(0u, a0); a0 |= sc_dpp<0x112, 0xF>(0u, a0); a0 |= sc_dpp<0x114, 0xF>(0u, a0); a0 |= sc_dpp<0x118, 0xF>(0u, a0);
            a0 |= sc_dpp<0x142, 0xA>(0u, a0); a0 |= sc_dpp<0x143, 0xC>(0u, a0);
            if constexpr (VF > 32) {
              a1 |= sc_dpp<0x111, 0xF>(0u, a1); a1 |= sc_dpp<0x112, 0xF>(0u, a1); a1 |= sc_dpp<0x114, 0xF>(0u, a1); a1 |= sc_dpp<0x118, 0xF>(0u, a1);
              a1 |= sc_dpp<0x142, 0xA>(0u, a1); a1 |= sc_dpp<0x143, 0xC>(0u, a1);
            }
            am = ((unsigned long long)sc_lane63(a1) << 32) | (unsigned long long)sc_lane63(a0);
          }
          wave_sync();
          if (RMI_SC_STOP == 3) { done = true; break; }
          SC_TICK(3);
          // ---- F4: the error pass.  The lane's keys in front of its start belong to the leaf of slot q - 1 (table entry q; entry 0: a leaf
          //      of an earlier tile, nothing is kept), those from the start on to slot q (entry q + 1)
          {
            double pa = m_ab[2 * pq], pb = m_ab[2 * pq + 1];
            double pa1 = 0.0, pb1 = 0.0;
            if (has) { pa1 = m_ab[2 * (pq + 1u)]; pb1 = m_ab[2 * (pq + 1u) + 1]; }
            unsigned int m = 0u, m0 = 0u, rn = 0u, rn0 = 0u, y_last = 0u;
            unsigned int amlo = (unsigned int)am, amhi = (unsigned int)(am >> 32);
            if (!dups) {
              auto pass = [&](auto cf_tag) {
              constexpr bool CF = decltype(cf_tag)::value;
#pragma unroll
              for (int rr = 0; rr < NSUB; rr++) {
                read_row(rr);
#pragma unroll
                for (int v = 0; v < V; v++) asm volatile("" : "+v"(kk[v]));
#pragma unroll
                for (int v = 0; v < V; v++) {
                  const int gv = rr * V + v;                                     // position among the lane's VF keys
#if RMI_SC_BRANCHFREE
                  {
                    const bool sw = p == (unsigned int)gv;
                    m0 = sw ? m : m0; m = sw ? 0u : m;
                    pa = sw ? pa1 : pa; pb = sw ? pb1 : pb;
                  }
#else
                  unsigned int& amr = gv < 32 ? amlo : amhi;
                  asm volatile("" : "+s"(amr));
                  if (amr & 1u) {                                                // (scalar) some lane's start lies at this position
                    const bool sw = p == (unsigned int)gv;
                    m0 = sw ? m : m0; m = sw ? 0u : m;
                    pa = sw ? pa1 : pa; pb = sw ? pb1 : pb;
                  }
                  amr >>= 1;
#endif
                  const double x = KeyTraits<K>::as_float(bits_to_key<K>(kk[v]));
                  const unsigned int pc = sg_cvt_u32(__builtin_fma(pb, x, pa));             // linear_spline.rs:52, models/mod.rs:735-737
                  const unsigned int pr = CF ? pc : min(pc, n32);                           // two_layer.rs:14-18
                  m = max(m, sg_absdiff(pr, f + (unsigned int)gv));
                }
              }
              };
              if (clampfree) pass(std::true_type{}); else pass(std::false_type{});
            } else {
              unsigned int y = y_in;
              bool ne = false;
#pragma unroll
              for (int rr = 0; rr < NSUB; rr++) {
                read_row(rr);
                const B krow_next = rr + 1 < NSUB ? bits_at(rowp + (rr + 1 < NSUB ? rr + 1 : rr) * S) : knext_b;   // the key behind this row
                if (rr == 0) ne = !(bits_to_key<K>(kk[0]) == bits_to_key<K>(kprev_b));
#pragma unroll
                for (int v = 0; v < V; v++) asm volatile("" : "+v"(kk[v]));
#pragma unroll
                for (int v = 0; v < V; v++) {
                  const int gv = rr * V + v;
#if RMI_SC_BRANCHFREE
                  {
                    const bool sw = p == (unsigned int)gv;
                    m0 = sw ? m : m0; m = sw ? 0u : m;
                    rn0 = sw ? rn : rn0; rn = sw ? 0u : rn;
                    pa = sw ? pa1 : pa; pb = sw ? pb1 : pb;
                  }
#else
                  unsigned int& amr = gv < 32 ? amlo : amhi;
                  asm volatile("" : "+s"(amr));
                  if (amr & 1u) {
                    const bool sw = p == (unsigned int)gv;
                    m0 = sw ? m : m0; m = sw ? 0u : m;
                    rn0 = sw ? rn : rn0; rn = sw ? 0u : rn;
                    pa = sw ? pa1 : pa; pb = sw ? pb1 : pb;
                  }
                  amr >>= 1;
#endif
                  const K kv = bits_to_key<K>(kk[v]);
                  const K kn = bits_to_key<K>(v + 1 < V ? kk[v + 1 < V ? v + 1 : v] : krow_next);
                  const unsigned int i = f + (unsigned int)gv;
                  y = ne ? i : y;
                  const double x = KeyTraits<K>::as_float(kv);
                  const unsigned int pr = min(sg_cvt_u32(__builtin_fma(pb, x, pa)), n32);
                  m = max(m, sg_absdiff(pr, y));
                  // a run of equal keys is recorded when the next different key arrives (lower_bound_correction.rs:108-119)
                  ne = !(kn == kv);
                  rn = max(rn, ne ? i + 1u - y : 0u);
                }
              }
              y_last = y;
            }
            const unsigned int mA = has ? m0 : m, rA = has ? rn0 : rn;
            if (q >= 1u) { if (mA) atomicMax(&m_err[pq], mA); if (rA > 1u) atomicMax(&m_run[pq], rA); }
            if (has) { if (m) atomicMax(&m_err[pq + 1u], m); if (rn > 1u) atomicMax(&m_run[pq + 1u], rn); }
            // ---- the keys of the open leaf behind the big tile: [A2 + BTILE, A2 + term_rel)
            {
              const double ta = m_ab[2 * (pend + nb)], tb = m_ab[2 * (pend + nb) + 1];
              unsigned int em = 0u, rm = 0u;
              unsigned int y_carry = (unsigned int)__builtin_amdgcn_readlane((int)y_last, 63);
              if constexpr (FAR != 2) {
                for (unsigned int o = (unsigned int)BTILE; o < term_rel; o += 64u) {
                  const unsigned int rel = o + (unsigned int)lane;
                  const bool in = rel < term_rel;
                  const int rr = (int)(in ? rel : o);
                  const K kv = key_rel(rr);
                  const unsigned int i = A2 + rel;
                  unsigned int y = i;
                  if (dups) {
                    const K kpv = key_rel(rr - 1), knx = key_rel(rr + 1);
                    const unsigned int hidx = (in && !(kv == kpv)) ? i : 0u;
                    const unsigned int pm = sc_scan_max(hidx);
                    y = max(pm, y_carry);
                    if (in && !(knx == kv)) rm = max(rm, i + 1u - y);
                    y_carry = sc_lane63(y);
                  }
                  if (in) {
                    const double x = KeyTraits<K>::as_float(kv);
                    const unsigned int pr = min(sg_cvt_u32(__builtin_fma(tb, x, ta)), n32);
                    em = max(em, sg_absdiff(pr, y));
                  }
                }
              } else {
                // one block of 64 keys of the open leaf: the lane's key kv at the relative index rel, the keys on either side of it
                auto open_block = [&](unsigned int rel, bool in, bool dq_, K kv, K kpv, K knx) {
                  const unsigned int i = A2 + rel;
                  unsigned int y = i;
                  if (dq_) {
                    const unsigned int hidx = (in && !(kv == kpv)) ? i : 0u;
                    const unsigned int pm = sc_scan_max(hidx);
                    y = max(pm, y_carry);
                    if (in && !(knx == kv)) rm = max(rm, i + 1u - y);
                    y_carry = sc_lane63(y);
                  }
                  if (in) {
                    const double x = KeyTraits<K>::as_float(kv);
                    const unsigned int pr = min(sg_cvt_u32(__builtin_fma(tb, x, ta)), n32);
                    em = max(em, sg_absdiff(pr, y));
                  }
                };
                unsigned int o = (unsigned int)BTILE;
                {
                  // the look-ahead's part (from the tile image)
                  const unsigned int t_la = term_rel < (unsigned int)(BTILE + EXTN) ? term_rel : (unsigned int)(BTILE + EXTN);
                  for (; o < t_la; o += 64u) {
                    const unsigned int rel = o + (unsigned int)lane;
                    const bool in = rel < term_rel;
                    const int rr = (int)(in ? rel : o);
                    const K kv = key_rel(rr);
                    K kpv = kv, knx = kv;
                    if (dups) { kpv = key_rel(rr - 1); knx = key_rel(rr + 1); }
                    open_block(rel, in, dups, kv, kpv, knx);
                  }
                }
                if constexpr (FAR == 2) {
                  // ... and what lies behind it, from the key array: FARB blocks a trip, their loads issued together (a trip is a memory round trip; whether these
                  // keys repeat has not been looked at: the duplicate form of the block, which is the plain form's values where they do not)
                  constexpr unsigned int FARB = RMI_SC_FARB;
                  if (!dups) y_carry = A2 + (unsigned int)(BTILE + EXTN) - 1u;   // (no key of the tile and its look-ahead repeats: the last of them is its own first occurrence)
                  // (a lane's neighbours' keys come from the neighbouring lanes -- three loads a block held 24 registers for 8-byte keys and spilled --, the key in
                  //  front of a block from the block before it, the key behind a trip's last block by one load of a uniform address)
                  auto lane_of = [&](B v, int src) -> B {
                    if constexpr (sizeof(B) == 8) {
                      const unsigned int lo_ = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src), hi_ = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), src);
                      return (B)(((unsigned long long)hi_ << 32) | lo_);
                    } else return (B)__builtin_amdgcn_readlane((int)v, src);
                  };
                  unsigned int relA2f = relA2;
                  asm volatile("" : "+s"(relA2f));                              // (the addresses below formed HERE, not in registers held across the tile's error pass: they were spilled)
                  const K* const pt = kb + (long long)relA2f;
                  B kfront = (B)key_to_bits<K>(pt[o - 1u]);
                  for (; o < term_rel; o += 64u * FARB) {
                    B kv[FARB];
  #pragma unroll
                    for (unsigned int u = 0; u < FARB; u++) {
                      const unsigned int rel = o + 64u * u + (unsigned int)lane;
                      kv[u] = (B)key_to_bits<K>(pt[rel < term_rel ? rel : term_rel]);   // (behind the leaf's end: the key that ends it)
                    }
                    const B kback = (B)key_to_bits<K>(pt[o + 64u * FARB < term_rel ? o + 64u * FARB : term_rel]);
  #pragma unroll
                    for (unsigned int u = 0; u < FARB; u++) {
                      if (o + 64u * u < term_rel) {
                        const unsigned int rel = o + 64u * u + (unsigned int)lane;
                        const B up = __shfl_up(kv[u], 1, 64), dn = __shfl_down(kv[u], 1, 64);
                        const B nxt0 = u + 1u < FARB ? lane_of(kv[u + 1u < FARB ? u + 1u : u], 0) : kback;
                        open_block(rel, rel < term_rel, true, bits_to_key<K>(kv[u]), bits_to_key<K>(lane == 0 ? kfront : up), bits_to_key<K>(lane == 63 ? nxt0 : dn));
                        kfront = lane_of(kv[u], 63);
                      }
                    }
                  }
                }
              }
              if (term_rel > (unsigned int)BTILE) {
                if (em) atomicMax(&m_err[pend + nb], em);
                if (rm > 1u) atomicMax(&m_run[pend + nb], rm);
              }
            }
          }
          wave_sync();
          if (RMI_SC_STOP == 4) { done = true; break; }
          SC_TICK(4);
          // ---- F5: the leaf ends (rmi_scan_ends.inc.h) -- at once, or (FB) at the head of the loop when the next tile's starts might not fit behind the pending ones
          if constexpr (FB) { pend += nb; nbmax = nb > nbmax ? nb : nbmax; }
          else {
            const unsigned int ends_count = nb, ends_A2 = A2;
#include "rmi_scan_ends.inc.h"
          }
          done = true;
          SC_TICK(5);
#if RMI_SC_PROF
          nfast++;
#endif
        } while (false);
        if (!done) leave_tile();
        continue;
      }
    }
    // ================= the general form: the big tile's NSUB tiles one after the other, lane l <-> row 64 h + l =================
    if constexpr (PHASE == 1)
    for (int h = 0; h < NSUB; h++) {
    const unsigned int relA = relA2 + (unsigned int)(h * TILE);                  // relative index of the tile's first key
    const unsigned int A = base32 + relA;                                       // ... and its global index
    unsigned int* const trow = trow0 + h * 64 * S;
    const bool edge = relA < rel_lo + 1u || relA + (unsigned int)TILE + (unsigned int)EXTN + 1u > rel_hi;   // (wave-uniform; implies edge2: the positions outside the launch are filled)
    // key (raw bits) at a global index near the tile: LDS where the big tile, its front chunks or its look-ahead hold it
    auto lds_bits = [&](int rel) -> B { return lds_bits0(rel + h * TILE); };     // rel in [-FHN - h TILE, (NSUB - h) TILE + EXTN)
    auto key_bits = [&](unsigned int i) -> B {
      const int rel = (int)(i - A);
      if (rel >= -FHN - h * TILE && rel < (NSUB - h) * TILE + EXTN) return lds_bits(rel);
      return key_to_bits<K>(keys[i]);                                            // (n < 2^32: i IS the global index)
    };
    auto key_at = [&](unsigned int i) -> K { return bits_to_key<K>(key_bits(i)); };
    // the key at a global index that may lie outside the launch -- a shard's halo: the point in front of its first leaf, the point behind its
    // last -- : the key array's value, never the nearest valid key the LDS image holds at such a position
    auto key_true = [&](unsigned int i) -> K { return ((unsigned int)(i - base32) - rel_lo < n_it) ? key_at(i) : keys[i]; };
    // ---- the lane's keys
    const unsigned int relf = relA + (unsigned int)(lane * V);                  // relative index of the lane's first key
    const unsigned int f = base32 + relf;
    B kk[V];
    {
      const unsigned int* rowp = trow + lane * S;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const uint4 q = *reinterpret_cast<const uint4*>(rowp + 4 * c);
        if constexpr (DW == 1) { kk[4 * c] = q.x; kk[4 * c + 1] = q.y; kk[4 * c + 2] = q.z; kk[4 * c + 3] = q.w; }
        else { kk[2 * c] = ((B)q.y << 32) | q.x; kk[2 * c + 1] = ((B)q.w << 32) | q.z; }
      }
    }
    const B kprev_b = key_bits(f - 1u), knext_b = key_bits(f + (unsigned int)V);
#if RMI_SC_PROF
    unsigned long long gprof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, glast = __builtin_readcyclecounter();
    const unsigned long long gstart = glast;
#endif
    unsigned int force_lo = 0u, force_hi = 0u;                                   // bit of the position it_lo / it_hi in this lane
    if (edge) {
      if (n_it > 0u && rel_lo >= relf && rel_lo - relf < (unsigned int)V) force_lo = 1u << (rel_lo - relf);
      if (rel_hi >= relf && rel_hi - relf < (unsigned int)V) force_hi = 1u << (rel_hi - relf);
    }
    // ---- P1: targets, leaf starts, heads
    unsigned int bnd = 0u, hd = 0u, t_lane_last;
    {
      bool oob = false, oob_any = false;
      unsigned int tp = target_of(bits_to_key<K>(kprev_b), oob);
      K kp = bits_to_key<K>(kprev_b);
#pragma unroll
      for (int v = 0; v < V; v++) {
        const K kv = bits_to_key<K>(kk[v]);
        const unsigned int t = target_of(kv, oob);
        if constexpr (!root_needs_bounds_check<ROOT>()) oob_any = oob_any || oob;
        bnd |= (t != tp) ? (1u << v) : 0u;
        hd |= !(kv == kp) ? (1u << v) : 0u;
        tp = t; kp = kv;
      }
      t_lane_last = tp;
      if constexpr (!root_needs_bounds_check<ROOT>()) { if (oob_any && n_it > 0u) flags |= EF_ROOT_OOB; }   // two_layer.rs:45-48
    }
    if (edge) {
      if (n_it == 0u) { bnd = 0u; hd = 0u; }
      bnd = (bnd & ~force_lo) | force_lo | force_hi;                             // (a start at it_lo by decree; `bnd` cannot hold one at it_hi or among the filled positions)
      hd |= force_lo;
    }
    // ---- slot numbers and the y carried into each lane
    const unsigned int nbl = (unsigned int)__builtin_popcount(bnd);
    const unsigned int cnt_incl = sc_scan_add(nbl);
    const unsigned int cnt_excl = cnt_incl - nbl;
    const unsigned int nb = sc_lane63(cnt_incl);                                 // leaf starts in this tile (the position it_hi counts as one)
    const unsigned int t_tile_last = sc_lane63(t_lane_last);
    const bool virt_here = edge && rel_hi >= relA && rel_hi - relA < (unsigned int)TILE;   // the position it_hi lies in this tile: its last "start" is the end of the data
    // the last key's leaf (Q7's owner), for the record
    // (only the launch that holds the data's last key, as in k_leaf_search: a shard whose it_hi < n falls exactly on this tile's end has
    //  rel_hi - relA == TILE too, and t_tile_last is then the target of ITS last key -- a leaf it owns, which would get the extra count)
    if (lane == 63 && edge && n_it > 0u && sp.it_hi == sp.n && rel_hi > relA && rel_hi - relA <= (unsigned int)TILE) st->last_target = (unsigned long long)t_tile_last;
    if (nb == 0u) continue;                                                      // the tile lies inside one leaf that started earlier
    if (RMI_SC_STOP == 1) continue;
    // duplicates: a valid key that equals the key before it
    const unsigned int vfull = sc_mask_below((unsigned int)V);
    const bool dups = __any(((~hd) & vfull) != 0u) != 0;
    unsigned int y_in = 0u;                                                      // y (FixDups offset) of the key in front of this lane's first key
    {
      const unsigned int lh = hd ? f + (31u - (unsigned int)__builtin_clz(hd)) + 1u : 0u;   // (index + 1 of this lane's last head)
      const unsigned int lh_prev = sc_dpp<0x138, 0xF>(0u, lh);                    // wave_shr:1
      const unsigned int lh_ex = sc_scan_max(lh_prev);
      const unsigned int y_tile = front_y(h * TILE, A, relA);
      y_in = lh_ex ? lh_ex - 1u : y_tile;
    }
    const unsigned int anyb = sc_wave_or(bnd);
    if (RMI_SC_DEBUG && bnd) printf("tile %u lane %d relf %u bnd %08x hd %08x cnt_excl %u nb %u y_in %u edge %d virt %d flo %x fhi %x\n", tile, lane, relf, bnd, hd, cnt_excl, nb, y_in, (int)edge, (int)virt_here, force_lo, force_hi);
    // ---- the split of the 2-way join (two_layer.rs:130-175): given by the host (a shard), else found where it is crossed
    unsigned int w_split = n32, w_stgt = 0u;                                     // the values leaf_container gets for this tile's leaves
    if (host_split) { w_split = (unsigned int)st->split_idx; w_stgt = (unsigned int)st->split_target; }

    SC_GT(0);
    // ---- batches of SC_SLOTS leaf starts
    for (unsigned int qb = 0; qb < nb; qb += (unsigned int)SC_SLOTS) {
      const unsigned int cnt_b = min((unsigned int)SC_SLOTS, nb - qb);          // records of this batch
      const bool last_batch = qb + cnt_b == nb;
      wave_sync();
      // ---- P2: records.  A lane walks its set bits; everything a start needs is recomputed from LDS (no register is indexed).
      {
        unsigned int m = bnd;
        while (m) {
          const unsigned int v = (unsigned int)__builtin_ctz(m);
          m &= m - 1u;
          const unsigned int q = cnt_excl + (unsigned int)__builtin_popcount(bnd & ((1u << v) - 1u));
          if (q - qb > (unsigned int)SC_SLOTS) continue;                        // (q - qb in [0, SLOTS]: the batch and the start behind it)
          const unsigned int i = f + v;
          const unsigned int hb = hd & ((1u << v) - 1u);
          unsigned int yp = hb ? f + (31u - (unsigned int)__builtin_clz(hb)) : y_in;   // y of key i - 1
          unsigned int t, g0;
          const unsigned int reli = relf + v;
          const bool owned = q - qb < (unsigned int)SC_SLOTS;                    // (not the batch's closing record: the next batch owns that start)
          auto note_split = [&](unsigned int tp) {                               // two_layer.rs:132-136, 152-156
            if (!host_split && tp < mid && t >= mid) {
              w_split = i; w_stgt = t;
              if (owned) {
                st->split_idx = (unsigned long long)i; st->split_target = (unsigned long long)t;
                if (i + 1u >= n32) flags |= EF_DEGENERATE_SPLIT;                 // second half empty -> two_layer.rs:27
              }
            }
          };
          if (edge && reli == rel_hi) {                                          // behind the last key: the end of the data, not a leaf
            bool oob;
            t = (unsigned int)sp.leaf_hi;
            g0 = n_it > 0u ? target_of(key_at(i - 1u), oob) + 1u : (unsigned int)sp.leaf_lo;
          } else {
            bool oob;
            t = target_of(key_at(i), oob);
            if (edge && reli == rel_lo) {
              g0 = (unsigned int)sp.leaf_lo;                                     // the launch's first key: the empty leaves from leaf_lo on are its gap
              yp = 0u;
              if (i == 0u) { if (t >= mid) flags |= EF_DEGENERATE_SPLIT; }       // split_idx == 0 -> two_layer.rs:27
              else if ((long long)reli - 1 >= rd_lo_rel) {
                // a shard's first key: the key in front of it is the previous shard's last -- the prev-last point of the first leaf
                const unsigned int tp = target_of(keys[i - 1u], oob);            // (from the key array: LDS holds the launch's first key there)
                if (t < tp) flags |= EF_NON_MONOTONE;
                yp = (unsigned int)first_occurrence(keys, (uint64_t)(tile0 + (long long)reli - 1), sp.rd_lo);
                note_split(tp);
              }
            } else {
              const unsigned int tp = target_of(key_at(i - 1u), oob);
              if (t < tp) flags |= EF_NON_MONOTONE;                              // two_layer.rs:50 / :144
              g0 = tp + 1u;
              note_split(tp);
            }
          }
          r_s[q - qb] = i; r_t[q - qb] = t; r_g0[q - qb] = g0; r_yp[q - qb] = yp;
        }
      }
      SC_GT(1);
      // the split, if a lane of this wave saw it (at most one key in the whole data set crosses L / 2)
      if (!host_split) {
        const unsigned long long sm = __ballot(w_split != n32);
        if (sm) {
          const int src = __builtin_ctzll(sm);
          w_split = (unsigned int)__builtin_amdgcn_readlane((int)w_split, src);
          w_stgt = (unsigned int)__builtin_amdgcn_readlane((int)w_stgt, src);
        } else if (relA >= rel_lo + 1u) {
          // a leaf start right in front of the tile (key A - 1) is not in `bnd`, but the leaf behind it has lost its prev point if that
          // key is the split (Q2/Q3)
          bool oob;
          const unsigned int t1 = target_of(key_at(A - 1u), oob);
          if (t1 >= mid && relA >= rel_lo + 2u) {
            const unsigned int t2 = target_of(key_at(A - 2u), oob);
            if (t2 < mid) { w_split = A - 1u; w_stgt = t1; }
          }
        }
      }
      wave_sync();
      // ---- the end of the leaf that is open at the tile's end
      unsigned int term_s = 0u;                                                  // start of the leaf behind the tile's last owned leaf
      bool handed = false;
      if (last_batch && !virt_here) {
        bool found = false;
        const unsigned int lim = long_min > (unsigned int)EXTN ? long_min : (unsigned int)EXTN;
        for (unsigned int o = 0; o < lim && !found; o += 64u) {
          const unsigned int rel = relA + (unsigned int)TILE + o + (unsigned int)lane;
          const bool inb = rel < rel_hi;
          bool oob, diff = false;
          unsigned int t = 0u;
          if (inb) { t = target_of(key_at(base32 + rel), oob); diff = t != t_tile_last; }
          const unsigned long long dm = __ballot(diff || !inb);
          if (dm) {
            const int src = __builtin_ctzll(dm);
            const unsigned int relt = relA + (unsigned int)TILE + o + (unsigned int)src;
            term_s = base32 + (relt < rel_hi ? relt : rel_hi);
            found = true;
            const unsigned int tt = (unsigned int)__builtin_amdgcn_readlane((int)t, src);
            if (relt < rel_hi) {
              if (!host_split && t_tile_last < mid && tt >= mid) { w_split = term_s; w_stgt = tt; }   // (the owner of that start records it)
            }
          }
        }
        if (!found) handed = true;
        if (lane == 0) r_s[cnt_b] = term_s;
      }
      wave_sync();
      if (RMI_SC_STOP == 2) continue;
      SC_GT(2);
      // ---- P3: the models.  Lane q: slot qb + q.
      const unsigned int own_b = last_batch ? (cnt_b - (virt_here ? 1u : 0u)) : cnt_b;     // real leaves among this batch's records
      const bool hand_q = handed && last_batch && (unsigned int)lane == own_b - 1u;
      unsigned int q_s = 0u, q_e = 0u, q_t = 0u;
      if (RMI_SC_DEBUG && lane == 0) printf("tile %u batch %u cnt_b %u own_b %u term_s %u handed %d wsplit %u\n", tile, qb, cnt_b, own_b, term_s, (int)handed, w_split);
      if ((unsigned int)lane < own_b) {
        q_s = r_s[lane]; q_e = r_s[lane + 1]; q_t = r_t[lane];
        if (RMI_SC_DEBUG) printf("  slot %d s %u e %u t %u g0 %u yp %u\n", lane, q_s, q_e, q_t, r_g0[lane], r_yp[lane]);
        double pa = 0.0, pb = 0.0;
        if (!hand_q) {
          uint64_t lo, hi;
          const uint64_t spl = (uint64_t)w_split;
          const int ck = leaf_container((uint64_t)q_t, (uint64_t)q_s, (uint64_t)q_e, sp.n, spl, (uint64_t)w_stgt, lo, hi);
          if (ck == 2) {
            // FixDups offsets of the two end points: the key in front of a leaf start carries the record's y; a leaf's first key and
            // the next leaf's first key are their own first occurrences
            const unsigned int lo32 = (unsigned int)lo, hi32 = (unsigned int)hi;
            const K k0 = key_true(lo32), k1 = key_true(hi32);
            unsigned int y0;
            if (lo32 + 1u == q_s) y0 = r_yp[lane];
            else if (lo32 == q_s) y0 = q_s;
            else y0 = (key_at(q_s + 1u) == key_at(q_s)) ? q_s : q_s + 1u;          // Q2: the container starts at the key behind the split key
            if (lo32 == hi32 || k0 == k1) { pa = (double)y0; pb = 0.0; }          // linear_spline.rs:18-20
            else {
              unsigned int y1;
              if (hi32 == q_e) y1 = q_e;
              else y1 = (unsigned int)first_occurrence(keys, (uint64_t)(tile0 + (long long)(unsigned int)(hi32 - base32)), sp.rd_lo);   // Q3: no next point (the split, the end of the data)
              const double x0 = KeyTraits<K>::as_float(k0), x1 = KeyTraits<K>::as_float(k1);
              const double y0f = (double)y0, y1f = (double)y1;
              pb = (y0f - y1f) / (x0 - x1);                                       // linear_spline.rs:27
              pa = y0f - pb * x0;                                                 // :28, plain multiply-subtract
            }
          } else if (ck == 1) { pa = (double)lo; pb = 0.0; }                      // Q4
        }
        m_ab[2 * (lane + 1)] = pa; m_ab[2 * (lane + 1) + 1] = pb;
        m_err[lane + 1] = 0u; m_run[lane + 1] = 0u;
      }
      wave_sync();
      if (RMI_SC_STOP == 3) continue;
      SC_GT(3);
      // ---- P4: the error pass over the lane's keys
      {
        // slot of the key in front of the lane, relative to the batch, + 1 (entry 0 and SLOTS + 1 of the tables are padding)
        auto rel_slot = [&](unsigned int q_abs_plus1) -> unsigned int {          // q_abs_plus1 = absolute slot + 1 (0: in front of the tile's first start)
          const unsigned int d = q_abs_plus1 - qb;                               // (wraps for slots in front of the batch)
          return d <= (unsigned int)SC_SLOTS ? d : (unsigned int)SC_SLOTS + 1u;
        };
        unsigned int slot1 = cnt_excl;                                           // absolute slot + 1 of the running leaf
        unsigned int rs = rel_slot(slot1);
        const unsigned int flush_n = own_b - ((handed && last_batch) ? 1u : 0u);   // slots 1 .. flush_n of the tables take maxima
        auto flush = [&](unsigned int rsl, unsigned int m, unsigned int rn) {
          if (rsl - 1u < flush_n) {
            if (m) atomicMax(&m_err[rsl], m);
            if (rn > 1u) atomicMax(&m_run[rsl], rn);
          }
        };
        unsigned int m = 0u, rn = 0u;
        double pa = m_ab[2 * rs], pb = m_ab[2 * rs + 1];
#pragma unroll
        for (int v = 0; v < V; v++) asm volatile("" : "+v"(kk[v]));              // (the conversions are redone here: 2 V registers are not worth keeping)
        // (the masks are shifted through registers the compiler cannot see through: it would otherwise form all V lane masks
        //  `bnd & (1 << v)` up front -- 2 V scalar registers, spilled -- and read them back lane by lane at every key)
        unsigned int bm = bnd, am = anyb;
        if (!dups) {
#pragma unroll
          for (int v = 0; v < V; v++) {
            asm volatile("" : "+s"(am));
            if (am & 1u) {
              asm volatile("" : "+v"(bm));
              if ((bm >> v) & 1u) {
                flush(rs, m, 0u);
                slot1++; rs = rel_slot(slot1); m = 0u;
                pa = m_ab[2 * rs]; pb = m_ab[2 * rs + 1];
              }
            }
            am >>= 1;
            const double x = KeyTraits<K>::as_float(bits_to_key<K>(kk[v]));
            const unsigned int pr = min(sg_cvt_u32(__builtin_fma(pb, x, pa)), n32);   // linear_spline.rs:52, models/mod.rs:735-737, two_layer.rs:14-18
            m = max(m, sg_absdiff(pr, f + (unsigned int)v));
          }
          flush(rs, m, 0u);
        } else {
          unsigned int y = y_in, hm = hd;
#pragma unroll
          for (int v = 0; v < V; v++) {
            asm volatile("" : "+s"(am));
            if (am & 1u) {
              asm volatile("" : "+v"(bm));
              if ((bm >> v) & 1u) {
                flush(rs, m, rn);
                slot1++; rs = rel_slot(slot1); m = 0u; rn = 0u;
                pa = m_ab[2 * rs]; pb = m_ab[2 * rs + 1];
              }
            }
            am >>= 1;
            const K kv = bits_to_key<K>(kk[v]);
            const K kn = bits_to_key<K>(v + 1 < V ? kk[v + 1 < V ? v + 1 : v] : knext_b);
            const unsigned int i = f + (unsigned int)v;
            asm volatile("" : "+v"(hm));
            y = ((hm >> v) & 1u) ? i : y;
            const double x = KeyTraits<K>::as_float(kv);
            const unsigned int pr = min(sg_cvt_u32(__builtin_fma(pb, x, pa)), n32);
            m = max(m, sg_absdiff(pr, y));
            // a run of equal keys is recorded when the next different key arrives (lower_bound_correction.rs:108-119); the
            // globally last run never is (Q5: knext of the last key is the key itself)
            if (!(kn == kv)) rn = max(rn, i + 1u - y);
          }
          flush(rs, m, rn);
        }
      }
      // ---- the keys of the open leaf behind the tile
      if (last_batch && !virt_here && !handed && own_b >= 1u) {
        const double pa = m_ab[2 * own_b], pb = m_ab[2 * own_b + 1];
        const unsigned int b0 = A + (unsigned int)TILE;
        unsigned int em = 0u, rm = 0u;
        for (unsigned int i0 = b0; i0 != term_s && (unsigned int)(term_s - i0) < 0x80000000u; i0 += 64u) {
          const unsigned int i = i0 + (unsigned int)lane;
          const bool in = (unsigned int)(term_s - i) - 1u < 0x7FFFFFFFu;        // i < term_s
          if (in) {
            const K kv = key_at(i);
            unsigned int y = i;
            {
              const K kpv = key_at(i - 1u);
              if (kpv == kv) y = (unsigned int)first_occurrence(keys, (uint64_t)(tile0 + (long long)(unsigned int)(i - base32)), sp.rd_lo);
              if (i + 1u < n32) { const K kn = key_true(i + 1u); if (!(kn == kv)) rm = max(rm, i + 1u - y); }
            }
            const double x = KeyTraits<K>::as_float(kv);
            const unsigned int pr = min(sg_cvt_u32(__builtin_fma(pb, x, pa)), n32);
            em = max(em, sg_absdiff(pr, y));
          }
        }
        em = sc_wave_max(em); rm = sc_wave_max(rm);
        if (lane == 0) { if (em) atomicMax(&m_err[own_b], em); if (rm > 1u) atomicMax(&m_run[own_b], rm); }
      }
      wave_sync();
      if (RMI_SC_STOP == 4) continue;
      SC_GT(4);
      // ---- P5: the leaf ends.  Lane q: slot qb + q; then the empty leaves in front of the batch's starts.
      const sc_kargp cp = cold();
      const ScanOut out = cold_out(cp);
      const int npeers = SC_ARG(cp, int, peers.n);
      auto store_leaf = [&](uint64_t j, uint64_t s, double a, double b2, uint64_t final_err, uint64_t cnt_j) {
        out.leaf_start[j] = s;
        if (out.params) { out.params[2 * j] = a; out.params[2 * j + 1] = b2; }
        if (out.leaf_err) out.leaf_err[j] = final_err;
        if (out.leaf_count) out.leaf_count[j] = cnt_j;
        double* rp = reinterpret_cast<double*>(out.rows + j * 24);
        rp[0] = a; rp[1] = b2;
        *reinterpret_cast<unsigned long long*>(out.rows + j * 24 + 16) = final_err;
        for (int p = 0; p < npeers; p++) {                                       // (wave-uniform trip count)
          unsigned char* const pt = cold_peer(cp, p);
          double* pr = reinterpret_cast<double*>(pt + j * 24);
          pr[0] = a; pr[1] = b2;
          *reinterpret_cast<unsigned long long*>(pt + j * 24 + 16) = final_err;
        }
      };
      if ((unsigned int)lane < own_b) {
        const uint64_t j = q_t, s = q_s, e = q_e;
        if (hand_q) {
          out.leaf_start[j] = s;
          { SgList fl; fl.ids = SC_ARG(cp, unsigned int*, fl.ids); fl.cnt = SC_ARG(cp, unsigned long long*, fl.cnt); fl.cap = SC_ARG(cp, unsigned long long, fl.cap); fl.push((unsigned int)j); }
        } else {
          double pp[2] = {m_ab[2 * (lane + 1)], m_ab[2 * (lane + 1) + 1]};
          const unsigned int ru = m_run[lane + 1];
          const K k_next = e < sp.n ? key_true((unsigned int)e) : KeyTraits<K>::max_value();
          const K k_prev = s > 0 ? key_true((unsigned int)s - 1u) : KeyTraits<K>::zero_value();
          uint64_t final_err, cnt_j;
          finalize_one_pre<K_LINEAR, K>(j, s, e, sp, r.L, keys, pp, (uint64_t)m_err[lane + 1], ru > 1u ? (uint64_t)ru : 0ull,
                                        e == sp.n ? j : ~0ull, k_next, k_prev, final_err, cnt_j);
          store_leaf(j, s, pp[0], pp[1], final_err, cnt_j);
          agg.add(j, final_err, cnt_j, nf);
        }
      }
      SC_GT(6);
      // empty leaves [g0, t) in front of a start at index s: s == e, the constant model (two_layer.rs:185-197; the last leaf of
      // all keeps the empty model, Q6), no key is read
      auto empty_leaf = [&](uint64_t j, uint64_t s) {
        double pp[2] = {0.0, 0.0};
        uint64_t final_err, cnt_j;
        finalize_one_pre<K_LINEAR, K>(j, s, s, sp, r.L, keys, pp, 0ull, 0ull, ~0ull, KeyTraits<K>::zero_value(), KeyTraits<K>::zero_value(), final_err, cnt_j);
        store_leaf(j, s, pp[0], pp[1], final_err, cnt_j);
        agg.add(j, final_err, cnt_j, nf);
      };
      {
        unsigned int g0 = 0u, gt = 0u, gs = 0u;
        if ((unsigned int)lane < cnt_b) { g0 = r_g0[lane]; gt = r_t[lane]; gs = r_s[lane]; }
        unsigned int glen = gt > g0 ? gt - g0 : 0u;
        if (glen >= SCAN_GAP_MIN) {                                              // a long stretch: k_scan_gaps writes it (if the list has room)
          const unsigned long long pos = atomicAdd(SC_ARG(cp, unsigned long long*, gap_cnt), 1ull);
          if (pos < (unsigned long long)SCAN_GAP_CAP) { SC_ARG(cp, GapRec*, gaps)[pos] = GapRec{g0, gt, gs, 0u}; glen = 0u; }
        }
#if RMI_SC_PROF
        if (edge && (unsigned int)lane < cnt_b) printf("  tile %u lane %d: g0 %u gt %u gs %u own_b %u cnt_b %u\n", tile, lane, g0, gt, gs, own_b, cnt_b);
#endif
        if (glen > 0u && glen <= 4u) for (unsigned int j = g0; j < gt; j++) empty_leaf((uint64_t)j, (uint64_t)gs);
        unsigned long long big = __ballot(glen > 4u);
        while (big) {
          const int src = __builtin_ctzll(big);
          big &= big - 1ull;
          const unsigned int b0 = (unsigned int)__builtin_amdgcn_readlane((int)g0, src), b1 = (unsigned int)__builtin_amdgcn_readlane((int)gt, src);
          const unsigned int bs = (unsigned int)__builtin_amdgcn_readlane((int)gs, src);
          for (unsigned int j = b0 + (unsigned int)lane; j < b1; j += 64u) empty_leaf((uint64_t)j, (uint64_t)bs);
        }
      }
      SC_GT(5);
    }
#if RMI_SC_PROF
    if (lane == 0 && (unsigned long long)__builtin_readcyclecounter() - gstart > 100000ull)
      printf("general tile %u (relA %u) edge %d nb %u: total %llu | P1 %llu P2 %llu ext %llu P3 %llu P4 %llu P5a %llu P5b %llu\n", tile, relA, (int)edge, nb,
             (unsigned long long)__builtin_readcyclecounter() - gstart, gprof[0], gprof[1], gprof[2], gprof[3], gprof[4], gprof[6], gprof[5]);
#endif
    }  // h
  }
#if RMI_SC_PROF
  if (lane == 0 && blockIdx.x % 61u == 0u)
    printf("wave %u xcd %u: tiles %u fast %u total %llu | stage %llu prefetch %llu F1 %llu F2-3 %llu F4 %llu F5 %llu looptop %llu\n", blockIdx.x, blockIdx.x & 7u, ntile, nfast,
           (unsigned long long)__builtin_readcyclecounter() - tstart, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[7]);
#endif
  if constexpr (FB) {
    if (pend) {
      const unsigned int ends_count = pend, ends_A2 = 0u;
#include "rmi_scan_ends.inc.h"
    }
  }
  if (flags) atomicOr(&st->err_flags, flags);
  if constexpr (PHASE == 1) { if (blockIdx.x == 0 && lane == 0) st->scan_listed = tlist != nullptr ? *SC_ARG(kp, unsigned long long*, tile_cnt) : 0ull; }
  // ---- this wave's aggregate record
  {
    unsigned long long mx = agg.mx, mi = agg.mi, sm = agg.sum;
    double l2 = agg.l2, lg = agg.lg;
    if constexpr (PHASE == 0) { mx = amx; mi = ami; sm = asum; l2 = aggp[0]; lg = aggp[1]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long omx = shfl_down_u64(mx, d), omi = shfl_down_u64(mi, d);
      if (omx > mx || (omx == mx && omi > mi)) { mx = omx; mi = omi; }
      sm += shfl_down_u64(sm, d);
      l2 += __shfl_down(l2, d);
      lg += __shfl_down(lg, d);
    }
    if (lane == 0) SC_ARG(cold(), StatsPartial*, out.partials)[blockIdx.x] = StatsPartial{mx, mi, sm, l2, lg};
  }
}

// k_scan_gaps: the stretches of empty leaves k_spline_scan has listed (GapRec).  Thread t of the grid takes the leaves g0 + t, g0 + t + T, ...
// of every record: empty model, constant next_index (two_layer.rs:185-197; the last leaf of all keeps the empty model, Q6), widening
// (:226-259) -- the same finalize_one_pre call as k_spline_scan's own empty leaves.  An empty leaf counts no key: its only share of the
// aggregates (:267-287) is the maximum and its leaf; block b leaves that in partials[b].
template <typename K>
__global__ void __launch_bounds__(256) k_scan_gaps(const K* __restrict__ keys, Span sp, uint64_t L, ScanOut out, PeerRows peers,
                                                   const GapRec* __restrict__ gaps, const unsigned long long* __restrict__ gap_cnt, StatsPartial* __restrict__ partials) {
  __shared__ unsigned long long s_mx[4], s_mi[4];
  const unsigned long long cnt_raw = *gap_cnt;
  const unsigned int cnt = cnt_raw < (unsigned long long)SCAN_GAP_CAP ? (unsigned int)cnt_raw : SCAN_GAP_CAP;
  const unsigned int T = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mx = 0ull, mi = 0ull;
  for (unsigned int rix = 0; rix < cnt; rix++) {
    const GapRec g = gaps[rix];
    for (unsigned int j = g.g0 + t0; j < g.gt; j += T) {
      double pp[2] = {0.0, 0.0};
      uint64_t final_err, cnt_j;
      finalize_one_pre<K_LINEAR, K>((uint64_t)j, (uint64_t)g.gs, (uint64_t)g.gs, sp, L, keys, pp, 0ull, 0ull, ~0ull, KeyTraits<K>::zero_value(), KeyTraits<K>::zero_value(), final_err, cnt_j);
      out.leaf_start[j] = (unsigned long long)g.gs;
      if (out.params) { out.params[2 * (size_t)j] = pp[0]; out.params[2 * (size_t)j + 1] = pp[1]; }
      if (out.leaf_err) out.leaf_err[j] = final_err;
      if (out.leaf_count) out.leaf_count[j] = cnt_j;
      double* rp = reinterpret_cast<double*>(out.rows + (size_t)j * 24);
      rp[0] = pp[0]; rp[1] = pp[1];
      *reinterpret_cast<unsigned long long*>(out.rows + (size_t)j * 24 + 16) = final_err;
      for (int p = 0; p < peers.n; p++) {
        double* pr = reinterpret_cast<double*>(peers.tab[p] + (size_t)j * 24);
        pr[0] = pp[0]; pr[1] = pp[1];
        *reinterpret_cast<unsigned long long*>(peers.tab[p] + (size_t)j * 24 + 16) = final_err;
      }
      if (final_err > mx || (final_err == mx && (unsigned long long)j > mi)) { mx = final_err; mi = (unsigned long long)j; }   // max_by_key: the LAST maximum
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long omx = shfl_down_u64(mx, d), omi = shfl_down_u64(mi, d);
    if (omx > mx || (omx == mx && omi > mi)) { mx = omx; mi = omi; }
  }
  if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_mi[threadIdx.x >> 6] = mi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) if (s_mx[w] > mx || (s_mx[w] == mx && s_mi[w] > mi)) { mx = s_mx[w]; mi = s_mi[w]; }
    partials[blockIdx.x] = StatsPartial{mx, mi, 0ull, 0.0, 0.0};
  }
}

}  // namespace rmi
