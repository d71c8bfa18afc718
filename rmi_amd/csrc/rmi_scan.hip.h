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
#include "rmi_scan_launch.h"

namespace rmi {

template <typename K, int V> struct ScGeom {
  static constexpr int DW = (int)sizeof(K) / 4;          // dwords per key
  static constexpr int KPC = 16 / (int)sizeof(K);        // keys per 16-byte chunk
  static constexpr int TILE = 64 * V;                    // keys per tile
  static constexpr int ROWD = V * DW;                    // dwords of a lane's keys
  static constexpr int S = ROWD + 4;                     // padded row stride in dwords: 4 x odd
  static constexpr int NCH = ROWD / 4;                   // 16-byte chunks per lane and tile
  static constexpr int EXTC = 32;                        // chunks of look-ahead behind the tile (lanes 1..32 of the aux load)
  static constexpr int EXTN = EXTC * KPC;                // ... in keys: 128 (4-byte keys), 64 (8-byte keys)
  static constexpr int LDS_DW = 4 + 64 * S + EXTN * DW;  // [left halo chunk][tile rows][look-ahead]
  static_assert(V <= 32 && (V & (V - 1)) == 0, "a bit per key in a 32-bit mask");
  static_assert(ROWD % 8 == 0, "row stride 4 x odd");
};
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

// (the builtin exists in the device pass only; the host pass merely parses the kernel)
__device__ __forceinline__ const void* sc_kernarg_ptr() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const void*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  return nullptr;
#endif
}

// The kernel's arguments: ONE plain struct.  The kernel never names its parameter: it reads the fields through the kernarg segment
// pointer, the hot ones once into registers, the cold ones (output pointers, peers' tables, the list) where they are used, behind a
// compiler barrier on the pointer -- named parameters are all loaded at the kernel's entry and then live (= spilled: 100+ SGPRs)
// for the whole kernel.
struct ScanArgs {
  const void* keys;               // pre-offset: keys[global index]
  long long tile0;                // global index of the first tile's first position (<= it_lo, a 128-byte line of the key array)
  unsigned int ntiles, tiles_per_xcd;
  unsigned int long_min;
  int host_split;
  DevState* st;
  Span sp;
  RootP r;
  // cold
  ScanOut out;
  SgList fl;
  PeerRows peers;
};

template <int ROOT, typename K, int V>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_spline_scan(ScanArgs) {
  const ScanArgs* const ka = reinterpret_cast<const ScanArgs*>(sc_kernarg_ptr());
  const K* const keys = (const K*)ka->keys;
  const long long tile0 = ka->tile0;
  const unsigned int ntiles = ka->ntiles, tiles_per_xcd = ka->tiles_per_xcd, long_min = ka->long_min;
  const int host_split = ka->host_split;
  DevState* const st = ka->st;
  const Span sp = ka->sp;
  const RootP r = ka->r;
  // the cold arguments, re-read where they are used
  auto cold = [&]() -> const ScanArgs* { const ScanArgs* p = ka; asm volatile("" : "+s"(p)); return p; };
  using G = ScGeom<K, V>;
  using B = typename LnBits<K>::type;
  constexpr int DW = G::DW, KPC = G::KPC, TILE = G::TILE, S = G::S, NCH = G::NCH, EXTN = G::EXTN;
  __shared__ __attribute__((aligned(16))) unsigned int lds[G::LDS_DW];
  __shared__ unsigned int r_s[SC_SLOTS + 1], r_t[SC_SLOTS + 1], r_g0[SC_SLOTS + 1], r_yp[SC_SLOTS + 1];   // boundary records of a batch
  __shared__ __attribute__((aligned(16))) double m_ab[2 * (SC_SLOTS + 2)];                                   // (alpha, beta) per slot, one entry of padding either side
  __shared__ unsigned int m_err[SC_SLOTS + 2], m_run[SC_SLOTS + 2];
  unsigned int* const hl = lds;                       // the 16 bytes in front of the tile
  unsigned int* const trow = lds + 4;                 // lane l's keys at trow[l S ...]
  unsigned int* const ext = lds + 4 + 64 * S;         // the keys behind the tile

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
  auto target_of = [&](K k, bool& oob) -> unsigned int { return s2_target<ROOT, K>(r, Lm1f, Lm1, k, KeyTraits<K>::as_float(k), oob); };

  // ---- tile loads: NCH chunks per lane (chunk c 64 + lane of the tile) + one aux chunk (lane 0: the chunk in front of the
  //      tile; lanes 1..EXTC: the look-ahead behind it).  A chunk is loaded iff it overlaps the readable keys [rd_lo, rd_hi).
  uint4 pf[NCH], pfx;
  auto chunk_load = [&](long long rel_first) -> uint4 {                         // rel_first: relative index of the chunk's first key
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (rel_first + KPC > rd_lo_rel && rel_first < rd_hi_rel) {
      typedef unsigned int raw_t __attribute__((ext_vector_type(4)));
      const raw_t rw = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(kb + rel_first));
      v = make_uint4(rw.x, rw.y, rw.z, rw.w);
    }
    return v;
  };
  auto load_tile = [&](unsigned int tile) {
    const long long a = (long long)tile * TILE;
#pragma unroll
    for (int c = 0; c < NCH; c++) pf[c] = chunk_load(a + (long long)(c * 64 + lane) * KPC);
    const long long ax = lane == 0 ? a - KPC : a + TILE + (long long)(lane - 1) * KPC;
    pfx = make_uint4(0u, 0u, 0u, 0u);
    if (lane <= G::EXTC) pfx = chunk_load(ax);
  };
  // persistent waves; block b runs on XCD b % 8 (observed, for speed only): each XCD streams a contiguous range of tiles, so
  // that the look-ahead of a tile is the neighbouring wave's tile in the same L2
  const unsigned int xcd = blockIdx.x & 7u, wix = blockIdx.x >> 3, wpx = (gridDim.x + 7u - xcd) >> 3;   // this wave's index among the wpx waves of its XCD
  const unsigned int t_lo = xcd * tiles_per_xcd, t_hi = min(ntiles, t_lo + tiles_per_xcd);
  ScAgg agg{0ull, 0ull, 0ull, 0.0, 0.0};
  unsigned int tile = t_lo + wix;
  if (tile < t_hi) load_tile(tile);

  for (; tile < t_hi; tile += wpx) {
    const unsigned int relA = tile * (unsigned int)TILE;                        // relative index of the tile's first key
    const unsigned int A = base32 + relA;                                       // ... and its global index
    // ---- validity.  The tiles at the two ends of the launch hold positions outside [it_lo, it_hi): those take the value of the
    //      nearest valid key BEFORE the tile goes to LDS (no leaf start, no new key value arises among them, and the key behind the
    //      last key equals it: no run is recorded there, Q5); the launch's first key and the position behind its last key are leaf
    //      starts by decree (below).
    const bool edge = relA < rel_lo + 1u || relA + (unsigned int)TILE + (unsigned int)EXTN + 1u > rel_hi;   // (wave-uniform)
    if (edge && n_it > 0u) {
      const B k_first = (B)key_to_bits<K>(kb[rel_lo]), k_last = (B)key_to_bits<K>(kb[rel_hi - 1u]);
      auto patch = [&](uint4& q, long long rel_first) {
        unsigned int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < KPC; k++) {
          const long long rel = rel_first + k;
          if (rel < (long long)rel_lo || rel >= (long long)rel_hi) {
            const B kv = rel < (long long)rel_lo ? k_first : k_last;
            if constexpr (DW == 1) w[k] = (unsigned int)kv;
            else { w[2 * k] = (unsigned int)kv; w[2 * k + 1] = (unsigned int)((unsigned long long)kv >> 32); }
          }
        }
        q = make_uint4(w[0], w[1], w[2], w[3]);
      };
      const long long a = (long long)relA;
#pragma unroll
      for (int c = 0; c < NCH; c++) patch(pf[c], a + (long long)(c * 64 + lane) * KPC);
      patch(pfx, lane == 0 ? a - KPC : a + TILE + (long long)(lane - 1) * KPC);
    }
    // ---- stage the tile (padded rows) and the aux chunks
    wave_sync();
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const unsigned int kk0 = (unsigned int)(c * 64 + lane) * (unsigned int)KPC;   // key offset of the chunk in the tile
      *reinterpret_cast<uint4*>(trow + (kk0 / (unsigned int)V) * (unsigned int)S + (kk0 % (unsigned int)V) * (unsigned int)DW) = pf[c];
    }
    if (lane == 0) *reinterpret_cast<uint4*>(hl) = pfx;
    else if (lane <= G::EXTC) *reinterpret_cast<uint4*>(ext + (lane - 1) * 4) = pfx;
    wave_sync();
    // key (raw bits) at a global index near the tile: LDS where the tile, its front chunk or its look-ahead hold it
    auto key_bits = [&](unsigned int i) -> B {
      const unsigned int rel = i - A;
      if (rel < (unsigned int)TILE) return bits_at(trow + (rel / (unsigned int)V) * (unsigned int)S + (rel % (unsigned int)V) * (unsigned int)DW);
      if (rel + (unsigned int)KPC < (unsigned int)KPC) return bits_at(hl + (rel + (unsigned int)KPC) * (unsigned int)DW);
      if (rel - (unsigned int)TILE < (unsigned int)EXTN) return bits_at(ext + (rel - (unsigned int)TILE) * (unsigned int)DW);
      return key_to_bits<K>(kb[(unsigned int)(i - base32)]);
    };
    auto key_at = [&](unsigned int i) -> K { return bits_to_key<K>(key_bits(i)); };
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
    // ---- the next tile's loads: in flight during everything below
    { const unsigned int nt = tile + wpx; if (nt < t_hi) load_tile(nt); }
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
    if (lane == 63 && edge && n_it > 0u && rel_hi > relA && rel_hi - relA <= (unsigned int)TILE) st->last_target = (unsigned long long)t_tile_last;
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
      // the key in front of the tile: its own first occurrence unless it equals the key before it (then: look it up)
      unsigned int y_tile = A - 1u;
      if (lane == 0 && relA > rel_lo) {
        const long long ia = (long long)relA - 1;                                // relative index of that key: a valid key of the launch
        if (ia - 1 >= rd_lo_rel && key_at(A - 2u) == key_at(A - 1u))
          y_tile = (unsigned int)first_occurrence(keys, (uint64_t)(tile0 + ia), sp.rd_lo);
      }
      y_tile = (unsigned int)__builtin_amdgcn_readfirstlane((int)y_tile);
      y_in = lh_ex ? lh_ex - 1u : y_tile;
    }
    const unsigned int anyb = sc_wave_or(bnd);
    if (RMI_SC_DEBUG && bnd) printf("tile %u lane %d relf %u bnd %08x hd %08x cnt_excl %u nb %u y_in %u edge %d virt %d flo %x fhi %x\n", tile, lane, relf, bnd, hd, cnt_excl, nb, y_in, (int)edge, (int)virt_here, force_lo, force_hi);
    // ---- the split of the 2-way join (two_layer.rs:130-175): given by the host (a shard), else found where it is crossed
    unsigned int w_split = n32, w_stgt = 0u;                                     // the values leaf_container gets for this tile's leaves
    if (host_split) { w_split = (unsigned int)st->split_idx; w_stgt = (unsigned int)st->split_target; }

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
                const unsigned int tp = target_of(kb[reli - 1u], oob);           // (from the key array: LDS holds the launch's first key there)
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
            const K k0 = key_at(lo32), k1 = key_at(hi32);
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
              if (i + 1u - base32 < rel_hi) { const K kn = key_at(i + 1u); if (!(kn == kv)) rm = max(rm, i + 1u - y); }
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
      // ---- P5: the leaf ends.  Lane q: slot qb + q; then the empty leaves in front of the batch's starts.
      const ScanArgs* const cp = cold();
      const ScanOut out = cp->out;
      const int npeers = cp->peers.n;
      auto store_leaf = [&](uint64_t j, uint64_t s, double a, double b2, uint64_t final_err, uint64_t cnt_j) {
        out.leaf_start[j] = s;
        if (out.params) { out.params[2 * j] = a; out.params[2 * j + 1] = b2; }
        if (out.leaf_err) out.leaf_err[j] = final_err;
        if (out.leaf_count) out.leaf_count[j] = cnt_j;
        double* rp = reinterpret_cast<double*>(out.rows + j * 24);
        rp[0] = a; rp[1] = b2;
        *reinterpret_cast<unsigned long long*>(out.rows + j * 24 + 16) = final_err;
        for (int p = 0; p < npeers; p++) {                                       // (wave-uniform trip count)
          unsigned char* const pt = cp->peers.tab[p];
          double* pr = reinterpret_cast<double*>(pt + j * 24);
          pr[0] = a; pr[1] = b2;
          *reinterpret_cast<unsigned long long*>(pt + j * 24 + 16) = final_err;
        }
      };
      if ((unsigned int)lane < own_b) {
        const uint64_t j = q_t, s = q_s, e = q_e;
        if (hand_q) {
          out.leaf_start[j] = s;
          cp->fl.push((unsigned int)j);
        } else {
          double pp[2] = {m_ab[2 * (lane + 1)], m_ab[2 * (lane + 1) + 1]};
          const unsigned int ru = m_run[lane + 1];
          const K k_next = e < sp.n ? key_at((unsigned int)e) : KeyTraits<K>::max_value();
          const K k_prev = s > 0 ? key_at((unsigned int)s - 1u) : KeyTraits<K>::zero_value();
          uint64_t final_err, cnt_j;
          finalize_one_pre<K_LINEAR, K>(j, s, e, sp, r.L, keys, pp, (uint64_t)m_err[lane + 1], ru > 1u ? (uint64_t)ru : 0ull,
                                        e == sp.n ? j : ~0ull, k_next, k_prev, final_err, cnt_j);
          store_leaf(j, s, pp[0], pp[1], final_err, cnt_j);
          agg.add(j, final_err, cnt_j, nf);
        }
      }
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
        const unsigned int glen = gt > g0 ? gt - g0 : 0u;
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
    }
  }
  if (flags) atomicOr(&st->err_flags, flags);
  // ---- this wave's aggregate record
  {
    unsigned long long mx = agg.mx, mi = agg.mi, sm = agg.sum;
    double l2 = agg.l2, lg = agg.lg;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long omx = shfl_down_u64(mx, d), omi = shfl_down_u64(mi, d);
      if (omx > mx || (omx == mx && omi > mi)) { mx = omx; mi = omi; }
      sm += shfl_down_u64(sm, d);
      l2 += __shfl_down(l2, d);
      lg += __shfl_down(lg, d);
    }
    if (lane == 0) cold()->out.partials[blockIdx.x] = StatsPartial{mx, mi, sm, l2, lg};
  }
}

}  // namespace rmi
