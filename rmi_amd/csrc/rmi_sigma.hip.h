// rmi_sigma.hip.h -- the ONE-PASS leaf path for linear leaves (gfx950, wave64): bucketing scan, per-leaf
// least-squares fit from shifted sufficient statistics and the last-level error pass with a single
// coalesced HBM read of the keys (two_layer.rs:43-90, linear.rs:12-59, two_layer.rs:207-217).
//
// Why a second formulation next to the exact streaming kernels (rmi_stream.hip.h): the reference's
// fit is a Welford recurrence, order dependent, so bit-identical coefficients make every leaf a
// sequential chain and the keys have to be read twice (fit, then errors with the finished
// coefficients).  Here a leaf's line comes from sums (n, S dx, S dx^2, S dx dy) that any number of
// lanes can add up, so a block finishes the leaves of a tile while the tile is still in LDS and
// runs the error pass from LDS.  The coefficients then differ from the reference's in the last
// digits (both are roundings of the same least-squares line); the per-leaf error INTEGERS stay
// bit-identical through a guard: every prediction of a leaf (its own keys and the two widening keys
// of two_layer.rs:226-259) must stay further from an integer than a bound `delta` on the distance
// between the two lines; a leaf that fails the test is handed to the exact kernels (list in HBM).
//
// Block = ROWS threads, tile = ROWS rows x 16 keys, thread r owns row r.  Per tile u (software
// pipeline, the keys of tile u+1 are in flight in registers meanwhile):
//   S2  every thread reads the 16 raw keys of its row from LDS (swizzled, conflict-free b128 reads)
//   S3  classification (root target per key, leaf boundaries, duplicates), row sums R relative to
//       the tile pivot, for boundary rows the tail sums T after the boundary, x = f64(key) written
//       back in place; inclusive scan of R over the rows (DPP in the wave, LDS across waves)
//   S4  the boundary rows are numbered (ballots) and publish Q = prefix sums up to their boundary
//   S5  the lane of a boundary row closes the leaf that ends there: sums = Q(end) - Q(start) (+ the
//       carry of the previous tiles) + prev-last + 2 x next-first (the Q1 tail duplicate), solves
//       (alpha, beta), computes delta, writes a leaf entry into an LDS table
//   S6  error pass of tile u-1 FROM LDS with the entries of tiles u-1 and u (a leaf that straddles
//       the two tiles has been closed by now): max |pred - y| and the closest approach of a prediction
//       to an integer per leaf (LDS atomics)
//   S7  entries of tile u-1: guard test -> leaf_maxerr, or the exact-fallback list
// Leaves the sums cannot describe (duplicate keys: y is a first-occurrence offset; the leaves next to
// the split of the 2-way join, Q2/Q3; first and last leaf; rows with several boundaries; leaves longer
// than a tile; variance 0) are "irregular": always handed to the exact kernels.
#pragma once
#include <type_traits>

#include "rmi_device.hip.h"
#include "rmi_kernels.hip.h"
#include "rmi_stream.hip.h"

namespace rmi {

constexpr int SG_CAP = 72;                 // listed boundary rows per tile (more: "dense" tile, all exact)
constexpr unsigned SGF_FOREIGN = 1u;       // entry: the leaf belongs to another block (started before this block's keys)
constexpr unsigned SGF_IRREG = 2u;         // entry: irregular leaf -> exact kernels
constexpr unsigned SGB_MULTI = 1u;         // boundary row: more than one boundary in the row
constexpr unsigned SGB_POISON_OPEN = 2u;   // the leaf opened at this boundary is irregular (split / start of data)
constexpr unsigned SGB_IRREG_CLOSE = 4u;   // the leaf closed at this boundary is irregular (split / end of data)
constexpr unsigned SGC_OWNED = 1u;         // carry: the open leaf started in this block
constexpr unsigned SGC_POISON = 2u;        // carry: the open leaf is irregular

struct SgEntry {                           // 64 bytes
  double alpha, beta;
  double x_next, x_prev;                   // f64 of key[e] and key[s-1]: the widening keys (two_layer.rs:229-247)
  unsigned int leaf;
  unsigned int emax;                       // max |pred - y| over the keys of the leaf
  unsigned int cmin;                       // f32 bits: closest distance of a prediction to an integer
  float delta;                             // guard distance
  unsigned int flags;
  unsigned int _pad[3];
};

struct SgParams {
  uint64_t chunk;                          // keys per block, a multiple of 16
  double guard_k;                          // safety factor of the guard bound
  int mode;                                // 1: guard-flagged leaves are re-fitted exactly; 2: only counted
  unsigned int* flist;                     // leaf ids handed to the exact kernels (capacity: DevState.flag_cap)
};

template <typename K> struct SgGeom {
  static constexpr int ROWB = 16 * (int)sizeof(K);    // bytes per row: 128 / 64
  static constexpr int SLOTS = ROWB / 16;             // 16-byte slots per row: 8 / 4
  static constexpr int KPS = 16 / (int)sizeof(K);     // keys per slot: 2 / 4
  static constexpr int RPB = 256 / ROWB;              // rows per 256-byte bank row: 2 / 4
  // XOR swizzle of the slot index: a lane-per-row ds_read_b128 of slot s then hits 16 distinct 16-byte
  // slots of the bank row in each of the instruction's 16-lane groups, and the 8 (4) lanes that
  // write a row with ds_write_b128 cover it exactly.
  static __device__ __forceinline__ unsigned slot_off(unsigned row, unsigned slot) {
    const unsigned h = (row / RPB) & (SLOTS - 1);
    return row * ROWB + ((slot ^ h) << 4);
  }
  static __device__ __forceinline__ unsigned key_off(unsigned i) {
    return slot_off(i >> 4, (i & 15u) / KPS) + (i & (KPS - 1)) * (unsigned)sizeof(K);
  }
};

// inclusive prefix sum over the 64 lanes of a wave (DPP: shifts inside rows of 16, then two row broadcasts)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double sg_dpp_add(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, ROWMASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, ROWMASK, 0xF, true);
  return v + __builtin_bit_cast(double, ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ __forceinline__ double sg_wave_scan(double v) {
  v = sg_dpp_add<0x111, 0xF>(v);   // row_shr:1
  v = sg_dpp_add<0x112, 0xF>(v);   // row_shr:2
  v = sg_dpp_add<0x114, 0xF>(v);   // row_shr:4
  v = sg_dpp_add<0x118, 0xF>(v);   // row_shr:8
  v = sg_dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
  v = sg_dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
  return v;
}

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

// distance of the leaf prediction at x from the nearest integer, as 0.5 - |fract(f) - 0.5|
__device__ __forceinline__ double sg_closeness(double a, double b, double x) {
  const double f = __builtin_fma(b, x, a);
  const double fr = f - floor(f);
  return 0.5 - fabs(fr - 0.5);
}

// Everything k_sigma keeps in LDS behind the two key buffers.
template <int ROWS> struct alignas(16) SgShared {
  SgEntry tab[2][SG_CAP];                  // leaves closed in the tile, by tile parity
  double bl_q[3][SG_CAP];                  // listed boundary rows: prefix sums of all keys before the (last) boundary of the row
  unsigned long long bl_e[SG_CAP];         // index of the first boundary of the row
  unsigned long long bl_el[SG_CAP];        // index of the last boundary of the row
  double bl_xpl[SG_CAP];                   // x of key[el - 1] (prev-last point of the leaf opened there)
  double bl_xe[SG_CAP];                    // x of key[e]
  unsigned int bl_told[SG_CAP];            // leaf closed at the first boundary
  unsigned int bl_flags[SG_CAP];
  unsigned short bl_row[2][SG_CAP];        // row of the listed boundary (kept one more iteration for the error pass)
  unsigned char bl_fb[2][SG_CAP];          // position of its first boundary (| 0x80: several boundaries)
  double wt[ROWS / 64][3];                 // wave totals of the row sums
  double c_sum[2][3];                      // carry by tile parity: sums of the open leaf's keys of earlier tiles
  double c_xpl[2];                         // x of key[c_s - 1]
  unsigned long long c_s[2];               // start index of the open leaf
  unsigned int c_flags[2];
  unsigned long long prevraw;              // bits of the last raw key of the previous tile
  int wc_lt[ROWS / 64], wc_ge[ROWS / 64];
  int nb[2], dense[2];
  int done;
};

template <int ROOT, typename K, int ROWS>
__global__ void __launch_bounds__(ROWS) k_sigma(const K* __restrict__ keys, Span sp, RootP r, SgParams sg,
                                                unsigned long long* __restrict__ leaf_start, double* __restrict__ params,
                                                unsigned long long* __restrict__ leaf_maxerr, DevState* __restrict__ st) {
  using G = SgGeom<K>;
  constexpr int WAVES = ROWS / 64;
  constexpr int TILE = ROWS * 16;
  constexpr int CAP = SG_CAP;
  constexpr bool WRITE_X = std::is_same<K, uint64_t>::value;     // u64: x = f64(key) replaces the key in LDS
  constexpr int KBUF = ROWS * G::ROWB;

  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  unsigned char* kbuf0 = sg_smem;                                  // [2][KBUF]
  SgShared<ROWS>& sh = *reinterpret_cast<SgShared<ROWS>*>(sg_smem + 2 * KBUF);
  SgEntry* tab0 = &sh.tab[0][0];
  double* bl_q = &sh.bl_q[0][0];
  unsigned long long* bl_e = sh.bl_e;
  unsigned long long* bl_el = sh.bl_el;
  double* bl_xpl = sh.bl_xpl;
  double* bl_xe = sh.bl_xe;
  unsigned int* bl_told = sh.bl_told;
  unsigned int* bl_flags = sh.bl_flags;
  unsigned short* bl_row0 = &sh.bl_row[0][0];
  unsigned char* bl_fb0 = &sh.bl_fb[0][0];
  double* wt = &sh.wt[0][0];
  int* wc_lt = sh.wc_lt;
  int* wc_ge = sh.wc_ge;
  int* s_nb = sh.nb;
  int* s_dense = sh.dense;
  int* s_done = &sh.done;
  unsigned long long* s_prevraw = &sh.prevraw;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint64_t c0 = sp.it_lo + (uint64_t)blockIdx.x * sg.chunk;
  if (c0 >= sp.it_hi) return;
  const uint64_t c1 = (c0 + sg.chunk < sp.it_hi) ? c0 + sg.chunk : sp.it_hi;
  const double Lm1f = (double)(r.L - 1);
  const double midf = (double)(r.L / 2);                            // two_layer.rs:131
  const unsigned int n32 = (unsigned int)sp.n;                      // (the host routes n >= 2^32 to the exact kernels)

  // ---- tile staging: thread `tid` moves SLOTS 16-byte pieces, piece g = k * ROWS + tid ----
  uint4 stage[G::SLOTS];
  auto load_tile = [&](uint64_t A) {
#pragma unroll
    for (int k = 0; k < G::SLOTS; k++) {
      const uint64_t gi = A + (uint64_t)(k * ROWS + tid) * G::KPS;
      if (gi + G::KPS <= sp.rd_hi) {
        stage[k] = sg_load16<K>(keys + gi);
      } else {
        K t[G::KPS];
        const uint64_t last = sp.rd_hi - 1;
#pragma unroll
        for (int q = 0; q < G::KPS; q++) t[q] = keys[gi + q < last ? gi + q : last];
        __builtin_memcpy(&stage[k], t, 16);
      }
    }
  };
  auto store_tile = [&](unsigned char* kb) {
#pragma unroll
    for (int k = 0; k < G::SLOTS; k++) {
      const unsigned g = (unsigned)(k * ROWS + tid);
      *reinterpret_cast<uint4*>(kb + G::slot_off(g / G::SLOTS, g % G::SLOTS)) = stage[k];
    }
  };
  // x = f64(key) of tile position i (after S3 of that tile)
  auto x_at = [&](const unsigned char* kb, unsigned i) -> double {
    if constexpr (WRITE_X || std::is_same<K, double>::value) return *reinterpret_cast<const double*>(kb + G::key_off(i));
    else return KeyTraits<K>::as_float(*reinterpret_cast<const K*>(kb + G::key_off(i)));
  };

  if (tid == 0) {
    sh.c_sum[0][0] = sh.c_sum[0][1] = sh.c_sum[0][2] = 0.0; sh.c_xpl[0] = 0.0; sh.c_s[0] = c0; sh.c_flags[0] = 0u;
    *s_prevraw = (c0 > sp.rd_lo) ? key_to_bits<K>(keys[c0 - 1]) : 0ull;
    s_nb[0] = s_nb[1] = 0; s_dense[0] = s_dense[1] = 0; *s_done = 0;
  }
  load_tile(c0);
  store_tile(kbuf0);
  __syncthreads();

  // per-row state of the previous tile, for its error pass one iteration later
  int prev_seq0 = 0;
  bool prev_listed = false;
  bool prev_dirty = false;
  bool active = true;

  for (unsigned u = 0;; u++) {
    const unsigned par = u & 1u;
    unsigned char* kb = kbuf0 + par * KBUF;
    SgEntry* tab = tab0 + par * CAP;
    const uint64_t A = c0 + (uint64_t)u * TILE;                     // first key index of the tile
    int seq0 = 0;
    bool listed = false;
    bool tile_dirty = false;
    double p = 0.0;
    if (active) {
      load_tile(A + TILE);                                          // prefetch: consumed at the end of the iteration
      // ------------------------------------------------------------------ S2
      K kk[16];
#pragma unroll
      for (int s = 0; s < G::SLOTS; s++) {
        const uint4 v = *reinterpret_cast<const uint4*>(kb + G::slot_off((unsigned)tid, (unsigned)s));
        K t[G::KPS];
        __builtin_memcpy(t, &v, 16);
#pragma unroll
        for (int q = 0; q < G::KPS; q++) kk[s * G::KPS + q] = t[q];
      }
      K kprev;
      if (tid > 0) kprev = *reinterpret_cast<const K*>(kb + G::key_off((unsigned)(16 * tid - 1)));
      else kprev = bits_to_key<K>(*s_prevraw);
      const K kfirst = *reinterpret_cast<const K*>(kb + G::key_off(0u));
      __syncthreads();
      // ------------------------------------------------------------------ S3
      p = KeyTraits<K>::as_float(kfirst);                           // tile pivot
      const uint64_t i0 = A + (uint64_t)(16 * tid);
      const bool interior = (i0 > sp.rd_lo) && (i0 + 16 <= sp.rd_hi);
      double R0 = 0.0, R1 = 0.0, R2 = 0.0;
      double xs[16];
      unsigned int bm = 0;
      bool anyd = false, nonmono = false, oobany = false;
      int fb = 16, lb = 16;
      double t_old1 = 0.0, x_e1 = 0.0, x_plL = 0.0;
      bool split_open = false, split_close = false, start_open = false, end_close = false;
      {
        bool oobp;
        double tp = (i0 > sp.rd_lo) ? root_target_f<ROOT, K>(r, Lm1f, kprev, oobp) : -1.0;
        K kp = kprev;
        double xp = KeyTraits<K>::as_float(kprev);
        uint64_t prevb = 0;                                         // index of the previous boundary of this row
        auto body = [&](auto edge_tag) {
          constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const K k = kk[j];
            const double x = KeyTraits<K>::as_float(k);
            xs[j] = x;
            bool oob;
            const double t = root_target_f<ROOT, K>(r, Lm1f, k, oob);
            bool cmp_ok = true, b, fstart = false, fend = false;
            if constexpr (EDGE) {
              const uint64_t idx = i0 + j;
              cmp_ok = idx > sp.rd_lo && idx < sp.rd_hi;
              fstart = (idx == sp.rd_lo && idx == sp.it_lo);        // the first key of the data: no previous key
              fend = (idx == sp.rd_hi);                             // the end of the (readable) data: no next key
              if constexpr (!root_needs_bounds_check<ROOT>()) oobany |= oob && idx < sp.rd_hi;
            } else {
              if constexpr (!root_needs_bounds_check<ROOT>()) oobany |= oob;
            }
            b = (cmp_ok && t != tp) || fstart || fend;
            nonmono |= cmp_ok && t < tp;
            anyd |= cmp_ok && (k == kp);
            if (b) {
              const uint64_t idx = i0 + j;
              const bool is_split = cmp_ok && tp < midf && t >= midf;      // idx == split_idx (two_layer.rs:132-136)
              if (idx >= c0 && idx < c1 && !fend) {                 // this block owns the leaf that starts here
                leaf_start[(unsigned int)t] = idx;
                if (is_split) {
                  st->split_idx = idx; st->split_target = (unsigned long long)t;
                  if (idx == 0 || idx + 1 >= sp.n) atomicOr(&st->err_flags, EF_DEGENERATE_SPLIT);   // two_layer.rs:27
                }
              }
              if (bm == 0u) {
                fb = j; t_old1 = tp < 0.0 ? 0.0 : tp; x_e1 = x;
                split_close = is_split; end_close = fend;
              } else if (prevb >= c0 && prevb < c1) {
                // a leaf that lies inside this row: exact kernels
                const unsigned long long pos = atomicAdd(&st->flag_count, 1ull);
                if (pos < st->flag_cap) sg.flist[pos] = (unsigned int)tp;
              }
              lb = j; x_plL = xp; prevb = idx;
              split_open = is_split; start_open = fstart;
              bm |= 1u << j;
            }
            const double dx = x - p;
            R0 += dx; R1 = __builtin_fma(dx, dx, R1); R2 = __builtin_fma(dx, (double)j, R2);
            tp = t; kp = k; xp = x;
          }
        };
        if (__all(interior)) body(std::false_type{}); else body(std::true_type{});
      }
      // tail sums after the last boundary of the row
      double T0 = 0.0, T1 = 0.0, T2 = 0.0;
      if (bm != 0u) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          if (j >= lb) {
            const double dx = xs[j] - p;
            T0 += dx; T1 = __builtin_fma(dx, dx, T1); T2 = __builtin_fma(dx, (double)j, T2);
          }
        }
      }
      if constexpr (WRITE_X) {
#pragma unroll
        for (int s = 0; s < G::SLOTS; s++) {
          double2 v = make_double2(xs[2 * s], xs[2 * s + 1]);
          *reinterpret_cast<double2*>(kb + G::slot_off((unsigned)tid, (unsigned)s)) = v;
        }
      }
      if (tid == ROWS - 1) *s_prevraw = key_to_bits<K>(kk[15]);
      if ((uint64_t)(sp.n - 1) >= i0 && (uint64_t)(sp.n - 1) < i0 + 16 && sp.n - 1 < sp.rd_hi && sp.n - 1 >= c0 && sp.n - 1 < c1) {
        bool oob_;
        st->last_target = (unsigned long long)root_target_f<ROOT, K>(r, Lm1f, keys[sp.n - 1], oob_);
      }
      {
        unsigned int ef = 0;
        if (nonmono) ef |= EF_NON_MONOTONE;                         // two_layer.rs:50 / :144
        if (oobany) ef |= EF_ROOT_OOB;                              // two_layer.rs:45-48
        if (ef) atomicOr(&st->err_flags, ef);
      }
      // dy = 16 * tid + j relative to the tile's first index
      const double rowf = (double)(16 * tid);
      R2 = __builtin_fma(rowf, R0, R2);
      T2 = __builtin_fma(rowf, T0, T2);
      double P0 = sg_wave_scan(R0), P1 = sg_wave_scan(R1), P2 = sg_wave_scan(R2);
      if (lane == 63) { wt[wv * 3 + 0] = P0; wt[wv * 3 + 1] = P1; wt[wv * 3 + 2] = P2; }
      const bool hasb = bm != 0u;
      const uint64_t e_first = i0 + (uint64_t)(fb & 15);
      const unsigned long long m_lt = __ballot(hasb && e_first < c1);
      const unsigned long long m_ge = __ballot(hasb && e_first >= c1);
      if (lane == 0) { wc_lt[wv] = __popcll(m_lt); wc_ge[wv] = m_ge != 0ull; }
      tile_dirty = __syncthreads_or(anyd) != 0;
      const bool dirty = tile_dirty || prev_dirty;                  // (prev-last y of a leaf may sit in the previous tile)
      // ------------------------------------------------------------------ S4
      double tot0 = 0.0, tot1 = 0.0, tot2 = 0.0;
      int base = 0, nlt = 0, ge_before = 0, ge_any = 0;
#pragma unroll
      for (int w = 0; w < WAVES; w++) {
        const double a0 = wt[w * 3 + 0], a1 = wt[w * 3 + 1], a2 = wt[w * 3 + 2];
        if (w < wv) { P0 += a0; P1 += a1; P2 += a2; base += wc_lt[w]; ge_before |= wc_ge[w]; }
        tot0 += a0; tot1 += a1; tot2 += a2;
        nlt += wc_lt[w]; ge_any |= wc_ge[w];
      }
      const unsigned long long below = (1ull << lane) - 1ull;
      const bool is_stop = hasb && e_first >= c1 && !ge_before && (m_ge & below) == 0ull;   // first boundary at or behind c1
      listed = (hasb && e_first < c1) || is_stop;
      seq0 = base + __popcll(m_lt & below) + ((ge_before || (m_ge & below) != 0ull) ? 1 : 0);
      const int q = base + __popcll(m_lt & below);                  // (for the stop row: all rows with e < c1 come first)
      const int nb = nlt + (ge_any ? 1 : 0);
      const bool dense = nb > CAP;
      if (listed && !dense) {
        bl_q[0 * CAP + q] = P0 - T0; bl_q[1 * CAP + q] = P1 - T1; bl_q[2 * CAP + q] = P2 - T2;
        bl_e[q] = e_first; bl_el[q] = i0 + (uint64_t)(lb & 15);
        bl_xpl[q] = x_plL; bl_xe[q] = x_e1; bl_told[q] = (unsigned int)t_old1;
        bl_flags[q] = (fb != lb ? SGB_MULTI : 0u) | ((split_open || start_open) ? SGB_POISON_OPEN : 0u) |
                      ((split_close || end_close) ? SGB_IRREG_CLOSE : 0u);
        bl_row0[par * CAP + q] = (unsigned short)tid;
        bl_fb0[par * CAP + q] = (unsigned char)(fb | (fb != lb ? 0x80 : 0));
      }
      __syncthreads();
      // ------------------------------------------------------------------ S5
      if (listed) {
        const bool first = (q == 0);
        const unsigned int cfl = sh.c_flags[par];
        const bool owned = first ? (cfl & SGC_OWNED) != 0u : true;
        if (dense) {
          if (owned) {
            const unsigned long long pos = atomicAdd(&st->flag_count, 1ull);
            if (pos < st->flag_cap) sg.flist[pos] = (unsigned int)t_old1;
          }
        } else {
          double S0, S1, S2, xpl;
          uint64_t s;
          unsigned int pfl;
          if (first) {
            S0 = bl_q[0 * CAP + q] + sh.c_sum[par][0]; S1 = bl_q[1 * CAP + q] + sh.c_sum[par][1]; S2 = bl_q[2 * CAP + q] + sh.c_sum[par][2];
            s = sh.c_s[par]; xpl = sh.c_xpl[par]; pfl = (cfl & SGC_POISON) ? SGB_POISON_OPEN : 0u;
          } else {
            S0 = bl_q[0 * CAP + q] - bl_q[0 * CAP + q - 1]; S1 = bl_q[1 * CAP + q] - bl_q[1 * CAP + q - 1];
            S2 = bl_q[2 * CAP + q] - bl_q[2 * CAP + q - 1];
            s = bl_el[q - 1]; xpl = bl_xpl[q - 1]; pfl = bl_flags[q - 1] & SGB_POISON_OPEN;
          }
          const unsigned int myfl = bl_flags[q];
          bool irregular = dirty || (myfl & (SGB_MULTI | SGB_IRREG_CLOSE)) != 0u || pfl != 0u;
          const uint64_t e = e_first;
          // prev-last point (two_layer.rs:74-78) and next-first point twice (two_layer.rs:58-59 + the Q1 tail duplicate)
          const double dxs = xpl - p, dys = (double)((long long)s - 1ll - (long long)A);
          const double dxe = x_e1 - p, dye = (double)((long long)e - (long long)A);
          S0 += dxs + 2.0 * dxe;
          S1 += dxs * dxs + 2.0 * (dxe * dxe);
          S2 += dxs * dys + 2.0 * (dxe * dye);
          const double cnt = (double)(e - s) + 3.0;
          const double m = dye - dys + 1.0;
          const double sy = m * (dys + dye) * 0.5 + dye;
          const double rn = 1.0 / cnt;
          const double mx = S0 * rn, my = sy * rn;
          const double m2 = __builtin_fma(-S0, mx, S1);
          const double cxy = __builtin_fma(-S0, my, S2);
          double beta = 0.0, alpha = 0.0, delta = 1.0;
          if (!(m2 > 0.0) || !(S1 < 1.7e308)) irregular = true;      // all keys on one f64 (linear.rs:50-53), overflow, NaN
          else {
            beta = cxy / m2;
            alpha = ((double)A + my) - beta * (p + mx);
            if (!(fabs(beta) < 1.7e308) || !(fabs(alpha) < 1.7e308)) irregular = true;
            // Guard distance: a first-order bound on |reference line - this line| over the keys of the leaf.
            //   reference (Welford, linear.rs:24-34): the running mean of x is rounded to ulp(X) at each of the n steps,
            //   and every step's dx inherits it -> n u |beta| X (offset) and n u |beta| W X / sigma_x (slope, over the range W);
            //   both lines: one rounding of alpha at the size of the prediction, u (|beta| X + Y);
            //   these sums: cancellation of S dx^2 against (S dx)^2 / n, u (S dx^2 / m2) |beta| W.
            const double X = fmax(fabs(xpl), fabs(x_e1)), W = x_e1 - xpl, ab = fabs(beta);
            const double sigma = sqrt(m2 * rn);
            const double cond = S1 / m2;
            delta = sg.guard_k * 1.1102230246251565e-16 * (cnt * ab * X * (1.0 + W / sigma) + ab * X + (double)e + 4.0 * cond * ab * W);
            if (!(delta < 0.5)) irregular = true;                    // nothing to gain from the sums
          }
          SgEntry en;
          en.alpha = alpha; en.beta = beta; en.x_next = x_e1; en.x_prev = xpl;
          en.leaf = (unsigned int)t_old1; en.emax = 0u; en.cmin = __float_as_uint(0.5f);
          en.delta = (float)(delta * 1.000001);
          en.flags = (owned ? 0u : SGF_FOREIGN) | (irregular ? SGF_IRREG : 0u);
          en._pad[0] = en._pad[1] = en._pad[2] = 0u;
          tab[q] = en;
          if (owned && !irregular) { params[2ull * en.leaf] = alpha; params[2ull * en.leaf + 1] = beta; }
        }
        if (q == nb - 1) {                                           // the open leaf behind the last listed boundary
          if (ge_any) *s_done = 1;
          else {
            sh.c_sum[par ^ 1u][0] = tot0 - (P0 - T0); sh.c_sum[par ^ 1u][1] = tot1 - (P1 - T1); sh.c_sum[par ^ 1u][2] = tot2 - (P2 - T2);
            sh.c_s[par ^ 1u] = i0 + (uint64_t)(lb & 15); sh.c_xpl[par ^ 1u] = x_plL;
            sh.c_flags[par ^ 1u] = SGC_OWNED | ((split_open || start_open || dirty || dense) ? SGC_POISON : 0u);
          }
        }
      }
      if (nb == 0 && tid == 0) {
        // no boundary in the whole tile: the open leaf is longer than a tile (its rows of the previous tile
        // cannot get their errors from LDS any more): irregular
        sh.c_sum[par ^ 1u][0] = sh.c_sum[par][0] + tot0; sh.c_sum[par ^ 1u][1] = sh.c_sum[par][1] + tot1; sh.c_sum[par ^ 1u][2] = sh.c_sum[par][2] + tot2;
        sh.c_s[par ^ 1u] = sh.c_s[par]; sh.c_xpl[par ^ 1u] = sh.c_xpl[par];
        sh.c_flags[par ^ 1u] = sh.c_flags[par] | SGC_POISON;
      }
      if (tid == 0) { s_nb[par] = dense ? 0 : nb; s_dense[par] = dense; }
      __syncthreads();
    }
    // -------------------------------------------------------------------- S6: error pass of tile u-1 from LDS
    if (u > 0) {
      const unsigned pv = par ^ 1u;
      const unsigned char* kv = kbuf0 + pv * KBUF;
      SgEntry* tv = tab0 + pv * CAP;
      const int nbv = s_nb[pv];
      const bool have_next = active && !s_dense[par] && s_nb[par] > 0;
      const uint64_t Av = A - TILE;
      if (!s_dense[pv]) {
        auto entry_of = [&](int q) -> SgEntry* {
          if (q < nbv) return tv + q;
          if (q == nbv && have_next) return tab;                    // the leaf straddles the two tiles: closed by tile u's first boundary
          return nullptr;
        };
        if (!prev_listed) {                                          // a row of one leaf
          SgEntry* en = entry_of(prev_seq0);
          if (en && !(en->flags & (SGF_FOREIGN | SGF_IRREG))) {
            const double a = en->alpha, b = en->beta;
            unsigned int emax = 0u;
            double hmax = 0.0;
            const unsigned int y0 = (unsigned int)(Av + (uint64_t)(16 * tid));
#pragma unroll
            for (int s = 0; s < G::SLOTS; s++) {
              const uint4 v = *reinterpret_cast<const uint4*>(kv + G::slot_off((unsigned)tid, (unsigned)s));
              double xv[G::KPS];
              if constexpr (sizeof(K) == 8) {
                double t[2]; __builtin_memcpy(t, &v, 16); xv[0] = t[0]; xv[1] = t[1];
              } else {
                K t[G::KPS]; __builtin_memcpy(t, &v, 16);
#pragma unroll
                for (int qq = 0; qq < G::KPS; qq++) xv[qq] = KeyTraits<K>::as_float(t[qq]);
              }
#pragma unroll
              for (int qq = 0; qq < G::KPS; qq++) {
                const double f = __builtin_fma(b, xv[qq], a);       // linear.rs:87-90
                const unsigned int pr = min(sg_cvt_u32(f), n32);    // models/mod.rs:735-737, two_layer.rs:14-18
                emax = max(emax, sg_absdiff(pr, y0 + (unsigned)(s * G::KPS + qq)));
                hmax = fmax_abs_raw(hmax, __builtin_amdgcn_fract(f) - 0.5);
              }
            }
            atomicMax(&en->emax, emax);
            atomicMin(&en->cmin, __float_as_uint((float)(0.5 - hmax) * 0.999999f));
          }
        }
        // rows with a boundary, one key per lane: keys before the boundary belong to the leaf it closes,
        // the others to the next entry (rows with several boundaries only touch irregular leaves)
        const unsigned short* brow = bl_row0 + pv * CAP;
        const unsigned char* bfb = bl_fb0 + pv * CAP;
        for (int item = tid; item < nbv * 16; item += ROWS) {
          const int q = item >> 4, j = item & 15;
          const unsigned int fbq = bfb[q];
          if (fbq & 0x80u) continue;
          SgEntry* en = entry_of((unsigned)j < fbq ? q : q + 1);
          if (!en || (en->flags & (SGF_FOREIGN | SGF_IRREG))) continue;
          const unsigned int ipos = (unsigned int)brow[q] * 16u + (unsigned int)j;
          const uint64_t idx = Av + ipos;
          if (idx >= sp.it_hi) continue;
          const double f = __builtin_fma(en->beta, x_at(kv, ipos), en->alpha);
          const unsigned int pr = min(sg_cvt_u32(f), n32);
          atomicMax(&en->emax, sg_absdiff(pr, (unsigned int)idx));
          atomicMin(&en->cmin, __float_as_uint((float)(0.5 - fabs(__builtin_amdgcn_fract(f) - 0.5)) * 0.999999f));
        }
      }
      __syncthreads();
      // ------------------------------------------------------------------ S7: the leaves closed in tile u-1
      if (tid < nbv) {
        const SgEntry en = tv[tid];
        if (!(en.flags & SGF_FOREIGN)) {
          bool exact = (en.flags & SGF_IRREG) != 0u;
          if (!exact) {
            // widening keys (two_layer.rs:229-247): key[e] - 1 and key[s-1] + 1 as f64.  For u64 keys only
            // x = RN(key) is at hand: RN(key -+ 1) is x -+ 1 below 2^53 and x or its neighbour above.
            double c = (double)__uint_as_float(en.cmin);
            const double xn = en.x_next, xp = en.x_prev;
            if constexpr (std::is_same<K, double>::value) {
              c = fmin(c, sg_closeness(en.alpha, en.beta, KeyTraits<K>::minus_eps(xn)));
              c = fmin(c, sg_closeness(en.alpha, en.beta, KeyTraits<K>::plus_eps(xp)));
            } else {
              c = fmin(c, sg_closeness(en.alpha, en.beta, xn - 1.0));
              c = fmin(c, sg_closeness(en.alpha, en.beta, xp + 1.0));
              if constexpr (std::is_same<K, uint64_t>::value) {
                c = fmin(c, sg_closeness(en.alpha, en.beta, xn));
                c = fmin(c, sg_closeness(en.alpha, en.beta, __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, xn) - 1ull)));
                c = fmin(c, sg_closeness(en.alpha, en.beta, xp));
                c = fmin(c, sg_closeness(en.alpha, en.beta, __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, xp) + 1ull)));
              }
            }
            if (!(c >= (double)en.delta)) {
              atomicAdd(&st->guard_count, 1ull);
              exact = sg.mode == 1;
            }
          }
          if (exact) {
            const unsigned long long pos = atomicAdd(&st->flag_count, 1ull);
            if (pos < st->flag_cap) sg.flist[pos] = en.leaf;
          } else leaf_maxerr[en.leaf] = (unsigned long long)en.emax;
        }
      }
    }
    if (!active) break;
    if (*s_done) active = false;
    prev_seq0 = seq0; prev_listed = listed; prev_dirty = tile_dirty;
    __syncthreads();                                                 // S6/S7 done with kbuf[par^1], s_done read by all
    if (active) {
      store_tile(kbuf0 + (par ^ 1u) * KBUF);
      __syncthreads();
      if (tid == 0) {
        // carry to the pivots of the next tile: x - p' = (x - p) + d,  i - A' = (i - A) - TILE
        double* cs = sh.c_sum[par ^ 1u];
        const double pn = KeyTraits<K>::as_float(*reinterpret_cast<const K*>(kbuf0 + (par ^ 1u) * KBUF + G::key_off(0u)));
        const double d = p - pn;
        const uint64_t s = sh.c_s[par ^ 1u];
        const double mcnt = (double)(A + TILE - s);                  // keys of the open leaf so far: [s, A + TILE)
        const double a = (double)((long long)s - (long long)A), b = (double)TILE - 1.0;
        const double sy = mcnt * (a + b) * 0.5;                      // S (i - A) over them
        const double sx = cs[0], sxx = cs[1], sxy = cs[2];
        const double dq = -(double)TILE;
        cs[0] = sx + mcnt * d;
        cs[1] = sxx + 2.0 * d * sx + mcnt * d * d;
        cs[2] = sxy + d * sy + dq * sx + mcnt * d * dq;
      }
      __syncthreads();
    }
  }
}

// LDS bytes of k_sigma
template <typename K, int ROWS>
constexpr size_t sg_smem_bytes() { return (size_t)2 * ROWS * SgGeom<K>::ROWB + sizeof(SgShared<ROWS>); }

// ---------------------------------------------------------------------------------------------
// Exact kernels for the leaves k_sigma hands over (DevState.flag_count entries of `flist`).
// k_fit_list: one lane per leaf, the reference's recurrence on the reference's container
// (fit_one_leaf); leaves beyond `long_min` points go on to k_fit_long (one wave each).
// k_err_list: one wave per leaf, error pass + run lengths of its keys (two_layer.rs:207-217,
// lower_bound_correction.rs:104-119), exactly what k_err computes per key.
// ---------------------------------------------------------------------------------------------
template <typename K>
__global__ void __launch_bounds__(256) k_fit_list(const K* __restrict__ keys, Span sp,
                                                  const unsigned long long* __restrict__ leaf_start, DevState* __restrict__ st,
                                                  double* __restrict__ params, const unsigned int* __restrict__ flist,
                                                  unsigned long long* __restrict__ long_idx, unsigned int long_min) {
  const unsigned long long cnt = st->flag_count < st->flag_cap ? st->flag_count : st->flag_cap;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint64_t j = flist[i];
    const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
    if (e - s > (uint64_t)long_min) {
      const unsigned long long pos = atomicAdd(&st->long_count, 1ull);
      if (pos < st->long_cap) long_idx[pos] = s;
    } else {
      fit_one_leaf<K_LINEAR, K>(j, keys, sp, leaf_start, st, params);
    }
  }
}

template <typename K>
__global__ void __launch_bounds__(64) k_err_list(const K* __restrict__ keys, Span sp,
                                                 const unsigned long long* __restrict__ leaf_start, const DevState* __restrict__ st,
                                                 const double* __restrict__ params, const unsigned int* __restrict__ flist,
                                                 unsigned long long* __restrict__ leaf_maxerr, unsigned long long* __restrict__ leaf_run) {
  const unsigned long long cnt = st->flag_count < st->flag_cap ? st->flag_count : st->flag_cap;
  const int lane = threadIdx.x;
  for (unsigned long long t = blockIdx.x; t < cnt; t += gridDim.x) {
    const uint64_t j = flist[t];
    const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
    unsigned long long err = 0, run = 0;
    for (uint64_t i = s + lane; i < e; i += 64) {
      const K k = keys[i];
      const uint64_t y = first_occurrence(keys, i, sp.rd_lo);
      const uint64_t pred = leaf_predict<K_LINEAR, K>(params + j * 2, k);
      const uint64_t er = error_between(pred, y, sp.n);
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
}


// =============================================================================================
// k_sigma2: the same one-pass path with AUTONOMOUS WAVES (what the counters of k_sigma asked for:
// it spent half of its wave time in barriers / LDS round trips between its phases, and ~78 vector
// instructions per 64 keys, a third of them boundary bookkeeping inside fixed 16-key rows).
//
// A wave (= a block of 64 threads) owns a contiguous chunk of the keys and never synchronises
// with another wave.  Per batch of BATCH keys:
//   phase 1  classification straight from the registers the coalesced 16-byte loads landed in (the
//            keys of the next batch are in flight meanwhile): x = f64(key), root target, leaf
//            boundary / duplicate test against the previous key (the previous lane's, by DPP);
//            x goes to a ring in LDS (a duplicate key as NaN: its leaf then fails the checks below
//            and goes to the exact kernels), a boundary appends (index, leaf id, flags) to a list.
//   phase 2  every leaf that is complete in the ring (both boundaries seen) is handled by a GROUP of
//            GL lanes, 64/GL leaves per round: the lanes stride over the leaf's container
//            [s-1, e] (+ the Q1 duplicate of e) adding up (S dx, S dx^2, S dx dy) relative to the
//            leaf's first key, butterfly all-reduce inside the group (DPP), every lane solves
//            (alpha, beta, delta), then the lanes stride over the leaf's own keys for
//            max |pred - y| and the closest approach of a prediction to an integer (+ the widening keys
//            on six lanes), butterfly max, lane 0 of the group decides: leaf_maxerr, or the exact list.
// A leaf that does not fit the ring (RING keys) is irregular ("long"); so are the leaves at the split
// of the 2-way join, the first and the last leaf, and leaves with duplicate keys -- as in k_sigma.
// =============================================================================================
constexpr unsigned S2_SPLIT = 1u, S2_START = 2u, S2_END = 4u, S2_LONG = 8u;

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
template <int GL> __device__ __forceinline__ double s2_group_sum(double v) {
  v += s2_dpp_f64<0xB1>(v);                        // quad_perm [1,0,3,2]
  v += s2_dpp_f64<0x4E>(v);                        // quad_perm [2,3,0,1]
  if constexpr (GL >= 8) v += s2_dpp_f64<0x141>(v);    // row_half_mirror
  if constexpr (GL >= 16) v += s2_dpp_f64<0x140>(v);   // row_mirror
  return v;
}
template <int GL> __device__ __forceinline__ double s2_group_max(double v) {
  v = fmax_raw(v, s2_dpp_f64<0xB1>(v));
  v = fmax_raw(v, s2_dpp_f64<0x4E>(v));
  if constexpr (GL >= 8) v = fmax_raw(v, s2_dpp_f64<0x141>(v));
  if constexpr (GL >= 16) v = fmax_raw(v, s2_dpp_f64<0x140>(v));
  return v;
}
template <int GL> __device__ __forceinline__ unsigned int s2_group_max(unsigned int v) {
  v = max(v, s2_dpp_u32<0xB1>(v));
  v = max(v, s2_dpp_u32<0x4E>(v));
  if constexpr (GL >= 8) v = max(v, s2_dpp_u32<0x141>(v));
  if constexpr (GL >= 16) v = max(v, s2_dpp_u32<0x140>(v));
  return v;
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

template <int ROOT, typename K, int RING, int BATCH, int GL>
__global__ void __launch_bounds__(64) k_sigma2(const K* __restrict__ keys, Span sp, RootP r, SgParams sg,
                                               unsigned long long* __restrict__ leaf_start, double* __restrict__ params,
                                               unsigned long long* __restrict__ leaf_maxerr, DevState* __restrict__ st) {
  constexpr int KPL = 16 / (int)sizeof(K);          // keys per lane and load
  constexpr int NLOAD = BATCH / (64 * KPL);         // loads per batch
  constexpr int NG = 64 / GL;                       // leaves per round
  constexpr int BCAP = 256;                         // boundary list (more boundaries in a batch: see `dense`)
  constexpr unsigned MASK = RING - 1;
  static_assert((RING & (RING - 1)) == 0 && BATCH % (64 * KPL) == 0 && RING >= 2 * BATCH, "geometry");
  __shared__ double xring[RING];
  __shared__ unsigned int b_idx[BCAP], b_t[BCAP];
  __shared__ unsigned char b_fl[BCAP];

  const int lane = threadIdx.x;
  const uint64_t c0 = sp.it_lo + (uint64_t)blockIdx.x * sg.chunk;
  if (c0 >= sp.it_hi) return;
  const uint64_t c1 = (c0 + sg.chunk < sp.it_hi) ? c0 + sg.chunk : sp.it_hi;
  const unsigned int c0u = (unsigned int)c0, c1u = (unsigned int)c1;
  const double Lm1f = (double)(r.L - 1);
  const unsigned int Lm1 = (unsigned int)(r.L - 1);
  const unsigned int mid = (unsigned int)(r.L / 2);                 // two_layer.rs:131
  const unsigned int n32 = (unsigned int)sp.n;
  const unsigned long long below = (1ull << lane) - 1ull;

  uint4 cur[NLOAD], nxt[NLOAD];
  auto load_batch = [&](uint4 (&dst)[NLOAD], uint64_t A) {
#pragma unroll
    for (int k = 0; k < NLOAD; k++) {
      const uint64_t gi = A + (uint64_t)(k * 64 + lane) * KPL;
      if (gi + KPL <= sp.rd_hi) dst[k] = sg_load16<K>(keys + gi);
      else {
        K t[KPL];
        const uint64_t last = sp.rd_hi - 1;
#pragma unroll
        for (int q = 0; q < KPL; q++) t[q] = keys[gi + q < last ? gi + q : last];
        __builtin_memcpy(&dst[k], t, 16);
      }
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
  }
  int bcnt = 0;                                                     // entries of the boundary list (wave-uniform)
  bool stop = false;                                                // a boundary at or behind c1 is in the list
  unsigned int eflags = 0;

  // ---- phase 2: the complete leaves [entry i, entry i+1) of the list ----
  auto rounds = [&](bool drop_open) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int g = lane / GL, l = lane % GL;
    for (int i0 = 0; i0 + 1 < bcnt; i0 += NG) {
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
      // sums over the container [s-1, e] relative to (x[s], s)
      double p = 0.0, xpl = 0.0, xe = 0.0;
      double R0 = 0.0, R1 = 0.0, R2 = 0.0;
      if (work) {
        p = xring[s & MASK]; xpl = xring[(s - 1u) & MASK]; xe = xring[e & MASK];
        double dy = (double)(l - 1);
        for (unsigned int k = s - 1u + (unsigned int)l; k <= e; k += GL) {
          const double dx = xring[k & MASK] - p;
          R0 += dx; R1 = __builtin_fma(dx, dx, R1); R2 = __builtin_fma(dx, dy, R2);
          dy += (double)GL;
        }
        if (l == 0) {                                               // the Q1 tail duplicate of the next-first point
          const double dx = xe - p, dyy = (double)(e - s);
          R0 += dx; R1 = __builtin_fma(dx, dx, R1); R2 = __builtin_fma(dx, dyy, R2);
        }
      }
      R0 = s2_group_sum<GL>(R0); R1 = s2_group_sum<GL>(R1); R2 = s2_group_sum<GL>(R2);
      // every lane of the group solves
      const double len = (double)(e - s);
      const double cnt = len + 3.0;
      const double sy = (len + 2.0) * (len - 1.0) * 0.5 + len;      // S (i - s) over i = s-1 .. e, plus (e - s) once more
      const double rn = 1.0 / cnt;
      const double mx = R0 * rn, my = sy * rn;
      const double m2 = __builtin_fma(-R0, mx, R1);
      const double cxy = __builtin_fma(-R0, my, R2);
      double alpha = 0.0, beta = 0.0, delta = 1.0;
      if (!(m2 > 0.0) || !(R1 < 1.7e308)) irregular = true;         // all keys on one f64 (linear.rs:50-53), NaN (duplicates), overflow
      else {
        beta = cxy / m2;
        alpha = ((double)s + my) - beta * (p + mx);
        if (!(fabs(beta) < 1.7e308) || !(fabs(alpha) < 1.7e308)) irregular = true;
        const double X = fmax(fabs(xpl), fabs(xe)), W = xe - xpl, ab = fabs(beta);
        const double sigma = sqrt(m2 * rn);
        const double cond = R1 / m2;
        delta = sg.guard_k * 1.1102230246251565e-16 * (cnt * ab * X * (1.0 + W / sigma) + ab * X + (double)e + 4.0 * cond * ab * W);
        if (!(delta < 0.5)) irregular = true;
      }
      // error pass over the own keys [s, e)
      unsigned int emax = 0u;
      double hmax = 0.0;
      if (work && !irregular) {
        for (unsigned int k = s + (unsigned int)l; k < e; k += GL) {
          const double f = __builtin_fma(beta, xring[k & MASK], alpha);    // linear.rs:87-90
          const unsigned int pr = min(sg_cvt_u32(f), n32);                  // models/mod.rs:735-737, two_layer.rs:14-18
          emax = max(emax, sg_absdiff(pr, k));
          hmax = fmax_abs_raw(hmax, __builtin_amdgcn_fract(f) - 0.5);
        }
        // widening keys (two_layer.rs:229-247), one candidate per lane: RN(key[e] - 1), RN(key[s-1] + 1)
        if (l < 6) {
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
      hmax = s2_group_max<GL>(hmax);
      if (owned && l == 0) {
        bool exact = irregular;
        if (!exact && !(0.5 - hmax >= delta)) {
          atomicAdd(&st->guard_count, 1ull);
          exact = sg.mode == 1;
        }
        if (!irregular) { params[2ull * leaf] = alpha; params[2ull * leaf + 1] = beta; }
        if (exact) {
          const unsigned long long pos = atomicAdd(&st->flag_count, 1ull);
          if (pos < st->flag_cap) sg.flist[pos] = leaf;
        } else leaf_maxerr[leaf] = (unsigned long long)emax;
      }
    }
    // the last entry stays: the start of the open leaf (after a dense batch the list restarts empty)
    if (drop_open) bcnt = 0;
    else if (bcnt > 1) {
      const unsigned int li = b_idx[bcnt - 1], lt = b_t[bcnt - 1];
      const unsigned char lf = b_fl[bcnt - 1];
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { b_idx[0] = li; b_t[0] = lt; b_fl[0] = lf; }
      bcnt = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  load_batch(nxt, c0);
  for (uint64_t A = c0; !stop; A += BATCH) {
#pragma unroll
    for (int k = 0; k < NLOAD; k++) cur[k] = nxt[k];
    load_batch(nxt, A + BATCH);
    // does the open leaf still fit the ring once this batch is in?  (its container starts at s - 1)
    if (bcnt > 0) {
      const uint64_t s_open = b_idx[0];
      if (A + BATCH - (s_open - 1) > (uint64_t)RING && lane == 0) b_fl[0] |= S2_LONG;
    }
    const bool interior = (A > sp.rd_lo) && (A + BATCH <= sp.rd_hi);
    auto phase1 = [&](auto edge_tag, bool& dense) {
      constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
      for (int k = 0; k < NLOAD; k++) {
        K kk[KPL];
        __builtin_memcpy(kk, &cur[k], 16);
        const uint64_t g0 = A + (uint64_t)(k * 64 + lane) * KPL;
        // the key / target before this lane's first key: the previous lane's last one (lane 0: the carry)
        K kp;
        unsigned int tp;
        double xs[KPL];
        unsigned int ts[KPL];
        bool oobany = false;
#pragma unroll
        for (int q = 0; q < KPL; q++) {
          xs[q] = KeyTraits<K>::as_float(kk[q]);
          bool oob;
          ts[q] = s2_target<ROOT, K>(r, Lm1f, Lm1, kk[q], xs[q], oob);
          if constexpr (!root_needs_bounds_check<ROOT>()) {
            if constexpr (EDGE) oobany |= oob && (g0 + q < sp.rd_hi); else oobany |= oob;
          }
        }
        {
          const unsigned long long lastb = key_to_bits<K>(kk[KPL - 1]), cb = key_to_bits<K>(carry_key);
          const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)cb, (int)(unsigned int)lastb, 0x138, 0xF, 0xF, false);   // wave_shr:1
          unsigned int hi = 0u;
          if constexpr (sizeof(K) == 8) hi = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)(cb >> 32), (int)(unsigned int)(lastb >> 32), 0x138, 0xF, 0xF, false);
          kp = bits_to_key<K>(((unsigned long long)hi << 32) | lo);
          tp = (unsigned int)__builtin_amdgcn_update_dpp((int)carry_t, (int)ts[KPL - 1], 0x138, 0xF, 0xF, false);
        }
        bool bq[KPL];
        bool anyd = false, nonmono = false;
        unsigned int fq[KPL];
        unsigned int tprev[KPL];
#pragma unroll
        for (int q = 0; q < KPL; q++) {
          bool cmp_ok = true, fstart = false, fend = false;
          if constexpr (EDGE) {
            const uint64_t idx = g0 + q;
            cmp_ok = idx > sp.rd_lo && idx < sp.rd_hi;
            fstart = (idx == sp.rd_lo && idx == sp.it_lo);
            fend = (idx == sp.rd_hi);
          }
          const bool dupq = cmp_ok && (kk[q] == kp);
          anyd |= dupq;
          nonmono |= cmp_ok && ts[q] < tp;
          bq[q] = (cmp_ok && ts[q] != tp) || fstart || fend;
          fq[q] = ((cmp_ok && tp < mid && ts[q] >= mid) || (fstart && ts[q] >= mid) ? S2_SPLIT : 0u) | (fstart ? S2_START : 0u) | (fend ? S2_END : 0u);
          tprev[q] = tp;
          if (dupq) xs[q] = __builtin_nan("");                      // y of a duplicate is not its index: the leaf goes to the exact kernels
          kp = kk[q]; tp = ts[q];
        }
        (void)tprev;
        // x into the ring (16 bytes per store)
        {
          const unsigned int off = (unsigned int)g0 & MASK;
#pragma unroll
          for (int q = 0; q < KPL; q += 2) *reinterpret_cast<double2*>(&xring[off + q]) = make_double2(xs[q], xs[q + 1]);
        }
        // boundaries -> list, in index order
        unsigned long long mq[KPL];
        unsigned long long many = 0ull;
#pragma unroll
        for (int q = 0; q < KPL; q++) { mq[q] = __ballot(bq[q]); many |= mq[q]; }
        if (many) {
          int add = 0;
#pragma unroll
          for (int q = 0; q < KPL; q++) add += __popcll(mq[q]);
          if (!dense && bcnt + add > BCAP) {
            // More boundaries than the list holds (leaves of a few keys): for the rest of this batch every leaf
            // goes straight to the exact kernels, the open one included; the list restarts with the next batch.
            dense = true;
            if (lane == 0 && bcnt > 0 && b_idx[bcnt - 1] >= c0u && b_idx[bcnt - 1] < c1u) {
              const unsigned long long pos = atomicAdd(&st->flag_count, 1ull);
              if (pos < st->flag_cap) sg.flist[pos] = b_t[bcnt - 1];
            }
          }
          int rank = bcnt;
#pragma unroll
          for (int q = 0; q < KPL; q++) rank += __popcll(mq[q] & below);
#pragma unroll
          for (int q = 0; q < KPL; q++) {
            if (bq[q]) {
              const uint64_t idx = g0 + q;
              const bool own = idx >= c0 && idx < c1 && !(fq[q] & S2_END);     // this wave owns the leaf that starts here
              if (!dense) { b_idx[rank] = (unsigned int)idx; b_t[rank] = ts[q]; b_fl[rank] = (unsigned char)fq[q]; }
              else if (own) {
                const unsigned long long pos = atomicAdd(&st->flag_count, 1ull);
                if (pos < st->flag_cap) sg.flist[pos] = ts[q];
              }
              rank++;
              if (own) {
                leaf_start[ts[q]] = idx;
                if (fq[q] & S2_SPLIT) {
                  st->split_idx = idx; st->split_target = ts[q];
                  if (idx == 0 || idx + 1 >= sp.n) eflags |= EF_DEGENERATE_SPLIT;   // two_layer.rs:27
                }
              }
              if (idx >= c1) stop = true;
            }
          }
          if (!dense) bcnt += add;
          stop = __any(stop);
        }
        if (nonmono) eflags |= EF_NON_MONOTONE;                     // two_layer.rs:50 / :144
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
        (void)anyd;
      }
    };
    bool dense = false;
    if (interior) phase1(std::false_type{}, dense); else phase1(std::true_type{}, dense);
    rounds(dense);
  }
  if (eflags) atomicOr(&st->err_flags, eflags);
}

}  // namespace rmi
