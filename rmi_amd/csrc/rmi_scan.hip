// rmi_scan.hip -- translation unit of pipeline 5 (k_spline_scan, rmi_scan.hip.h) and its launcher.
#include <hip/hip_runtime.h>


#include "rmi_scan.hip.h"

namespace rmi {

template <int ROOT, typename K, int V>
static int scan_launch_t(ScanLaunch& a, hipStream_t s) {
  using G = ScGeom<K, V>;
  const K* keys = (const K*)a.keys;
  // tiles start at 128-byte lines of the key array (the 16-byte chunks of a tile are aligned by ADDRESS)
  const uint64_t align = 128 / sizeof(K);
  const uint64_t mis = (uint64_t)((reinterpret_cast<uintptr_t>(keys + a.sp.it_lo) / sizeof(K)) % align);
  const long long tile0 = (long long)a.sp.it_lo - (long long)mis;
  const uint64_t rel_hi = a.sp.it_hi - (uint64_t)tile0;
  const unsigned int ntiles = (unsigned int)(rel_hi / G::BTILE + 1);            // big tiles (the position behind the last key lies in one)
  const unsigned int tpx = (ntiles + 7u) / 8u;
  auto grid_for = [&](int phase) -> unsigned int {
    unsigned int cap = rmi_scan_waves_per_cu(phase) * (a.n_cu ? a.n_cu : 256u);
    if (a.max_waves && a.max_waves < cap) cap = a.max_waves;
    unsigned int grid = (ntiles + 7u) & ~7u;
    if (grid > cap) grid = cap & ~7u;
    if (grid > SCAN_MAX_WAVES / 2) grid = SCAN_MAX_WAVES / 2;
    if (grid < 8u) grid = 8u;
    return grid;
  };
  ScanArgs ka;
  ka.keys = a.keys; ka.tile0 = tile0; ka.ntiles = ntiles; ka.tiles_per_xcd = tpx; ka.long_min = a.long_min; ka.host_split = a.host_split; ka.mono = a.mono;
  ka.st = a.st; ka.sp = a.sp; ka.r = a.rp; ka.out = a.out; ka.fl = a.fl; ka.peers = a.peers; ka.gaps = a.gaps; ka.gap_cnt = a.gap_cnt;
  ka.tile_list = a.tile_list; ka.tile_cnt = a.tile_cnt;
  a.waves = 0;
  bool listed = false;
  if constexpr (ScMono<ROOT>::value && RMI_SC_FAST) {
    // leaves shorter than a lane's row on average (8-byte keys: 16 keys, 4-byte keys: 32; the limit: 1.25 rows): nearly every tile has a lane with two leaf
    // starts, the short form would list them all -- one counter -- and the general form take them from the list: 3.7 ms for 200 M u64 keys in 2^24 leaves,
    // 4.1 for 400 M u32 keys, against 1.3 with the general form over all tiles at once (what roots that are not monotone by arithmetic get)
    const uint64_t n_keys = a.sp.it_hi - a.sp.it_lo, n_leaves = a.sp.leaf_hi > a.sp.leaf_lo ? a.sp.leaf_hi - a.sp.leaf_lo : 1;
    const bool short_leaves = 4ull * n_keys < 5ull * (uint64_t)G::VF * n_leaves;
    if (a.mono && a.tile_list != nullptr && !short_leaves) {
      // the short form's kernel over all tiles; what it leaves goes on the list
      const unsigned int g0 = grid_for(0);
      // leaves longer than the look-ahead on average (8-byte keys: 64 keys, 4-byte keys: 128): the variant that looks for an open leaf's end behind it
      // (... and where they are several hundred keys long, the variant that finds that end with one gather and reads the open leaf's far keys eight blocks a trip)
      // (... or where an earlier training of the configuration listed hundreds of tiles -- books-shaped keys in 2^20 leaves: 4 700 leaves run on for more than
      //  long_min keys, each a search of 64 round trips before its tile is listed, 1.9 ms: the long-leaf instance keeps them)
      if (n_keys > 384ull * n_leaves || a.long_leaves) {
        // Leaves of several tiles: a leaf start every P tiles, and the wave that meets one reads the whole leaf.  A wave of an XCD takes every wpx-th tile
        // (wpx = the waves of the XCD): if wpx / P is a fraction of a small denominator q, every start meets the same wpx q / P waves -- 400 M u32 keys
        // (a jittered grid of stride 10) under a radix root of 2^14 leaves: P = 2^18 / 10 / 2 048 = 12.8 tiles, wpx = 384 = 30 P, 30 waves of 384 did all
        // the work (6.7 ms against 0.77).  The shapes that do this are made of powers of two, like 384 = 3 * 2^7: wpx is the largest PRIME the device
        // holds (383).  P itself is not known here -- the leaves a radix root leaves empty are not.  (Measured instead: a wave's place rotating from
        // round to round 0.81 ms, but 0.74 -> 0.97 for the shapes without such a fraction; tiles drawn from a counter per XCD, four a ticket, 0.96, and
        // 0.55 -> 0.77: a fixed stride deals evenly filled leaves out perfectly, any other rule like a Poisson process, and the tickets cost.)
        unsigned int gf = g0;
        if (g0 >= 8u * 32u) {
          unsigned int w = g0 / 8u;
          auto prime = [](unsigned int v) { for (unsigned int d = 2; d * d <= v; d++) if (v % d == 0) return false; return true; };
          while (!prime(w)) w--;
          gf = 8u * w;
        }
        hipLaunchKernelGGL((k_spline_scan<ROOT, K, V, 0, 2>), dim3(gf), dim3(64), 0, s, ka);
        a.waves += gf;
        ka.out.partials = a.out.partials + gf;
        listed = true;
      } else {
      if (n_keys > (uint64_t)G::EXTN * n_leaves) hipLaunchKernelGGL((k_spline_scan<ROOT, K, V, 0, 1>), dim3(g0), dim3(64), 0, s, ka);
      else hipLaunchKernelGGL((k_spline_scan<ROOT, K, V, 0, 0>), dim3(g0), dim3(64), 0, s, ka);
      a.waves += g0;
      ka.out.partials = a.out.partials + g0;
      listed = true;
      }
    }
  }
  if (!listed) ka.tile_list = nullptr;                                            // the general form's kernel takes every tile
  unsigned int g1 = grid_for(1);
  if (listed && a.listed_hint != ~0u) {                                           // (a hint for the launch's size only: the kernel strides over whatever the list holds)
    const unsigned long long want = 2ull * a.listed_hint + 64ull;
    if (want < (unsigned long long)g1) g1 = (unsigned int)((want + 7ull) & ~7ull);
  }
  hipLaunchKernelGGL((k_spline_scan<ROOT, K, V, 1>), dim3(g1), dim3(64), 0, s, ka);
  a.waves += g1;
  return 0;
}

template <int ROOT>
static int scan_launch_k(int dtype, ScanLaunch& a, hipStream_t s) {
  switch (dtype) {
    case 0: return scan_launch_t<ROOT, uint64_t, 16>(a, s);
    case 1: return scan_launch_t<ROOT, uint32_t, 32>(a, s);
    case 2: return scan_launch_t<ROOT, double, 16>(a, s);
  }
  return -1;
}

// (what the registers and the LDS of a CU hold: 4 SIMDs x the waves per SIMD the phase is compiled for, 160 KB over the kernel's static LDS;
//  the same for every instance.  A launch of more waves than are resident would run its surplus as a second round behind the first.)
unsigned int rmi_scan_waves_per_cu(int phase) {
  const unsigned int lds_bytes = (((unsigned int)ScGeom<uint32_t, 32>::LDS_DW * 4u + 2864u) + 1279u) / 1280u * 1280u;   // (the tile image + the slot tables, in the allocation's granules of 1 280 B)
  const unsigned int by_lds = 163840u / lds_bytes;
  const unsigned int by_regs = 4u * (phase == 0 ? RMI_SC_WPE0 : RMI_SC_WPE);
  return by_lds < by_regs ? by_lds : by_regs;
}
unsigned long long rmi_scan_tiles(int dtype, unsigned long long n_it) {
  const unsigned long long tile = dtype == 1 ? ScGeom<uint32_t, 32>::BTILE : ScGeom<uint64_t, 16>::BTILE;
  return n_it / tile + 8;
}

int rmi_scan_gaps_launch(int dtype, ScanLaunch& a, hipStream_t s) {
  StatsPartial* const part = a.out.partials + a.waves;
  switch (dtype) {
    case 0: hipLaunchKernelGGL((k_scan_gaps<uint64_t>), dim3(SCAN_GAP_BLOCKS), dim3(256), 0, s, (const uint64_t*)a.keys, a.sp, a.rp.L, a.out, a.peers, (const GapRec*)a.gaps, (const unsigned long long*)a.gap_cnt, part); break;
    case 1: hipLaunchKernelGGL((k_scan_gaps<uint32_t>), dim3(SCAN_GAP_BLOCKS), dim3(256), 0, s, (const uint32_t*)a.keys, a.sp, a.rp.L, a.out, a.peers, (const GapRec*)a.gaps, (const unsigned long long*)a.gap_cnt, part); break;
    case 2: hipLaunchKernelGGL((k_scan_gaps<double>), dim3(SCAN_GAP_BLOCKS), dim3(256), 0, s, (const double*)a.keys, a.sp, a.rp.L, a.out, a.peers, (const GapRec*)a.gaps, (const unsigned long long*)a.gap_cnt, part); break;
    default: return -1;
  }
  a.waves += SCAN_GAP_BLOCKS;
  return 0;
}

int rmi_scan_launch(int root, int dtype, ScanLaunch& a, hipStream_t s) {
  switch (root) {
    case K_LINEAR: return scan_launch_k<K_LINEAR>(dtype, a, s);
    case K_RADIX: return scan_launch_k<K_RADIX>(dtype, a, s);
#ifndef RMI_SC_DEV                    // (development builds: two roots compile in a third of the time)
    case K_CUBIC: return scan_launch_k<K_CUBIC>(dtype, a, s);
    case K_RADIX_TABLE: return scan_launch_k<K_RADIX_TABLE>(dtype, a, s);
    case K_LOGLINEAR: return scan_launch_k<K_LOGLINEAR>(dtype, a, s);
    case K_NORMAL: return scan_launch_k<K_NORMAL>(dtype, a, s);
#endif
  }
  return -1;
}

}  // namespace rmi
