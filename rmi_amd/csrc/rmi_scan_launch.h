// rmi_scan_launch.h -- host interface of pipeline 5 (rmi_scan.hip.h): the kernel is compiled in a translation unit of its own
// (rmi_scan.hip), rmi_hip.hip launches it through this function.
#pragma once
#include "rmi_lanes.hip.h"

namespace rmi {

struct ScanOut {
  unsigned long long* leaf_start;
  double* params;                 // null: not written (the host derives them from the rows)
  unsigned long long* leaf_err;   // null: likewise
  unsigned long long* leaf_count; // null: likewise
  unsigned char* rows;
  StatsPartial* partials;         // one record per wave
};

// A long stretch of empty leaves [g0, gt) in front of the leaf start at key index gs (or behind the last key): one wave would write them
// 64 at a time -- 288 000 leaves behind the last key of the C5 key set took one wave 1.5 ms --, so stretches of SCAN_GAP_MIN leaves
// or more are listed here and written by k_scan_gaps, all of the device at once.
struct GapRec { unsigned int g0, gt, gs, pad; };
constexpr unsigned int SCAN_GAP_MIN = 512, SCAN_GAP_CAP = 4096, SCAN_GAP_BLOCKS = 128;

struct ScanLaunch {
  const void* keys;               // pre-offset: keys[global index]
  Span sp;
  RootP rp;
  DevState* st;
  SgList fl;
  unsigned int long_min;
  ScanOut out;
  PeerRows peers;
  GapRec* gaps;                   // SCAN_GAP_CAP records
  unsigned long long* gap_cnt;    // their counter (zero before the launch)
  unsigned int* tile_list;        // the tiles the short form's kernel leaves to the general form's (room for every tile of the launch: rmi_scan_tiles)
  unsigned long long* tile_cnt;   // their counter (zero before the launch)
  unsigned int n_cu;              // compute units of the device
  unsigned int listed_hint;       // tiles the short form left to the general form in the last training of this configuration (~0: unknown)
  int long_leaves;                // an earlier training of this configuration left hundreds of tiles: long leaves among short ones (a skewed key set) -- the long-leaf instance
  int host_split;                 // the split of the 2-way join is in *st already (a shard)
  int mono;                       // the root's targets are monotone in the key by arithmetic: a linear root with finite coefficients and a slope >= 0, a radix
                                  // root whose prefix is common to all resident keys (then equal targets at two keys prove that no leaf starts between them)
  unsigned int max_waves;         // persistent waves to launch at most (0: what the device holds)
  unsigned int waves;             // out: waves launched = aggregate records written
};
constexpr unsigned int SCAN_MAX_WAVES = 8192;    // both kernels of a training together

// root: K_LINEAR, K_CUBIC, K_RADIX, K_RADIX_TABLE, K_LOGLINEAR, K_NORMAL; dtype: RMI_KEY_*.  Returns 0, or -1 for a combination
// that is not compiled.
int rmi_scan_launch(int root, int dtype, ScanLaunch& a, hipStream_t s);
// k_scan_gaps behind it on the same stream: the listed stretches of empty leaves; its SCAN_GAP_BLOCKS aggregate records go behind the
// waves' (a.out.partials[a.waves ...]), a.waves counts them in afterwards
int rmi_scan_gaps_launch(int dtype, ScanLaunch& a, hipStream_t s);
// persistent waves a CU holds of this build of the kernel (4 SIMDs x the waves per SIMD it is compiled for, the LDS allowing): phase 0 = the
// short form's kernel, 1 = the general form's
unsigned int rmi_scan_waves_per_cu(int phase);
// an upper bound of the tiles of a launch over n_it keys of `dtype` (the capacity of ScanLaunch::tile_list)
unsigned long long rmi_scan_tiles(int dtype, unsigned long long n_it);

}  // namespace rmi
