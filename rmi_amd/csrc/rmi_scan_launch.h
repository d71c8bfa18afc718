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

struct ScanLaunch {
  const void* keys;               // pre-offset: keys[global index]
  Span sp;
  RootP rp;
  DevState* st;
  SgList fl;
  unsigned int long_min;
  ScanOut out;
  PeerRows peers;
  int host_split;                 // the split of the 2-way join is in *st already (a shard)
  unsigned int max_waves;         // persistent waves the device holds
  unsigned int waves;             // out: waves launched = aggregate records written
};
constexpr unsigned int SCAN_MAX_WAVES = 4096;

// root: K_LINEAR, K_CUBIC, K_RADIX, K_RADIX_TABLE, K_LOGLINEAR, K_NORMAL; dtype: RMI_KEY_*.  Returns 0, or -1 for a combination
// that is not compiled.
int rmi_scan_launch(int root, int dtype, ScanLaunch& a, hipStream_t s);

}  // namespace rmi
