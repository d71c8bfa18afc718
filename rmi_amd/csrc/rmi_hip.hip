// rmi_hip.hip -- C ABI (include/rmi_hip.h) over the gfx950 kernels.  Host orchestration only:
// every per-key / per-leaf computation of the hot path runs in the HIP kernels of
// rmi_kernels.hip.h.  There is no CPU fallback: without a HIP device every compute entry point
// returns RMI_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rmi_hip.h"
#include "rmi_kernels.hip.h"
#include "rmi_stream.hip.h"
#include "rmi_sigma.hip.h"
#include "rmi_lanes.hip.h"
#include "rmi_regs.hip.h"
#include "rmi_scan_launch.h"
#include "rmi_root_host.h"

using namespace rmi;

struct rmi_hip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // keys
  const void* d_keys = nullptr;
  void* d_keys_owned = nullptr;
  uint64_t n = 0;
  int dtype = RMI_KEY_U64;
  // outputs (capacity in leaves)
  uint64_t cap_leaves = 0;
  int cap_ppl = 0;
  unsigned long long* d_leaf_start = nullptr;   // L+1
  double* d_params = nullptr;                   // L*ppl
  unsigned long long* d_maxerr = nullptr;       // L
  unsigned long long* d_run = nullptr;          // L
  unsigned long long* d_err = nullptr;          // L
  unsigned long long* d_count = nullptr;        // L
  unsigned char* d_rows = nullptr;              // L*(ppl*8+8)
  unsigned long long* d_tilemin = nullptr;
  // radix-table root (radix8/18/22/26/28): hint_table of the last rmi_hip_fit_root / rmi_hip_set_root_table
  std::vector<uint32_t> h_table;
  uint32_t* d_table = nullptr;
  uint64_t d_table_cap = 0;
  std::vector<uint64_t> h_spline;               // last rmi_hip_cache_fix result, (key, offset) pairs
  double* d_cube = nullptr;                     // cubic leaves: span, then pow(span, 3.0), per leaf
  uint64_t cube_cap = 0;
  std::vector<double> h_cube;
  unsigned long long* d_long = nullptr;         // long-leaf hand-over list (pass A -> k_fit_long)
  uint64_t long_cap = 0;
  DevState* d_state = nullptr;
  StatsPartial* d_partials = nullptr;           // per-block aggregate records of k_finalize
  DevState* h_state = nullptr;                  // pinned
  unsigned long long* h_sentinel = nullptr;     // pinned
  // shard of a multi-GPU run (rmi_hip_set_shard); keys resident = global [rd_lo, rd_hi)
  bool have_shard = false;
  Span shard = {};
  unsigned long long shard_split_idx = ~0ull, shard_split_target = 0;
  void* d_rows_ext = nullptr;                   // caller-provided row buffer (e.g. the all-gather buffer)
  const void* last_rows = nullptr;              // where the LAST training wrote its rows (d_rows, or the caller's buffer at that time)
  // streamed training (rmi_hip_train_streamed): the key buffer and the output arrays are those of the WHOLE key set, a
  // launch works on the shard in c->shard; its aggregates go to slot stream_slot of the pinned state array
  bool stream_mode = false;
  int stream_slot = 0;
  void* h_stage[2] = {nullptr, nullptr};        // pinned staging buffers of the chunked upload
  hipStream_t copy_stream = nullptr;
  bool opt_tail = true;                         // k_lane_reduce publishes the result behind k_leaf_lanes; the list kernels run behind the synchronisation, and only if a leaf was handed over
  bool tail_armed = false;
  std::function<int()> tail_fn;                 // the list kernels + k_finalize_listed of the last launch (listed_epilogue)
  std::function<int(unsigned int)> regs_listed_fn;          // pipeline 4: k_leaf_lanes_listed + k_lane_reduce once more, run behind the synchronisation and only if k_leaf_regs listed a group
  std::function<void()> refinalize_fn;          // one-pass modes: k_finalize + k_stats_reduce once more, behind the host fit of giant leaves
  unsigned int* d_tickets = nullptr;            // arrival counter of k_lane_reduce's blocks
  int peer_fuse_n = 0;                          // direct exchange, <= 8 ranks: the peers' tables of the running epoch (k_leaf_lanes stores its rows there too)
  unsigned char* peer_fuse_tab[7] = {};
  bool rows_pushed = false;                     // ... and whether the last launch did
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  hipEvent_t ev[10] = {};
  int profile_level = 0;                        // -1: no events at all (device_ns = 0); 0: whole call only; 1: + the first (dominant) kernel; 2: every kernel group
  DevState* h_state_dev = nullptr;              // device address of the pinned h_state (written by the last kernel)
  int pipeline = 3;                             // 1 = one kernel per reference pass; 2 = streaming passes A/B; 3 = leaf-lane kernels (rmi_lanes.hip.h)
  double* d_lntab = nullptr;                    // RN(1 / k) for the running count of the leaf-lane walk (k_lane_table)
  bool lanes_fuse = true;                       // error pass fused behind the fit in k_leaf_lanes (else k_err_range)
  // pipeline 4 (rmi_regs.hip.h): k_leaf_regs -- one read of the keys, a leaf's keys stay in registers between its fit and its error pass --
  // in place of k_leaf_lanes where the leaves are short enough on average; the groups it does not take go through k_leaf_lanes_listed
  bool regs = true;                             // RMI_HIP_REGS=0: k_leaf_lanes for everything
  int regs_u32 = 2;                             // 4-byte keys through k_leaf_regs: 2 = two waves per SIMD with the raw keys stashed (k_leaf_regs<K, 2>), 1 = the
                                                // one-wave kernel in half-line panels, 0 = k_leaf_lanes (RMI_HIP_REGS_U32)
  bool regs_forced = false;                     // RMI_HIP_REGS=1: k_leaf_regs wherever it applies, whatever the number of groups
  unsigned int regs_grid = 0;                   // persistent waves of k_leaf_regs (0: 4 per CU)
  unsigned int regs_max_avg = 208;              // average keys per leaf above which most groups would not fit (RG_MAXPTS = 240 per container)
  unsigned int regs_long_max_avg = 640;         // ... and up to which k_leaf_regs<K, LONG> takes them (the steps behind the stash through the ring twice); above: k_leaf_lanes
  static constexpr unsigned int regs_slow = 0;  // (debugging aid of round 4: every group on the list)
  bool regs_backoff = true;                     // RMI_HIP_REGS_BACKOFF=0: k_leaf_regs also for key sets on which it listed most groups last time
  uint64_t regs_off_epoch = 0; uint64_t regs_off_L[8] = {}; int regs_off_n = 0;   // ... the (key set, leaves) pairs remembered
  static constexpr bool regs_queue = false;     // (groups dealt to the waves from a counter instead of by wave number: measured 473 against 460 us; the switch is gone)
  double* d_regtab = nullptr;                   // the interleaved step table of k_leaf_regs
  unsigned long long* d_regprof = nullptr;      // RG_PROF builds: cycles per phase, summed over the waves
  std::vector<rmi_hip_ctx*> many_views;         // rmi_hip_train_many: contexts that borrow this one's keys (kept for the next call)
  void* d_bnext = nullptr;                      // the keys on either side of every leaf, for k_regs_finalize: [2][64 groups]
  unsigned char* d_tile_slow = nullptr;         // per group: finished by k_leaf_lanes_listed
  unsigned int* d_slow_list = nullptr;          // groups of 64 leaves left to k_leaf_lanes_listed (counter: d_tickets[1])
  uint64_t slow_cap = 0;
  int n_cu = 256;
  bool last_regs = false;
  void* d_gaps = nullptr;                       // pipeline 5: the listed stretches of empty leaves (GapRec)
  uint64_t scan_hint_epoch = 0, scan_hint_L = 0; unsigned int scan_hint_n = ~0u;   // ... and how many it left the last time (key set, leaves per launch)
  uint64_t scan_skew_epoch = ~0ull, scan_skew_L = 0;   // (key set, leaves) whose short form listed hundreds of tiles: a skewed key set -- its next trainings take the long-leaf instance
  unsigned int* d_tile_list = nullptr;          // ... the tiles the short form's kernel leaves to the general form's
  uint64_t tile_list_cap = 0;
  // pipeline 5 writes the rows (codegen.rs:288-315: alpha, beta, error -- the 24 L bytes of SURVEY 8d) and the bucket table only; the separate
  // coefficient / error / count arrays hold the same values and are filled from them when somebody downloads one (k_lean_arrays)
  bool lean = true;                             // RMI_HIP_LEAN=0: the kernel writes all five arrays
  bool last_lean = false, lean_derived = false;
  unsigned long long lean_last_target = ~0ull, lean_leaf_lo = 0;   // (of the training the arrays belong to: the shard may be gone when they are asked for)
  bool last_scan = false;
  unsigned int scan_waves = 0;                  // its persistent waves per kernel at most (0: as many as the device holds, rmi_scan_waves_per_cu)
  bool lanes_search = true;                     // leaf boundaries by k_leaf_search where the root allows it (else the bucketing scan)
  uint64_t edge_epoch = 0;                      // first / last resident key of the key set `keys_epoch` (radix roots: is the prefix common?)
  uint64_t edge_first = 0, edge_last = 0;
  uint64_t edgef_epoch = 0;                     // ... as doubles (cubic roots: is the polynomial increasing between them?)
  double edge_first_f = 0.0, edge_last_f = 0.0;
  bool cubic_margin = true;                     // RMI_HIP_CUBIC_MARGIN=0: cubic roots always with the per-key verification (k_leaf_lanes, pipeline 3)
  double cubic_margin_scale = 1.0;              // testing: RMI_HIP_CUBIC_MARGIN_SCALE widens the margin (1e13: every leaf is verified key by key)
  bool last_lanes = false;
  // giant leaves (containers of more than host_min points): recorded by k_list, fitted on host cores after the device
  // pipeline, their error pass and finalize in a short epilogue (giant_epilogue).  Plain single-context trainings only.
  GiantLeaf* d_giant = nullptr;
  uint64_t giant_cap = 0;
  uint64_t host_min = 262144;                   // RMI_HIP_HOST_MIN; 0: never.  (A wave walks 262 144 points in ~7 ms, a host core in ~1: below that the
                                                //  leaves of a skewed key set are many and run side by side on the device)
  bool host_min_set = false;                    // RMI_HIP_HOST_MIN given: the threshold alone decides (else giant leaves only where the AVERAGE leaf is far below it)
  bool giant_armed = false;                     // the last launch recorded giant leaves for the host
  // the giant list early (k_giant_scan in front of k_list, read through pinned memory): the host walks the chains while
  // the device fits the other listed leaves
  static constexpr uint64_t GIANT_EARLY_MAX = 256;
  unsigned long long* h_giant = nullptr;        // pinned: [0] count, then GIANT_EARLY_MAX entries of 4 words
  hipEvent_t ev_giant = nullptr;
  bool giant_early = false;                     // the running tail carries the early list
  bool giant_fitted = false;                    // ... and the host has fitted it (giant_list / giant_ab)
  std::vector<GiantLeaf> giant_list;
  std::vector<double> giant_ab;
  struct { const void* keys; Span sp; uint64_t L; unsigned long long* leaf_start; double* params; unsigned long long* maxerr; unsigned long long* run;
           unsigned long long* err; unsigned long long* count; unsigned char* rows; uint64_t waves; } lp = {};
  uint64_t fit_threads = 131072;                // lanes of pass A (256 CUs x 8 waves x 64)
  uint64_t err_threads = 262144;                // lanes of pass B (4 waves/SIMD)
  int fit_min_chunk = 64;
  bool robust_leaf = false;                     // this call's leaves are robust_linear (fitted by k_fit_leaf; predict like linear)
  unsigned int long_min = 4096;                 // leaves with more points go to k_fit_long (>= FS_TMAX)
  // fit mode of linear leaves: 0 = exact (two streaming passes), 1 = one pass from sufficient statistics with the
  // guard (error integers bit-identical, flagged leaves re-fitted exactly), 2 = one pass, guard only counted
  int fit_mode = 0;
  double guard_k = 2.0;
  uint64_t sigma_waves = 4096;                  // k_sigma2: chunks the keys are cut into (one wave each)
  unsigned int sigma_min_leaf = 32;             // average keys per leaf below which the exact kernels are used
  unsigned int* d_flist = nullptr;              // leaves handed to the exact kernels: SG_REGIONS regions of flist_cap ids ...
  unsigned long long* d_flist_cnt = nullptr;    // ... and their counters
  void* d_recs = nullptr;                       // one-pass mode 2: partial sums of long leaves, [blocks][rpw] + the counts
  uint64_t recs_bytes = 0;
  unsigned long long* d_segs = nullptr;         // one-pass modes: the stretches of the long listed leaves for their error pass
  uint64_t segs_cap = 0;
  SgParams last_sg;                             // the parameters of the last k_sigma2 launch (k_list reads the records)
  // A key set on which the one-pass kernel hands most leaves to the exact list kernels (duplicate-heavy keys; keys
  // whose f64 images collapse, in the guarded mode) is served faster by the exact streaming passes: 1.1 ms against
  // 13.8 ms on 200 M duplicate-heavy keys.  Remembered per (key set, leaf count, mode); the next call takes the exact path.
  uint64_t keys_epoch = 1;                      // bumped whenever the context gets new keys
  uint64_t hint_epoch = 0;                      // the key set the remembered leaf counts belong to
  uint64_t hint_L[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (a parameter grid trains several leaf counts on one key set)
  int hint_n = 0, hint_mode = -1;
  void* d_bkeys = nullptr;                      // one-pass mode: key[e] and key[s-1] of every leaf (2 x leaves keys), see k_finalize
  uint64_t bkeys_cap = 0;
  uint64_t flist_cap = 0;                       // entries per region
  bool last_sigma = false;
  bool last_spline = false;                     // ... for linear_spline leaves (bit-identical in one pass)
  // last result
  uint64_t last_L = 0;
  int last_ppl = 2;
  uint64_t generation = 0;                      // train calls so far on this context
  std::thread upload_thread;                    // rmi_hip_upload_keys_async
  int upload_rc = RMI_OK;
  bool defer_sync = false;                      // rmi_hip_train_sharded: the caller queues the exchange, synchronises and finishes
  struct rmi_hip_multi* multi = nullptr;        // multi-GPU state (rmi_multi.inc.h)
  std::string err;
};

static int finish_train(rmi_hip_ctx* c, int leaf_kind, uint64_t num_leaves, rmi_hip_result* out);

// Is the cubic ((a x + b) x + c) x + d increasing on [xlo, xhi] as an exact polynomial?  Its derivative 3 a x^2 + 2 b x + c takes its minimum
// at an end of the interval or at the vertex -b / (3 a); evaluated in long double (64-bit mantissa: the coefficients are exact there) and
// required to exceed 1e-15 of the sum of its terms' magnitudes -- a thousand times long double's own rounding.  "No" (flat, decreasing
// somewhere, not finite) only costs the per-key verification.
static bool cubic_increasing_on(const RootP& rp, double xlo, double xhi) {
  if (!(std::isfinite(rp.p0) && std::isfinite(rp.p1) && std::isfinite(rp.p2) && std::isfinite(rp.p3) && std::isfinite(xlo) && std::isfinite(xhi)) || !(xlo <= xhi)) return false;
  const long double a = rp.p0, b = rp.p1, cc = rp.p2;
  auto fp = [&](long double x) -> long double { return (3.0L * a * x + 2.0L * b) * x + cc; };
  long double m = fp((long double)xlo) < fp((long double)xhi) ? fp((long double)xlo) : fp((long double)xhi);
  if (a != 0.0L) {
    const long double xv = -b / (3.0L * a);
    if (xv > (long double)xlo && xv < (long double)xhi && fp(xv) < m) m = fp(xv);
  }
  const long double xm = fabsl((long double)xlo) > fabsl((long double)xhi) ? fabsl((long double)xlo) : fabsl((long double)xhi);
  const long double scale = fabsl(3.0L * a) * xm * xm + fabsl(2.0L * b) * xm + fabsl(cc);
  return std::isfinite((double)m) && m > 1e-15L * scale;
}

static void set_err(rmi_hip_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (c) c->err = buf;
}

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return RMI_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

extern "C" {

int rmi_hip_abi_version(void) { return RMI_HIP_ABI_VERSION; }
int rmi_hip_last_pipeline(rmi_hip_ctx* c) { return !c ? RMI_ERR_BAD_ARG : (c->last_scan ? 5 : c->last_regs ? 4 : (c->last_lanes ? 3 : 2)); }

int rmi_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* rmi_hip_strerror(int code) {
  switch (code) {
    case RMI_OK: return "ok";
    case RMI_ERR_UNKNOWN_MODEL: return "unknown model type (train/mod.rs:53)";
    case RMI_ERR_RESTRICTION: return "model layer restriction violated (train/mod.rs:69-82)";
    case RMI_ERR_NON_MONOTONE: return "root model is not monotone on the data (two_layer.rs:50)";
    case RMI_ERR_DEGENERATE_SPLIT: return "degenerate split: a half of the 2-way join is empty (two_layer.rs:27)";
    case RMI_ERR_ROOT_OUT_OF_BOUNDS: return "root prediction out of bounds for a root without bounds check (two_layer.rs:45)";
    case RMI_ERR_BAD_ARG: return "bad argument";
    case RMI_ERR_NEGATIVE_VARIANCE: return "negative variance in SLR (linear.rs:48)";
    case RMI_ERR_ROBUST_TOO_SMALL: return "robust_linear needs more data (linear.rs:248)";
    case RMI_ERR_NUM_BITS: return "radix: num_bits assertion (utils.rs:18)";
    case RMI_ERR_CUBIC_DEGENERATE: return "cubic: no interior point (cubic_spline.rs:50/61)";
    case RMI_ERR_UNSUPPORTED_MODEL: return "model type is in the registry but not on the device path";
    case RMI_ERR_LAYERS: return "only two-layer RMIs are supported (train/mod.rs:125)";
    case RMI_ERR_NO_KEYS: return "no keys resident";
    case RMI_ERR_HIP: return "HIP runtime error";
    case RMI_ERR_NO_DEVICE: return "no HIP device";
    case RMI_ERR_NO_RCCL: return "RCCL (librccl.so) could not be loaded";
    case RMI_ERR_RCCL: return "RCCL error; see rmi_hip_last_error";
    default: return "unknown error";
  }
}

static const char* const kModelNames[] = {
    "linear", "linear_spline", "cubic", "radix", "robust_linear", "loglinear", "normal", "lognormal",
    "radix8", "radix18", "radix22", "radix26", "radix28", "bradix", "histogram"};
constexpr int kNumModels = sizeof(kModelNames) / sizeof(kModelNames[0]);

int rmi_hip_model_from_name(const char* name) {
  if (!name) return RMI_ERR_UNKNOWN_MODEL;
  for (int i = 0; i < kNumModels; i++)
    if (std::strcmp(name, kModelNames[i]) == 0) return i;
  return RMI_ERR_UNKNOWN_MODEL;
}

const char* rmi_hip_model_name(int kind) {
  if (kind < 0 || kind >= kNumModels) return nullptr;
  return kModelNames[kind];
}

// Roots the device path evaluates.  Not: lognormal (libm's ln per key, normal.rs:189-193: no device
// routine reproduces it bit for bit) and histogram (a binary search over L pivots per key).
static bool root_on_device_path(int kind) {
  switch (kind) {
    case RMI_MODEL_LOGNORMAL: case RMI_MODEL_HISTOGRAM: return false;
    default: return kind >= 0 && kind <= RMI_MODEL_HISTOGRAM;
  }
}

static bool must_be_top(int kind) {
  // radix.rs:75-80, balanced_radix.rs:167-169, histogram.rs:102 (MustBeTop).  RadixTable has no
  // restriction (radix.rs:163-165).
  switch (kind) {
    case RMI_MODEL_RADIX: case RMI_MODEL_BRADIX: case RMI_MODEL_HISTOGRAM:
      return true;
    default: return false;
  }
}

int rmi_hip_parse_spec(const char* spec, int* root_kind, int* leaf_kind) {
  if (!spec) return RMI_ERR_BAD_ARG;
  std::vector<std::string> parts;
  std::string cur;
  for (const char* p = spec;; p++) {
    if (*p == ',' || *p == 0) { parts.push_back(cur); cur.clear(); if (!*p) break; }
    else cur.push_back(*p);
  }
  std::vector<int> kinds;
  for (size_t i = 0; i < parts.size(); i++) {      // validate(): train/mod.rs:59-85
    int k = rmi_hip_model_from_name(parts[i].c_str());
    if (k < 0) return RMI_ERR_UNKNOWN_MODEL;
    if (must_be_top(k) && i != 0) return RMI_ERR_RESTRICTION;
    kinds.push_back(k);
  }
  if (kinds.size() != 2) return RMI_ERR_LAYERS;    // train/mod.rs:111-125
  if (root_kind) *root_kind = kinds[0];
  if (leaf_kind) *leaf_kind = kinds[1];
  return RMI_OK;
}

int rmi_hip_create(int device_id, rmi_hip_ctx** out) {
  if (!out) return RMI_ERR_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return RMI_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= ndev) return RMI_ERR_BAD_ARG;
  rmi_hip_ctx* c = new rmi_hip_ctx();
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete c; return RMI_ERR_HIP; }
  // (rmi_hip_destroy releases whatever has been created so far)
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  c->stream = c->own_stream;
  if (hipMalloc(&c->d_state, sizeof(DevState)) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  if (hipHostMalloc((void**)&c->h_state, sizeof(DevState) * RMI_STREAM_MAX_CHUNKS, hipHostMallocDefault) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->h_state_dev, c->h_state, 0) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  if (hipHostMalloc((void**)&c->h_sentinel, 64, hipHostMallocDefault) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  c->profile_level = 0;                                 // (rmi_hip_set_profile_level)
  const char* pl = std::getenv("RMI_HIP_PIPELINE");
  if (pl && *pl) c->pipeline = std::atoi(pl) >= 3 ? 3 : 2;       // 2: the streaming passes of round 2 (what tiny and huge key sets and cubic / robust leaves take anyway); default 3: everything newer
  { const char* lsr = std::getenv("RMI_HIP_LANES_SEARCH"); if (lsr && *lsr) c->lanes_search = std::atoi(lsr) != 0; }
  { const char* otl = std::getenv("RMI_HIP_OPT_TAIL"); if (otl && *otl) c->opt_tail = std::atoi(otl) != 0; }
  { const char* hm = std::getenv("RMI_HIP_HOST_MIN"); if (hm && *hm) { c->host_min = std::strtoull(hm, nullptr, 10); c->host_min_set = true; } }
  { const char* rg = std::getenv("RMI_HIP_REGS"); if (rg && *rg) { c->regs = std::atoi(rg) != 0; c->regs_forced = c->regs; } }   // (=1 also overrides the choice by group count)
  { const char* ru = std::getenv("RMI_HIP_REGS_U32"); if (ru && *ru) c->regs_u32 = std::atoi(ru); }
  { const char* cm = std::getenv("RMI_HIP_CUBIC_MARGIN"); if (cm && *cm) c->cubic_margin = std::atoi(cm) != 0; }
  { const char* ln = std::getenv("RMI_HIP_LEAN"); if (ln && *ln) c->lean = std::atoi(ln) != 0; }
  { const char* cm = std::getenv("RMI_HIP_CUBIC_MARGIN_SCALE"); if (cm && *cm) c->cubic_margin_scale = std::atof(cm); }
  { const char* sc = std::getenv("RMI_HIP_SCAN_WAVES"); if (sc && *sc) c->scan_waves = (unsigned int)std::atoi(sc); }
  { const char* rg = std::getenv("RMI_HIP_REGS_LONG_MAX_AVG"); if (rg && *rg) c->regs_long_max_avg = (unsigned int)std::atoi(rg); }
  { const char* rg = std::getenv("RMI_HIP_REGS_GRID"); if (rg && *rg) c->regs_grid = (unsigned int)std::atoi(rg); }
  { const char* rg = std::getenv("RMI_HIP_REGS_MAX_AVG"); if (rg && *rg) c->regs_max_avg = (unsigned int)std::atoi(rg); }
  { const char* rg = std::getenv("RMI_HIP_REGS_BACKOFF"); if (rg && *rg) c->regs_backoff = std::atoi(rg) != 0; }
  { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cu > 0) c->n_cu = cu; }
  if (hipMalloc(&c->d_lntab, sizeof(double) * (3 * LN_TMAX + 2 * (LS_SAMPLES + 1))) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  hipLaunchKernelGGL(k_lane_table, dim3((LN_TMAX + 255) / 256), dim3(256), 0, c->stream, c->d_lntab, LN_TMAX);
  if (hipMalloc(&c->d_regtab, sizeof(double) * 4 * RG_TMAX) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  if (RG_PROF) { if (hipMalloc(&c->d_regprof, 128) != hipSuccess || hipMemset(c->d_regprof, 0, 128) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; } }
  hipLaunchKernelGGL(k_regs_table, dim3((RG_TMAX + 255) / 256), dim3(256), 0, c->stream, c->d_regtab, RG_TMAX);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rmi_hip_destroy(c); return RMI_ERR_HIP; }
  // (the chunk geometry of pass A, pipeline 2: its tests vary it)
  const char* ft = std::getenv("RMI_HIP_FIT_THREADS");
  if (ft && *ft) c->fit_threads = std::strtoull(ft, nullptr, 10);
  const char* fc = std::getenv("RMI_HIP_FIT_MIN_CHUNK");
  if (fc && *fc) c->fit_min_chunk = std::atoi(fc);
  const char* lm = std::getenv("RMI_HIP_LONG_MIN");
  if (lm && *lm) { long v = std::atol(lm); if (v >= 64) c->long_min = (unsigned int)v; }
  *out = c;
  return RMI_OK;
}

static void free_outputs(rmi_hip_ctx* c) {
  (void)hipFree(c->d_leaf_start); (void)hipFree(c->d_params); (void)hipFree(c->d_maxerr); (void)hipFree(c->d_run);
  (void)hipFree(c->d_err); (void)hipFree(c->d_count); (void)hipFree(c->d_rows); (void)hipFree(c->d_tilemin);
  (void)hipFree(c->d_partials); c->d_partials = nullptr;
  (void)hipFree(c->d_tickets); c->d_tickets = nullptr;
  if (c->d_long) { (void)hipFree(c->d_long); c->d_long = nullptr; c->long_cap = 0; }
  if (c->d_cube) { (void)hipFree(c->d_cube); c->d_cube = nullptr; c->cube_cap = 0; }
  if (c->d_flist) { (void)hipFree(c->d_flist); c->d_flist = nullptr; c->flist_cap = 0; }
  if (c->d_flist_cnt) { (void)hipFree(c->d_flist_cnt); c->d_flist_cnt = nullptr; }
  if (c->d_gaps) { (void)hipFree(c->d_gaps); c->d_gaps = nullptr; }
  if (c->d_tile_list) { (void)hipFree(c->d_tile_list); c->d_tile_list = nullptr; c->tile_list_cap = 0; }
  if (c->d_bkeys) { (void)hipFree(c->d_bkeys); c->d_bkeys = nullptr; c->bkeys_cap = 0; }
  if (c->d_recs) { (void)hipFree(c->d_recs); c->d_recs = nullptr; c->recs_bytes = 0; }
  if (c->d_segs) { (void)hipFree(c->d_segs); c->d_segs = nullptr; c->segs_cap = 0; }
  c->d_leaf_start = nullptr; c->d_params = nullptr; c->d_maxerr = nullptr; c->d_run = nullptr;
  c->d_err = nullptr; c->d_count = nullptr; c->d_rows = nullptr; c->d_tilemin = nullptr;
  c->cap_leaves = 0; c->cap_ppl = 0;
}

static void free_multi(rmi_hip_ctx* c);
void rmi_hip_destroy(rmi_hip_ctx* c) {
  if (!c) return;
  for (rmi_hip_ctx* v : c->many_views) rmi_hip_destroy(v);
  c->many_views.clear();
  if (c->upload_thread.joinable()) c->upload_thread.join();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  free_multi(c);
  free_outputs(c);
  if (c->d_table) (void)hipFree(c->d_table);       // the root table is an input, not an output: it outlives re-sizing
  if (c->d_keys_owned) (void)hipFree(c->d_keys_owned);
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->d_lntab) (void)hipFree(c->d_lntab);
  if (c->d_regtab) (void)hipFree(c->d_regtab);
  if (c->d_regprof) {
    unsigned long long h[16] = {};
    if (hipMemcpy(h, c->d_regprof, 128, hipMemcpyDeviceToHost) == hipSuccess)
      std::fprintf(stderr, "k_leaf_regs cycles (sum over waves and trainings): fit %llu, epilogue+tail %llu, hand-over %llu, error pass %llu, finalize %llu; (unused %llu) hand-over requests %llu, (unused %llu), finalize stores %llu, finalize up to the shuffles' end %llu; RG_PROF 2, inside the fit: panel requests %llu, stash %llu, arithmetic %llu, constants + landed %llu, wait for the panel %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[5], h[7], h[8], h[9], h[11]);
    (void)hipFree(c->d_regprof);
  }
  if (c->d_slow_list) (void)hipFree(c->d_slow_list);
  if (c->d_tile_slow) (void)hipFree(c->d_tile_slow);
  if (c->d_bnext) (void)hipFree(c->d_bnext);
  if (c->d_giant) (void)hipFree(c->d_giant);
  if (c->h_state) (void)hipHostFree(c->h_state);
  if (c->h_sentinel) (void)hipHostFree(c->h_sentinel);
  for (int b = 0; b < 2; b++) { if (c->h_stage[b]) (void)hipHostFree(c->h_stage[b]); if (c->ev_stage[b]) (void)hipEventDestroy(c->ev_stage[b]); }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->h_giant) (void)hipHostFree(c->h_giant);
  if (c->ev_giant) (void)hipEventDestroy(c->ev_giant);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* rmi_hip_last_error(const rmi_hip_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rmi_hip_set_profile_level(rmi_hip_ctx* c, int level) {
  if (!c || level < -1 || level > 2) return RMI_ERR_BAD_ARG;
  c->profile_level = level;
  return RMI_OK;
}

int rmi_hip_set_fit_mode(rmi_hip_ctx* c, int mode, double guard_k) {
  if (!c || mode < 0 || mode > 2) return RMI_ERR_BAD_ARG;
  c->fit_mode = mode;
  if (guard_k > 0.0) c->guard_k = guard_k;
  c->hint_epoch = 0;                               // (what the last mode learned about the key set does not carry over)
  return RMI_OK;
}

int rmi_hip_set_stream(rmi_hip_ctx* c, void* s) {
  if (!c) return RMI_ERR_BAD_ARG;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return RMI_OK;
}

static size_t key_size(int dtype) { return dtype == RMI_KEY_U32 ? 4 : 8; }

int rmi_hip_upload_keys(rmi_hip_ctx* c, const void* host_keys, uint64_t n, int dtype) {
  if (!c || !host_keys || n == 0 || dtype < 0 || dtype > 2) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  c->d_keys = nullptr; c->n = 0;                               // (no stale pointer if anything below fails)
  if (c->d_keys_owned) { HIPCHK(c, hipFree(c->d_keys_owned)); c->d_keys_owned = nullptr; }
  HIPCHK(c, hipMalloc(&c->d_keys_owned, n * key_size(dtype)));
  HIPCHK(c, hipMemcpyAsync(c->d_keys_owned, host_keys, n * key_size(dtype), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->d_keys = c->d_keys_owned; c->n = n; c->dtype = dtype; c->keys_epoch++;
  return RMI_OK;
}

// The upload of a new key set next to host work on the same keys (the exact root fit of linear / robust_linear /
// cubic roots is a sequential host pass of ~0.8 s per 200 M keys; the 1.6 GB upload takes ~29 ms of it): the copy
// runs on a thread of the library, rmi_hip_upload_wait joins it.  Until then the context has no keys.
int rmi_hip_upload_keys_async(rmi_hip_ctx* c, const void* host_keys, uint64_t n, int dtype) {
  if (!c || !host_keys || n == 0 || dtype < 0 || dtype > 2) return RMI_ERR_BAD_ARG;
  if (c->upload_thread.joinable()) c->upload_thread.join();
  c->d_keys = nullptr; c->n = 0;
  c->upload_rc = RMI_OK;
  c->upload_thread = std::thread([c, host_keys, n, dtype]() { c->upload_rc = rmi_hip_upload_keys(c, host_keys, n, dtype); });
  return RMI_OK;
}
int rmi_hip_upload_wait(rmi_hip_ctx* c) {
  if (!c) return RMI_ERR_BAD_ARG;
  if (c->upload_thread.joinable()) c->upload_thread.join();
  return c->upload_rc;
}

int rmi_hip_attach_device_keys(rmi_hip_ctx* c, const void* device_keys, uint64_t n, int dtype) {
  if (!c || !device_keys || n == 0 || dtype < 0 || dtype > 2) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  c->d_keys = nullptr; c->n = 0;
  if (c->d_keys_owned) { HIPCHK(c, hipFree(c->d_keys_owned)); c->d_keys_owned = nullptr; }
  c->d_keys = device_keys; c->n = n; c->dtype = dtype; c->keys_epoch++;
  return RMI_OK;
}

uint64_t rmi_hip_num_keys(const rmi_hip_ctx* c) { return c ? c->n : 0; }

int rmi_hip_key_buffer(const rmi_hip_ctx* c, const void** device_keys, uint64_t* n, int* dtype) {
  if (!c || !device_keys || !n || !dtype) return RMI_ERR_BAD_ARG;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  *device_keys = c->d_keys; *n = c->n; *dtype = c->dtype;
  return RMI_OK;
}

// optimizer.rs:220-231 / src/main.rs:241-248 (`par_iter` over the configurations): `count` trainings on the context's resident keys,
// `in_flight` at a time, each on a context of its own that borrows the keys -- the threads are the library's.
int rmi_hip_train_many(rmi_hip_ctx* c, const rmi_hip_train_config* cfgs, uint64_t count, int in_flight, rmi_hip_result* results, int* rcs) {
  if (!c || (count && (!cfgs || !results))) return RMI_ERR_BAD_ARG;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  if (c->upload_thread.joinable()) { const int urc = rmi_hip_upload_wait(c); if (urc != RMI_OK) return urc; }
  // a training of the batch is a plain training of the WHOLE resident key set on a context of its own: a shard, a running streamed
  // training or a deferred synchronisation belong to one context and cannot be handed to the views
  if (c->have_shard || c->stream_mode || c->defer_sync) {
    set_err(c, "rmi_hip_train_many: the context holds a shard / a streamed or sharded training; train those one at a time");
    return RMI_ERR_BAD_ARG;
  }
  for (uint64_t i = 0; i < count; i++)
    if (cfgs[i].root.kind >= RMI_MODEL_RADIX8 && cfgs[i].root.kind <= RMI_MODEL_RADIX28 && !cfgs[i].root_table) {
      set_err(c, "rmi_hip_train_many: configuration %llu has a radix-table root without its table", (unsigned long long)i);
      return RMI_ERR_BAD_ARG;
    }
  if (in_flight < 1) in_flight = 1;
  if ((uint64_t)in_flight > count) in_flight = (int)(count ? count : 1);
  if (in_flight > 16) in_flight = 16;
  // the views: kept with the context (rmi_hip_release_views frees them); their keys are attached anew (the key set may have
  // changed since the last call)
  while ((int)c->many_views.size() < in_flight - 1) {
    rmi_hip_ctx* v = nullptr;
    const int crc = rmi_hip_create(c->device, &v);
    if (crc != RMI_OK) { set_err(c, "rmi_hip_train_many: a view could not be created (%s)", rmi_hip_strerror(crc)); return crc; }
    c->many_views.push_back(v);
  }
  for (int t = 0; t + 1 < in_flight; t++) {
    rmi_hip_ctx* v = c->many_views[t];
    const int arc = rmi_hip_attach_device_keys(v, c->d_keys, c->n, c->dtype);
    if (arc != RMI_OK) return arc;
    // what the caller has set on the context holds for every training of the batch
    v->fit_mode = c->fit_mode; v->guard_k = c->guard_k; v->profile_level = c->profile_level; v->host_min = c->host_min; v->host_min_set = c->host_min_set;
    v->long_min = c->long_min; v->pipeline = c->pipeline; v->regs = c->regs; v->regs_forced = c->regs_forced; v->regs_u32 = c->regs_u32; v->opt_tail = c->opt_tail;
  }
  std::atomic<uint64_t> next{0};
  std::vector<int> lrc(count, RMI_OK);
  std::vector<std::string> cerr((size_t)count);                        // a failing configuration's message; merged behind the join
  auto work = [&](rmi_hip_ctx* w, int slot) {
    (void)hipSetDevice(w->device);
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= count) return;
      std::memset(&results[i], 0, sizeof results[i]);
      lrc[i] = cfgs[i].root_table ? rmi_hip_set_root_table(w, cfgs[i].root_table, cfgs[i].root_table_entries) : RMI_OK;
      if (lrc[i] == RMI_OK) lrc[i] = rmi_hip_train_two_layer(w, &cfgs[i].root, cfgs[i].leaf_kind, cfgs[i].num_leaves, &results[i]);
      if (lrc[i] != RMI_OK) {
        char head[64]; snprintf(head, sizeof head, "configuration %llu: ", (unsigned long long)i);
        cerr[(size_t)i] = std::string(head) + w->err;
      }
    }
  };
  c->err.clear();
  std::vector<std::thread> th;
  for (int t = 0; t + 1 < in_flight; t++) th.emplace_back(work, c->many_views[t], t + 1);
  work(c, 0);
  for (auto& t : th) t.join();
  int first = RMI_OK;
  for (uint64_t i = 0; i < count; i++) {
    if (rcs) rcs[i] = lrc[i];
    if (first == RMI_OK && lrc[i] != RMI_OK) first = lrc[i];
  }
  if (first != RMI_OK) {                                               // (every failing configuration's message, "configuration <i>: ..." joined by "; ")
    std::string all;
    for (const auto& e : cerr) if (!e.empty()) { if (!all.empty()) all += "; "; all += e; }
    c->err = all;
  }
  return first;
}

// the contexts rmi_hip_train_many keeps between calls (each holds per-leaf buffers of the largest leaf count it trained)
int rmi_hip_release_views(rmi_hip_ctx* c) {
  if (!c) return RMI_ERR_BAD_ARG;
  for (rmi_hip_ctx* v : c->many_views) rmi_hip_destroy(v);
  c->many_views.clear();
  return RMI_OK;
}

int rmi_hip_set_shard(rmi_hip_ctx* c, const rmi_hip_shard* sh) {
  if (!c) return RMI_ERR_BAD_ARG;
  if (!sh) { c->have_shard = false; return RMI_OK; }
  if (!(sh->read_lo <= sh->key_lo && sh->key_lo <= sh->key_hi && sh->key_hi <= sh->read_hi && sh->read_hi <= sh->n_global &&
        sh->leaf_lo < sh->leaf_hi)) return RMI_ERR_BAD_ARG;
  if (sh->key_lo > 0 && sh->read_lo == sh->key_lo) return RMI_ERR_BAD_ARG;         // needs a left halo
  if (sh->key_hi < sh->n_global && sh->read_hi == sh->key_hi) return RMI_ERR_BAD_ARG;  // needs a right halo
  c->shard.it_lo = sh->key_lo; c->shard.it_hi = sh->key_hi;
  c->shard.rd_lo = sh->read_lo; c->shard.rd_hi = sh->read_hi;
  c->shard.n = sh->n_global; c->shard.leaf_lo = sh->leaf_lo; c->shard.leaf_hi = sh->leaf_hi;
  c->shard_split_idx = sh->split_idx; c->shard_split_target = sh->split_target;
  c->have_shard = true;
  return RMI_OK;
}

int rmi_hip_set_rows_output(rmi_hip_ctx* c, void* device_rows) {
  if (!c) return RMI_ERR_BAD_ARG;
  c->d_rows_ext = device_rows;
  return RMI_OK;
}

int rmi_hip_generate_keys(rmi_hip_ctx* c, int generator, int dtype, uint64_t n_global, uint64_t start,
                          uint64_t count, uint64_t seed) {
  if (!c || n_global == 0 || count == 0 || start + count > n_global) return RMI_ERR_BAD_ARG;
  if (generator < 0 || generator > 1 || (dtype != RMI_KEY_U64 && dtype != RMI_KEY_U32)) return RMI_ERR_BAD_ARG;
  const unsigned long long span = dtype == RMI_KEY_U64 ? 0xFFFFFFFFFFFFFFFEull : 0xFFFFFFFDull;  // 2^64-2 / 2^32-3
  const unsigned long long stride = span / n_global;
  if (stride == 0) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  c->d_keys = nullptr; c->n = 0;
  if (c->d_keys_owned) { HIPCHK(c, hipFree(c->d_keys_owned)); c->d_keys_owned = nullptr; }
  HIPCHK(c, hipMalloc(&c->d_keys_owned, count * key_size(dtype)));
  const unsigned long long base_seed = seed ? seed : (dtype == RMI_KEY_U64 ? 42ull : 46ull);
  const unsigned long long dup_seed = dtype == RMI_KEY_U64 ? 45ull : 47ull;
  const uint64_t want = (count + 255) / 256;
  const unsigned blocks = (unsigned)(want < (1u << 20) ? want : (1u << 20));   // grid-stride beyond 2^28 keys
  if (dtype == RMI_KEY_U64)
    hipLaunchKernelGGL((k_generate<uint64_t>), dim3(blocks), dim3(256), 0, c->stream, (uint64_t*)c->d_keys_owned, start, count, stride, base_seed, generator, dup_seed);
  else
    hipLaunchKernelGGL((k_generate<uint32_t>), dim3(blocks), dim3(256), 0, c->stream, (uint32_t*)c->d_keys_owned, start, count, stride, base_seed, generator, dup_seed);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->d_keys = c->d_keys_owned; c->n = count; c->dtype = dtype; c->keys_epoch++;
  return RMI_OK;
}

int rmi_hip_download_keys(rmi_hip_ctx* c, void* host_out) {
  if (!c || !host_out) return RMI_ERR_BAD_ARG;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(host_out, c->d_keys, c->n * key_size(c->dtype), hipMemcpyDeviceToHost));
  return RMI_OK;
}

const void* rmi_hip_device_keys(const rmi_hip_ctx* c) { return c ? c->d_keys : nullptr; }

int rmi_hip_selftest_div(rmi_hip_ctx* c, uint64_t trials, uint64_t seed, uint64_t* mismatches) {
  if (!c || !mismatches || trials == 0) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIPCHK(c, hipMalloc(&d, 8));
  HIPCHK(c, hipMemsetAsync(d, 0, 8, c->stream));
  const unsigned blocks = 2048, threads = 256;
  const unsigned long long per = (trials + (unsigned long long)blocks * threads - 1) / ((unsigned long long)blocks * threads);
  hipLaunchKernelGGL(k_selftest_div, dim3(blocks), dim3(threads), 0, c->stream, per, (unsigned long long)seed, d);
  unsigned long long h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  *mismatches = h;
  return RMI_OK;
}

// Host twin of rmi_hip_selftest_div for the reciprocal form of the root recurrence (rmi_root_host.h):
// fma(a, r, a*rl) against a / n on random and near-midpoint numerators, counts up to 2^40.
RMI_HOST_FMA static uint64_t host_div_mismatches(uint64_t trials, uint64_t seed) {
  uint64_t bad = 0, state = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (uint64_t it = 0; it < trials; it++) {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    uint64_t z = state ^ (state >> 29);
    z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    const uint64_t n = ((z >> 6) & 1ull) ? 1ull + ((state >> 11) % ((1ull << 40) - 1ull)) : 1ull + (z % 4096ull);
    const double nf = (double)n;
    const int ex = (int)((z >> 40) % 150ull) - 84;                     // 2^-84 .. 2^65, the operands of the recurrence
    const double frac = 1.0 + (double)((z >> 8) & 0xFFFFFFFFFFFFFull) * 0x1p-52;
    double a;
    const unsigned mode = (unsigned)(z >> 60) & 3u;
    if (mode == 0) a = std::ldexp(frac, ex);
    else {
      const double q = std::ldexp(frac, ex), half = std::ldexp(1.0, ex - 53);
      a = ((mode & 1u) ? q + half : q - half) * nf;                    // quotient next to a rounding midpoint
      if (mode == 3) a = std::nextafter(a, (z & 1ull) ? 1e300 : -1e300);
    }
    if ((z >> 7) & 1ull) a = -a;
    const double r = 1.0 / nf, rl = __builtin_fma(-nf, r, 1.0) * r;
    const double got = __builtin_fma(a, r, a * rl), want = a / nf;
    uint64_t gb, wb; std::memcpy(&gb, &got, 8); std::memcpy(&wb, &want, 8);
    if (gb != wb) bad++;
  }
  return bad;
}
int rmi_hip_selftest_host_div(uint64_t trials, uint64_t seed, uint64_t* mismatches) {
  if (!mismatches || trials == 0) return RMI_ERR_BAD_ARG;
  *mismatches = rmi_host::host_has_fma() ? host_div_mismatches(trials, seed) : 0;
  return RMI_OK;
}

int rmi_hip_selftest_recip(rmi_hip_ctx* c, uint64_t n_lo, uint64_t n_hi, uint64_t* mismatches) {
  if (!c || !mismatches || n_lo == 0 || n_hi <= n_lo || n_hi > (1ull << 40)) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIPCHK(c, hipMalloc(&d, 8));
  HIPCHK(c, hipMemsetAsync(d, 0, 8, c->stream));
  hipLaunchKernelGGL(k_selftest_recip, dim3(4096), dim3(256), 0, c->stream, (unsigned long long)n_lo, (unsigned long long)n_hi, d);
  unsigned long long h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  *mismatches = h;
  return RMI_OK;
}

int rmi_hip_measure_read_bandwidth(rmi_hip_ctx* c, int iters, double* gb_per_s) { return rmi_hip_measure_read_bandwidth_ex(c, iters, 0, gb_per_s); }
int rmi_hip_measure_read_bandwidth_ex(rmi_hip_ctx* c, int iters, int pattern, double* gb_per_s) {
  if (!c || !gb_per_s || iters <= 0 || pattern < 0 || pattern > 1) return RMI_ERR_BAD_ARG;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t bytes = c->n * key_size(c->dtype);
  const uint64_t n16 = bytes / 16;
  if (n16 == 0 || ((uintptr_t)c->d_keys & 15)) return RMI_ERR_BAD_ARG;
  auto go = [&]() {
    if (pattern == 1) hipLaunchKernelGGL(k_read_bw_chunk, dim3(256 * 2), dim3(256), 0, c->stream, (const uint4*)c->d_keys, n16, (unsigned int*)c->d_state);
    else hipLaunchKernelGGL(k_read_bw, dim3(256 * 8), dim3(256), 0, c->stream, (const uint4*)c->d_keys, n16, (unsigned int*)c->d_state);
  };
  go();
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  for (int i = 0; i < iters; i++) go();
  HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[9]));
  *gb_per_s = (double)(n16 * 16) * iters / ((double)ms * 1e-3) / 1e9;
  return RMI_OK;
}

}  // extern "C"

// Bucketing scan of the resident keys with a radix-family root over `bins` bins:
// d_first[j] = index of the first key whose bin is >= j (j in [0, bins]; d_first[bins] = n), on
// the stream of the context.  d_first: bins + 1 entries; d_tmin: bin_starts_tiles(bins) + 1.
static uint64_t bin_starts_tiles(uint64_t bins) { return (bins + 1 + FILL_TILE - 1) / FILL_TILE; }
template <typename K>
static void launch_bin_starts(rmi_hip_ctx* c, const RootP& rp, uint64_t bins, unsigned long long* d_first, unsigned long long* d_tmin) {
  hipStream_t s = c->stream;
  const uint64_t ntiles = bin_starts_tiles(bins);
  Span sp; sp.it_lo = 0; sp.it_hi = c->n; sp.rd_lo = 0; sp.rd_hi = c->n; sp.n = c->n; sp.leaf_lo = 0; sp.leaf_hi = bins;
  DevState init; std::memset(&init, 0, sizeof init);
  init.split_idx = c->n; init.last_target = ~0ull;
  (void)hipMemcpyAsync(c->d_state, &init, sizeof init, hipMemcpyHostToDevice, s);
  hipLaunchKernelGGL(k_table_init, dim3(1024), dim3(256), 0, s, d_first, bins);
  constexpr uint64_t V = 16 / sizeof(K);
  const uint64_t blocks = ((c->n + V - 1) / V + 256 * BV_UNROLL - 1) / (256 * BV_UNROLL);
  hipLaunchKernelGGL((k_bounds_vec<K_RADIX, K>), dim3((unsigned)blocks), dim3(256), 0, s, (const K*)c->d_keys, sp, rp, d_first, c->d_state);
  hipLaunchKernelGGL(k_fill_tilemin, dim3((unsigned)ntiles), dim3(256), 0, s, d_first, bins + 1, d_tmin);
  hipLaunchKernelGGL(k_fill_scan_tiles, dim3(1), dim3(1024), 0, s, d_tmin, ntiles);
  hipLaunchKernelGGL(k_fill_apply, dim3((unsigned)ntiles), dim3(256), 0, s, d_first, bins + 1, d_tmin);
}

// Radix-table root fitted where the keys are: bucketing scan over the table's slots, fill, scale.
template <typename K>
static int fit_radix_table_device(rmi_hip_ctx* c, int kind, uint64_t num_leaves, rmi_hip_model_params* out) {
  const uint64_t bits = (uint64_t)rmi_host::radix_table_bits(kind);
  const uint64_t slots = 1ull << bits;
  hipStream_t s = c->stream;
  HIPCHK(c, hipSetDevice(c->device));
  K k0{}, kl{};
  HIPCHK(c, hipMemcpy(&k0, c->d_keys, sizeof(K), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(&kl, (const K*)c->d_keys + (c->n - 1), sizeof(K), hipMemcpyDeviceToHost));
  const uint64_t x = rmi_host::as_uint(k0) ^ rmi_host::as_uint(kl);
  const uint64_t prefix = x == 0 ? 64 : (uint64_t)__builtin_clzll(x);          // common_prefix_size of sorted keys
  std::memset(out, 0, sizeof *out);
  out->kind = kind; out->ip[0] = prefix; out->ip[1] = bits;
  unsigned long long* d_first = nullptr;
  unsigned long long* d_tmin = nullptr;
  HIPCHK(c, hipMalloc(&d_first, (slots + 1) * 8));
  if (hipMalloc(&d_tmin, (bin_starts_tiles(slots) + 1) * 8) != hipSuccess) { (void)hipFree(d_first); return RMI_ERR_HIP; }
  if (c->d_table_cap < slots) {
    if (c->d_table) (void)hipFree(c->d_table);
    c->d_table = nullptr; c->d_table_cap = 0;
    if (hipMalloc(&c->d_table, slots * 4) != hipSuccess) { (void)hipFree(d_first); (void)hipFree(d_tmin); return RMI_ERR_HIP; }
    c->d_table_cap = slots;
  }
  int rc = RMI_OK;
  if (prefix == 64) {
    // every key is the same value x: the masked shifts of radix.rs:98-99 leave slot = x
    const uint64_t xv = rmi_host::as_uint(k0);
    if (xv >= slots) rc = RMI_ERR_BAD_ARG;                                     // assert!, radix.rs:101
    else {
      c->h_table.assign(slots, (uint32_t)slots);
      for (uint64_t i = 0; i <= xv; i++) c->h_table[i] = 0;
      if (hipMemcpy(c->d_table, c->h_table.data(), slots * 4, hipMemcpyHostToDevice) != hipSuccess) rc = RMI_ERR_HIP;
    }
  } else {
    RootP rp; rp.p0 = rp.p1 = rp.p2 = rp.p3 = 0.0; rp.prefix = (uint32_t)prefix; rp.L = slots; rp.table = nullptr;
    rp.cap = slots - 1; rp.oob_cap = slots - 1;
    // slot = ((x << p) >> p) >> max(0, 64 - p - bits)  ==  the `radix` function with min(bits, 64 - p) bits
    rp.bits = (uint32_t)(bits < 64 - prefix ? bits : 64 - prefix);
    launch_bin_starts<K>(c, rp, slots, d_first, d_tmin);
    const double scale = (double)num_leaves / (double)c->n;                    // two_layer.rs:109
    const int scaled = std::fabs(scale - 1.0) > DBL_EPSILON ? 1 : 0;           // map_scale!, models/mod.rs:238-250
    hipLaunchKernelGGL(k_table_from_starts, dim3(1024), dim3(256), 0, s, d_first, slots, scale, scaled, c->d_table);
    c->h_table.resize(slots);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(c->h_table.data(), c->d_table, slots * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) rc = RMI_ERR_HIP;
  }
  (void)hipFree(d_first); (void)hipFree(d_tmin);
  if (rc == RMI_ERR_HIP) c->err = "radix table fit on the device failed";
  if (rc != RMI_OK) c->h_table.clear();
  return rc;
}

// bradix root (balanced_radix.rs:40-101): the chi-square scores of its candidates come from bin
// counts; the bins of a monotone function over sorted keys are index ranges, so one bucketing scan
// per `high` candidate gives the counts exactly; the sums run on the host in the reference's order.
template <typename K>
static int fit_bradix_device(rmi_hip_ctx* c, uint64_t num_leaves, rmi_hip_model_params* out) {
  const uint64_t n = c->n;
  HIPCHK(c, hipSetDevice(c->device));
  std::memset(out, 0, sizeof *out);
  out->kind = RMI_MODEL_BRADIX;
  bool failed = false;
  auto get = [&](uint64_t i) { K k{}; if (hipMemcpy(&k, (const K*)c->d_keys + i, sizeof(K), hipMemcpyDeviceToHost) != hipSuccess) failed = true; return k; };
  const K k0 = get(0), kl = get(n - 1);
  uint64_t lo = 0, hi = n - 1;                                                 // first occurrence of the last key
  while (lo < hi && !failed) { const uint64_t mid = lo + (hi - lo) / 2; if (get(mid) < kl) lo = mid + 1; else hi = mid; }
  if (failed) { c->err = "hipMemcpy of a key failed"; return RMI_ERR_HIP; }
  const rmi_host::Data<K> d{nullptr, n, (double)num_leaves / (double)n};
  const uint64_t max_output = d.scale_y(lo);                                   // largest y over iter(), balanced_radix.rs:98
  if (rmi_host::num_bits(max_output) < 0) return RMI_ERR_NUM_BITS;
  const uint64_t x = rmi_host::as_uint(k0) ^ rmi_host::as_uint(kl);
  const int prefix = x == 0 ? 64 : __builtin_clzll(x);
  const uint64_t bins = max_output;
  unsigned long long* d_first = nullptr;
  unsigned long long* d_tmin = nullptr;
  HIPCHK(c, hipMalloc(&d_first, (bins + 1) * 8));
  if (hipMalloc(&d_tmin, (bin_starts_tiles(bins) + 1) * 8) != hipSuccess) { (void)hipFree(d_first); return RMI_ERR_HIP; }
  std::vector<unsigned long long> first(bins + 1);
  int rc = RMI_OK;
  auto high_counts = [&](int test_bits) {
    RootP rp; rp.p0 = rp.p1 = rp.p2 = rp.p3 = 0.0; rp.prefix = (uint32_t)prefix; rp.bits = (uint32_t)test_bits; rp.L = bins; rp.table = nullptr;
    rp.cap = max_output - 1; rp.oob_cap = ~0ull;
    launch_bin_starts<K>(c, rp, bins, d_first, d_tmin);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(first.data(), d_first, (bins + 1) * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) rc = RMI_ERR_HIP;
    for (auto& v : first) if (v > n) v = n;                                   // bins after the last key: no start
    // the tail duplicate of iter_model_input() (Q1) lands in the last key's bin: the last non-empty one
    uint64_t last_bin = bins - 1;
    while (last_bin > 0 && first[last_bin] >= n) last_bin--;
    return [&first, last_bin](uint64_t b) { return (uint64_t)(first[b + 1] - first[b]) + (b == last_bin ? 1 : 0); };
  };
  const int rc2 = rmi_host::bradix_choose(n, max_output, prefix, high_counts, out);
  (void)hipFree(d_first); (void)hipFree(d_tmin);
  if (rc == RMI_ERR_HIP) c->err = "bradix fit on the device failed";
  return rc != RMI_OK ? rc : rc2;
}

// Cubic root of HBM-resident keys: coefficients from O(log n) fetched keys (exact), the model choice
// from a device reduction when it is clear-cut (see k_cubic_root_sums), else *decided = false and the
// caller runs the reference's sequential pass.
template <typename K>
static int fit_cubic_device(rmi_hip_ctx* c, uint64_t num_leaves, rmi_hip_model_params* out, bool* decided) {
  *decided = false;
  const uint64_t n = c->n;
  HIPCHK(c, hipSetDevice(c->device));
  bool failed = false;
  auto get = [&](uint64_t i) { K k{}; if (hipMemcpy(&k, (const K*)c->d_keys + i, sizeof(K), hipMemcpyDeviceToHost) != hipSuccess) failed = true; return k; };
  const rmi_host::Data<K> d{nullptr, n, (double)num_leaves / (double)n};
  double cp[4];
  int rc = rmi_host::cubic_coeffs_get<K>(get, d, cp);
  if (failed) { c->err = "hipMemcpy of a key failed"; return RMI_ERR_HIP; }
  if (rc) return rc;
  rmi_hip_model_params lin;
  rc = rmi_host::fit_root_sparse<K>(RMI_MODEL_LINEAR_SPLINE, get, n, num_leaves, &lin);
  if (failed) { c->err = "hipMemcpy of a key failed"; return RMI_ERR_HIP; }
  if (rc) return rc;
  const double la = lin.p[0], lb = lin.p[1];
  CubicPartial* d_p = nullptr;
  HIPCHK(c, hipMalloc(&d_p, sizeof(CubicPartial) * RF_BLOCKS));
  const int scaled = std::fabs(d.scale - 1.0) > DBL_EPSILON ? 1 : 0;
  hipLaunchKernelGGL((k_cubic_root_sums<K>), dim3(RF_BLOCKS), dim3(256), 0, c->stream, (const K*)c->d_keys, n, d.scale, scaled,
                     cp[0], cp[1], cp[2], cp[3], la, lb, d_p);
  std::vector<CubicPartial> h(RF_BLOCKS);
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h.data(), d_p, sizeof(CubicPartial) * RF_BLOCKS, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipFree(d_p); c->err = "cubic root reduction failed"; return RMI_ERR_HIP; }
  (void)hipFree(d_p);
  long double eo = 0, el = 0;
  for (const CubicPartial& p : h) { eo += p.our_err; el += p.lin_err; }
  // Any summation order of n <= 2^32 non-negative f64 terms is within n * 2^-53 < 1e-6 relative of
  // the exact sum; decide only when the two sums are further apart than that, both ways.
  const long double gap = eo > el ? eo - el : el - eo;
  if (!(eo == eo) || !(el == el) || !(gap > 4e-6L * (eo + el))) return RMI_OK;   // (NaN / too close: undecided)
  std::memset(out, 0, sizeof *out);
  out->kind = RMI_MODEL_CUBIC;
  if (el < eo) { out->p[0] = 0.0; out->p[1] = 0.0; out->p[2] = lb; out->p[3] = la; }
  else std::memcpy(out->p, cp, sizeof cp);
  *decided = true;
  return RMI_OK;
}

// linear / robust_linear from parallel sums (opt-in fast mode)
template <typename K>
static int fit_linear_fast_device(rmi_hip_ctx* c, int kind, uint64_t num_leaves, rmi_hip_model_params* out) {
  const uint64_t n = c->n;
  std::memset(out, 0, sizeof *out);
  out->kind = kind;
  uint64_t lo = 0, hi = n;
  int tail = 1;
  if (kind == RMI_MODEL_ROBUST_LINEAR) {                       // linear.rs:239-260
    uint64_t bnd = rmi_host::sat_u64((double)n * 0.0001);
    if (bnd < 1) bnd = 1;
    if (!(bnd * 2 + 1 < n)) return RMI_ERR_ROBUST_TOO_SMALL;
    lo = bnd; hi = n - bnd; tail = 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  RootPartial* d_p = nullptr;
  HIPCHK(c, hipMalloc(&d_p, sizeof(RootPartial) * RF_BLOCKS));
  const double scale = (double)num_leaves / (double)n;
  const int scaled = std::fabs(scale - 1.0) > DBL_EPSILON ? 1 : 0;
  hipLaunchKernelGGL((k_root_sums<K>), dim3(RF_BLOCKS), dim3(256), 0, c->stream, (const K*)c->d_keys, n, lo, hi, scale, scaled, tail, d_p);
  std::vector<RootPartial> h(RF_BLOCKS);
  K k0{};
  int rc = RMI_OK;
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h.data(), d_p, sizeof(RootPartial) * RF_BLOCKS, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&k0, c->d_keys, sizeof(K), hipMemcpyDeviceToHost) != hipSuccess) rc = RMI_ERR_HIP;
  (void)hipFree(d_p);
  if (rc != RMI_OK) { c->err = "fast root fit failed"; return rc; }
  long double cn = 0, sx = 0, sy = 0, sxx = 0, sxy = 0;       // 1024 partials: combine in extended precision
  for (const RootPartial& p : h) { cn += p.n; sx += p.sx; sy += p.sy; sxx += p.sxx; sxy += p.sxy; }
  if (cn < 1.5L) { out->p[0] = (double)(cn > 0 ? sy : 0); out->p[1] = 0.0; return RMI_OK; }
  const long double mx = sx / cn, my = sy / cn;
  const long double m2 = sxx - sx * mx, cxy = sxy - sx * my;
  if (!(m2 > 0)) { out->p[0] = (double)my; out->p[1] = 0.0; return RMI_OK; }          // linear.rs:50-53
  const long double beta = cxy / m2;
  const long double x0 = (long double)rmi_host::as_float(k0);
  out->p[1] = (double)beta;
  out->p[0] = (double)(my - beta * (mx + x0));                                           // undo the shift by the first key
  return RMI_OK;
}

extern "C" {

int rmi_hip_fit_root_fast(rmi_hip_ctx* c, int root_kind, uint64_t num_leaves, rmi_hip_model_params* out) {
  if (!c || !out || num_leaves == 0) return RMI_ERR_BAD_ARG;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  if (root_kind != RMI_MODEL_LINEAR && root_kind != RMI_MODEL_ROBUST_LINEAR)
    return rmi_hip_fit_root(c, root_kind, num_leaves, nullptr, out);          // the other roots are exact and cheap, or (cubic) host work
  switch (c->dtype) {
    case RMI_KEY_U64: return fit_linear_fast_device<uint64_t>(c, root_kind, num_leaves, out);
    case RMI_KEY_U32: return fit_linear_fast_device<uint32_t>(c, root_kind, num_leaves, out);
    case RMI_KEY_F64: return fit_linear_fast_device<double>(c, root_kind, num_leaves, out);
  }
  return RMI_ERR_BAD_ARG;
}

// The host part of rmi_hip_fit_root on its own (no context, no device): the reference's fit over a host array.
int rmi_hip_fit_root_host(int root_kind, int dtype, const void* host_keys, uint64_t n, uint64_t num_leaves, rmi_hip_model_params* out) {
  if (!host_keys || !out || n == 0 || num_leaves == 0) return RMI_ERR_BAD_ARG;
  if (root_kind < 0 || root_kind >= kNumModels) return RMI_ERR_UNKNOWN_MODEL;
  if (!root_on_device_path(root_kind) || rmi_host::radix_table_bits(root_kind) > 0 || root_kind == RMI_MODEL_BRADIX) return RMI_ERR_UNSUPPORTED_MODEL;
  switch (dtype) {
    case RMI_KEY_U64: return rmi_host::fit_root<uint64_t>(root_kind, (const uint64_t*)host_keys, n, num_leaves, out);
    case RMI_KEY_U32: return rmi_host::fit_root<uint32_t>(root_kind, (const uint32_t*)host_keys, n, num_leaves, out);
    case RMI_KEY_F64: return rmi_host::fit_root<double>(root_kind, (const double*)host_keys, n, num_leaves, out);
  }
  return RMI_ERR_BAD_ARG;
}

int rmi_hip_fit_root(rmi_hip_ctx* c, int root_kind, uint64_t num_leaves, const void* host_keys,
                     rmi_hip_model_params* out) {
  if (!c || !out || num_leaves == 0) return RMI_ERR_BAD_ARG;
  if (root_kind < 0 || root_kind >= kNumModels) return RMI_ERR_UNKNOWN_MODEL;
  const bool is_table = rmi_host::radix_table_bits(root_kind) > 0;
  if (!root_on_device_path(root_kind)) return RMI_ERR_UNSUPPORTED_MODEL;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  if (root_kind == RMI_MODEL_BRADIX) {                         // bin counts of the resident keys: exact on the device
    switch (c->dtype) {
      case RMI_KEY_U64: return fit_bradix_device<uint64_t>(c, num_leaves, out);
      case RMI_KEY_U32: return fit_bradix_device<uint32_t>(c, num_leaves, out);
      case RMI_KEY_F64: return fit_bradix_device<double>(c, num_leaves, out);
    }
    return RMI_ERR_BAD_ARG;
  }
  if (is_table) {                                              // integer work on the resident keys: exact on the device
    switch (c->dtype) {
      case RMI_KEY_U64: return fit_radix_table_device<uint64_t>(c, root_kind, num_leaves, out);
      case RMI_KEY_U32: return fit_radix_table_device<uint32_t>(c, root_kind, num_leaves, out);
      case RMI_KEY_F64: return fit_radix_table_device<double>(c, root_kind, num_leaves, out);
    }
    return RMI_ERR_BAD_ARG;
  }
  if (root_kind == RMI_MODEL_CUBIC && c->n < (1ull << 32)) {    // exact and usually O(log n) host work + one device reduction
    bool decided = false;
    int rc = RMI_ERR_BAD_ARG;
    switch (c->dtype) {
      case RMI_KEY_U64: rc = fit_cubic_device<uint64_t>(c, num_leaves, out, &decided); break;
      case RMI_KEY_U32: rc = fit_cubic_device<uint32_t>(c, num_leaves, out, &decided); break;
      case RMI_KEY_F64: rc = fit_cubic_device<double>(c, num_leaves, out, &decided); break;
    }
    if (rc != RMI_OK || decided) return rc;
    // (too close to call from a parallel sum: the reference's sequential pass below)
  }
  std::vector<unsigned char> tmp;
  const void* hk = host_keys;
  if (!hk && (root_kind == RMI_MODEL_RADIX || root_kind == RMI_MODEL_LINEAR_SPLINE)) {
    // these two need a handful of keys only: fetch them from HBM one by one
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ks = key_size(c->dtype);
    const unsigned char* base = (const unsigned char*)c->d_keys;
    bool failed = false;
    auto fetch = [&](uint64_t i, void* dst) { if (hipMemcpy(dst, base + i * ks, ks, hipMemcpyDeviceToHost) != hipSuccess) failed = true; };
    int rc = RMI_ERR_BAD_ARG;
    switch (c->dtype) {
      case RMI_KEY_U64: rc = rmi_host::fit_root_sparse<uint64_t>(root_kind, [&](uint64_t i) { uint64_t k = 0; fetch(i, &k); return k; }, c->n, num_leaves, out); break;
      case RMI_KEY_U32: rc = rmi_host::fit_root_sparse<uint32_t>(root_kind, [&](uint64_t i) { uint32_t k = 0; fetch(i, &k); return k; }, c->n, num_leaves, out); break;
      case RMI_KEY_F64: rc = rmi_host::fit_root_sparse<double>(root_kind, [&](uint64_t i) { double k = 0; fetch(i, &k); return k; }, c->n, num_leaves, out); break;
    }
    if (failed) { c->err = "hipMemcpy of a key failed"; return RMI_ERR_HIP; }
    return rc;
  }
  if (!hk) {
    // (linear, robust_linear, cubic and the radix tables are passes over all keys)
    HIPCHK(c, hipSetDevice(c->device));
    tmp.resize(c->n * key_size(c->dtype));
    HIPCHK(c, hipMemcpy(tmp.data(), c->d_keys, tmp.size(), hipMemcpyDeviceToHost));
    hk = tmp.data();
  }
  switch (c->dtype) {
    case RMI_KEY_U64: return rmi_host::fit_root<uint64_t>(root_kind, (const uint64_t*)hk, c->n, num_leaves, out);
    case RMI_KEY_U32: return rmi_host::fit_root<uint32_t>(root_kind, (const uint32_t*)hk, c->n, num_leaves, out);
    case RMI_KEY_F64: return rmi_host::fit_root<double>(root_kind, (const double*)hk, c->n, num_leaves, out);
  }
  return RMI_ERR_BAD_ARG;
}

// The hint table of a radix-table root lives in the context (host copy + HBM copy): it is set by
// rmi_hip_fit_root, or by the caller (another process of a multi-GPU job, a cached root).
int rmi_hip_set_root_table(rmi_hip_ctx* c, const uint32_t* table, uint64_t entries) {
  if (!c || !table || entries == 0 || (entries & (entries - 1))) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->d_table_cap < entries) {
    if (c->d_table) (void)hipFree(c->d_table);
    c->d_table = nullptr; c->d_table_cap = 0;
    HIPCHK(c, hipMalloc(&c->d_table, entries * 4));
    c->d_table_cap = entries;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));                  // a train call may still be reading the old table
  if (table != c->h_table.data()) c->h_table.assign(table, table + entries);
  HIPCHK(c, hipMemcpy(c->d_table, c->h_table.data(), entries * 4, hipMemcpyHostToDevice));
  return RMI_OK;
}
int rmi_hip_cache_fix(rmi_hip_ctx* c, const uint64_t* host_keys, uint64_t n, uint64_t line_size, uint64_t* num_points) {
  if (!c || !host_keys || !num_points) return RMI_ERR_BAD_ARG;
  int rc = rmi_host::cache_fix(host_keys, n, line_size, c->h_spline);
  *num_points = c->h_spline.size() / 2;
  return rc;
}
int rmi_hip_download_cache_fix(const rmi_hip_ctx* c, uint64_t* out_pairs) {
  if (!c || !out_pairs || c->h_spline.empty()) return RMI_ERR_BAD_ARG;
  std::memcpy(out_pairs, c->h_spline.data(), c->h_spline.size() * 8);
  return RMI_OK;
}
int rmi_hip_root_table_entries(const rmi_hip_ctx* c, uint64_t* entries) {
  if (!c || !entries) return RMI_ERR_BAD_ARG;
  *entries = c->h_table.size();
  return RMI_OK;
}
int rmi_hip_download_root_table(const rmi_hip_ctx* c, uint32_t* out) {
  if (!c || !out || c->h_table.empty()) return RMI_ERR_BAD_ARG;
  std::memcpy(out, c->h_table.data(), c->h_table.size() * 4);
  return RMI_OK;
}

}  // extern "C"

extern "C" {

int rmi_hip_root_target(const rmi_hip_model_params* root, int dtype, uint64_t key_bits, uint64_t num_leaves, uint64_t* out) {
  if (!root || !out || num_leaves == 0) return RMI_ERR_BAD_ARG;
  if (rmi_host::radix_table_bits(root->kind) > 0 || !root_on_device_path(root->kind))
    return RMI_ERR_UNSUPPORTED_MODEL;                          // (the radix tables need the table: plan on the caller's side)
  switch (dtype) {
    case RMI_KEY_U64: *out = rmi_host::root_target<uint64_t>(*root, (uint64_t)key_bits, num_leaves); return RMI_OK;
    case RMI_KEY_U32: *out = rmi_host::root_target<uint32_t>(*root, (uint32_t)key_bits, num_leaves); return RMI_OK;
    case RMI_KEY_F64: { double d; std::memcpy(&d, &key_bits, 8); *out = rmi_host::root_target<double>(*root, d, num_leaves); return RMI_OK; }
  }
  return RMI_ERR_BAD_ARG;
}

struct rmi_hip_root_stream {
  int dtype;
  rmi_host::LinearRootStream<uint64_t> s64;
  rmi_host::LinearRootStream<uint32_t> s32;
  rmi_host::LinearRootStream<double> sf;
};

int rmi_hip_root_stream_begin(int root_kind, int dtype, uint64_t n_global, uint64_t num_leaves, rmi_hip_root_stream** out) {
  if (!out || n_global == 0 || num_leaves == 0 || dtype < 0 || dtype > 2) return RMI_ERR_BAD_ARG;
  if (root_kind != RMI_MODEL_LINEAR) return RMI_ERR_UNSUPPORTED_MODEL;
  rmi_hip_root_stream* r = new rmi_hip_root_stream();
  r->dtype = dtype;
  r->s64.begin(n_global, num_leaves); r->s32.begin(n_global, num_leaves); r->sf.begin(n_global, num_leaves);
  *out = r;
  return RMI_OK;
}
int rmi_hip_root_stream_push(rmi_hip_root_stream* r, const void* host_keys, uint64_t count) {
  if (!r || (!host_keys && count)) return RMI_ERR_BAD_ARG;
  switch (r->dtype) {
    case RMI_KEY_U64: r->s64.push((const uint64_t*)host_keys, count); break;
    case RMI_KEY_U32: r->s32.push((const uint32_t*)host_keys, count); break;
    default: r->sf.push((const double*)host_keys, count); break;
  }
  return RMI_OK;
}
int rmi_hip_root_stream_finish(rmi_hip_root_stream* r, rmi_hip_model_params* out) {
  if (!r || !out) return RMI_ERR_BAD_ARG;
  std::memset(out, 0, sizeof *out);
  out->kind = RMI_MODEL_LINEAR;
  int rc;
  switch (r->dtype) {
    case RMI_KEY_U64: rc = r->s64.finish(out); break;
    case RMI_KEY_U32: rc = r->s32.finish(out); break;
    default: rc = r->sf.finish(out); break;
  }
  delete r;
  return rc;
}

}  // extern "C"

// aggregate records of a training: one per wave of 64 leaves (k_leaf_lanes, k_leaf_regs) or one per persistent wave (k_spline_scan)
static inline uint64_t agg_records(uint64_t leaves) { const uint64_t w = (leaves + 63) / 64, m = (uint64_t)SCAN_MAX_WAVES + SCAN_GAP_BLOCKS; return w > m ? w : m; }
static inline uint64_t lane_slices(uint64_t leaves) { return (agg_records(leaves) + LF_SLICE - 1) / LF_SLICE; }
static int ensure_outputs(rmi_hip_ctx* c, uint64_t L, int ppl) {
  if (L <= c->cap_leaves && ppl <= c->cap_ppl) return RMI_OK;
  free_outputs(c);
  HIPCHK(c, hipMalloc(&c->d_leaf_start, (L + 1) * 8));
  HIPCHK(c, hipMalloc(&c->d_params, L * ppl * 8));
  HIPCHK(c, hipMalloc(&c->d_maxerr, L * 8));
  HIPCHK(c, hipMalloc(&c->d_run, L * 8));
  HIPCHK(c, hipMalloc(&c->d_err, L * 8));
  HIPCHK(c, hipMalloc(&c->d_count, L * 8));
  HIPCHK(c, hipMalloc(&c->d_rows, L * (ppl * 8 + 8)));
  HIPCHK(c, hipMalloc(&c->d_tilemin, ((L + 1 + FILL_TILE - 1) / FILL_TILE + 1) * 8));
  // (per block of k_finalize, or: per wave of k_leaf_lanes, the slice records of k_list_tail, the blocks of
  //  k_finalize_listed, then the slice records of k_leaf_lanes' own reduction)
  HIPCHK(c, hipMalloc(&c->d_partials, sizeof(StatsPartial) * (agg_records(L) + SG_REGIONS + 2 * FL_BLOCKS + 1 + lane_slices(L) + 1)));
  HIPCHK(c, hipMalloc(&c->d_tickets, 64));
  c->cap_leaves = L; c->cap_ppl = ppl;
  return RMI_OK;
}

// pow(x, 3.0) with the host's libm, a few threads
static void host_cubes(double* v, uint64_t n) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
  if (n < 65536) nt = 1;
  auto work = [v](uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; i++) v[i] = std::pow(v[i], 3.0); };
  if (nt == 1) { work(0, n); return; }
  std::vector<std::thread> th;
  const uint64_t per = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    const uint64_t a = (uint64_t)t * per, b = a + per < n ? a + per : n;
    if (a < b) th.emplace_back(work, a, b);
  }
  for (auto& x : th) x.join();
}

template <int ROOT, int LEAF, typename K>
static int launch_pipeline(rmi_hip_ctx* c, const RootP& rp, uint64_t L) {
  hipStream_t s = c->stream;
  constexpr int PPL = (LEAF == K_CUBIC) ? 4 : 2;
  constexpr int ROWB = PPL * 8 + 8;
  // An event between two kernels costs ~5.5 us of idle device time (measured in the kernel trace):
  // level 0 records the start and the end of the call only, level 1 also brackets the first,
  // dominant kernel, level 2 every kernel group.
  const int pl = c->profile_level;
  int evi = 0;
  auto mark = [&]() { if (pl >= 2 || (pl == 1 && evi == 0)) (void)hipEventRecord(c->ev[1 + evi], s); evi++; };

  // ---- index space of this launch (global indices; see Span) ----
  Span sp;
  if (c->have_shard) sp = c->shard;
  else { sp.it_lo = 0; sp.it_hi = c->n; sp.rd_lo = 0; sp.rd_hi = c->n; sp.n = c->n; sp.leaf_lo = 0; sp.leaf_hi = L; }
  const uint64_t n_it = sp.it_hi - sp.it_lo;            // keys this launch works on
  const uint64_t L_own = sp.leaf_hi - sp.leaf_lo;
  // pointers pre-offset so that ptr[global index] is the right element
  // (streamed training: buffers of the whole key set / all leaves, indexed globally)
  const uint64_t kb = c->stream_mode ? sp.rd_lo : 0, lb = c->stream_mode ? sp.leaf_lo : 0;
  const K* keys = (const K*)c->d_keys + kb - sp.rd_lo;
  unsigned long long* const a_leaf_start = c->d_leaf_start + lb;
  unsigned long long* const a_maxerr = c->d_maxerr + lb;
  unsigned long long* const a_run = c->d_run + lb;
  unsigned long long* leaf_start = a_leaf_start - sp.leaf_lo;
  double* params = c->d_params + lb * PPL - sp.leaf_lo * PPL;
  unsigned long long* maxerr = a_maxerr - sp.leaf_lo;
  unsigned long long* run = a_run - sp.leaf_lo;
  unsigned long long* err = c->d_err + lb - sp.leaf_lo;
  unsigned long long* count = c->d_count + lb - sp.leaf_lo;
  unsigned char* rows_base = c->d_rows_ext ? (unsigned char*)c->d_rows_ext : c->d_rows;
  c->last_rows = rows_base;
  unsigned char* rows = rows_base + lb * ROWB - sp.leaf_lo * ROWB;

  // --- init ---
  // long-leaf list: a long leaf has at least long_min points, so n/long_min entries always suffice
  const unsigned int long_min_a = c->long_min < (unsigned int)FS_TMAX ? (unsigned int)FS_TMAX : c->long_min;   // pass A's table covers FS_TMAX counts
  uint64_t need_long = n_it / long_min_a + 1024;
  if (c->fit_mode != 0 && need_long < L_own + 1024) need_long = L_own + 1024;      // one-pass mode: every listed leaf goes through k_fit_long
  if (c->long_cap < need_long) {
    if (c->d_long) (void)hipFree(c->d_long);
    c->d_long = nullptr; c->long_cap = 0;
    HIPCHK(c, hipMalloc(&c->d_long, need_long * 8));
    c->long_cap = need_long;
  }
  DevState init; std::memset(&init, 0, sizeof init);
  init.long_cap = c->long_cap;
  init.flag_cap = (uint64_t)L_own + 64;
  init.seg_cap = n_it / SG_SEG + L_own + 16;
  init.giant_cap = c->host_min > 0 ? n_it / c->host_min + 64 : 0;
  init.split_idx = (c->have_shard && c->shard_split_idx != ~0ull) ? c->shard_split_idx : sp.n;
  init.split_target = (c->have_shard && c->shard_split_idx != ~0ull) ? c->shard_split_target : 0;
  init.last_target = ~0ull;
  c->giant_armed = false;
  if (!c->d_flist_cnt) HIPCHK(c, hipMalloc(&c->d_flist_cnt, (2 * SG_REGIONS + 8) * 8));  // (the one-pass mode's list + merge counters: zeroed by k_init)
  // pipeline 1 launches one thread per key: a grid dimension holds fewer than 2^32 threads
  const bool stream_fit = (LEAF == K_LINEAR) && !c->robust_leaf;
  // one pass from sufficient statistics: leaves of at least a few dozen keys on average (the rows of a tile hold
  // 16 keys; shorter leaves are cheap chains for the exact kernels anyway), indices below 2^32
  bool hinted = false;
  if (c->hint_epoch == c->keys_epoch && c->hint_mode == c->fit_mode)
    for (int h = 0; h < c->hint_n && h < 8; h++) hinted = hinted || c->hint_L[h] == L_own;
  // linear_spline leaves (the line through a container's two end points) need no sums and no guard: the one-pass kernel
  // reproduces them bit for bit, so it serves EVERY fit mode (RMI_HIP_SPLINE_ONEPASS=0: the per-pass kernels)
  // ... and, in front of it, the leaf-lane kernel (two end points per leaf, then its error pass with duplicates handled natively)
  // pipeline 5 (rmi_scan.hip.h): linear_spline leaves in ONE key-parallel pass, the bucketing scan included -- every root, every key type,
  // any number of keys (an empty shard as well); 32-bit indices
  const bool scan5 = c->pipeline >= 3 && (LEAF == K_LINEAR_SPLINE) && sp.n < (1ull << 32) - (1ull << 16);
  c->last_scan = scan5;
  const bool spline_l = scan5;                             // (round 6: k_leaf_lanes<.., K_LINEAR_SPLINE>, the second route for these leaves, is gone)
  // (round 5: the one-pass kernel's linear_spline variant is gone -- k_spline_scan serves them, and where it does not (RMI_HIP_SCAN=0 with
  //  RMI_HIP_SPLINE_LANES=0, 2^32 keys and more, RMI_HIP_PIPELINE <= 2) the per-pass kernels do; the variant missed the borrowed point of
  //  a leaf behind an emptied split leaf, Q4)
  const bool spline1 = false;
  const bool sigma = ((stream_fit && c->fit_mode != 0) || spline1) && !hinted && sp.n < (1ull << 32) - (1ull << 16) &&
                     n_it >= (uint64_t)c->sigma_min_leaf * L_own && n_it >= 4096;
  c->last_sigma = sigma;
  c->last_spline = sigma && LEAF == K_LINEAR_SPLINE;
  // exact linear leaves, pipeline 3: the leaf-lane kernels (rmi_lanes.hip.h)
  const bool lanes = (c->pipeline >= 3 && stream_fit && !sigma && n_it >= 1024) || spline_l;   // (tiny key sets: the streaming passes)
  c->last_lanes = lanes;
  // the fused error pass of k_leaf_lanes needs 32-bit indices; with it and the search, k_init has no array to prepare
  const bool lanes_fused_plan = lanes && c->lanes_fuse && sp.n < (1ull << 32) - (1ull << 16);
  bool lanes_search_plan = false;
  if constexpr (ROOT == K_LINEAR) lanes_search_plan = lanes && c->lanes_search && rp.p1 >= 0.0 && std::isfinite(rp.p0) && std::isfinite(rp.p1);
  // cubic roots: not monotone by arithmetic, but the search may assume it when k_leaf_lanes verifies every key's target
  // during its (fused) error pass -- no bucketing scan, no fill
  if constexpr (ROOT == K_CUBIC) lanes_search_plan = lanes && c->lanes_search && lanes_fused_plan && LEAF == K_LINEAR && std::isfinite(rp.p0) && std::isfinite(rp.p1) && std::isfinite(rp.p2) && std::isfinite(rp.p3);
  bool scan_mono = false;                                  // pipeline 5: the root is monotone by arithmetic (its short form of a tile relies on it)
  if constexpr (ROOT == K_LINEAR) scan_mono = scan5 && rp.p1 >= 0.0 && std::isfinite(rp.p0) && std::isfinite(rp.p1);
  if constexpr (ROOT == K_RADIX) {
    // (key << prefix) >> (64 - bits) is monotone in the key exactly when no key loses a distinguishing bit to the
    // shift: all resident keys share their top `prefix` bits.  First and last key of the sorted set decide; fetched once per key set.
    if ((lanes && c->lanes_search) || scan5) {
      if (c->edge_epoch != c->keys_epoch) {
        K k0{}, k1{};
        HIPCHK(c, hipMemcpy(&k0, c->d_keys, sizeof(K), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&k1, (const K*)c->d_keys + (c->n - 1), sizeof(K), hipMemcpyDeviceToHost));
        c->edge_first = rmi_host::as_uint(k0); c->edge_last = rmi_host::as_uint(k1); c->edge_epoch = c->keys_epoch;
      }
      const unsigned pfx = rp.prefix & 63u;
      lanes_search_plan = pfx == 0u || ((c->edge_first ^ c->edge_last) >> (64u - pfx)) == 0ull;
      scan_mono = scan5 && lanes_search_plan;
    }
  }
  if (scan5) lanes_search_plan = false;                    // (the scan finds the leaf starts itself)
  const bool init_arrays = scan5 ? false : !(lanes_fused_plan && lanes_search_plan);
  if ((!c->stream_mode || c->stream_slot == 0) && pl >= 0) HIPCHK(c, hipEventRecord(c->ev[8], s));   // start of the device work of this call
  // (the leaf-lane pipeline with its search and its fused error pass: the launch of k_leaf_samples carries the init)
  const bool init_folded = !scan5 && !init_arrays && (LEAF == K_LINEAR || LEAF == K_LINEAR_SPLINE);
  if (!init_folded) {
    const uint64_t ib = init_arrays ? (L_own + 1 + 255) / 256 : 1;
    hipLaunchKernelGGL(k_init, dim3((unsigned)(ib < 2048 ? ib : 2048)), dim3(256), 0, s, a_leaf_start, a_maxerr, a_run,
                       L_own, (unsigned long long)sp.it_hi, c->d_state, init, c->d_flist_cnt, 2 * SG_REGIONS + 8, init_arrays, scan5 ? c->d_tickets : (unsigned int*)nullptr);
  }
  c->tail_armed = false; c->tail_fn = nullptr; c->regs_listed_fn = nullptr; c->giant_early = false; c->giant_fitted = false;
  c->last_lean = false; c->lean_derived = false;

  auto ensure_lists = [&]() -> int {
    const uint64_t rcap = (L_own + SG_REGIONS - 1) / SG_REGIONS + 8;      // a region holds every leaf with its residue, and the odd re-listed one
    if (c->flist_cap < rcap) {
      if (c->d_flist) (void)hipFree(c->d_flist);
      c->d_flist = nullptr; c->flist_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_flist, rcap * SG_REGIONS * 4));
      c->flist_cap = rcap;
    }
    const uint64_t scap = n_it / SG_SEG + L_own + 16;                     // stretches of the long listed leaves (k_list -> k_list_tail)
    if (c->segs_cap < scap) {
      if (c->d_segs) (void)hipFree(c->d_segs);
      c->d_segs = nullptr; c->segs_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_segs, scap * 8));
      c->segs_cap = scap;
    }
    return RMI_OK;
  };
  // giant leaves go to the host when this call ends with its own synchronisation (not inside a streamed or a sharded training)
  auto arm_giants = [&]() -> int {
    const uint64_t gcap = n_it / c->host_min + 64;
    if (c->giant_cap < gcap) {
      if (c->d_giant) (void)hipFree(c->d_giant);
      c->d_giant = nullptr; c->giant_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_giant, gcap * sizeof(GiantLeaf)));
      c->giant_cap = gcap;
    }
    c->giant_armed = true;
    if (!c->h_giant) HIPCHK(c, hipHostMalloc((void**)&c->h_giant, 8 + rmi_hip_ctx::GIANT_EARLY_MAX * sizeof(GiantLeaf), hipHostMallocDefault));
    if (!c->ev_giant) HIPCHK(c, hipEventCreateWithFlags(&c->ev_giant, hipEventDisableTiming));
    c->lp.keys = keys; c->lp.sp = sp; c->lp.L = L; c->lp.leaf_start = leaf_start; c->lp.params = params; c->lp.maxerr = maxerr; c->lp.run = run;
    c->lp.err = err; c->lp.count = count; c->lp.rows = rows; c->lp.waves = (L_own + 63) / 64;
    return RMI_OK;
  };
  c->refinalize_fn = nullptr;
  bool lanes_fused = false;
  if (lanes) {
    if constexpr (LEAF == K_LINEAR || LEAF == K_LINEAR_SPLINE) {
      { const int lrc = ensure_lists(); if (lrc != RMI_OK) return lrc; }
      // --- leaf boundaries: lower bounds by search where the root is monotone by arithmetic, else the bucketing scan + fill ---
      bool searched = false;
      const uint64_t wb = (L_own + 63) / 64;
      uint64_t nrec = wb;                                               // aggregate records the fitting kernel leaves (one per wave)
      // the result published by k_lane_reduce right behind k_leaf_lanes; the list kernels behind the synchronisation, if a leaf was handed over
      const bool optimistic = lanes_fused_plan && c->opt_tail && !c->stream_mode;
      if constexpr (ROOT == K_LINEAR || ROOT == K_RADIX || ROOT == K_CUBIC) {
        if (lanes_search_plan) {
          const uint64_t sb = (L_own + (uint64_t)LS_BLOCK * LS_ILP - 1) / ((uint64_t)LS_BLOCK * LS_ILP);
          double* smp = c->d_lntab + 3 * LN_TMAX;                       // 2 (LS_SAMPLES + 1) doubles behind the step tables
          LaneInit li; std::memset(&li, 0, sizeof li);
          if (init_folded) {
            li.st = c->d_state; li.init = init; li.leaf_start = a_leaf_start; li.L_own = L_own; li.sentinel = (unsigned long long)sp.it_hi;
            li.list_cnt = c->d_flist_cnt; li.n_list_cnt = 2 * SG_REGIONS + 8; li.tickets = c->d_tickets; li.n_tickets = 3;
          }
          hipLaunchKernelGGL((k_leaf_samples<ROOT, K>), dim3((LS_SAMPLES + 256) / 256), dim3(256), 0, s, keys, sp, rp, smp, li);
          hipLaunchKernelGGL((k_leaf_search<ROOT, K, LS_ILP>), dim3((unsigned)sb), dim3(LS_BLOCK), 0, s, keys, sp, rp, leaf_start, c->d_state, (const double*)smp);
          searched = true;
        }
      }
      if (!(searched && init_folded) && !scan5) HIPCHK(c, hipMemsetAsync(c->d_tickets, 0, 12, s));   // (k_lane_reduce's arrival counter, k_leaf_regs' list counter and group counter)
      if (!searched && !scan5) {
        constexpr uint64_t V = 16 / sizeof(K);
        const uint64_t blocks = ((n_it + V - 1) / V + 256 * BV_UNROLL - 1) / (256 * BV_UNROLL);
        hipLaunchKernelGGL((k_bounds_vec<ROOT, K>), dim3((unsigned)blocks), dim3(256), 0, s, keys, sp, rp, leaf_start, c->d_state);
        const uint64_t count_e = L_own + 1;
        const uint64_t ntiles = (count_e + FILL_TILE - 1) / FILL_TILE;
        hipLaunchKernelGGL(k_fill_tilemin, dim3((unsigned)ntiles), dim3(256), 0, s, a_leaf_start, count_e, c->d_tilemin);
        hipLaunchKernelGGL(k_fill_scan_tiles, dim3(1), dim3(1024), 0, s, c->d_tilemin, ntiles);
        hipLaunchKernelGGL(k_fill_apply, dim3((unsigned)ntiles), dim3(256), 0, s, a_leaf_start, count_e, c->d_tilemin);
      }
      if (pl >= 1) HIPCHK(c, hipEventRecord(c->ev[0], s));              // (the bracket of the dominant kernel starts here)
      // --- exact fit of 64 leaves per wave in lockstep, and their error pass behind it ---
      lanes_fused = lanes_fused_plan;
      SgList fl; fl.ids = c->d_flist; fl.cnt = c->d_flist_cnt; fl.cap = c->flist_cap;
      // (giant leaves are for the OUTLIERS of a skewed key set: where the average leaf is within a factor of four of the threshold nearly every leaf would go
      //  to the host -- 400 M u32 keys in 1 024 leaves: 680 ms, the keys over PCIe and 1 024 chains on a few cores, against 14 ms of 1 024 waves side by side)
      const bool giants_pay = c->host_min > 0 && (c->host_min_set || n_it / (L_own ? L_own : 1) <= c->host_min / 4);
      const bool giants = LEAF == K_LINEAR && lanes_fused_plan && giants_pay && !c->stream_mode && !c->defer_sync;
      if (giants) { const int grc = arm_giants(); if (grc != RMI_OK) return grc; }
      const unsigned int lmin = c->long_min < (unsigned int)LN_LONG_MAX ? c->long_min : (unsigned int)LN_LONG_MAX;
      StatsPartial* const part = c->d_partials;
      // sharded training with the direct exchange: the rows of the leaves k_leaf_lanes finishes go to every peer's table from
      // the kernel itself (train_sharded_direct has set the tables of this epoch); leaves handed to the list kernels follow
      // with the whole slot in the second exchange of the `pending` protocol
      PeerRows peers; std::memset(&peers, 0, sizeof peers);
      c->rows_pushed = false;
      // (only with the published-early result: with the list kernels in-stream, RMI_HIP_OPT_TAIL=0, `pending` is 0, no second exchange
      //  follows, and the rows of the listed leaves would never reach the peers -- k_peer_push then carries the whole slot)
      if (lanes_fused_plan && c->peer_fuse_n > 0 && optimistic) {   // (linear and linear_spline leaves: rows of 24 bytes)
        peers.n = c->peer_fuse_n;
        for (int p = 0; p < peers.n; p++) peers.tab[p] = c->peer_fuse_tab[p];
        c->rows_pushed = true;
      }
      // pipeline 4's conditions besides the root's: 8-byte keys, linear leaves, leaves short enough on average that most groups of 64 qualify,
      // not a key set on which k_leaf_regs listed most groups last time
      bool regs_plan = false, regs_long = false;
      if constexpr (LEAF == K_LINEAR) {
        bool regs_off = false;
        if (c->regs_off_epoch == c->keys_epoch)
          for (int h = 0; h < c->regs_off_n && h < 8; h++) regs_off = regs_off || c->regs_off_L[h] == L_own;
        regs_long = n_it > (uint64_t)c->regs_max_avg * L_own;            // long leaves on average: the LONG variant of the kernel
        const unsigned int cap = c->regs_long_max_avg > c->regs_max_avg ? c->regs_long_max_avg : c->regs_max_avg;
        regs_plan = lanes_fused_plan && c->regs && !regs_off && c->pipeline >= 3 && n_it <= (uint64_t)cap * L_own && (sizeof(K) == 8 || c->regs_u32);
        // Between one and two and a half groups per resident wave (M's shard at 8 GPUs: 2 048 groups on 1 024 waves) k_leaf_regs runs two rounds of a
        // group each behind its 20 us of start-up, k_leaf_lanes ONE round on twice the waves: measured 0.110 against 0.120 ms at 2 048 groups, equal
        // at 1 024, 0.186 against 0.177 at 4 096.
        {
          const uint64_t resident = 4ull * (uint64_t)c->n_cu;
          if (!regs_long && !c->regs_forced && c->regs_grid == 0 && wb > resident && 2 * wb <= 5 * resident) regs_plan = false;
        }
      }
      // a cubic root on pipeline 4: increasing over the resident keys' range as an exact polynomial (here), every leaf's end keys clear
      // their leaf's interval by the rounding bound (k_regs_finalize<K, K_CUBIC>) -- else the per-key verification of k_leaf_lanes
      bool cubic_margin = false;
      if constexpr (ROOT == K_CUBIC && LEAF == K_LINEAR) {
        if (searched && regs_plan && c->cubic_margin) {
          if (c->edgef_epoch != c->keys_epoch) {
            K k0{}, k1{};
            HIPCHK(c, hipMemcpy(&k0, c->d_keys, sizeof(K), hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(&k1, (const K*)c->d_keys + (c->n - 1), sizeof(K), hipMemcpyDeviceToHost));
            c->edge_first_f = rmi_host::as_float(k0); c->edge_last_f = rmi_host::as_float(k1); c->edgef_epoch = c->keys_epoch;
          }
          cubic_margin = cubic_increasing_on(rp, c->edge_first_f, c->edge_last_f);
        }
      }
      bool verify = false;
      if constexpr (ROOT == K_CUBIC && LEAF == K_LINEAR) {
        if (searched && !cubic_margin) {                                // (searched implies fused: the verification rides on the error pass)
          hipLaunchKernelGGL((k_leaf_lanes<K, true, K_LINEAR, K_CUBIC>), dim3((unsigned)wb), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, c->d_lntab, fl, lmin, maxerr, run,
                             L, err, count, rows, part, rp, peers);
          verify = true;
        }
      }
      // pipeline 4: one read of the keys (8-byte keys, linear leaves, leaves short enough on average that most groups of 64 qualify)
      bool regs = false;
      if constexpr (LEAF == K_LINEAR) {
        // (a key set on which k_leaf_regs listed most groups -- duplicate-heavy keys: every group meets a duplicate -- is remembered, like
        //  the one-pass modes' hint: the next trainings of it with that many leaves go straight to k_leaf_lanes, 0.80 against 2.25 ms)
        regs = !verify && lanes_fused && regs_plan;
        if (regs) {
          if (c->slow_cap < wb) {
            if (c->d_slow_list) (void)hipFree(c->d_slow_list);
            if (c->d_tile_slow) (void)hipFree(c->d_tile_slow);
            if (c->d_bnext) (void)hipFree(c->d_bnext);
            c->d_slow_list = nullptr; c->d_tile_slow = nullptr; c->d_bnext = nullptr; c->slow_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_slow_list, wb * 4));
            HIPCHK(c, hipMalloc(&c->d_tile_slow, wb));
            HIPCHK(c, hipMalloc(&c->d_bnext, wb * 64 * 16));                // (the keys on either side of every leaf: 2 x 8 bytes)
            c->slow_cap = wb;
          }
          HIPCHK(c, hipGetLastError());
          const bool w2 = sizeof(K) == 4 && c->regs_u32 >= 2;           // two waves per SIMD
          unsigned int grid = c->regs_grid ? c->regs_grid : (w2 ? 8u : 4u) * (unsigned int)c->n_cu;
          if ((uint64_t)grid > wb) grid = (unsigned int)wb;
          // (the variant: 0 = leaves that fit the stash and the ring, 1 = LONG, 2 = 4-byte keys at two waves per SIMD)
          auto launch_regs = [&](auto variant_tag) {
            constexpr int VARIANT = decltype(variant_tag)::value;
            hipLaunchKernelGGL((k_leaf_regs<K, VARIANT>), dim3(grid), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, c->d_lntab, c->d_regtab, fl, lmin, maxerr, run,
                               L, err, count, rows, part, rp, peers, (unsigned int)wb, (c->regs_slow & 1u) | (c->regs_backoff ? 2u : 0u), c->d_slow_list, c->d_tickets + 1, c->d_regprof,
                               (K*)c->d_bnext, (K*)c->d_bnext + wb * 64, c->d_tile_slow, c->regs_queue ? c->d_tickets + 2 : (unsigned int*)nullptr);
          };
          bool launched = false;
          if constexpr (sizeof(K) == 4) { if (w2) { launch_regs(std::integral_constant<int, 2>{}); launched = true; } }
          if (!launched) { if (regs_long) launch_regs(std::integral_constant<int, 1>{}); else launch_regs(std::integral_constant<int, 0>{}); }
          mark();                                                       // (slot 0: k_leaf_regs alone; slot 1: the listed groups + k_regs_finalize)
          // The groups k_leaf_regs listed.  As a rule there are none: with the result published early (k_lane_reduce) and the host
          // synchronising itself, the kernel is launched only when the record says a group was listed (DevState::regs_listed) --
          // an empty launch cost 4.6 us and a gap in every step; in-stream otherwise (a sharded or streamed training).
          const unsigned int lgrid = wb < 512 ? (unsigned int)wb : 512u;
          unsigned int* const slow_list = c->d_slow_list;
          unsigned int* const slow_cnt = c->d_tickets + 1;
          double* const lntab = c->d_lntab;
          DevState* const dstate = c->d_state;
          auto listed = [=](unsigned int grid_l) {
            hipLaunchKernelGGL((k_leaf_lanes_listed<K>), dim3(grid_l), dim3(64), 0, s, slow_list, slow_cnt, keys, sp, leaf_start, dstate, params, lntab, fl, lmin,
                               maxerr, run, L, err, count, rows, part, rp, peers);
          };
          const bool listed_late = optimistic && !c->defer_sync && peers.n == 0;
          if (listed_late) {
            StatsPartial* const slices = part + nrec + SG_REGIONS + 2 * FL_BLOCKS + 1;
            unsigned int* const tickets = c->d_tickets;
            DevState* const hcopy_l = c->h_state_dev;
            const unsigned int nrec_l = (unsigned int)nrec;
            const unsigned int wb_l = (unsigned int)wb;
            c->regs_listed_fn = [=](unsigned int groups) -> int {                 // (as many waves as groups: a duplicate-heavy key set lists them all)
              listed(groups < wb_l ? (groups > 0u ? groups : 1u) : wb_l);
              hipLaunchKernelGGL(k_lane_reduce, dim3((nrec_l + LF_SLICE - 1) / LF_SLICE), dim3(LF_SLICE), 0, s, (const StatsPartial*)part, nrec_l, slices, tickets, fl, dstate, hcopy_l);
              return RMI_OK;
            };
          } else listed(lgrid);
          if (cubic_margin)
            hipLaunchKernelGGL((k_regs_finalize<K, K_CUBIC>), dim3((unsigned)((wb * 64 + 255) / 256)), dim3(256), 0, s, keys, sp, L, leaf_start, c->d_state, (const unsigned int*)(c->d_tickets + 1), params,
                               (const unsigned long long*)maxerr, (const K*)c->d_bnext, (const K*)c->d_bnext + wb * 64, (const unsigned char*)c->d_tile_slow, (unsigned int)wb,
                               err, count, rows, part, peers, rp, c->cubic_margin_scale, listed_late);
          else
            hipLaunchKernelGGL((k_regs_finalize<K>), dim3((unsigned)((wb * 64 + 255) / 256)), dim3(256), 0, s, keys, sp, L, leaf_start, c->d_state, (const unsigned int*)(c->d_tickets + 1), params,
                               (const unsigned long long*)maxerr, (const K*)c->d_bnext, (const K*)c->d_bnext + wb * 64, (const unsigned char*)c->d_tile_slow, (unsigned int)wb,
                               err, count, rows, part, peers, rp, 1.0, listed_late);
          HIPCHK(c, hipGetLastError());
        }
      }
      c->last_regs = regs;
      if (scan5) {
        ScanLaunch sl; std::memset(&sl, 0, sizeof sl);
        sl.keys = keys; sl.sp = sp; sl.rp = rp; sl.st = c->d_state; sl.fl = fl; sl.long_min = lmin;
        // (a streamed / sharded training fills the arrays shard by shard: kept whole there; rows in a caller's buffer -- rmi_hip_set_rows_output --
        //  may be gone or overwritten when the arrays are asked for: lean_fill derives them from the rows, so no lean outputs then)
        const bool lean5 = c->lean && !c->stream_mode && !c->defer_sync && c->d_rows_ext == nullptr;
        sl.out.leaf_start = leaf_start; sl.out.rows = rows; sl.out.partials = part;
        sl.out.params = lean5 ? nullptr : params; sl.out.leaf_err = lean5 ? nullptr : err; sl.out.leaf_count = lean5 ? nullptr : count;
        c->last_lean = lean5;
        sl.peers = peers;
        sl.host_split = (c->have_shard && c->shard_split_idx != ~0ull) ? 1 : 0;
        sl.mono = scan_mono ? 1 : 0;
        sl.max_waves = c->scan_waves;
        sl.n_cu = (unsigned int)c->n_cu;
        // the general form's kernel gets as many waves as tiles were left to it by the last training of this (key set, leaf count) -- twice
        // that and 64 more; a first training, or one in which the list outgrows them, launches what the device holds
        sl.listed_hint = (c->scan_hint_epoch == c->keys_epoch && c->scan_hint_L == L_own) ? c->scan_hint_n : ~0u;
        sl.long_leaves = (c->scan_skew_epoch == c->keys_epoch && c->scan_skew_L == L_own) ? 1 : 0;
        {
          const uint64_t need = rmi_scan_tiles(c->dtype, n_it);
          if (c->tile_list_cap < need) {
            if (c->d_tile_list) (void)hipFree(c->d_tile_list);
            c->d_tile_list = nullptr; c->tile_list_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_tile_list, need * 4));
            c->tile_list_cap = need;
          }
        }
        sl.tile_list = c->d_tile_list; sl.tile_cnt = c->d_flist_cnt + 2 * SG_REGIONS + 2;   // (zeroed by k_init with the list counters)
        if (!c->d_gaps) HIPCHK(c, hipMalloc(&c->d_gaps, (size_t)SCAN_GAP_CAP * sizeof(GapRec)));
        sl.gaps = (GapRec*)c->d_gaps; sl.gap_cnt = c->d_flist_cnt + 2 * SG_REGIONS + 1;   // (zeroed by k_init with the list counters)
        if (rmi_scan_launch(ROOT, c->dtype, sl, s) != 0) { set_err(c, "internal: k_spline_scan is not built for root %d / key type %d", ROOT, c->dtype); return RMI_ERR_HIP; }
        (void)rmi_scan_gaps_launch(c->dtype, sl, s);
        nrec = sl.waves;
      } else if (verify || regs) {
      } else if constexpr (LEAF == K_LINEAR) {                            // (linear_spline leaves reach this point only with scan5)
        if (lanes_fused)
          hipLaunchKernelGGL((k_leaf_lanes<K, true, K_LINEAR>), dim3((unsigned)wb), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, c->d_lntab, fl, lmin, maxerr, run,
                             L, err, count, rows, part, rp, peers);
        else
          hipLaunchKernelGGL((k_leaf_lanes<K, false, K_LINEAR>), dim3((unsigned)wb), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, c->d_lntab, fl, lmin, maxerr, run,
                             L, err, count, rows, part, rp, peers);
      }
      mark();
      // --- the leaves handed over (containers too long for the lockstep walk): one wave each, fit + error pass; the
      //     listed leaves' share of the finalize, the aggregates, the result record ---
      SgParams sgp; std::memset(&sgp, 0, sizeof sgp);
      sgp.flist = fl; sgp.segs = c->d_segs; sgp.mode = 0; sgp.guard_k = c->guard_k;
      c->last_sg = sgp;
      unsigned long long* const segs = c->d_segs;
      DevState* const dst = c->d_state;
      unsigned long long* const fticket = c->d_flist_cnt + 2 * SG_REGIONS;
      DevState* const hcopy = c->h_state_dev + (c->stream_mode ? c->stream_slot : 0);
      GiantLeaf* const dgiant = giants ? c->d_giant : (GiantLeaf*)nullptr;
      const unsigned long long hmin = giants ? (unsigned long long)c->host_min : ~0ull;
      const bool fused = lanes_fused;
      // (the early giant list only where the host runs the tail itself, behind its synchronisation)
      const bool early = optimistic && giants && !c->defer_sync && c->h_giant != nullptr;
      unsigned long long* const hg = c->h_giant;
      hipEvent_t const evg = c->ev_giant;
      const uint64_t hgn = c->giant_cap < rmi_hip_ctx::GIANT_EARLY_MAX ? c->giant_cap : rmi_hip_ctx::GIANT_EARLY_MAX;
      auto tail = [=](auto&& mk) {
        if constexpr (ROOT == K_CUBIC && LEAF == K_LINEAR) {
          if (verify) hipLaunchKernelGGL((k_verify_listed<K_CUBIC, K>), dim3(1024), dim3(256), 0, s, keys, sp, rp, leaf_start, dst, fl);
        }
        // (one wave per listed leaf wherever possible: on skewed keys thousands of leaves are listed and each is a sequential chain)
        if (early) {
          hipLaunchKernelGGL((k_giant_scan<K>), dim3(SG_REGIONS), dim3(64), 0, s, keys, sp, leaf_start, dst, fl, dgiant, hmin);
          (void)hipMemcpyAsync(hg, &dst->giant_count, 8, hipMemcpyDeviceToHost, s);
          (void)hipMemcpyAsync(hg + 1, dgiant, hgn * sizeof(GiantLeaf), hipMemcpyDeviceToHost, s);
          (void)hipEventRecord(evg, s);
        }
        hipLaunchKernelGGL((k_list<K, LEAF>), dim3(128 * SG_REGIONS), dim3(64), 0, s, keys, sp, leaf_start, dst, params, fl, sgp, maxerr, run, dgiant, hmin, !early);
        mk();
        // (waves' aggregate records: [0, wb); their 64 slice sums, by k_list_tail: [wb, wb + 64); the records of k_finalize_listed behind)
        hipLaunchKernelGGL((k_list_tail<K>), dim3(2048), dim3(64), 0, s, keys, sp, leaf_start, dst, params, fl, segs, maxerr, run,
                           fused ? (const StatsPartial*)part : (const StatsPartial*)nullptr, (unsigned int)nrec, part + nrec);
        mk();
        if (fused)
          hipLaunchKernelGGL((k_finalize_listed<K>), dim3(FL_BLOCKS), dim3(FL_THREADS), 0, s, keys, sp, L, leaf_start, dst, params, maxerr, run, err, count, rows,
                             fl, part + nrec, (unsigned int)SG_REGIONS, part + nrec + SG_REGIONS, fticket, dst, hcopy, (const GiantLeaf*)nullptr, hmin);
      };
      if (optimistic) {
        const unsigned int nsl = (unsigned int)((nrec + LF_SLICE - 1) / LF_SLICE);
        hipLaunchKernelGGL(k_lane_reduce, dim3(nsl), dim3(LF_SLICE), 0, s, (const StatsPartial*)part, (unsigned int)nrec, part + nrec + SG_REGIONS + 2 * FL_BLOCKS + 1,
                           c->d_tickets, fl, dst, hcopy);
        mark(); mark();
        c->tail_armed = true;
        c->giant_early = early;
        c->tail_fn = [tail]() -> int { tail([]() {}); return RMI_OK; };
      } else tail(mark);
    }
  } else if (pl >= 1) HIPCHK(c, hipEventRecord(c->ev[0], s));
  if (lanes) {
  } else if (sigma) {
    if constexpr (LEAF == K_LINEAR || LEAF == K_LINEAR_SPLINE) {
      { const int lrc = ensure_lists(); if (lrc != RMI_OK) return lrc; }
      if (c->bkeys_cap < L_own) {
        if (c->d_bkeys) (void)hipFree(c->d_bkeys);
        c->d_bkeys = nullptr; c->bkeys_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_bkeys, 2 * L_own * 8));
        c->bkeys_cap = L_own;
      }
      SgParams sgp; sgp.guard_k = c->guard_k; sgp.mode = c->fit_mode;
      sgp.flist.ids = c->d_flist; sgp.flist.cnt = c->d_flist_cnt; sgp.flist.cap = c->flist_cap;
      sgp.dbg = 0;
      {
        auto launch2 = [&](auto ring_tag, auto batch_tag) -> int {
          constexpr int RING = decltype(ring_tag)::value, BATCH = decltype(batch_tag)::value;
          uint64_t chunk = (n_it + c->sigma_waves - 1) / c->sigma_waves;
          chunk = ((chunk + BATCH - 1) / BATCH) * BATCH;
          if (chunk < (uint64_t)BATCH * 4) chunk = (uint64_t)BATCH * 4;     // (>= the window of the chunk rule, RING / 2)
          sgp.chunk = chunk;
          const uint64_t sblocks = (n_it + chunk - 1) / chunk;
          sgp.recs = nullptr; sgp.rec_cnt = nullptr; sgp.rpw = 0; sgp.segs = c->d_segs;
          if (LEAF == K_LINEAR && c->fit_mode == 2) {
            // a stretch of a long leaf is at least RING / 2 - BATCH keys, or the only one of its wave
            const uint64_t rpw = chunk / (RING / 2 - BATCH) + 3;
            const uint64_t need = sblocks * rpw * sizeof(SgRec) + sblocks * 4;
            if (c->recs_bytes < need) {
              if (c->d_recs) (void)hipFree(c->d_recs);
              c->d_recs = nullptr; c->recs_bytes = 0;
              HIPCHK(c, hipMalloc(&c->d_recs, need));
              c->recs_bytes = need;
            }
            sgp.recs = (SgRec*)c->d_recs; sgp.rec_cnt = (unsigned int*)((char*)c->d_recs + sblocks * rpw * sizeof(SgRec)); sgp.rpw = (unsigned int)rpw;
          }
          c->last_sg = sgp;
          if (c->fit_mode == 2)
            hipLaunchKernelGGL((k_sigma2<ROOT, K, RING, BATCH, true>), dim3((unsigned)sblocks), dim3(64), 0, s, keys, sp, rp, sgp, leaf_start, params, maxerr, c->d_state,
                               (K*)c->d_bkeys - sp.leaf_lo, (K*)c->d_bkeys + c->bkeys_cap - sp.leaf_lo);
          else
            hipLaunchKernelGGL((k_sigma2<ROOT, K, RING, BATCH, false>), dim3((unsigned)sblocks), dim3(64), 0, s, keys, sp, rp, sgp, leaf_start, params, maxerr, c->d_state,
                               (K*)c->d_bkeys - sp.leaf_lo, (K*)c->d_bkeys + c->bkeys_cap - sp.leaf_lo);
          return RMI_OK;
        };
        // (4-byte keys: the ring holds them raw, twice as many in the same LDS, and the batch is four loads as well)
        int lrc;
        if constexpr (sizeof(K) == 4) lrc = launch2(std::integral_constant<int, 4096>{}, std::integral_constant<int, 1024>{});
        else lrc = launch2(std::integral_constant<int, 2048>{}, std::integral_constant<int, 512>{});
        if (lrc != RMI_OK) return lrc;
        mark();
      }
    }
  } else if (n_it == 0) {
    mark();                                              // a shard without keys: every leaf is empty
  } else if (!stream_fit) {
    // --- bucketing scan ---
    {
      constexpr uint64_t V = 16 / sizeof(K);
      const uint64_t blocks = ((n_it + V - 1) / V + 256 * BV_UNROLL - 1) / (256 * BV_UNROLL);
      hipLaunchKernelGGL((k_bounds_vec<ROOT, K>), dim3((unsigned)blocks), dim3(256), 0, s, keys, sp, rp, leaf_start, c->d_state);
    }
    mark();
  } else {
    // --- pass A: bucketing scan + exact per-leaf fit in one streaming pass ---
    uint64_t C = (n_it + c->fit_threads - 1) / c->fit_threads;
    C = ((C + FS_ROW - 1) / FS_ROW) * FS_ROW;
    if (C < (uint64_t)c->fit_min_chunk) C = c->fit_min_chunk;
    const uint64_t chunks = (n_it + C - 1) / C;
    const uint64_t waves = (chunks + 63) / 64;
    const uint64_t fblocks = (waves + FA_WAVES - 1) / FA_WAVES;
    hipLaunchKernelGGL((k_fit_stream<ROOT, K>), dim3((unsigned)fblocks), dim3(64 * FA_WAVES), 0, s, keys, sp, rp, C, leaf_start, params, c->d_state, c->d_long, long_min_a);
    mark();
  }
  // --- fill empty leaves ---
  if (!lanes) {
    const uint64_t count_e = L_own + 1;
    const uint64_t ntiles = (count_e + FILL_TILE - 1) / FILL_TILE;
    hipLaunchKernelGGL(k_fill_tilemin, dim3((unsigned)ntiles), dim3(256), 0, s, a_leaf_start, count_e, c->d_tilemin);
    hipLaunchKernelGGL(k_fill_scan_tiles, dim3(1), dim3(1024), 0, s, c->d_tilemin, ntiles);
    hipLaunchKernelGGL(k_fill_apply, dim3((unsigned)ntiles), dim3(256), 0, s, a_leaf_start, count_e, c->d_tilemin);
  }
  if (!lanes) mark();
  bool sigma_giants = false;
  if (lanes) {
  } else if (sigma) {
    if constexpr (LEAF == K_LINEAR || LEAF == K_LINEAR_SPLINE) {
      // --- the leaves the one-pass kernel handed over: fit (or merge) + error pass, one wave per leaf; long ones in stretches ---
      SgList fl; fl.ids = c->d_flist; fl.cnt = c->d_flist_cnt; fl.cap = c->flist_cap;
      // (the guarded mode re-fits its long leaves exactly: a giant one is a chain for a host core, as on the exact path;
      //  RMI_FIT_ONEPASS merges their partial sums instead -- tagged entries, never handed to the host)
      sigma_giants = LEAF == K_LINEAR && c->host_min > 0 && (c->host_min_set || n_it / (L_own ? L_own : 1) <= c->host_min / 4) && !c->stream_mode && !c->defer_sync;
      if (sigma_giants) {
        const int grc = arm_giants(); if (grc != RMI_OK) return grc;
        // (their list first, through pinned memory: the host walks those chains while k_list fits the other listed leaves)
        const uint64_t hgn = c->giant_cap < rmi_hip_ctx::GIANT_EARLY_MAX ? c->giant_cap : rmi_hip_ctx::GIANT_EARLY_MAX;
        hipLaunchKernelGGL((k_giant_scan<K>), dim3(SG_REGIONS), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, fl, c->d_giant, (unsigned long long)c->host_min);
        HIPCHK(c, hipMemcpyAsync(c->h_giant, &c->d_state->giant_count, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(c->h_giant + 1, c->d_giant, hgn * sizeof(GiantLeaf), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev_giant, s));
        c->giant_early = true;
      }
      hipLaunchKernelGGL((k_list<K, LEAF>), dim3(128 * SG_REGIONS), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, fl, c->last_sg, maxerr, run,
                         sigma_giants ? c->d_giant : (GiantLeaf*)nullptr, sigma_giants ? (unsigned long long)c->host_min : ~0ull, !sigma_giants);
      mark();
      hipLaunchKernelGGL((k_list_tail<K>), dim3(8192), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, fl, c->d_segs, maxerr, run);
    }
  } else if (n_it == 0) {
  } else if (!stream_fit) {
    // --- per-leaf fit ---
    const uint64_t blocks = (L_own + 255) / 256;
    const double* cube = nullptr;
    if constexpr (LEAF == K_CUBIC) {
      // The cube of every container's key range is the platform libm's pow on the host (that is
      // what the reference's coefficient is defined by); the kernels do everything else.
      if (c->cube_cap < L_own) {
        if (c->d_cube) (void)hipFree(c->d_cube);
        c->d_cube = nullptr; c->cube_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_cube, L_own * 8));
        c->cube_cap = L_own;
      }
      c->h_cube.resize(L_own);
      hipLaunchKernelGGL((k_cubic_span<K>), dim3((unsigned)blocks), dim3(256), 0, s, keys, sp, leaf_start, c->d_state, c->d_cube - sp.leaf_lo);
      HIPCHK(c, hipMemcpyAsync(c->h_cube.data(), c->d_cube, L_own * 8, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
      host_cubes(c->h_cube.data(), L_own);
      HIPCHK(c, hipMemcpyAsync(c->d_cube, c->h_cube.data(), L_own * 8, hipMemcpyHostToDevice, s));
      cube = c->d_cube - sp.leaf_lo;
    }
    hipLaunchKernelGGL((k_fit_leaf<LEAF, K>), dim3((unsigned)blocks), dim3(256), 0, s, keys, sp, leaf_start, c->d_state, params, cube, c->robust_leaf);
    if constexpr (LEAF == K_CUBIC) {
      if (n_it + 2 > (uint64_t)CUBIC_LONG) {                 // long containers are possible
        const uint64_t wb = L_own < 4096 ? L_own : 4096;
        hipLaunchKernelGGL((k_fit_cubic_long<K>), dim3((unsigned)wb), dim3(64), 0, s, keys, sp, leaf_start, c->d_state, params, cube);
      }
    }
  } else {
    // --- leaves handed over by pass A (more than long_min points): one wave each ---
    const uint64_t blocks = c->long_cap < 2048 ? c->long_cap : 2048;   // ~2 waves per SIMD saturate its f64 issue
    hipLaunchKernelGGL((k_fit_long<ROOT, K>), dim3((unsigned)blocks), dim3(64), 0, s, keys, sp, rp, leaf_start, c->d_state, params, c->d_long);
  }
  if (!sigma && !lanes) mark();
  // --- error pass ---
  if (n_it == 0 || sigma || lanes_fused) {
  } else {
    uint64_t C = (n_it + c->err_threads - 1) / c->err_threads;
    C = ((C + FS_ROW - 1) / FS_ROW) * FS_ROW;
    if (C < (uint64_t)c->fit_min_chunk) C = c->fit_min_chunk;
    const uint64_t chunks = (n_it + C - 1) / C;
    const uint64_t waves = (chunks + 63) / 64;
    hipLaunchKernelGGL((k_err_range<ROOT, LEAF, K>), dim3((unsigned)waves), dim3(64), 0, s, keys, sp, rp, C, leaf_start, params, maxerr, run);
  }
  mark();
  // --- finalize + stats ---
  if (!lanes_fused) {
    const uint64_t blocks = (L_own + 255) / 256;
    const K* bn = sigma ? (const K*)c->d_bkeys - sp.leaf_lo : nullptr;
    const K* bp = sigma ? (const K*)c->d_bkeys + c->bkeys_cap - sp.leaf_lo : nullptr;
    hipLaunchKernelGGL((k_finalize<LEAF, K>), dim3((unsigned)blocks), dim3(256), 0, s, keys, sp, L, leaf_start, c->d_state,
                       params, maxerr, run, err, count, rows, c->d_partials, bn, bp);
    // (the last kernel also copies the device state into the pinned host copy: a separate 100-byte
    // copy command would cost ~15 us of the call)
    hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, s, c->d_partials, (int)blocks, c->d_state, c->h_state_dev + (c->stream_mode ? c->stream_slot : 0));
    if (sigma_giants) {
      // (behind the host fit of giant leaves, giant_epilogue: every leaf finalized once more -- ~25 us -- and the aggregates)
      StatsPartial* const part = c->d_partials;
      DevState* const dst = c->d_state;
      DevState* const hcopy = c->h_state_dev;
      c->refinalize_fn = [=]() {
        hipLaunchKernelGGL((k_finalize<LEAF, K>), dim3((unsigned)blocks), dim3(256), 0, s, keys, sp, L, leaf_start, dst, params, maxerr, run, err, count, rows, part, bn, bp);
        hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, s, part, (int)blocks, dst, hcopy);
      };
    }
  }
  mark();
  if (pl >= 0) HIPCHK(c, hipEventRecord(c->ev[9], s));
  HIPCHK(c, hipGetLastError());
  return RMI_OK;
}

// The giant leaves of the last launch (k_list recorded them: containers of more than host_min points).  The exact fit of
// a leaf is a sequential chain of its length (linear.rs:24-34): ~28 ns per point on a wave, ~4 on a host core with the
// reciprocal form of the step (rmi_root_host.h).  So: their keys come to the host, one thread per leaf fits them in
// reference order, the coefficients go back, and a short device epilogue runs their error pass (k_list_tail in
// stretches), their finalize and the aggregates again.  books-shaped 200 M keys / 262 144 leaves (one leaf of 2.5 M
// keys): 78 ms -> ~13 ms per training, same bits.
// (the chains on host threads, given the list: the keys come over on `cs` -- the context's stream, or a stream of its own
//  while that one is still busy with k_list)
template <typename K>
static int giant_host_fit(rmi_hip_ctx* c, const GiantLeaf* list, uint64_t cnt, hipStream_t cs) {
  c->giant_list.assign(list, list + cnt);
  c->giant_ab.assign(2 * cnt, 0.0);
  const std::vector<GiantLeaf>& g = c->giant_list;
  std::vector<double>& ab = c->giant_ab;
  const K* keys = (const K*)c->lp.keys;
  // every container's keys on their way first (one queue of copies, ONE synchronisation), the longest chain first in the fit
  std::vector<std::vector<K>> bufs(cnt);
  std::vector<uint64_t> order(cnt);
  for (uint64_t i = 0; i < cnt; i++) { bufs[i].resize(g[i].hi - g[i].lo + 1); order[i] = i; }
  std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return bufs[a].size() > bufs[b].size(); });
  for (uint64_t q = 0; q < cnt; q++) {
    const uint64_t i = order[q];
    HIPCHK(c, hipMemcpyAsync(bufs[i].data(), keys + g[i].lo, bufs[i].size() * sizeof(K), hipMemcpyDeviceToHost, cs));
  }
  HIPCHK(c, hipStreamSynchronize(cs));
  // (no early return below: every thread is joined before the function is left)
  std::vector<int> rcs(cnt, RMI_OK);
  std::atomic<uint64_t> next{0};
  const unsigned hw = std::thread::hardware_concurrency();
  uint64_t nth = hw ? hw : 16;
  if (nth > cnt) nth = cnt;
  if (nth > 128) nth = 128;
  auto work = [&]() {
    for (;;) {
      const uint64_t q = next.fetch_add(1);
      if (q >= cnt) return;
      const uint64_t i = order[q];
      rcs[i] = rmi_host::leaf_slr<K>(bufs[i].data(), bufs[i].size(), g[i].lo, g[i].y0, &ab[2 * i], &ab[2 * i + 1]);
    }
  };
  std::vector<std::thread> th;
  for (uint64_t t = 1; t < nth; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  for (uint64_t i = 0; i < cnt; i++)
    if (rcs[i] != RMI_OK) { set_err(c, "%s", rmi_hip_strerror(rcs[i])); return rcs[i]; }
  return RMI_OK;
}

template <typename K>
static int giant_epilogue(rmi_hip_ctx* c) {
  hipStream_t s = c->stream;
  const DevState& st0 = *c->h_state;
  if (st0.giant_count > c->giant_cap) { set_err(c, "internal: giant-leaf list overflow"); return RMI_ERR_HIP; }
  const uint64_t cnt = st0.giant_count;
  const K* keys = (const K*)c->lp.keys;
  if (!(c->giant_fitted && c->giant_list.size() == cnt)) {           // (not fitted beside k_list: now)
    std::vector<GiantLeaf> g(cnt);
    HIPCHK(c, hipMemcpy(g.data(), c->d_giant, cnt * sizeof(GiantLeaf), hipMemcpyDeviceToHost));
    const int frc = giant_host_fit<K>(c, g.data(), cnt, s);
    if (frc != RMI_OK) return frc;
  }
  const std::vector<GiantLeaf>& g = c->giant_list;
  for (uint64_t i = 0; i < cnt; i++)
    HIPCHK(c, hipMemcpyAsync(c->lp.params + 2 * g[i].j, &c->giant_ab[2 * i], 16, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(&c->d_state->seg_count, 0, 8, s));
  HIPCHK(c, hipMemsetAsync(c->d_flist_cnt + 2 * SG_REGIONS, 0, 8, s));
  SgList fl; fl.ids = c->d_flist; fl.cnt = c->d_flist_cnt; fl.cap = c->flist_cap;
  hipLaunchKernelGGL(k_giant_segments, dim3(16), dim3(64), 0, s, c->d_giant, c->lp.leaf_start, c->d_state, c->d_segs, c->lp.maxerr, c->lp.run);
  hipLaunchKernelGGL((k_list_tail<K>), dim3(2048), dim3(64), 0, s, keys, c->lp.sp, c->lp.leaf_start, c->d_state, c->lp.params, fl, c->d_segs, c->lp.maxerr, c->lp.run);
  if (c->refinalize_fn) {                                           // one-pass modes: k_finalize over all leaves + the aggregates again
    c->refinalize_fn();
    c->refinalize_fn = nullptr;
    if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[9], s));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    return RMI_OK;
  }
  StatsPartial* first = c->d_partials + c->lp.waves + SG_REGIONS;    // the records of the launch's own k_finalize_listed
  hipLaunchKernelGGL((k_finalize_listed<K>), dim3(FL_BLOCKS), dim3(FL_THREADS), 0, s, keys, c->lp.sp, c->lp.L, c->lp.leaf_start, c->d_state, c->lp.params,
                     c->lp.maxerr, c->lp.run, c->lp.err, c->lp.count, c->lp.rows, fl, first, (unsigned int)FL_BLOCKS, first + FL_BLOCKS,
                     c->d_flist_cnt + 2 * SG_REGIONS, c->d_state, c->h_state_dev, (const GiantLeaf*)c->d_giant, ~0ull);
  if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[9], s)); // (the device time of the call covers the epilogue, host fit included)
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(s));
  return RMI_OK;
}

// The giant list is out before k_list starts (k_giant_scan + a copy into pinned memory + an event): the host walks those
// chains while the device fits the other listed leaves.
static int giant_early_fit(rmi_hip_ctx* c) {
  HIPCHK(c, hipEventSynchronize(c->ev_giant));
  const uint64_t gc = c->h_giant[0];
  if (gc == 0 || gc > rmi_hip_ctx::GIANT_EARLY_MAX || gc > c->giant_cap) return RMI_OK;   // (none, or more than the early list holds: giant_epilogue)
  if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  const GiantLeaf* gl = reinterpret_cast<const GiantLeaf*>(c->h_giant + 1);
  int rc;
  switch (c->dtype) {
    case RMI_KEY_U64: rc = giant_host_fit<uint64_t>(c, gl, gc, c->copy_stream); break;
    case RMI_KEY_U32: rc = giant_host_fit<uint32_t>(c, gl, gc, c->copy_stream); break;
    default: rc = giant_host_fit<double>(c, gl, gc, c->copy_stream); break;
  }
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  c->giant_fitted = true;
  return RMI_OK;
}

template <int ROOT, typename K>
static int dispatch_leaf(rmi_hip_ctx* c, const RootP& rp, int leaf_kind, uint64_t L) {
  switch (leaf_kind) {
    case RMI_MODEL_LINEAR: case RMI_MODEL_ROBUST_LINEAR: return launch_pipeline<ROOT, K_LINEAR, K>(c, rp, L);
    case RMI_MODEL_LINEAR_SPLINE: return launch_pipeline<ROOT, K_LINEAR_SPLINE, K>(c, rp, L);
    case RMI_MODEL_CUBIC: return launch_pipeline<ROOT, K_CUBIC, K>(c, rp, L);
    default: return RMI_ERR_UNSUPPORTED_MODEL;
  }
}

template <typename K>
static int dispatch_root(rmi_hip_ctx* c, int root_kind, const RootP& rp, int leaf_kind, uint64_t L) {
  switch (root_kind) {
    case RMI_MODEL_LINEAR: case RMI_MODEL_ROBUST_LINEAR: case RMI_MODEL_LINEAR_SPLINE:
      return dispatch_leaf<K_LINEAR, K>(c, rp, leaf_kind, L);     // all three predict with fma(beta, x, alpha)
    case RMI_MODEL_CUBIC: return dispatch_leaf<K_CUBIC, K>(c, rp, leaf_kind, L);
    case RMI_MODEL_LOGLINEAR: return dispatch_leaf<K_LOGLINEAR, K>(c, rp, leaf_kind, L);
    case RMI_MODEL_NORMAL: return dispatch_leaf<K_NORMAL, K>(c, rp, leaf_kind, L);
    case RMI_MODEL_RADIX: case RMI_MODEL_BRADIX: return dispatch_leaf<K_RADIX, K>(c, rp, leaf_kind, L);
    case RMI_MODEL_RADIX8: case RMI_MODEL_RADIX18: case RMI_MODEL_RADIX22: case RMI_MODEL_RADIX26: case RMI_MODEL_RADIX28:
      return dispatch_leaf<K_RADIX_TABLE, K>(c, rp, leaf_kind, L);
    default: return RMI_ERR_UNSUPPORTED_MODEL;
  }
}

extern "C" {

int rmi_hip_train_two_layer(rmi_hip_ctx* c, const rmi_hip_model_params* root, int leaf_kind,
                            uint64_t num_leaves, rmi_hip_result* out) {
  if (!c || !root || !out || num_leaves == 0 || num_leaves > (1ull << 31)) return RMI_ERR_BAD_ARG;
  if (!c->d_keys || c->n == 0) return RMI_ERR_NO_KEYS;
  if (root->kind < 0 || root->kind >= kNumModels || leaf_kind < 0 || leaf_kind >= kNumModels) return RMI_ERR_UNKNOWN_MODEL;
  if (must_be_top(leaf_kind)) return RMI_ERR_RESTRICTION;
  const int table_bits = rmi_host::radix_table_bits(root->kind);
  if (!root_on_device_path(root->kind) || leaf_kind > RMI_MODEL_ROBUST_LINEAR) return RMI_ERR_UNSUPPORTED_MODEL;
  if (table_bits > 0 && (c->h_table.size() != (1ull << table_bits) || root->ip[1] != (uint64_t)table_bits || !c->d_table))
    return RMI_ERR_BAD_ARG;                                    // no (matching) table in this context: rmi_hip_set_root_table
  c->last_L = 0;                                               // the arrays of the previous call are gone from here on
  c->generation++;
  // robust_linear as a leaf trims 0.01 % tails of each container (linear.rs:247-252): its own fit, then a linear leaf
  c->robust_leaf = (leaf_kind == RMI_MODEL_ROBUST_LINEAR);
  HIPCHK(c, hipSetDevice(c->device));
  const int ppl = leaf_kind == RMI_MODEL_CUBIC ? 4 : 2;
  uint64_t L_own = num_leaves;
  if (c->have_shard) {
    const Span& sp = c->shard;
    if (sp.leaf_hi > num_leaves || (c->stream_mode ? sp.rd_hi > c->n : sp.rd_hi - sp.rd_lo != c->n)) return RMI_ERR_BAD_ARG;
    L_own = c->stream_mode ? num_leaves : sp.leaf_hi - sp.leaf_lo;
  }
  int rc = ensure_outputs(c, L_own, ppl);
  if (rc) return rc;
  RootP rp;
  rp.p0 = root->p[0]; rp.p1 = root->p[1]; rp.p2 = root->p[2]; rp.p3 = root->p[3];
  rp.prefix = (uint32_t)root->ip[0]; rp.bits = (uint32_t)root->ip[1];
  rp.L = num_leaves;
  rp.table = nullptr;
  rp.cap = num_leaves - 1; rp.oob_cap = num_leaves - 1;        // two_layer.rs:45-49
  if (root->kind == RMI_MODEL_BRADIX) {
    // balanced_radix.rs:104-116 on the radix kernels: clamp-high is the radix function with a lower
    // cap and no bounds error; clamp-low with a clamp no prediction reaches (every fit of
    // balanced_radix.rs:63 in a release build, see rmi_root_host.h) sends every key to leaf 0.
    const uint64_t clamp = root->ip[2];
    rp.oob_cap = ~0ull;
    if (root->ip[3]) rp.cap = clamp < num_leaves - 1 ? clamp : num_leaves - 1;
    else if (root->ip[1] < 64 && (clamp >> root->ip[1]) != 0) rp.cap = 0;
    else return RMI_ERR_UNSUPPORTED_MODEL;
  }
  if (table_bits > 0) {                                        // radix.rs:125-131: slot = ((x << p) >> p) >> shift
    rp.table = c->d_table;
    rp.bits = (root->ip[0] + root->ip[1] > 64) ? 0u : (uint32_t)(64 - (root->ip[0] + root->ip[1]));
  }
  switch (c->dtype) {
    case RMI_KEY_U64: rc = dispatch_root<uint64_t>(c, root->kind, rp, leaf_kind, num_leaves); break;
    case RMI_KEY_U32: rc = dispatch_root<uint32_t>(c, root->kind, rp, leaf_kind, num_leaves); break;
    case RMI_KEY_F64: rc = dispatch_root<double>(c, root->kind, rp, leaf_kind, num_leaves); break;
    default: rc = RMI_ERR_BAD_ARG;
  }
  if (rc) return rc;
  if (c->defer_sync) return RMI_OK;                            // (rmi_hip_train_sharded goes on from here)
  if (c->giant_early && !c->tail_armed) { rc = giant_early_fit(c); if (rc) return rc; }   // (one-pass modes: k_list is already running)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->regs_listed_fn) {
    // pipeline 4 published its record without the groups k_leaf_regs listed: they run now, and the aggregates are combined once more
    if (c->h_state->regs_listed > 0) {
      const unsigned int listed_groups = c->h_state->regs_listed;
      rc = c->regs_listed_fn(listed_groups);
      c->regs_listed_fn = nullptr;
      if (rc) return rc;
      if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->h_state->regs_listed = listed_groups;                       // (the second record's copy of the state holds it as well; kept explicit)
    }
    c->regs_listed_fn = nullptr;
  }
  if (c->tail_armed) {
    // k_lane_reduce has published the result: final unless k_leaf_lanes handed leaves to the list kernels
    c->tail_armed = false;
    if (c->h_state->pending > 0) {
      rc = c->tail_fn();
      c->tail_fn = nullptr;
      if (rc) return rc;
      if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
      if (c->giant_early) { rc = giant_early_fit(c); if (rc) return rc; }
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->tail_fn = nullptr;
  }
  if (c->giant_armed && c->h_state->giant_count > 0 && !c->h_state->err_flags) {
    switch (c->dtype) {
      case RMI_KEY_U64: rc = giant_epilogue<uint64_t>(c); break;
      case RMI_KEY_U32: rc = giant_epilogue<uint32_t>(c); break;
      default: rc = giant_epilogue<double>(c); break;
    }
    if (rc) return rc;
  }
  return finish_train(c, leaf_kind, num_leaves, out);
}

}  // extern "C"

// After the stream has been synchronised: the reference's panics as error codes, the result record, timings.
static int finish_train(rmi_hip_ctx* c, int leaf_kind, uint64_t num_leaves, rmi_hip_result* out) {
  int rc = RMI_OK;
  const int ppl = leaf_kind == RMI_MODEL_CUBIC ? 4 : 2;
  const uint64_t L_own = c->have_shard ? c->shard.leaf_hi - c->shard.leaf_lo : num_leaves;
  const DevState& st = *c->h_state;
  if (st.err_flags) {
    if (st.err_flags & EF_NON_MONOTONE) rc = RMI_ERR_NON_MONOTONE;
    else if (st.err_flags & EF_ROOT_OOB) rc = RMI_ERR_ROOT_OUT_OF_BOUNDS;
    else if (st.err_flags & EF_DEGENERATE_SPLIT) rc = RMI_ERR_DEGENERATE_SPLIT;
    else if (st.err_flags & EF_NEG_VARIANCE) rc = RMI_ERR_NEGATIVE_VARIANCE;
    else if (st.err_flags & EF_ROBUST_TOO_SMALL) rc = RMI_ERR_ROBUST_TOO_SMALL;
    else if (st.err_flags & EF_CUBIC_DEGENERATE) rc = RMI_ERR_CUBIC_DEGENERATE;
    else if (st.err_flags & EF_LIST_OVERFLOW) { set_err(c, "internal: a hand-over list of the leaf kernels overflowed (flags 0x%x)", st.err_flags); return RMI_ERR_HIP; }
    set_err(c, "%s", rmi_hip_strerror(rc));
    return rc;
  }
  c->last_L = L_own; c->last_ppl = ppl;
  c->lean_last_target = st.last_target;
  if (c->last_scan && !c->stream_mode) {
    c->scan_hint_epoch = c->keys_epoch; c->scan_hint_L = L_own; c->scan_hint_n = (unsigned int)(st.scan_listed < 0xFFFFFFFFull ? st.scan_listed : 0xFFFFFFFFull);
    if (st.scan_listed > 512ull) { c->scan_skew_epoch = c->keys_epoch; c->scan_skew_L = L_own; }
  }
  c->lean_leaf_lo = c->have_shard ? c->shard.leaf_lo : 0;
  std::memset(out, 0, sizeof *out);
  out->generation = c->generation;
  const uint64_t n_glob = c->have_shard ? c->shard.n : c->n;
  out->num_rows = n_glob; out->num_leaves = num_leaves; out->leaf_kind = leaf_kind;
  out->shard_leaf_lo = c->have_shard ? c->shard.leaf_lo : 0; out->shard_leaves = L_own;
  out->sum_n_err = st.sum_n_err; out->sum_l2 = st.sum_l2; out->sum_log2 = st.sum_log2;
  out->params_per_leaf = ppl; out->row_bytes = (uint64_t)ppl * 8 + 8;
  out->model_max_error = st.max_err; out->model_max_error_idx = st.max_err_idx;
  out->model_avg_error = (double)st.sum_n_err / (double)n_glob;
  out->model_avg_l2_error = st.sum_l2;
  out->model_avg_log2_error = st.sum_log2 / (double)n_glob;
  out->model_max_log2_error = std::log2((double)st.max_err);
  out->split_idx = st.split_idx; out->split_target = st.split_target;
  out->long_leaves = c->last_lanes ? st.flag_count : st.long_count;
  if (c->last_sigma && (st.flag_count - st.merged_count) * 4 > L_own) {   // most leaves went through the list kernels: see hint_epoch
    if (c->hint_epoch != c->keys_epoch || c->hint_mode != c->fit_mode) { c->hint_epoch = c->keys_epoch; c->hint_mode = c->fit_mode; c->hint_n = 0; }
    c->hint_L[c->hint_n % 8] = L_own; c->hint_n++;
  }
  if (c->last_regs && c->regs_backoff && st.regs_listed >= (L_own + 63) / 64) c->last_regs = false;   // every group listed (regs_dups): k_leaf_lanes_listed did the work -- pipeline 3
  if (c->regs_backoff && c->regs && (uint64_t)st.regs_listed * 4 > (L_own + 63) / 64) {   // most groups went on the list: see regs_off
    if (c->regs_off_epoch != c->keys_epoch) { c->regs_off_epoch = c->keys_epoch; c->regs_off_n = 0; }
    c->regs_off_L[c->regs_off_n % 8] = L_own; c->regs_off_n++;
  }
  out->fit_mode_used = c->last_sigma ? (c->last_spline ? RMI_FIT_USED_ONEPASS_EXACT : c->fit_mode) : 0;
  out->exact_leaves = c->last_sigma ? st.flag_count - st.merged_count : 0;
  out->merged_leaves = c->last_sigma ? (int32_t)(st.merged_count < 0x7fffffffull ? st.merged_count : 0x7fffffffull) : 0;
  out->guard_leaves = c->last_sigma ? st.guard_count : 0;
  float ms = 0.f;
  if (c->profile_level >= 0) HIPCHK(c, hipEventElapsedTime(&ms, c->ev[8], c->ev[9]));
  out->device_ns = (uint64_t)((double)ms * 1e6);
  for (int k = 0; k < (c->profile_level >= 2 ? 5 : c->profile_level); k++) {
    float m2 = 0.f;
    if (hipEventElapsedTime(&m2, c->ev[k], c->ev[k + 1]) == hipSuccess) out->kernel_ns[k] = (uint64_t)((double)m2 * 1e6);
  }
  return RMI_OK;
}

extern "C" {

}  // extern "C"

// alpha, beta, error of every leaf from its row, its count from the bucket table (+ the last key's second visit, Q7)
static __global__ void __launch_bounds__(256) k_lean_arrays(const unsigned char* __restrict__ rows, const unsigned long long* __restrict__ leaf_start, uint64_t L_own,
                                                            uint64_t leaf_lo, unsigned long long last_target, double* __restrict__ params,
                                                            unsigned long long* __restrict__ err, unsigned long long* __restrict__ count) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= L_own) return;
  const double* rp = reinterpret_cast<const double*>(rows + j * 24);
  params[2 * j] = rp[0]; params[2 * j + 1] = rp[1];
  err[j] = *reinterpret_cast<const unsigned long long*>(rows + j * 24 + 16);
  count[j] = leaf_start[j + 1] - leaf_start[j] + ((leaf_lo + j == last_target) ? 1ull : 0ull);
}
static int lean_fill(rmi_hip_ctx* c) {
  if (!c->last_lean || c->lean_derived || !c->last_L) return RMI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t leaf_lo = c->lean_leaf_lo;
  hipLaunchKernelGGL(k_lean_arrays, dim3((unsigned)((c->last_L + 255) / 256)), dim3(256), 0, c->stream, (const unsigned char*)c->last_rows, (const unsigned long long*)c->d_leaf_start,
                     c->last_L, leaf_lo, c->lean_last_target, c->d_params, c->d_err, c->d_count);
  HIPCHK(c, hipGetLastError());
  c->lean_derived = true;
  return RMI_OK;
}

extern "C" {

static int dl(rmi_hip_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || !dst) return RMI_ERR_BAD_ARG;
  if (!c->last_L) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (src == (const void*)c->d_params || src == (const void*)c->d_err || src == (const void*)c->d_count) { const int lrc = lean_fill(c); if (lrc) return lrc; }
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RMI_OK;
}
int rmi_hip_download_leaf_params(rmi_hip_ctx* c, double* o) { return dl(c, o, c ? c->d_params : nullptr, c ? c->last_L * c->last_ppl * 8 : 0); }
int rmi_hip_download_leaf_errors(rmi_hip_ctx* c, uint64_t* o) { return dl(c, o, c ? c->d_err : nullptr, c ? c->last_L * 8 : 0); }
int rmi_hip_download_leaf_counts(rmi_hip_ctx* c, uint64_t* o) { return dl(c, o, c ? c->d_count : nullptr, c ? c->last_L * 8 : 0); }
int rmi_hip_download_leaf_starts(rmi_hip_ctx* c, uint64_t* o) { return dl(c, o, c ? c->d_leaf_start : nullptr, c ? (c->last_L + 1) * 8 : 0); }
int rmi_hip_download_rows(rmi_hip_ctx* c, void* o) { return dl(c, o, c ? c->last_rows : nullptr, c ? c->last_L * (c->last_ppl * 8 + 8) : 0); }
int rmi_hip_download_checked(rmi_hip_ctx* c, int what, uint64_t generation, void* o, uint64_t capacity) {
  if (!c || !o || !c->last_L || generation != c->generation) return RMI_ERR_BAD_ARG;
  const void* src = nullptr;
  uint64_t bytes = 0;
  switch (what) {
    case RMI_DL_PARAMS: src = c->d_params; bytes = c->last_L * c->last_ppl * 8; break;
    case RMI_DL_ERRORS: src = c->d_err; bytes = c->last_L * 8; break;
    case RMI_DL_COUNTS: src = c->d_count; bytes = c->last_L * 8; break;
    case RMI_DL_STARTS: src = c->d_leaf_start; bytes = (c->last_L + 1) * 8; break;
    case RMI_DL_ROWS: src = c->last_rows; bytes = c->last_L * (c->last_ppl * 8 + 8); break;   // (the rows of the last training, wherever they were written)
    default: return RMI_ERR_BAD_ARG;
  }
  if (capacity < bytes) return RMI_ERR_BAD_ARG;
  return dl(c, o, src, bytes);
}
void* rmi_hip_device_rows(rmi_hip_ctx* c) { return c ? (c->last_rows ? const_cast<void*>(c->last_rows) : (c->d_rows_ext ? c->d_rows_ext : (void*)c->d_rows)) : nullptr; }

}  // extern "C"

#include "rmi_multi.inc.h"

static rmi_hip_multi* multi_of(rmi_hip_ctx* c) {
  if (!c->multi) c->multi = new rmi_hip_multi();
  return c->multi;
}
static void free_multi(rmi_hip_ctx* c) {
  rmi_hip_multi* m = c->multi;
  if (!m) return;
  if (m->comm && rmi_multi::api().CommDestroy) (void)rmi_multi::api().CommDestroy(m->comm);
  if (m->d_rows_full) (void)hipFree(m->d_rows_full);
  for (int r = 0; r < 64; r++) {
    if (!m->peer_open[r] || r == m->rank) continue;
    if (m->peer_rows[r]) (void)hipIpcCloseMemHandle(m->peer_rows[r]);
    if (m->peer_mail[r]) (void)hipIpcCloseMemHandle(m->peer_mail[r]);
  }
  if (m->d_rows2) (void)hipFree(m->d_rows2);
  if (m->d_mail) (void)hipFree(m->d_mail);
  if (m->d_peer_rows) (void)hipFree(m->d_peer_rows);
  if (m->d_peer_mail) (void)hipFree(m->d_peer_mail);
  if (m->d_stats_all) (void)hipFree(m->d_stats_all);
  if (m->h_stats_all) (void)hipHostFree(m->h_stats_all);
  delete m;
  c->multi = nullptr;
}
