/*
 * rmi_hip.h -- C ABI of the MI355X-native replacement for the leaf-fitting hot path of
 * learnedsystems/RMI's `rmi_lib::train` (two-layer RMIs).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  A Rust caller
 * binds it with `extern "C"` (see INTEGRATION.md for the stub that replaces the body of
 * rmi_lib::train::two_layer::train_two_layer).  All citations are file:line in the reference
 * repository (/root/reference at build time).
 *
 * What each entry point replaces:
 *   rmi_hip_upload_keys / rmi_hip_attach_device_keys
 *       RMITrainingData<T> over the mmap'd key file (src/load.rs:21-95, 132-157;
 *       rmi_lib/src/models/mod.rs:233-317).  Keys stay resident in HBM across many train calls
 *       (the optimizer / param-grid callers, optimizer.rs:220-231, src/main.rs:241-248).
 *   rmi_hip_fit_root
 *       `train_model(layer1_model, data)` with scale = L/N (two_layer.rs:109-110; factory
 *       train/mod.rs:35-57).
 *   rmi_hip_train_two_layer
 *       two_layer.rs:126-287: bucketing of every key with the root model, split for the 2-way
 *       join (:130-175), build_models_from (:20-99) incl. the per-leaf fits (linear.rs:12-59,
 *       linear_spline.rs:13-35, cubic_spline.rs:18-137), LowerBoundCorrection::new
 *       (lower_bound_correction.rs:92-137), empty-leaf fix (:185-197), last-level error pass
 *       (:207-217), lower-bound widening (:226-259) and the aggregate statistics (:267-287).
 *   rmi_hip_download_* / rmi_hip_result
 *       the fields of TrainedRMI (train/mod.rs:18-33) that codegen::output_rmi consumes
 *       (codegen.rs:450-788).  `rows` is byte-for-byte the reference's L1_PARAMETERS file
 *       (codegen.rs:288-315, 164-182; models/mod.rs:613-651).
 *
 * Error behaviour: the reference panics (process abort); this ABI never aborts -- every
 * reference panic on this path is mapped to a negative return code (enum below).
 *
 * Threading: a context is not thread-safe; use one context per caller thread (the reference
 * calls train() concurrently from Rayon workers on shared read-only data -- here each worker
 * would own a context that attaches the same device key buffer).
 */
#ifndef RMI_HIP_H
#define RMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMI_HIP_ABI_VERSION 6

/* src/load.rs:15-19 */
enum rmi_hip_key_dtype { RMI_KEY_U64 = 0, RMI_KEY_U32 = 1, RMI_KEY_F64 = 2 };

/* model registry, train/mod.rs:37-54.  On the device path: roots linear, linear_spline, cubic,
 * radix, robust_linear, loglinear, normal, the radix tables (ids 8-12) and bradix; leaves linear, linear_spline, cubic,
 * robust_linear (radix is top-only in the reference; radix tables as leaves are rejected).
 * The rest of the registry is recognised by name and rejected with RMI_ERR_UNSUPPORTED_MODEL. */
enum rmi_hip_model_kind {
  RMI_MODEL_LINEAR = 0,
  RMI_MODEL_LINEAR_SPLINE = 1,
  RMI_MODEL_CUBIC = 2,
  RMI_MODEL_RADIX = 3,
  RMI_MODEL_ROBUST_LINEAR = 4,
  RMI_MODEL_LOGLINEAR = 5,
  RMI_MODEL_NORMAL = 6,
  RMI_MODEL_LOGNORMAL = 7,
  RMI_MODEL_RADIX8 = 8,
  RMI_MODEL_RADIX18 = 9,
  RMI_MODEL_RADIX22 = 10,
  RMI_MODEL_RADIX26 = 11,
  RMI_MODEL_RADIX28 = 12,
  RMI_MODEL_BRADIX = 13,
  RMI_MODEL_HISTOGRAM = 14
};

enum rmi_hip_error {
  RMI_OK = 0,
  RMI_ERR_UNKNOWN_MODEL = -1,      /* panic!("Unknown model type") train/mod.rs:53 */
  RMI_ERR_RESTRICTION = -2,        /* MustBeTop / MustBeBottom, train/mod.rs:69-82 */
  RMI_ERR_NON_MONOTONE = -3,       /* assert!(target >= last_target) two_layer.rs:50; :144 */
  RMI_ERR_DEGENERATE_SPLIT = -4,   /* assert!(end_idx > start_idx) two_layer.rs:27 */
  RMI_ERR_ROOT_OUT_OF_BOUNDS = -5, /* two_layer.rs:45-48 (roots without bounds check) */
  RMI_ERR_BAD_ARG = -6,
  RMI_ERR_NEGATIVE_VARIANCE = -7,  /* linear.rs:48 */
  RMI_ERR_ROBUST_TOO_SMALL = -8,   /* linear.rs:248 */
  RMI_ERR_NUM_BITS = -9,           /* utils.rs:18 */
  RMI_ERR_CUBIC_DEGENERATE = -10,  /* cubic_spline.rs:50 / :61 unwrap() on None */
  RMI_ERR_UNSUPPORTED_MODEL = -11, /* registry name known, not on the device path */
  RMI_ERR_LAYERS = -12,            /* != 2 layers: panic!() train/mod.rs:125 */
  RMI_ERR_NO_KEYS = -13,
  RMI_ERR_HIP = -14,               /* HIP runtime failure; see rmi_hip_last_error */
  RMI_ERR_NO_DEVICE = -15,
  RMI_ERR_NO_RCCL = -16,           /* librccl.so could not be loaded (multi-GPU entry points only) */
  RMI_ERR_RCCL = -17               /* an RCCL call failed; see rmi_hip_last_error */
};

typedef struct rmi_hip_ctx rmi_hip_ctx;

/* Model parameters in `params()` order of each plugin: linear / linear_spline / robust_linear:
 * p = (alpha, beta) (linear.rs:99-101); cubic: p = (a, b, c, d) (cubic_spline.rs:160-167);
 * loglinear: p = (alpha, beta) (linear.rs:189-191); normal: p = (mean, stdev, scale) (normal.rs:94-100);
 * radix: ip = (prefix_len, bits) (radix.rs:60-62); bradix: ip = (prefix_len, bits, clamp)
 * (balanced_radix.rs:124-130) and ip[3] = 1 for the clamp-high function, 0 for clamp-low
 * (the `high` member, :17, which selects the emitted function, :132-164). */
typedef struct {
  int32_t kind;
  int32_t _pad;
  double p[4];
  uint64_t ip[4];
} rmi_hip_model_params;

/* Aggregates of two_layer.rs:267-287 + timings.  Per-leaf arrays stay in HBM until downloaded. */
typedef struct {
  uint64_t num_rows;              /* TrainedRMI::num_rmi_rows / num_data_rows */
  uint64_t num_leaves;            /* branching_factor */
  int32_t leaf_kind;
  int32_t params_per_leaf;        /* 2 or 4 */
  uint64_t row_bytes;             /* params_per_leaf*8 + 8 */
  double model_avg_error;
  double model_avg_l2_error;
  double model_avg_log2_error;
  double model_max_log2_error;
  uint64_t model_max_error;
  uint64_t model_max_error_idx;
  uint64_t split_idx;             /* two_layer.rs:132 (== num_rows when there is no split) */
  uint64_t split_target;          /* two_layer.rs:152-156 */
  /* shard bookkeeping: partial sums over the leaves of this shard, to be combined across ranks:
   * avg = sum(sum_n_err)/N, avg_l2 = sum(sum_l2), avg_log2 = sum(sum_log2)/N, max over max. */
  uint64_t shard_leaf_lo, shard_leaves;
  uint64_t sum_n_err;
  double sum_l2, sum_log2;
  uint64_t device_ns;             /* hipEvent time of all device work of this call */
  uint64_t kernel_ns[8];          /* per-kernel hipEvent times, see RMI_K_* */
  uint64_t long_leaves;           /* leaves too long for the lockstep fit pass, fitted one lane each
                                   * (skew diagnostic: each is a sequential chain of its own length) */
  /* one-pass mode (rmi_hip_set_fit_mode): did this call run it, how many leaves it handed to the exact
   * kernels (irregular leaves + guard), how many leaves the guard flagged, and (RMI_FIT_ONEPASS) how many long
   * leaves -- longer than a wave's LDS ring, or cut at a chunk border -- were fitted from merged partial sums */
  int32_t fit_mode_used;
  int32_t merged_leaves;
  uint64_t exact_leaves;
  uint64_t guard_leaves;
  uint64_t generation;            /* number of this train call on the context, for rmi_hip_download_checked */
} rmi_hip_result;

/* kernel_ns slots of the streaming passes (pipeline 2).  The default leaf-lane pipeline fills slot 0 with
 * k_leaf_lanes (bracketed behind the search), slot 1 with k_lane_reduce (or k_list when RMI_HIP_OPT_TAIL=0, then k_list_tail
 * and k_finalize_listed in 2 and 3); kernel_ns[7] is the exchange of rmi_hip_train_sharded. */
enum { RMI_K_BOUNDARIES = 0, RMI_K_FILL = 1, RMI_K_FIT = 2, RMI_K_ERR = 3, RMI_K_FINALIZE = 4 };

/* ---- lifetime ---- */
int rmi_hip_abi_version(void);
/* Which leaf kernels the context's last training ran: 5 = k_spline_scan (linear_spline leaves: one key-parallel read, v6), 4 = k_leaf_regs +
 * k_regs_finalize (linear leaves, one read of the keys: up to 640 keys per leaf on average, 4-byte keys at two waves per SIMD;
 * kernel_ns[0] = k_leaf_regs, [1] = the groups it listed for k_leaf_lanes + k_regs_finalize, [2] = k_lane_reduce), 3 = k_leaf_lanes,
 * 2 = the streaming passes (tiny key sets, 2^32 keys and more, cubic leaves, the one-pass modes).  (v5) */
int rmi_hip_last_pipeline(rmi_hip_ctx* ctx);
int rmi_hip_device_count(void);
int rmi_hip_create(int device_id, rmi_hip_ctx** out);
void rmi_hip_destroy(rmi_hip_ctx* ctx);
const char* rmi_hip_last_error(const rmi_hip_ctx* ctx);
const char* rmi_hip_strerror(int code);
/* Timing detail of rmi_hip_result: -1 = none (no event is recorded; device_ns = 0), 0 = device_ns only (default),
 * 1 = also kernel_ns[0], the first and dominant kernel of the call, 2 = every kernel group (RMI_K_*).  An event
 * between two kernels costs ~5 us of idle device time, so the detail is not free. */
int rmi_hip_set_profile_level(rmi_hip_ctx* ctx, int level);
/* How linear leaves (linear.rs:12-59) are fitted:
 *   RMI_FIT_EXACT (default, and the fastest mode): the reference's recurrence in the reference's order; coefficients,
 *     error integers and counts bit-identical to the reference.  Leaf boundaries by search, 64 leaves per wave in lockstep.
 *     Up to 640 keys a leaf on average: k_leaf_regs (rmi_regs.hip.h) -- the keys are read ONCE and stay in the lanes' registers
 *     between a leaf's fit and its error pass (the steps behind the 192nd of a long leaf come through the LDS ring a second time;
 *     4-byte keys: two waves per SIMD with the raw keys stashed), the leaf ends in k_regs_finalize.  Otherwise k_leaf_lanes
 *     (rmi_lanes.hip.h): error pass and finalize fused behind the fit, the keys are read twice (the second read largely from
 *     the Infinity Cache while the leaves are short).  Containers of more than 4 096 points one wave each, of more than 262 144
 *     points on a host core (RMI_HIP_HOST_MIN).  This is the mode every figure of merit is quoted in; DESIGN.md holds the
 *     table configuration -> kernels.
 *   RMI_FIT_ONEPASS_GUARDED (an opt-in FAST mode: its coefficients do NOT meet a 1e-9 relative tolerance on every
 *     leaf): ONE pass over the keys; a leaf's line from shifted sums (n, S dx, S dx^2,
 *     S dx dy) reduced in parallel, the error pass from LDS.  Bucket ids, per-leaf error integers and
 *     counts stay bit-identical: a leaf with any prediction closer to an integer than a bound on the
 *     distance between the two lines (guard_k times a first-order rounding bound, see rmi_sigma.hip.h),
 *     or that the sums do not describe (duplicate keys, the leaves at the split of two_layer.rs:130-175,
 *     first / last leaf, leaves longer than a tile), is re-fitted by the exact kernels.  Coefficients of
 *     the other leaves agree with the reference's to its own rounding noise (up to 2.3e-9 relative on 200M u64
 *     keys), not bit for bit.  The guard is a first-order bound with an empirical factor, not a proof: the integers are
 *     bit-identical on every key set tested (tests/test_gpu_sigma.py, test_gpu_lanes.py::test_adversarial_guard).
 *   RMI_FIT_ONEPASS: the least-squares line of the sums wherever the sums are defined.  No re-fit of guard-flagged
 *     leaves (counted in rmi_hip_result.guard_leaves); leaves longer than a wave's LDS ring, or cut at the border
 *     of two waves' chunks, are summed piecewise and merged (rmi_hip_result.merged_leaves) -- also the first leaf,
 *     the last leaf and the two leaves at the split of the reference's 2-way join, with their containers' rules.
 *     Only leaves with duplicate keys, or without spread, go to the exact kernels.  Error bounds are those of the
 *     emitted coefficients (the index is sound).  On well-conditioned keys a few bounds differ by one from the
 *     reference's; where the reference's own recurrence is dominated by its rounding noise (long leaves, keys far
 *     from 0 relative to their spread) its coefficients are not reproducible by sums and the bounds differ more.
 *     This is the mode for heavy-tailed key sets: one leaf of millions of keys costs the exact kernels its whole
 *     length as a sequential chain (28 ns per key) and this mode one more streaming read of its keys.
 * When a one-pass call hands more than a quarter of the leaves to the exact kernels (duplicate-heavy keys in either
 * mode; f64-collapsed keys in the guarded mode) the context remembers it for this key set, leaf count and mode, and the
 * following calls run the exact streaming passes instead (rmi_hip_result.fit_mode_used == 0; 1.1 ms against 13.8 ms
 * on 200 M duplicate-heavy keys).  rmi_hip_set_fit_mode and new keys forget it.
 * guard_k <= 0 keeps the current factor (default 2: the largest distance observed between the two lines, over
 * uniform / heavy-tailed / clustered key sets of 200 M keys, is 0.46 of the bound with factor 1).  Leaf kinds other
 * than `linear` ignore the mode. */
enum rmi_hip_fit_mode { RMI_FIT_EXACT = 0, RMI_FIT_ONEPASS_GUARDED = 1, RMI_FIT_ONEPASS = 2 };
/* linear_spline leaves (the line through the two end points of a container, linear_spline.rs:13-35) have no recurrence and are
 * exact in every mode (fit_mode_used 0): k_spline_scan (rmi_scan.hip.h) reads the keys ONCE, key-parallel -- bucketing scan,
 * end points, error pass and leaf ends in the same pass, for every root and key type. */
enum { RMI_FIT_USED_ONEPASS_EXACT = 3 };   /* (rounds 2-4: linear_spline leaves through the one-pass kernel; not reported any more) */
int rmi_hip_set_fit_mode(rmi_hip_ctx* ctx, int mode, double guard_k);
/* Run on a caller-provided hipStream_t (e.g. torch's current stream); NULL = context's own. */
int rmi_hip_set_stream(rmi_hip_ctx* ctx, void* hip_stream);

/* ---- registry ---- */
int rmi_hip_model_from_name(const char* name);       /* id, or RMI_ERR_UNKNOWN_MODEL */
const char* rmi_hip_model_name(int kind);
/* validate() + layer-count check of train() (train/mod.rs:59-85, 100-126) for a spec string
 * such as "linear,linear"; on success writes the two kinds. */
int rmi_hip_parse_spec(const char* spec, int* root_kind, int* leaf_kind);

/* ---- data ---- */
/* Copy n keys from host memory into HBM (buffer borrowed for the duration of the call). */
int rmi_hip_upload_keys(rmi_hip_ctx* ctx, const void* host_keys, uint64_t n, int dtype);
/* The same copy on a thread of the library, so that the caller can work on the host keys meanwhile -- the exact
 * root fit of a new key set (rmi_hip_fit_root with host_keys) is sequential host work that dwarfs the upload
 * (src/load.rs:132-157 + two_layer.rs:109-110 in the reference).  The buffer must stay valid until
 * rmi_hip_upload_wait returns (the copy's return code); no other call on the context in between. */
int rmi_hip_upload_keys_async(rmi_hip_ctx* ctx, const void* host_keys, uint64_t n, int dtype);
int rmi_hip_upload_wait(rmi_hip_ctx* ctx);
/* Borrow a device buffer that already holds the sorted keys (stays owned by the caller). */
int rmi_hip_attach_device_keys(rmi_hip_ctx* ctx, const void* device_keys, uint64_t n, int dtype);
uint64_t rmi_hip_num_keys(const rmi_hip_ctx* ctx);
/* The resident key buffer of a context (owned or borrowed), e.g. to attach it to further contexts:
 * independent trainings on one key set can then be in flight together, one context per caller thread
 * (optimizer.rs:220-231 trains its configurations with par_iter).  The buffer stays owned by `ctx`. */
int rmi_hip_key_buffer(const rmi_hip_ctx* ctx, const void** device_keys, uint64_t* n, int* dtype);
/* Many trainings on the resident keys in ONE call: what optimizer.rs:220-231 and the --param-grid mode (src/main.rs:241-248) do
 * with `par_iter` over their configurations.  `count` configurations (fitted root, leaf kind, branching factor), `in_flight`
 * (1 .. 16) at a time, each on a context of the library's own that borrows the keys (most configurations fill the GPU alone, but
 * those with few, long leaves are a handful of sequential chains: in flight together they cost the time of one).  results[i] holds
 * the aggregates of configs[i] (the per-leaf arrays are not kept); rcs (may be NULL) the per-configuration return codes -- where
 * the reference panics for ONE configuration the others still train.  Returns the first non-zero code, or RMI_OK; then
 * rmi_hip_last_error holds every failing worker's first message.  What rmi_hip_set_fit_mode / rmi_hip_set_profile_level have set
 * on `ctx` holds for every training of the batch.  RMI_ERR_BAD_ARG for a context that holds a shard (rmi_hip_set_shard) or a
 * streamed training, and for a radix-table root without its table.  The worker contexts stay with `ctx` for the next call (each
 * keeps per-leaf buffers of the largest leaf count it trained): rmi_hip_release_views frees them.  (v5; the rules v6) */
typedef struct {
  rmi_hip_model_params root;      /* from rmi_hip_fit_root / rmi_hip_fit_root_fast */
  int32_t leaf_kind;              /* RMI_MODEL_* */
  int32_t _pad;
  uint64_t num_leaves;
  const uint32_t* root_table;     /* radix-table roots (radix8/18/22/26/28): the hint table on the HOST (rmi_hip_download_root_table), else NULL */
  uint64_t root_table_entries;
} rmi_hip_train_config;
int rmi_hip_train_many(rmi_hip_ctx* ctx, const rmi_hip_train_config* configs, uint64_t count, int in_flight,
                       rmi_hip_result* results, int* rcs);
int rmi_hip_release_views(rmi_hip_ctx* ctx);
/* Synthetic sorted keys generated in HBM (bit-identical to rmi_amd/datagen.py):
 * generator 0 = uniform, 1 = uniform with duplicate runs.  Produces indices
 * [start, start+count) of the n_global-key array (so ranks can generate their own shard).
 * seed 0 = the documented default seed.  dtype: RMI_KEY_U64 or RMI_KEY_U32. */
int rmi_hip_generate_keys(rmi_hip_ctx* ctx, int generator, int dtype, uint64_t n_global,
                          uint64_t start, uint64_t count, uint64_t seed);
int rmi_hip_download_keys(rmi_hip_ctx* ctx, void* host_out);
const void* rmi_hip_device_keys(const rmi_hip_ctx* ctx);
/* Device self-test: the reciprocal-table division used inside the SLR recurrence against IEEE
 * division, on `trials` pseudo-random and near-midpoint operands; *mismatches must come back 0. */
int rmi_hip_selftest_div(rmi_hip_ctx* ctx, uint64_t trials, uint64_t seed, uint64_t* mismatches);
/* Host twin: the reciprocal form of the host root recurrence (no context, no GPU). */
int rmi_hip_selftest_host_div(uint64_t trials, uint64_t seed, uint64_t* mismatches);
/* The computed reciprocal used for counts beyond the table: checks RN(1/n) == 1.0/n for every
 * integer n in [n_lo, n_hi), 1 <= n_lo < n_hi <= 2^40. */
int rmi_hip_selftest_recip(rmi_hip_ctx* ctx, uint64_t n_lo, uint64_t n_hi, uint64_t* mismatches);
/* Achieved HBM read bandwidth (GB/s) of a read-only streaming kernel over the resident keys:
 * the measured denominator reported next to the 8 TB/s spec peak. */
int rmi_hip_measure_read_bandwidth(rmi_hip_ctx* ctx, int iters, double* gb_per_s);
/* ... by access pattern: 0 = grid-stride 16-byte loads (the function above), 1 = every wave reads contiguous pieces of 8 KB with
 * non-temporal loads, the pattern of the one-read kernels and the best read-only pattern found on this machine.  (v6) */
int rmi_hip_measure_read_bandwidth_ex(rmi_hip_ctx* ctx, int iters, int pattern, double* gb_per_s);

/* ---- multi-GPU sharding ----
 * A rank owns the contiguous leaf range [leaf_lo, leaf_hi) and the keys [key_lo, key_hi) that the
 * root maps to it (key_lo = first index with target >= leaf_lo; key_hi likewise for leaf_hi), and
 * keeps the keys [read_lo, read_hi) resident: the owned keys plus a halo -- on the left the whole
 * duplicate run of key[key_lo-1] and one more key, on the right at least key[key_hi].
 * rmi_hip_upload_keys / attach / generate then refer to exactly the keys [read_lo, read_hi).
 * split_idx / split_target: the 2-way-join split of two_layer.rs:132-156 (a global property;
 * UINT64_MAX if no key reaches leaf L/2).  All indices are global.  NULL clears the shard. */
typedef struct {
  uint64_t n_global;
  uint64_t read_lo, read_hi;
  uint64_t key_lo, key_hi;
  uint64_t leaf_lo, leaf_hi;
  uint64_t split_idx, split_target;
} rmi_hip_shard;
int rmi_hip_set_shard(rmi_hip_ctx* ctx, const rmi_hip_shard* shard);
/* Write the packed rows of this shard to a caller-owned device buffer (row 0 of the shard at
 * `device_rows`) instead of the context's own buffer, e.g. straight into the slot of an
 * all-gather buffer.  NULL restores the internal buffer. */
int rmi_hip_set_rows_output(rmi_hip_ctx* ctx, void* device_rows);

/* ---- multi-GPU inside the library: planner + RCCL all-gather of the rows ----
 * One context per GPU -- one process per GPU, or several contexts in one process.  What it replaces: the only
 * parallelism inside one reference training, the 2-way rayon::join of two_layer.rs:161-169; rmi_lib has no
 * distributed surface at all (rmi_lib/src/lib.rs:6-12), so a caller binds these four calls next to train():
 *   1. every rank:   rmi_hip_plan_shards(...)                   -> the same G shards everywhere (O(G log N) key probes)
 *   2. rank 0:       rmi_hip_comm_unique_id(id); hand the 128 bytes to the other ranks (any channel)
 *   3. every rank:   rmi_hip_comm_init(ctx, rank, G, id); rmi_hip_set_shard(ctx, &shards[rank]);
 *                    upload / attach / generate exactly the keys [read_lo, read_hi) of its shard
 *   4. every rank:   rmi_hip_train_sharded(ctx, root, leaf_kind, L, &result)   (as often as wanted)
 * After step 4 the full row buffer of every rank (rmi_hip_device_rows_full / rmi_hip_download_rows_full) holds
 * the L rows -- byte for byte what one GPU computes -- and `result` the aggregates of the whole model. */
#define RMI_HIP_COMM_ID_BYTES 128
/* key source of the planner: the bits of the key with global index i (u32 keys widened, f64 keys as their bits) */
typedef uint64_t (*rmi_hip_key_at_fn)(void* user, uint64_t index);
/* ctx may be NULL unless the root is a radix table (its hint table lives in the context).  num_leaves must be a
 * multiple of world.  out: `world` shards. */
int rmi_hip_plan_shards(const rmi_hip_ctx* ctx, const rmi_hip_model_params* root, int dtype, uint64_t n_global,
                        uint64_t num_leaves, int world, rmi_hip_key_at_fn key_at, void* user, rmi_hip_shard* out);
/* Closed form of the synthetic generators of rmi_hip_generate_keys (a key source for the planner when every rank
 * generates its own shard in HBM). */
int rmi_hip_generated_key(int generator, int dtype, uint64_t n_global, uint64_t seed, uint64_t index, uint64_t* key_bits);
/* `radix` / `linear_spline` roots (O(1) keys: radix.rs:18-39, linear_spline.rs:13-35) through a key source. */
int rmi_hip_fit_root_from_source(int root_kind, int dtype, uint64_t n_global, uint64_t num_leaves, rmi_hip_key_at_fn key_at,
                                 void* user, rmi_hip_model_params* out);
int rmi_hip_comm_unique_id(void* id_out /* RMI_HIP_COMM_ID_BYTES */);
int rmi_hip_comm_init(rmi_hip_ctx* ctx, int rank, int world, const void* id /* RMI_HIP_COMM_ID_BYTES; NULL with world == 1 */);
int rmi_hip_comm_info(rmi_hip_ctx* ctx, int* world, int* rank);   /* ncclCommCount / ncclCommUserRank of the context's communicator */
int rmi_hip_comm_destroy(rmi_hip_ctx* ctx);
int rmi_hip_train_sharded(rmi_hip_ctx* ctx, const rmi_hip_model_params* root, int leaf_kind, uint64_t num_leaves,
                          rmi_hip_result* out);
void* rmi_hip_device_rows_full(rmi_hip_ctx* ctx);
int rmi_hip_download_rows_full(rmi_hip_ctx* ctx, void* host_out, uint64_t capacity_bytes);
/* ---- the exchange as direct peer stores over xGMI, next to the RCCL all-gather ----
 * xGMI is a full mesh: a rank can store its 24 L / G bytes into all G - 1 peers at once (~ slice / 153 GB/s per link),
 * where a ring all-gather takes G - 1 dependent steps.  Every rank exports its (double-buffered) full row table and a
 * mailbox as IPC handles, imports its peers' (any channel carries the RMI_HIP_PEER_HANDLE_BYTES), and with
 * rmi_hip_set_exchange(ctx, RMI_EXCHANGE_DIRECT) rmi_hip_train_sharded ends with: one kernel that stores this rank's
 * slice into every peer's table, one that publishes the aggregates and an epoch flag in every peer's mailbox (system
 * scope), one that waits for the G flags of this epoch (bounded: 60 s, then RMI_ERR_HIP).  No RCCL needed.
 * Works across processes on ONE device too (how it is tested here); across devices it has NOT run yet -- it is opt-in,
 * and bench.py --exchange auto validates its table against the RCCL exchange before preferring it. */
#define RMI_HIP_PEER_HANDLE_BYTES 256
enum { RMI_EXCHANGE_RCCL = 0, RMI_EXCHANGE_DIRECT = 1 };
int rmi_hip_peer_export(rmi_hip_ctx* ctx, int rank, int world, int leaf_kind, uint64_t num_leaves, void* handle_out);
int rmi_hip_peer_import(rmi_hip_ctx* ctx, int peer_rank, const void* handle);
int rmi_hip_set_exchange(rmi_hip_ctx* ctx, int mode);

/* ---- root model ---- */
/* Fit the root exactly as the reference does.  `host_keys` may be NULL, in which case the keys
 * are read back from HBM for the order-dependent fits (linear / robust_linear / cubic). */
int rmi_hip_fit_root(rmi_hip_ctx* ctx, int root_kind, uint64_t num_leaves, const void* host_keys,
                     rmi_hip_model_params* out);

/* The host part alone (no context, no device), e.g. beside rmi_hip_upload_keys_async: the reference's fit of
 * linear / robust_linear / linear_spline / cubic / radix / loglinear / normal roots over a host array. */
int rmi_hip_fit_root_host(int root_kind, int dtype, const void* host_keys, uint64_t n, uint64_t num_leaves,
                          rmi_hip_model_params* out);

/* FAST root fit, opt-in: `linear` / `robust_linear` from parallel sums on the device
 * instead of the reference's sequential recurrence -- same points, coefficients within ~1e-12
 * relative, NOT bit-identical (a handful of keys next to leaf boundaries may change bucket; the leaf
 * path then trains, exactly, the RMI of THAT root, and the emitted index is as sound as any).  Other
 * root kinds fall through to rmi_hip_fit_root.  Milliseconds instead of ~6 ns per key on one core. */
int rmi_hip_fit_root_fast(rmi_hip_ctx* ctx, int root_kind, uint64_t num_leaves, rmi_hip_model_params* out);

/* min(L-1, root.predict_to_int(key)) evaluated on the host (two_layer.rs:49): lets a caller plan
 * leaf-aligned shard cuts with exactly the bucketing the kernels use.  key_bits: the key's bits. */
int rmi_hip_root_target(const rmi_hip_model_params* root, int dtype, uint64_t key_bits,
                        uint64_t num_leaves, uint64_t* out);
/* Radix-table roots (radix8/18/22/26/28 = RadixTable::new(data, bits), radix.rs:83-121): the
 * parameters are ip = (prefix_bits, table_bits) plus the hint table of 2^table_bits u32, which
 * lives in the context.  rmi_hip_fit_root stores it there; these calls read it back (for the
 * emitted L0_PARAMETERS file) or install one (a cached root, another rank of a multi-GPU job).
 * rmi_hip_train_two_layer uses the table of the context for such a root; rmi_hip_root_target
 * has no context and returns RMI_ERR_UNSUPPORTED_MODEL for them. */
int rmi_hip_root_table_entries(const rmi_hip_ctx* ctx, uint64_t* entries);
int rmi_hip_download_root_table(const rmi_hip_ctx* ctx, uint32_t* host_out);
int rmi_hip_set_root_table(rmi_hip_ctx* ctx, const uint32_t* host_table, uint64_t entries);

/* Error-bounded mode (`--bounded line_size`, train_bounded train/mod.rs:156-184): cache_fix
 * (cache_fix.rs:109-150) is a greedy, sequential pass over the unique u64 keys on the host; the RMI
 * is then trained, on the device as usual, over the spline keys (upload them as a key set of their
 * own).  The spline ((key, offset) pairs, 16 B each) stays in the context until the next call. */
int rmi_hip_cache_fix(rmi_hip_ctx* ctx, const uint64_t* host_keys, uint64_t n, uint64_t line_size, uint64_t* num_points);
int rmi_hip_download_cache_fix(const rmi_hip_ctx* ctx, uint64_t* host_out_pairs /* 2 * num_points */);

/* `linear` root fit fed with consecutive chunks of the global key array (same recurrence and
 * result as rmi_hip_fit_root; for data that is produced or held shard by shard). */
typedef struct rmi_hip_root_stream rmi_hip_root_stream;
int rmi_hip_root_stream_begin(int root_kind, int dtype, uint64_t n_global, uint64_t num_leaves,
                              rmi_hip_root_stream** out);
int rmi_hip_root_stream_push(rmi_hip_root_stream* rs, const void* host_keys, uint64_t count);
int rmi_hip_root_stream_finish(rmi_hip_root_stream* rs, rmi_hip_model_params* out);  /* frees rs */

/* ---- the hot path ---- */
int rmi_hip_train_two_layer(rmi_hip_ctx* ctx, const rmi_hip_model_params* root, int leaf_kind,
                            uint64_t num_leaves, rmi_hip_result* out);

/* Upload and train at once (replaces the reference's loader + train for a key set in host memory, src/load.rs:132-157,
 * train/mod.rs:100-126).  The keys go to HBM in chunks through two pinned staging buffers (host copy of chunk c+1 beside
 * the DMA of chunk c); the key set is cut into `chunks` leaf-aligned shards (rmi_hip_plan_shards on the host keys) and
 * shard s is trained as soon as its keys have arrived, behind the upload of the following ones.  When the call returns
 * the keys are resident (as after rmi_hip_upload_keys), the per-leaf arrays hold all `num_leaves` leaves and `out` the
 * aggregates of the whole model: the same results as upload + rmi_hip_train_two_layer, bit for bit in RMI_FIT_EXACT
 * (the f64 sums of the aggregates are recombined per shard: equal to 1e-12).  num_leaves must be a multiple of chunks
 * (1 .. RMI_STREAM_MAX_CHUNKS); the root has to be known (fit it from the host keys: rmi_hip_fit_root_host). */
#define RMI_STREAM_MAX_CHUNKS 64
int rmi_hip_train_streamed(rmi_hip_ctx* ctx, const void* host_keys, uint64_t n, int dtype, const rmi_hip_model_params* root,
                           int leaf_kind, uint64_t num_leaves, int chunks, rmi_hip_result* out);

/* ---- results (valid until the next train call on this context) ---- */
int rmi_hip_download_leaf_params(rmi_hip_ctx* ctx, double* host_out /* L*ppl */);
int rmi_hip_download_leaf_errors(rmi_hip_ctx* ctx, uint64_t* host_out /* L */);
int rmi_hip_download_leaf_counts(rmi_hip_ctx* ctx, uint64_t* host_out /* L */);
int rmi_hip_download_leaf_starts(rmi_hip_ctx* ctx, uint64_t* host_out /* L+1 */);
int rmi_hip_download_rows(rmi_hip_ctx* ctx, void* host_out /* L*row_bytes */);
/* The same downloads for callers that hold on to a result while the context trains on: `what` is one of
 * RMI_DL_*, `generation` the rmi_hip_result.generation of the call whose arrays are wanted, `capacity_bytes`
 * the size of host_out.  RMI_ERR_BAD_ARG if the context has trained again since (the arrays are gone) or
 * if the buffer is too small -- nothing is written then. */
enum { RMI_DL_PARAMS = 0, RMI_DL_ERRORS = 1, RMI_DL_COUNTS = 2, RMI_DL_STARTS = 3, RMI_DL_ROWS = 4 };
int rmi_hip_download_checked(rmi_hip_ctx* ctx, int what, uint64_t generation, void* host_out, uint64_t capacity_bytes);
/* Device pointer of the packed rows (for an all-gather over RCCL without a host bounce). */
void* rmi_hip_device_rows(rmi_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
