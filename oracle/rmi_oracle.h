/*
 * rmi_oracle.h -- CPU restatement of the reference's two-layer RMI training path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped library (librmi_hip.so)
 * never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (learnedsystems/RMI, Rust) holds no golden numbers
 * for this path and cannot be compiled in this image (no cargo/rustc).  The oracle is
 * a line-by-line restatement of the cited Rust sources and is pinned only by the
 * reference's stale in-file KATs and its end-to-end soundness property
 * (see tests/test_oracle_kats.py, tests/test_oracle_property.py), by independent
 * transcriptions of single functions in those tests, and by a second, independent
 * restatement of the whole trainer in Python (tests/pyref.py, tests/test_pyref.py).
 * None of that is an execution of the reference itself.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef RMI_ORACLE_H
#define RMI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* key dtypes: src/load.rs:15-19 (UINT64 / UINT32 / FLOAT64) */
enum { ORC_KEY_U64 = 0, ORC_KEY_U32 = 1, ORC_KEY_F64 = 2 };

/* model registry subset on the hot path: rmi_lib/src/train/mod.rs:37-54 */
enum {
  ORC_MODEL_LINEAR = 0,
  ORC_MODEL_LINEAR_SPLINE = 1,
  ORC_MODEL_CUBIC = 2,
  ORC_MODEL_RADIX = 3,
  ORC_MODEL_ROBUST_LINEAR = 4,
  ORC_MODEL_LOGLINEAR = 5,         /* LogLinearModel, linear.rs:152-210 (as a root) */
  ORC_MODEL_NORMAL = 6,            /* NormalModel, normal.rs:70-126 (as a root): p = (mean, stdev, scale) */
  /* RadixTable::new(data, bits), train/mod.rs:46-50: same ids as include/rmi_hip.h */
  ORC_MODEL_RADIX8 = 8,
  ORC_MODEL_RADIX18 = 9,
  ORC_MODEL_RADIX22 = 10,
  ORC_MODEL_RADIX26 = 11,
  ORC_MODEL_RADIX28 = 12,
  ORC_MODEL_BRADIX = 13            /* BalancedRadixModel, balanced_radix.rs */
};

/* return codes: 0 ok; negative = the reference would have panicked at the cited line */
enum {
  ORC_OK = 0,
  ORC_ERR_UNKNOWN_MODEL = -1,      /* train/mod.rs:53 */
  ORC_ERR_RESTRICTION = -2,        /* train/mod.rs:69-82 */
  ORC_ERR_NON_MONOTONE = -3,       /* two_layer.rs:50 */
  ORC_ERR_DEGENERATE_SPLIT = -4,   /* two_layer.rs:27 / :144 */
  ORC_ERR_ROOT_OUT_OF_BOUNDS = -5, /* two_layer.rs:45 */
  ORC_ERR_BAD_ARG = -6,
  ORC_ERR_NEGATIVE_VARIANCE = -7,  /* linear.rs:48 */
  ORC_ERR_ROBUST_TOO_SMALL = -8,   /* linear.rs:248 */
  ORC_ERR_NUM_BITS = -9,           /* utils.rs:18 */
  ORC_ERR_CUBIC_DEGENERATE = -10   /* cubic_spline.rs:50 / :61 (.unwrap() on None) */
};

/* A model: up to 4 f64 params (linear: alpha,beta; cubic: a,b,c,d) or 2 int params
 * (radix: prefix, bits).  Mirrors `params()` ordering of each plugin. */
typedef struct {
  int kind;
  double p[4];
  uint64_t ip[4];   /* bradix: (prefix, bits, clamp) = params() (balanced_radix.rs:124-130), ip[3] = `high` (:17) */
  /* radix tables (radix.rs:83-121): ip = (prefix_bits, table_bits), hint_table of 2^table_bits u32.
   * malloc'd by the fit (release with orc_model_free) or borrowed from the caller. */
  uint32_t* table;
  uint64_t table_len;
} orc_model;

void orc_model_free(orc_model* m);

/* Result of train_two_layer: the fields of TrainedRMI (train/mod.rs:18-33) that the
 * hot path produces, plus diagnostics (leaf_start = bucket assignment). */
typedef struct {
  uint64_t n;                /* num_rmi_rows == num_data_rows */
  uint64_t num_leaves;
  orc_model root;
  int leaf_kind;
  int params_per_leaf;       /* 2 (linear, linear_spline) or 4 (cubic) */
  double* leaf_params;       /* [num_leaves * params_per_leaf], caller-allocated */
  uint64_t* leaf_err;        /* [num_leaves] last_layer_max_l1s, caller-allocated */
  uint64_t* leaf_count;      /* [num_leaves] `n` of two_layer.rs:216 (incl. Q7), caller-allocated */
  uint64_t* leaf_start;      /* [num_leaves+1] first index with target>=j, caller-allocated or NULL */
  double model_avg_error;
  double model_avg_l2_error;
  double model_avg_log2_error;
  uint64_t model_max_error;
  uint64_t model_max_error_idx;
  double model_max_log2_error;
} orc_trained_rmi;

/* --- model-level entry points (fit on explicit (key, offset) pairs; used by KAT tests) --- */

/* Fit `kind` on `len` (key, y) pairs exactly as train_model() would on a
 * Vec<(K,usize)> provider (models/mod.rs:124-140) with the given scale. */
int orc_fit_pairs(int kind, int dtype, const void* keys, const uint64_t* ys, size_t len,
                  double scale, orc_model* out);

/* Model::predict_to_float / predict_to_int for a key given as raw bits of `dtype`. */
double orc_predict_to_float(const orc_model* m, int dtype, uint64_t key_bits);
uint64_t orc_predict_to_int(const orc_model* m, int dtype, uint64_t key_bits);

/* utils.rs:13-21 and :23-36 */
int orc_num_bits(uint64_t largest_target);
int orc_common_prefix_size(int dtype, const void* keys, size_t len);

/* --- the hot path --- */

/* Root fit only: train_model(layer1, data) with scale = L/N (two_layer.rs:109-110). */
int orc_fit_root(int root_kind, int dtype, const void* keys, uint64_t n, uint64_t num_leaves,
                 orc_model* out);

/* train_two_layer (two_layer.rs:101-306).  If root_override != NULL the root model is
 * taken from it instead of being fitted (used to test the leaf path in isolation).
 * threads: 1 = fully sequential; 2 = the two halves of two_layer.rs:161-169 run on two
 * threads (mirrors rayon::join). */
int orc_train_two_layer(int root_kind, int leaf_kind, int dtype, const void* keys, uint64_t n,
                        uint64_t num_leaves, const orc_model* root_override, int threads,
                        orc_trained_rmi* out);

/* per-key bucket ids: min(L-1, root.predict_to_int(key)) (two_layer.rs:49,134,210) */
int orc_bucket_ids(const orc_model* root, int dtype, const void* keys, uint64_t n,
                   uint64_t num_leaves, uint64_t* out_ids);

/* The reference tests' soundness property (tests/simple_model_wiki/main.cpp:26-41):
 * for every key, |lookup(key) - lower_bound(key)| <= err.  Returns the number of
 * violating keys (0 == sound); first violating index in *first_bad (or n). */
uint64_t orc_check_lookup_property(const orc_trained_rmi* rmi, int dtype, const void* keys,
                                   uint64_t n, uint64_t* first_bad);

/* --- error-bounded mode (`--bounded line_size`): cache_fix.rs, train/mod.rs:156-184 --- */

/* cache_fix(data, line_size) (cache_fix.rs:109-150): greedy spline over the unique keys (and their
 * predecessors key-1) such that interpolating between consecutive spline points lands in the right
 * `line_size`-aligned block.  *pairs_out = malloc'd [count][2] = (key, offset); orc_free() it.
 * Returns ORC_ERR_BAD_ARG where the reference asserts. */
int orc_cache_fix(const uint64_t* keys, uint64_t n, uint64_t line_size, uint64_t** pairs_out, uint64_t* count_out);
void orc_free(void* p);

/* The emitted lookup() of a bounded RMI (codegen.rs:396-447) -- `rmi` trained on the spline keys --
 * and the reference tests' property |lookup(key) - lower_bound(key)| <= line_size
 * (tests/cache_fix_wiki/main.cpp:26-44).  Returns the number of violating keys. */
uint64_t orc_check_bounded_property(const orc_trained_rmi* rmi, const uint64_t* spline_pairs, uint64_t num_spline,
                                    uint64_t line_size, const uint64_t* keys, uint64_t n, uint64_t* first_bad);

const char* orc_version(void);

#ifdef __cplusplus
}
#endif
#endif
