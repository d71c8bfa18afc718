// rmi_multi.inc.h -- multi-GPU form of the leaf path BEHIND the C ABI (included at the end of rmi_hip.hip).
//
// One context per GPU.  Given the root parameters, every leaf's container, fit and error bound depend only on a
// contiguous key range plus a one/two-key halo and on global indices: rank r owns the leaves
// [r L/G, (r+1) L/G) and the keys the root maps to them; rmi_hip_plan_shards finds the cuts by binary search with
// exactly the bucketing of the kernels; rmi_hip_train_sharded runs the usual kernels on the shard, with the rows
// written straight into the rank's slot of the full row buffer, and exchanges the rows with ONE ncclAllGather
// (RCCL over xGMI) on the context's stream; the aggregates are recombined exactly from per-shard partial sums.
// Replaces the reference's only parallelism inside one training, the 2-way rayon::join (two_layer.rs:161-169).
//
// RCCL is bound at run time (dlopen): a process that already holds RCCL (torch ships one) shares it, and the
// library keeps loading on a machine without RCCL (single-GPU use).
#include <dlfcn.h>

#include <mutex>
namespace rmi_multi {

struct NcclId { char internal[RMI_HIP_COMM_ID_BYTES]; };
typedef void* NcclComm;
struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(NcclComm, int*) = nullptr;
  int (*CommUserRank)(NcclComm, int*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather; }
};

static Api& api() {
  static Api a;
  static std::once_flag once;                                          // (several contexts may bring up communicators from different threads)
  std::call_once(once, [&]() {
    if (std::getenv("RMI_HIP_NO_RCCL")) return;                          // (tests: behave like a machine without RCCL)
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) { a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (a.handle) break; }   // already in the process
    if (!a.handle) {
      const char* paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
      for (const char* n : paths) { a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.handle) break; }
    }
    if (a.handle) {
      a.GetUniqueId = (int (*)(NcclId*))dlsym(a.handle, "ncclGetUniqueId");
      a.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))dlsym(a.handle, "ncclCommInitRank");
      a.CommDestroy = (int (*)(NcclComm))dlsym(a.handle, "ncclCommDestroy");
      a.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(a.handle, "ncclAllGather");
      a.GetErrorString = (const char* (*)(int))dlsym(a.handle, "ncclGetErrorString");
      a.CommCount = (int (*)(NcclComm, int*))dlsym(a.handle, "ncclCommCount");
      a.CommUserRank = (int (*)(NcclComm, int*))dlsym(a.handle, "ncclCommUserRank");
      a.GroupStart = (int (*)())dlsym(a.handle, "ncclGroupStart");
      a.GroupEnd = (int (*)())dlsym(a.handle, "ncclGroupEnd");
    }
  });
  return a;
}

// closed forms of the synthetic generators (rmi_hip_generate_keys / rmi_amd/datagen.py)
static inline uint64_t splitmix64_h(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace rmi_multi

struct rmi_hip_multi {
  rmi_multi::NcclComm comm = nullptr;
  int rank = 0, world = 1;
  unsigned char* d_rows_full = nullptr;     // L * row_bytes
  uint64_t rows_full_bytes = 0;                 // allocated
  uint64_t rows_last_bytes = 0;                 // num_leaves * row_bytes of the last sharded training
  unsigned char* d_stats_all = nullptr;     // world * 40 bytes
  unsigned char* h_stats_all = nullptr;     // pinned
  // direct exchange (peer stores): the row table double-buffered by the parity of the epoch, a mailbox (64 flags + 64 x 40 bytes)
  int exchange = 0;                         // RMI_EXCHANGE_*
  unsigned char* d_rows2 = nullptr;         // 2 x slot_bytes (exported)
  uint64_t rows2_slot = 0;
  unsigned char* d_mail = nullptr;          // fine-grained device memory (exported)
  unsigned char* peer_rows[64] = {};        // [rank]: that rank's d_rows2 as mapped here (own rank: local)
  unsigned char* peer_mail[64] = {};
  bool peer_open[64] = {};
  int peers = 0;
  unsigned char** d_peer_rows = nullptr;    // the two tables above for the kernels
  unsigned char** d_peer_mail = nullptr;
  unsigned long long epoch = 0;
  bool fuse_peer_stores = true;             // direct exchange: rows to the peers from k_leaf_lanes itself (false: k_peer_push behind the kernels -- more than 8 ranks)
};
// A rank's record in the exchange of the aggregates: the 6 words of DevState from max_err on (max_err, max_err_idx, sum_n_err,
// sum_l2, sum_log2, pending): the last one tells every rank whether SOME rank still has listed leaves to finish.
constexpr int RMI_STATS_WORDS = 6;
constexpr size_t RMI_STATS_BYTES = 8 * RMI_STATS_WORDS;
constexpr size_t RMI_MAIL_BYTES = 64 * 8 + 64 * RMI_STATS_BYTES;
struct rmi_peer_handle { hipIpcMemHandle_t rows, mail; uint64_t slot_bytes; int rank, world; };
static_assert(sizeof(rmi_peer_handle) <= RMI_HIP_PEER_HANDLE_BYTES, "handle size");

// this rank's slice into every peer's table (16-byte stores; a block reads its piece once and writes it G - 1 times)
__global__ void __launch_bounds__(256) k_peer_push(const uint4* __restrict__ src, uint64_t n16, unsigned char* const* __restrict__ peer_rows,
                                                   uint64_t byte_off, int rank, int world) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = src[i];
    for (int r = 0; r < world; r++)
      if (r != rank) reinterpret_cast<uint4*>(peer_rows[r] + byte_off)[i] = v;
  }
}
// the aggregates and the epoch flag into every mailbox (the kernel boundary before this launch has made the slice visible)
__global__ void __launch_bounds__(64) k_peer_signal(unsigned char* const* __restrict__ peer_mail, const unsigned long long* __restrict__ my_stats,
                                                    int rank, int world, unsigned long long epoch) {
  const int r = threadIdx.x;
  if (r >= world) return;
  unsigned long long* mail = reinterpret_cast<unsigned long long*>(peer_mail[r]);
  for (int q = 0; q < RMI_STATS_WORDS; q++) __hip_atomic_store(&mail[64 + rank * RMI_STATS_WORDS + q], my_stats[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&mail[rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wait for the `world` flags of this epoch in the own mailbox, then hand the aggregates to the host copy
__global__ void __launch_bounds__(64) k_peer_wait(unsigned long long* __restrict__ mail, int world, unsigned long long epoch,
                                                  unsigned long long* __restrict__ stats_all, rmi::DevState* __restrict__ st,
                                                  unsigned long long timeout_ticks) {
  const int r = threadIdx.x;
  if (r >= world) return;
  const unsigned long long t0 = wall_clock64();                      // 100 MHz
  bool ok = true;
  while (__hip_atomic_load(&mail[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > timeout_ticks) { ok = false; break; }  // (100 MHz ticks; RMI_HIP_PEER_TIMEOUT_S, default 60 s)
  }
  if (!ok) { atomicOr(&st->err_flags, rmi::EF_PEER_TIMEOUT); return; }
  for (int q = 0; q < RMI_STATS_WORDS; q++) stats_all[r * RMI_STATS_WORDS + q] = __hip_atomic_load(&mail[64 + r * RMI_STATS_WORDS + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static rmi_hip_multi* multi_of(rmi_hip_ctx* c);    // (accessor defined in rmi_hip.hip)

extern "C" {

int rmi_hip_generated_key(int generator, int dtype, uint64_t n_global, uint64_t seed, uint64_t index, uint64_t* key_bits) {
  if (!key_bits || n_global == 0 || index >= n_global || generator < 0 || generator > 1 || (dtype != RMI_KEY_U64 && dtype != RMI_KEY_U32))
    return RMI_ERR_BAD_ARG;
  const uint64_t span = dtype == RMI_KEY_U64 ? 0xFFFFFFFFFFFFFFFEull : 0xFFFFFFFDull;
  const uint64_t stride = span / n_global;
  if (stride == 0) return RMI_ERR_BAD_ARG;
  const uint64_t base_seed = seed ? seed : (dtype == RMI_KEY_U64 ? 42ull : 46ull);
  const uint64_t dup_seed = dtype == RMI_KEY_U64 ? 45ull : 47ull;
  uint64_t i = index;
  if (generator == 1) {
    const uint64_t cc = rmi_multi::splitmix64_h((i >> 3) + dup_seed) % 5ull;
    const uint64_t rr = cc < 3 ? 1ull : (cc == 3 ? 2ull : 8ull);
    i = i - (i % rr);
  }
  *key_bits = 1ull + i * stride + (rmi_multi::splitmix64_h(i + base_seed) % stride);
  return RMI_OK;
}

int rmi_hip_plan_shards(const rmi_hip_ctx* c, const rmi_hip_model_params* root, int dtype, uint64_t n, uint64_t L, int world,
                        rmi_hip_key_at_fn key_at, void* user, rmi_hip_shard* out) {
  if (!root || !key_at || !out || n == 0 || L == 0 || world < 1 || dtype < 0 || dtype > 2) return RMI_ERR_BAD_ARG;
  if (L % (uint64_t)world != 0) return RMI_ERR_BAD_ARG;                 // equal leaf counts: one all-gather of equal pieces
  const int table_bits = rmi_host::radix_table_bits(root->kind);
  if (table_bits > 0 && (!c || c->h_table.size() != (1ull << table_bits))) return RMI_ERR_BAD_ARG;
  if (!root_on_device_path(root->kind)) return RMI_ERR_UNSUPPORTED_MODEL;
  auto target = [&](uint64_t i) -> uint64_t {
    const uint64_t kb = key_at(user, i);
    if (table_bits > 0) {                                                 // radix.rs:124-134, clamp two_layer.rs:49
      uint64_t v = kb;
      if (dtype == RMI_KEY_F64) { double d; std::memcpy(&d, &kb, 8); v = rmi_host::sat_u64(d); }
      const uint64_t prefix = root->ip[0], bits = root->ip[1];
      const uint64_t nb = prefix + bits > 64 ? 0 : 64 - (prefix + bits);
      const uint64_t slot = ((v << (prefix & 63)) >> (prefix & 63)) >> (nb & 63);
      const uint64_t t = c->h_table[slot];
      return t < L - 1 ? t : L - 1;
    }
    uint64_t t = 0;
    (void)rmi_hip_root_target(root, dtype, kb, L, &t);
    return t;
  };
  auto first_ge = [&](uint64_t leaf) -> uint64_t {                       // lower_bound over the monotone targets (two_layer.rs:132-136)
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (target(mid) < leaf) lo = mid + 1; else hi = mid; }
    return lo;
  };
  auto key_less = [&](uint64_t a, uint64_t b) -> bool {                  // order of the key TYPE, on its bits
    if (dtype == RMI_KEY_F64) { double x, y; std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8); return x < y; }
    return a < b;
  };
  auto first_occurrence = [&](uint64_t i) -> uint64_t {
    const uint64_t v = key_at(user, i);
    if (i == 0 || key_at(user, i - 1) != v) return i;
    uint64_t lo = 0, hi = i - 1;
    while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (key_less(key_at(user, mid), v)) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const uint64_t per = L / (uint64_t)world;
  std::vector<uint64_t> cuts((size_t)world + 1);
  cuts[0] = 0; cuts[(size_t)world] = n;
  for (int r = 1; r < world; r++) cuts[(size_t)r] = first_ge((uint64_t)r * per);
  const uint64_t split = first_ge(L / 2);
  const uint64_t split_idx = split >= n ? ~0ull : split;
  const uint64_t split_target = split >= n ? 0 : target(split);
  for (int r = 0; r < world; r++) {
    rmi_hip_shard& s = out[r];
    s.n_global = n;
    s.key_lo = cuts[(size_t)r]; s.key_hi = cuts[(size_t)r + 1];
    s.leaf_lo = (uint64_t)r * per; s.leaf_hi = (uint64_t)(r + 1) * per;
    // left halo: the whole duplicate run of key[key_lo-1] plus one more key (prev-last point with its first-occurrence
    // offset; "was the previous key the split key" needs key_lo-2); right halo: key[key_hi] plus one spare
    if (s.key_lo == 0) s.read_lo = 0;
    else {
      const uint64_t fo = first_occurrence(s.key_lo - 1);
      const uint64_t a = fo < s.key_lo - 1 ? fo : s.key_lo - 1;
      s.read_lo = a > 0 ? a - 1 : 0;
    }
    s.read_hi = s.key_hi + 2 < n ? s.key_hi + 2 : n;
    s.split_idx = split_idx; s.split_target = split_target;
  }
  return RMI_OK;
}

// `radix` and `linear_spline` roots need a handful of keys (first, last, start of the last run): fitted through the
// key source, for key sets no single rank holds (same code as rmi_hip_fit_root uses on HBM-resident keys).
int rmi_hip_fit_root_from_source(int root_kind, int dtype, uint64_t n, uint64_t num_leaves, rmi_hip_key_at_fn key_at, void* user,
                                 rmi_hip_model_params* out) {
  if (!key_at || !out || n == 0 || num_leaves == 0) return RMI_ERR_BAD_ARG;
  switch (dtype) {
    case RMI_KEY_U64: return rmi_host::fit_root_sparse<uint64_t>(root_kind, [&](uint64_t i) { return (uint64_t)key_at(user, i); }, n, num_leaves, out);
    case RMI_KEY_U32: return rmi_host::fit_root_sparse<uint32_t>(root_kind, [&](uint64_t i) { return (uint32_t)key_at(user, i); }, n, num_leaves, out);
    case RMI_KEY_F64: return rmi_host::fit_root_sparse<double>(root_kind, [&](uint64_t i) { const uint64_t b = key_at(user, i); double d; std::memcpy(&d, &b, 8); return d; }, n, num_leaves, out);
  }
  return RMI_ERR_BAD_ARG;
}

int rmi_hip_comm_unique_id(void* id_out) {
  if (!id_out) return RMI_ERR_BAD_ARG;
  rmi_multi::Api& a = rmi_multi::api();
  if (!a.ok()) return RMI_ERR_NO_RCCL;
  rmi_multi::NcclId id;
  if (a.GetUniqueId(&id) != 0) return RMI_ERR_RCCL;
  std::memcpy(id_out, &id, sizeof id);
  return RMI_OK;
}

int rmi_hip_comm_init(rmi_hip_ctx* c, int rank, int world, const void* id_bytes) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !id_bytes)) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  HIPCHK(c, hipSetDevice(c->device));
  if (m->comm) { (void)rmi_multi::api().CommDestroy(m->comm); m->comm = nullptr; }
  m->rank = rank; m->world = world;
  if (world == 1 && !id_bytes) return RMI_OK;                            // a communicator of one: nothing to exchange
  rmi_multi::Api& a = rmi_multi::api();
  if (!a.ok()) return RMI_ERR_NO_RCCL;
  rmi_multi::NcclId id;
  std::memcpy(&id, id_bytes, sizeof id);
  const int rc = a.CommInitRank(&m->comm, world, id, rank);
  if (rc != 0) { set_err(c, "ncclCommInitRank: %s", a.GetErrorString ? a.GetErrorString(rc) : "error"); m->comm = nullptr; return RMI_ERR_RCCL; }
  return RMI_OK;
}

// what the communicator itself says (ncclCommCount / ncclCommUserRank): the proof that a collective spans `world` ranks
int rmi_hip_comm_info(rmi_hip_ctx* c, int* world, int* rank) {
  if (!c || !world || !rank) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  *world = m->world; *rank = m->rank;
  if (!m->comm) return RMI_OK;
  rmi_multi::Api& a = rmi_multi::api();
  if (!a.CommCount || !a.CommUserRank) return RMI_ERR_NO_RCCL;
  if (a.CommCount(m->comm, world) != 0 || a.CommUserRank(m->comm, rank) != 0) return RMI_ERR_RCCL;
  return RMI_OK;
}

int rmi_hip_comm_destroy(rmi_hip_ctx* c) {
  if (!c) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  if (m->comm) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); (void)rmi_multi::api().CommDestroy(m->comm); m->comm = nullptr; }
  m->world = 1; m->rank = 0;
  return RMI_OK;
}

// the aggregates of the whole model from the ranks' records (two_layer.rs:267-287 over all shards: lexicographic maximum, the
// last maximum wins; exact integer sum; f64 sums); true if some rank still has listed leaves to finish
static bool sharded_totals(rmi_hip_ctx* c, rmi_hip_multi* m, rmi_hip_result* out) {
  unsigned long long mx = 0, mi = 0, sn = 0; double l2 = 0.0, lg = 0.0;
  bool listed = false;
  for (int r = 0; r < m->world; r++) {
    unsigned long long v[RMI_STATS_WORDS]; double d[2];
    std::memcpy(v, m->h_stats_all + RMI_STATS_BYTES * r, sizeof v); std::memcpy(d, m->h_stats_all + RMI_STATS_BYTES * r + 24, 16);
    if (v[0] > mx || (v[0] == mx && v[1] >= mi)) { mx = v[0]; mi = v[1]; }
    sn += v[2]; l2 += d[0]; lg += d[1];
    listed = listed || v[RMI_STATS_WORDS - 1] != 0;
  }
  if (out) {
    const double ng = (double)c->shard.n;
    out->model_max_error = mx; out->model_max_error_idx = mi;
    out->model_avg_error = (double)sn / ng; out->model_avg_l2_error = l2; out->model_avg_log2_error = lg / ng;
    out->model_max_log2_error = std::log2((double)mx);
  }
  return listed;
}

// The exchange as peer stores (include/rmi_hip.h).  Epoch e uses half e & 1 of every rank's table: a rank that runs ahead
// stores into the half its peers are not reading.
// how long k_peer_wait waits for a peer's flag: a peer may still be walking a listed leaf of 10^8 points on one wave (28 ns a point)
static unsigned long long peer_timeout_ticks() {
  static const unsigned long long t = [] {
    const char* e = std::getenv("RMI_HIP_PEER_TIMEOUT_S");
    const double s = (e && *e) ? std::atof(e) : 60.0;
    return (unsigned long long)((s > 0.0 ? s : 60.0) * 1e8);              // wall_clock64: 100 MHz
  }();
  return t;
}
static int direct_exchange(rmi_hip_ctx* c, rmi_hip_multi* m, unsigned long long epoch, uint64_t off, uint64_t bytes, bool rows_there) {
  unsigned char* table = m->d_rows2 + (epoch & 1ull) * m->rows2_slot;
  // (peer pointers of this epoch's half: the tables hold the bases, the half is part of the byte offset)
  const uint64_t n16 = bytes / 16;
  const uint64_t byte_off = (epoch & 1ull) * m->rows2_slot + off;
  // (rows_there: k_leaf_lanes has stored its rows into the peers' tables itself)
  if (!rows_there) hipLaunchKernelGGL(k_peer_push, dim3(512), dim3(256), 0, c->stream, (const uint4*)(table + off), n16, (unsigned char* const*)m->d_peer_rows, byte_off, m->rank, m->world);
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, c->stream, (unsigned char* const*)m->d_peer_mail, (const unsigned long long*)&c->d_state->max_err, m->rank, m->world, epoch);
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, c->stream, (unsigned long long*)m->d_mail, m->world, epoch, (unsigned long long*)m->d_stats_all, c->d_state,
                     peer_timeout_ticks());
  HIPCHK(c, hipMemcpyAsync(m->h_stats_all, m->d_stats_all, RMI_STATS_BYTES * (size_t)m->world, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&c->h_state->err_flags, &c->d_state->err_flags, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));   // (k_peer_wait may have raised the timeout)
  if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[9], c->stream));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->h_state->err_flags & EF_PEER_TIMEOUT) { set_err(c, "direct exchange: a peer's flag of epoch %llu did not arrive within %.0f s (RMI_HIP_PEER_TIMEOUT_S); the peers' tables of this epoch are undefined", epoch, (double)peer_timeout_ticks() / 1e8); return RMI_ERR_HIP; }
  return RMI_OK;
}

static int train_sharded_direct(rmi_hip_ctx* c, rmi_hip_multi* m, const rmi_hip_model_params* root, int leaf_kind, uint64_t num_leaves,
                                uint64_t rowb, uint64_t L_own, rmi_hip_result* out) {
  unsigned long long epoch = ++m->epoch;
  unsigned char* table = m->d_rows2 + (epoch & 1ull) * m->rows2_slot;
  const uint64_t off = (uint64_t)m->rank * L_own * rowb;
  m->rows_last_bytes = num_leaves * rowb;
  void* const saved_ext = c->d_rows_ext;
  c->d_rows_ext = table + off;
  c->defer_sync = true;
  // (<= 8 ranks: the peers' tables of this epoch go to k_leaf_lanes, which stores every row it finishes there as well)
  c->peer_fuse_n = 0;
  if (m->world <= 8 && m->fuse_peer_stores) {
    for (int r = 0; r < m->world; r++)
      if (r != m->rank) c->peer_fuse_tab[c->peer_fuse_n++] = m->peer_rows[r] + (epoch & 1ull) * m->rows2_slot;
  }
  c->rows_pushed = false;                                              // (set by the launch only if its kernels really store to the peers)
  int rc = rmi_hip_train_two_layer(c, root, leaf_kind, num_leaves, out);
  c->defer_sync = false;
  c->peer_fuse_n = 0;
  c->d_rows_ext = saved_ext;
  if (rc != RMI_OK) return rc;
  if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
  rc = direct_exchange(c, m, epoch, off, L_own * rowb, c->rows_pushed);
  if (rc != RMI_OK) return rc;
  {
    // the ranks' results may have been published without their list kernels (launch_pipeline): if SOME rank handed leaves
    // over, every rank learns it from the records, the ranks concerned finish theirs, and all exchange once more -- in the
    // next epoch, i.e. the other half of the tables (this rank's rows move there first)
    const bool listed = sharded_totals(c, m, nullptr);
    if (listed) {
      if (c->tail_armed && c->h_state->pending > 0) { rc = c->tail_fn(); if (rc != RMI_OK) { c->tail_armed = false; c->tail_fn = nullptr; return rc; } }
      epoch = ++m->epoch;
      unsigned char* table2 = m->d_rows2 + (epoch & 1ull) * m->rows2_slot;
      HIPCHK(c, hipMemcpyAsync(table2 + off, table + off, L_own * rowb, hipMemcpyDeviceToDevice, c->stream));
      rc = direct_exchange(c, m, epoch, off, L_own * rowb, false);
    }
    c->tail_armed = false; c->tail_fn = nullptr;
    if (rc != RMI_OK) return rc;
  }
  rc = finish_train(c, leaf_kind, num_leaves, out);
  if (rc != RMI_OK) return rc;
  { float xms = 0.f; if (c->profile_level >= 0 && hipEventElapsedTime(&xms, c->ev[7], c->ev[9]) == hipSuccess) out->kernel_ns[7] = (uint64_t)((double)xms * 1e6); }
  (void)sharded_totals(c, m, out);
  return RMI_OK;
}

// One training of this rank's shard (rmi_hip_set_shard) + the exchange: when the call returns, the full row
// buffer of EVERY rank holds the L rows, and `out` carries the aggregates of the whole model.
int rmi_hip_train_sharded(rmi_hip_ctx* c, const rmi_hip_model_params* root, int leaf_kind, uint64_t num_leaves, rmi_hip_result* out) {
  if (!c || !root || !out) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  if (!c->have_shard) return RMI_ERR_BAD_ARG;
  const int ppl = leaf_kind == RMI_MODEL_CUBIC ? 4 : 2;
  const uint64_t rowb = (uint64_t)ppl * 8 + 8;
  const uint64_t L_own = c->shard.leaf_hi - c->shard.leaf_lo;
  if (L_own * (uint64_t)m->world != num_leaves || c->shard.leaf_lo != (uint64_t)m->rank * L_own) return RMI_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const bool direct = m->exchange == RMI_EXCHANGE_DIRECT && m->world > 1 && m->peers == m->world && m->rows2_slot >= num_leaves * rowb &&
                      (L_own * rowb) % 16 == 0;
  if (m->exchange == RMI_EXCHANGE_DIRECT && m->world > 1 && !direct) { set_err(c, "direct exchange: peers not imported (or another table size than exported)"); return RMI_ERR_BAD_ARG; }
  if (direct) return train_sharded_direct(c, m, root, leaf_kind, num_leaves, rowb, L_own, out);
  if (m->rows_full_bytes < num_leaves * rowb) {
    if (m->d_rows_full) (void)hipFree(m->d_rows_full);
    m->d_rows_full = nullptr; m->rows_full_bytes = 0;
    HIPCHK(c, hipMalloc(&m->d_rows_full, num_leaves * rowb));
    m->rows_full_bytes = num_leaves * rowb;
  }
  if (!m->d_stats_all) {
    HIPCHK(c, hipMalloc(&m->d_stats_all, RMI_STATS_BYTES * 64));
    HIPCHK(c, hipHostMalloc((void**)&m->h_stats_all, RMI_STATS_BYTES * 64, hipHostMallocDefault));
  }
  if (m->world > 64) return RMI_ERR_BAD_ARG;
  m->rows_last_bytes = num_leaves * rowb;
  void* const saved_ext = c->d_rows_ext;
  c->d_rows_ext = m->d_rows_full + (uint64_t)m->rank * L_own * rowb;    // the kernels write this rank's rows into its slot
  c->defer_sync = m->comm != nullptr;                                  // the exchange is queued behind the kernels, one sync for both
  int rc = rmi_hip_train_two_layer(c, root, leaf_kind, num_leaves, out);
  c->defer_sync = false;
  if (rc != RMI_OK) { c->d_rows_ext = saved_ext; return rc; }
  if (m->world > 1 && !m->comm) { c->d_rows_ext = saved_ext; return RMI_ERR_BAD_ARG; }
  if (m->comm) {
    rmi_multi::Api& a = rmi_multi::api();
    if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[7], c->stream));   // (kernel_ns[7] of the result: the exchange alone)
    auto exchange = [&]() -> int {
      // rows: in place (this rank's piece already sits in its slot); aggregates: the record of RMI_STATS_WORDS words.
      // One group: RCCL launches the two gathers as one kernel (one launch latency instead of two on a ~0.1 ms step)
      const bool grouped = a.GroupStart && a.GroupEnd;
      int n0 = grouped ? a.GroupStart() : 0;
      int n1 = a.AllGather(c->d_rows_ext, m->d_rows_full, (size_t)(L_own * rowb), 1 /* ncclUint8 */, m->comm, c->stream);
      int n2 = a.AllGather(&c->d_state->max_err, m->d_stats_all, RMI_STATS_BYTES, 1, m->comm, c->stream);
      int n3 = grouped ? a.GroupEnd() : 0;
      if (n1 == 0 && n2 == 0) n1 = n0 ? n0 : n3;
      if (n1 != 0 || n2 != 0) { set_err(c, "ncclAllGather: %s", a.GetErrorString ? a.GetErrorString(n1 ? n1 : n2) : "error"); return RMI_ERR_RCCL; }
      HIPCHK(c, hipMemcpyAsync(m->h_stats_all, m->d_stats_all, RMI_STATS_BYTES * (size_t)m->world, hipMemcpyDeviceToHost, c->stream));
      if (c->profile_level >= 0) HIPCHK(c, hipEventRecord(c->ev[9], c->stream));   // device time of the call now includes the exchange
      HIPCHK(c, hipStreamSynchronize(c->stream));
      return RMI_OK;
    };
    rc = exchange();
    if (rc == RMI_OK && sharded_totals(c, m, nullptr)) {
      // the ranks' results may have been published without their list kernels (launch_pipeline): if SOME rank handed leaves
      // over, every rank learns it from the records, the ranks concerned finish theirs, and all gather once more
      if (c->tail_armed && c->h_state->pending > 0) rc = c->tail_fn();
      if (rc == RMI_OK) rc = exchange();
    }
    c->tail_armed = false; c->tail_fn = nullptr;
    if (rc != RMI_OK) { c->d_rows_ext = saved_ext; return rc; }
    rc = finish_train(c, leaf_kind, num_leaves, out);                     // error flags, per-shard result, timings
    if (rc != RMI_OK) { c->d_rows_ext = saved_ext; return rc; }
    { float xms = 0.f; if (c->profile_level >= 0 && hipEventElapsedTime(&xms, c->ev[7], c->ev[9]) == hipSuccess) out->kernel_ns[7] = (uint64_t)((double)xms * 1e6); }
    (void)sharded_totals(c, m, out);
  }
  c->d_rows_ext = saved_ext;
  return RMI_OK;
}

// ---------------------------------------------------------------------------------------------
// Upload + train, overlapped (include/rmi_hip.h): pinned double-buffered staging, shards trained behind the upload.
// ---------------------------------------------------------------------------------------------
namespace rmi_stream_up {
struct HostKeys { const unsigned char* p; int dtype; };
static uint64_t key_at(void* user, uint64_t i) {
  const HostKeys* h = (const HostKeys*)user;
  if (h->dtype == RMI_KEY_U32) { uint32_t v; std::memcpy(&v, h->p + i * 4, 4); return v; }
  uint64_t v; std::memcpy(&v, h->p + i * 8, 8); return v;
}
// host copy with a few threads: one thread moves ~10 GB/s, the link wants 55
static void par_memcpy(void* dst, const void* src, size_t bytes, int threads) {
  if (bytes < (8u << 20) || threads <= 1) { std::memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t per = ((bytes + threads - 1) / threads + 4095) & ~(size_t)4095;
  for (int t = 1; t < threads; t++) {
    const size_t off = per * t;
    if (off >= bytes) break;
    const size_t len = off + per <= bytes ? per : bytes - off;
    th.emplace_back([=]() { std::memcpy((char*)dst + off, (const char*)src + off, len); });
  }
  std::memcpy(dst, src, per < bytes ? per : bytes);
  for (auto& t : th) t.join();
}
}  // namespace rmi_stream_up

int rmi_hip_train_streamed(rmi_hip_ctx* c, const void* host_keys, uint64_t n, int dtype, const rmi_hip_model_params* root,
                           int leaf_kind, uint64_t num_leaves, int chunks, rmi_hip_result* out) {
  if (!c || !host_keys || !root || !out || n == 0 || dtype < 0 || dtype > 2 || num_leaves == 0 || chunks < 1 || chunks > RMI_STREAM_MAX_CHUNKS)
    return RMI_ERR_BAD_ARG;
  if (num_leaves % (uint64_t)chunks != 0) return RMI_ERR_BAD_ARG;
  if (c->upload_thread.joinable()) c->upload_thread.join();
  HIPCHK(c, hipSetDevice(c->device));
  constexpr size_t STG = 64u << 20;                                     // bytes per staging buffer
  const size_t ks = key_size(dtype);
  // ---- plan on the host keys
  rmi_stream_up::HostKeys hk{(const unsigned char*)host_keys, dtype};
  std::vector<rmi_hip_shard> sh((size_t)chunks);
  int rc = rmi_hip_plan_shards(c, root, dtype, n, num_leaves, chunks, rmi_stream_up::key_at, &hk, sh.data());
  if (rc != RMI_OK) return rc;
  // ---- buffers
  for (int b = 0; b < 2; b++) {
    if (!c->h_stage[b]) HIPCHK(c, hipHostMalloc(&c->h_stage[b], STG, hipHostMallocDefault));
    if (!c->ev_stage[b]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage[b], hipEventDisableTiming));
  }
  if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  HIPCHK(c, hipStreamSynchronize(c->stream));                           // (a train call may still read the old keys)
  c->d_keys = nullptr; c->n = 0;
  if (c->d_keys_owned) { HIPCHK(c, hipFree(c->d_keys_owned)); c->d_keys_owned = nullptr; }
  HIPCHK(c, hipMalloc(&c->d_keys_owned, n * ks));
  c->d_keys = c->d_keys_owned; c->n = n; c->dtype = dtype; c->keys_epoch++;
  // ---- upload in chunks; a shard is launched behind the chunk that completes its key range
  const int threads = 8;
  const size_t total = n * ks;
  int next = 0;
  bool used[2] = {false, false};
  bool any_sigma = false;
  c->stream_mode = true;
  auto fail = [&](int code) {                                           // (no half-uploaded key set stays behind)
    c->stream_mode = false; c->defer_sync = false;
    (void)rmi_hip_set_shard(c, nullptr);
    (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(c->stream);
    c->d_keys = nullptr; c->n = 0; c->last_L = 0;
    return code;
  };
  size_t off = 0;
  for (int q = 0; off < total; q++) {
    const int b = q & 1;
    const size_t len = total - off < STG ? total - off : STG;
    if (used[b] && hipEventSynchronize(c->ev_stage[b]) != hipSuccess) return fail(RMI_ERR_HIP);
    rmi_stream_up::par_memcpy(c->h_stage[b], (const char*)host_keys + off, len, threads);
    if (hipMemcpyAsync((char*)c->d_keys_owned + off, c->h_stage[b], len, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
        hipEventRecord(c->ev_stage[b], c->copy_stream) != hipSuccess) return fail(RMI_ERR_HIP);
    used[b] = true;
    off += len;
    const uint64_t have = off / ks;                                       // keys on their way
    while (next < chunks && sh[(size_t)next].read_hi <= have) {
      if (hipStreamWaitEvent(c->stream, c->ev_stage[b], 0) != hipSuccess) return fail(RMI_ERR_HIP);
      rc = rmi_hip_set_shard(c, &sh[(size_t)next]);
      if (rc != RMI_OK) return fail(rc);
      c->stream_slot = next;
      c->defer_sync = true;
      rc = rmi_hip_train_two_layer(c, root, leaf_kind, num_leaves, out);
      c->defer_sync = false;
      if (rc != RMI_OK) return fail(rc);
      any_sigma = any_sigma || c->last_sigma;                             // (a shard with few keys per leaf takes the exact passes)
      next++;
    }
  }
  if (next != chunks) return fail(RMI_ERR_BAD_ARG);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(RMI_ERR_HIP);
  c->stream_mode = false;
  (void)rmi_hip_set_shard(c, nullptr);
  // ---- the aggregates of the whole model from the shards' (two_layer.rs:267-287: lexicographic maximum, last one wins;
  //      exact integer sum; f64 sums), the other counters summed; finish_train turns slot 0 into the result
  DevState tot = c->h_state[0];
  for (int r = 1; r < chunks; r++) {
    const DevState& d = c->h_state[r];
    tot.err_flags |= d.err_flags;
    if (d.max_err > tot.max_err || (d.max_err == tot.max_err && d.max_err_idx >= tot.max_err_idx)) { tot.max_err = d.max_err; tot.max_err_idx = d.max_err_idx; }
    tot.sum_n_err += d.sum_n_err; tot.sum_l2 += d.sum_l2; tot.sum_log2 += d.sum_log2;
    tot.long_count += d.long_count; tot.flag_count += d.flag_count; tot.guard_count += d.guard_count; tot.merged_count += d.merged_count;
  }
  c->h_state[0] = tot;
  c->last_sigma = any_sigma;
  return finish_train(c, leaf_kind, num_leaves, out);
}

void* rmi_hip_device_rows_full(rmi_hip_ctx* c) {
  if (!c) return nullptr;
  rmi_hip_multi* m = multi_of(c);
  if (m->exchange == RMI_EXCHANGE_DIRECT && m->d_rows2 && m->epoch) return m->d_rows2 + (m->epoch & 1ull) * m->rows2_slot;
  return m->d_rows_full;
}

int rmi_hip_peer_export(rmi_hip_ctx* c, int rank, int world, int leaf_kind, uint64_t num_leaves, void* handle_out) {
  if (!c || !handle_out || world < 1 || world > 64 || rank < 0 || rank >= world || num_leaves == 0) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t rowb = (leaf_kind == RMI_MODEL_CUBIC ? 4 : 2) * 8 + 8;
  const uint64_t slot = ((num_leaves * rowb + 255) / 256) * 256;
  if (m->d_rows2 && m->rows2_slot != slot) return RMI_ERR_BAD_ARG;     // (one table size per context: the peers hold mappings of it)
  if (!m->d_rows2) {
    HIPCHK(c, hipMalloc(&m->d_rows2, 2 * slot));
    m->rows2_slot = slot;
    HIPCHK(c, hipExtMallocWithFlags((void**)&m->d_mail, RMI_MAIL_BYTES, hipDeviceMallocFinegrained));
    HIPCHK(c, hipMemset(m->d_mail, 0, RMI_MAIL_BYTES));
    HIPCHK(c, hipMalloc(&m->d_peer_rows, 64 * sizeof(void*)));
    HIPCHK(c, hipMalloc(&m->d_peer_mail, 64 * sizeof(void*)));
    if (!m->d_stats_all) {
      HIPCHK(c, hipMalloc(&m->d_stats_all, RMI_STATS_BYTES * 64));
      HIPCHK(c, hipHostMalloc((void**)&m->h_stats_all, RMI_STATS_BYTES * 64, hipHostMallocDefault));
    }
  }
  m->rank = rank; m->world = world;
  m->peer_rows[rank] = m->d_rows2; m->peer_mail[rank] = m->d_mail;
  if (!m->peer_open[rank]) { m->peer_open[rank] = true; m->peers++; }
  rmi_peer_handle h; std::memset(&h, 0, sizeof h);
  HIPCHK(c, hipIpcGetMemHandle(&h.rows, m->d_rows2));
  HIPCHK(c, hipIpcGetMemHandle(&h.mail, m->d_mail));
  h.slot_bytes = slot; h.rank = rank; h.world = world;
  std::memset(handle_out, 0, RMI_HIP_PEER_HANDLE_BYTES);
  std::memcpy(handle_out, &h, sizeof h);
  return RMI_OK;
}

int rmi_hip_peer_import(rmi_hip_ctx* c, int peer_rank, const void* handle) {
  if (!c || !handle) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  rmi_peer_handle h; std::memcpy(&h, handle, sizeof h);
  if (!m->d_rows2 || peer_rank < 0 || peer_rank >= m->world || h.rank != peer_rank || h.world != m->world || h.slot_bytes != m->rows2_slot) return RMI_ERR_BAD_ARG;
  if (peer_rank == m->rank || m->peer_open[peer_rank]) return RMI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  void* pr = nullptr; void* pm = nullptr;
  HIPCHK(c, hipIpcOpenMemHandle(&pr, h.rows, hipIpcMemLazyEnablePeerAccess));
  HIPCHK(c, hipIpcOpenMemHandle(&pm, h.mail, hipIpcMemLazyEnablePeerAccess));
  m->peer_rows[peer_rank] = (unsigned char*)pr; m->peer_mail[peer_rank] = (unsigned char*)pm;
  m->peer_open[peer_rank] = true; m->peers++;
  if (m->peers == m->world) {
    HIPCHK(c, hipMemcpy(m->d_peer_rows, m->peer_rows, 64 * sizeof(void*), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(m->d_peer_mail, m->peer_mail, 64 * sizeof(void*), hipMemcpyHostToDevice));
  }
  return RMI_OK;
}

int rmi_hip_set_exchange(rmi_hip_ctx* c, int mode) {
  if (!c || (mode != RMI_EXCHANGE_RCCL && mode != RMI_EXCHANGE_DIRECT)) return RMI_ERR_BAD_ARG;
  multi_of(c)->exchange = mode;
  return RMI_OK;
}

int rmi_hip_download_rows_full(rmi_hip_ctx* c, void* host_out, uint64_t capacity_bytes) {
  if (!c || !host_out) return RMI_ERR_BAD_ARG;
  rmi_hip_multi* m = multi_of(c);
  const unsigned char* table = (const unsigned char*)rmi_hip_device_rows_full(c);
  if (!table || !m->rows_last_bytes || capacity_bytes < m->rows_last_bytes) return RMI_ERR_BAD_ARG;   // (the table of the LAST training, not the high-water allocation)
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(host_out, table, m->rows_last_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RMI_OK;
}

}  // extern "C"
