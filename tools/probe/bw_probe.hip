// Development aid (GPU box): what a read-only kernel gets out of HBM on this machine, by access pattern.
// build: hipcc --offload-arch=gfx950 -O3 -o bw_probe tools/probe/bw_probe.hip ; run: ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// A: grid-stride, U independent 16-byte loads per lane per trip, far apart (the library's k_read_bw is U = 4)
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stride(const u4* __restrict__ src, uint64_t n16, unsigned int* sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned int acc = 0;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// B: every wave owns contiguous chunks of U KB (U loads of 1 KB per wave back to back), chunks dealt round-robin to waves
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_chunk(const u4* __restrict__ src, uint64_t n16, unsigned int* sink) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x % 64;
  const uint64_t nwaves = (uint64_t)gridDim.x * blockDim.x / 64;
  unsigned int acc = 0;
  for (uint64_t c = wave; (c + 1) * U * 64 <= n16; c += nwaves) {
    const u4* p = src + c * U * 64 + lane;
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(p + u * 64) : p[u * 64];
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// C: 64 row streams per wave like k_leaf_lanes: lane group of 8 lanes reads one 128-byte line of its row per load, rows are
// contiguous pieces of `rowlen` lines, 8 loads (64 rows) per step, U steps in flight
// D dependent scattered loads in front of the streaming (what a wave of k_leaf_lanes does before its first panel: the
// entries of leaf_start, then single keys, then the panels): each is a round trip with nothing else in flight
template <int U, int D>
__global__ void __launch_bounds__(64) k_rows_dep(const u4* __restrict__ src, uint64_t n16, unsigned int rowlines, unsigned int* sink) {
  extern __shared__ unsigned int lds_pad[];
  if (rowlines == 0xFFFFFFFFu) lds_pad[threadIdx.x] = 1;
  const uint64_t wave = blockIdx.x;
  const int lane = threadIdx.x;
  uint64_t base = wave * 64ull * rowlines * 8ull;
  if (base + 64ull * rowlines * 8ull > n16) return;
  unsigned int acc = 0;
  uint64_t chase = (wave * 2654435761ull + lane * 40503ull) % (n16 - 1);
#pragma unroll
  for (int d = 0; d < D; d++) { const u4 v = src[chase]; acc ^= v.x; chase = (chase * 31ull + (v.y & 1u) + 977ull * (d + 1)) % (n16 - 1); }
  base += (acc == 0x7fffffffu) ? 1 : 0;                            // (the streaming waits for the chain)
  for (unsigned int l = 0; l + U <= rowlines; l += U) {
    u4 v[U][8];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) v[u][i] = src[base + (uint64_t)(i * 8 + lane / 8) * rowlines * 8ull + (uint64_t)(l + u) * 8ull + (lane % 8)];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc ^= v[u][i].x ^ v[u][i].y ^ v[u][i].z ^ v[u][i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int U, bool NT = false>
__global__ void __launch_bounds__(64) k_rows(const u4* __restrict__ src, uint64_t n16, unsigned int rowlines, unsigned int* sink) {
  extern __shared__ unsigned int lds_pad[];                      // (dynamic LDS only to set the occupancy like k_leaf_lanes' ring does)
  if (rowlines == 0xFFFFFFFFu) lds_pad[threadIdx.x] = 1;
  const uint64_t wave = blockIdx.x;
  const int lane = threadIdx.x;
  const uint64_t base = wave * 64ull * rowlines * 8ull;          // 16-byte units: a line = 8 of them
  if (base + 64ull * rowlines * 8ull > n16) return;
  unsigned int acc = 0;
  for (unsigned int l = 0; l + U <= rowlines; l += U) {
    u4 v[U][8];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) { const u4* q = src + base + (uint64_t)(i * 8 + lane / 8) * rowlines * 8ull + (uint64_t)(l + u) * 8ull + (lane % 8); v[u][i] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc ^= v[u][i].x ^ v[u][i].y ^ v[u][i].z ^ v[u][i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const uint64_t bytes = 1600000000ull, n16 = bytes / 16;
  u4* d; unsigned int* sink;
  CHK(hipMalloc(&d, bytes)); CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(d, 1, bytes));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& launch) {
    for (int w = 0; w < 3; w++) launch();
    hipEventRecord(e0);
    const int it = 10;
    for (int k = 0; k < it; k++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
  };
  for (int g : {8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "stride U=4 grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_stride<4, false>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
    snprintf(nm, sizeof nm, "stride U=8 grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_stride<8, false>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
    snprintf(nm, sizeof nm, "stride U=8 nt grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_stride<8, true>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
    snprintf(nm, sizeof nm, "chunk U=4 grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_chunk<4, false>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
    snprintf(nm, sizeof nm, "chunk U=8 grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_chunk<8, false>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
    snprintf(nm, sizeof nm, "chunk U=8 nt grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_chunk<8, true>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
    snprintf(nm, sizeof nm, "chunk U=16 grid=%d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_chunk<16, false>), dim3(g), dim3(256), 0, 0, d, n16, sink); });
  }
  {
    const unsigned int rl = 12u, lds = 20000u;
    const unsigned int waves = (unsigned int)(n16 / (64ull * rl * 8ull));
    timeit("64 rows x 12 lines, 8 waves/CU, 0 dependent loads first", [&]() { hipLaunchKernelGGL((k_rows<2>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    timeit("64 rows x 12 lines, 8 waves/CU, 1 dependent load first", [&]() { hipLaunchKernelGGL((k_rows_dep<2, 1>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    timeit("64 rows x 12 lines, 8 waves/CU, 2 dependent loads first", [&]() { hipLaunchKernelGGL((k_rows_dep<2, 2>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    timeit("64 rows x 12 lines, 8 waves/CU, 3 dependent loads first", [&]() { hipLaunchKernelGGL((k_rows_dep<2, 3>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
  }
  // row streams: rows of 12 lines (190 keys of 8 bytes); LDS per wave sets the waves per CU (20 000 B: 8 like k_leaf_lanes)
  for (unsigned int lds : {0u, 20000u, 13000u}) {
    const unsigned int rl = 12u;
    const unsigned int waves = (unsigned int)(n16 / (64ull * rl * 8ull));
    char nm[96];
    snprintf(nm, sizeof nm, "64 rows x 12 lines, lds %u, 2 in flight", lds); timeit(nm, [&]() { hipLaunchKernelGGL((k_rows<2>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    snprintf(nm, sizeof nm, "64 rows x 12 lines, lds %u, 3 in flight", lds); timeit(nm, [&]() { hipLaunchKernelGGL((k_rows<3>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    snprintf(nm, sizeof nm, "64 rows x 12 lines, lds %u, 4 in flight", lds); timeit(nm, [&]() { hipLaunchKernelGGL((k_rows<4>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    snprintf(nm, sizeof nm, "64 rows x 12 lines, lds %u, 6 in flight", lds); timeit(nm, [&]() { hipLaunchKernelGGL((k_rows<6>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    snprintf(nm, sizeof nm, "64 rows x 12 lines, lds %u, 2 in flight nt", lds); timeit(nm, [&]() { hipLaunchKernelGGL((k_rows<2, true>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
    snprintf(nm, sizeof nm, "64 rows x 12 lines, lds %u, 4 in flight nt", lds); timeit(nm, [&]() { hipLaunchKernelGGL((k_rows<4, true>), dim3(waves), dim3(64), lds, 0, d, n16, rl, sink); });
  }
  return 0;
}
