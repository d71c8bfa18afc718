// Development aid (GPU box): does global_load_lds_dwordx4 take a source that is only 8-byte aligned, and what does it cost?
// One wave copies 64 x 16 bytes from keys + shift (shift in 8-byte units) to LDS and checks them; then a timing loop of
// 8-instruction panels (64 rows x 128 B, rows 1.5 KB apart) with aligned and with shifted rows, 1 024 waves.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_align_probe tools/probe/dma_align_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void* base, unsigned lds, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o], %[kb]\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep) : [lds] "s"(lds), [kb] "s"(base), [o] "v"(voff) : "memory");
}
__global__ void __launch_bounds__(64) k_check(const unsigned long long* keys, unsigned shift, unsigned* bad) {
  __shared__ __attribute__((aligned(1024))) unsigned long long buf[128];
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)buf);
  dma16(keys, lds, threadIdx.x * 16u + shift * 8u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const unsigned long long a = buf[threadIdx.x * 2], b = buf[threadIdx.x * 2 + 1];
  if (a != keys[threadIdx.x * 2 + shift] || b != keys[threadIdx.x * 2 + 1 + shift]) atomicAdd(bad, 1u);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_time(const unsigned long long* keys, unsigned shift, unsigned trips, unsigned long long* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[32768];
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
  const unsigned lane = threadIdx.x;
  // rows: 64 per wave, 1536 B apart, wave regions 96 KB apart; instruction i serves rows 8 i + lane / 8, piece lane % 8
  const unsigned char* base = reinterpret_cast<const unsigned char*>(keys) + (size_t)blockIdx.x * 64u * 1536u * 4u;
  unsigned long long acc = 0;
  for (unsigned t = 0; t < trips; t++) {
#pragma unroll
    for (unsigned i = 0; i < 8; i++) {
      const unsigned row = i * 8u + (lane >> 3);
      dma16(base, lds + (t & 3u) * 8192u + i * 1024u, row * 1536u + t * 128u + (lane & 7u) * 16u + shift * 8u);
    }
    if (t >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    acc += *reinterpret_cast<unsigned long long*>(ring + ((t + 2u) & 3u) * 8192u + lane * 128u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x1234567ull) *sink = acc;
}
int main() {
  const size_t n = (size_t)1024 * 64 * 1536 * 4 / 8 + 4096;
  std::vector<unsigned long long> h(n);
  for (size_t i = 0; i < n; i++) h[i] = i * 0x9E3779B97F4A7C15ull + 1;
  unsigned long long* d; unsigned* bad; unsigned long long* sink;
  CHK(hipMalloc(&d, n * 8)); CHK(hipMalloc(&bad, 4)); CHK(hipMalloc(&sink, 8));
  CHK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
  for (unsigned shift = 0; shift < 4; shift++) {
    CHK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d, shift, bad);
    unsigned hb = 0; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("shift %u x 8 B: %u lanes wrong\n", shift, hb);
  }
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (unsigned shift = 0; shift < 2; shift++) for (int rep = 0; rep < 2; rep++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_time, dim3(1024), dim3(64), 0, 0, d, shift * 5u, 48u, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("panels of %s rows: %.1f us for %.2f GB requested (%.2f TB/s)\n", shift ? "shifted (40 B)" : "aligned", ms * 1e3, 1024.0 * 48 * 8192 / 1e9, 1024.0 * 48 * 8192 / ms / 1e9);
  }
  return 0;
}
