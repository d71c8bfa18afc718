// Probe: achievable HBM read bandwidth of the lane-per-chunk access pattern (64 streams per wave),
// as a function of bytes read contiguously per lane per visit and of the number of lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SEG16>   // 16-byte pieces per lane per visit (SEG16*16 bytes contiguous per lane)
__global__ void __launch_bounds__(64) k_probe(const uint4* __restrict__ src, uint64_t n16, uint64_t chunk16, unsigned int* sink) {
  const int lane = threadIdx.x;
  const uint64_t wave_base = (uint64_t)blockIdx.x * 64 * chunk16;
  // visit v: lane l reads pieces [l*chunk16 + v*SEG16, +SEG16); instruction q covers pieces q*(64/ (128/16))...
  // coalesced form: instruction k covers rows r = k*RPI .. with LPR lanes per row
  constexpr int LPR = SEG16;                 // lanes per row (each lane 16 B)
  constexpr int RPI = 64 / LPR;              // rows per instruction
  constexpr int NI = 64 / RPI;               // instructions per visit
  unsigned int acc = 0;
  const int rsub = lane / LPR, col = lane % LPR;
  for (uint64_t v = 0; v * SEG16 < chunk16; v++) {
    uint4 t[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
      const uint64_t row = (uint64_t)k * RPI + rsub;
      uint64_t gi = wave_base + row * chunk16 + v * SEG16 + col;
      gi = gi < n16 ? gi : n16 - 1;
      t[k] = src[gi];
    }
#pragma unroll
    for (int k = 0; k < NI; k++) acc ^= t[k].x ^ t[k].y ^ t[k].z ^ t[k].w;
  }
  if (acc == 0x12345u) sink[0] = acc;
}

template <int SEG16>
double run(const uint4* d, uint64_t n16, uint64_t lanes, unsigned int* sink) {
  uint64_t chunk16 = (n16 + lanes - 1) / lanes;
  chunk16 = ((chunk16 + SEG16 - 1) / SEG16) * SEG16;
  uint64_t waves = (n16 + chunk16 * 64 - 1) / (chunk16 * 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_probe<SEG16>, dim3((unsigned)waves), dim3(64), 0, 0, d, n16, chunk16, sink);
  hipEventRecord(a, 0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_probe<SEG16>, dim3((unsigned)waves), dim3(64), 0, 0, d, n16, chunk16, sink);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return (double)n16 * 16 * 5 / (ms * 1e-3) / 1e9;
}

int main() {
  const uint64_t bytes = 1600000000ull; const uint64_t n16 = bytes / 16;
  uint4* d; unsigned int* sink; hipMalloc(&d, bytes); hipMalloc(&sink, 64); hipMemset(d, 1, bytes);
  for (uint64_t lanes : {65536ull, 131072ull, 262144ull, 524288ull}) {
    printf("lanes %8llu : 128B/visit %7.0f GB/s | 256B %7.0f | 512B %7.0f | 1024B %7.0f\n", (unsigned long long)lanes,
           run<8>(d, n16, lanes, sink), run<16>(d, n16, lanes, sink), run<32>(d, n16, lanes, sink), run<64>(d, n16, lanes, sink));
  }
  return 0;
}
