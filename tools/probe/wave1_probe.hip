// Development aid (GPU box): what ONE wave per SIMD gets on gfx950 -- the regime of k_leaf_regs (512 registers per lane).
// Cycles per operation (s_memtime around a loop, lane 0 of block 0; all 1 024 waves of the chip run the same loop).
//   dependent / independent f64 chains, the step of the exact recurrence (14 f64 instructions), scalar table loads in front
//   of their use, lane-per-row ds_read_b64 with random slots, 32-bit VALU, v_accvgpr moves.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o wave1_probe tools/probe/wave1_probe.hip ; run: ./wave1_probe [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define WPE __attribute__((amdgpu_waves_per_eu(1, 1)))

struct Out { unsigned long long cyc; double sink; };

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long stime() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// 1: N dependent fma
__global__ void __launch_bounds__(64) WPE k_dep(Out* o, int iters, double a, double b) {
  double x = (double)threadIdx.x;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) x = __builtin_fma(x, a, b);
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (x == 12345.678) o->sink = x;
}
// 2: 4 independent chains
__global__ void __launch_bounds__(64) WPE k_indep(Out* o, int iters, double a, double b) {
  double x0 = (double)threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); }
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (x0 + x1 + x2 + x3 == 12345.678) o->sink = x0;
}
// 3: the recurrence step, constants as kernel arguments (SGPRs), 16 steps per trip
__global__ void __launch_bounds__(64) WPE k_step(Out* o, int iters, double r, double kf, double hh, const unsigned long long* keys) {
  double mx = 0.0, cc = 0.0, m2 = 0.0;
  unsigned long long kk[16];
#pragma unroll
  for (int u = 0; u < 16; u++) kk[u] = keys[threadIdx.x * 16 + u];
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const double x = __builtin_fma((double)(unsigned int)(kk[u] >> 32), 4294967296.0, (double)(unsigned int)kk[u]);
      const double dx = x - mx;
      const double q = dx * r;
      const double e = __builtin_fma(-q, kf, dx);
      mx += __builtin_fma(e, r, q);
      cc += dx * hh;
      m2 += dx * (x - mx);
    }
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (mx + cc + m2 == 12345.678) o->sink = mx;
}
// 4: the same with the constants loaded from a table right in front of their use (per step: 32 bytes through the scalar cache)
template <int AHEAD>
__global__ void __launch_bounds__(64) WPE k_step_tab(Out* o, int iters, const double* __restrict__ tab, const unsigned long long* keys) {
  double mx = 0.0, cc = 0.0, m2 = 0.0;
  unsigned long long kk[16];
#pragma unroll
  for (int u = 0; u < 16; u++) kk[u] = keys[threadIdx.x * 16 + u];
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
    const double* tb = tab + 64 * (i & 15);
    if constexpr (AHEAD) {
      double rr[16], kq[16], hq[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { rr[u] = tb[4 * u]; kq[u] = tb[4 * u + 1]; hq[u] = tb[4 * u + 2]; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const double x = __builtin_fma((double)(unsigned int)(kk[u] >> 32), 4294967296.0, (double)(unsigned int)kk[u]);
        const double dx = x - mx;
        const double q = dx * rr[u];
        const double e = __builtin_fma(-q, kq[u], dx);
        mx += __builtin_fma(e, rr[u], q);
        cc += dx * hq[u];
        m2 += dx * (x - mx);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const double r = tb[4 * u], kf = tb[4 * u + 1], hh = tb[4 * u + 2];
        const double x = __builtin_fma((double)(unsigned int)(kk[u] >> 32), 4294967296.0, (double)(unsigned int)kk[u]);
        const double dx = x - mx;
        const double q = dx * r;
        const double e = __builtin_fma(-q, kf, dx);
        mx += __builtin_fma(e, r, q);
        cc += dx * hh;
        m2 += dx * (x - mx);
      }
    }
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (mx + cc + m2 == 12345.678) o->sink = mx;
}
// 5: 16 lane-per-row ds_read_b64 (row = 128 bytes of a 8 KB panel, slot offset per lane), then their sum
template <int RANDOM>
__global__ void __launch_bounds__(64) WPE k_lds(Out* o, int iters) {
  __shared__ unsigned long long ring[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) ring[i] = i * 2654435761ull;
  __syncthreads();
  const int lane = threadIdx.x;
  const unsigned int rowpart = (lane >> 3) * 1024u + (lane & 7) * 128u;
  const unsigned int a0 = RANDOM ? ((lane * 7u + 3u) * 2654435761u >> 28) : 0u;      // 0..15
  unsigned long long acc = 0;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
    unsigned long long v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const unsigned int slot = (a0 + u) & 15u;
      v[u] = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned char*>(ring) + (((i & 3) * 8192u + rowpart + slot * 8u) & 32767u));
    }
#pragma unroll
    for (int u = 0; u < 16; u++) acc += v[u];
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (acc == 12345ull) o->sink = (double)acc;
}
// 6: 32-bit VALU dependent chain, and v_mov / accvgpr traffic
__global__ void __launch_bounds__(64) WPE k_i32(Out* o, int iters, unsigned int a) {
  unsigned int x = threadIdx.x;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) x = x * 3u + a;   // v_mad / v_mul_lo + add
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (x == 12345u) o->sink = x;
}
__global__ void __launch_bounds__(64) WPE k_add32(Out* o, int iters, unsigned int a) {
  unsigned int x = threadIdx.x, y = x + 1, z = x + 2, w = x + 3;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) { x = (x + a) ^ y; y = (y + a) ^ z; z = (z + a) ^ w; w = (w + a) ^ x; }   // 8 x 4 VOP2
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (x + y + z + w == 12345u) o->sink = x;
}
// 7: the error step (fma, cvt, min, sad, max) x 16, independent inputs
__global__ void __launch_bounds__(64) WPE k_err(Out* o, int iters, double pa, double pb, unsigned int n32) {
  double xs[16];
#pragma unroll
  for (int u = 0; u < 16; u++) xs[u] = (double)(threadIdx.x * 16 + u) * 1e15;
  unsigned int emax = 0, lo = threadIdx.x;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const double f = __builtin_fma(pb, xs[u], pa);
      unsigned int pr; asm("v_cvt_u32_f64 %0, %1" : "=v"(pr) : "v"(f));
      pr = pr < n32 ? pr : n32;
      unsigned int d; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(pr), "v"(lo + (unsigned)u + (unsigned)i));
      emax = emax > d ? emax : d;
    }
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (emax == 12345u) o->sink = emax;
}

// 8: the error step with its 16 doubles in AGPRs (v_accvgpr_read x 2 in front of every fma): what k_leaf_regs' error pass pays for
//    the two thirds of the stash that live there
__global__ void __launch_bounds__(64) WPE k_err_acc(Out* o, int iters, double pa, double pb, unsigned int n32) {
  unsigned int al[16], ah[16];                      // (held in AGPRs by the constraints)
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const double x = (double)(threadIdx.x * 16 + u) * 1e15;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(al[u]) : "v"(__double2loint(x)));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ah[u]) : "v"(__double2hiint(x)));
  }
  unsigned int emax = 0, lo = threadIdx.x;
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      int xl, xh;
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(xl) : "a"(al[u]));
      asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(xh) : "a"(ah[u]));
      const double f = __builtin_fma(pb, __hiloint2double(xh, xl), pa);
      unsigned int pr; asm("v_cvt_u32_f64 %0, %1" : "=v"(pr) : "v"(f));
      pr = pr < n32 ? pr : n32;
      unsigned int d; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(pr), "v"(lo + (unsigned)u + (unsigned)i));
      emax = emax > d ? emax : d;
    }
  }
  const unsigned long long t1 = stime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (emax == 12345u) o->sink = emax;
}
// 9: 16 doubles to AGPRs per trip (the stash's writes)
__global__ void __launch_bounds__(64) WPE k_acc_write(Out* o, int iters, double a) {
  double x = (double)threadIdx.x;
  unsigned int acc[32];
  const unsigned long long t0 = stime();
  for (int i = 0; i < iters; i++) {
    x += a;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[2 * u]) : "v"(__double2loint(x) + u));
      asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc[2 * u + 1]) : "v"(__double2hiint(x)));
    }
  }
  const unsigned long long t1 = stime();
  unsigned int sum = 0;
#pragma unroll
  for (int u = 0; u < 32; u++) { unsigned int v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[u])); sum += v; }
  if (threadIdx.x == 0 && blockIdx.x == 0) { o->cyc = t1 - t0; }
  if (sum == 12345u) o->sink = (double)sum;
}

int main(int argc, char** argv) {
  const int wpc = argc > 1 ? atoi(argv[1]) : 4;
  int ncu = 256;
  CHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  const int grid = ncu * wpc;
  Out* o; CHK(hipMalloc(&o, sizeof(Out)));
  double* tab; CHK(hipMalloc(&tab, 8 * 1024 * 4));
  std::vector<double> ht(1024 * 4);
  for (int i = 0; i < 1024; i++) { ht[4 * i] = 1.0 / (i + 1); ht[4 * i + 1] = i + 1; ht[4 * i + 2] = i * 0.5; ht[4 * i + 3] = 0; }
  CHK(hipMemcpy(tab, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
  unsigned long long* keys; CHK(hipMalloc(&keys, 64 * 16 * 8));
  std::vector<unsigned long long> hk(64 * 16);
  for (size_t i = 0; i < hk.size(); i++) hk[i] = (i + 1) * 0x9E3779B97F4A7C15ull;
  CHK(hipMemcpy(keys, hk.data(), hk.size() * 8, hipMemcpyHostToDevice));
  const int iters = 2000;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  auto report = [&](const char* name, double per_trip_ops) {
    Out h; (void)hipMemcpy(&h, o, sizeof h, hipMemcpyDeviceToHost);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    // s_memtime counts at 100 MHz on this family; wall-clock cycles from the event time at an assumed 2.4 GHz are printed beside it
    printf("%-34s %9.2f memtime ticks/op   wall %8.3f us  = %7.2f cycles@2.4GHz per op\n", name, (double)h.cyc / iters / per_trip_ops, ms * 1e3,
           ms * 1e-3 * 2.4e9 / iters / per_trip_ops);
  };
#define RUN(name, ops, ...) do { CHK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(__VA_ARGS__); CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); report(name, ops); } while (0)
  printf("grid %d waves (%d per CU), %d trips\n", grid, wpc, iters);
  for (int rep = 0; rep < 2; rep++) {
    RUN("dependent v_fma_f64", 16, k_dep, dim3(grid), dim3(64), 0, 0, o, iters, 1.0000001, 0.5);
    RUN("4 independent v_fma_f64 chains", 16, k_indep, dim3(grid), dim3(64), 0, 0, o, iters, 1.0000001, 0.5);
    RUN("recurrence step, SGPR constants", 16, k_step, dim3(grid), dim3(64), 0, 0, o, iters, 0.37, 3.0, 1.0, keys);
    RUN("recurrence step, table per step", 16, (k_step_tab<0>), dim3(grid), dim3(64), 0, 0, o, iters, tab, keys);
    RUN("recurrence step, table 16 ahead", 16, (k_step_tab<1>), dim3(grid), dim3(64), 0, 0, o, iters, tab, keys);
    RUN("ds_read_b64 lane-per-row aligned", 16, (k_lds<0>), dim3(grid), dim3(64), 0, 0, o, iters);
    RUN("ds_read_b64 lane-per-row random", 16, (k_lds<1>), dim3(grid), dim3(64), 0, 0, o, iters);
    RUN("dependent 32-bit mul+add", 16, k_i32, dim3(grid), dim3(64), 0, 0, o, iters, 7u);
    RUN("32-bit VOP2 (add, xor) x 8", 16, k_add32, dim3(grid), dim3(64), 0, 0, o, iters, 7u);
    RUN("error step (5 ops)", 16, k_err, dim3(grid), dim3(64), 0, 0, o, iters, 0.5, 1e-15, 200000000u);
    RUN("error step, x from AGPRs (7 ops)", 16, k_err_acc, dim3(grid), dim3(64), 0, 0, o, iters, 0.5, 1e-15, 200000000u);
    RUN("a double to AGPRs (2 writes)", 16, k_acc_write, dim3(grid), dim3(64), 0, 0, o, iters, 0.25);
  }
  return 0;
}
