// Probe: issue cost of vector instructions on gfx950 relative to v_fma_f64 (4 waves per SIMD, 8
// independent chains per lane, so that latency is hidden and the SIMD's issue rate is what is measured).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define ITERS 4096

#define KERNEL(NAME, DECL, ...)                                                               \
  __global__ void __launch_bounds__(256) NAME(double* out, double seed) {                      \
    DECL;                                                                                       \
    for (int i = 0; i < ITERS; i++) {                                                           \
      _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { __VA_ARGS__; }                              \
    }                                                                                           \
    double acc = 0; _Pragma("unroll") for (int c = 0; c < CHAINS; c++) acc += (double)v[c];     \
    if (acc == 12345.678) out[0] = acc;                                                         \
  }

#define DV double v[CHAINS]; for (int c = 0; c < CHAINS; c++) v[c] = seed + c + threadIdx.x
#define UV unsigned int v[CHAINS]; for (int c = 0; c < CHAINS; c++) v[c] = (unsigned)seed + c + threadIdx.x

KERNEL(k_fma, DV, asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[c]) : "v"(seed)))
KERNEL(k_add, DV, asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
KERNEL(k_mul, DV, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
KERNEL(k_max, DV, asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
KERNEL(k_min, DV, asm volatile("v_min_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed)))
KERNEL(k_floor, DV, asm volatile("v_floor_f64 %0, %0" : "+v"(v[c])))
KERNEL(k_rcp, DV, asm volatile("v_rcp_f64 %0, %0" : "+v"(v[c])))
KERNEL(k_cmp, DV, asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v[c]), "v"(seed) : "vcc"))
KERNEL(k_cvt_f64_u32, DV, { unsigned int t = (unsigned int)c; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[c]) : "v"(t)); })
KERNEL(k_cvt_u32_f64, DV, { unsigned int t; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(t) : "v"(v[c])); asm volatile("" :: "v"(t)); })
KERNEL(k_ldexp, DV, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v[c])))
KERNEL(k_mov64, DV, asm volatile("v_mov_b64 %0, %1" : "=v"(v[c]) : "v"(seed)))
KERNEL(k_cndmask, UV, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"((unsigned)seed) : ))
KERNEL(k_add_u32, UV, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_sad_u32, UV, asm volatile("v_sad_u32 %0, %0, %1, 0" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_max_u32, UV, asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_cmp_u64, DV, asm volatile("v_cmp_eq_u64 vcc, %0, %1" : : "v"(v[c]), "v"(seed) : "vcc"))
KERNEL(k_salu, UV, { unsigned int t; asm volatile("s_add_u32 %0, %1, 1" : "=s"(t) : "s"(i) : "scc"); asm volatile("" :: "s"(t)); })
KERNEL(k_cnd_vcc_set, UV, { if (c == 0 && i == 0) asm volatile("s_mov_b64 vcc, exec" ::: "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"((unsigned)seed)); })
KERNEL(k_cnd_e64, UV, { unsigned long long m = 0x5555555555555555ull; asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[c]) : "v"((unsigned)seed), "s"(m)); })
KERNEL(k_cnd_2dst, UV, { unsigned long long m = 0x5555555555555555ull; unsigned int t; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t) : "v"(v[c]), "v"((unsigned)seed), "s"(m)); asm volatile("" :: "v"(t)); })
KERNEL(k_and_b32, UV, asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_or_b32, UV, asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_lshl_b32, UV, asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[c])))
KERNEL(k_mov_b32, UV, asm volatile("v_mov_b32 %0, %1" : "=v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_or3_b32, UV, asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_add_f32, UV, asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_fma_f32, UV, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[c]) : "v"((unsigned)seed)))
KERNEL(k_cmp_u32, UV, asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(v[c]), "v"((unsigned)seed) : "vcc"))
KERNEL(k_cmp_u32_e64, UV, { unsigned long long m; asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "v"(v[c]), "v"((unsigned)seed)); asm volatile("" :: "s"(m)); })
KERNEL(k_readlane, UV, { unsigned int t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(v[c])); asm volatile("" :: "s"(t)); })
KERNEL(k_dpp_mov, UV, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[c])))
KERNEL(k_branch, UV, { asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n v_add_u32 %1, %1, 1\n1:" : : "s"(i), "v"(v[c]) : "scc"); })
KERNEL(k_saveexec, UV, { unsigned long long sv; asm volatile("s_and_saveexec_b64 %0, vcc\n v_add_u32 %1, %1, 1\n s_or_b64 exec, exec, %0" : "=&s"(sv), "+v"(v[c]) : : "scc"); })
KERNEL(k_cnd_e32_dst, UV, { unsigned int t; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t) : "v"(v[c]), "v"((unsigned)seed)); asm volatile("" :: "v"(t)); })
KERNEL(k_cmp_cnd_e32, UV, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"((unsigned)seed) : "vcc"))
KERNEL(k_cmp_cnd_e64, UV, { unsigned long long m; asm volatile("v_cmp_lt_u32_e64 %1, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(v[c]), "=&s"(m) : "v"((unsigned)seed)); })
KERNEL(k_cmp_2cnd_e32, UV, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(v[c]) : "v"((unsigned)seed) : "vcc"))
KERNEL(k_cmpf_2cnd_e32, DV, { unsigned int lo, hi; asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, 0, 1, vcc\n v_cndmask_b32 %1, 0, 1, vcc" : "=v"(lo), "=v"(hi) : "v"(v[c]), "v"(seed) : "vcc"); asm volatile("" :: "v"(lo), "v"(hi)); })
// VALU and SALU interleaved: do they share the issue slot?
KERNEL(k_fma_salu, DV, { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[c]) : "v"(seed)); unsigned int t; asm volatile("s_add_u32 %0, %1, 1" : "=s"(t) : "s"(i) : "scc"); asm volatile("" :: "s"(t)); })

template <typename F>
double timeit(F k, double* out) {
  const int blocks = 256 * 4 * 2;   // 4 waves per block, 16 waves per CU = 4 per SIMD (x2 rounds)
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.5);
  hipEventRecord(a, 0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.5);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 3.0;
}

int main() {
  double* out; hipMalloc(&out, 64);
  const double base = timeit(k_fma, out);
#define R(NAME) printf("%-16s %8.3f ms  %5.2f x v_fma_f64\n", #NAME, timeit(NAME, out), timeit(NAME, out) / base)
  R(k_fma); R(k_add); R(k_mul); R(k_max); R(k_min); R(k_floor); R(k_rcp); R(k_cmp); R(k_cvt_f64_u32); R(k_cvt_u32_f64);
  R(k_ldexp); R(k_mov64); R(k_cndmask); R(k_add_u32); R(k_sad_u32); R(k_max_u32); R(k_cmp_u64); R(k_salu); R(k_fma_salu);
  R(k_cnd_vcc_set); R(k_cnd_e64); R(k_cnd_2dst); R(k_and_b32); R(k_or_b32); R(k_lshl_b32); R(k_mov_b32); R(k_or3_b32); R(k_add_f32); R(k_fma_f32);
  R(k_cnd_e32_dst); R(k_cmp_cnd_e32); R(k_cmp_cnd_e64); R(k_cmp_2cnd_e32); R(k_cmpf_2cnd_e32);
  R(k_cmp_u32); R(k_cmp_u32_e64); R(k_readlane); R(k_dpp_mov); R(k_branch); R(k_saveexec);
  return 0;
}
