// rmi_regs.hip.h -- the REGISTER-RESIDENT leaf kernel of the exact leaf path (gfx950, wave64): pipeline 4.
//
//   k_leaf_regs    what k_leaf_lanes does (exact per-leaf SLR, linear.rs:12-59 on the container of two_layer.rs:52-90, then
//                  the last-level error pass, two_layer.rs:207-217, lower_bound_correction.rs:104-119, then the leaf's row)
//                  with ONE read of the keys.
//
// Why registers.  The error of a key needs the final (alpha, beta) of its leaf, known only behind the leaf's last key, and
// bit-identical (alpha, beta) need the order-dependent recurrence: one sequential chain per leaf, 64 chains per wave in
// lockstep (rmi_lanes.hip.h).  The keys of a wave's 64 leaves (64 x ~191 x 8 B = 98 KB) must therefore stay on chip between
// the two phases, for every wave in flight: with all 1 024 SIMDs busy that is 100 MB.  The chip has 40 MB of LDS and 32 MB of
// L2 -- and 128 MB of vector registers.  So: ONE wave per SIMD with the full 512-register budget; a lane keeps the keys of its
// leaf (as the doubles the recurrence consumed: `key as f64`, models/mod.rs:83) in registers while it walks them -- the first
// RG_STASH = 192 steps in registers, the following <= 48 in the LDS ring the keys arrive through, which is simply not
// refilled at the end of a group of leaves -- and replays them for the error pass.  Nothing is read twice.
//
// Data movement.  The keys arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, the register file belongs to the
// stash): as in k_leaf_lanes a row (= a leaf's container) is fetched in aligned 128-byte lines, a panel = 16 steps of all 64
// rows = 8 instructions = 8 KB; the ring holds 4 panels (32 KB per wave, 4 waves per CU).  The destination of an LDS-DMA
// instruction is linear in the lane (M0 + 16 lane): instruction i of a panel carries the rows 8 g + i of the 8 groups g of 8 loader
// lanes, so that a loader lane's rows are the lanes of its own group and their offsets reach it by DPP; lane r reads slot
// (a0_r + k) of its row: the panel that slot lies in differs between lanes that have and have not crossed a line boundary
// (one select per step).  The wave is PERSISTENT: it takes every gridDim.x-th group of 64 leaves, and the first panels of the
// next group are requested before the register part of this group's error pass.
//
// Code shape.  The walk is rolled over blocks of 16 steps (rmi_regs_block.inc.h); for a group whose containers all hold more than
// RG_UBLK x 16 points the first RG_UBLK blocks are written out per block (static registers of the stash, no masks).  A lone
// wave has nobody to fill an LDS or scalar-cache round trip: wave-wide maxima and the loaders' row offsets go by DPP, the
// step constants are requested a quarter block ahead, the single keys a group needs late are parked in LDS by DMA.
//
// What takes this path: groups of 64 consecutive leaves whose containers hold at most RG_FARPTS = 1 008 points (the lanes with more
// than RG_MAXPTS = 240 go on from the key array by themselves), cover their leaves (every leaf but the one behind the split,
// two_layer.rs:166-169) and hold no duplicate key (found while walking: compared as doubles, a superset of key equality);
// 8-byte keys; linear leaves.  Every other group is put on a list and runs the body of k_leaf_lanes in k_leaf_lanes_listed,
// launched behind this kernel: the same bits either way.  The leaf ends (widening, rows, counts, aggregates): k_regs_finalize.
#pragma once
#include <type_traits>

#include "rmi_lanes.hip.h"

namespace rmi {

#ifndef RG_STASH_N
#define RG_STASH_N 192
#endif
constexpr int RG_ROW = 16;                          // steps (keys) per row and panel: one aligned 128-byte line
constexpr int RG_RING = 4;                          // panels in the LDS ring
constexpr int RG_PANEL_B = 64 * RG_ROW * 8;         // bytes of a panel: 64 rows x 128 B
constexpr int RG_STASH = RG_STASH_N;                       // steps whose keys stay in registers
#ifndef RG_STASH2_N
#define RG_STASH2_N 160          // (U32, 400 M keys in 2^21 leaves: 128: 535 us, 160: 510 us, 176: 548 us with 21 spilled registers; one wave per SIMD: 655 us)
#endif
constexpr int RG_STASH2 = RG_STASH2_N;                     // ... of k_leaf_regs<K, 2> (two waves per SIMD: half the registers)
constexpr int RG_MAXPTS = RG_STASH_N + 48;                      // longest container whose steps behind the stash are still in the ring at the end
constexpr int RG_FARPTS = 1008;                                 // longest container of a group that takes the register path at all
constexpr int RG_NBLK = RG_MAXPTS / RG_ROW;         // 15 blocks of 16 steps
constexpr int RG_SBLK = RG_STASH / RG_ROW;          // 12 of them stashed in registers
static_assert(RG_NBLK - RG_SBLK <= RG_RING - 1, "the tail of a group must still be in the ring when its fit ends");
static_assert(RG_FARPTS + 16 <= 1024, "the step table (RG_TMAX) covers the walk");

template <int I, int N, typename F>
__device__ __forceinline__ void rg_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); rg_static_for<I + 1, N>(f); }
}

// Timing experiments (results wrong), compiled in by macro: RG_KO & 1 no recurrence arithmetic, & 2 no error pass, & 4 no loads.
#ifndef RG_KO
#define RG_KO 0
#endif
#ifndef RG_PROF
#define RG_PROF 0                // 1: cycles per phase of a group, summed over the waves into prof[] (printed by the host at destroy)
#endif
#ifndef RG_UBLK
#define RG_UBLK 9                // blocks of the walk whose code is written out per block (STATIC in k_leaf_regs); 0: none
#endif
#ifndef RG_LONG_F32
#define RG_LONG_F32 0            // 1: the LONG variant stashes (float)(x - x0) for integer keys -- 384 steps in registers, the error pass over them certified per
                                 // leaf, flagged leaves redone by k_regs_finalize.  Measured SLOWER (C4's shard shape: kernel 278 against 254 us with the doubles and
                                 // the second trip through the ring): the kernel is bound by its instructions, not by its reads, and the float path costs a conversion
                                 // and three operations of the certificate per step more.  Kept as a build switch (tests: tools/build_var.sh).
#endif
#ifndef RG_PRE
#define RG_PRE 3                 // LONG: banks of the stash replayed in front of the walk through the ring (while its first panels land); 0: none
#endif
#ifndef RG_WALK_PIPE
#define RG_WALK_PIPE 1           // LONG: the walk through the ring asks for the next half block's keys before this half's steps run
#endif
#ifndef RG_DIAG
#define RG_DIAG 0                // & 1 the constants of the first half block for all, & 2 no duplicate test, & 8 no lane ever tests, & 16 only bank 0 stashed, & 64 all panels from one place (cache hits: results wrong)
#endif

// The step tables of this kernel, four arrays of RG_TMAX doubles for the running count k = i + 1: RN(1 / k), the tail of the
// reciprocal (1 / k = r + rl to 2^-105: div_by_count2 of rmi_stream.hip.h, one FMA fewer than div_by_count -- an FMA costs 7.3
// cycles on this chip, an addition or a multiplication 4), (k - 1) / 2, k.  The walk takes them a quarter block (4 steps) at a time,
// one ahead: 128 bytes per quarter, two scalar loads.
static __global__ void __launch_bounds__(256) k_regs_table(double* __restrict__ tab, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    const double nf = (double)(i + 1), r = 1.0 / nf;
    double* const t = tab + (i >> 2) * 16 + (i & 3);                     // (the 16 doubles of a quarter block -- 4 steps -- lie together)
    t[0] = r; t[4] = recip_tail(nf, r); t[8] = (double)i * 0.5; t[12] = nf;
  }
}
constexpr int RG_TMAX = 1024;                       // steps the table covers

// One panel: 8 LDS-DMA instructions, instruction i = rows 8 i + lane / 8, 128 bytes each; off[i] = this lane's byte offset
// from `kb` (a wave-uniform pointer) of its 16-byte piece.  The destination is M0 + 16 lane.  M0 is not the compiler's to
// keep around an asm statement (cdna_hip_programming.md section 5): saved and restored here.
#define RG_DMA8(NTS)                                                                                                   \
  asm volatile("s_mov_b32 %[keep], m0\n\t"                                                                             \
               "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[kb]" NTS "\n\t"                     \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[kb]" NTS "\n\t"                  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o2], %[kb]" NTS "\n\t"                  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o3], %[kb]" NTS "\n\t"                  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o4], %[kb]" NTS "\n\t"                  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o5], %[kb]" NTS "\n\t"                  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o6], %[kb]" NTS "\n\t"                  \
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o7], %[kb]" NTS "\n\t"                  \
               "s_mov_b32 m0, %[keep]"                                                                                 \
               : [keep] "=&s"(keep)                                                                                    \
               : [lds] "s"(lds), [kb] "s"(kb), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]), \
                 [o4] "v"(off[4]), [o5] "v"(off[5]), [o6] "v"(off[6]), [o7] "v"(off[7])                                \
               : "memory", "scc")
template <bool NT>
__device__ __forceinline__ void rg_dma_panel(const void* kb, unsigned int lds, const unsigned int (&off)[8]) {
  unsigned int keep;
  if constexpr (NT) RG_DMA8(" nt"); else RG_DMA8("");
}
// 4-byte keys: a panel is 64 rows of 64 bytes (16 keys: HALF a line; the other half comes with the next panel and hits the L2 -- plain loads,
// the non-temporal hint would have evicted it): 4 instructions, instruction i = rows 4 g + i of the 16 groups g of 4 loader lanes
__device__ __forceinline__ void rg_dma_panel4(const void* kb, unsigned int lds, const unsigned int (&off)[4]) {
  unsigned int keep;
  asm volatile("s_mov_b32 %[keep], m0\n\t"
               "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[kb]\n\t"
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[kb]\n\t"
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o2], %[kb]\n\t"
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o3], %[kb]\n\t"
               "s_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep)
               : [lds] "s"(lds), [kb] "s"(kb), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3])
               : "memory", "scc");
}
// one dword per lane to LDS (M0 + 4 lane): the single keys and the leaf boundaries a group needs late are parked in LDS this way --
// no register holds them meanwhile and, more important, no wait of the compiler's stands behind a panel request for them
__device__ __forceinline__ void rg_dma_dword(const void* base, unsigned int lds, unsigned int voff) {
  unsigned int keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dword %[o], %[kb]\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep)
               : [lds] "s"(lds), [kb] "s"(base), [o] "v"(voff)
               : "memory");
}
// the loads of rg_dma_panel are not in the compiler's count: waits for them are written here (at most N VMEM operations
// outstanding; operations complete in order, so whatever else the compiler has in flight only makes a wait longer)
__device__ __forceinline__ unsigned long long rg_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
template <int N> __device__ __forceinline__ void rg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the maximum over the wave, in an SGPR: four DPP steps inside the rows of 16 lanes (swap neighbours, swap pairs, mirror the half
// rows, mirror the rows), then the four rows' values by readlane.  (As six __shfl_xor it is six LDS round trips, each waited
// out by a wave that has nobody to fill them: ~1 000 cycles per group for the two maxima a group descriptor needs.)
__device__ __forceinline__ unsigned int rg_wave_max(unsigned int v) {
  auto step = [](unsigned int x, auto ctrl_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value;
    const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, false);
    return o > x ? o : x;
  };
  v = step(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1, 0, 3, 2]
  v = step(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2, 3, 0, 1]
  v = step(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
  v = step(v, std::integral_constant<int, 0x140>{});     // row_mirror
  const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned int r2 = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned int a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// `key as f64` from the two halves of the key as they lie in LDS (models/mod.rs:83; see KeyTraits<uint64_t>::as_float -- written
// on the halves because the compiler turns (double)(uint32_t)(k >> 32) back into a 64-bit conversion with one addition more)
__device__ __forceinline__ unsigned int rg_block_index(unsigned int b) { return b; }
template <int B> __device__ __forceinline__ unsigned int rg_block_index(std::integral_constant<int, B>) { return (unsigned int)B; }
template <typename K> __device__ __forceinline__ unsigned long long key_to_bits_rg(K k) {
  if constexpr (std::is_same<K, double>::value) return (unsigned long long)__double_as_longlong(k); else return (unsigned long long)k;
}
template <typename K> __device__ __forceinline__ double rg_as_float(unsigned int v) { return (double)v; }   // 4-byte keys: one exact conversion
__device__ __forceinline__ unsigned int rg_raw_lo(uint2 v) { return v.x; }
__device__ __forceinline__ unsigned int rg_raw_lo(unsigned int v) { return v; }
template <typename K> __device__ __forceinline__ double rg_as_float(uint2 v) {
  if constexpr (std::is_same<K, double>::value) return __hiloint2double((int)v.y, (int)v.x);
  else return __builtin_fma((double)v.y, 4294967296.0, (double)v.x);
}

__device__ __forceinline__ double rg_fract(double v) { return __builtin_amdgcn_fract(v); }   // v_fract_f64: v - floor(v), in [0, 1)

// ---------------------------------------------------------------------------------------------
// k_leaf_regs
// ---------------------------------------------------------------------------------------------
// LONG: the variant for groups of LONG leaves (averages above 208 keys a leaf: C4's shard shape, 381).  The walk goes on through the
// ring behind the stash -- up to RG_FARPTS points per container, the ring refilled all the way -- and the error steps behind the stash
// come through the ring ONCE MORE after the fit (panels 12 ...: 2 - 192 / n reads of a key instead of k_leaf_lanes' 2; at 381 keys a
// leaf 1.5).  Why not everything on chip: 65 536 chains must be in flight to fill the device's 1 024 SIMDs with lockstep waves, and
// their keys between fit and error pass are 65 536 x n x 8 B -- 200 MB at n = 381, more than every register and LDS byte of the chip
// (172 MB); above ~330 keys a leaf NO exact one-read design exists at full lane efficiency.
// LONG == 2 (4-byte keys only): TWO waves per SIMD, 256 registers each -- the stash holds the RAW keys (one register a step; the conversion
// to the double is exact and is made again in the error pass), RG_STASH2 = 128 steps of them, and everything behind goes the LONG way.  The
// experiment behind it: a lone wave issues ~57 % of the cycles (the recurrence's dependent chain); a second wave on the SIMD fills the gaps.
template <typename K, int LONG>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LONG == 2 ? 2 : 1, LONG == 2 ? 2 : 1))) k_leaf_regs(const K* __restrict__ keys, Span sp,
                                                   const unsigned long long* __restrict__ leaf_start,
                                                   DevState* __restrict__ st, double* __restrict__ params,
                                                   const double* __restrict__ rtab, const double* __restrict__ rtab4, SgList fl, unsigned int long_min,
                                                   unsigned long long* __restrict__ leaf_maxerr,
                                                   unsigned long long* __restrict__ leaf_run, uint64_t L,
                                                   unsigned long long* __restrict__ leaf_err,
                                                   unsigned long long* __restrict__ leaf_count,
                                                   unsigned char* __restrict__ rows, StatsPartial* __restrict__ partials, RootP vr,
                                                   PeerRows peers, unsigned int ntiles, unsigned int slow,
                                                   unsigned int* __restrict__ slow_list, unsigned int* __restrict__ slow_count,
                                                   unsigned long long* __restrict__ prof,
                                                   K* __restrict__ bnext, K* __restrict__ bprev, unsigned char* __restrict__ tile_slow,
                                                   unsigned int* __restrict__ tile_queue) {
  static_assert(sizeof(K) == 8 || sizeof(K) == 4, "8-byte keys, or 4-byte keys in half-line panels");
  constexpr unsigned int KB = (unsigned int)sizeof(K);                 // bytes of a key
  constexpr unsigned int ROWB = (unsigned int)RG_ROW * KB;             // bytes of a row's share of a panel: a 128-byte line, or half of one
  constexpr unsigned int PANEL_B = 64u * ROWB;                         // bytes of a panel: 8 KB / 4 KB
  using RAW = typename std::conditional<sizeof(K) == 8, uint2, unsigned int>::type;   // a key as it lies in the ring
  constexpr int NI = KB == 8u ? 8 : 4;                                 // LDS-DMA instructions of a panel (the hand-written waits count them)
  constexpr bool NT = true;                                          // non-temporal LDS-DMA loads (plain ones measured equal: the switch is gone)
  constexpr unsigned int WALK = LONG ? (unsigned int)RG_FARPTS : (unsigned int)RG_MAXPTS;   // steps of a container that come through the ring
  constexpr bool DIVK = !UseRecipTable<K>::value;                     // f64 keys: plain IEEE division
  // LONG, integer keys: the stash holds (float)(x - x0), x0 = the double of the container's first key -- ONE register a step, 384 steps.
  // The error pass over the stash then works on x~ with |x~ - x| <= 2^-23.9 (x - x0) and is CERTIFIED per leaf (rg_stash_err below): a
  // leaf one of whose predictions lies within the bound of an integer is flagged and its error pass redone from the key array by
  // k_regs_finalize (about 2 % of the leaves at 381 keys a leaf); every other leaf's maximum is the exact one.
  constexpr bool W2 = LONG == 2;                                      // two waves per SIMD: raw 4-byte keys in the stash
  static_assert(!W2 || sizeof(K) == 4, "the two-wave variant stashes raw 4-byte keys");
  constexpr bool F32 = LONG && !W2 && !DIVK && (RG_LONG_F32 != 0);
  using XT = typename std::conditional<W2, unsigned int, typename std::conditional<F32, float, double>::type>::type;
  constexpr int STASH = W2 ? RG_STASH2 : (F32 ? 2 * RG_STASH : RG_STASH);   // steps whose keys stay in registers
  constexpr int SBLK = STASH / RG_ROW;
  static_assert(STASH % RG_ROW == 0 && SBLK > RG_PRE, "whole banks");
  __shared__ __attribute__((aligned(1024))) unsigned char ringc[RG_RING * 64 * RG_ROW * sizeof(K)];   // 32 KB (16 KB for 4-byte keys): 4 waves per CU
  __shared__ unsigned int park[10][64];                              // k_hi, k_lom1, k_next, k_prev (two words each), next group's s, e
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned int ring_lds = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(lds_byte*)ringc);
  const unsigned int park_lds = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(lds_byte*)&park[0][0]);

  const int lane = threadIdx.x;
  // A panel in LDS: instruction i of a request carries row 8 g + i for each group g of 8 loader lanes (LDS: i * 1 KB + g * 128 B),
  // so that a loader lane's 8 rows are the 8 lanes of its own group -- their offsets reach it by DPP, not through LDS.
  const unsigned int rowpart = KB == 8u ? (unsigned int)(lane & 7) * 1024u + (unsigned int)(lane >> 3) * 128u    // this lane's row inside a panel
                                        : (unsigned int)(lane & 3) * 1024u + (unsigned int)(lane >> 2) * 64u;   // (4-byte keys: 4 loader lanes a row)
  const unsigned int piece = (unsigned int)(lane & (KB == 8u ? 7 : 3)) * 16u;   // its 16-byte piece of a row as a loader
  const unsigned int n32 = (unsigned int)sp.n;
  const uint64_t split_idx = st->split_idx, split_target = st->split_target;
  // duplicate-heavy keys (DevState::regs_dups): every group would meet a duplicate and go on the list after its panels were requested
  // and its walk begun; listed at once instead, k_leaf_lanes_listed takes them all
  slow = (slow & 1u) | (((slow & 2u) && st->regs_dups * (64ull * 16ull) > (unsigned long long)(sp.leaf_hi - sp.leaf_lo)) ? 1u : 0u);   // (bit 1 of the argument: the routing is on)

  unsigned long long pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                   // (RG_PROF)
  unsigned long long pt = 0;
  auto mark = [&](int i) { if (RG_PROF) { const unsigned long long t = rg_now(); pf[i] += t - pt; pt = t; } };
  unsigned long long pst = 0;                                                   // (RG_PROF 2: inside the fit's blocks)
  auto sub0 = [&]() { if (RG_PROF > 1) pst = rg_now(); };
  auto sub = [&](int i) { if (RG_PROF > 1) { const unsigned long long t = rg_now(); pf[i] += t - pst; pst = t; } };
  // ---- a group of 64 leaves ("tile"): per-lane view and what is wave-uniform
  struct Tile {
    bool fast;                          // uniform: the register path takes it
    bool valid, act;                    // leaf exists / container walked here
    int ck;
    unsigned int s, e, lo, npts;        // leaf [s, e), container [lo, lo + npts)
    unsigned int a0;                    // slot of the container's first point in its first line
    unsigned int maxlen, lastp;         // uniform: longest walk through the ring (<= RG_MAXPTS steps), last panel any row needs
    unsigned int maxfar;                // uniform: longest container (> RG_MAXPTS: those lanes go on from the key array)
    bool ulong;                         // uniform: every container walked here has more than RG_UBLK * 16 points
    const K* kb;                        // uniform: keys + wave base (line aligned)
    unsigned int off, lim;              // this lane's row: byte offset from kb of its first line, of its last
  };
  // (all lanes: a loader lane serves the rows 8 (lane / 8) + i, whatever its own leaf does).  Every lane clamps its OWN row's offset
  // (a finished row keeps re-reading its last line); lane 8 g + i's value reaches the 8 lanes of group g in two DPP moves: lane
  // i mod 4 of every quad to its quad, then the right quad of the pair to both.  (Through LDS it was four round trips per panel.)
  auto issue_panel = [&](const K* kb, unsigned int p, unsigned int off_own, unsigned int lim_own) {
    const unsigned int o = off_own + p * ROWB;
    const int mine = (int)(o < lim_own ? o : lim_own);
    if constexpr (KB == 4u) {
      // (a loader lane's 4 rows are the lanes of its own quad: one DPP move each)
      unsigned int off4[4];
      rg_static_for<0, 4>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        off4[i] = (unsigned int)__builtin_amdgcn_update_dpp(mine, mine, i * 0x55, 0xF, 0xF, false) + piece;
      });
      rg_dma_panel4(kb, ring_lds + (p & (unsigned int)(RG_RING - 1)) * PANEL_B, off4);
      return;
    }
    unsigned int off[8];
    rg_static_for<0, 8>([&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value;
      constexpr int QP = (i & 3) * 0x55;                                  // quad_perm [a, a, a, a]
      const int t = __builtin_amdgcn_update_dpp(mine, mine, QP, 0xF, 0xF, false);
      const int g = (i < 4) ? __builtin_amdgcn_update_dpp(t, t, 0x114, 0xF, 0xA, false)      // row_shr:4 into the upper quads
                            : __builtin_amdgcn_update_dpp(t, t, 0x104, 0xF, 0x5, false);     // row_shl:4 into the lower quads
      off[i] = (unsigned int)g + piece;
    });
    rg_dma_panel<NT>((RG_DIAG & 64) ? (const K*)keys + 4096 : kb, ring_lds + (p & (unsigned int)(RG_RING - 1)) * PANEL_B, off);   // (& 64: every group's panels from the same 100 KB)
  };
  unsigned int nxt_off = 0u, nxt_lim = 0u;                             // row descriptors of the group requested last (this lane's row)
  // descriptor of tile `tl` from its leaves' boundaries; a fast tile's first panels are requested at once
  auto make_tile = [&](unsigned int tl, uint64_t s, uint64_t e) -> Tile {
    Tile t;
    const uint64_t j = sp.leaf_lo + (uint64_t)tl * 64 + (uint64_t)lane;
    t.valid = j < sp.leaf_hi;
    uint64_t lo = 0, hi = 0;
    t.ck = t.valid ? leaf_container(j, s, e, sp.n, split_idx, split_target, lo, hi) : 0;
    uint64_t wb;
    {
      const unsigned int s0l = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)s);
      const unsigned int s0h = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(s >> 32));
      const uint64_t s0 = ((uint64_t)s0h << 32) | s0l;
      wb = s0 > sp.rd_lo ? s0 - 1 : sp.rd_lo;
    }
    wb -= (uint64_t)((reinterpret_cast<uintptr_t>(keys + wb) / sizeof(K)) % (uintptr_t)RG_ROW);
    t.kb = keys + wb;
    const uint64_t np64 = t.ck == 2 ? hi - lo + 1 : 0;
    const bool handed = t.valid && t.ck == 2 && np64 + 1 > (uint64_t)long_min;
    const bool ok = !t.valid || (!handed && (t.ck == 2 ? (np64 <= (uint64_t)RG_FARPTS && lo <= s + 1 && e <= hi + 1) : e == s));
    t.fast = __all(ok) && !slow;
    t.act = t.valid && t.ck == 2;
    t.s = (unsigned int)s; t.e = (unsigned int)e; t.lo = (unsigned int)lo; t.npts = t.act ? (unsigned int)np64 : 0u;
    const unsigned int rel = t.act ? (unsigned int)(lo - wb) : 0u;
    t.a0 = rel & (unsigned int)(RG_ROW - 1);
    // the walk through the ring covers RG_MAXPTS = 240 steps (panels 0 .. 15: the ring then still holds the panels behind the
    // stash); the few lanes with more points go on from there by themselves (far_fit below)
    const unsigned int wl = t.npts < WALK ? t.npts : WALK;
    t.maxfar = rg_wave_max(t.npts);
    t.maxlen = t.maxfar < WALK ? t.maxfar : WALK;
    t.lastp = rg_wave_max(t.act ? (t.a0 + wl - 1u) >> 4 : 0u);
    t.ulong = __all(!t.act || t.npts > (unsigned int)(RG_UBLK * RG_ROW));
    if (t.fast) {
      nxt_off = (rel & ~(unsigned int)(RG_ROW - 1)) * KB;
      nxt_lim = t.act ? ((rel + wl - 1u) & ~(unsigned int)(RG_ROW - 1)) * KB : 0u;
      t.off = nxt_off; t.lim = nxt_lim;
      unsigned long long m0t = 0;
      if (RG_PROF) m0t = rg_now();
      if (!(RG_KO & 4)) {
        // the single keys of this group's end -- the container's last key (Q1), the key in front of the container (FixDups offset
        // of its first point), the keys on either side of the leaf (finalize_one) -- in FRONT of the panels: landed before them
        const K* const kbm = t.kb - RG_ROW;                              // (offsets from one line in front of the wave base: never negative)
        const unsigned int o_hi = t.act ? (rel + t.npts - 1u + (unsigned int)RG_ROW) * KB : (unsigned int)RG_ROW * KB;
        const unsigned int o_lom1 = (t.act && lo > sp.rd_lo) ? (rel - 1u + (unsigned int)RG_ROW) * KB : (unsigned int)RG_ROW * KB;
        const unsigned int o_next = (t.valid && e < sp.n) ? ((unsigned int)(e - wb) + (unsigned int)RG_ROW) * KB : (unsigned int)RG_ROW * KB;
        const unsigned int o_prev = (t.valid && s > 0) ? ((unsigned int)(s - wb) - 1u + (unsigned int)RG_ROW) * KB : (unsigned int)RG_ROW * KB;
        constexpr unsigned int HW = KB == 8u ? 4u : 0u;                  // (the key's second word; 4-byte keys: the same word again, nothing behind it is read)
        rg_dma_dword(kbm, park_lds + 0u * 256u, o_hi); rg_dma_dword(kbm, park_lds + 1u * 256u, o_hi + HW);
        rg_dma_dword(kbm, park_lds + 2u * 256u, o_lom1); rg_dma_dword(kbm, park_lds + 3u * 256u, o_lom1 + HW);
        rg_dma_dword(kbm, park_lds + 4u * 256u, o_next); rg_dma_dword(kbm, park_lds + 5u * 256u, o_next + HW);
        rg_dma_dword(kbm, park_lds + 6u * 256u, o_prev); rg_dma_dword(kbm, park_lds + 7u * 256u, o_prev + HW);
#pragma unroll
        for (unsigned int p = 0; p < (unsigned int)RG_RING; p++)
          if (p <= t.lastp) issue_panel(t.kb, p, nxt_off, nxt_lim);
      }
      if (RG_PROF) pf[6] += rg_now() - m0t;
    }
    return t;
  };
  // the boundaries of tile `tl`'s leaves to the parking rows 8, 9 (low words: the fused path has n < 2^32)
  auto request_bounds = [&](unsigned int tl) {
    const uint64_t j0 = sp.leaf_lo + (uint64_t)tl * 64;
    const uint64_t left = sp.leaf_hi - j0;                               // > 0
    const unsigned int jl = (uint64_t)lane < left ? (unsigned int)lane : (unsigned int)(left - 1);
    const unsigned long long* const base = leaf_start + j0;
    if (!(RG_KO & 4)) { rg_dma_dword(base, park_lds + 8u * 256u, jl * 8u); rg_dma_dword(base, park_lds + 9u * 256u, jl * 8u + 8u); }
  };
  auto parked_key = [&](int row) -> K {
    if constexpr (KB == 4u) return (K)park[row][lane];
    else {
      const unsigned long long v = ((unsigned long long)park[row + 1][lane] << 32) | (unsigned long long)park[row][lane];
      return bits_to_key<K>(v);
    }
  };
  auto load_bounds = [&](unsigned int tl, uint64_t& s, uint64_t& e) {
    const uint64_t j = sp.leaf_lo + (uint64_t)tl * 64 + (uint64_t)lane;
    s = 0; e = 0;
    if (j < sp.leaf_hi) { s = leaf_start[j]; e = leaf_start[j + 1]; }
  };

  // The loop is skewed by the hand-over: iteration i finishes tile i - 1 around the descriptor (and first panels) of tile i,
  // so that there is ONE copy of everything in the code.
  // Which groups a wave takes: its first two by its number, the others from a counter (a group with a long leaf, 2 % of them on
  // the metric configuration, takes its wave 20 us longer: dealt out statically, the waves with two or three of those end 40-60 us
  // behind the others).  A group's number is needed two hand-overs ahead (its leaves' boundaries are requested one ahead).
  unsigned int tile = blockIdx.x;                                      // the group the NEXT hand-over describes
  if (tile >= ntiles) return;
  unsigned int tile2 = tile + gridDim.x;                               // ... and the one behind it
  unsigned int done_tile = 0u;                                         // the group `cur` describes (when `have`)
  uint64_t sn, en;
  load_bounds(tile, sn, en);
  Tile cur;                                                            // (set at the first hand-over)
  bool have = false;
  for (;;) {
    if (RG_PROF) pt = rg_now();
    const bool more = tile < ntiles;
    // (asked for at the top of the iteration, looked at behind the walk: no wait of the compiler's for it stands behind a panel request)
    unsigned int tile3 = 0xFFFFFFFFu;
    if (tile_queue == nullptr) tile3 = tile2 + gridDim.x;               // (static dealing: RMI_HIP_REGS_QUEUE=0)
    else if (lane == 0 && tile2 < ntiles) tile3 = 2u * gridDim.x + atomicAdd(tile_queue, 1u);
    unsigned int flags = 0;
    double pa = 0.0, pb = 0.0;
    const bool fast = have && cur.fast;
    unsigned int lane_j = (unsigned int)lane;
    asm volatile("" : "+v"(lane_j));                                   // (the addresses of this group's results are formed when they are stored: not kept, not spilled)
    const uint64_t j = sp.leaf_lo + (uint64_t)done_tile * 64 + (uint64_t)lane_j;
    const unsigned int a8 = cur.a0 * KB;
    // LDS address of step 16 b + qq of this lane's row: slot a0 + qq of the two-panel window that starts at panel b -- in panel
    // b (ring slot b mod 4) for the lanes that have not crossed their line's end yet, in panel b + 1 for the others
    // (the select: bit qq of `cm` says "crossed"; times the distance between the two places: two VOP3 operations, no VCC)
    const unsigned int cm = 0xFFFFu << ((unsigned int)RG_ROW - cur.a0);  // crossed at qq >= 16 - a0 (a0 = 0: never)
    auto block_base = [&](unsigned int b, unsigned int& in_b, unsigned int& delta) {
      const unsigned int sb = b & (unsigned int)(RG_RING - 1), sb1 = (b + 1u) & (unsigned int)(RG_RING - 1);
      in_b = rowpart + a8 + sb * PANEL_B;
      delta = (sb1 - sb) * PANEL_B - ROWB;                               // (wrapping: the ring's last slot is followed by its first)
    };
    auto slot_key = [&](unsigned int in_b, unsigned int delta, int qq) -> RAW {
      // (written out: from C the compiler makes 16 comparisons of it, kept in 32 SGPRs across the whole walk)
      unsigned int crossed, addr0;
      asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(crossed) : "v"(cm), "n"(qq));
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(addr0) : "v"(crossed), "s"(delta), "v"(in_b));   // (signed: the ring wraps backwards)
      const unsigned int addr = addr0 + (unsigned int)qq * KB;
      return *reinterpret_cast<const RAW*>(ringc + addr);
    };
    K k_hi = KeyTraits<K>::zero_value(), k_lom1 = KeyTraits<K>::zero_value();
    K k_next = KeyTraits<K>::max_value(), k_prev = KeyTraits<K>::zero_value();    // the two boundary keys of finalize_one
    // =========================== the error pass over the leaves' own keys ===========================
    // the leaf's keys are the steps [es, eend) of its container's walk: es = s - lo is 0 or 1 -- or -1 for the leaf behind the split
    // (two_layer.rs:166-169: its container starts behind its first key, which is the parked key in front of the container)
    const unsigned int es = cur.s - cur.lo;
    unsigned int eend = (have && cur.act) ? cur.e - cur.lo : 0u;
    unsigned int emax = 0u;
    auto err_step = [&](double x, unsigned int k) {
      const double f = __builtin_fma(pb, x, pa);                            // linear.rs:87-90
      const unsigned int pr = min(sg_cvt_u32(f), n32);                      // models/mod.rs:735-737, two_layer.rs:14-18
      emax = max(emax, sg_absdiff(pr, cur.lo + k));
    };
    // two steps at once: the conversions and absolute differences are asm statements, behind which the compiler places a wait state
    // when the next instruction reads their result -- of two interleaved steps each fills the other's, and one v_max3 closes both
    auto err_pair = [&](double x1, unsigned int k1, double x2, unsigned int k2) {
      const double f1 = __builtin_fma(pb, x1, pa), f2 = __builtin_fma(pb, x2, pa);
      const unsigned int c1 = sg_cvt_u32(f1), c2 = sg_cvt_u32(f2);
      const unsigned int d1 = sg_absdiff(min(c1, n32), cur.lo + k1), d2 = sg_absdiff(min(c2, n32), cur.lo + k2);
      emax = max(emax, max(d1, d2));
    };
    // W2 (raw 4-byte keys in the stash): the conversion is made HERE, on an opaque copy -- the stash does not change inside the loop over its
    // banks, and the compiler would otherwise convert all of it in front of the loop (twice the stash in registers: spills)
    auto err_pair_w = [&](unsigned int r1, unsigned int k1, unsigned int r2, unsigned int k2) {
      asm volatile("" : "+v"(r1), "+v"(r2));
      err_pair((double)r1, k1, (double)r2, k2);
    };
    auto err_step_w = [&](unsigned int r, unsigned int k) {
      asm volatile("" : "+v"(r));
      err_step((double)r, k);
    };
    // F32 stash: the same steps on x~ = x0 + d (d the stashed float of x - x0), as fma(beta, d, A) with A = fma(beta, x0, alpha), and
    // the certificate beside them: the largest distance of a prediction's fraction from 1/2 (three more operations a step)
    double sA = 0.0, tmax = 0.0;
    auto err_pair_f = [&](float d1, unsigned int k1, float d2, unsigned int k2) {
      // (opaque: the stash does not change inside the loop over its banks, and the compiler would compute all 384 predictions and
      //  fractions in front of it -- one block of 2 500 instructions with its results in scratch memory)
      asm volatile("" : "+v"(d1), "+v"(d2));
      const double f1 = __builtin_fma(pb, (double)d1, sA), f2 = __builtin_fma(pb, (double)d2, sA);
      const unsigned int c1 = sg_cvt_u32(f1), c2 = sg_cvt_u32(f2);
      const unsigned int e1 = sg_absdiff(min(c1, n32), cur.lo + k1), e2 = sg_absdiff(min(c2, n32), cur.lo + k2);
      emax = max(emax, max(e1, e2));
      const double g1 = __builtin_fabs(rg_fract(f1) - 0.5), g2 = __builtin_fabs(rg_fract(f2) - 0.5);
      tmax = __builtin_fmax(tmax, __builtin_fmax(g1, g2));
      // (pair by pair: left alone, the chain of maxima is rebalanced into a tree over the whole bank, every pair's result parked in scratch)
      asm volatile("" : "+v"(tmax));
    };
    auto err_step_f = [&](float d, unsigned int k) {
      asm volatile("" : "+v"(d));
      const double f = __builtin_fma(pb, (double)d, sA);
      emax = max(emax, sg_absdiff(min(sg_cvt_u32(f), n32), cur.lo + k));
      tmax = __builtin_fmax(tmax, __builtin_fabs(rg_fract(f) - 0.5));
      asm volatile("" : "+v"(tmax));
    };
    // ---- hand-over: the ring is free -- the next group's descriptor and its first panels, under the rest of this group's work
    Tile nxt = cur;
    auto hand_over = [&]() {
      asm volatile("" : "+v"(tile3));                                   // (the counter's answer is waited for HERE, in front of the panel requests)
      if (more) {
        uint64_t s_nx = sn, e_nx = en;                                   // (the first group's: loaded in front of the loop)
        if (have && !(RG_KO & 4)) { s_nx = park[8][lane]; e_nx = park[9][lane]; }
        else if (have) load_bounds(tile, s_nx, e_nx);
        nxt = make_tile(tile, s_nx, e_nx);
        if (tile2 < ntiles) request_bounds(tile2);                       // (for the hand-over after this one)
      }
    };
    bool done = false;
    if (fast) {
      XT xs[STASH];                                                    // the stash: the doubles of this lane's first 192 points (F32: 384 floats, x - x0)
      double x0 = 0.0;                                                 // (F32) the double of the container's first key
      // =========================== the fit ===========================
      // Rolled over the blocks of 16 steps (the code of a block is ~3 KB; unrolled over the whole walk it would be 60 KB, and two
      // CUs share 64 KB of instruction cache): a block leaves its 16 doubles in T[], and a switch on the block index copies
      // them to their places in the stash (static register names in every case: xs[] never becomes memory).
      double mx = 0.0, cc = 0.0, m2 = 0.0;
      double fmx = 0.0, fcc = 0.0, fm2 = 0.0;                            // ... as they stand behind the lane's last point
      // Duplicates.  8-byte integer keys: the minimum over the walk of lo(key_k) ^ lo(key_k-1) -- zero where two keys in a row share
      // their low word, a superset of "equal" that costs two 32-bit operations per step (false alarms: 2^-32 per pair);
      // f64 keys (low words of round numbers ARE equal): cleared by a comparison of the doubles.
      unsigned int dmin = 0xFFFFFFFFu, fdmin = 0xFFFFFFFFu;
      unsigned int plo = 0u;
      double xp = __builtin_nan("");
      const unsigned int npts = cur.npts;
      RAW rawA[8], rawB[8];                                              // the keys of the half block in work / of the next one
      double cA[12], cB[12];                                             // the constants of the quarter block in work / of the next one
      auto request = [&](double (&c)[12], unsigned int quarter_index) {
        const double* const tq = rtab4 + ((RG_DIAG & 1) ? 0u : quarter_index * 16u);
#pragma unroll
        for (int u = 0; u < 4; u++) { c[u] = tq[u]; c[4 + u] = DIVK ? tq[12 + u] : tq[4 + u]; c[8 + u] = tq[8 + u]; }
      };
      request(cA, 0u);
      // panel 1 has landed once at most the panels behind it are outstanding (requested at the hand-over: up to panel 3)
      if (cur.maxlen > 0u) {
        if (!(RG_KO & 4)) {
          if (cur.lastp >= 3u) rg_wait_vm<2 * NI>();
          else if (cur.lastp == 2u) rg_wait_vm<NI>();
          else rg_wait_vm<0>();
        }
        unsigned int in_b, dlt;
        block_base(0u, in_b, dlt);
#pragma unroll
        for (int q = 0; q < 8; q++) rawA[q] = slot_key(in_b, dlt, q);
        plo = ~rg_raw_lo(rawA[0]);                                         // (the walk's first step has nothing in front of it)
      }
      // One block of 16 steps.  STATIC: `b` is a constant of the call (the first RG_UBLK blocks of a group whose containers are
      // all longer than that: no lane ends there, so no test, no sums put aside, and every double goes straight to its register
      // of the stash); else rolled (any block of any group).
      // (the block index comes as a type in the STATIC case: every written-out block is then its own instantiation with ONE call,
      //  which the inliner takes in its normal order; as `always_inline` the rolled loop came out a third longer, with spills)
      // the first RG_UBLK blocks of a group whose containers are all longer than that, written out: no lane ends there, so no
      // test, no sums put aside, no search for the bank (every double goes straight to its register of the stash)
      unsigned int b0 = 0u;
      if (RG_UBLK > 0 && cur.ulong) {
#pragma unroll
        for (unsigned int b = 0; b < (unsigned int)RG_UBLK; b++) {
          constexpr bool STATIC = true;
#include "rmi_regs_block.inc.h"
        }
        b0 = (unsigned int)RG_UBLK;
      }
#pragma nounroll
      for (unsigned int b = b0; b * (unsigned int)RG_ROW < cur.maxlen; b++) {
        constexpr bool STATIC = false;
#include "rmi_regs_block.inc.h"
      }
      if (!LONG && cur.maxfar > (unsigned int)RG_MAXPTS) {
        // ---- the lanes with more than 240 points (3 in 10 000 leaves of the metric configuration) go on from the key array: the same
        //      steps, 8 keys a trip, every step tested; the other lanes' sums were put aside where their walks ended
        for (unsigned int k0 = (unsigned int)RG_MAXPTS; k0 < cur.maxfar; k0 += 8u) {
          K kk[8];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const unsigned int k = k0 + (unsigned int)q;
            kk[q] = keys[(uint64_t)cur.lo + (k < npts ? k : 0u)];
          }
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const unsigned int k = k0 + (unsigned int)q;
            const double* const th = rtab4 + ((k0 >> 2) + (unsigned int)(q >> 2)) * 16u + (unsigned int)(q & 3);
            const unsigned long long bits = key_to_bits_rg<K>(kk[q]);
            const double x = KeyTraits<K>::as_float(kk[q]);
            if (k < npts) {
              if constexpr (DIVK) { if (x == xp) dmin = 0u; }
              else { const unsigned int d = (unsigned int)bits ^ plo; dmin = dmin < d ? dmin : d; }
              const double dx = x - mx;
              if constexpr (DIVK) mx += dx / th[12]; else mx += div_by_count2(dx, th[0], th[4]);
              cc += dx * th[8];
              m2 += dx * (x - mx);
              xp = x; plo = (unsigned int)bits;
            }
          }
        }
        if (npts > (unsigned int)RG_MAXPTS) { fmx = mx; fcc = cc; fm2 = m2; fdmin = dmin; }
      }
      // LONG: the fit's panels have all landed (its last wait was vmcnt(0)) and been read: the ring is free, and the panels of the
      // error steps behind the stash are requested once more -- under the arithmetic that ends the fit
      // (a walk of at most 3 panels behind the stash: they are the ring's last panels, still there -- nothing is read again)
      const bool reread = LONG && cur.lastp > (unsigned int)(SBLK + RG_RING - 1);
      if (LONG && !(RG_KO & 4) && reread) {
#pragma unroll
        for (unsigned int p = (unsigned int)SBLK; p < (unsigned int)(SBLK + RG_RING); p++)
          if (p <= cur.lastp) issue_panel(cur.kb, p, cur.off, cur.lim);
      }
      mark(0);
      if (!(RG_KO & 4)) { k_hi = parked_key(0); k_lom1 = parked_key(2); k_next = parked_key(4); k_prev = parked_key(6); }
      if (!cur.valid || !((uint64_t)cur.e < sp.n)) k_next = KeyTraits<K>::max_value();
      if (!cur.valid || !(cur.s > 0u)) k_prev = KeyTraits<K>::zero_value();
      // a duplicate key somewhere in the group, or in front of a container's first point (compared as doubles: a superset
      // of key equality): the closed form of the y half does not hold -- the whole group goes through the general walk
      const bool dup0 = cur.act && (uint64_t)cur.lo > sp.rd_lo && KeyTraits<K>::as_float(k_lom1) == (F32 ? x0 : (double)xs[0]);   // (W2: the raw key's exact double)
      if (!__any(dup0 || (cur.act && fdmin == 0u))) {
        // ---- the container's last item once more (Q1, models/mod.rs:180), then linear.rs:36-58
        if (cur.act) {
          const double x = KeyTraits<K>::as_float(k_hi);
          const double nn = (double)npts;
          const double y0f = (double)cur.lo;
          const double my0 = y0f + (nn - 1.0) * 0.5, yprev = y0f + (nn - 1.0);   // the y half in closed form (no duplicate)
          const double nf = nn + 1.0;
          mx = fmx; cc = fcc; m2 = fm2;
          const double dx = x - mx;
          mx += dx / nf;
          const double my = my0 + (yprev - my0) / nf;
          cc += dx * (yprev - my);
          m2 += dx * (x - mx);
          const double cov = cc / (nf - 1.0), var = m2 / (nf - 1.0);
          if (!(var >= 0.0)) flags |= EF_NEG_VARIANCE;                     // linear.rs:48
          if (var == 0.0) { pa = my; pb = 0.0; }                           // linear.rs:50-53
          else { pb = cov / var; pa = my - pb * mx; }                      // no fma: linear.rs:56
        } else if (cur.ck == 1) { pa = (double)cur.lo; pb = 0.0; }         // Q4: one borrowed point (two identical items)
      if (!LONG && !(RG_KO & 2) && cur.maxfar > (unsigned int)RG_MAXPTS) {
        // (the steps of the long containers behind the ring's reach: from the key array once more, 16 keys a lane and trip)
        for (unsigned int k0 = (unsigned int)RG_MAXPTS; __any(k0 < eend); k0 += 16u) {
          K kk[16];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const unsigned int k = k0 + (unsigned int)q;
            kk[q] = keys[(uint64_t)cur.lo + (k < eend ? k : 0u)];
          }
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const unsigned int k = k0 + (unsigned int)q;
            if (k < eend) err_step(KeyTraits<K>::as_float(kk[q]), k);
          }
        }
      }
      // ---- LONG: the error steps behind the stash, through the ring once more (or, for a walk of at most 3 panels behind the stash, from
      //      the ring as the fit left it).  Half blocks of 8 steps, the next half's keys requested from LDS before this half's steps run
      //      (a lone wave has nobody to fill an LDS round trip); half (b, 1)'s reads landed, panel b + 4 takes the slot of panel b;
      //      panel b + 2 has landed once at most the panels requested behind it are outstanding -- the waits of the fit.
      auto ring_walk = [&]() {
        if (!(LONG && !(RG_KO & 2) && cur.maxlen > (unsigned int)STASH)) return;
        const unsigned int hlast = (cur.maxlen - 1u) >> 3;               // last half block of the walk
        RAW rkA[8], rkB[8];
        auto read_half = [&](unsigned int h, RAW (&rk)[8]) {
          unsigned int in_b, dlt;
          block_base(h >> 1, in_b, dlt);
          if (h & 1u) {
#pragma unroll
            for (int q = 0; q < 8; q++) rk[q] = slot_key(in_b, dlt, 8 + q);
          } else {
#pragma unroll
            for (int q = 0; q < 8; q++) rk[q] = slot_key(in_b, dlt, q);
          }
        };
        auto steps = [&](unsigned int h, const RAW (&rk)[8]) {
          const unsigned int k0 = h * 8u;
          if (__all(eend >= k0 + 8u || eend <= k0)) {                      // no leaf ends inside the half: one test for its steps
            if (eend > k0) {
#pragma unroll
              for (int q = 0; q < 8; q += 2) err_pair(rg_as_float<K>(rk[q]), k0 + (unsigned int)q, rg_as_float<K>(rk[q + 1]), k0 + (unsigned int)(q + 1));
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; q++) { if (k0 + (unsigned int)q < eend) err_step(rg_as_float<K>(rk[q]), k0 + (unsigned int)q); }
          }
        };
        auto landed = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); };   // lgkmcnt(0)
        auto wait_panel = [&](unsigned int b) {                            // panel b + 2 (for block b + 1)
          if (RG_KO & 4) return;
          if (cur.lastp >= b + 4u) rg_wait_vm<2 * NI>();
          else if (cur.lastp == b + 3u) rg_wait_vm<NI>();
          else rg_wait_vm<0>();
        };
        if (reread) wait_panel((unsigned int)SBLK - 1u);
        if (!RG_WALK_PIPE) {                                               // (the plain form: a half block's reads, the wait, its steps)
#pragma nounroll
          for (unsigned int h = 2u * (unsigned int)SBLK; h <= hlast; h++) {
            const unsigned int b = h >> 1;
            read_half(h, rkA);
            landed();
            if ((h & 1u) && reread && !(RG_KO & 4) && b + (unsigned int)RG_RING <= cur.lastp) issue_panel(cur.kb, b + (unsigned int)RG_RING, cur.off, cur.lim);
            steps(h, rkA);
            if ((h & 1u) && h < hlast && reread) wait_panel(b);
          }
          return;
        }
        read_half(2u * (unsigned int)SBLK, rkA);
#pragma nounroll
        for (unsigned int h = 2u * (unsigned int)SBLK; h <= hlast; h += 2u) {
          const unsigned int b = h >> 1;
          landed();
          if (h + 1u <= hlast) read_half(h + 1u, rkB);                     // (the same two panels)
          steps(h, rkA);
          if (h + 1u <= hlast) {
            landed();                                                      // (every key of block b is in registers: panel b's slot is free)
            if (reread && !(RG_KO & 4) && b + (unsigned int)RG_RING <= cur.lastp) issue_panel(cur.kb, b + (unsigned int)RG_RING, cur.off, cur.lim);
            if (h + 2u <= hlast) {
              if (reread) wait_panel(b);
              read_half(h + 2u, rkA);
            }
            steps(h + 1u, rkB);
          }
        }
      };
      if (!LONG && !(RG_KO & 2)) {
        // the steps behind the stash: still in the ring (masked the plain way: a few blocks)
#pragma nounroll
        for (unsigned int b = (unsigned int)RG_SBLK; b * (unsigned int)RG_ROW < cur.maxlen; b++) {
          if (!__any(eend > b * (unsigned int)RG_ROW)) break;              // (a container's last point is not a key of its leaf: often nothing is left here)
          unsigned int in_b, dlt;
          block_base(b, in_b, dlt);
          // (the 8 reads of a half block together, then its steps: read one by one as they are used, every step would wait
          //  out an LDS round trip -- this wave has no other to fill it)
#pragma unroll
          for (int hb = 0; hb < 2; hb++) {
            if (hb == 1 && !(b * (unsigned int)RG_ROW + 8u < cur.maxlen)) break;
            RAW rk[8];
#pragma unroll
            for (int q = 0; q < 8; q++) rk[q] = slot_key(in_b, dlt, hb * 8 + q);
            __builtin_amdgcn_s_waitcnt(0xC07F);                            // lgkmcnt(0)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const unsigned int k = b * (unsigned int)RG_ROW + (unsigned int)(hb * 8 + q);
              if (k < eend) err_step(rg_as_float<K>(rk[q]), k);
            }
            if (b * (unsigned int)RG_ROW + (unsigned int)(hb * 8 + 4) < cur.maxlen) {
#pragma unroll
              for (int q = 4; q < 8; q++) {
                const unsigned int k = b * (unsigned int)RG_ROW + (unsigned int)(hb * 8 + q);
                if (k < eend) err_step(rg_as_float<K>(rk[q]), k);
              }
            }
          }
        }
      }
      mark(1);
      asm volatile("" : "+v"(k_next), "+v"(k_prev));                   // (read from their parking rows before the next group's keys are sent there)
      const unsigned int sl = cur.maxlen < (unsigned int)STASH ? cur.maxlen : (unsigned int)STASH;
      // Every load of the compiler's own is waited for in front of the hand-over, by the builtin its score keeping understands: behind the
      // hand-over the next group's panels are in flight, which it does not count -- a wait it places there for a load of the far lanes' (a
      // register about to be overwritten) comes out as vmcnt(0) and stands until the panels have landed, at the head of the error pass that
      // was to run under them.  (stores count as well: the coefficients are stored behind this wait, not in front of it)
      auto step0 = [&]() {
        if ((int)es <= 0 && eend > 0u) err_step(F32 ? x0 : (double)xs[0], 0u);   // (step 0 belongs to the leaf only where the container starts with it or behind it)
        if (es == 0xFFFFFFFFu && cur.act) err_step(KeyTraits<K>::as_float(k_lom1), 0xFFFFFFFFu);
      };
      // (LONG, tried: the replay in two phases -- four banks while the error panels requested behind the fit are on their way, then the walk and
      //  the hand-over, then the other banks: the loop around the banks' code makes the compiler keep copies of the stash values it reads, 0.298
      //  against 0.279 ms for C4's shard shape)
      // LONG: the banks 1 .. RG_PRE of the stash are replayed HERE, in front of the walk through the ring, while its first panels (requested behind
      // the fit) are on their way -- written out once, for groups in which every leaf covers them whole or not at all; the loop behind the
      // hand-over then goes on behind them.
      if constexpr (F32) sA = __builtin_fma(pb, x0, pa);
      unsigned int b_after0 = 1u;
      if constexpr (LONG != 0 && RG_PRE > 0) {
        constexpr unsigned int PE = (unsigned int)((RG_PRE + 1) * RG_ROW);
        if (!(RG_KO & 2) && reread && sl >= PE && __all(eend >= PE || eend <= (unsigned int)RG_ROW)) {
          if (eend >= PE) {
            rg_static_for<1, RG_PRE + 1>([&](auto i_tag) {
              constexpr int i = decltype(i_tag)::value;
#pragma unroll
              for (int qq = 0; qq < RG_ROW; qq += 2) {
                if constexpr (F32) err_pair_f(xs[i * RG_ROW + qq], (unsigned int)(i * RG_ROW + qq), xs[i * RG_ROW + qq + 1], (unsigned int)(i * RG_ROW + qq + 1));
                else if constexpr (W2) err_pair_w(xs[i * RG_ROW + qq], (unsigned int)(i * RG_ROW + qq), xs[i * RG_ROW + qq + 1], (unsigned int)(i * RG_ROW + qq + 1));
                else err_pair(xs[i * RG_ROW + qq], (unsigned int)(i * RG_ROW + qq), xs[i * RG_ROW + qq + 1], (unsigned int)(i * RG_ROW + qq + 1));
              }
              asm volatile("; stash bank %0 (ahead)" ::"n"(i));
            });
          }
          b_after0 = (unsigned int)RG_PRE + 1u;
        }
      }
      if constexpr (LONG != 0) ring_walk();
      __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0)
      if (cur.valid) { params[2 * j] = pa; params[2 * j + 1] = pb; }
      hand_over();
      mark(2);
      if (!(RG_KO & 2)) {
        step0();
        const unsigned int b_from = 0u, b_to = (sl + (unsigned int)RG_ROW - 1u) >> 4;
#include "rmi_regs_replay.inc.h"
      }
      mark(3);
      // ---- what k_regs_finalize needs to finish the leaf: the raw maximum and the keys on either side of the leaf (they lie in this
      //      wave's LDS; there they would be two scattered loads per leaf).  The rest of the leaf's end -- widening, row, counts, the
      //      group's aggregate record: ~400 instructions a lane, a logarithm and a division among them -- costs a lone wave 6 us per
      //      group here and a fully occupied launch ~15 us for ALL groups.
      unsigned long long flagged = 0ull;
      if constexpr (F32) {
        // The certificate.  x~ - x0 = d with |d - (x - x0)| <= 2^-23.9 (x - x0) (the subtraction's rounding, then the float's), so
        // |fma(beta, d, A) - RN(beta x + alpha)| <= E = 2^-23.5 |beta| D + 2^-51 (|A| + |beta| D), D = x_hi - x0 >= x - x0 for every key of
        // the leaf (the container's last key; the second term: A's rounding and the two results').  sg_cvt_u32 and the clamp are monotone:
        // if every prediction's fraction lies in [E, 1 - E] -- tmax <= 1/2 - E -- the integers are the reference's.  Anything not finite
        // fails the comparison and flags the leaf.
        const double D = KeyTraits<K>::as_float(k_hi) - x0;
        const double bD = __builtin_fabs(pb) * D;
        const double E = 0x1p-23 * 0.7072 * bD + 0x1p-51 * (__builtin_fabs(sA) + bD);
        if (cur.act && eend > 0u && !(tmax <= 0.5 - E)) flagged = 1ull << 63;
      }
      if (cur.valid) {
        const uint64_t jl = (uint64_t)done_tile * 64 + (uint64_t)lane_j;
        leaf_maxerr[j] = (unsigned long long)emax | flagged;
        bnext[jl] = k_next; bprev[jl] = k_prev;
      }
      if (lane == 0) tile_slow[done_tile] = 0;
      if (flags) atomicOr(&st->err_flags, flags);
      mark(4);
      done = true;
      }
    }
    if (!done) {
      // not taken here: the group goes on the list of k_leaf_lanes_listed, launched behind this kernel (the general walk inside
      // this kernel would share its registers with the stash: the compiler then keeps a third of the stash in scratch memory)
      if (have && lane == 0) { slow_list[atomicAdd(slow_count, 1u)] = done_tile; tile_slow[done_tile] = 1; }
      rg_wait_vm<0>();                                               // (hand_over reads the parked bounds of the next group: DMA loads the compiler does not see)
      hand_over();
    }
    if (!more) break;
    cur = nxt;
    have = true;
    done_tile = tile;
    tile = tile2;
    tile2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)tile3);
  }
  if (RG_PROF && prof != nullptr && lane == 0) {
#pragma unroll
    for (int i = 0; i < 12; i++) atomicAdd(prof + i, pf[i]);
  }
}

// The end of the leaves k_leaf_regs has fitted and measured (two_layer.rs:185-197, 226-259, the row of codegen.rs:288-315, the terms
// of :267-287): one thread per leaf, a wave = one group of 64 leaves = one aggregate record, as in k_leaf_lanes.  Groups that went
// through k_leaf_lanes_listed were finished there.
// VROOT == K_CUBIC: the root's targets are not monotone by arithmetic (three nested fmas round independently) and k_leaf_regs has no
// root evaluation in its error pass.  The search rests on the assumption that they are non-decreasing (two_layer.rs:50); it is PROVEN
// here in O(L) instead of per key: the host has checked that the exact cubic is increasing over the resident keys' range (its
// derivative's minimum there, rmi_hip.hip), the computed value c(x) differs from the exact one by at most E(x) = 2^-53 (|v3| + |x| |v2| +
// x^2 |v1|) (one rounding per fma of the Horner form, cubic_spline.rs:146-148), so if the first key of leaf j computes to >= j + 2 E and
// its last key to < j + 1 - 2 E -- E an upper bound of E(x) over the WHOLE leaf, taken from the coefficients' magnitudes at the larger |x| of
// the two ends --, every key between them has an exact value in [j + E, j + 1 - E] and therefore the target j.  The margin used is 4 E.  A leaf that does not clear it (an end key within ~1e-9 of a leaf border: about one leaf in a training of 2^20) is verified
// key by key on the spot, by the 64 lanes of its wave together: the check k_leaf_lanes<.., K_CUBIC> makes for every key.
__device__ __forceinline__ uint64_t rg_readlane_u64(uint64_t v, int src) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src), hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
template <typename K, int VROOT = -1>
__global__ void __launch_bounds__(256) k_regs_finalize(const K* __restrict__ keys, Span sp, uint64_t L,
                                                       const unsigned long long* __restrict__ leaf_start, DevState* st, const unsigned int* __restrict__ slow_count,
                                                       double* __restrict__ params, const unsigned long long* __restrict__ leaf_maxerr,
                                                       const K* __restrict__ bnext, const K* __restrict__ bprev,
                                                       const unsigned char* __restrict__ tile_slow, unsigned int ntiles,
                                                       unsigned long long* __restrict__ leaf_err, unsigned long long* __restrict__ leaf_count,
                                                       unsigned char* __restrict__ rows, StatsPartial* __restrict__ partials, PeerRows peers, RootP vr,
                                                       double margin_scale,     // (1; a test widens the margin until every leaf falls back)
                                                       bool listed_behind) {    // the listed groups' kernel runs BEHIND this one (and behind the first k_lane_reduce)
  const uint64_t jl = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned int tile = (unsigned int)(jl >> 6);
  if (jl == 0) st->regs_listed = *slow_count;                          // (into the record the host reads: rmi_hip.hip, regs_off)
  if (tile >= ntiles) return;                                          // (wave-uniform)
  const uint64_t j = sp.leaf_lo + jl;
  if constexpr (VROOT == K_CUBIC) {                                    // every leaf, whoever fitted it
    bool ok = true;
    uint64_t vs = 0, ve = 0;
    if (j < sp.leaf_hi) {
      const uint64_t s = leaf_start[j], e = leaf_start[j + 1];
      vs = s; ve = e;
      if (s < e) {
        auto val = [&](K k) -> double {
          const double x = KeyTraits<K>::as_float(k);
          const double v1 = __builtin_fma(vr.p0, x, vr.p1), v2 = __builtin_fma(v1, x, vr.p2), v3 = __builtin_fma(v2, x, vr.p3);
          return v3;
        };
        // the leaf's first key is the key behind leaf j - 1 (bnext[j - 1]), its last key the key in front of leaf j + 1 (bprev[j + 1]): k_leaf_regs
        // has handed both over for the groups it took -- coalesced reads instead of two scattered lines per leaf
        const uint64_t nl = sp.leaf_hi - sp.leaf_lo;
        const K ka = (jl > 0 && tile_slow[(jl - 1) >> 6] == 0) ? bnext[jl - 1] : keys[s];
        const K kb = (jl + 1 < nl && tile_slow[(jl + 1) >> 6] == 0) ? bprev[jl + 1] : keys[e - 1];
        // The margin must hold for EVERY key of the leaf, not only for its two end keys: E(x) taken at the ends can be smaller than at an
        // interior key when the cubic's terms cancel there.  So E is bounded over the whole leaf from the magnitudes alone -- with
        // xm = max(|x|) over the leaf, |v1| <= |p0| xm + |p1| = V1, |v2| <= V1 xm + |p2| = V2, |v3| <= V2 xm + |p3| = V3: monotone in |x|, so
        // the bound at xm covers every key between the ends.  (Weaker than the pointwise E where the terms cancel; a leaf that fails it is
        // verified key by key below, which costs a wave one pass over that leaf.)
        const double ca = val(ka), cb = val(kb);
        const double xm = fmax(fabs(KeyTraits<K>::as_float(ka)), fabs(KeyTraits<K>::as_float(kb)));
        const double V1 = fabs(vr.p0) * xm + fabs(vr.p1), V2 = V1 * xm + fabs(vr.p2), V3 = V2 * xm + fabs(vr.p3);
        const double E4 = margin_scale * 0x1p-51 * (V3 + xm * V2 + (xm * xm) * V1);
        const double Lf = (double)vr.L;
        if (j > 0) ok = ok && (ca >= (double)j + E4);
        if (j + 1 < L) ok = ok && (cb < (double)(j + 1) - E4);
        else ok = ok && (cb < Lf - E4);                                 // (the largest prediction of all: below L, two_layer.rs:45-48)
      }                                                                 // (NaN and infinities: undecided as well)
    }
    unsigned long long um = __ballot(!ok);
    unsigned int vflags = 0;
    const unsigned int Lm1 = (unsigned int)(vr.L - 1);
    while (um) {                                                        // an undecided leaf: its keys, 64 at a time
      const int src = __builtin_ctzll(um);
      um &= um - 1ull;
      const uint64_t us = rg_readlane_u64(vs, src), ue = rg_readlane_u64(ve, src);
      const unsigned int uj = (unsigned int)(sp.leaf_lo + (jl & ~63ull) + (uint64_t)src);
      for (uint64_t i = us + (threadIdx.x & 63); i < ue; i += 64) {
        const unsigned int rt = sg_cvt_u32(root_eval_f<K_CUBIC>(vr, KeyTraits<K>::as_float(keys[i])));
        if (rt > Lm1) vflags |= EF_ROOT_OOB;
        if ((rt < Lm1 ? rt : Lm1) != uj) vflags |= EF_NON_MONOTONE;
      }
    }
    if (vflags) atomicOr(&st->err_flags, vflags);
  }
  if (tile_slow[tile] != 0) {                                          // (wave-uniform)
    // a listed group's aggregate record is written by k_leaf_lanes_listed; where that runs only behind the host's first synchronisation
    // the first k_lane_reduce reads the record before: a neutral one, so that what it publishes meanwhile is well defined
    if (listed_behind && (threadIdx.x & 63) == 0) partials[tile] = StatsPartial{0ull, 0ull, 0ull, 0.0, 0.0};
    return;
  }
  unsigned long long st_mx = 0, st_mi = 0, st_sum = 0;
  double st_l2 = 0.0, st_lg = 0.0;
  // A leaf k_leaf_regs<K, LONG> could not certify (bit 63 of its maximum: a prediction from the float stash within the rounding bound of an
  // integer): its error pass once more, from the key array, on the keys themselves -- the 64 lanes of the wave together, 64 keys a trip
  // (two_layer.rs:207-217; no duplicate among them: the group would have been listed).
  unsigned long long me = 0ull;
  uint64_t fs = 0, fe = 0;
  double fa = 0.0, fb = 0.0;
  if (j < sp.leaf_hi) { me = leaf_maxerr[j]; fs = leaf_start[j]; fe = leaf_start[j + 1]; fa = params[2 * j]; fb = params[2 * j + 1]; }
  {
    unsigned long long um = __ballot((me >> 63) != 0ull);
    const unsigned int n32 = (unsigned int)sp.n;
    const int ln = (int)(threadIdx.x & 63);
    while (um) {
      const int src = __builtin_ctzll(um);
      um &= um - 1ull;
      const uint64_t us = rg_readlane_u64(fs, src), ue = rg_readlane_u64(fe, src);
      const double ua = __longlong_as_double((long long)rg_readlane_u64((uint64_t)__double_as_longlong(fa), src));
      const double ub = __longlong_as_double((long long)rg_readlane_u64((uint64_t)__double_as_longlong(fb), src));
      unsigned int em = 0u;
      for (uint64_t i = us + (uint64_t)ln; i < ue; i += 64) {
        const double f = __builtin_fma(ub, KeyTraits<K>::as_float(keys[i]), ua);      // linear.rs:87-90
        em = max(em, sg_absdiff(min(sg_cvt_u32(f), n32), (unsigned int)i));            // models/mod.rs:735-737, two_layer.rs:14-18
      }
      em = rg_wave_max(em);
      if (ln == src) me = (unsigned long long)em;
    }
  }
  if (j < sp.leaf_hi) {
    const uint64_t s = fs, e = fe;
    double pp[2] = {fa, fb};
    const K k_next = e < sp.n ? bnext[jl] : KeyTraits<K>::max_value();
    const K k_prev = s > 0 ? bprev[jl] : KeyTraits<K>::zero_value();
    uint64_t final_err, cnt_j;
    finalize_one_pre<K_LINEAR, K>(j, s, e, sp, L, keys, pp, me, 0ull, st->last_target, k_next, k_prev, final_err, cnt_j);
    if (!(s < e)) { params[2 * j] = pp[0]; params[2 * j + 1] = pp[1]; }
    leaf_err[j] = final_err;
    leaf_count[j] = cnt_j;
    double* rp = reinterpret_cast<double*>(rows + j * 24);
    rp[0] = pp[0]; rp[1] = pp[1];
    *reinterpret_cast<unsigned long long*>(rows + j * 24 + 16) = final_err;
    for (int p = 0; p < peers.n; p++) {                                // (the direct exchange of a sharded training: rmi_lanes.hip.h)
      double* pr = reinterpret_cast<double*>(peers.tab[p] + j * 24);
      pr[0] = pp[0]; pr[1] = pp[1];
      *reinterpret_cast<unsigned long long*>(peers.tab[p] + j * 24 + 16) = final_err;
    }
    st_mx = final_err; st_mi = j;
    st_sum = cnt_j * final_err;                                        // wrapping u64, like the reference's sum
    const double v = (double)st_sum;
    st_l2 = (v * v) / (double)sp.n;
    st_lg = (double)cnt_j * log2((double)(2 * final_err + 2));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {                                   // (lexicographic maximum and sums: the order of k_leaf_lanes)
    const unsigned long long omx = shfl_down_u64(st_mx, d), omi = shfl_down_u64(st_mi, d);
    if (omx > st_mx || (omx == st_mx && omi > st_mi)) { st_mx = omx; st_mi = omi; }
    st_sum += shfl_down_u64(st_sum, d);
    st_l2 += __shfl_down(st_l2, d);
    st_lg += __shfl_down(st_lg, d);
  }
  if ((threadIdx.x & 63) == 0) partials[tile] = StatsPartial{st_mx, st_mi, st_sum, st_l2, st_lg};
}

// The groups k_leaf_regs did not take: the body of k_leaf_lanes per listed group (a fixed grid; nothing listed: the blocks leave).
template <typename K>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RMI_LN_WPE, RMI_LN_WPE))) k_leaf_lanes_listed(const unsigned int* __restrict__ slow_list,
                                                   const unsigned int* __restrict__ slow_count, const K* __restrict__ keys, Span sp,
                                                   const unsigned long long* __restrict__ leaf_start,
                                                   DevState* __restrict__ st, double* __restrict__ params,
                                                   const double* __restrict__ rtab, SgList fl, unsigned int long_min,
                                                   unsigned long long* __restrict__ leaf_maxerr,
                                                   unsigned long long* __restrict__ leaf_run, uint64_t L,
                                                   unsigned long long* __restrict__ leaf_err,
                                                   unsigned long long* __restrict__ leaf_count,
                                                   unsigned char* __restrict__ rows, StatsPartial* __restrict__ partials, RootP vr,
                                                   PeerRows peers) {
  __shared__ typename LnBits<K>::type panel[64 * LnGeom<K>::STRIDE];
  const unsigned int n = *slow_count;
  for (unsigned int i = blockIdx.x; i < n; i += gridDim.x)
    leaf_lanes_body<K, true, K_LINEAR, -1>(slow_list[i], panel, keys, sp, leaf_start, st, params, rtab, fl, long_min, leaf_maxerr, leaf_run, L, leaf_err, leaf_count,
                                           rows, partials, vr, peers);
}

}  // namespace rmi
