// txm_i8g.h -- the int8 bootstrap path with the per-sample counts in HBM (round 5): a generator kernel writes the u8
// counts of a replicate slab in the order the contraction kernel's MFMA A operands take them, the contraction kernel
// (txm_resample_i8g.hip) has no sampler inside -- no Philox, no count tile, no fill barriers.
#pragma once
#include "txm_i8t_common.h"

namespace txm {

constexpr int G_REPS = 128;                 // replicates per workgroup of the contraction (4 MFMA row blocks of 32)
constexpr int G_KSTEP_BYTES = 32 * G_REPS;  // 4096: the counts of one k-step (32 samples) of a replicate group
constexpr int G_TILE_BYTES = SM_T * G_REPS; // 131072: one sampler tile of a replicate group

// Count table of a slab of replicate groups (u8):
//   table[((g * ntiles + t) * 32 + s) * 4096 + q * 1024 + L * 16 + b]
//     = draws of sample tile_base(t) + 32 s + 16 (L >> 5) + b in replicate 128 g + 32 q + (L & 31) of the slab
// (tile_base(t) = min(1024 t, N - 1024): the last tile slides its window back, foreign samples count 0) -- lane L of
// a wave reads the A operand of replicate quarter q and k-step s as ONE 16-byte word, a k-step is 4 KiB contiguous,
// a tile 128 KiB, a replicate group's table one contiguous stream.  Replicates past nrep count 0.
static inline size_t count_table_bytes(int64_t ntiles, int64_t nreps) {
  return (size_t)cdiv(nreps, G_REPS) * (size_t)ntiles * G_TILE_BYTES;
}

// ---- pieces shared by the table-fed contraction kernels (txm_resample_i8g.hip: wide states; txm_resample_i8gn.hip: narrow states)
constexpr int G_BS = 4;  // k-steps per block (one barrier per block)
constexpr int G_RAW = 2 * G_BS * 256;  // one raw buffer: u then w of a block's chunks
#ifndef TXM_G_LEAD
#define TXM_G_LEAD 2
#endif
constexpr uint32_t G_LEAD = TXM_G_LEAD;  // tiles a replicate group may run ahead of the slowest one of its window

// LDS-DMA: 16 (4) bytes per lane from saddr + voff to the LDS address in M0 + 16 (4) * lane
__device__ __forceinline__ void g_dma16(const void *sbase, uint32_t voff, uint32_t lds_dst) {
#ifdef TXM_G_NO_DMA  // ablation build
  return;
#endif
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_dst) : "memory", "m0");
}
// ... the same with the non-temporal hint: the count words of the table, which ONE workgroup reads once per pass (x, which the
// replicate groups of a window share through the L2, keeps the plain form).  Same box, north-star call, three rounds: 159.9 / 160.8 /
// 161.0 ms plain against 159.4 / 159.0 / 160.2 ms (profiles/r06_table_nt_ab.txt); -DTXM_G_TABLE_PLAIN is the A/B build.
__device__ __forceinline__ void g_dma16_stream(const void *sbase, uint32_t voff, uint32_t lds_dst) {
#ifdef TXM_G_NO_DMA
  return;
#endif
#ifdef TXM_G_TABLE_PLAIN  // (A/B build)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_dst) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_dst) : "memory", "m0");
#endif
}
__device__ __forceinline__ void g_dma4(const void *sbase, uint32_t voff, uint32_t lds_dst) {
#ifdef TXM_G_NO_DMA
  return;
#endif
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_dst) : "memory", "m0");
}
// (TXM_G_ADIR experiment) 16 bytes per lane from saddr + voff into registers, as an asm the compiler does not track: it cannot count
// the DMA pieces above either, and for a load it DOES see across the loop's back edge it falls back to s_waitcnt vmcnt(0) at the top of
// every block -- the waits are written by hand (g_wait_vm) from the issue order of the block
__device__ __forceinline__ void g_load16(v4i &dst, const void *sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// sixteen progress words of a window, read past the scalar cache (glc): SMEM counts on lgkmcnt, so a poll does not touch
// the wave's vmcnt queue of DMAs
typedef uint32_t g_v16u __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t g_min_progress(const uint32_t *pg) {
  g_v16u v;
  asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(pg) : "memory");
  uint32_t m = 0xffffffffu;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t e = v[i] == 0u ? 0xffffffffu : v[i];  // 0: a group that has not started (or does not exist)
    m = e < m ? e : m;
  }
  return m;
}
// the lane id, recomputed where it is used: addresses that are a function of the lane and are needed once a block (the DMA
// offsets, the staging addresses) would otherwise be held in registers across the k-steps -- the kernel has none to spare,
// they were spilled, and every reload put a scratch round trip + s_waitcnt vmcnt on the block's critical path
__device__ __forceinline__ uint32_t g_lane_now() {
  uint32_t l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int N>
__device__ __forceinline__ void g_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// counts: [nrep][ntiles] rows of the CALL (row r = stream replicate rep_base + r); the slab holds the call's
// replicates rep_begin .. rep_begin + 128 n_groups - 1
int launch_count_table(const uint32_t *counts, int64_t nrep, int64_t N, uint32_t k0, uint32_t k1, uint32_t rep_base,
                       int64_t rep_begin, int64_t n_groups, unsigned char *table, hipStream_t st);

// the contraction over a count table (txm_resample_i8g.hip): one 32-column group (a.col0, a.C), every pass of the order
bool i8g_applicable(const double *x, int64_t ldx_s, int64_t C, const double *y, int64_t ldy_s);  // C = all columns of the call
int launch_resample_i8g(const I8Args &a, int K, bool weighted, const unsigned char *table, int64_t rep_begin, int n_grp,
                        hipStream_t st);

// ... and for narrow states (C <= 16; txm_resample_i8gn.hip): every pass of the order over the tables of the S states of a batched
// launch (table_state_stride bytes apart; 0 for a single state)
bool i8gn_applicable(const double *x, int64_t ldx_s, int64_t C, int K);
int launch_resample_i8gn(const I8Args &a, int K, bool weighted, const unsigned char *table, int64_t rep_begin, int n_grp,
                         size_t table_state_stride, hipStream_t st);

}  // namespace txm
