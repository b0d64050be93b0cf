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

// counts: [nrep][ntiles] rows of the CALL (row r = stream replicate rep_base + r); the slab holds the call's
// replicates rep_begin .. rep_begin + 128 n_groups - 1
int launch_count_table(const uint32_t *counts, int64_t nrep, int64_t N, uint32_t k0, uint32_t k1, uint32_t rep_base,
                       int64_t rep_begin, int64_t n_groups, unsigned char *table, hipStream_t st);

// the contraction over a count table (txm_resample_i8g.hip): one 32-column group (a.col0, a.C), every pass of the order
bool i8g_applicable(const double *x, int64_t ldx_s, int64_t C, const double *y, int64_t ldy_s);  // C = all columns of the call
int launch_resample_i8g(const I8Args &a, int K, bool weighted, const unsigned char *table, int64_t rep_begin, int n_grp,
                        hipStream_t st);

}  // namespace txm
