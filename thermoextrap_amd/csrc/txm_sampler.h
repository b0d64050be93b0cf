// txm_sampler.h -- device side of the counter-based multinomial sampler.
// Normative statement of the stream: oracle/philox_oracle.c (bit-for-bit).
#pragma once
#include "txm_common.h"

namespace txm {

constexpr int SM_LT = 10;
constexpr int SM_T = 1 << SM_LT;  // samples per tile

struct SamplerGeom {
  int64_t ndat, ntiles, last_tile_size;
  int k;  // 2^k >= ntiles: depth of the tile tree
};

static inline int sampler_geometry(int64_t ndat, SamplerGeom *g) {
  if (ndat > ((int64_t)1 << 30)) return -1;
  g->ndat = ndat;
  g->ntiles = (ndat + SM_T - 1) / SM_T;
  g->last_tile_size = ndat - (g->ntiles - 1) * SM_T;
  g->k = 0;
  while (((int64_t)1 << g->k) < g->ntiles) g->k++;
  return 0;
}

struct Philox4 {
  uint32_t w[4];
};

// XOR3: fold the two XORs of a round into one v_bitop3_b32.  Measured on gfx950: the standalone sampler
// kernels gain 9 % (round-1 tile-count kernels: 58.8 -> 53.9 ms at N=1e8, nrep=1000), the fill phases fused into the
// bootstrap kernels lose (int8 kernel 262 -> 271 ms), so only the former ask for it.
template <bool XOR3 = false>
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 64-bit product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: on gfx950
    // every VALU instruction costs FP64-MFMA pipe time, so instruction count matters
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0, n2;
    if constexpr (XOR3) {
      n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);  // a ^ b ^ c
      n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    } else {
      n0 = hi1 ^ c1 ^ k0;
      n2 = hi0 ^ c3 ^ k1;
    }
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += W0;
    k1 += W1;
  }
  Philox4 o;
  o.w[0] = c0;
  o.w[1] = c1;
  o.w[2] = c2;
  o.w[3] = c3;
  return o;
}

__device__ __forceinline__ uint32_t slot16(const Philox4 &o, int e) {
  return (o.w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
}

// Per-sample counts ("stage 3") for one (replicate r, tile t) executed by one wave: every draw calls
// hit(off) with off in [0, tile_size).  `n` = counts[r][t].
//   full tile   : draw d = field d % 12 of Philox call d / 12 (three 10-bit
//                 fields per word), call c handled by lane c % 64 -- no random
//                 bit is wasted and a tile costs ceil(n / 768) wave iterations.
//   partial tile: per-lane quotas with 16-bit slots and rejection.
template <class Hit>
__device__ __forceinline__ void sampler_fine_tile(uint32_t k0, uint32_t k1, uint32_t r, uint32_t t,
                                                  uint32_t n, uint32_t tile_size, int lane, Hit hit) {
  if (tile_size == (uint32_t)SM_T) {
    for (uint32_t c0 = 0; c0 * 12u < n; c0 += 64u) {  // wave-uniform trip count
      const uint32_t c = c0 + (uint32_t)lane;
      const uint32_t first = c * 12u;
      if (first < n) {
        const Philox4 o = philox4x32_10(t, c, r, 3u, k0, k1);
        const uint32_t nd = n - first;  // >= 1; fields beyond nd are unused
#pragma unroll
        for (int q = 0; q < 12; ++q)
          if ((uint32_t)q < nd) hit((o.w[q / 3] >> (10 * (q % 3))) & 1023u);
      }
    }
  } else {
    uint32_t quota = n / 64u + ((uint32_t)lane < (n % 64u) ? 1u : 0u);
    uint32_t j = 0;
    const uint32_t c1 = t * 64u + (uint32_t)lane;
    while (quota) {
      const Philox4 o = philox4x32_10(j++, c1, r, 6u, k0, k1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t off = slot16(o, e) & (uint32_t)(SM_T - 1);
        if (quota && off < tile_size) {
          hit(off);
          --quota;
        }
      }
    }
  }
}

}  // namespace txm
