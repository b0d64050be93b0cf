/*
 * oracle/philox_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement, bit for bit, of libtxmom's device multinomial sampler
 * ("scale mode", include/txmom.h: txm_sampler_*).  The reference's sampler is
 *     indices = rng.choice(ndat, (nrep, nsamp), replace=True); freq = bincount
 * (cmomy 0.24.0 factory_sampler as reached from src/thermoextrap/data.py:
 * 1782-1789; semantics verified in SURVEY App. B).  Its PCG64 stream is
 * sequential and cannot be reproduced at N = 1e8, so the device generates the
 * same DISTRIBUTION (exact multinomial(nsamp; 1/ndat ...)) from a counter-based
 * Philox4x32-10 stream using integer arithmetic only.  This file is the
 * normative statement of that stream; tests compare the GPU's tables with it
 * bit for bit.
 *
 * Spec (all integers):
 *   T = 1024 samples per tile;  ntiles = ceil(ndat / T)
 *   s = smallest shift with ceil(ntiles / 2^s) <= 16384   (requires s <= 6)
 *   nb1 = ceil(ntiles / 2^s) coarse bins of BS = T * 2^s samples
 *   Stage 1 (per replicate r): 16384 virtual lanes v, quota q_v = nsamp/16384
 *     (+1 for v < nsamp % 16384).  Lane v reads Philox(key=seed,
 *     ctr=(j, v, r, 1)), j = 0,1,...; each call yields 8 16-bit slots (word e>>1,
 *     half e&1).  Slot value z: c = z & (2^k1 - 1), k1 = bits(nb1 - 1).
 *     Reject if c >= nb1.  If c is the last bin and that bin is partial, draw
 *     off = Philox(ctr=(m, v, r, 4)).w0 & (BS-1) with m = running count of such
 *     events on this lane, and reject unless off < size(last bin).  Accepted
 *     draws increment n1[r][c] until the quota is met.
 *   Stage 2 (per replicate r, coarse bin b; only when s > 0), n = n1[r][b]:
 *     full bin (covers 2^s whole tiles): draw d in [0, n) uses field d % F of
 *     Philox call c = d / F, F = 4 * floor(32 / s); field k of word w is bits
 *     [k*s, (k+1)*s) (k < floor(32/s)), fields ordered word-major;
 *     ctr = (c, b, r, 2).  The field value is the tile inside the bin.
 *     last bin when partial: 64 lanes, quota split of n; ctr = (j, b*64 + lane,
 *     r, 5); 16-bit slot z: off = z & (BS-1), reject unless off < size(bin b);
 *     tile inside the bin = off >> 10.
 *   Stage 3 (per replicate r, tile t), n = n2[r][t]:
 *     full tile: draw d uses field d % 12 of call c = d / 12 (three 10-bit
 *     fields per word: bits 0-9, 10-19, 20-29); ctr = (c, t, r, 3); the field
 *     value is the sample inside the tile.
 *     last tile when partial: 64 lanes, quota split of n; ctr = (j, t*64 + lane,
 *     r, 6); 16-bit slot z: off = z & 1023, reject unless off < size(tile t).
 *   freq[r][t*1024 + off] += 1 for every stage-3 draw.
 * Acceptance regions are exactly proportional to the number of samples a bin
 * covers, so every draw is uniform over [0, ndat) and the tables are exactly
 * multinomial.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SM_LT 10
#define SM_T 1024
#define SM_V1 16384
#define SM_NB1_MAX 16384

static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                          uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

typedef struct {
  int64_t ntiles, nb1, BS, last_bin_size, last_tile_size;
  int s, k1;
} sm_geom;

static int sm_geometry(int64_t ndat, sm_geom *g) {
  g->ntiles = (ndat + SM_T - 1) / SM_T;
  g->s = 0;
  while (((g->ntiles + ((int64_t)1 << g->s) - 1) >> g->s) > SM_NB1_MAX) g->s++;
  if (g->s > 6) return -1;
  g->nb1 = (g->ntiles + ((int64_t)1 << g->s) - 1) >> g->s;
  g->BS = (int64_t)SM_T << g->s;
  g->last_bin_size = ndat - (g->nb1 - 1) * g->BS;
  g->last_tile_size = ndat - (g->ntiles - 1) * SM_T;
  g->k1 = 0;
  while (((int64_t)1 << g->k1) < g->nb1) g->k1++;
  return 0;
}

int64_t orc_sampler_ntiles(int64_t ndat) { return (ndat + SM_T - 1) / SM_T; }

/* slot e of a 4-word Philox output */
static inline uint32_t slot16(const uint32_t o[4], int e) { return (o[e >> 1] >> (16 * (e & 1))) & 0xffffu; }

/* counts [nrep][ntiles] uint32 (stage 1 + stage 2) */
int orc_sampler_tile_counts(uint64_t seed, int64_t nrep, int64_t ndat, int64_t nsamp,
                            uint32_t *counts) {
  sm_geom g;
  if (ndat < 1 || nrep < 1 || sm_geometry(ndat, &g)) return -1;
  if (nsamp <= 0) nsamp = ndat;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t *n1 = (uint32_t *)calloc((size_t)g.nb1, sizeof(uint32_t));
  if (!n1) return -2;
  for (int64_t r = 0; r < nrep; ++r) {
    memset(n1, 0, sizeof(uint32_t) * (size_t)g.nb1);
    uint32_t *n2 = counts + r * g.ntiles;
    memset(n2, 0, sizeof(uint32_t) * (size_t)g.ntiles);
    /* ---- stage 1 ---- */
    if (g.nb1 == 1) {
      n1[0] = (uint32_t)nsamp;
    } else {
      const uint32_t mask = ((uint32_t)1 << g.k1) - 1;
      for (int64_t v = 0; v < SM_V1; ++v) {
        int64_t quota = nsamp / SM_V1 + (v < nsamp % SM_V1 ? 1 : 0);
        uint32_t j = 0, m = 0;
        while (quota > 0) {
          uint32_t o[4];
          philox4x32_10(j++, (uint32_t)v, (uint32_t)r, 1u, k0, k1, o);
          for (int e = 0; e < 8 && quota > 0; ++e) {
            const uint32_t c = slot16(o, e) & mask;
            if (c >= (uint32_t)g.nb1) continue;
            if (c == (uint32_t)(g.nb1 - 1) && g.last_bin_size < g.BS) {
              uint32_t o2[4];
              philox4x32_10(m++, (uint32_t)v, (uint32_t)r, 4u, k0, k1, o2);
              if ((int64_t)(o2[0] & (uint32_t)(g.BS - 1)) >= g.last_bin_size) continue;
            }
            n1[c]++;
            quota--;
          }
        }
      }
    }
    /* ---- stage 2 ---- */
    if (g.s == 0) {
      for (int64_t b = 0; b < g.nb1; ++b) n2[b] = n1[b];
    } else {
      const int fpw = 32 / g.s, F = 4 * fpw; /* fields per word / per call */
      const uint32_t fmask = ((uint32_t)1 << g.s) - 1;
      for (int64_t b = 0; b < g.nb1; ++b) {
        const int64_t n = n1[b];
        const int64_t size_b = (b == g.nb1 - 1) ? g.last_bin_size : g.BS;
        if (size_b == g.BS) {
          for (int64_t c = 0; c * F < n; ++c) {
            uint32_t o[4];
            philox4x32_10((uint32_t)c, (uint32_t)b, (uint32_t)r, 2u, k0, k1, o);
            const int64_t nd = (n - c * F < F) ? n - c * F : F;
            for (int64_t q = 0; q < nd; ++q)
              n2[(b << g.s) + ((o[q / fpw] >> (g.s * (int)(q % fpw))) & fmask)]++;
          }
        } else {
          for (int lane = 0; lane < 64; ++lane) {
            int64_t quota = n / 64 + (lane < n % 64 ? 1 : 0);
            uint32_t j = 0;
            while (quota > 0) {
              uint32_t o[4];
              philox4x32_10(j++, (uint32_t)(b * 64 + lane), (uint32_t)r, 5u, k0, k1, o);
              for (int e = 0; e < 8 && quota > 0; ++e) {
                const int64_t off = slot16(o, e) & (uint32_t)(g.BS - 1);
                if (off >= size_b) continue;
                n2[(b << g.s) + (off >> SM_LT)]++;
                quota--;
              }
            }
          }
        }
      }
    }
  }
  free(n1);
  return 0;
}

/* freq [nrep][ndat] int64 from counts (stage 3) */
int orc_sampler_freq(uint64_t seed, int64_t nrep, int64_t ndat, const uint32_t *counts,
                     int64_t *freq) {
  sm_geom g;
  if (ndat < 1 || nrep < 1 || sm_geometry(ndat, &g)) return -1;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  memset(freq, 0, sizeof(int64_t) * (size_t)nrep * ndat);
  for (int64_t r = 0; r < nrep; ++r)
    for (int64_t t = 0; t < g.ntiles; ++t) {
      const int64_t n = counts[r * g.ntiles + t];
      const int64_t size_t_ = (t == g.ntiles - 1) ? g.last_tile_size : SM_T;
      if (size_t_ == SM_T) {
        for (int64_t c = 0; c * 12 < n; ++c) {
          uint32_t o[4];
          philox4x32_10((uint32_t)c, (uint32_t)t, (uint32_t)r, 3u, k0, k1, o);
          const int64_t nd = (n - c * 12 < 12) ? n - c * 12 : 12;
          for (int64_t q = 0; q < nd; ++q)
            freq[r * ndat + t * SM_T + ((o[q / 3] >> (10 * (int)(q % 3))) & 1023u)]++;
        }
      } else {
        for (int lane = 0; lane < 64; ++lane) {
          int64_t quota = n / 64 + (lane < n % 64 ? 1 : 0);
          uint32_t j = 0;
          while (quota > 0) {
            uint32_t o[4];
            philox4x32_10(j++, (uint32_t)(t * 64 + lane), (uint32_t)r, 6u, k0, k1, o);
            for (int e = 0; e < 8 && quota > 0; ++e) {
              const int64_t off = slot16(o, e) & (SM_T - 1);
              if (off >= size_t_) continue;
              freq[r * ndat + t * SM_T + off]++;
              quota--;
            }
          }
        }
      }
    }
  return 0;
}
