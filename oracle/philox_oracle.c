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
 * Spec (all integers), stream version 2:
 *   T = 1024 samples per tile;  ntiles = ceil(ndat / T);  k = smallest integer with 2^k >= ntiles.
 *   Tile counts (per replicate r) by recursive binomial splitting over a binary tree of tile ranges:
 *     node (l, i), l = 0..k, i in [0, 2^l), covers tiles [i * 2^(k-l), (i+1) * 2^(k-l));
 *     size(l, i) = number of samples of [0, ndat) inside that tile range;  heap index h = 2^l + i.
 *     n(0, 0) = nsamp.  A node with l < k and n = n(l, i) > 0 sends `left` of its draws to child
 *     (l+1, 2i) and n - left to (l+1, 2i+1), where left ~ Binomial(n, A / (A + B)) exactly,
 *     A = size(l+1, 2i), B = size(l+1, 2i+1):
 *       B == 0: left = n (no random bits).
 *       otherwise every draw compares a uniform binary fraction U = 0.b1 b2 ... with p = A / (A + B) =
 *       0.p1 p2 ... bit by bit (left iff U < p), all draws of the node at once:
 *         rem = A, den = A + B, tie = n, left = 0;  for j = 0, 1, ...:
 *           rem = 2 * rem;  pj = (rem >= den);  if pj: rem -= den
 *           K = number of 1 bits among the first `tie` bits of bit stream (h, j)
 *           if pj: left += tie - K, tie = K      (b = 0 < pj = 1: decided left;  b = 1: still tied)
 *           else : tie = tie - K                 (b = 1 > pj = 0: decided right; b = 0: still tied)
 *           stop when tie == 0, or rem == 0 (p exhausted: the ties have U >= p, right), or j == 254.
 *       (A == B is the one-step case p = 1/2: left = number of 0 bits among n.)  A >= B always, because
 *       only the last real node of a level can be partial.
 *     bit stream (h, j): bit q is bit (q & 31) of word ((q >> 5) & 3) of Philox(key = seed,
 *     ctr = (h, q >> 7, r, 1 + 256 * j)).  (The call index sits in the second counter word so that the
 *     first Philox rounds are partly the same for all calls of a node.)
 *     counts[r][t] = n(k, t).
 *   Replicate offset (txm_sampler_spec.rep0): row r of a call's tables is replicate rep0 + r of the stream, i.e.
 *     every `r` in a Philox counter above and below is rep0 + r.  A replicate's draws depend on (seed, stream
 *     replicate, tile) only, so rows [a, b) of the (seed, nrep) tables equal the (seed, b - a, rep0 = a) tables.
 *   Per-sample counts (per replicate r, tile t), n = counts[r][t]:
 *     full tile: draw d uses field d % 12 of call c = d / 12 (three 10-bit
 *     fields per word: bits 0-9, 10-19, 20-29); ctr = (t, c, r, 3); the field
 *     value is the sample inside the tile.
 *     last tile when partial: 64 lanes, quota split of n; ctr = (j, t*64 + lane,
 *     r, 6); 16-bit slot z: off = z & 1023, reject unless off < size(tile t).
 *   freq[r][t*1024 + off] += 1 for every such draw.
 * Every binomial split is exact for its ratio of sample counts, so the tile counts are exactly
 * multinomial(nsamp; size(t) / ndat) and, with the uniform draws inside a tile, the tables are exactly
 * multinomial(nsamp; 1/ndat ...).  Cost: one random bit per draw and level (two on the k nodes whose split
 * is not 1/2) instead of a 16-bit slot per draw -- version 1 of the stream binned every draw separately.
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
  int64_t ntiles, last_tile_size;
  int k;
} sm_geom;

static int sm_geometry(int64_t ndat, sm_geom *g) {
  if (ndat > ((int64_t)1 << 30)) return -1;
  g->ntiles = (ndat + SM_T - 1) / SM_T;
  g->last_tile_size = ndat - (g->ntiles - 1) * SM_T;
  g->k = 0;
  while (((int64_t)1 << g->k) < g->ntiles) g->k++;
  return 0;
}

int64_t orc_sampler_ntiles(int64_t ndat) { return (ndat + SM_T - 1) / SM_T; }

/* slot e of a 4-word Philox output */
static inline uint32_t slot16(const uint32_t o[4], int e) { return (o[e >> 1] >> (16 * (e & 1))) & 0xffffu; }

/* samples of [0, ndat) under node (l, i) of the tile tree */
static int64_t node_size(int64_t ndat, int k, int l, int64_t i) {
  const int64_t span = (int64_t)SM_T << (k - l);
  const int64_t lo = i * span;
  if (lo >= ndat) return 0;
  return (ndat - lo < span) ? ndat - lo : span;
}

static int popc32(uint32_t v) {
  int c = 0;
  while (v) { v &= v - 1; ++c; }
  return c;
}

/* number of 1 bits among the first nbits bits of bit stream (h, j) of replicate r */
static uint32_t stream_ones(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t j, uint32_t nbits) {
  uint32_t ones = 0;
  const uint32_t full = nbits >> 7, tail = nbits & 127u;
  for (uint32_t c = 0; c < full; ++c) {
    uint32_t o[4];
    philox4x32_10(h, c, r, 1u + 256u * j, k0, k1, o);
    ones += (uint32_t)(popc32(o[0]) + popc32(o[1]) + popc32(o[2]) + popc32(o[3]));
  }
  if (tail) {
    uint32_t o[4];
    philox4x32_10(h, full, r, 1u + 256u * j, k0, k1, o);
    for (int w = 0; w < 4; ++w) {
      const int nb = (int)tail - 32 * w;
      if (nb <= 0) break;
      ones += (uint32_t)popc32(nb >= 32 ? o[w] : (o[w] & (((uint32_t)1 << nb) - 1u)));
    }
  }
  return ones;
}

/* left ~ Binomial(n, A / (A + B)), bitwise comparison of all n uniforms with p at once */
static uint32_t split_left(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n, int64_t A, int64_t B) {
  if (B == 0) return n;
  uint64_t rem = (uint64_t)A;
  const uint64_t den = (uint64_t)A + (uint64_t)B;
  uint32_t tie = n, left = 0;
  for (uint32_t j = 0; tie > 0 && j < 255u; ++j) {
    rem <<= 1;
    const int pj = rem >= den;
    if (pj) rem -= den;
    const uint32_t K = stream_ones(k0, k1, h, r, j, tie);
    if (pj) { left += tie - K; tie = K; }
    else tie -= K;
    if (rem == 0) break;
  }
  return left;
}

/* counts [nrep][ntiles] uint32; row r = stream replicate rep0 + r */
int orc_sampler_tile_counts_rep0(uint64_t seed, int64_t nrep, int64_t ndat, int64_t nsamp, int64_t rep0,
                                 uint32_t *counts) {
  sm_geom g;
  if (ndat < 1 || nrep < 1 || rep0 < 0 || rep0 + nrep > ((int64_t)1 << 32) || sm_geometry(ndat, &g)) return -1;
  if (nsamp <= 0) nsamp = ndat;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const size_t P = (size_t)1 << g.k;
  uint32_t *cur = (uint32_t *)calloc(P, sizeof(uint32_t)), *nxt = (uint32_t *)calloc(P, sizeof(uint32_t));
  if (!cur || !nxt) { free(cur); free(nxt); return -2; }
  for (int64_t r = 0; r < nrep; ++r) {
    cur[0] = (uint32_t)nsamp;
    for (int l = 0; l < g.k; ++l) {
      const int64_t nn = (int64_t)1 << l;
      for (int64_t i = 0; i < nn; ++i) {
        const uint32_t n = cur[i];
        uint32_t left = 0;
        if (n > 0)
          left = split_left(k0, k1, (uint32_t)(nn + i), (uint32_t)(rep0 + r), n, node_size(ndat, g.k, l + 1, 2 * i),
                            node_size(ndat, g.k, l + 1, 2 * i + 1));
        nxt[2 * i] = left;
        nxt[2 * i + 1] = n - left;
      }
      uint32_t *t = cur; cur = nxt; nxt = t;
    }
    for (int64_t t = 0; t < g.ntiles; ++t) counts[r * g.ntiles + t] = cur[t];
  }
  free(cur);
  free(nxt);
  return 0;
}

int orc_sampler_tile_counts(uint64_t seed, int64_t nrep, int64_t ndat, int64_t nsamp, uint32_t *counts) {
  return orc_sampler_tile_counts_rep0(seed, nrep, ndat, nsamp, 0, counts);
}

/* freq [nrep][ndat] int64 from counts (stage 3); row r = stream replicate rep0 + r */
int orc_sampler_freq_rep0(uint64_t seed, int64_t nrep, int64_t ndat, int64_t rep0, const uint32_t *counts,
                          int64_t *freq) {
  sm_geom g;
  if (ndat < 1 || nrep < 1 || rep0 < 0 || rep0 + nrep > ((int64_t)1 << 32) || sm_geometry(ndat, &g)) return -1;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  memset(freq, 0, sizeof(int64_t) * (size_t)nrep * ndat);
  for (int64_t r = 0; r < nrep; ++r)
    for (int64_t t = 0; t < g.ntiles; ++t) {
      const int64_t n = counts[r * g.ntiles + t];
      const int64_t size_t_ = (t == g.ntiles - 1) ? g.last_tile_size : SM_T;
      if (size_t_ == SM_T) {
        for (int64_t c = 0; c * 12 < n; ++c) {
          uint32_t o[4];
          philox4x32_10((uint32_t)t, (uint32_t)c, (uint32_t)(rep0 + r), 3u, k0, k1, o);
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
            philox4x32_10(j++, (uint32_t)(t * 64 + lane), (uint32_t)(rep0 + r), 6u, k0, k1, o);
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

int orc_sampler_freq(uint64_t seed, int64_t nrep, int64_t ndat, const uint32_t *counts, int64_t *freq) {
  return orc_sampler_freq_rep0(seed, nrep, ndat, 0, counts, freq);
}
