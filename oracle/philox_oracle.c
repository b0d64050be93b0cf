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
 * Philox4x32-10 stream.  This file is the
 * normative statement of that stream; tests compare the GPU's tables with it
 * bit for bit.
 *
 * Spec, stream version 3 (round 4; version 2 spent one random bit per draw and tree level -- 1.85e12 bits at
 * N = 1e8, nrep = 1000 -- version 3 spends O(1) uniforms per tree node):
 *   T = 1024 samples per tile;  nt = ceil(ndat / T);  k = smallest integer with 2^k >= nt.
 *   Tile counts (per replicate r) by recursive binomial splitting over a COUNT-BALANCED binary tree of tile ranges:
 *     node (l, i), l = 0..k, i in [0, 2^l), covers tiles [b(l, i), b(l, i+1)),  b(l, i) = floor(i * nt / 2^l)
 *     (so the two children of a node differ by at most one tile, and a leaf (k, i) covers one tile or none);
 *     size(l, i) = number of samples of [0, ndat) inside that tile range;  heap index h = 2^l + i.
 *     n(0, 0) = nsamp.  A node with l < k and n = n(l, i) > 0 sends `left` of its draws to child (l+1, 2i) and
 *     n - left to (l+1, 2i+1), left ~ Binomial(n, A / (A + B)), A = size(l+1, 2i), B = size(l+1, 2i+1):
 *       B == 0: left = n;  A == 0: left = 0 (no random numbers).
 *       otherwise S = min(A, B), p = (double)S / (double)(A + B) <= 1/2 and
 *         n * p >= 10 (as doubles): x = BTRS(n, p) below, the draws of the SMALLER child (A <= B: left = x, else
 *                      left = n - x);
 *         else       : the version-2 rule -- every draw compares a uniform binary fraction with A / (A + B) bit by
 *                      bit, all draws of the node at once (split_left_bits; integers only).
 *     BTRS(n, p): Hormann's transformed rejection with squeeze ("The generation of binomial random variates",
 *       J. Statist. Comput. Simul. 46 (1993), algorithm BTRS), restated with IEEE-754 double + - * / and floor ONLY,
 *       in the order written in btrs() below, no fused multiply-add -- sqrt and log are the fixed sequences
 *       det_sqrt / det_log of this file -- so that a device and a CPU produce the SAME integers.  Attempt a = 0, 1, ...
 *       takes its two uniforms from Philox(key = seed, ctr = (h, a, r, 7)) = (w0, w1, w2, w3):
 *         U = ((w0 << 20 | w1 >> 12) + 0.5) 2^-52,  V = ((w2 << 20 | w3 >> 12) + 0.5) 2^-52   (52-bit, never 0 or 1).
 *       Exactness: BTRS is an exact rejection sampler in real arithmetic; this restatement differs from it by double
 *       rounding, the 52-bit uniforms and the truncation of Stirling's series in the acceptance bound (< 1e-10
 *       relative, reached on ~14 % of the attempts only) -- the class of numpy's Generator.binomial, which the
 *       reference's rng.choice + bincount tables do not need; the version-2 stream was exact in integers at 3-5 x the
 *       cost of the whole bootstrap's sampler stage.
 *     bit stream (h, j) of the version-2 rule: bit q is bit (q & 31) of word ((q >> 5) & 3) of Philox(key = seed,
 *     ctr = (h, q >> 7, r, 1 + 256 * j)).
 *     counts[r][t] = n(k, i) for the leaf (k, i) that covers tile t.
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
 * Conditional binomials of the children's sample-count ratios compose to multinomial(nsamp; size(t) / ndat) tile
 * counts and, with the uniform draws inside a tile, to multinomial(nsamp; 1/ndat ...) tables (to the accuracy stated
 * for BTRS above; the small-n rule and the per-sample stage are exact in integers).
 */
#include <math.h>
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

/* tile boundary b(l, i) = floor(i * nt / 2^l) of the count-balanced tree and the samples of [0, ndat) below it */
static int64_t node_bound(int64_t nt, int l, int64_t i) { return (i * nt) >> l; }
static int64_t samples_below(int64_t ndat, int64_t tile) {
  const int64_t s = tile * SM_T;
  return s < ndat ? s : ndat;
}
/* samples of [0, ndat) under node (l, i) */
static int64_t node_size(int64_t ndat, int64_t nt, int l, int64_t i) {
  return samples_below(ndat, node_bound(nt, l, i + 1)) - samples_below(ndat, node_bound(nt, l, i));
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

/* small nodes: left ~ Binomial(n, A / (A + B)), bitwise comparison of all n uniforms with p at once (integers only):
 *   rem = A, den = A + B, tie = n, left = 0;  for j = 0, 1, ...:
 *     rem = 2 * rem;  pj = (rem >= den);  if pj: rem -= den          (the next bit of p = A / (A + B))
 *     K = number of 1 bits among the first `tie` bits of bit stream (h, j)
 *     if pj: left += tie - K, tie = K      (b = 0 < pj = 1: decided left;  b = 1: still tied)
 *     else : tie = tie - K                 (b = 1 > pj = 0: decided right; b = 0: still tied)
 *     stop when tie == 0, or rem == 0 (p exhausted: the ties have U >= p, right), or j == 254. */
static uint32_t split_left_bits(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n, int64_t A, int64_t B) {
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

/* ---- deterministic double arithmetic: + - * / floor and bit moves only, evaluated exactly as written ------------ */
static inline uint64_t d2b(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
static inline double b2d(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

/* sqrt(x), x > 0 normal: x = m 4^e, m in [1, 4); linear seed (5 % off at most), four Newton steps */
static double det_sqrt(double x) {
  const int E = (int)((d2b(x) >> 52) & 0x7ffu) - 1023;
  const int e = E >> 1;                                   /* floor(E / 2) */
  const double m = x * b2d((uint64_t)(1023 - 2 * e) << 52); /* x 4^-e */
  double s = m / 3.0 + 0.72;
  s = 0.5 * (s + m / s);
  s = 0.5 * (s + m / s);
  s = 0.5 * (s + m / s);
  s = 0.5 * (s + m / s);
  return s * b2d((uint64_t)(1023 + e) << 52);
}

/* log(x), x > 0 normal: x = 2^k (1 + f), sqrt(1/2) < 1 + f <= sqrt(2); s = f / (2 + f);
 * log(1 + f) = 2 s + s R(s^2) with the classical degree-14 minimax polynomial (coefficients L1..L7 below) */
static double det_log(double x) {
  static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  static const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                      L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                      L7 = 1.479819860511658591e-01;
  uint64_t bits = d2b(x);
  int k = (int)((bits >> 52) & 0x7ffu) - 1023;
  bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull; /* mantissa in [1, 2) */
  double m = b2d(bits);
  if (m > 1.4142135623730951) { m = m * 0.5; k = k + 1; }
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (L2 + w * (L4 + w * L6));
  const double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

/* log(k!) - [(k + 1/2) log(k + 1) - (k + 1) + log(2 pi) / 2]: table for k <= 9, three terms of the series above */
static double stirling_tail(double k) {
  static const double T[10] = {0.08106146679532726, 0.04134069595540929, 0.02767792568499834, 0.02079067210376509,
                               0.01664469118982119, 0.01387612882307075, 0.01189670994589177, 0.01041126526197209,
                               0.009255462182712733, 0.008330563433362871};
  if (k <= 9.0) return T[(int)k];
  const double kp1 = k + 1.0;
  const double kp1sq = kp1 * kp1;
  return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0) / kp1sq) / kp1sq) / kp1;
}

static inline double u52(uint32_t hi, uint32_t lo) {
  const uint64_t j = ((uint64_t)hi << 20) | (uint64_t)(lo >> 12);
  return ((double)j + 0.5) * 0x1p-52;
}

/* x ~ Binomial(n, p), p <= 1/2, n p >= 10 */
static uint32_t btrs(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n, double p) {
  const double dn = (double)n;
  const double q = 1.0 - p;
  const double spq = det_sqrt(dn * p * q);
  const double b = 1.15 + 2.53 * spq;
  const double a = -0.0873 + 0.0248 * b + 0.01 * p;
  const double c = dn * p + 0.5;
  const double vr = 0.92 - 4.2 / b;
  for (uint32_t att = 0;; ++att) {
    uint32_t o[4];
    philox4x32_10(h, att, r, 7u, k0, k1, o);
    const double u = u52(o[0], o[1]) - 0.5;
    double v = u52(o[2], o[3]);
    const double us = 0.5 - (u < 0.0 ? -u : u);
    const double kf = floor((2.0 * a / us + b) * u + c);
    if (us >= 0.07 && v <= vr) return (uint32_t)kf;
    if (kf < 0.0 || kf > dn) continue;
    const double alpha = (2.83 + 5.1 / b) * spq;
    const double rr = p / q;
    const double m = floor((dn + 1.0) * p);
    v = det_log(v * alpha / (a / (us * us) + b));
    const double bound = (m + 0.5) * det_log((m + 1.0) / (rr * (dn - m + 1.0))) +
                         (dn + 1.0) * det_log((dn - m + 1.0) / (dn - kf + 1.0)) +
                         (kf + 0.5) * det_log(rr * (dn - kf + 1.0) / (kf + 1.0)) +
                         ((stirling_tail(m) + stirling_tail(dn - m)) - (stirling_tail(kf) + stirling_tail(dn - kf)));
    if (v <= bound) return (uint32_t)kf;
  }
}

static uint32_t split_left(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n, int64_t A, int64_t B) {
  if (n == 0 || A == 0) return 0;
  if (B == 0) return n;
  const int64_t S = A <= B ? A : B;
  const double p = (double)S / (double)(A + B);
  if ((double)n * p >= 10.0) {
    const uint32_t x = btrs(k0, k1, h, r, n, p);
    return A <= B ? x : n - x;
  }
  return split_left_bits(k0, k1, h, r, n, A, B);
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
        const uint32_t left = split_left(k0, k1, (uint32_t)(nn + i), (uint32_t)(rep0 + r), n,
                                         node_size(ndat, g.ntiles, l + 1, 2 * i), node_size(ndat, g.ntiles, l + 1, 2 * i + 1));
        nxt[2 * i] = left;
        nxt[2 * i + 1] = n - left;
      }
      uint32_t *t = cur; cur = nxt; nxt = t;
    }
    for (int64_t i = 0; i < (int64_t)P; ++i) {
      const int64_t lo = node_bound(g.ntiles, g.k, i), hi = node_bound(g.ntiles, g.k, i + 1);
      if (hi > lo) counts[r * g.ntiles + lo] = cur[i];
    }
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
