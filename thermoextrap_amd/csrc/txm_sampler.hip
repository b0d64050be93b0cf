// txm_sampler.hip -- sampler kernels: indices -> freq histogram (parity mode,
// cmomy's indices_to_freq) and the counter-based exact multinomial sampler
// (scale mode) that replaces `rng.choice(ndat, (nrep, ndat))` + histogram
// (cmomy factory_sampler as reached from data.py:1782-1789) without ever
// materialising the (nrep, ndat) index/freq tables.
//
// Tile counts: recursive binomial splitting over a count-balanced tree of tile ranges, one binomial variate
//          (BTRS) per node, one workgroup per replicate (sampler_tree_kernel; stream version 3).
// Per-sample counts inside a tile ("stage 3"): txm_sampler.h, executed inside the
//          bootstrap kernels (txm_resample*.hip), tile by tile, straight into LDS.
#include "txm_sampler.h"

namespace txm {

// `err` is a word of the CALLER's workspace: concurrent calls on different streams do not share it
__global__ __launch_bounds__(256) void indices_to_freq_kernel(const int64_t *__restrict__ idx,
                                                              int64_t nrep, int64_t nsamp,
                                                              int64_t ndat,
                                                              int64_t *__restrict__ freq,
                                                              int *__restrict__ err) {
  for (int64_t r = blockIdx.y; r < nrep; r += gridDim.y) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nsamp;
         k += (int64_t)gridDim.x * blockDim.x) {
      const int64_t j = idx[r * nsamp + k];
      if (j < 0 || j >= ndat) {
        *err = 1;
      } else {
        atomicAdd(reinterpret_cast<unsigned long long *>(freq + r * ndat + j), 1ULL);
      }
    }
  }
}

// ---- tile counts: recursive binomial splitting, stream v3 (oracle/philox_oracle.c is the normative statement) --------
// Version 2 counted one random bit per draw and tree level (1.85e12 bits at N = 1e8, nrep = 1000: 20.9 ms, the
// vector-issue bound of Philox).  Version 3 draws ONE binomial variate per tree node with O(1) uniforms -- Hormann's
// transformed rejection with squeeze (BTRS) -- over a count-balanced tree of tile ranges, and keeps the bitwise rule
// for the nodes with n p < 10.  Everything floating point below is IEEE double + - * / floor in a fixed order (this
// file is compiled with -ffp-contract=off; sqrt and log are the fixed sequences det_sqrt / det_log), so that the
// tables equal the CPU restatement bit for bit.
#pragma clang fp contract(off)

__device__ __forceinline__ uint32_t popc4(const Philox4 &o) {
  return (uint32_t)(__popc(o.w[0]) + __popc(o.w[1]) + __popc(o.w[2]) + __popc(o.w[3]));
}

// number of 1 bits among the first `nbits` bits of bit stream (h, tagj) of replicate r (one lane walks the calls)
__device__ __forceinline__ uint32_t stream_ones(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t tagj,
                                                uint32_t nbits) {
  const uint32_t full = nbits >> 7, tail = nbits & 127u;
  uint32_t acc = 0;
  for (uint32_t c = 0; c < full; ++c) acc += popc4(philox4x32_10<true>(h, c, r, tagj, k0, k1));
  if (tail != 0u) {
    const Philox4 o = philox4x32_10<true>(h, full, r, tagj, k0, k1);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int nb = (int)tail - 32 * w;
      const uint32_t m = nb >= 32 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
      acc += (uint32_t)__popc(o.w[w] & m);
    }
  }
  return acc;
}

// small nodes: left ~ Binomial(n, A / (A + B)), all n uniforms compared with p bit by bit at once (integers only)
__device__ __forceinline__ uint32_t split_left_bits(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n,
                                                    int64_t A, int64_t B) {
  uint64_t rem = (uint64_t)A;
  const uint64_t den = (uint64_t)A + (uint64_t)B;
  uint32_t tie = n, left = 0;
  for (uint32_t j = 0; tie > 0u && j < 255u; ++j) {
    rem <<= 1;
    const bool pj = rem >= den;
    if (pj) rem -= den;
    const uint32_t K = stream_ones(k0, k1, h, r, 1u + 256u * j, tie);
    if (pj) {
      left += tie - K;
      tie = K;
    } else {
      tie -= K;
    }
    if (rem == 0) break;
  }
  return left;
}

__device__ __forceinline__ double bits_f64(uint64_t b) { return __longlong_as_double((long long)b); }

// sqrt(x), x > 0 normal: x = m 4^e, m in [1, 4); linear seed, four Newton steps (same sequence as the oracle's)
__device__ __forceinline__ double det_sqrt(double x) {
  const int E = (int)(((uint64_t)__double_as_longlong(x) >> 52) & 0x7ffu) - 1023;
  const int e = E >> 1;
  const double m = x * bits_f64((uint64_t)(1023 - 2 * e) << 52);
  double s = m / 3.0 + 0.72;
  s = 0.5 * (s + m / s);
  s = 0.5 * (s + m / s);
  s = 0.5 * (s + m / s);
  s = 0.5 * (s + m / s);
  return s * bits_f64((uint64_t)(1023 + e) << 52);
}

// log(x), x > 0 normal: x = 2^k (1 + f), s = f / (2 + f), log(1 + f) = 2 s + s R(s^2)
__device__ __forceinline__ double det_log(double x) {
  constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  constexpr double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                   L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                   L7 = 1.479819860511658591e-01;
  uint64_t bits = (uint64_t)__double_as_longlong(x);
  int k = (int)((bits >> 52) & 0x7ffu) - 1023;
  bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double m = bits_f64(bits);
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    k = k + 1;
  }
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

__device__ __forceinline__ double stirling_tail(double k) {
  if (k <= 9.0) {
    // (a select chain instead of a table: per-lane indexing of a constant array is a memory load)
    const int i = (int)k;
    double t = 0.08106146679532726;
    t = i == 1 ? 0.04134069595540929 : t;
    t = i == 2 ? 0.02767792568499834 : t;
    t = i == 3 ? 0.02079067210376509 : t;
    t = i == 4 ? 0.01664469118982119 : t;
    t = i == 5 ? 0.01387612882307075 : t;
    t = i == 6 ? 0.01189670994589177 : t;
    t = i == 7 ? 0.01041126526197209 : t;
    t = i == 8 ? 0.009255462182712733 : t;
    t = i == 9 ? 0.008330563433362871 : t;
    return t;
  }
  const double kp1 = k + 1.0;
  const double kp1sq = kp1 * kp1;
  return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0) / kp1sq) / kp1sq) / kp1;
}

__device__ __forceinline__ double u52(uint32_t hi, uint32_t lo) {
  const uint64_t j = ((uint64_t)hi << 20) | (uint64_t)(lo >> 12);
  return ((double)j + 0.5) * 0x1p-52;
}

// x ~ Binomial(n, p), p <= 1/2, n p >= 10: Hormann (1993), algorithm BTRS.  Attempt `att` of node h, replicate r takes
// its two uniforms from Philox(ctr = (h, att, r, 7)).  ~86 % of the attempts end in the first test.
// (the constants of a node, and ONE attempt: the walk below interleaves the attempts of different nodes across the lanes)
struct Btrs {
  double dn, p, q, spq, b, a, c, vr;
};
__device__ __forceinline__ Btrs btrs_setup(uint32_t n, double p) {
  Btrs t;
  t.dn = (double)n;
  t.p = p;
  t.q = 1.0 - p;
  t.spq = det_sqrt(t.dn * p * t.q);
  t.b = 1.15 + 2.53 * t.spq;
  t.a = -0.0873 + 0.0248 * t.b + 0.01 * p;
  t.c = t.dn * p + 0.5;
  t.vr = 0.92 - 4.2 / t.b;
  return t;
}
// -> accepted?; *x = the variate
__device__ __forceinline__ bool btrs_attempt(const Btrs &t, uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t att, uint32_t *x) {
  const Philox4 o = philox4x32_10<true>(h, att, r, 7u, k0, k1);
  const double u = u52(o.w[0], o.w[1]) - 0.5;
  double v = u52(o.w[2], o.w[3]);
  const double us = 0.5 - (u < 0.0 ? -u : u);
  const double kf = __builtin_floor((2.0 * t.a / us + t.b) * u + t.c);
  if (us >= 0.07 && v <= t.vr) {
    *x = (uint32_t)kf;
    return true;
  }
  if (kf < 0.0 || kf > t.dn) return false;
  *x = (uint32_t)kf;
  const double alpha = (2.83 + 5.1 / t.b) * t.spq;
  const double rr = t.p / t.q;
  const double m = __builtin_floor((t.dn + 1.0) * t.p);
  v = det_log(v * alpha / (t.a / (us * us) + t.b));
  const double bound = (m + 0.5) * det_log((m + 1.0) / (rr * (t.dn - m + 1.0))) +
                       (t.dn + 1.0) * det_log((t.dn - m + 1.0) / (t.dn - kf + 1.0)) +
                       (kf + 0.5) * det_log(rr * (t.dn - kf + 1.0) / (kf + 1.0)) +
                       ((stirling_tail(m) + stirling_tail(t.dn - m)) - (stirling_tail(kf) + stirling_tail(t.dn - kf)));
  return v <= bound;
}
__device__ __forceinline__ uint32_t btrs(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n, double p) {
  const Btrs t = btrs_setup(n, p);
  for (uint32_t att = 0;; ++att) {
    uint32_t x;
    if (btrs_attempt(t, k0, k1, h, r, att, &x)) return x;
  }
}

// tile boundary floor(i nt / 2^l) of the count-balanced tree; samples of [0, ndat) under node (l, i)
__device__ __forceinline__ int64_t tree_node_size(int64_t ndat, int64_t nt, int l, int64_t i) {
  const int64_t lo = ((i * nt) >> l) * SM_T, hi = (((i + 1) * nt) >> l) * SM_T;
  return (hi < ndat ? hi : ndat) - (lo < ndat ? lo : ndat);
}

// draws of node (l, i) (heap index h, n draws) that go to its left child
__device__ __forceinline__ uint32_t split_left(uint32_t k0, uint32_t k1, const SamplerGeom &g, int l, int64_t i,
                                               uint32_t r, uint32_t n) {
  if (n == 0u) return 0u;
  const int64_t A = tree_node_size(g.ndat, g.ntiles, l + 1, 2 * i), B = tree_node_size(g.ndat, g.ntiles, l + 1, 2 * i + 1);
  if (A == 0) return 0u;
  if (B == 0) return n;
  const uint32_t h = (1u << l) + (uint32_t)i;
  const int64_t S = A <= B ? A : B;
  const double p = (double)S / (double)(A + B);
  if ((double)n * p >= 10.0) {
    const uint32_t x = btrs(k0, k1, h, r, n, p);
    return A <= B ? x : n - x;
  }
  return split_left_bits(k0, k1, h, r, n, A, B);
}

// One workgroup per replicate.  Levels 0 .. L0 - 1 (L0 = k - D, D = the depth of the lane-private subtrees) are split
// breadth first, one thread per node, in a heap in LDS.  Below level L0 every LANE owns a subtree: it walks its 2^D
// leaves left to right and splits the nodes it enters on the way down (depth-first; the pending right siblings of the
// path sit in a D-entry stack in LDS).  All lanes run the same loop -- the tree has one shape -- so the only divergence
// is inside a split (empty nodes, the rejection loop).
// A node's variate is a function of (level, index, replicate) alone: D and the workgroup size are execution parameters, the
// table does not depend on them.  D = k - log2(threads) gives every lane a subtree (round 6: a fixed D = 5 left 32 of 256
// lanes busy on the 977-tile series of BASELINE config 5 -- 5 + 63 dependent splits a replicate where 8 + 3 do; 690 us of
// an 8 ms step); calls with few replicates take 1024 threads a replicate and several workgroups a replicate (below).
// grid (nrep, 2^M), block THREADS, dynamic LDS: heap[2^(L0+1)] + stack[(D + 1) * THREADS]
constexpr int TR_MAX_L0 = 13;  // the heap of the top levels: 2^(L0+1) words of LDS
// log2 of the workgroups a replicate may be cut into.  The kernel takes any M (the tables do not depend on it), the launcher uses 0:
// measured at N = 1e7, 200 replicates, four workgroups a replicate took 0.27 ms against 0.19 ms with one -- the dependent splits of
// the top levels are the chain either way, and every part repeats its own path from the root.
constexpr int TR_MAX_M = 0;

template <int THREADS>
__global__ __launch_bounds__(THREADS) void sampler_tree_kernel(uint32_t k0, uint32_t k1key, uint32_t nsamp,
                                                               SamplerGeom g, uint32_t rep0, int D, int M,
                                                               uint32_t *__restrict__ counts) {
  extern __shared__ uint32_t tree_lds[];
  // this workgroup's part of the replicate's tree: the subtree under node (M, part) -- 2^M workgroups a replicate where the
  // replicates alone do not fill the chip; each walks the M splits of its own path from the root first
  const int64_t part = (int64_t)blockIdx.y;
  const int k = g.k - M, L0 = k - D;
  uint32_t *heap = tree_lds;                       // levels 0 .. L0 of the part, heap indexed
  uint32_t *stk = heap + ((size_t)2 << L0);        // [D + 1][THREADS]
  const uint32_t r = rep0 + blockIdx.x;            // replicate of the STREAM; row blockIdx.x of this call's table
  const int tid = (int)threadIdx.x;

  if (tid == 0) {
    uint32_t n = nsamp;
    for (int l = 0; l < M; ++l) {
      const uint32_t left = split_left(k0, k1key, g, l, part >> (M - l), r, n);
      n = ((part >> (M - l - 1)) & 1) != 0 ? n - left : left;
    }
    heap[1] = n;
  }
  __syncthreads();
  for (int l = 0; l < L0; ++l) {
    const uint32_t nn = 1u << l;
    for (uint32_t i = (uint32_t)tid; i < nn; i += THREADS) {
      const uint32_t n = heap[nn + i];
      const uint32_t left = split_left(k0, k1key, g, M + l, (part << l) + (int64_t)i, r, n);
      heap[2 * nn + 2 * i] = left;
      heap[2 * nn + 2 * i + 1] = n - left;
    }
    __syncthreads();
  }

  // The subtrees: every lane walks its own (depth first), ONE BTRS attempt per turn of the loop.  A lane whose attempt is accepted
  // moves on to its next node in the same turn; one whose attempt is rejected tries again in the next -- no lane waits for the
  // slowest of its wave (a loop over the nodes with the rejection loop inside ran max-over-64-lanes attempts a node, ~3.5 where a
  // lane needs 1.16).  The variates are those of split_left: same constants, same attempts, same order per node.
  const uint32_t nroots = 1u << L0, nleaf = 1u << D;
  uint32_t sub = (uint32_t)tid, c = 0, cur = 0, att = 0, h = 0;
  int d = 0;
  bool run = sub < nroots, fresh = true, flip = false;
  Btrs bt = {};
  if (run) cur = heap[nroots + sub];
#pragma unroll 1
  for (;;) {
    // leaves: store the count, back up to the pending right sibling (the next subtree of this lane behind the last leaf)
    while (run && d == D) {
      const int64_t leaf = (part << k) + ((int64_t)sub << D) + c;
      const int64_t lo = (leaf * g.ntiles) >> g.k, hi = ((leaf + 1) * g.ntiles) >> g.k;
      if (hi > lo) counts[(size_t)blockIdx.x * g.ntiles + lo] = cur;
      ++c;
      if (c == nleaf) {
        sub += THREADS;
        run = sub < nroots;
        c = 0;
        d = 0;
        if (run) cur = heap[nroots + sub];
      } else {
        d = D - __builtin_ctz(c);
        cur = stk[d * THREADS + tid];
      }
    }
    if (__ballot(run) == 0ull) break;
    if (run) {
      uint32_t left = 0;
      bool have = false;
      if (fresh) {  // a new node: the cases split_left decides without a variate, the bitwise rule, or the constants of BTRS
        const int l = M + L0 + d;
        const int64_t gi = (part << (L0 + d)) + ((int64_t)sub << d) + (int64_t)(c >> (D - d));
        const int64_t A = tree_node_size(g.ndat, g.ntiles, l + 1, 2 * gi), B = tree_node_size(g.ndat, g.ntiles, l + 1, 2 * gi + 1);
        h = (1u << l) + (uint32_t)gi;
        if (cur == 0u || A == 0) {
          have = true;
        } else if (B == 0) {
          left = cur;
          have = true;
        } else {
          const int64_t S = A <= B ? A : B;
          const double p = (double)S / (double)(A + B);
          if ((double)cur * p >= 10.0) {
            bt = btrs_setup(cur, p);
            flip = !(A <= B);
            att = 0;
            fresh = false;
          } else {
            left = split_left_bits(k0, k1key, h, r, cur, A, B);
            have = true;
          }
        }
      }
      if (!have) {
        uint32_t x;
        if (btrs_attempt(bt, k0, k1key, h, r, att, &x)) {
          left = flip ? cur - x : x;
          have = true;
          fresh = true;
        } else {
          ++att;
        }
      }
      if (have) {
        stk[(d + 1) * THREADS + tid] = cur - left;
        cur = left;
        ++d;
      }
    }
  }
}

// ---- materialise freq (testing / small sizes) -------------------------------
// one wave per (r, t); block 256 = 4 waves, each with a private 1024-bin tile.
__global__ __launch_bounds__(256) void sampler_freq_kernel(uint32_t k0, uint32_t k1key,
                                                           int64_t nrep, int64_t ndat,
                                                           SamplerGeom g, uint32_t rep0,
                                                           const uint32_t *__restrict__ counts,
                                                           int64_t *__restrict__ freq) {
  __shared__ uint32_t tile[4][SM_T];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t task = (int64_t)blockIdx.x * 4 + wave;
  const bool active = task < nrep * g.ntiles;
  const uint32_t r = active ? (uint32_t)(task / g.ntiles) : 0u;
  const uint32_t t = active ? (uint32_t)(task % g.ntiles) : 0u;
  for (int i = lane; i < SM_T; i += 64) tile[wave][i] = 0u;
  __syncthreads();
  const uint32_t tsize = (t == (uint32_t)g.ntiles - 1u) ? (uint32_t)g.last_tile_size : (uint32_t)SM_T;
  if (active) {
    const uint32_t n = counts[(size_t)r * g.ntiles + t];
    uint32_t *tl = tile[wave];
    sampler_fine_tile(k0, k1key, rep0 + r, t, n, tsize, lane, [&](uint32_t off) { atomicAdd(&tl[off], 1u); });
  }
  __syncthreads();
  if (active) {
    for (uint32_t i = lane; i < tsize; i += 64)
      freq[(size_t)r * ndat + (size_t)t * SM_T + i] = (int64_t)tile[wave][i];
  }
}

}  // namespace txm

using namespace txm;

extern "C" size_t txm_indices_to_freq_ws_bytes(void) { return 256; }

extern "C" int txm_indices_to_freq(const int64_t *indices, int64_t nrep, int64_t nsamp,
                                   int64_t ndat, int64_t *freq, void *ws, size_t ws_bytes,
                                   txm_stream stream) {
  TXM_REQUIRE(indices && freq && ws, "indices_to_freq: null pointer");
  TXM_REQUIRE(nrep >= 1 && nsamp >= 1 && ndat >= 1, "indices_to_freq: bad sizes");
  if (ws_bytes < txm_indices_to_freq_ws_bytes()) {
    set_error("indices_to_freq: workspace too small (%zu < 256)", ws_bytes);
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int *err = reinterpret_cast<int *>(ws);
  TXM_HIP(hipMemsetAsync(freq, 0, sizeof(int64_t) * (size_t)nrep * ndat, st));
  TXM_HIP(hipMemsetAsync(err, 0, sizeof(int), st));
  int gx = (int)cdiv(nsamp, 256 * 4);
  if (gx > 4096) gx = 4096;
  const unsigned gy = (unsigned)(nrep < 65535 ? nrep : 65535);
  hipLaunchKernelGGL(indices_to_freq_kernel, dim3(gx, gy), dim3(256), 0, st, indices, nrep, nsamp, ndat, freq, err);
  TXM_LAUNCH_CHECK();
  int flag = 0;
  TXM_HIP(hipMemcpyAsync(&flag, err, sizeof(int), hipMemcpyDeviceToHost, st));
  TXM_HIP(hipStreamSynchronize(st));
  if (flag) {
    set_error("indices_to_freq: index outside [0, %lld)", (long long)ndat);
    return TXM_ERR_INVALID;
  }
  return TXM_OK;
}

extern "C" int64_t txm_sampler_ntiles(int64_t ndat) { return ndat < 1 ? 0 : (ndat + SM_T - 1) / SM_T; }

static int check_spec(const txm_sampler_spec *sp, SamplerGeom *g, int64_t *nsamp) {
  TXM_REQUIRE(sp, "sampler: null spec");
  TXM_REQUIRE(sp->nrep >= 1 && sp->nrep <= ((int64_t)1 << 24), "sampler: nrep=%lld outside [1, 2^24]",
              (long long)sp->nrep);
  TXM_REQUIRE(sp->rep0 >= 0 && sp->rep0 + sp->nrep <= ((int64_t)1 << 32),
              "sampler: stream replicates [%lld, %lld) outside [0, 2^32)", (long long)sp->rep0,
              (long long)(sp->rep0 + sp->nrep));
  TXM_REQUIRE(sp->ndat >= 1, "sampler: ndat < 1");
  *nsamp = sp->nsamp > 0 ? sp->nsamp : sp->ndat;
  TXM_REQUIRE(*nsamp < ((int64_t)1 << 32), "sampler: nsamp >= 2^32 unsupported");
  // per-sample counts live in 8-bit LDS counters inside the bootstrap kernel: keep
  // the mean count <= 16 so that an overflow (count >= 256) is beyond any reachable tail
  TXM_REQUIRE(*nsamp <= 16 * sp->ndat, "sampler: nsamp > 16 * ndat unsupported (8-bit per-sample counters)");
  if (sampler_geometry(sp->ndat, g) != 0) {
    set_error("sampler: ndat=%lld too large (max 2^30)", (long long)sp->ndat);
    return TXM_ERR_UNSUPPORTED;
  }
  return TXM_OK;
}

extern "C" size_t txm_sampler_counts_ws_bytes(const txm_sampler_spec *sp) {
  SamplerGeom g;
  int64_t nsamp;
  if (check_spec(sp, &g, &nsamp) != TXM_OK) return 0;
  return 256;  // the tile tree lives in LDS; kept so that callers' workspace plumbing stays valid
}

extern "C" int txm_sampler_tile_counts(const txm_sampler_spec *sp, uint32_t *counts, void *ws,
                                       size_t ws_bytes, txm_stream stream) {
  SamplerGeom g;
  int64_t nsamp;
  int rc = check_spec(sp, &g, &nsamp);
  if (rc != TXM_OK) return rc;
  TXM_REQUIRE(counts, "sampler: null counts");
  (void)ws;
  (void)ws_bytes;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t k0 = (uint32_t)sp->seed, k1 = (uint32_t)(sp->seed >> 32);
  // few replicates: more lanes a replicate, and 2^M workgroups a replicate (each the subtree under a node of level M) until the
  // grid covers the chip twice -- while a part keeps a subtree a lane
  const bool wide = sp->nrep < 1024;
  const int lg = wide ? 10 : 8, threads = wide ? 1024 : 256;
  int M = 0;
  while (M < TR_MAX_M && (sp->nrep << M) < 512 && g.k - M > lg) ++M;
  // lane-private subtrees below a heap of at most 2^TR_MAX_L0 nodes: one subtree a lane where the tree is deep enough
  const int kp = g.k - M;
  int D = kp > lg ? kp - lg : 0;
  if (kp - D > TR_MAX_L0) D = kp - TR_MAX_L0;
  const size_t lds = (((size_t)2 << (kp - D)) + (size_t)(D + 1) * threads) * sizeof(uint32_t);
  const dim3 grid((unsigned)sp->nrep, 1u << M);
  if (wide) {
    TXM_SET_MAX_LDS(sampler_tree_kernel<1024>, 160 * 1024);
    hipLaunchKernelGGL(sampler_tree_kernel<1024>, grid, dim3(1024), lds, st, k0, k1, (uint32_t)nsamp, g, (uint32_t)sp->rep0, D, M, counts);
  } else {
    TXM_SET_MAX_LDS(sampler_tree_kernel<256>, 160 * 1024);
    hipLaunchKernelGGL(sampler_tree_kernel<256>, grid, dim3(256), lds, st, k0, k1, (uint32_t)nsamp, g, (uint32_t)sp->rep0, D, M, counts);
  }
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" int txm_sampler_freq(const txm_sampler_spec *sp, const uint32_t *counts, int64_t *freq,
                                txm_stream stream) {
  SamplerGeom g;
  int64_t nsamp;
  int rc = check_spec(sp, &g, &nsamp);
  if (rc != TXM_OK) return rc;
  TXM_REQUIRE(counts && freq, "sampler_freq: null pointer");
  const int64_t tasks = sp->nrep * g.ntiles;
  TXM_REQUIRE(cdiv(tasks, 4) < ((int64_t)1 << 31), "sampler_freq: too many tiles");
  const uint32_t k0 = (uint32_t)sp->seed, k1 = (uint32_t)(sp->seed >> 32);
  hipLaunchKernelGGL(sampler_freq_kernel, dim3((unsigned)cdiv(tasks, 4)), dim3(256), 0,
                     (hipStream_t)stream, k0, k1, sp->nrep, sp->ndat, g, (uint32_t)sp->rep0, counts, freq);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}
