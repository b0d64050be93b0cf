// txm_resample.hip -- sample-level bootstrap of central comoments
// (cmomy.wrap_resample_vals as called from thermoextrap data.py:1803-1810,
// 1354-1366).
//
// out[r] = comoment state of the data with weights w_i * f[r][i].  With the
// pivot-shifted monomials m[i][c][j] = (x_ic - px_c) (u_i - pu)^j this is a
// dense contraction over samples
//        S1[r][c][j] = sum_i f[r][i] * w_i * du_i^j * dx_ic
//        S0[r][j]    = sum_i f[r][i] * w_i * du_i^j
// i.e. ~2*K*(C+1) flop per (replicate, sample) against 8*(C+1) bytes per
// sample read once: FP64-bound for nrep >~ 12, so it runs on the FP64 matrix
// pipe.  Per wave and per 4 samples:
//        A_j (16 reps x 4 samples)  = f * w * du^j      one f64 per lane
//        B   (4 samples x 16 cols)  = dx                one f64 per lane
//        D_j (16 reps x 16 cols)   += A_j * B           v_mfma_f64_16x16x4_f64
// so the power index j rides on the A operand (one multiply per j) and B is
// loaded once per column block; S0 is the row-sum of A_j, kept on the VALU.
// The MFMA broadcasts f over columns and dx over replicates for free, and the
// VALU (Philox draws, LDS histogram, u-row sums) co-executes with the matrix
// pipe of the other resident wave.
//
// f comes either from the Philox stage-3 stream (scale mode; the wave fills a
// private [1024 samples][16 reps] u8 tile in LDS, never touching HBM) or from an
// explicit int64 freq table (parity mode).
#include <type_traits>

#include <stdlib.h>

#include "txm_sampler.h"
#include "txm_pivot.h"
#include "txm_resample_i8.h"
#include "txm_i8g.h"

namespace txm {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int RS_BLOCK = 256;          // 4 waves
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_REPS = 16;            // replicates per wave (MFMA M)
// f tile of one wave: u8 counts, layout [sample slot 0..1023][rep 0..15] (16 KiB).
// A draw of slot `off` for replicate rr is one ds_add_u32 of (1 << 8*(rr&3)) at byte
// off*16 + (rr & ~3): one shift-add of VALU work per draw.  The contraction reads
// one byte per step (ds_read_u8, immediate offset): no VALU unpacking.
// On gfx950 the FP64 MFMA shares the VALU datapath (tools/mfma_f64_peak4.hip:
// every VALU instruction costs 4.5-8 cycles of matrix-pipe time), so the design
// rule of this kernel is: as few VALU instructions per MFMA as possible.
constexpr int RS_QUARTER = SM_T / 4;                  // 256 samples per lane-group per tile
constexpr int RS_TILE_BYTES = SM_T * RS_REPS;         // 16384 bytes per wave
// steps per register-prefetch group: as many as the VGPR budget (2 waves/SIMD,
// 256 VGPRs) allows next to the K*NBLK accumulator tiles.
constexpr int rs_group(int K, int NBLK, bool weighted, bool explicit_, int pack = 1) {
  const int per_step = 2 + 2 * NBLK + (weighted ? 2 : 0) + (explicit_ ? 10 : 0);
  for (int g = 8; g > 2; g >>= 1)
#ifndef TXM_EXP_BUDGET
#define TXM_EXP_BUDGET 160
#endif
    if ((K + pack - 1) / pack * NBLK * 8 + 2 * g * per_step <= TXM_EXP_BUDGET) return g;
  return 2;
}

struct ResampleArgs {
  const double *x;
  int64_t ldx_s;
  const double *u;
  const double *w;
  int64_t N;
  int64_t C;
  int64_t nrep;
  const int64_t *freq;      // parity mode
  const uint32_t *counts;   // scale mode: [nrep][ntiles]
  uint32_t k0, k1;          // philox key
  uint32_t rep_base;        // replicate r of the call draws stream replicate rep_base + r (txm_sampler_spec.rep0)
  int64_t ntiles;
  uint32_t last_tile_size;
  const double *pivot;      // [1 + C]
  double *part_x;           // [n_chunks][nrep_pad][C_pad][K]
  double *part_u;           // [n_chunks][nrep_pad][K]
  int n_chunks;             // multiple of 8
  int n_rbg;                // replicate-block groups (4 blocks of 16 reps each)
  int64_t tiles_per_chunk;
  int64_t nrep_pad;         // n_rbg * 64
  int64_t C_pad;            // col groups * NBLK * 16
  // listed mode (precision-guard fallback of the int8 path): chunk c contracts the runs list[c], list[c + n_chunks], ...
  // of sub_tiles tiles each instead of its contiguous share; n_list[0] = number of runs (0: the launch is a no-op)
  const uint32_t *list;
  const uint32_t *n_list;
  int sub_tiles;
  int64_t col_off;          // x already points at the column group; its pivots start at pivot[1 + col_off]
  // batched mode (txm_resample_vals_batched): blockIdx.z = state s of S state points of one shape.  x/u/w come
  // from batch[s]; pivot, the partial sums, counts / freq rows and the sampler's replicate ids are those of the
  // single-state layout shifted by s (replicate s * nrep + r of one sampler over S * nrep replicates)
  const txm_state_ptrs *batch;
  // listed mode of a BATCHED int8 call: the fallback runs of state blockIdx.z, operands from the int8 path's state table
  const I8State *i8states = nullptr;
  // L2-sharing hint (nullptr: off).  The waves that contract one sample chunk -- 4 per workgroup x the replicate
  // groups of the chunk, placed on one XCD by the block map -- read the same samples; left alone they drift apart
  // by many tiles and every one of them streams the chunk from HBM again (measured: 19x the algorithmic bytes).
  // Each wave publishes the number of tiles it has finished in progress[group][slot] and, before the next tile,
  // sleeps (bounded) while it is more than RS_LEAD tiles ahead of the slowest STARTED, UNFINISHED wave of its
  // group.  Purely a performance hint: every wait is bounded and no result depends on it.
  uint32_t *progress;       // [groups = n_chunks * colgroups (* S)][64], zeroed by the launcher
};
constexpr uint32_t RS_LEAD = 1;          // tiles a wave may run ahead of its group
constexpr int RS_THROTTLE_SPINS = 48;    // x ~0.3 us: bound of one wait

// One wave: 16 replicates x (NBLK*16 columns) x one chunk of tiles.
//
// MFMA operand roles for step s of a tile (lane = (row = lane & 15, kk = lane >> 4)):
//   sample    i  = tile_base + kk * 256 + s          (each kk owns a contiguous quarter)
//   A (rep = row, k = kk)   = f[rep][i] * w_i * du_i^j
//   B (k = kk,  col = row)  = x[i][col] - px[col]
// A lane therefore walks CONTIGUOUS samples: u/w/freq come as 16-byte loads of
// 8 consecutive steps, f as one 8-byte LDS read per 8 steps, and the loads of
// group g+1 are issued before the 8*K*NBLK MFMAs of group g (register double
// buffer), so HBM/L2 latency hides under the matrix pipe.
// Stage 3 of the sampler for a FULL tile and the 16 replicates of one wave, with the
// minimum of VALU work (the stream itself is the one oracle/philox_oracle.c defines:
// draw d of (r, t) = 10-bit field d % 12 of Philox call d / 12 -- which lane runs a
// call is free).  Replicates are taken in pairs: calls 0..63 of each go to one wave
// iteration each, and their remaining calls 64..95 (n - 768 draws, ~1/3 of a wave)
// share a third iteration, 32 lanes per replicate.  Fields past the replicate's
// draw count add 0 instead of being branched around.
template <bool ALL_VALID>
__device__ __forceinline__ void tile_calls(uint32_t *tw, uint32_t k0, uint32_t k1, uint32_t r, uint32_t t,
                                           uint32_t c, uint32_t n, uint32_t inc, uint32_t wsel) {
  const uint32_t first = c * 12u;
  if (!ALL_VALID && first >= n) return;  // ALL_VALID: 12 (c + 1) <= n by construction, no branch
  // call index in the second counter word: the tile's and the replicate's share of rounds 1-3 is wave-uniform
  const Philox4 o = philox4x32_10(t, c, r, 3u, k0, k1);
  const uint32_t nd = n - first;
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
    const uint32_t word = o.w[wi];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t f = (word >> (10 * k)) & 1023u;
      const uint32_t add = ALL_VALID ? inc : (((uint32_t)(wi * 3 + k) < nd) ? inc : 0u);
      atomicAdd(&tw[f * (RS_REPS / 4) + wsel], add);
    }
  }
}

__device__ __forceinline__ void fill_tile_full(const ResampleArgs &a, const uint32_t *counts, uint32_t rid0,
                                               uint32_t *tw, int64_t rep0, int64_t t, int lane) {
#pragma unroll 1
  for (int p = 0; p < RS_REPS / 2; ++p) {
    const int64_t ra = rep0 + 2 * p, rb = ra + 1;
    if (ra >= a.nrep) break;  // wave-uniform
    const uint32_t na = counts[(size_t)ra * a.ntiles + t];
    const uint32_t nb = rb < a.nrep ? counts[(size_t)rb * a.ntiles + t] : 0u;
    const uint32_t wsel = (uint32_t)p >> 1;                  // (2p) >> 2 == (2p+1) >> 2
    const uint32_t inc_a = 1u << (8u * ((2u * p) & 3u)), inc_b = inc_a << 8;
    // iterations 1 and 2: calls 0..63 of each replicate; when n >= 768 (wave-uniform,
    // practically always) every field is a real draw and no select is needed
    if (na >= 768u) tile_calls<true>(tw, a.k0, a.k1, rid0 + (uint32_t)ra, (uint32_t)t, (uint32_t)lane, na, inc_a, wsel);
    else tile_calls<false>(tw, a.k0, a.k1, rid0 + (uint32_t)ra, (uint32_t)t, (uint32_t)lane, na, inc_a, wsel);
    if (nb >= 768u) tile_calls<true>(tw, a.k0, a.k1, rid0 + (uint32_t)rb, (uint32_t)t, (uint32_t)lane, nb, inc_b, wsel);
    else tile_calls<false>(tw, a.k0, a.k1, rid0 + (uint32_t)rb, (uint32_t)t, (uint32_t)lane, nb, inc_b, wsel);
    // iteration 3: calls 64..95 of both, 32 lanes each
    const bool hb = lane >= 32;
    tile_calls<false>(tw, a.k0, a.k1, rid0 + (uint32_t)(hb ? rb : ra), (uint32_t)t, 64u + ((uint32_t)lane & 31u),
                      hb ? nb : na, hb ? inc_b : inc_a, wsel);
    // calls >= 96 (n > 1152, a > 4 sigma event): plain loop
    const uint32_t nmax = na > nb ? na : nb;
    for (uint32_t c0 = 96u; c0 * 12u < nmax; c0 += 64u) {
      tile_calls<false>(tw, a.k0, a.k1, rid0 + (uint32_t)ra, (uint32_t)t, c0 + (uint32_t)lane, na, inc_a, wsel);
      tile_calls<false>(tw, a.k0, a.k1, rid0 + (uint32_t)rb, (uint32_t)t, c0 + (uint32_t)lane, nb, inc_b, wsel);
    }
  }
}

//
// Tiles are always contracted over a full 1024-sample window.  The last
// (partial) tile slides its window back to [N - 1024, N) and gives the samples
// that belong to the previous tile a zero count, so the hot loop has no
// bounds handling at all.  Only data sets shorter than one tile (SMALLN) use
// clamped addresses, in a separate instantiation.
// MODE: RS_PLAIN = one state point, contiguous chunk of tiles; RS_LISTED = the runs of the fallback list;
// RS_BATCHED = state blockIdx.z of a batch.  Compile-time so that the plain kernel carries none of the other modes'
// loop state (scalar registers are what this kernel runs out of first: spilled SGPRs cost vector instructions,
// and every vector instruction here costs FP64-MFMA time).
constexpr int RS_PLAIN = 0, RS_LISTED = 1, RS_BATCHED = 2;
//
// PACK (1, 2 or 4; NBLK == 1): with C <= 16 / PACK value columns the 16 B-operand columns carry PACK powers of du
// per column, B[k][jp * CPK + c] = dx_c * du^jp, and the A operands step by du^PACK: ceil(K / PACK) MFMAs per
// k-step instead of K, so a narrow state does not pay for 16 columns it does not have.
template <int K, int NBLK, bool WEIGHTED, bool EXPLICIT, bool SMALLN, int MODE = RS_PLAIN, int PACK = 1>
__global__ __launch_bounds__(RS_BLOCK, 2) void resample_kernel(const ResampleArgs a) {
  static_assert(PACK == 1 || (NBLK == 1 && (PACK == 2 || PACK == 4)), "power packing needs a single column block");
  constexpr int CPK = 16 / PACK;            // value columns per power slot
  constexpr int KJ = (K + PACK - 1) / PACK;  // accumulator tiles (MFMAs per k-step and column block)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // readfirstlane: tell the compiler the wave index is wave-uniform, so everything
  // derived from it (replicate base, LDS tile base, loop bounds) lives in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int row = lane & 15;   // A: replicate within block;  B: column within block
  const int kk = lane >> 4;    // which quarter of the tile this lane group walks
  const int crow = PACK == 1 ? row : row % CPK;  // B: value column within the block
  const int jp = PACK == 1 ? 0 : row / CPK;      // B: power of du folded into this lane's column

  // XCD-aware task map: workgroups that share a sample chunk sit on one XCD
  // (blocks b and b+8 share an XCD) and are dispatched back to back, so the
  // chunk is streamed from HBM once and re-read from that XCD's L2.
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int chunk = (q / a.n_rbg) * 8 + xcd;
  const int rbg = q % a.n_rbg;
  const int colgrp = blockIdx.y;
  const int64_t rep0 = ((int64_t)rbg * RS_WAVES + wave) * RS_REPS;
  const int64_t col0 = (int64_t)colgrp * (NBLK * 16);

  const int64_t t_begin = (int64_t)chunk * a.tiles_per_chunk;
  int64_t t_end = t_begin + a.tiles_per_chunk;
  if (t_end > a.ntiles) t_end = a.ntiles;
  // operands of this launch's state (batched mode: state blockIdx.z)
  const double *X = a.x, *U = a.u, *W = a.w, *PIV = a.pivot;
  const uint32_t *CNT = a.counts;
  const int64_t *FREQ = a.freq;
  double *PX = a.part_x, *PU = a.part_u;
  uint32_t rid0 = a.rep_base;
  if constexpr (MODE == RS_BATCHED) {
    const int64_t sidx = blockIdx.z;
    const txm_state_ptrs bs = a.batch[sidx];
    X = bs.x;
    U = bs.u;
    W = bs.w;
    PIV += sidx * (1 + a.C);
    if (CNT) CNT += (size_t)sidx * a.nrep * a.ntiles;
    if (FREQ) FREQ += (size_t)sidx * a.nrep * a.N;
    PX += (size_t)sidx * a.n_chunks * a.nrep_pad * a.C_pad * K;
    PU += (size_t)sidx * a.n_chunks * a.nrep_pad * K;
    rid0 += (uint32_t)(sidx * a.nrep);
  }
  constexpr bool listed = MODE == RS_LISTED;
  int64_t nruns = 1;
  const uint32_t *LIST = a.list, *NLIST = a.n_list;
  if constexpr (listed) {
    if (a.i8states != nullptr) {  // batched int8 call: state blockIdx.z
      const I8State e = a.i8states[blockIdx.z];
      X = e.x + a.col_off;  // (a.x of a listed launch is x + the column group's offset)
      U = e.u;
      W = e.w;
      PIV = e.pivot;
      CNT = e.counts;
      PX = e.fb_x;
      PU = e.fb_u;
      rid0 = e.rep_base;
      LIST = e.list;
      NLIST = e.n_list;
    }
    nruns = (int64_t)*NLIST;
    if (nruns == 0) return;  // nothing flagged: the finalize kernel does not read this launch's partial sums
  }

  const double pu = PIV[0];
  // Columns beyond C (padding of the last 16-column block) re-read column 0:
  // their sums are finite garbage that the finalize kernel never looks at, which
  // is cheaper than a mask multiply on the FP64 pipe the MFMAs need.
  double px[NBLK];
  int64_t ccol[NBLK];
#pragma unroll
  for (int bl = 0; bl < NBLK; ++bl) {
    const int64_t c = col0 + bl * 16 + crow;
    ccol[bl] = c < a.C ? c : 0;
    px[bl] = PIV[1 + a.col_off + ccol[bl]];
  }

  v4f64 acc[KJ][NBLK];
  double usum[K];
#pragma unroll
  for (int j = 0; j < K; ++j) usum[j] = 0.0;
#pragma unroll
  for (int j = 0; j < KJ; ++j)
#pragma unroll
    for (int bl = 0; bl < NBLK; ++bl) acc[j][bl] = (v4f64){0.0, 0.0, 0.0, 0.0};

  unsigned char *tile = lds_raw + (size_t)wave * RS_TILE_BYTES;
  const int64_t my_rep = rep0 + row;
  const bool rep_ok = my_rep < a.nrep;
  const int64_t last = a.N - 1;

  // register double buffer for one group of RS_GROUP steps
  constexpr int RS_GROUP = SMALLN ? 2 : rs_group(K, NBLK, WEIGHTED, EXPLICIT, PACK);
  struct Grp {
    double u[RS_GROUP];
    double w[RS_GROUP];
    double x[RS_GROUP][NBLK];
    uint32_t fb[RS_GROUP];    // scale mode: u8 counts straight from LDS, one per step
    int64_t fi[EXPLICIT ? RS_GROUP : 1];  // parity mode: raw int64 counts
  };

  uint32_t *pg = nullptr;
  uint32_t tiles_done = 0;
  const int pslot = (rbg * RS_WAVES + wave) & 63;
  if constexpr (MODE == RS_PLAIN && !SMALLN)
    if (a.progress != nullptr) pg = a.progress + ((size_t)(colgrp * a.n_chunks + chunk)) * 64;
  double cnt = 0.0;  // lanes kk == 0, unweighted: the replicate's draws in the contracted tiles (exact integers)
  for (int64_t run = listed ? chunk : 0; run < nruns; run += listed ? a.n_chunks : 1) {
  int64_t tb = t_begin, te = t_end;
  if constexpr (listed) {
    tb = (int64_t)LIST[run];
    te = tb + a.sub_tiles;
    if (te > a.ntiles) te = a.ntiles;
  }
  for (int64_t t = tb; t < te; ++t) {
    const int64_t i_tile = t * SM_T;
    if constexpr (!WEIGHTED && !EXPLICIT)
      if (kk == 0 && rep_ok) cnt += (double)CNT[(size_t)my_rep * a.ntiles + t];
    const uint32_t tsize = (t == a.ntiles - 1) ? a.last_tile_size : (uint32_t)SM_T;
    // window of 1024 samples that is contracted for this tile
    int64_t wbase = i_tile;
    if constexpr (!SMALLN)
      if (wbase > a.N - SM_T) wbase = a.N - SM_T;
    const uint32_t shift = (uint32_t)(i_tile - wbase);  // window slots owned by the previous tile

    if constexpr (!EXPLICIT) {
      // ---- stage 3 of the sampler: fill this wave's private f tile ---------
      // (wave-private LDS: DS ops of one wave execute in order, so only the
      // compiler needs fencing -- no s_barrier, waves of a workgroup drift
      // apart and their VALU/LDS phases overlap other waves' MFMA phases)
      uint4 *z = reinterpret_cast<uint4 *>(tile);
#pragma unroll
      for (int e = 0; e < RS_TILE_BYTES / 16 / 64; ++e) z[e * 64 + lane] = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      uint32_t *tw = reinterpret_cast<uint32_t *>(tile);
      if (tsize == (uint32_t)SM_T) {
        fill_tile_full(a, CNT, rid0, tw, rep0, t, lane);
      } else {
        for (int rr = 0; rr < RS_REPS; ++rr) {
          const int64_t r = rep0 + rr;
          if (r >= a.nrep) break;  // wave-uniform
          const uint32_t n = CNT[(size_t)r * a.ntiles + t];
          sampler_fine_tile(a.k0, a.k1, rid0 + (uint32_t)r, (uint32_t)t, n, tsize, lane, [&](uint32_t off0) {
            const uint32_t off = off0 + shift;
            atomicAdd(&tw[off * (RS_REPS / 4) + ((uint32_t)rr >> 2)], 1u << (8u * ((uint32_t)rr & 3u)));
          });
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }

    const unsigned char *frow = tile + (kk * RS_QUARTER) * RS_REPS + row;  // + s * 16 per step
    const int64_t ibase = wbase + (int64_t)kk * RS_QUARTER;
    const uint32_t lbase = (uint32_t)kk * RS_QUARTER;  // window slot of this lane's step 0
    // unsigned 32-bit BYTE offsets: base (SGPR pair) + zext(VGPR) is the saddr form of global_load
    const uint32_t lane_uoff = (uint32_t)(kk * RS_QUARTER) * 8u;
    uint32_t lane_xoff[NBLK];
#pragma unroll
    for (int bl = 0; bl < NBLK; ++bl) lane_xoff[bl] = (uint32_t)((kk * RS_QUARTER * a.ldx_s + ccol[bl]) * 8);
    auto ld = [](const void *base, uint32_t byteoff) {
      return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byteoff);
    };

    // One-step operand lookahead: the A operands (f * w * du^j) and B operands
    // (x - px) of step e+1 are computed into a second register set BEFORE the
    // K*NBLK MFMAs of step e are issued, so the MFMAs of a step go out back to back
    // and no VALU write targets a register an in-flight MFMA still reads.
    struct Ops {
      double a[KJ];
      double b[NBLK];
    };
    auto prep_step = [&](const Grp &G, int e, Ops &o) {
      double av;
      if constexpr (EXPLICIT) av = (double)G.fi[e];
      else av = (double)G.fb[e];
      if constexpr (WEIGHTED) av *= G.w[e];
      const double du = G.u[e] - pu;
#pragma unroll
      for (int bl = 0; bl < NBLK; ++bl) o.b[bl] = G.x[e][bl] - px[bl];
      if constexpr (PACK > 1) {
        double f = (jp & 1) ? du : 1.0;
        if constexpr (PACK == 4) f *= (jp & 2) ? du * du : 1.0;
        o.b[0] *= f;
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (j % PACK == 0) o.a[j / PACK] = av;
        // S0[0] = sum of counts: accumulated only when weights make it non-trivial
        // (unweighted it is the replicate's draw count, summed from `counts` below)
        if (j > 0 || WEIGHTED || EXPLICIT) usum[j] += av;
        av *= du;
      }
    };
    auto mfma_step = [&](const Ops &o) {
#pragma unroll
      for (int j = 0; j < KJ; ++j)
#pragma unroll
        for (int bl = 0; bl < NBLK; ++bl)
          acc[j][bl] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[j], o.b[bl], acc[j][bl], 0, 0, 0);
    };
    auto compute_group = [&](const Grp &G) {
      Ops P, Q;
      prep_step(G, 0, P);
#pragma unroll
      for (int e = 0; e < RS_GROUP; e += 2) {
        if (e + 1 < RS_GROUP) prep_step(G, e + 1, Q);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(P);
        __builtin_amdgcn_sched_barrier(0);
        if (e + 1 < RS_GROUP) {
          if (e + 2 < RS_GROUP) prep_step(G, e + 2, P);
          __builtin_amdgcn_sched_barrier(0);
          mfma_step(Q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    // Branch-free on purpose: a conditional around the prefetch loads makes the
    // compiler's s_waitcnt insertion fall back to vmcnt(0) at the join, which
    // serialises loads and MFMAs.
    auto load_group = [&](int g, Grp &G) {
      const int s0 = g * RS_GROUP;
      if constexpr (!SMALLN) {
        // wave-uniform row base (SGPRs, scalar ALU) + a loop-invariant 32-bit lane
        // offset: the loads need no per-step vector address arithmetic
        const double *ur = U + (wbase + s0);
#pragma unroll
        for (int e = 0; e < RS_GROUP; ++e) G.u[e] = ld(ur + e, lane_uoff);
        if constexpr (WEIGHTED) {
          const double *wr = W + (wbase + s0);
#pragma unroll
          for (int e = 0; e < RS_GROUP; ++e) G.w[e] = ld(wr + e, lane_uoff);
        }
#pragma unroll
        for (int e = 0; e < RS_GROUP; ++e) {
          const double *xr = X + (wbase + s0 + e) * a.ldx_s;
#pragma unroll
          for (int bl = 0; bl < NBLK; ++bl) G.x[e][bl] = ld(xr, lane_xoff[bl]);
        }
        if constexpr (EXPLICIT) {
          const int64_t *fp = FREQ + (size_t)(rep_ok ? my_rep : 0) * a.N + ibase + s0;
#pragma unroll
          for (int e = 0; e < RS_GROUP; ++e)
            G.fi[e] = (rep_ok && lbase + (uint32_t)(s0 + e) >= shift) ? fp[e] : 0;
        }
      } else {
        // N < 1024: a single short tile; clamp addresses, zero the missing samples
#pragma unroll
        for (int e = 0; e < RS_GROUP; ++e) {
          const int64_t ii = ibase + s0 + e;
          const int64_t ic = ii <= last ? ii : last;
          G.u[e] = U[ic];
          if constexpr (WEIGHTED) G.w[e] = W[ic];
#pragma unroll
          for (int bl = 0; bl < NBLK; ++bl) G.x[e][bl] = X[ic * a.ldx_s + ccol[bl]];
          if constexpr (EXPLICIT)
            G.fi[e] = (rep_ok && ii <= last) ? FREQ[(size_t)(rep_ok ? my_rep : 0) * a.N + ic] : 0;
        }
      }
      if constexpr (!EXPLICIT) {
#pragma unroll
        for (int e = 0; e < RS_GROUP; ++e) G.fb[e] = frow[(s0 + e) * RS_REPS];
      }
    };
    // groups per lane quarter (even); SMALLN: only those that hold samples
    int ngroups = RS_QUARTER / RS_GROUP;
    if constexpr (SMALLN) {
      const uint32_t q0 = tsize < (uint32_t)RS_QUARTER ? tsize : (uint32_t)RS_QUARTER;
      ngroups = (int)((q0 + RS_GROUP - 1) / RS_GROUP);
      ngroups = (ngroups + 1) & ~1;
    }
    {
      Grp A, B;
      load_group(0, A);
      // sched_barrier: keep "issue the next group's loads, then run this group's
      // MFMAs" in program order (the machine scheduler otherwise sinks the loads
      // next to their uses to save registers and exposes their latency).
      for (int g = 0; g < ngroups; g += 2) {
        load_group(g + 1, B);  // g + 1 < ngroups (even count)
        __builtin_amdgcn_sched_barrier(0);
        compute_group(A);
        __builtin_amdgcn_sched_barrier(0);
        const int g2 = (g + 2 < ngroups) ? g + 2 : g + 1;  // tail: harmless re-load
        load_group(g2, A);
        __builtin_amdgcn_sched_barrier(0);
        compute_group(B);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    if constexpr (!EXPLICIT) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    if (pg != nullptr) {  // kernel argument: uniform
      ++tiles_done;
      if (lane == 0) __hip_atomic_store(&pg[pslot], tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
      for (int spin = 0; spin < RS_THROTTLE_SPINS; ++spin) {
        uint32_t v = __hip_atomic_load(&pg[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0u) v = 0xffffffffu;  // slot unused or its wave not started yet: not a wave to wait for
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t w2 = (uint32_t)__shfl_xor((int)v, o);
          v = w2 < v ? w2 : v;
        }
        if (tiles_done <= v + RS_LEAD) break;
        __builtin_amdgcn_s_sleep(32);
      }
    }
  }
  }
  if (pg != nullptr && lane == 0)  // finished: never hold the others back
    __hip_atomic_store(&pg[pslot], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- write partial sums ---------------------------------------------------
  // D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *px_out = PX + ((size_t)chunk * a.nrep_pad + rep0) * a.C_pad * K;
#pragma unroll
  for (int jj = 0; jj < KJ; ++jj)
#pragma unroll
    for (int bl = 0; bl < NBLK; ++bl)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rrow = kk + 4 * rg;
        const int64_t c = col0 + bl * 16 + crow;
        const int j = jj * PACK + jp;
        if (PACK == 1 || j < K) px_out[((size_t)rrow * a.C_pad + c) * K + j] = acc[jj][bl][rg];
      }
  if (colgrp == 0) {
    if constexpr (!WEIGHTED && !EXPLICIT) usum[0] = cnt;  // other lane groups contribute 0
#pragma unroll
    for (int j = 0; j < K; ++j) {
      double v = usum[j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kk == 0) PU[((size_t)chunk * a.nrep_pad + rep0 + row) * K + j] = v;
    }
  }
}

// finalize: thread per (rep, col): fixed-order sum over chunks, shift to the
// cmomy state, out[rep][c][2][K].
template <int K>
__global__ __launch_bounds__(256) void resample_finalize_kernel(
    const double *__restrict__ part_x, const double *__restrict__ part_u, int n_chunks,
    int64_t nrep_pad, int64_t C_pad, int64_t nrep, int64_t C, const double *__restrict__ pivot,
    double *__restrict__ out, int64_t c_off = 0, int64_t C_total = 0) {
  // C columns of this launch are the columns c_off .. c_off + C - 1 of the C_total output columns.
  // Thread = (output e of the block's 32, chunk segment seg of 8): segment seg adds the chunks seg, seg + 8, ... in
  // ascending order, then the eight segments are added in order -- a fixed tree (one thread per output walking all
  // chunks was a chain of dependent-latency loads: 0.3 ms for 400 outputs x 1024 chunks).
  constexpr int NSEG = 8, NOUT = 256 / NSEG;
  __shared__ double sh[NSEG][NOUT][2 * K];
  const int eo = threadIdx.x % NOUT, seg = threadIdx.x / NOUT;
  const int64_t e = (int64_t)blockIdx.x * NOUT + eo;
  const bool live = e < nrep * C;
  const int64_t r = live ? e / C : 0, c = live ? e % C : 0;
  if (C_total == 0) C_total = C;
  // batched mode: blockIdx.y = state; every per-state array follows the previous state's
  const int64_t sidx = blockIdx.y;
  part_x += (size_t)sidx * n_chunks * nrep_pad * C_pad * K;
  part_u += (size_t)sidx * n_chunks * nrep_pad * K;
  pivot += sidx * (1 + C);
  out += (size_t)sidx * nrep * C_total * 2 * K;
  double S0[K], S1[K];
#pragma unroll
  for (int j = 0; j < K; ++j) S0[j] = S1[j] = 0.0;
  if (live)
    for (int ch = seg; ch < n_chunks; ch += NSEG) {
      const double *pu_ = part_u + ((size_t)ch * nrep_pad + r) * K;
      const double *px_ = part_x + (((size_t)ch * nrep_pad + r) * C_pad + c) * K;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        S0[j] += pu_[j];
        S1[j] += px_[j];
      }
    }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    sh[seg][eo][j] = S0[j];
    sh[seg][eo][K + j] = S1[j];
  }
  __syncthreads();
  if (seg != 0 || !live) return;
  for (int g = 1; g < NSEG; ++g)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      S0[j] += sh[g][eo][j];
      S1[j] += sh[g][eo][K + j];
    }
  double st[2 * K];
  pivot_sums_to_state<K>(S0, S1, pivot[0], pivot[1 + c_off + c], st);
  double *o = out + (r * C_total + c_off + c) * 2 * K;
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) o[q] = st[q];
}

// finalize of the int8 path: one slot of partial sums per scaling window, [window][replicate][power][8 digit slots]
// [column] (u-row: [window][replicate][power][8]); windows in ascending order, digits in ascending order -- a fixed order
// that does not depend on the launch geometry.  Windows the precision guard flagged hold nothing: they were contracted
// by the FP64 kernel, whose sums are added behind (same pivot).
template <int K, int CPAD>
__global__ __launch_bounds__(256) void resample_finalize_i8_kernel(
    const double *__restrict__ part_x, const double *__restrict__ part_u, int64_t nwin,
    const uint32_t *__restrict__ wflag, int64_t nrep_pad, int64_t nrep, int64_t C,
    const double *__restrict__ pivot, double *__restrict__ out, int64_t c_off, int64_t C_total,
    const double *__restrict__ fb_x, const double *__restrict__ fb_u, int fb_chunks, int64_t fb_cpad,
    const uint32_t *__restrict__ n_list, const I8State *__restrict__ states = nullptr, const int summed = 0) {
  // summed != 0: the x slots hold the digit sums already -- [window][replicate][power][column] -- written by the narrow kernels and
  // resample_i8g_kernel with the expression below; summed == 1: the u slots too ([window][replicate][power])
  if (states != nullptr) {  // batched int8 call: state blockIdx.y
    const I8State e = states[blockIdx.y];
    part_x = e.part_x; part_u = e.part_u; wflag = e.wflag; pivot = e.pivot; out = e.out;
    fb_x = e.fb_x; fb_u = e.fb_u; n_list = e.n_list;
  }
  // CPAD = columns of a row of part_x (32; 4, 8 or 16 where the narrow-state kernel wrote it)
  constexpr int cpad = CPAD;
  // one workgroup per replicate: thread = (column c < cpad, window segment seg of 256 / cpad).  Segment seg adds the
  // windows seg, seg + nseg, ... in ascending order, then the segments are added in order: a fixed tree that depends on
  // the number of windows (i.e. on N) and on the state's width only, not on the launch geometry.  (One thread per output
  // walking all windows was latency-bound on short series with few outputs: 1600 threads x 611 windows at BASELINE
  // config 2; so were 8 segments for an 8-column state.)
  __shared__ double sh[256][2 * K];
  constexpr int nseg = 256 / cpad;
  const int64_t r = blockIdx.x;
  const int c = threadIdx.x % cpad, seg = threadIdx.x / cpad;
  double S0[K], S1[K];
#pragma unroll
  for (int j = 0; j < K; ++j) S0[j] = S1[j] = 0.0;
  if (c < C) {
    for (int64_t w = seg; w < nwin; w += nseg) {
      if (wflag[w] != 0u) continue;
      if (summed) {  // (uniform) 1: x and u slots digit-summed; 2: x summed, u per digit (resample_i8g_kernel: its u-row digits ride
        // in other waves' columns)
        const double *px_ = part_x + ((size_t)w * nrep_pad + r) * K * cpad + c;
        if (summed == 1) {
          const double *pu_ = part_u + ((size_t)w * nrep_pad + r) * K;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            S0[j] += pu_[j];
            S1[j] += px_[j * cpad];
          }
        } else {
          const double *pu_ = part_u + ((size_t)w * nrep_pad + r) * K * 8;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const double4 ua = *reinterpret_cast<const double4 *>(pu_ + j * 8), ub = *reinterpret_cast<const double4 *>(pu_ + j * 8 + 4);
            S0[j] += ((((((ua.x + ua.y) + ua.z) + ua.w) + ub.x) + ub.y) + ub.z);  // digit slots 0..6, ascending
            S1[j] += px_[j * cpad];
          }
        }
        continue;
      }
      const double *pu_ = part_u + ((size_t)w * nrep_pad + r) * K * 8;
      const double *px_ = part_x + ((size_t)w * nrep_pad + r) * K * 8 * cpad + c;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const double4 ua = *reinterpret_cast<const double4 *>(pu_ + j * 8), ub = *reinterpret_cast<const double4 *>(pu_ + j * 8 + 4);
        const double *q = px_ + (size_t)j * 8 * cpad;
        S0[j] += ((((((ua.x + ua.y) + ua.z) + ua.w) + ub.x) + ub.y) + ub.z);  // digit slots 0..6, ascending
        S1[j] += ((((((q[0] + q[cpad]) + q[2 * cpad]) + q[3 * cpad]) + q[4 * cpad]) + q[5 * cpad]) + q[6 * cpad]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    sh[threadIdx.x][j] = S0[j];
    sh[threadIdx.x][K + j] = S1[j];
  }
  __syncthreads();
  if (seg != 0 || c >= C) return;
  for (int g = 1; g < nseg; ++g)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      S0[j] += sh[g * cpad + c][j];
      S1[j] += sh[g * cpad + c][K + j];
    }
  // windows the precision guard handed to the FP64 kernel (same pivot: the sums simply add)
  if (n_list[0] != 0u)
    for (int ch = 0; ch < fb_chunks; ++ch) {
      const double *pu_ = fb_u + ((size_t)ch * nrep_pad + r) * K;
      const double *px_ = fb_x + (((size_t)ch * nrep_pad + r) * fb_cpad + c) * K;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        S0[j] += pu_[j];
        S1[j] += px_[j];
      }
    }
  double st[2 * K];
  pivot_sums_to_state<K>(S0, S1, pivot[0], pivot[1 + c_off + c], st);
  double *o = out + (r * C_total + c_off + c) * 2 * K;
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) o[q] = st[q];
  (void)nrep;
}

// the second sample matrix: per-replicate weighted means  out_y[r][c] = py[c] + S1y / S0  from the per-window slots
// written by the y row set (same fixed tree as above), the sum of weights S0 from the u-row slots (power 0)
__global__ __launch_bounds__(256) void resample_finalize_y_kernel(
    const double *__restrict__ part_y, const double *__restrict__ part_u, int K, int64_t nwin,
    const uint32_t *__restrict__ wflag, int64_t nrep_pad, int64_t C, const double *__restrict__ ypivot,
    double *__restrict__ out_y, int64_t c_off, int64_t C_total, const double *__restrict__ fb_y,
    const double *__restrict__ fb_u, int fb_chunks, int64_t fb_cpad, const uint32_t *__restrict__ n_list) {
  constexpr int NSEG = 8;
  __shared__ double sh[NSEG][32][2];
  const int64_t r = blockIdx.x;
  const int c = threadIdx.x & 31, seg = threadIdx.x >> 5;
  double S0 = 0.0, S1 = 0.0;
  if (c < C) {
    for (int64_t w = seg; w < nwin; w += NSEG) {
      if (wflag[w] != 0u) continue;
      const double *pu_ = part_u + ((size_t)w * nrep_pad + r) * K * 8;
      const double *q = part_y + ((size_t)w * nrep_pad + r) * 8 * I8_CPAD + c;
      const double4 ua = *reinterpret_cast<const double4 *>(pu_), ub = *reinterpret_cast<const double4 *>(pu_ + 4);
      S0 += ((((((ua.x + ua.y) + ua.z) + ua.w) + ub.x) + ub.y) + ub.z);
      S1 += ((((((q[0] + q[I8_CPAD]) + q[2 * I8_CPAD]) + q[3 * I8_CPAD]) + q[4 * I8_CPAD]) + q[5 * I8_CPAD]) + q[6 * I8_CPAD]);
    }
  }
  sh[seg][c][0] = S0;
  sh[seg][c][1] = S1;
  __syncthreads();
  if (seg != 0 || c >= C) return;
  S0 = sh[0][c][0];
  S1 = sh[0][c][1];
  for (int g = 1; g < NSEG; ++g) {
    S0 += sh[g][c][0];
    S1 += sh[g][c][1];
  }
  if (n_list[0] != 0u)
    for (int ch = 0; ch < fb_chunks; ++ch) {
      S0 += fb_u[((size_t)ch * nrep_pad + r) * K];
      S1 += fb_y[((size_t)ch * nrep_pad + r) * fb_cpad + c];
    }
  out_y[r * C_total + c_off + c] = ypivot[1 + c_off + c] + S1 * (1.0 / S0);  // as pivot_sums_to_state forms a mean
}

struct ResamplePlan {
  int nblk, colgroups, n_rbg, n_chunks;
  int64_t tiles_per_chunk, nrep_pad, C_pad, ntiles;
  size_t off_pivot, off_px, off_pu, off_prog, prog_bytes, total;
};

// tiles per sample chunk: ntiles / div chunks, at least 8 and at most 1024 of them (rounded to octets by the callers)
static int64_t resample_chunk_tiles(int64_t ntiles, int64_t div) {
  int64_t nc = ntiles / div;
  if (nc > 1024) nc = 1024;
  if (nc < 8) nc = 8;
  if (nc > ntiles) nc = ntiles;
  nc = cdiv(nc, 8) * 8;
  return cdiv(ntiles, nc);
}

static ResamplePlan plan_resample(int64_t N, int64_t C, int64_t nrep, int K) {
  ResamplePlan p;
  p.nblk = C <= 16 ? 1 : 2;
  p.colgroups = (int)cdiv(C, p.nblk * 16);
  p.C_pad = (int64_t)p.colgroups * p.nblk * 16;
  p.n_rbg = (int)cdiv(nrep, RS_WAVES * RS_REPS);
  p.nrep_pad = (int64_t)p.n_rbg * RS_WAVES * RS_REPS;
  p.ntiles = cdiv(N, SM_T);
  // Sample chunks: their number is a function of N ALONE (ntiles / 2, between 8 and 1024).  A replicate's sums are formed
  // per chunk and the chunks are added in a fixed order by resample_finalize_kernel, so with a chunking that does not
  // look at nrep (it used to: "enough workgroups to fill the chip") the rows [a, b) of a bootstrap equal the (b - a)-
  // replicate call with rep0 = a BIT FOR BIT on this kernel too, whatever the slab sizes (multi-GPU replicate slabs).
  p.tiles_per_chunk = resample_chunk_tiles(p.ntiles, 2);
  const int64_t nc = cdiv(cdiv(p.ntiles, p.tiles_per_chunk), 8) * 8;
  p.n_chunks = (int)nc;
  p.off_pivot = 0;
  p.off_px = align_up((size_t)(1 + C) * sizeof(double), 256);
  p.off_pu = p.off_px + align_up((size_t)p.n_chunks * p.nrep_pad * p.C_pad * K * sizeof(double), 256);
  p.off_prog = p.off_pu + align_up((size_t)p.n_chunks * p.nrep_pad * K * sizeof(double), 256);
  p.prog_bytes = (size_t)p.n_chunks * p.colgroups * 64 * sizeof(uint32_t);
  p.total = p.off_prog + align_up(p.prog_bytes, 256);
  return p;
}

// int8-sliced path (txm_resample_i8.hip): 64 replicates x all columns per workgroup,
// chunks of whole scaling windows, one workgroup per CU.
struct I8Plan {
  int n_rbg, n_chunks, ngroups;
  int64_t tiles_per_chunk, nrep_pad, ntiles, nwin, win_tiles;
  size_t off_px, off_pu, total;
  // precision-guard fallback: the FP64 kernel's plan / partial sums for one column group
  int sub_tiles;
  size_t off_fbx, off_fbu, off_prog, prog_bytes, off_stats, off_prep;
  // the per-sample count table of the call's replicates (txm_count_table.hip; wide states: the contraction kernel without a
  // sampler inside, txm_resample_i8g.hip) -- 0 bytes for shapes that never take that kernel
  size_t off_table, table_bytes, off_gprog, total_table;  // (+ sixteen progress words per window: the L2-sharing hint of that kernel)
  // second sample matrix (txm_resample_opts.y) carried by the int8 kernel: its per-window partial sums, the FP64
  // fallback's sums for it, the sums themselves [nrep][C] (2 doubles each) and its pre-pass tables inside the prep block
  size_t off_py, off_fby, prep_ypiv, prep_ywt, prep_yflag;
  // the pre-pass block ("prep": what depends on the data and the shape, not on the sampler):
  //   [pivot (1 + C)] then per 32-column group [window table | guard flags | fallback run list | n_list (256 B)]
  size_t prep_wt, prep_flag, prep_list, prep_nlist, prep_group0, prep_group_stride, prep_total;
  ResamplePlan fb;
};

static I8Plan plan_i8(int64_t N, int64_t C, int64_t nrep, int K) {
  I8Plan p;
  p.n_rbg = (int)cdiv(nrep, I8_REPS);
  p.nrep_pad = (int64_t)p.n_rbg * I8_REPS;
  p.ntiles = cdiv(N, SM_T);
  p.ngroups = (int)cdiv(C, I8_CPAD);
  // scaling window = the slot of the partial sums: 256 tiles (262144 samples) on long series, shorter on short ones.
  // A function of N ALONE: the window decides the fixed-point scale of every monomial and the order of the final
  // summation, so a replicate's bits must not depend on nrep / the chunking (txm_sampler_spec.rep0: replicate slabs of
  // a multi-GPU run equal the one-GPU rows bit for bit).
  p.win_tiles = I8_WIN_TILES;
#ifndef TXM_WIN_MIN  // (A/B builds: fewer, longer windows on short series)
#define TXM_WIN_MIN 256
#endif
  while (p.win_tiles > 4 && p.ntiles < TXM_WIN_MIN * p.win_tiles) p.win_tiles /= 4;  // >= 256 windows where N allows
  p.nwin = cdiv(p.ntiles, p.win_tiles);
  // one workgroup per CU: chunks (whole windows) x replicate groups should fill the CUs once, not 1.1 times
  int64_t nc = (int64_t)num_cus() / p.n_rbg / 8 * 8;
  if (nc < 8) nc = 8;
  if (nc > p.nwin) nc = cdiv(p.nwin, 8) * 8;
  p.tiles_per_chunk = cdiv(p.nwin, nc) * p.win_tiles;
  p.n_chunks = (int)(cdiv(cdiv(p.ntiles, p.tiles_per_chunk), 8) * 8);
  p.sub_tiles = p.win_tiles < I8_SUB_TILES ? (int)p.win_tiles : I8_SUB_TILES;
  // prep block
  p.prep_wt = 0;
  p.prep_flag = p.prep_wt + align_up((size_t)p.nwin * I8_WT_STRIDE * sizeof(double) + 2048, 256);  // + timing slots of debug builds
  p.prep_list = p.prep_flag + align_up((size_t)p.nwin * sizeof(uint32_t), 256);
  p.prep_nlist = p.prep_list + align_up((size_t)p.nwin * (size_t)(p.win_tiles / p.sub_tiles) * sizeof(uint32_t), 256);
  p.prep_ywt = p.prep_nlist + 256;
  p.prep_yflag = p.prep_ywt + align_up((size_t)p.nwin * I8_WT_STRIDE * sizeof(double), 256);
  p.prep_group_stride = p.prep_yflag + align_up((size_t)p.nwin * sizeof(uint32_t), 256);
  p.prep_ypiv = align_up((size_t)(1 + C) * sizeof(double), 256);
  p.prep_group0 = 2 * p.prep_ypiv;
  p.prep_total = p.prep_group0 + (size_t)p.ngroups * p.prep_group_stride;
  // workspace
  p.off_px = 0;
  p.off_pu = p.off_px + align_up((size_t)p.nwin * p.nrep_pad * K * I8_CPAD * 8 * sizeof(double), 256);
  p.fb = plan_resample(N, C < I8_CPAD ? C : I8_CPAD, nrep, K);
  p.off_fbx = p.off_pu + align_up((size_t)p.nwin * p.nrep_pad * K * 8 * sizeof(double), 256);
  p.off_fbu = p.off_fbx + align_up((size_t)p.fb.n_chunks * p.fb.nrep_pad * p.fb.C_pad * K * sizeof(double), 256);
  p.off_prog = p.off_fbu + align_up((size_t)p.fb.n_chunks * p.fb.nrep_pad * (K + 1) * sizeof(double), 256);  // + the y run's
  p.prog_bytes = (size_t)p.n_chunks * 64 * sizeof(uint32_t);
  p.off_stats = p.off_prog + align_up(p.prog_bytes, 256);
  p.off_py = p.off_stats + align_up((size_t)cdiv(p.ntiles, p.win_tiles < 16 ? p.win_tiles : 16) * 100 * sizeof(double), 256);
  p.off_fby = p.off_py + align_up((size_t)p.nwin * p.nrep_pad * 8 * I8_CPAD * sizeof(double), 256);
  p.off_prep = p.off_fby + align_up((size_t)p.fb.n_chunks * p.fb.nrep_pad * p.fb.C_pad * sizeof(double), 256);
  // the count table sits LAST: `total` is the size without it, `total_table` with it -- a call runs the table kernel when
  // the rule (table_kernel_pays) or the caller asks for it AND the workspace it was given reaches total_table
  p.off_gprog = p.off_prep + align_up(p.prep_total, 256);
  p.off_table = p.off_gprog + align_up((size_t)cdiv(p.nwin, 8) * 8 * 16 * sizeof(uint32_t), 256);
  // (narrow states -- C <= 16, order >= 1 -- have a table-fed kernel of their own since round 6: txm_resample_i8gn.hip)
  p.table_bytes = (N >= SM_T && (C > 16 || K >= 2)) ? count_table_bytes(p.ntiles, nrep) : 0;
  p.total = p.off_table;
  p.total_table = p.off_table + align_up(p.table_bytes, 256);
  return p;
}

// the process-wide default of TXM_PATH_AUTO calls (txm_set_resample_path; tests and A/B timing): -1 = the shape rule
// below, TXM_PATH_FP64 / TXM_PATH_INT8 = forced wherever the kernel supports the shape.  No environment variable is read
// anywhere in the library (rounds 1-4 read TXM_I8 / TXM_THROTTLE / TXM_PACK / TXM_I8W); a call that passes
// txm_resample_opts.path does not look at this word either.
static int g_path_override = -1;
static int path_override() { return g_path_override; }

// the L2-sharing hint of the bootstrap kernels; an A/B build without it: -DTXM_NO_THROTTLE (tools/build_variant.sh)
static constexpr bool throttle_on() {
#ifdef TXM_NO_THROTTLE
  return false;
#else
  return true;
#endif
}

// Wide states on the int8 path: which of its two contraction kernels.  They agree bit for bit (the same int32 sums, the same
// flush), so this is a rule on speed alone -- measured on MI355X, pre-pass block kept, ms per call fused / table
// (tools/i8g_sweep.py, profiles/r05_sweep.txt; N = 1e8, C = 32, nrep = 1000):
//   order 0: 92.5 / 74.1   1: 95.1 / 82.9   2: 115.8 / 99.3   3: 134.2 / 137.2   4: 162.8 / 162.7   5: 231.7 / 174.0
//   6: 259.9 / 217.7   7: 268.2 / 226.4;   with a second matrix (N = 1e7, nrep = 256): order 1: 4.47 / 3.94, 4: 8.36 / 5.97,
//   6: 8.79 / 7.58.
// The table kernel takes at most three row sets per pass and pays the count-table generator (28 ms per 1e11 counts) once per
// call; every call with a second matrix and every order but 3 is faster on it.  Orders 3 and 4 -- four and five row sets, two
// passes against the fused kernel's one -- were ties on the kernel's first cut; after its re-cut (reads a slot ahead, no
// spills; same box, fused / table, profiles/r05_orders34_ab.txt): order 4 168.3 / 163.2 (N = 1e8, nrep = 1000), 37.0 / 35.5
// (2e7), 6.2 / 5.95 (1e7, nrep = 200), 1.55 / 1.42 (1e6, 384), 1.72 / 1.76 (3e6, 128); order 3 145.6 / 145.1, 31.8 / 31.9,
// 5.47 / 5.55, 1.47 / 1.56 -- order 4 moved to the table kernel; order 3 followed once its passes were split 3 + 1 (below).
// Replicates come in groups of 128 there, 64 on the fused kernel: a call whose padding to 128 wastes much more than its
// padding to 64 stays fused (nrep = 64: 1.80 / 2.66 ms; 130: 3.45 / 3.92; 100: 2.87 / 2.93; 200, 256, 1000: the table above).
static bool table_kernel_pays(int64_t nrep, int K, bool has_y) {
  const int64_t pad128 = cdiv(nrep, G_REPS) * G_REPS, pad64 = cdiv(nrep, I8_REPS) * I8_REPS;
  if (4 * pad128 > 5 * pad64) return false;
  if (has_y) return true;
  // order 4 is only ~3 % ahead on the table kernel, and one 128-replicate group is 382 workgroups per 1e8 samples -- 1.5 rounds of
  // the 256 CUs -- where two 64-replicate groups of the fused kernel are 3.0 (a 125-replicate slab, what one of 8 ranks runs in
  // bench.py --mode replicas: 22.6 ms fused, 24.4 table)
  // order 3 likewise since its four row sets go as 3 + 1 (the single-row-set pass takes 256 replicates per workgroup: 101 ms for
  // the two passes against 109 for 2 + 2): 142.2 / 150.1 ms table / fused at N = 1e8, 31.1 / 32.7 at 2e7, 5.4 / 5.5 at 1e7 x 200,
  // 1.27 / 1.32 at 1e6 x 384, 1.59 / 1.50 at 3e6 x 128 (profiles/r05_pass_split_ab.txt)
  if (K == 4 || K == 5) return pad128 >= 2 * G_REPS;
  return true;
}

// ONE statement of "does this call run the count-table kernel" for the call itself, the workspace query and
// txm_resample_kernel (what the host keys a kept pre-pass block on): `eff` is the call's path after the process-wide
// override, `applicable` = i8g_applicable() of the operands.  The call additionally needs the table's bytes in its workspace.
// Narrow states (C <= 16): the table-fed kernel of txm_resample_i8gn.hip (128 replicates per workgroup, no fill phase) against the
// quad-sharing variant of the kernel that draws in place -- bit for bit the same sums, so again a rule on speed alone.  Measured on
// MI355X (tools/narrow_table_sweep.py: N = 1e6 .. 3e7, C = 4 .. 16, orders 1 .. 6, 64 .. 1000 replicates), fused ms / table ms:
//   first sweep (profiles/r06_narrow_table_sweep.txt): 0.6 - 0.9 at 64 replicates (the padding to 128 doubles the work), 0.9 - 1.2
//   with ONE group of 128, 1.0 - 1.27 from two groups on -- the rule took the table kernel there for every long series.
//   Re-swept twice in round 6's second session, because both kernels' window boundary got cheaper in turn:
//   (profiles/r06_narrow_table_sweep2.txt, _sweep3.txt) after the FUSED kernel's chunk groups got their cheaper flush and digit-summed
//   slots (6 % faster at the median, 12 % on short series whose windows are four tiles): from two groups on the two kernels were within
//   +- 5 % on most shapes, the table kernel clearly ahead only on long series at some orders -- a rule of four cases followed;
//   (profiles/r06_narrow_table_sweep4.txt: 560 shapes, N = 1e6, 3e6, 4.5e6, 1e7) after the TABLE kernel got digit-summed slots too
//   (one workgroup per window: a short series' call wrote and re-read as many bytes of per-digit slots as of counts): 0.65 - 1.04 at
//   64 replicates, 0.91 - 1.33 with ONE group of 128 (100, 128 replicates: below 1 only for low orders of the widest states around
//   N = 4.5e6), 0.92 - 1.40 from two groups on (median 1.10; N = 1e6: 0.95 - 1.38) -- four column quads at order 4 included (1e6:
//   1.15 - 1.36, 1e7: 0.96 - 1.11).  A one-line rule followed (mean regret 0.5 % on that sweep; the four-case rule: 6.3 %);
//   (profiles/r06_narrow_table_sweep5.txt: 420 shapes) after the fused instances WITHOUT chunk groups got digit-summed slots as well:
//   four column quads at order 4 -- one fused pass against two table passes -- 16 - 19 % faster on the fused kernel, 0.88 - 1.05 now:
//   the exception of the first rule is back.  The rule below picks the slower kernel by more than 3 % on 19 of the 420 shapes (worst
//   10 %), mean regret 0.4 %.
static bool narrow_table_pays(int64_t N, int64_t C, int64_t nrep, int K) {
  const int64_t pad128 = cdiv(nrep, G_REPS) * G_REPS, pad64 = cdiv(nrep, I8_REPS) * I8_REPS;
  if (N < 786432 || nrep <= I8_REPS || pad128 > pad64) return false;
  return !(i8t_narrow_nq(C, K) == 4 && K == 5);
}
static bool table_call_rule(size_t table_bytes, int eff, int64_t N, int64_t C, int64_t nrep, int K, bool has_y, bool applicable) {
  if (eff == TXM_PATH_FP64 || eff == TXM_PATH_INT8_FUSED || table_bytes == 0 || !applicable) return false;
  if (eff == TXM_PATH_INT8_TABLE) return true;
  return C > 16 ? table_kernel_pays(nrep, K, has_y) : narrow_table_pays(N, C, nrep, K);
}
// what the table-fed kernels ask of the operands (16-byte LDS-DMA pieces): wide states txm_resample_i8g.hip, narrow ones
// txm_resample_i8gn.hip (a second matrix never rides a narrow call: it is bootstrapped on its own)
static bool table_operands_ok(const double *x, int64_t ldx_s, int64_t C, int K, const double *y, int64_t ldy_s) {
  return C > 16 ? i8g_applicable(x, ldx_s, C, y, ldy_s) : i8gn_applicable(x, ldx_s, C, K);
}

static bool use_i8(int64_t N, int64_t C, int64_t nrep, int K, int call_path = TXM_PATH_AUTO) {
  if (!i8_supported(N, C, nrep, K)) return false;
  const int ov = call_path != TXM_PATH_AUTO ? call_path : path_override();
  if (ov == TXM_PATH_FP64) return false;
  if (ov == TXM_PATH_INT8 || ov == TXM_PATH_INT8_FUSED || ov == TXM_PATH_INT8_TABLE) return true;
  // measured on MI355X (tools/i8_sweep.py, N = 1e7; tools/ab_order.py, N = 1e8): C <= 16 runs one 16-column FP64
  // block and stays ahead; with two blocks the int8 kernel wins from one full replicate group on at order >= 3
  // (order 4: 1.2x at 64 replicates, 1.5x at 128, 1.6x from 400), from 128 replicates at orders 1 and 2 (nrep = 128:
  // 25 vs 32 ms and 29 vs 43 ms; nrep = 1000: 157 vs 254 ms and 181 vs 342 ms) and from ~400 at order 0 (nrep = 200:
  // a tie; 400: 74 vs 80 ms; 1000: 151 vs 176 ms) -- round 3's numbers; the long-series thresholds below are round 4's re-measurement.
  // Narrow states (C <= 16, order >= 1: the quad-sharing variant of the transposing-read kernel with chunk groups, round 4):
  // ahead of the power-packed FP64 kernel at EVERY replicate count from 4 to 200 once the series is long (tools/
  // i8_sweep_narrow.py, N = 1e7, C = 1 .. 16, orders 1 .. 4: 1.5 - 2.4 x with the pre-pass block kept by the data object,
  // 0.93 - 2.1 x on a first call that computes it; profiles/r04_narrow_sweep.txt).  On short series the pre-pass weighs more:
  // N = 3e5: 1.05 - 1.6 x kept, 0.6 - 1.2 x on the first call, ahead on both from 128 replicates.  A rule on (N, nrep) that a
  // replicate slab of a long series never crosses: from 3 x 2^18 samples every slab takes the kernel the whole call takes
  // (N = 1e6: 1.2 - 1.35 x on the first call, 1.9 x kept).
  if (C <= 16) return K >= 2 && (N >= 786432 || (N >= 262144 && nrep >= 128));
  // Wide states, re-measured in round 4 (tools/i8_sweep_narrow.py 1e7 32 ..., profiles/r04_narrow_sweep.txt; pre-pass block kept /
  // first call): N = 1e7, order 4: 2.1 / 1.4 x at 32 replicates, 2.4 / 1.9 at 100; orders 1-2: 1.8-1.9 / 1.0-1.15 at 32, 2.1-2.25 /
  // 1.4-1.6 at 100; order 0: 1.4-1.7 x kept at every count, but a first call only pays from 100 replicates (1.03; 0.72 at 64).
  // N = 3e5: ahead on a first call from 128 replicates only (order 4: 1.17, order 1: 0.86).
  // (a tail group of <= 16 columns runs the narrow-state variant; at order 0, which has none, such a call stays on the FP64 kernel)
  const int64_t ctail = C % I8_CPAD;
  const bool long_series = N >= 786432;
  const int64_t min_rep = long_series ? (K >= 2 ? 32 : 100) : (K >= 4 ? 64 : (K >= 2 ? 128 : 384));  // (short series: round 3's rule)
  return C > 16 && (ctail == 0 || ctail > 16 || K >= 2) && nrep >= min_rep && N >= 262144;
}

}  // namespace txm

using namespace txm;

extern "C" int txm_set_resample_path(int path) {
  TXM_REQUIRE(path == -1 || path == TXM_PATH_FP64 || path == TXM_PATH_INT8 || path == TXM_PATH_INT8_FUSED || path == TXM_PATH_INT8_TABLE, "set_resample_path: %d is not a path", path);
  g_path_override = path;
  return TXM_OK;
}

namespace txm {
// info [4] on the device: path, windows x groups, windows the guard sent to the FP64 kernel, tables came from prep
__global__ void i8_info_kernel(const unsigned char *prep_groups, size_t group_stride, size_t off_nlist, int ngroups,
                               int64_t nwin, int from_prep, int64_t *info) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t flagged = 0;
    for (int g = 0; g < ngroups; ++g)
      flagged += reinterpret_cast<const uint32_t *>(prep_groups + (size_t)g * group_stride + off_nlist)[1];
    info[0] = TXM_PATH_INT8;
    info[1] = nwin * ngroups;
    info[2] = flagged;
    info[3] = from_prep;  // bit 0: tables came from the caller's block; bit 1: wide groups ran the count-table kernel
  }
}
__global__ void fp64_info_kernel(int64_t *info) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    info[0] = TXM_PATH_FP64;
    info[1] = info[2] = info[3] = 0;
  }
}
}  // namespace txm
using namespace txm;

extern "C" int txm_resample_vals_info(const void *ws, int64_t N, int64_t C, int64_t nrep, int order,
                                      int64_t *info_host, txm_stream stream) {
  TXM_REQUIRE(ws && info_host, "resample_vals_info: null pointer");
  TXM_REQUIRE(N >= 1 && C >= 1 && nrep >= 1 && order >= 0 && order <= TXM_MAX_ORDER, "resample_vals_info: bad shape");
  info_host[0] = TXM_PATH_FP64;
  info_host[1] = info_host[2] = 0;
  if (!use_i8(N, C, nrep, order + 1)) return TXM_OK;
  const I8Plan q = plan_i8(N, C, nrep, order + 1);
  // the last call's info block sits at the head of the workspace's own prep region's n_list words; read them back
  int64_t flagged = 0;
  for (int g = 0; g < q.ngroups; ++g) {
    uint32_t nl[2] = {0, 0};
    TXM_HIP(hipMemcpyAsync(nl, (const char *)ws + q.off_prep + q.prep_group0 + (size_t)g * q.prep_group_stride + q.prep_nlist,
                           sizeof(nl), hipMemcpyDeviceToHost, (hipStream_t)stream));
    TXM_HIP(hipStreamSynchronize((hipStream_t)stream));
    flagged += nl[1];
  }
  info_host[0] = TXM_PATH_INT8;
  info_host[1] = q.nwin * q.ngroups;
  info_host[2] = flagged;
  return TXM_OK;
}

extern "C" int txm_resample_path(int64_t N, int64_t C, int64_t nrep, int order) {
  if (N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return TXM_PATH_FP64;
  return use_i8(N, C, nrep, order + 1) ? TXM_PATH_INT8 : TXM_PATH_FP64;
}

// The contraction kernel ONE device-sampler call with these options runs -- TXM_PATH_FP64, TXM_PATH_INT8_FUSED or
// TXM_PATH_INT8_TABLE -- given a workspace of txm_resample_vals_ws_bytes_opts(same arguments) bytes, | TXM_KERNEL_WITH_Y when
// that kernel carries the second matrix (so that the call's pre-pass block holds y's pivot, scales and guard flags).
// `aligned`: x (and y) 16-byte aligned with an even row pitch >= C rounded up to 4 -- what the table kernel's DMA needs.
// A kept pre-pass block serves exactly the calls that return the same word here (and the same N, C, order, tensors).
extern "C" int txm_resample_kernel(int64_t N, int64_t C, int64_t nrep, int order, int path, int has_y, int aligned) {
  if (N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return TXM_PATH_FP64;
  const int K = order + 1;
  if (!use_i8(N, C, nrep, K, path)) return TXM_PATH_FP64;
  const I8Plan q = plan_i8(N, C, nrep, K);
  const int eff = path != TXM_PATH_AUTO ? path : path_override();
  const bool table = table_call_rule(q.table_bytes, eff, N, C, nrep, K, has_y != 0, aligned != 0);
  const bool with_y = has_y != 0 && ((table && C > 16) || i8t_carries_y(C, K));
  return (table ? TXM_PATH_INT8_TABLE : TXM_PATH_INT8_FUSED) | (with_y ? TXM_KERNEL_WITH_Y : 0);
}

// the `aligned` argument of txm_resample_kernel for a given pair of operands (y may be NULL): what the count-table kernel's
// LDS-DMA asks of them -- ONE statement of it, the one the call itself applies (i8g_applicable)
extern "C" int txm_resample_operands_aligned(const double *x, int64_t ldx_s, int64_t C, const double *y, int64_t ldy_s) {
  if (x == nullptr || C < 1) return 0;
  return table_operands_ok(x, ldx_s, C, 2, y, ldy_s) ? 1 : 0;  // (K only decides whether a narrow shape is served at all: txm_resample_kernel's business)
}

extern "C" size_t txm_resample_prep_bytes(int64_t N, int64_t C, int64_t nrep, int order) {
  if (N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  if (!i8_supported(N, C, nrep, order + 1)) return 256;
  return plan_i8(N, C, nrep, order + 1).prep_total;
}

// the second sample matrix of txm_resample_opts (y -> per-replicate means) runs as an order-0 bootstrap of its own
// behind the main call unless the kernel of the main call carries it: its states [nrep][C][2][1] and its own
// workspace sit behind the main plan's
static size_t y_main_bytes(int64_t N, int64_t C, int64_t nrep) {
  size_t n = plan_resample(N, C, nrep, 1).total;
  if (i8_supported(N, C, nrep, 1)) {
    const size_t m = plan_i8(N, C, nrep, 1).total;
    if (m > n) n = m;
  }
  return align_up(n, 256);
}
static size_t y_extra_bytes(int64_t N, int64_t C, int64_t nrep) {
  return align_up((size_t)nrep * C * 2 * sizeof(double), 256) + y_main_bytes(N, C, nrep);
}

// path / has_y: what the call will pass in txm_resample_opts -- they decide whether the int8 path's count table (one byte
// per replicate and sample, replicates padded to 128) is part of the workspace: TXM_PATH_AUTO / TXM_PATH_INT8 follow the
// library's rule (table_kernel_pays), TXM_PATH_INT8_TABLE always asks for it, TXM_PATH_FP64 / _INT8_FUSED never.  A call
// handed less than this runs the kernel without the table.
extern "C" size_t txm_resample_vals_ws_bytes_opts(int64_t N, int64_t C, int64_t nrep, int order, int path, int has_y) {
  if (N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  size_t n = plan_resample(N, C, nrep, order + 1).total;
  if (i8_supported(N, C, nrep, order + 1)) {
    const I8Plan q = plan_i8(N, C, nrep, order + 1);
    const int eff = path != TXM_PATH_AUTO ? path : path_override();
    const bool table = table_call_rule(q.table_bytes, eff, N, C, nrep, order + 1, has_y != 0, true);
    const size_t m = table ? q.total_table : q.total;
    if (m > n) n = m;
  }
  return align_up(n, 256);
}

extern "C" size_t txm_resample_vals_ws_bytes(int64_t N, int64_t C, int64_t nrep, int order) {
  return txm_resample_vals_ws_bytes_opts(N, C, nrep, order, TXM_PATH_AUTO, 0);
}

// ... and the scratch a call with opts.y needs BEHIND those bytes when the kernel of the main call does not carry the
// second matrix (an order-0 bootstrap of its own: its states and its plan).  Opt-in: at the north-star shape it is
// ~2 GB that a call without opts.y never touches.
extern "C" size_t txm_resample_y_ws_bytes(int64_t N, int64_t C, int64_t nrep) {
  if (N < 1 || C < 1 || nrep < 1) return 0;
  return y_extra_bytes(N, C, nrep);
}

// whether the int8 kernel can take the shape at all (TXM_PATH_INT8 on a shape it cannot take runs the FP64 kernel)
extern "C" int txm_resample_i8_supported(int64_t N, int64_t C, int64_t nrep, int order) {
  if (N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  return i8_supported(N, C, nrep, order + 1) ? 1 : 0;
}

#define TXM_K_SWITCH(K_, CALL)                          \
  switch (K_) {                                         \
    case 1: { constexpr int KK = 1; CALL; } break;      \
    case 2: { constexpr int KK = 2; CALL; } break;      \
    case 3: { constexpr int KK = 3; CALL; } break;      \
    case 4: { constexpr int KK = 4; CALL; } break;      \
    case 5: { constexpr int KK = 5; CALL; } break;      \
    case 6: { constexpr int KK = 6; CALL; } break;      \
    case 7: { constexpr int KK = 7; CALL; } break;      \
    case 8: { constexpr int KK = 8; CALL; } break;      \
    case 9: { constexpr int KK = 9; CALL; } break;      \
    default: set_error("order out of range"); return TXM_ERR_INVALID; \
  }

namespace txm {
// narrow states: several powers per B-operand column; an A/B build with one power per column: -DTXM_NO_PACK
static constexpr bool pack_on() {
#ifdef TXM_NO_PACK
  return false;
#else
  return true;
#endif
}
template <int K>
static int run_resample(ResampleArgs a, const ResamplePlan &p, bool weighted, bool explicit_,
                        double *out, hipStream_t st, int64_t S = 1) {
  dim3 grid((unsigned)(p.n_chunks * p.n_rbg), (unsigned)p.colgroups, (unsigned)S), block(RS_BLOCK);
  const size_t lds = explicit_ ? 0 : (size_t)RS_WAVES * RS_TILE_BYTES;
#define TXM_RS_LAUNCH(NB, WT, EX)                                                                       \
  do {                                                                                                 \
    if (S > 1 || a.batch != nullptr) {                                                                 \
      if (a.N < SM_T)                                                                                  \
        hipLaunchKernelGGL((resample_kernel<K, NB, WT, EX, true, RS_BATCHED>), grid, block, lds, st, a);  \
      else                                                                                             \
        hipLaunchKernelGGL((resample_kernel<K, NB, WT, EX, false, RS_BATCHED>), grid, block, lds, st, a); \
    } else if (a.N < SM_T)                                                                             \
      hipLaunchKernelGGL((resample_kernel<K, NB, WT, EX, true>), grid, block, lds, st, a);             \
    else                                                                                               \
      hipLaunchKernelGGL((resample_kernel<K, NB, WT, EX, false>), grid, block, lds, st, a);            \
  } while (0)
  // narrow states (C <= 8, device sampler, full tiles): PACK powers per B column -> ceil(K / PACK) MFMAs per k-step
  bool packed = false;
  if constexpr (K >= 2 && K <= 6) {
    if (p.nblk == 1 && p.colgroups == 1 && !explicit_ && a.N >= SM_T && a.C <= 8 && pack_on()) {
      packed = true;
      const bool bat = S > 1 || a.batch != nullptr;
#define TXM_RS_PACKED(PK)                                                                                      \
  do {                                                                                                         \
    if (bat) {                                                                                                 \
      if (weighted) hipLaunchKernelGGL((resample_kernel<K, 1, true, false, false, RS_BATCHED, PK>), grid, block, lds, st, a); \
      else hipLaunchKernelGGL((resample_kernel<K, 1, false, false, false, RS_BATCHED, PK>), grid, block, lds, st, a);         \
    } else {                                                                                                   \
      if (weighted) hipLaunchKernelGGL((resample_kernel<K, 1, true, false, false, RS_PLAIN, PK>), grid, block, lds, st, a);   \
      else hipLaunchKernelGGL((resample_kernel<K, 1, false, false, false, RS_PLAIN, PK>), grid, block, lds, st, a);           \
    }                                                                                                          \
  } while (0)
      if (a.C <= 4 && K >= 3) TXM_RS_PACKED(4);
      else TXM_RS_PACKED(2);
#undef TXM_RS_PACKED
    }
  }
  if (packed) {
  } else if (p.nblk == 1) {
    if (weighted) { if (explicit_) TXM_RS_LAUNCH(1, true, true); else TXM_RS_LAUNCH(1, true, false); }
    else          { if (explicit_) TXM_RS_LAUNCH(1, false, true); else TXM_RS_LAUNCH(1, false, false); }
  } else {
    if (weighted) { if (explicit_) TXM_RS_LAUNCH(2, true, true); else TXM_RS_LAUNCH(2, true, false); }
    else          { if (explicit_) TXM_RS_LAUNCH(2, false, true); else TXM_RS_LAUNCH(2, false, false); }
  }
#undef TXM_RS_LAUNCH
  TXM_LAUNCH_CHECK();
  const int64_t ne = a.nrep * a.C;
  hipLaunchKernelGGL((resample_finalize_kernel<K>), dim3((unsigned)cdiv(ne, 32), (unsigned)S), dim3(256), 0, st,
                     a.part_x, a.part_u, p.n_chunks, p.nrep_pad, p.C_pad, a.nrep, a.C, a.pivot, out);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// the bootstrap kernel alone, in listed mode (device sampler, N >= one tile), partial sums left for the caller's finalize
template <int K>
static int run_listed_k(const ResampleArgs &a, const ResamplePlan &p, bool weighted, hipStream_t st, int64_t S = 1) {
  dim3 grid((unsigned)(p.n_chunks * p.n_rbg), (unsigned)p.colgroups, (unsigned)S), block(RS_BLOCK);
  const size_t lds = (size_t)RS_WAVES * RS_TILE_BYTES;
  if (p.nblk == 1) {
    if (weighted) hipLaunchKernelGGL((resample_kernel<K, 1, true, false, false, RS_LISTED>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((resample_kernel<K, 1, false, false, false, RS_LISTED>), grid, block, lds, st, a);
  } else {
    if (weighted) hipLaunchKernelGGL((resample_kernel<K, 2, true, false, false, RS_LISTED>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((resample_kernel<K, 2, false, false, false, RS_LISTED>), grid, block, lds, st, a);
  }
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

static int run_listed(const ResampleArgs &a, const ResamplePlan &p0, int K, bool weighted, hipStream_t st, int64_t S = 1) {
  // the plan was made for a full 32-column group; a narrower last group uses one 16-column block
  ResamplePlan p = p0;
  if (a.C <= 16) p.nblk = 1;
  p.colgroups = 1;
  TXM_K_SWITCH(K, return run_listed_k<KK>(a, p, weighted, st, S));
  return TXM_OK;
}
}  // namespace txm

namespace txm {
// y[N][C] -> the mean column of its order-0 replicate states
__global__ void y_means_kernel(const double *__restrict__ st /*[n][2]*/, int64_t n, double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = st[2 * e + 1];
}

static int resample_vals_impl(const double *x, int64_t ldx_s, const double *u, const double *w, int64_t N, int64_t C,
                              int order, int64_t nrep, const int64_t *freq, const txm_sampler_spec *spec,
                              const uint32_t *counts, const double *pivot, double *out, int path, void *prep,
                              size_t prep_bytes, bool prep_valid, int64_t *info, void *ws, size_t ws_bytes,
                              hipStream_t st, const double *y = nullptr, int64_t ldy_s = 0, double *out_y = nullptr,
                              bool *y_done = nullptr) {
  // y / out_y: the second sample matrix of txm_resample_opts.  Carried by this call when the int8 kernel can take it
  // as a row set of its last pass (*y_done = true); otherwise left to the caller (a separate order-0 bootstrap).
  const bool explicit_ = freq != nullptr;
  const int K = order + 1;
  if (y_done) *y_done = false;
  if (!explicit_ && use_i8(N, C, nrep, K, path)) {
    const I8Plan q = plan_i8(N, C, nrep, K);
    // wide states: the per-sample counts of the call's replicates as a table in HBM (built once, shared by the column groups
    // and the passes) and the contraction kernel without a sampler inside (txm_resample_i8g.hip).  Misaligned operands and
    // TXM_PATH_INT8_FUSED keep the kernel that draws in place; narrow states and narrow tail groups always run it.
    const int eff_path = path != TXM_PATH_AUTO ? path : path_override();
    const bool table_call = ws_bytes >= q.total_table &&
                            table_call_rule(q.table_bytes, eff_path, N, C, nrep, K, y != nullptr, table_operands_ok(x, ldx_s, C, K, y, ldy_s));
    const bool with_y = y != nullptr && ((table_call && C > 16) || i8t_carries_y(C, K));
    if (ws_bytes < q.total) {
      set_error("resample_vals: workspace too small (%zu < %zu)", ws_bytes, q.total);
      return TXM_ERR_WORKSPACE;
    }
    TXM_REQUIRE(spec->ndat == N && spec->nrep == nrep, "resample_vals: sampler spec does not match N/nrep");
    TXM_REQUIRE(spec->rep0 >= 0 && spec->rep0 + nrep <= ((int64_t)1 << 32), "resample_vals: sampler rep0 out of range");
    // the pre-pass block: the caller's persistent buffer (tables reused when prep_valid) or scratch in ws
    unsigned char *pb = (unsigned char *)ws + q.off_prep;
    bool have_tables = false;
    if (prep != nullptr) {
      if (prep_bytes < q.prep_total) {
        set_error("resample_vals: prep buffer too small (%zu < %zu)", prep_bytes, q.prep_total);
        return TXM_ERR_WORKSPACE;
      }
      pb = (unsigned char *)prep;
      have_tables = prep_valid;
    }
    double *piv = (double *)pb;
    double *ypiv = (double *)(pb + q.prep_ypiv);
    if (!have_tables) {
      if (pivot) {
        TXM_HIP(hipMemcpyAsync(piv, pivot, sizeof(double) * (size_t)(1 + C), hipMemcpyDeviceToDevice, st));
      } else {
        hipLaunchKernelGGL(pivot_kernel, dim3((unsigned)(1 + C)), dim3(256), 0, st, x, ldx_s, (int64_t)1,
                           u, (int64_t)1, N, piv);
        TXM_LAUNCH_CHECK();
      }
      if (with_y) {
        hipLaunchKernelGGL(pivot_kernel, dim3((unsigned)(1 + C)), dim3(256), 0, st, y, ldy_s, (int64_t)1, u, (int64_t)1,
                           N, ypiv);
        TXM_LAUNCH_CHECK();
        // one u pivot for the call (the monomial scales of both matrices are built on it)
        TXM_HIP(hipMemcpyAsync(ypiv, piv, sizeof(double), hipMemcpyDeviceToDevice, st));
      }
    }
    I8Args b;
    b.x = x; b.ldx_s = ldx_s; b.u = u; b.w = w; b.N = N; b.nrep = nrep; b.C_call = C;
    b.cpad = i8_cpad(C, K);
    b.counts = counts;
    b.k0 = (uint32_t)spec->seed;
    b.k1 = (uint32_t)(spec->seed >> 32);
    b.rep_base = (uint32_t)spec->rep0;
    b.ntiles = q.ntiles;
    b.last_tile_size = (uint32_t)(N - (q.ntiles - 1) * SM_T);
    b.pivot = piv;
    b.stats = (double *)((char *)ws + q.off_stats);
    b.nwin = q.nwin;
    b.part_x = (double *)((char *)ws + q.off_px);
    b.part_u = (double *)((char *)ws + q.off_pu);
    b.n_chunks = q.n_chunks; b.n_rbg = q.n_rbg; b.tiles_per_chunk = q.tiles_per_chunk;
    b.nrep_pad = q.nrep_pad;
    b.win_tiles = q.win_tiles;
    b.sub_tiles = q.sub_tiles;
    b.y = with_y ? y : nullptr;
    b.ldy_s = ldy_s;
    b.ypivot = ypiv;
    b.part_y = (double *)((char *)ws + q.off_py);
    b.part_summed = i8t_wide_summed(with_y) ? 1 : 0;  // (what the wide fused kernel's instances of this call store: the finalize's mode below)
    b.progress = (throttle_on() && q.n_rbg > 1) ? (uint32_t *)((char *)ws + q.off_prog) : nullptr;
    // the FP64 kernel in listed mode: contracts the windows the precision guard flags (none on ordinary data)
    ResampleArgs f;
    f.ldx_s = ldx_s; f.u = u; f.w = w; f.N = N; f.nrep = nrep;
    f.freq = nullptr; f.counts = counts;
    f.k0 = b.k0; f.k1 = b.k1; f.rep_base = b.rep_base;
    f.ntiles = q.ntiles; f.last_tile_size = b.last_tile_size;
    f.pivot = piv;
    f.part_x = (double *)((char *)ws + q.off_fbx);
    f.part_u = (double *)((char *)ws + q.off_fbu);
    f.n_chunks = q.fb.n_chunks; f.n_rbg = q.fb.n_rbg; f.tiles_per_chunk = q.fb.tiles_per_chunk;
    f.nrep_pad = q.fb.nrep_pad; f.C_pad = q.fb.C_pad;
    f.sub_tiles = q.sub_tiles; f.batch = nullptr; f.progress = nullptr;
    TXM_REQUIRE(q.fb.nrep_pad == q.nrep_pad, "resample_vals: replicate padding of the two kernels differs");
    // one launch (or two, orders 5-7) per group of 32 columns; the groups reuse the partial buffers
    int g = 0;
    bool have_table = false;
    for (int64_t col0 = 0; col0 < C; col0 += I8_CPAD, ++g) {
      unsigned char *pg = pb + q.prep_group0 + (size_t)g * q.prep_group_stride;
      b.wtab = (double *)(pg + q.prep_wt);
      b.wflag = (uint32_t *)(pg + q.prep_flag);
      b.list = (uint32_t *)(pg + q.prep_list);
      b.n_list = (uint32_t *)(pg + q.prep_nlist);
      b.ywtab = (double *)(pg + q.prep_ywt);
      b.yflag = (uint32_t *)(pg + q.prep_yflag);
      f.list = b.list; f.n_list = b.n_list;
      b.col0 = col0;
      b.C = C - col0 < I8_CPAD ? C - col0 : I8_CPAD;
      // a tail group of <= 16 columns behind full groups runs the narrow-state variant (its own partial-sum row width) --
      // unless a second matrix rides on the pass, which only the wide variant carries
      const bool narrow_tail = C > I8_CPAD && b.C <= 16 && K >= 2 && !with_y;
      b.C_call = narrow_tail ? b.C : C;
      b.cpad = i8_cpad(b.C_call, K);
      if (!have_tables) {
        const int rc0 = launch_i8_prepass(b, K, st);
        if (rc0 != TXM_OK) return rc0;
      }
      const bool table_kernel = table_call && !narrow_tail;
      int rc;
      if (table_kernel) {
        unsigned char *table = (unsigned char *)ws + q.off_table;
        if (!have_table) {
          rc = launch_count_table(counts, nrep, N, b.k0, b.k1, b.rep_base, 0, cdiv(nrep, G_REPS), table, st);
          if (rc != TXM_OK) return rc;
          have_table = true;
        }
        I8Args bg = b;
        bg.progress = (throttle_on() && nrep > G_REPS) ? (uint32_t *)((char *)ws + q.off_gprog) : nullptr;
        rc = C > 16 ? launch_resample_i8g(bg, K, w != nullptr, table, 0, (int)cdiv(nrep, G_REPS), st)
                    : launch_resample_i8gn(bg, K, w != nullptr, table, 0, (int)cdiv(nrep, G_REPS), 0, st);
      } else {
        rc = launch_resample_i8(b, K, w != nullptr, q.prog_bytes, st);
      }
      if (rc != TXM_OK) return rc;
      f.x = x + col0; f.C = b.C; f.col_off = col0;
      {
        const int rc2 = run_listed(f, q.fb, K, w != nullptr, st);
        if (rc2 != TXM_OK) return rc2;
      }
      // (the slots' layout: digit sums where the fused narrow kernel with chunk groups wrote them, per-digit slots otherwise)
      // (table-fed kernels: always -- txm_resample_i8gn.hip both x and u, txm_resample_i8g.hip x only; the fused kernel: its chunk-group instances)
      // the fused kernel: narrow instances always, the wide one unless a second matrix rides the call (its finalize reads per-digit u slots)
      const int fin_summed = table_kernel ? (C <= 16 ? 1 : 2)
                                          : (i8t_narrow_nq(b.C_call, K) != 0 ? (i8t_partials_summed(b.C_call, K) ? 1 : 0) : b.part_summed);
#define TXM_I8_FIN2(KK, CP)                                                                            \
  hipLaunchKernelGGL((resample_finalize_i8_kernel<KK, CP>), dim3((unsigned)nrep), dim3(256), 0, st,      \
                     b.part_x, b.part_u, q.nwin, b.wflag, q.nrep_pad, nrep, b.C, piv, out, col0, C,        \
                     f.part_x, f.part_u, q.fb.n_chunks, q.fb.C_pad, b.n_list, (const I8State *)nullptr, fin_summed)
#define TXM_I8_FIN(KK)                                                                                 \
  do {                                                                                                 \
    if (b.cpad == 32) TXM_I8_FIN2(KK, 32);                                                             \
    else if (b.cpad == 16) TXM_I8_FIN2(KK, 16);                                                        \
    else if (b.cpad == 8) TXM_I8_FIN2(KK, 8);                                                          \
    else TXM_I8_FIN2(KK, 4);                                                                           \
  } while (0)
      switch (K) {
        case 1: TXM_I8_FIN(1); break;
        case 2: TXM_I8_FIN(2); break;
        case 3: TXM_I8_FIN(3); break;
        case 4: TXM_I8_FIN(4); break;
        case 5: TXM_I8_FIN(5); break;
        case 6: TXM_I8_FIN(6); break;
        case 7: TXM_I8_FIN(7); break;
        default: TXM_I8_FIN(8); break;
      }
#undef TXM_I8_FIN
      TXM_LAUNCH_CHECK();
      if (with_y) {
        // the windows the guard flagged (for either matrix): the FP64 kernel in listed mode on y, order 0
        ResampleArgs fy = f;
        fy.x = y + col0; fy.ldx_s = ldy_s; fy.pivot = ypiv;
        fy.part_x = (double *)((char *)ws + q.off_fby);
        fy.part_u = f.part_u + (size_t)q.fb.n_chunks * q.fb.nrep_pad * K;  // scratch behind x's u-row sums (unused by the finalize)
        const int rc3 = run_listed(fy, q.fb, 1, w != nullptr, st);
        if (rc3 != TXM_OK) return rc3;
        hipLaunchKernelGGL(resample_finalize_y_kernel, dim3((unsigned)nrep), dim3(256), 0, st, b.part_y, b.part_u, K, q.nwin,
                           b.wflag, q.nrep_pad, b.C, ypiv, out_y, col0, C, fy.part_x, f.part_u, q.fb.n_chunks, q.fb.C_pad,
                           b.n_list);
        TXM_LAUNCH_CHECK();
      }
    }
    if (with_y && y_done) *y_done = true;
    // a reused prep block keeps its n_list words; a fresh one inside ws is what txm_resample_vals_info reads back
    if (prep != nullptr)
      TXM_HIP(hipMemcpyAsync((char *)ws + q.off_prep, pb, q.prep_total, hipMemcpyDeviceToDevice, st));
    if (info != nullptr) {
      hipLaunchKernelGGL(i8_info_kernel, dim3(1), dim3(64), 0, st, pb + q.prep_group0, q.prep_group_stride, q.prep_nlist,
                         q.ngroups, q.nwin, (have_tables ? 1 : 0) | (have_table ? 2 : 0), info);
      TXM_LAUNCH_CHECK();
    }
    return TXM_OK;
  }
  if (info != nullptr) {
    hipLaunchKernelGGL(fp64_info_kernel, dim3(1), dim3(64), 0, st, info);
    TXM_LAUNCH_CHECK();
  }
  const ResamplePlan p = plan_resample(N, C, nrep, K);
  if (ws_bytes < p.total) {
    set_error("resample_vals: workspace too small (%zu < %zu)", ws_bytes, p.total);
    return TXM_ERR_WORKSPACE;
  }
  TXM_REQUIRE((int64_t)p.n_chunks * p.n_rbg < ((int64_t)1 << 31), "resample_vals: grid too large");
  double *piv = (double *)((char *)ws + p.off_pivot);
  if (pivot) {
    TXM_HIP(hipMemcpyAsync(piv, pivot, sizeof(double) * (size_t)(1 + C), hipMemcpyDeviceToDevice, st));
  } else {
    hipLaunchKernelGGL(pivot_kernel, dim3((unsigned)(1 + C)), dim3(256), 0, st, x, ldx_s, (int64_t)1,
                       u, (int64_t)1, N, piv);
    TXM_LAUNCH_CHECK();
  }
  ResampleArgs a;
  a.x = x; a.ldx_s = ldx_s; a.u = u; a.w = w; a.N = N; a.C = C; a.nrep = nrep;
  a.freq = freq; a.counts = counts;
  a.k0 = a.k1 = 0;
  a.rep_base = 0;
  a.ntiles = p.ntiles;
  a.last_tile_size = (uint32_t)(N - (p.ntiles - 1) * SM_T);
  if (!explicit_) {
    TXM_REQUIRE(spec->ndat == N && spec->nrep == nrep, "resample_vals: sampler spec does not match N/nrep");
    TXM_REQUIRE(spec->rep0 >= 0 && spec->rep0 + nrep <= ((int64_t)1 << 32), "resample_vals: sampler rep0 out of range");
    a.k0 = (uint32_t)spec->seed;
    a.k1 = (uint32_t)(spec->seed >> 32);
    a.rep_base = (uint32_t)spec->rep0;
  }
  a.pivot = piv;
  a.part_x = (double *)((char *)ws + p.off_px);
  a.part_u = (double *)((char *)ws + p.off_pu);
  a.n_chunks = p.n_chunks; a.n_rbg = p.n_rbg; a.tiles_per_chunk = p.tiles_per_chunk;
  a.nrep_pad = p.nrep_pad; a.C_pad = p.C_pad;
  a.list = nullptr; a.n_list = nullptr; a.sub_tiles = 0; a.col_off = 0; a.batch = nullptr;
  a.progress = nullptr;
  if (throttle_on() && p.n_rbg > 1 && a.N >= SM_T) {
    a.progress = (uint32_t *)((char *)ws + p.off_prog);
    TXM_HIP(hipMemsetAsync(a.progress, 0, p.prog_bytes, st));
  }
  TXM_K_SWITCH(K, return run_resample<KK>(a, p, w != nullptr, explicit_, out, st));
  return TXM_OK;
}
}  // namespace txm

extern "C" int txm_resample_vals(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                                 const double *w, int64_t N, int64_t C, int order, int64_t nrep,
                                 const int64_t *freq, const txm_sampler_spec *spec,
                                 const uint32_t *counts, const double *pivot, double *out,
                                 const txm_resample_opts *opts, void *ws, size_t ws_bytes, txm_stream stream) {
  TXM_REQUIRE(x && u && out && ws, "resample_vals: null pointer");
  TXM_REQUIRE(N >= 1 && C >= 1 && nrep >= 1, "resample_vals: need N, C, nrep >= 1");
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "resample_vals: order %d outside [0, %d]", order,
              TXM_MAX_ORDER);
  TXM_REQUIRE(ldx_c == 1 && ldx_s >= C, "resample_vals: x must be (rec, val) row-major (ldx_c == 1)");
  const bool explicit_ = freq != nullptr;
  TXM_REQUIRE(explicit_ != (spec != nullptr && counts != nullptr),
              "resample_vals: give either freq or (spec, counts)");
  txm_resample_opts o;
  o.path = TXM_PATH_AUTO; o.prep_valid = 0; o.prep = nullptr; o.prep_bytes = 0; o.info = nullptr;
  o.y = nullptr; o.ldy_s = 0; o.out_y = nullptr;
  if (opts) o = *opts;
  TXM_REQUIRE(o.path == TXM_PATH_AUTO || o.path == TXM_PATH_FP64 || o.path == TXM_PATH_INT8 || o.path == TXM_PATH_INT8_FUSED || o.path == TXM_PATH_INT8_TABLE,
              "resample_vals: opts.path %d is not a path", (int)o.path);
  TXM_REQUIRE((o.y == nullptr) == (o.out_y == nullptr), "resample_vals: opts.y and opts.out_y go together");
  TXM_REQUIRE(o.y == nullptr || o.ldy_s >= C, "resample_vals: opts.ldy_s < C");
  hipStream_t st = (hipStream_t)stream;
  const size_t main_bytes = txm_resample_vals_ws_bytes_opts(N, C, nrep, order, o.path, o.y != nullptr);
  const size_t avail = ws_bytes < main_bytes ? ws_bytes : main_bytes;
  bool y_done = false;
  int rc = resample_vals_impl(x, ldx_s, u, w, N, C, order, nrep, freq, spec, counts, pivot, out, o.path, o.prep,
                              o.prep_bytes, o.prep_valid != 0, o.info, ws, avail, st, o.y, o.ldy_s, o.out_y, &y_done);
  if (rc != TXM_OK || o.y == nullptr || y_done) return rc;
  // second sample matrix: an order-0 bootstrap on the same sampler draw, then its mean column
  if (ws_bytes < main_bytes + y_extra_bytes(N, C, nrep)) {
    set_error("resample_vals: workspace too small for opts.y (%zu < %zu)", ws_bytes, main_bytes + y_extra_bytes(N, C, nrep));
    return TXM_ERR_WORKSPACE;
  }
  double *ystates = (double *)((char *)ws + main_bytes);
  void *yws = (char *)ws + main_bytes + align_up((size_t)nrep * C * 2 * sizeof(double), 256);
  rc = resample_vals_impl(o.y, o.ldy_s, u, w, N, C, 0, nrep, freq, spec, counts, nullptr, ystates, o.path, nullptr, 0,
                          false, nullptr, yws, y_main_bytes(N, C, nrep), st);
  if (rc != TXM_OK) return rc;
  hipLaunchKernelGGL(y_means_kernel, dim3((unsigned)cdiv(nrep * C, 256)), dim3(256), 0, st, ystates, nrep * C, o.out_y);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// ---- S state points of one shape: the same kernels with the state on grid axis z -----------------------
namespace txm {
struct BatchPlan {
  ResamplePlan one;  // plan of one state (the chunk count is chosen for S states together)
  size_t off_tab, off_pivot, off_px, off_pu, total;
};
static BatchPlan plan_batched(int64_t S, int64_t N, int64_t C, int64_t nrep, int K) {
  BatchPlan b;
  ResamplePlan &p = b.one;
  p = plan_resample(N, C, nrep, K);
  // S states fill the chip together: fewer sample chunks per state (ntiles / 8) -- again a function of N alone, so that
  // the states [s0, s1) of a collection give the same bits whichever call (rank) they are part of
  (void)S;
  p.tiles_per_chunk = resample_chunk_tiles(p.ntiles, 8);
  p.n_chunks = (int)(cdiv(cdiv(p.ntiles, p.tiles_per_chunk), 8) * 8);
  b.off_tab = 0;
  b.off_pivot = align_up((size_t)S * sizeof(txm_state_ptrs), 256);
  b.off_px = b.off_pivot + align_up((size_t)S * (1 + C) * sizeof(double), 256);
  b.off_pu = b.off_px + align_up((size_t)S * p.n_chunks * p.nrep_pad * p.C_pad * K * sizeof(double), 256);
  b.total = b.off_pu + align_up((size_t)S * p.n_chunks * p.nrep_pad * K * sizeof(double), 256);
  return b;
}
}  // namespace txm

// ---- S narrow state points on the int8 path (BASELINE config 5: 64 states x 1e6 samples x 4 observables) ----------------
// The reference loops over states in Python (models.py:635-641, gpr_active/active_utils.py:896-925).  Every kernel of the
// single-state int8 call (pre-pass, transposing-read kernel in its narrow-state variant, FP64 fallback in listed mode,
// finalize) takes the state from a grid axis and its operands from an I8State table, so the batch is the same arithmetic
// on the same per-window slots as S single calls: the same bits (tests/test_batched_gpu.py).
namespace txm {
struct I8StateLayout {
  size_t stride, s_px, s_pu, s_fbx, s_fbu, s_stats;     // a state's scratch block in the workspace
  size_t p_stride, p_wt, p_flag, p_list, p_nlist;       // a state's pre-pass block (workspace, or the caller's prep buffer)
};
struct BatchI8Plan {
  I8Plan q;                  // one state's plan: windows, replicate groups, the fallback's chunking
  int cpad, n_chunks;
  int64_t tiles_per_chunk;
  I8StateLayout L;
  size_t off_tab, off_states, off_bargs, off_state0, off_prep, total;
  size_t prep_piv, prep_state0, prep_total;  // the pre-pass block: pivots [S][1 + C], then the states' tables
};
static BatchI8Plan plan_batched_i8(int64_t S, int64_t N, int64_t C, int64_t nrep, int K) {
  BatchI8Plan b;
  b.q = plan_i8(N, C, nrep, K);
  const I8Plan &q = b.q;
  b.cpad = i8_cpad(C, K);
  // chunks of whole windows: the S states fill the chip together (the per-window slots make the sums independent of it)
  int64_t nc = (int64_t)num_cus() / ((int64_t)q.n_rbg * S) / 8 * 8;
  if (nc < 8) nc = 8;
  if (nc > q.nwin) nc = cdiv(q.nwin, 8) * 8;
  b.tiles_per_chunk = cdiv(q.nwin, nc) * q.win_tiles;
  b.n_chunks = (int)(cdiv(cdiv(q.ntiles, b.tiles_per_chunk), 8) * 8);
  I8StateLayout &L = b.L;
  L.s_px = 0;
  L.s_pu = L.s_px + align_up((size_t)q.nwin * q.nrep_pad * K * 8 * b.cpad * sizeof(double), 256);
  L.s_fbx = L.s_pu + align_up((size_t)q.nwin * q.nrep_pad * K * 8 * sizeof(double), 256);
  L.s_fbu = L.s_fbx + align_up((size_t)q.fb.n_chunks * q.fb.nrep_pad * q.fb.C_pad * K * sizeof(double), 256);
  L.s_stats = L.s_fbu + align_up((size_t)q.fb.n_chunks * q.fb.nrep_pad * (K + 1) * sizeof(double), 256);
  L.stride = L.s_stats + align_up((size_t)cdiv(q.ntiles, q.win_tiles < 16 ? q.win_tiles : 16) * 100 * sizeof(double), 256);
  L.p_wt = 0;
  L.p_flag = L.p_wt + align_up((size_t)q.nwin * I8_WT_STRIDE * sizeof(double) + 2048, 256);
  L.p_list = L.p_flag + align_up((size_t)q.nwin * sizeof(uint32_t), 256);
  L.p_nlist = L.p_list + align_up((size_t)q.nwin * (size_t)(q.win_tiles / q.sub_tiles) * sizeof(uint32_t), 256);
  L.p_stride = L.p_nlist + 256;
  b.prep_piv = 0;
  b.prep_state0 = align_up((size_t)S * (1 + C) * sizeof(double), 256);
  b.prep_total = b.prep_state0 + (size_t)S * L.p_stride;
  b.off_tab = 0;
  b.off_states = align_up((size_t)S * sizeof(txm_state_ptrs), 256);
  b.off_bargs = b.off_states + align_up((size_t)S * sizeof(I8State), 256);
  b.off_state0 = b.off_bargs + align_up((size_t)S * sizeof(I8Args), 256);
  b.off_prep = b.off_state0 + (size_t)S * L.stride;
  b.total = b.off_prep + align_up(b.prep_total, 256);
  return b;
}

// which batched calls take the int8 path: narrow states (C <= 16: the quad-sharing variant of the transposing-read kernel)
// from order 2 on, with at least one full replicate group per state and windows long enough for the guard's statistics.
// Measured at BASELINE config 5's work (64e6 samples x 4 observables, order 3, nrep = 100): int8 9.6 ms incl. the pre-pass
// against 11.5 ms on the power-packed FP64 kernel (gpurun_out/r4_c5_probe.log).
static bool use_i8_batched(int64_t S, int64_t N, int64_t C, int64_t nrep, int K, int call_path = TXM_PATH_AUTO) {
  if (!i8_supported(N, C, nrep, K) || i8t_narrow_nq(C, K) == 0) return false;
  const int ov = call_path != TXM_PATH_AUTO ? call_path : path_override();
  if (ov == TXM_PATH_FP64) return false;
  if (ov == TXM_PATH_INT8 || ov == TXM_PATH_INT8_FUSED || ov == TXM_PATH_INT8_TABLE) return true;
  // A rule on the STATE's shape only -- never on how many states share the launch: the states of a collection must take the
  // same kernel whichever rank (or workspace-bounded group) they are bootstrapped in, or a sharded run would differ from the
  // one-GPU run in the last bits.  The single-state rule for narrow states (use_i8): long series at any replicate count,
  // short ones from 128 replicates (64 x 1e6 x 4, order 3, nrep = 100: 6.7 against 11.6 ms on the FP64 kernel).
  (void)S;
  return K >= 2 && (N >= 786432 || (N >= 262144 && nrep >= 128));
}

__global__ void i8_states_kernel(const txm_state_ptrs *__restrict__ tab, int64_t S, unsigned char *base, unsigned char *pbase,
                                 I8StateLayout L, const double *piv, int64_t C, const uint32_t *counts, int64_t nrep, int64_t ntiles,
                                 double *out, int K, uint32_t rep_base, I8State *__restrict__ states) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  unsigned char *b = base + (size_t)s * L.stride, *pb = pbase + (size_t)s * L.p_stride;
  I8State e;
  e.x = tab[s].x; e.u = tab[s].u; e.w = tab[s].w;
  e.pivot = piv + s * (1 + C);
  e.stats = (double *)(b + L.s_stats);
  e.wtab = (double *)(pb + L.p_wt);
  e.wflag = (uint32_t *)(pb + L.p_flag);
  e.list = (uint32_t *)(pb + L.p_list);
  e.n_list = (uint32_t *)(pb + L.p_nlist);
  e.part_x = (double *)(b + L.s_px);
  e.part_u = (double *)(b + L.s_pu);
  e.fb_x = (double *)(b + L.s_fbx);
  e.fb_u = (double *)(b + L.s_fbu);
  e.counts = counts + (size_t)s * nrep * ntiles;
  e.out = out + (size_t)s * nrep * C * 2 * K;
  e.rep_base = rep_base + (uint32_t)(s * nrep);
  e.pad_ = 0;
  states[s] = e;
}

// the bootstrap kernel's arguments of state s: the call's, with the state's operands
__global__ void i8_batch_args_kernel(const I8Args base, int64_t S, I8Args *__restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const I8State e = base.states[s];
  I8Args r = base;
  r.x = e.x; r.u = e.u; r.w = e.w; r.pivot = e.pivot; r.wtab = e.wtab; r.wflag = e.wflag;
  r.part_x = e.part_x; r.part_u = e.part_u; r.counts = e.counts; r.rep_base = e.rep_base;
  out[s] = r;
}

static int resample_batched_i8(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N, int64_t C, int K,
                               int64_t nrep, bool weighted, const txm_sampler_spec *spec, const uint32_t *counts, double *out,
                               void *prep, size_t prep_bytes, bool prep_valid, int64_t *info, void *ws, size_t ws_bytes,
                               hipStream_t st) {
  const BatchI8Plan b = plan_batched_i8(S, N, C, nrep, K);
  const I8Plan &q = b.q;
  if (ws_bytes < b.total) {
    set_error("resample_vals_batched: workspace too small (%zu < %zu)", ws_bytes, b.total);
    return TXM_ERR_WORKSPACE;
  }
  TXM_REQUIRE(spec->ndat == N && spec->nrep == S * nrep, "resample_vals_batched: the sampler must span S * nrep replicates of N samples");
  TXM_REQUIRE(spec->rep0 >= 0 && spec->rep0 + spec->nrep <= ((int64_t)1 << 32), "resample_vals_batched: sampler replicates outside [0, 2^32)");
  TXM_REQUIRE((int64_t)b.n_chunks * q.n_rbg < ((int64_t)1 << 31) && S <= 65535, "resample_vals_batched: grid too large");
  txm_state_ptrs *tab = (txm_state_ptrs *)((char *)ws + b.off_tab);
  TXM_HIP(hipMemcpyAsync(tab, states_host, (size_t)S * sizeof(txm_state_ptrs), hipMemcpyHostToDevice, st));
  // the pre-pass block (pivots, window tables, guard flags, fallback lists of the S states): the caller's persistent buffer
  // (reused when prep_valid) or scratch in ws
  unsigned char *pb = (unsigned char *)ws + b.off_prep;
  bool have_tables = false;
  if (prep != nullptr) {
    if (prep_bytes < b.prep_total) {
      set_error("resample_vals_batched: prep buffer too small (%zu < %zu)", prep_bytes, b.prep_total);
      return TXM_ERR_WORKSPACE;
    }
    pb = (unsigned char *)prep;
    have_tables = prep_valid;
  }
  double *piv = (double *)(pb + b.prep_piv);
  if (!have_tables) {
    hipLaunchKernelGGL(pivot_batch_kernel, dim3((unsigned)(1 + C), (unsigned)S), dim3(256), 0, st, tab, ldx_s, N, C, piv);
    TXM_LAUNCH_CHECK();
  }
  I8State *states = (I8State *)((char *)ws + b.off_states);
  hipLaunchKernelGGL(i8_states_kernel, dim3((unsigned)cdiv(S, 64)), dim3(64), 0, st, tab, S, (unsigned char *)ws + b.off_state0,
                     pb + b.prep_state0, b.L, piv, C, counts, nrep, q.ntiles, out, K, (uint32_t)spec->rep0, states);
  TXM_LAUNCH_CHECK();
  bool vec_ok = ldx_s % 2 == 0;
  for (int64_t s = 0; s < S; ++s) vec_ok = vec_ok && (reinterpret_cast<uintptr_t>(states_host[s].x) & 15) == 0;
  I8Args a;
  a.states = states;
  a.S = S;
  a.x = vec_ok ? states_host[0].x : reinterpret_cast<const double *>((uintptr_t)8);  // (the pre-pass reads its alignment only)
  a.ldx_s = ldx_s; a.u = states_host[0].u; a.w = states_host[0].w; a.N = N; a.C = C; a.nrep = nrep;
  a.col0 = 0; a.C_call = C;
  a.part_summed = 1;  // (narrow instances always store digit-summed slots; the field is the wide instance's)
  a.counts = counts;
  a.k0 = (uint32_t)spec->seed;
  a.k1 = (uint32_t)(spec->seed >> 32);
  a.rep_base = (uint32_t)spec->rep0;
  a.ntiles = q.ntiles;
  a.last_tile_size = (uint32_t)(N - (q.ntiles - 1) * SM_T);
  a.pivot = piv; a.wtab = nullptr; a.stats = nullptr; a.nwin = q.nwin;
  a.part_x = nullptr; a.cpad = b.cpad; a.part_u = nullptr;
  a.y = nullptr; a.ldy_s = 0; a.ypivot = nullptr; a.ywtab = nullptr; a.yflag = nullptr; a.part_y = nullptr;
  a.n_chunks = b.n_chunks; a.n_rbg = q.n_rbg; a.tiles_per_chunk = b.tiles_per_chunk; a.win_tiles = q.win_tiles;
  a.nrep_pad = q.nrep_pad;
  a.wflag = nullptr; a.list = nullptr; a.n_list = nullptr; a.sub_tiles = q.sub_tiles;
  a.progress = nullptr;
  I8Args *bargs = (I8Args *)((char *)ws + b.off_bargs);
  hipLaunchKernelGGL(i8_batch_args_kernel, dim3((unsigned)cdiv(S, 64)), dim3(64), 0, st, a, S, bargs);
  TXM_LAUNCH_CHECK();
  a.batch_args = bargs;
  int rc = have_tables ? TXM_OK : launch_i8_prepass(a, K, st);
  if (rc != TXM_OK) return rc;
  rc = launch_resample_i8t(a, K, weighted, 0, st);
  if (rc != TXM_OK) return rc;
  // the windows the precision guard flags, state by state on grid axis z (a state with an empty list: its blocks exit)
  ResampleArgs f;
  f.x = states_host[0].x; f.ldx_s = ldx_s; f.u = states_host[0].u; f.w = states_host[0].w; f.N = N; f.C = C; f.nrep = nrep;
  f.freq = nullptr; f.counts = counts;
  f.k0 = a.k0; f.k1 = a.k1; f.rep_base = a.rep_base;
  f.ntiles = q.ntiles; f.last_tile_size = a.last_tile_size;
  f.pivot = piv; f.part_x = nullptr; f.part_u = nullptr;
  f.n_chunks = q.fb.n_chunks; f.n_rbg = q.fb.n_rbg; f.tiles_per_chunk = q.fb.tiles_per_chunk;
  f.nrep_pad = q.fb.nrep_pad; f.C_pad = q.fb.C_pad;
  f.list = nullptr; f.n_list = nullptr; f.sub_tiles = q.sub_tiles; f.col_off = 0; f.batch = nullptr; f.progress = nullptr;
  f.i8states = states;
  TXM_REQUIRE(q.fb.nrep_pad == q.nrep_pad, "resample_vals_batched: replicate padding of the two kernels differs");
  rc = run_listed(f, q.fb, K, weighted, st, S);
  if (rc != TXM_OK) return rc;
  const int bfin_summed = i8t_partials_summed(C, K) ? 1 : 0;  // (batched launches always run the fused narrow kernel)
#define TXM_I8_BFIN2(KK, CP)                                                                                        \
  hipLaunchKernelGGL((resample_finalize_i8_kernel<KK, CP>), dim3((unsigned)nrep, (unsigned)S), dim3(256), 0, st, nullptr, \
                     nullptr, q.nwin, nullptr, q.nrep_pad, nrep, C, nullptr, nullptr, (int64_t)0, C, nullptr, nullptr,      \
                     q.fb.n_chunks, q.fb.C_pad, nullptr, states, bfin_summed)
#define TXM_I8_BFIN(KK)                                  \
  do {                                                   \
    if (b.cpad == 16) TXM_I8_BFIN2(KK, 16);              \
    else if (b.cpad == 8) TXM_I8_BFIN2(KK, 8);           \
    else TXM_I8_BFIN2(KK, 4);                            \
  } while (0)
  switch (K) {
    case 2: TXM_I8_BFIN(2); break;
    case 3: TXM_I8_BFIN(3); break;
    case 4: TXM_I8_BFIN(4); break;
    case 5: TXM_I8_BFIN(5); break;
    case 6: TXM_I8_BFIN(6); break;
    case 7: TXM_I8_BFIN(7); break;
    default: TXM_I8_BFIN(8); break;
  }
#undef TXM_I8_BFIN
#undef TXM_I8_BFIN2
  TXM_LAUNCH_CHECK();
  if (info != nullptr) {
    hipLaunchKernelGGL(i8_info_kernel, dim3(1), dim3(64), 0, st, pb + b.prep_state0, b.L.p_stride, b.L.p_nlist, (int)S, q.nwin,
                       have_tables ? 1 : 0, info);
    TXM_LAUNCH_CHECK();
  }
  return TXM_OK;
}
}  // namespace txm

extern "C" size_t txm_resample_vals_batched_ws_bytes(int64_t S, int64_t N, int64_t C, int64_t nrep, int order) {
  if (S < 1 || N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  // (the larger of the two paths: which one a call takes may be forced per process, txm_set_resample_path)
  const size_t f = plan_batched(S, N, C, nrep, order + 1).total;
  if (!i8_supported(N, C, nrep, order + 1) || i8t_narrow_nq(C, order + 1) == 0) return f;
  const size_t i = plan_batched_i8(S, N, C, nrep, order + 1).total;
  return i > f ? i : f;
}

// the kernel a TXM_PATH_AUTO batched call takes (the batched counterpart of txm_resample_path): callers that keep a
// pre-pass block bind it only when the call really runs the int8 path
extern "C" int txm_resample_batched_path(int64_t S, int64_t N, int64_t C, int64_t nrep, int order) {
  if (S < 1 || N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return TXM_PATH_FP64;
  return use_i8_batched(S, N, C, nrep, order + 1) ? TXM_PATH_INT8 : TXM_PATH_FP64;
}

extern "C" size_t txm_resample_batched_prep_bytes(int64_t S, int64_t N, int64_t C, int64_t nrep, int order) {
  if (S < 1 || N < 1 || C < 1 || nrep < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  if (!i8_supported(N, C, nrep, order + 1) || i8t_narrow_nq(C, order + 1) == 0) return 0;
  return plan_batched_i8(S, N, C, nrep, order + 1).prep_total;
}

extern "C" int txm_resample_vals_batched(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N,
                                         int64_t C, int order, int64_t nrep, const int64_t *freq,
                                         const txm_sampler_spec *spec, const uint32_t *counts, double *out,
                                         void *ws, size_t ws_bytes, txm_stream stream) {
  return txm_resample_vals_batched_opts(states_host, S, ldx_s, N, C, order, nrep, freq, spec, counts, out, nullptr, ws, ws_bytes,
                                        stream);
}

extern "C" int txm_resample_vals_batched_opts(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N,
                                              int64_t C, int order, int64_t nrep, const int64_t *freq,
                                              const txm_sampler_spec *spec, const uint32_t *counts, double *out,
                                              const txm_resample_opts *opts, void *ws, size_t ws_bytes, txm_stream stream) {
  const int call_path = opts ? opts->path : TXM_PATH_AUTO;
  TXM_REQUIRE(call_path == TXM_PATH_AUTO || call_path == TXM_PATH_FP64 || call_path == TXM_PATH_INT8 || call_path == TXM_PATH_INT8_FUSED || call_path == TXM_PATH_INT8_TABLE,
              "resample_vals_batched: opts.path %d is not a path", call_path);
  TXM_REQUIRE(!(opts && (opts->y || opts->out_y)), "resample_vals_batched: no second sample matrix on the batched entry");
  TXM_REQUIRE(states_host && out && ws, "resample_vals_batched: null pointer");
  TXM_REQUIRE(S >= 1 && S <= 65535 && N >= 1 && C >= 1 && nrep >= 1, "resample_vals_batched: need S, N, C, nrep >= 1");
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "resample_vals_batched: order %d outside [0, %d]", order, TXM_MAX_ORDER);
  TXM_REQUIRE(ldx_s >= C, "resample_vals_batched: row pitch ldx_s < C");
  const bool explicit_ = freq != nullptr;
  TXM_REQUIRE(explicit_ != (spec != nullptr && counts != nullptr), "resample_vals_batched: give either freq or (spec, counts)");
  const bool weighted = states_host[0].w != nullptr;
  for (int64_t s = 0; s < S; ++s) {
    TXM_REQUIRE(states_host[s].x && states_host[s].u, "resample_vals_batched: state %lld has a null pointer", (long long)s);
    TXM_REQUIRE((states_host[s].w != nullptr) == weighted, "resample_vals_batched: weights for all states or for none");
  }
  const int K = order + 1;
  if (!explicit_ && use_i8_batched(S, N, C, nrep, K, call_path))
    return resample_batched_i8(states_host, S, ldx_s, N, C, K, nrep, weighted, spec, counts, out, opts ? opts->prep : nullptr,
                               opts ? opts->prep_bytes : 0, opts && opts->prep_valid != 0, opts ? opts->info : nullptr, ws, ws_bytes,
                               (hipStream_t)stream);
  if (opts && opts->info != nullptr) {
    hipLaunchKernelGGL(fp64_info_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, opts->info);
    TXM_LAUNCH_CHECK();
  }
  const BatchPlan b = plan_batched(S, N, C, nrep, K);
  if (ws_bytes < b.total) {
    set_error("resample_vals_batched: workspace too small (%zu < %zu)", ws_bytes, b.total);
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  txm_state_ptrs *tab = (txm_state_ptrs *)((char *)ws + b.off_tab);
  TXM_HIP(hipMemcpyAsync(tab, states_host, (size_t)S * sizeof(txm_state_ptrs), hipMemcpyHostToDevice, st));
  double *piv = (double *)((char *)ws + b.off_pivot);
  hipLaunchKernelGGL(pivot_batch_kernel, dim3((unsigned)(1 + C), (unsigned)S), dim3(256), 0, st, tab, ldx_s, N, C, piv);
  TXM_LAUNCH_CHECK();
  const ResamplePlan &p = b.one;
  ResampleArgs a;
  a.x = nullptr; a.ldx_s = ldx_s; a.u = nullptr; a.w = nullptr; a.N = N; a.C = C; a.nrep = nrep;
  a.freq = freq; a.counts = counts;
  a.k0 = a.k1 = 0;
  a.rep_base = 0;
  a.ntiles = p.ntiles;
  a.last_tile_size = (uint32_t)(N - (p.ntiles - 1) * SM_T);
  if (!explicit_) {
    TXM_REQUIRE(spec->ndat == N && spec->nrep == S * nrep, "resample_vals_batched: the sampler must span S * nrep replicates of N samples");
    TXM_REQUIRE(spec->rep0 >= 0 && spec->rep0 + spec->nrep <= ((int64_t)1 << 32), "resample_vals_batched: sampler replicates outside [0, 2^32)");
    a.k0 = (uint32_t)spec->seed;
    a.k1 = (uint32_t)(spec->seed >> 32);
    a.rep_base = (uint32_t)spec->rep0;
  }
  a.pivot = piv;
  a.part_x = (double *)((char *)ws + b.off_px);
  a.part_u = (double *)((char *)ws + b.off_pu);
  a.n_chunks = p.n_chunks; a.n_rbg = p.n_rbg; a.tiles_per_chunk = p.tiles_per_chunk;
  a.nrep_pad = p.nrep_pad; a.C_pad = p.C_pad;
  a.list = nullptr; a.n_list = nullptr; a.sub_tiles = 0; a.col_off = 0;
  a.batch = tab;
  a.progress = nullptr;
  TXM_K_SWITCH(K, return run_resample<KK>(a, p, weighted, explicit_, out, st, S));
  return TXM_OK;
}
