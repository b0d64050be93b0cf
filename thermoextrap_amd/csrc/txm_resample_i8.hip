// txm_resample_i8.hip -- the sample-level bootstrap contraction on the int8 matrix
// pipe (v_mfma_i32_32x32x32_i8), exact to FP64 accuracy by fixed-point slicing.
//
// Same sums as txm_resample.hip (cmomy.wrap_resample_vals as called from thermoextrap
// data.py:1803-1810, 1354-1366):
//        S1[r][c][j] = sum_i f[r][i] * w_i * du_i^j * dx_ic      S0[r][j] = sum_i f[r][i] * w_i * du_i^j
// Why another kernel: on gfx950 the FP64 MFMA shares the VALU datapath (tools/
// mfma_f64_peak4.hip), which caps the FP64 formulation near 64 TFLOP/s.  The int8 MFMA
// runs at 4.4 POP/s beside the VALU (tools/mfma_i8_probe2.hip), and here ONE operand is
// already an exact small integer: the bootstrap count f[r][i] (a u8 from the sampler).
// Only the data operand needs slicing, and that work is shared by all 64 replicates of
// a workgroup:
//
//   per window of 16 tiles (16384 samples) the pre-pass measures max|du|, max|w|,
//   max|dx_c|; inside a window every monomial m = (w/wmax)(du/dumax)^j * (dx_c/dxmax_c)
//   lies in [-1, 1] and  X = rint(m * 2^50)  is a 52-bit signed integer obtained with ONE
//   v_fma_f64 against the magic constant 1.5*2^52 (+ a per-byte bias of 0x80), whose
//   mantissa bytes -- after an XOR with 0x80 -- are seven signed base-256 digits
//   X = sum_i d_i 256^i, d_i in [-128, 127].  Then
//        sum_k f_k X_k = sum_i 256^i * (sum_k f_k d_ik)      exactly, in int32 accumulators,
//   flushed per window through a 7-term Horner in FP64 and the window's descale.
//   Rounding: one rint per monomial at 2^-51 of the WINDOW maximum (unbiased), against
//   2^-53 per element in FP64 -- far below the FP64 accumulation error of the sums.
//
// Workgroup = 4 waves x one wave per SIMD (512 VGPRs: 21 int32 accumulator tiles each),
// 64 replicates x all (K+1) column blocks [block j < K: power j of the 32 observables,
// block K: the K u-row sums].  Per sampler tile (1024 samples):
//   1. stage 3 of the sampler fills the WG's count tile   cnt[sample/4][rep][4 x u8]  (64 KiB)
//   2. 32 k-steps of 32 samples: every lane slices (1 column) x (4 samples) x (K powers)
//      of chunk s+1 into the other B buffer while the MFMAs of chunk s run.
// LDS: 64 KiB counts + 2 x (K+1)*7 KiB B chunks.
#include <type_traits>

#include "txm_resample_i8.h"
#include "txm_sampler.h"

namespace txm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int I8_BLOCK = 256;
constexpr int I8_CNT_BYTES = SM_T * I8_REPS;  // 65536
constexpr int I8_FRAG = 1024;                 // one 32 x 32 int8 MFMA operand
constexpr int I8_STEPS = SM_T / 32;           // k-steps per tile
constexpr int64_t I8_WIN_SAMPLES = (int64_t)I8_WIN_TILES * SM_T;

// ---------------------------------------------------------------------------
// pre-pass: per-window maxima -> scale / descale table
__global__ __launch_bounds__(256) void i8_window_kernel(const double *__restrict__ x, int64_t ldx,
                                                        const double *__restrict__ u,
                                                        const double *__restrict__ w, int64_t N,
                                                        int64_t C, const double *__restrict__ pivot,
                                                        double *__restrict__ wtab) {
  const int64_t win = blockIdx.x;
  const int64_t i0 = win * I8_WIN_SAMPLES;
  const int64_t i1 = (i0 + I8_WIN_SAMPLES < N) ? i0 + I8_WIN_SAMPLES : N;
  const int tid = threadIdx.x, c = tid & 31, r = tid >> 5;
  __shared__ double shx[256], shu[256], shw[256];
  double mx = 0.0, mu = 0.0, mw = 0.0;
  if (c < C) {
    const double px = pivot[1 + c];
    for (int64_t i = i0 + r; i < i1; i += 8) mx = fmax(mx, fabs(x[i * ldx + c] - px));
  }
  const double pu = pivot[0];
  for (int64_t i = i0 + tid; i < i1; i += 256) {
    mu = fmax(mu, fabs(u[i] - pu));
    if (w) mw = fmax(mw, fabs(w[i]));
  }
  shx[tid] = mx;
  shu[tid] = mu;
  shw[tid] = mw;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      shu[tid] = fmax(shu[tid], shu[tid + off]);
      shw[tid] = fmax(shw[tid], shw[tid + off]);
      if (off >= 32) shx[tid] = fmax(shx[tid], shx[tid + off]);  // keeps the column = tid & 31
    }
    __syncthreads();
  }
  double *wt = wtab + win * I8_WT_STRIDE;
  const double dumax = shu[0], wmax = w ? shw[0] : 1.0;
  if (tid == 0) {
    wt[I8_WT_INVDU] = dumax > 0.0 ? 1.0 / dumax : 0.0;
    wt[I8_WT_INVW] = wmax > 0.0 ? 1.0 / wmax : 0.0;
    double d = wmax;
    for (int j = 0; j < 10; ++j) {
      wt[I8_WT_DSP + j] = d;
      d *= dumax;
    }
  }
  if (tid < 32) {
    const double m = shx[tid];
    wt[I8_WT_SC + tid] = m > 0.0 ? 0x1p50 / m : 0.0;
    wt[I8_WT_DSC + tid] = m * 0x1p-50;
  }
}

// ---------------------------------------------------------------------------
// stage 3 of the sampler into the workgroup count tile (same stream as
// txm_sampler.h / oracle/philox_oracle.c; only the histogram layout differs)
template <bool ALL_VALID>
__device__ __forceinline__ void i8_tile_calls(uint32_t *cnt, uint32_t k0, uint32_t k1, uint32_t r, uint32_t t,
                                              uint32_t c, uint32_t n, uint32_t rl) {
  const uint32_t first = c * 12u;
  if (first >= n) return;
  const Philox4 o = philox4x32_10(c, t, r, 3u, k0, k1);
  const uint32_t nd = n - first;
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
    const uint32_t word = o.w[wi];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t f = (word >> (10 * k)) & 1023u;
      uint32_t inc = 1u << ((f & 3u) << 3);
      if (!ALL_VALID) inc = ((uint32_t)(wi * 3 + k) < nd) ? inc : 0u;
      atomicAdd(&cnt[(f >> 2) * I8_REPS + rl], inc);
    }
  }
}

__device__ __forceinline__ void i8_fill_full(const I8Args &a, uint32_t *cnt, int64_t rep0w, uint32_t rl0,
                                             int64_t t, int lane) {
#pragma unroll 1
  for (int p = 0; p < 8; ++p) {
    const int64_t ra = rep0w + 2 * p, rb = ra + 1;
    if (ra >= a.nrep) break;  // wave-uniform
    const uint32_t na = a.counts[(size_t)ra * a.ntiles + t];
    const uint32_t nb = rb < a.nrep ? a.counts[(size_t)rb * a.ntiles + t] : 0u;
    const uint32_t la = rl0 + 2u * p, lb = la + 1u;
    if (na >= 768u) i8_tile_calls<true>(cnt, a.k0, a.k1, (uint32_t)ra, (uint32_t)t, (uint32_t)lane, na, la);
    else i8_tile_calls<false>(cnt, a.k0, a.k1, (uint32_t)ra, (uint32_t)t, (uint32_t)lane, na, la);
    if (nb >= 768u) i8_tile_calls<true>(cnt, a.k0, a.k1, (uint32_t)rb, (uint32_t)t, (uint32_t)lane, nb, lb);
    else i8_tile_calls<false>(cnt, a.k0, a.k1, (uint32_t)rb, (uint32_t)t, (uint32_t)lane, nb, lb);
    const bool hb = lane >= 32;
    i8_tile_calls<false>(cnt, a.k0, a.k1, (uint32_t)(hb ? rb : ra), (uint32_t)t, 64u + ((uint32_t)lane & 31u),
                         hb ? nb : na, hb ? lb : la);
    const uint32_t nmax = na > nb ? na : nb;
    for (uint32_t c0 = 96u; c0 * 12u < nmax; c0 += 64u) {
      i8_tile_calls<false>(cnt, a.k0, a.k1, (uint32_t)ra, (uint32_t)t, c0 + (uint32_t)lane, na, la);
      i8_tile_calls<false>(cnt, a.k0, a.k1, (uint32_t)rb, (uint32_t)t, c0 + (uint32_t)lane, nb, lb);
    }
  }
}

// ---------------------------------------------------------------------------
// 4 fixed-point words -> 7 digit words (byte e of word i = digit i of sample e)
__device__ __forceinline__ void i8_slice4(const double (&r)[4], uint32_t (&W)[I8_NSL]) {
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint64_t b = (uint64_t)__double_as_longlong(r[e]);
    lo[e] = (uint32_t)b ^ 0x80808080u;
    hi[e] = ((uint32_t)(b >> 32) ^ 0x00008080u) - 0x00380000u;
  }
  const uint32_t a01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u);
  const uint32_t b01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);
  const uint32_t a23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u);
  const uint32_t b23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
  W[0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);
  W[1] = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
  W[2] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
  W[3] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
  const uint32_t c01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u);
  const uint32_t d01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x07030602u);
  const uint32_t c23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u);
  const uint32_t d23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x07030602u);
  W[4] = __builtin_amdgcn_perm(c23, c01, 0x05040100u);
  W[5] = __builtin_amdgcn_perm(c23, c01, 0x07060302u);
  W[6] = __builtin_amdgcn_perm(d23, d01, 0x05040100u);
}

// 1.5 * 2^52 + 0x80 in each of the six low mantissa bytes
constexpr double I8_MAGIC = 6755399441055744.0 + 141289400074368.0;

struct I8Chunk {
  double x[4], u[4], w[4];
};

template <int K, bool WEIGHTED>
__global__ __launch_bounds__(I8_BLOCK, 1) void resample_i8_kernel(const I8Args a) {
  constexpr int NBLK = K + 1, NFR = NBLK * I8_NSL;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *cnt = reinterpret_cast<uint32_t *>(lds);
  unsigned char *bb0 = lds + I8_CNT_BYTES;
  unsigned char *bb1 = bb0 + NFR * I8_FRAG;

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int n32 = lane & 31, half = lane >> 5;
  // slicing role: column c, sample group g (4 samples) of every chunk
  const int g = lane & 7, cslot = lane >> 3;
  const int c = wave * 8 + cslot;
  const bool c_ok = c < a.C;
  const int64_t cc = c_ok ? c : 0;
  // u-row role: power jsel (lanes of column slots 0 and 1)
  const int jsel = wave + 4 * cslot;
  const bool urow = cslot < 2 && jsel < K;
  const uint32_t woff = (uint32_t)(c * 32 + g * 4);
  const uint32_t uoff = (uint32_t)((K * I8_NSL * 32 + (urow ? jsel : 0)) * 32 + g * 4);
  const uint32_t roff = (uint32_t)(n32 * 32 + half * 16);
  const uint32_t aoff = (uint32_t)(half * 4 * I8_REPS + n32);
  // MFMA role: block `wave` for both replicate halves, plus half `wave & 1` of block 4 + wave / 2
  static_assert(NBLK >= 4 && NBLK <= 6, "every wave owns one full block; blocks 4, 5 are shared by wave pairs");
  const int sb = 4 + (wave >> 1), shh = wave & 1;
  const bool has_sh = sb < NBLK;

  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int chunk = (q / a.n_rbg) * 8 + xcd;
  const int rbg = q % a.n_rbg;
  const int64_t rep0 = (int64_t)rbg * I8_REPS;
  const int64_t t_begin = (int64_t)chunk * a.tiles_per_chunk;
  int64_t t_end = t_begin + a.tiles_per_chunk;
  if (t_end > a.ntiles) t_end = a.ntiles;

  const double pu = a.pivot[0];
  const double px = a.pivot[1 + cc];

  v16i acc[3][I8_NSL];
#pragma unroll
  for (int e = 0; e < 3; ++e)
#pragma unroll
    for (int i = 0; i < I8_NSL; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][i][r] = 0;

  // rows of the B buffers that are never written (columns >= C, u-row block rows >= K) stay zero
  for (int e = threadIdx.x; e < 2 * NFR * I8_FRAG / 16; e += I8_BLOCK)
    reinterpret_cast<uint4 *>(bb0)[e] = make_uint4(0, 0, 0, 0);

  double inv_du = 0.0, inv_w = 1.0, sc = 0.0;

  // ---- flush the int32 accumulators of one window into the FP64 partial sums ----
  // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
#define TXM_I8_FLUSH_TILES(E, BASE, STRIDE, DSC, VALID)                                     \
  do {                                                                                       \
    double *const base_ = (BASE);                                                            \
    const double dsc_ = (DSC);                                                               \
    const bool valid_ = (VALID);                                                             \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
      double v = (double)acc[E][I8_NSL - 1][r];                                              \
      _Pragma("unroll") for (int i = I8_NSL - 2; i >= 0; --i) v = fma(v, 256.0, (double)acc[E][i][r]); \
      if (valid_) base_[((r >> 2) * 8 + (r & 3)) * (STRIDE)] += v * dsc_;                    \
    }                                                                                        \
    _Pragma("unroll") for (int i = 0; i < I8_NSL; ++i) acc[E][i] = (v16i)(0);                \
  } while (0)
#define TXM_I8_FLUSH_UNIT(E, BLK, H)                                                         \
  do {                                                                                       \
    const int blk_ = (BLK);                                                                  \
    const size_t row0 = (size_t)chunk * a.nrep_pad + rep0 + 32 * (H) + 4 * half;             \
    if (blk_ < K) {                                                                          \
      TXM_I8_FLUSH_TILES(E, a.part_x + (row0 * I8_CPAD + n32) * K + blk_, I8_CPAD * K,       \
                         wt[I8_WT_DSP + blk_] * wt[I8_WT_DSC + n32], n32 < a.C);             \
    } else {                                                                                 \
      const int j_ = n32 < K ? n32 : 0;                                                      \
      TXM_I8_FLUSH_TILES(E, a.part_u + row0 * K + j_, K, wt[I8_WT_DSP + j_] * 0x1p-50, n32 < K); \
    }                                                                                        \
  } while (0)
  auto flush = [&](int64_t win) {
    const double *wt = a.wtab + win * I8_WT_STRIDE;
    // the asm MFMAs are invisible to the hazard recognizer: let the matrix pipe drain
    // before the VALU reads their destination registers
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    TXM_I8_FLUSH_UNIT(0, wave, 0);
    TXM_I8_FLUSH_UNIT(1, wave, 1);
    if (has_sh) TXM_I8_FLUSH_UNIT(2, sb, shh);
  };

  auto load_chunk = [&](int64_t wbase, int s, I8Chunk &r) {
    const int64_t i = wbase + s * 32 + g * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r.x[e] = a.x[(i + e) * a.ldx_s + cc];
      r.u[e] = a.u[i + e];
      if constexpr (WEIGHTED) r.w[e] = a.w[i + e];
    }
  };

  // ---- slice chunk `r` into B buffer `bb`; then refill r with chunk `snext` ----
  auto produce = [&](unsigned char *bb, I8Chunk &r, int64_t wbase, int snext) {
    double du[4], dx[4], p[4], p0[WEIGHTED ? 4 : 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      du[e] = (r.u[e] - pu) * inv_du;
      dx[e] = (r.x[e] - px) * sc;
      if constexpr (WEIGHTED) p[e] = p0[e] = r.w[e] * inv_w;
      else p[e] = 1.0;
    }
    load_chunk(wbase, snext, r);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (j > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] *= du[e];
      }
      double rr[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) rr[e] = fma(p[e], dx[e], I8_MAGIC);
      uint32_t W[I8_NSL];
      i8_slice4(rr, W);
      if (c_ok) {
#pragma unroll
        for (int i = 0; i < I8_NSL; ++i)
          *reinterpret_cast<uint32_t *>(bb + (j * I8_NSL + i) * I8_FRAG + woff) = W[i];
      }
    }
    // u-row block: lanes of column slot 0 slice w * du^wave, those of slot 1 (wave 0 only,
    // K = 5) w * du^4, which is the p the loop above ends with.  One pass per wave.
    if (urow) {
      double ps[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double v = WEIGHTED ? p0[e] : 1.0;
        if (wave >= 1) v *= du[e];
        if (wave >= 2) v *= du[e];
        if (wave >= 3) v *= du[e];
        ps[e] = (cslot == 1) ? p[e] : v;
      }
      double rr[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) rr[e] = fma(ps[e], 0x1p50, I8_MAGIC);
      uint32_t W[I8_NSL];
      i8_slice4(rr, W);
#pragma unroll
      for (int i = 0; i < I8_NSL; ++i) *reinterpret_cast<uint32_t *>(bb + i * I8_FRAG + uoff) = W[i];
    }
  };

  auto mfma_step = [&](const unsigned char *bb, int s) {
    const uint32_t *cw = cnt + s * (8 * I8_REPS) + aoff;
    v4i A0, A1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      A0[e] = (int)cw[e * I8_REPS];
      A1[e] = (int)cw[e * I8_REPS + 32];
    }
#pragma unroll
    for (int i = 0; i < I8_NSL; ++i) {
      const v4i B = *reinterpret_cast<const v4i *>(bb + (wave * I8_NSL + i) * I8_FRAG + roff);
      acc[0][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B, acc[0][i], 0, 0, 0);
      acc[1][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B, acc[1][i], 0, 0, 0);
    }
    // The third unit's tiles are pinned through asm constraints (1 to the AGPRs left over by
    // the 14 builtin tiles, 6 to VGPRs): 21 tiles do not fit the 256 AGPRs that the register
    // allocator gives builtin accumulators.  Its A operand is read from LDS a second time
    // (wave-uniform half) so that one straight-line sequence serves both halves and the
    // allocator has no reason to move the tiles: a compiler-made copy right after an asm
    // MFMA would read the destination before the matrix pipe has written it (the hazard
    // recognizer does not look into asm).  tools/check_i8_isa.py verifies the emitted code.
    if (has_sh) {
      v4i As;
#pragma unroll
      for (int e = 0; e < 4; ++e) As[e] = (int)cw[e * I8_REPS + 32 * shh];
#pragma unroll
      for (int i = 0; i < I8_NSL; ++i) {
        const v4i B = *reinterpret_cast<const v4i *>(bb + (sb * I8_NSL + i) * I8_FRAG + roff);
        if (i < 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[2][i]) : "v"(As), "v"(B));
        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[2][i]) : "v"(As), "v"(B));
      }
    }
  };

  // chunks are made of whole windows (tiles_per_chunk is a multiple of I8_WIN_TILES)
  for (int64_t win = t_begin / I8_WIN_TILES; win * I8_WIN_TILES < t_end; ++win) {
    {
      const double *wt = a.wtab + win * I8_WT_STRIDE;
      inv_du = wt[I8_WT_INVDU];
      if constexpr (WEIGHTED) inv_w = wt[I8_WT_INVW];
      sc = wt[I8_WT_SC + cc];
    }
    int64_t tt_end = (win + 1) * I8_WIN_TILES;
    if (tt_end > t_end) tt_end = t_end;
#pragma unroll 1
    for (int64_t t = win * I8_WIN_TILES; t < tt_end; ++t) {
    const int64_t i_tile = t * SM_T;
    const uint32_t tsize = (t == a.ntiles - 1) ? a.last_tile_size : (uint32_t)SM_T;
    int64_t wbase = i_tile;
    if (wbase > a.N - SM_T) wbase = a.N - SM_T;          // the last tile slides its window back
    const uint32_t shift = (uint32_t)(i_tile - wbase);

    I8Chunk r0;
    load_chunk(wbase, 0, r0);

    // ---- stage 3 of the sampler: the workgroup's 64 x 1024 count tile -------
    for (int e = threadIdx.x; e < I8_CNT_BYTES / 16; e += I8_BLOCK)
      reinterpret_cast<uint4 *>(cnt)[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    {
      const int64_t rep0w = rep0 + wave * 16;
      const uint32_t rl0 = (uint32_t)wave * 16u;
      if (tsize == (uint32_t)SM_T) {
        i8_fill_full(a, cnt, rep0w, rl0, t, lane);
      } else {
        for (int rr = 0; rr < 16; ++rr) {
          const int64_t r = rep0w + rr;
          if (r >= a.nrep) break;  // wave-uniform
          const uint32_t n = a.counts[(size_t)r * a.ntiles + t];
          sampler_fine_tile(a.k0, a.k1, (uint32_t)r, (uint32_t)t, n, tsize, lane, [&](uint32_t off0) {
            const uint32_t off = off0 + shift;
            atomicAdd(&cnt[(off >> 2) * I8_REPS + rl0 + (uint32_t)rr], 1u << ((off & 3u) << 3));
          });
        }
      }
    }
    __syncthreads();

    // ---- contraction: chunk s on the matrix pipe, chunk s+1 through the slicer ----
    produce(bb0, r0, wbase, 1);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < I8_STEPS; s += 2) {
      mfma_step(bb0, s);
      produce(bb1, r0, wbase, s + 2 < I8_STEPS ? s + 2 : I8_STEPS - 1);
      __syncthreads();
      mfma_step(bb1, s + 1);
      if (s + 2 < I8_STEPS) produce(bb0, r0, wbase, s + 3 < I8_STEPS ? s + 3 : I8_STEPS - 1);
      __syncthreads();
    }
  }
    flush(win);
  }
}

// ---------------------------------------------------------------------------
bool i8_supported(int64_t N, int64_t C, int64_t nrep, int K) {
  (void)nrep;
  return N >= SM_T && C >= 1 && C <= I8_CPAD && K >= 3 && K <= 5;
}

int launch_resample_i8(const I8Args &a, int K, bool weighted, hipStream_t st) {
  hipLaunchKernelGGL(i8_window_kernel, dim3((unsigned)a.nwin), dim3(256), 0, st, a.x, a.ldx_s, a.u, a.w,
                     a.N, a.C, a.pivot, a.wtab);
  TXM_LAUNCH_CHECK();
  const dim3 grid((unsigned)(a.n_chunks * a.n_rbg)), block(I8_BLOCK);
  const size_t lds = (size_t)I8_CNT_BYTES + 2u * (size_t)(K + 1) * I8_NSL * I8_FRAG;
#define TXM_I8_LAUNCH(KK)                                                                      \
  do {                                                                                         \
    if (weighted) {                                                                            \
      TXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_i8_kernel<KK, true>),  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      hipLaunchKernelGGL((resample_i8_kernel<KK, true>), grid, block, lds, st, a);             \
    } else {                                                                                   \
      TXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_i8_kernel<KK, false>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      hipLaunchKernelGGL((resample_i8_kernel<KK, false>), grid, block, lds, st, a);            \
    }                                                                                          \
  } while (0)
  switch (K) {
    case 3: TXM_I8_LAUNCH(3); break;
    case 4: TXM_I8_LAUNCH(4); break;
    case 5: TXM_I8_LAUNCH(5); break;
    default: set_error("resample_i8: order out of range"); return TXM_ERR_INVALID;
  }
#undef TXM_I8_LAUNCH
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

}  // namespace txm
