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
//   per window of 64 tiles (65536 samples) the pre-pass measures max|du|, max|w|,
//   max|dx_c|; inside a window every monomial m = (w/wmax)(du/dumax)^j * (dx_c/dxmax_c)
//   lies in [-1, 1] and  X = rint(m * 2^50)  is a 52-bit signed integer obtained with ONE
//   v_fma_f64 against the magic constant 1.5*2^52 (+ a per-byte bias of 0x80), whose
//   mantissa bytes -- after an XOR with 0x80 -- are seven signed base-256 digits
//   X = sum_i d_i 256^i, d_i in [-128, 127].  Then
//        sum_k f_k X_k = sum_i 256^i * (sum_k f_k d_ik)      exactly, in int32 accumulators,
//   flushed per window into one FP64 partial sum per digit (x 256^i x the window descale).
//   Rounding: one rint per monomial at 2^-51 of the WINDOW maximum (unbiased), against
//   2^-53 per element in FP64 -- far below the FP64 accumulation error of the sums.
//
// Workgroup = 8 waves, two per SIMD (256 registers each, 10 int32 accumulator tiles),
// 64 replicates x all operand rows of the 32 observables (layout: see the kernel); the K
// u-row monomials (dx = 1) pack their digits into the columns 8 j + i of ceil(8K/32) further
// fragments.  Per sampler tile (1024 samples):
//   1. stage 3 of the sampler fills the WG's count tile   cnt[rep][sample/4 (+pad)][4 x u8]  (65 KiB)
//   2. 32 k-steps of 32 samples: every lane slices (1 column) x (2 samples) x (K powers)
//      of chunk s+1 into the other B buffer, the MFMAs of chunk s issued between the powers.
// LDS: 65 KiB counts + 2 x (6K + K + ceil(8K/32)) KiB of B chunks (139 KiB at order 4).
#include "txm_resample_i8.h"
#include "txm_sampler.h"

namespace txm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int I8_BLOCK = 512;
constexpr int I8_WAVES = I8_BLOCK / 64;
constexpr int I8_REPS_WAVE = I8_REPS / I8_WAVES;  // replicates whose counts one wave draws
// count tile: cnt[rep][260 words], word g = the u8 counts of samples 4g..4g+3.  The 4-word pad
// spreads one replicate's words (the scatter of a wave) and one word of 32 replicates (the A
// operand read of a wave) over all LDS banks.
constexpr int I8_CNT_ROW = SM_T / 4 + 4;          // words per replicate row
constexpr int I8_CNT_BYTES = I8_REPS * I8_CNT_ROW * 4;  // 66560
constexpr int I8_FRAG = 1024;                     // one 32 x 32 int8 MFMA operand
constexpr int I8_STEPS = SM_T / 32;               // k-steps per tile

// ---------------------------------------------------------------------------
// pre-pass: per-window maxima -> scale / descale table, and the precision guard (txm_resample_i8.h).
//
// i8_stats_kernel: one block per SUB-BLOCK of <= 16 tiles (a 256-tile window alone would give the chip 1.5 blocks
// per CU).  Next to the maxima it takes the mean of |w du^J dx_c| (J = the top power) over groups of samples and
// keeps the smallest group mean: a robust "typical monomial".  typ must not be the plain mean: ONE 1e4-sigma sample
// owns the mean of du^4 over its window, and the replicates that do not draw it (37 %) see only the other samples --
// which the window's scale would have rounded away.  Groups: a window of several sub-blocks uses 4 groups per
// sub-block and column (quarters of its rows); a window that is one sub-block uses 64 (quarter x row phase).
// i8_table_kernel: one block per window: combines its sub-blocks, writes the window table entry and the flag.
constexpr int I8_ST_MAXDU = 0, I8_ST_MAXW = 1, I8_ST_UMIN = 2, I8_ST_BADU = 3, I8_ST_MAXX = 4, I8_ST_GMIN = 36,
              I8_ST_BADX = 68, I8_ST_STRIDE = 100;
constexpr int I8_STAT_TILES = 16;  // tiles per sub-block

template <bool VEC2>
__global__ __launch_bounds__(256) void i8_stats_kernel(const double *__restrict__ x, int64_t ldx,
                                                       const double *__restrict__ u,
                                                       const double *__restrict__ w, int64_t N, int64_t C,
                                                       int64_t col0, int64_t sub_samples, int fine_groups,
                                                       const double *__restrict__ pivot, int J,
                                                       double *__restrict__ stats) {
  // thread = (column pair cp, row phase r): 16 lanes read one 256-byte row with 16-byte loads
  const int64_t sb = blockIdx.x;
  const int64_t i0 = sb * sub_samples;
  const int64_t i1 = (i0 + sub_samples < N) ? i0 + sub_samples : N;
  const int tid = threadIdx.x, cp = tid & 15, r = tid >> 4;
  __shared__ double shq[4][2][256];  // per-thread quarter sums of the two columns
  __shared__ int shqn[4][256];
  __shared__ double shx[2][256], shg[2][256], shu[256], shw[256], shs[256];
  __shared__ int shn[256];
  __shared__ int badx[32], badu;
  if (tid < 32) badx[tid] = 0;
  if (tid == 0) badu = 0;
  __syncthreads();
  const double pu = pivot[0];
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  double mx[2] = {0.0, 0.0}, gmin[2] = {kInf, kInf}, mu = 0.0, mw = 0.0;
  bool bx[2] = {false, false}, bu = false;
  const int c0 = 2 * cp;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    shq[g][0][tid] = shq[g][1][tid] = 0.0;
    shqn[g][tid] = 0;
  }
  if (c0 < C) {
    const bool two = c0 + 1 < C;
    const double px0 = pivot[1 + col0 + c0], px1 = two ? pivot[1 + col0 + c0 + 1] : 0.0;
    const int64_t gs = sub_samples / 4;  // rows per quarter (sub_samples is a multiple of 4096)
    for (int g = 0; g < 4; ++g) {
      const int64_t ia = i0 + g * gs;
      const int64_t ib = (ia + gs < i1) ? ia + gs : i1;
      double s0 = 0.0, s1 = 0.0;
      int n = 0;
      auto body = [&](double x0, double x1, double ui, double wi) {
        double a = fabs(wi);
        const double du = fabs(ui - pu);
        for (int q = 0; q < J; ++q) a *= du;
        const double v0 = fabs(x0 - px0), v1 = fabs(x1 - px1);
        bx[0] |= !(v0 <= 1.7976931348623157e308);
        bx[1] |= !(v1 <= 1.7976931348623157e308);
        mx[0] = fmax(mx[0], v0);
        mx[1] = fmax(mx[1], v1);
        s0 += a * v0;
        s1 += a * v1;
        ++n;
      };
      auto load = [&](int64_t i, double &x0, double &x1, double &ui, double &wi) {
        x1 = px1;
        if constexpr (VEC2) {
          if (two) {
            const double2 t2 = *reinterpret_cast<const double2 *>(x + i * ldx + col0 + c0);
            x0 = t2.x;
            x1 = t2.y;
          } else {
            x0 = x[i * ldx + col0 + c0];
          }
        } else {
          x0 = x[i * ldx + col0 + c0];
          if (two) x1 = x[i * ldx + col0 + c0 + 1];
        }
        ui = u[i];
        wi = w ? w[i] : 1.0;
      };
      int64_t i = ia + r;
      constexpr int UNR = 4;  // rows in flight per lane
      for (; i + (UNR - 1) * 16 < ib; i += UNR * 16) {
        double x0[UNR], x1[UNR], ui[UNR], wi[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) load(i + q * 16, x0[q], x1[q], ui[q], wi[q]);
#pragma unroll
        for (int q = 0; q < UNR; ++q) body(x0[q], x1[q], ui[q], wi[q]);
      }
      for (; i < ib; i += 16) {
        double x0, x1, ui, wi;
        load(i, x0, x1, ui, wi);
        body(x0, x1, ui, wi);
      }
      shq[g][0][tid] = s0;
      shq[g][1][tid] = s1;
      shqn[g][tid] = n;
      if (n > 0) {
        gmin[0] = fmin(gmin[0], s0 / (double)n);
        gmin[1] = fmin(gmin[1], s1 / (double)n);
      }
    }
  }
  double su = 0.0;
  int nu = 0;
  for (int64_t i = i0 + tid; i < i1; i += 256) {
    const double v = fabs(u[i] - pu);
    bu |= !(v <= 1.7976931348623157e308);
    mu = fmax(mu, v);
    double a = 1.0;
    if (w) {
      const double vw = fabs(w[i]);
      bu |= !(vw <= 1.7976931348623157e308);
      mw = fmax(mw, vw);
      a = vw;
    }
    for (int q = 0; q < J; ++q) a *= v;
    su += a;
    ++nu;
  }
  if (bx[0]) atomicOr(&badx[c0], 1);
  if (bx[1]) atomicOr(&badx[c0 + 1], 1);
  if (bu) atomicOr(&badu, 1);
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    shx[v][tid] = mx[v];
    shg[v][tid] = gmin[v];
  }
  shu[tid] = mu;
  shw[tid] = mw;
  shs[tid] = su;
  shn[tid] = nu;
  __syncthreads();
  if (!fine_groups) {
    // coarse groups: the 16 row phases of a quarter together (fixed order) -> 4 group means per column
#pragma unroll
    for (int v = 0; v < 2; ++v) shg[v][tid] = kInf;
    __syncthreads();
    if (tid < 128) {
      const int g = tid >> 5, col = tid & 31, ccp = col >> 1, v = col & 1;
      double sg = 0.0;
      int ng = 0;
      for (int rr = 0; rr < 16; ++rr) {
        sg += shq[g][v][rr * 16 + ccp];
        ng += shqn[g][rr * 16 + ccp];
      }
      // park the group mean where the tree below (which keeps tid & 15 = column pair) finds it: row phase g
      if (ng > 0) shg[v][g * 16 + ccp] = sg / (double)ng;
    }
    __syncthreads();
  }
  // u row: groups of 4 adjacent threads (fine: 64 groups) or of 64 (coarse: 4 groups) -> group means -> their minimum
  double umin = kInf;
  const int gw = fine_groups ? 4 : 64;
  if (tid < 256 / gw) {
    double sg = 0.0;
    int ng = 0;
    for (int k = 0; k < gw; ++k) {
      sg += shs[gw * tid + k];
      ng += shn[gw * tid + k];
    }
    if (ng > 0) umin = sg / (double)ng;
  }
  __syncthreads();
  shs[tid] = umin;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      shu[tid] = fmax(shu[tid], shu[tid + off]);
      shw[tid] = fmax(shw[tid], shw[tid + off]);
      shs[tid] = fmin(shs[tid], shs[tid + off]);
      if (off >= 16) {  // keeps the column pair = tid & 15
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          shx[v][tid] = fmax(shx[v][tid], shx[v][tid + off]);
          shg[v][tid] = fmin(shg[v][tid], shg[v][tid + off]);
        }
      }
    }
    __syncthreads();
  }
  double *st = stats + sb * I8_ST_STRIDE;
  if (tid == 0) {
    st[I8_ST_MAXDU] = shu[0];
    st[I8_ST_MAXW] = shw[0];
    st[I8_ST_UMIN] = shs[0];
    st[I8_ST_BADU] = (double)badu;
  }
  if (tid < 32) {  // column tid = 2 * (tid >> 1) + (tid & 1)
    st[I8_ST_MAXX + tid] = shx[tid & 1][tid >> 1];
    st[I8_ST_GMIN + tid] = shg[tid & 1][tid >> 1];
    st[I8_ST_BADX + tid] = (double)badx[tid];
  }
}

__global__ __launch_bounds__(64) void i8_table_kernel(const double *__restrict__ stats, int nsub, int64_t nsub_total,
                                                      int64_t N, int64_t C, int64_t win_samples, bool weighted,
                                                      int J, double *__restrict__ wtab,
                                                      uint32_t *__restrict__ wflag) {
  const int64_t win = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t i0 = win * win_samples;
  const int64_t i1 = (i0 + win_samples < N) ? i0 + win_samples : N;
  const double kInf = __longlong_as_double(0x7ff0000000000000ll);
  __shared__ int flagged;
  if (tid == 0) flagged = 0;
  __syncthreads();
  double dumax = 0.0, wmaxv = 0.0, umin = kInf, mx = 0.0, gmin = kInf;
  bool badu = false, badx = false;
  for (int k = 0; k < nsub; ++k) {
    const int64_t sb = win * nsub + k;
    if (sb >= nsub_total) break;
    const double *st = stats + sb * I8_ST_STRIDE;
    dumax = fmax(dumax, st[I8_ST_MAXDU]);
    wmaxv = fmax(wmaxv, st[I8_ST_MAXW]);
    umin = fmin(umin, st[I8_ST_UMIN]);
    badu |= st[I8_ST_BADU] != 0.0;
    if (tid < 32) {
      mx = fmax(mx, st[I8_ST_MAXX + tid]);
      gmin = fmin(gmin, st[I8_ST_GMIN + tid]);
      badx |= st[I8_ST_BADX + tid] != 0.0;
    }
  }
  const double wmax = weighted ? wmaxv : 1.0;
  double *wt = wtab + win * I8_WT_STRIDE;
  // guard: scale of the top power against the typical monomial (NaN / inf compare false: those windows keep
  // the int8 path, whose poisoned descale reproduces the FP64 kernel's non-finite output)
  double mtop = wmax;
  for (int q = 0; q < J; ++q) mtop *= dumax;
  const double theta = I8_GUARD * sqrt((double)(i1 - i0));
  if (tid == 0) {
    wt[I8_WT_INVDU] = dumax > 0.0 ? 1.0 / dumax : 0.0;
    wt[I8_WT_INVW] = wmax > 0.0 ? 1.0 / wmax : 0.0;
    double d = badu ? __longlong_as_double(0x7ff8000000000000ll) : wmax;
    for (int j = 0; j < 10; ++j) {
      wt[I8_WT_DSP + j] = d;
      d *= dumax;
    }
    if (mtop > theta * umin) atomicOr(&flagged, 1);
  }
  if (tid < 32) {
    wt[I8_WT_SC + tid] = mx > 0.0 ? 0x1p50 / mx : 0.0;
    wt[I8_WT_DSC + tid] = badx ? __longlong_as_double(0x7ff8000000000000ll) : mx * 0x1p-50;
    if (tid < C && mtop * mx > theta * gmin) atomicOr(&flagged, 1);
  }
  __syncthreads();
  if (tid == 0) wflag[win] = (uint32_t)flagged;
}

// flagged windows -> sorted list of tile runs for the FP64 kernel (one block: the scan keeps the order, so
// the fallback's partial sums are added in a fixed order and results stay bitwise reproducible)
__global__ __launch_bounds__(256) void i8_list_kernel(const uint32_t *__restrict__ wflag, int64_t nwin,
                                                      int64_t win_tiles, int sub_tiles,
                                                      uint32_t *__restrict__ list,
                                                      uint32_t *__restrict__ n_list) {
  __shared__ uint32_t sh[256];
  __shared__ uint32_t running;
  const int tid = threadIdx.x;
  if (tid == 0) running = 0;
  __syncthreads();
  const int per = (int)(win_tiles / sub_tiles);
  for (int64_t base = 0; base < nwin; base += 256) {
    const int64_t wi = base + tid;
    const uint32_t f = (wi < nwin && wflag[wi] != 0u) ? 1u : 0u;
    sh[tid] = f;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive scan
      const uint32_t v = tid >= off ? sh[tid - off] : 0u;
      __syncthreads();
      sh[tid] += v;
      __syncthreads();
    }
    const uint32_t pos = running + sh[tid] - f;
    if (f)
      for (int k = 0; k < per; ++k) list[(size_t)pos * per + k] = (uint32_t)(wi * win_tiles + (int64_t)k * sub_tiles);
    __syncthreads();
    if (tid == 255) running += sh[255];
    __syncthreads();
  }
  if (tid == 0) {
    n_list[0] = running * (uint32_t)per;
    n_list[1] = running;  // flagged windows of this column group
  }
}

// ---------------------------------------------------------------------------
// stage 3 of the sampler into the workgroup count tile (same stream as
// txm_sampler.h / oracle/philox_oracle.c; only the histogram layout differs)
template <bool ALL_VALID>
__device__ __forceinline__ void i8_tile_calls(uint32_t *cnt, uint32_t k0, uint32_t k1, uint32_t r, uint32_t t,
                                              uint32_t c, uint32_t n, uint32_t rl) {
  const uint32_t first = c * 12u;
  // ALL_VALID: the caller guarantees 12 (c + 1) <= n, so there is no branch and the two
  // replicates of a pair share a basic block: their Philox chains interleave
  if (!ALL_VALID && first >= n) return;
  // call index in the second counter word: the tile's and the replicate's share of rounds 1-3 is wave-uniform
  const Philox4 o = philox4x32_10(t, c, r, 3u, k0, k1);
  const uint32_t nd = n - first;
  uint32_t *row = cnt + rl * I8_CNT_ROW;
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
    const uint32_t word = o.w[wi];
    // byte lane of every field at once: keep bits {0,1} of the three fields, so that a plain shift leaves
    // 8 * (f & 3) in the five bits the shifter reads and zeros below them.  The two opaque values keep this
    // selection (mask once; bfe + lshl_add for the address): 4 instead of 6 vector instructions per draw.
    uint32_t lo2 = word & 0x00300C03u;
    asm volatile("" : "+v"(lo2));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint32_t q = __builtin_amdgcn_ubfe(word, 10 * k + 2, 8);  // f >> 2: the word of the row
      asm volatile("" : "+v"(q));
      uint32_t inc = 1u << (((k == 0) ? (lo2 << 3) : (lo2 >> (10 * k - 3))) & 31u);
      if (!ALL_VALID) inc = ((uint32_t)(wi * 3 + k) < nd) ? inc : 0u;
      atomicAdd(&row[q], inc);
    }
  }
}

// `wcnt`: lane i < I8_REPS_WAVE holds counts[rep0w + i][t] (0 past the last replicate), fetched one tile ahead
__device__ __forceinline__ void i8_fill_full(const I8Args &a, uint32_t *cnt, int64_t rep0w, uint32_t rl0,
                                             int64_t t, int lane, uint32_t wcnt) {
#pragma unroll 1
  for (int p = 0; p < I8_REPS_WAVE / 2; ++p) {
    const int64_t ra = rep0w + 2 * p, rb = ra + 1;
    if (ra >= a.nrep) break;  // wave-uniform
    // The SIMD issues oldest-first: its second wave (waves 4..7) trails through the fill and finishes it alone, bound
    // by the latency of its own Philox chain, while the first waits at the barrier (phase timing: 26 % vs 19 % of the
    // kernel in the fill).  Alternating the second wave's priority per replicate pair keeps both in flight to the end:
    // 1.5 % of the kernel (same-box A/B: 205.4 -> 202.3 ms).  The same trick inside the k-steps loses: per power +3 %,
    // per k-step +0.6 %.
    if (rl0 >= (uint32_t)(I8_REPS / 2)) {
      if (p & 1) __builtin_amdgcn_s_setprio(0);
      else __builtin_amdgcn_s_setprio(2);
    }
    const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)wcnt, 2 * p);
    const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)wcnt, 2 * p + 1);
    const uint32_t la = rl0 + 2u * p, lb = la + 1u;
    const uint32_t sa = a.rep_base + (uint32_t)ra, sb = a.rep_base + (uint32_t)rb;  // replicates of the stream
    if (na >= 768u) i8_tile_calls<true>(cnt, a.k0, a.k1, sa, (uint32_t)t, (uint32_t)lane, na, la);
    else i8_tile_calls<false>(cnt, a.k0, a.k1, sa, (uint32_t)t, (uint32_t)lane, na, la);
    if (nb >= 768u) i8_tile_calls<true>(cnt, a.k0, a.k1, sb, (uint32_t)t, (uint32_t)lane, nb, lb);
    else i8_tile_calls<false>(cnt, a.k0, a.k1, sb, (uint32_t)t, (uint32_t)lane, nb, lb);
    const bool hb = lane >= 32;
    i8_tile_calls<false>(cnt, a.k0, a.k1, hb ? sb : sa, (uint32_t)t, 64u + ((uint32_t)lane & 31u),
                         hb ? nb : na, hb ? lb : la);
    const uint32_t nmax = na > nb ? na : nb;
    for (uint32_t c0 = 96u; c0 * 12u < nmax; c0 += 64u) {
      i8_tile_calls<false>(cnt, a.k0, a.k1, sa, (uint32_t)t, c0 + (uint32_t)lane, na, la);
      i8_tile_calls<false>(cnt, a.k0, a.k1, sb, (uint32_t)t, c0 + (uint32_t)lane, nb, lb);
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// ---------------------------------------------------------------------------
// 2 fixed-point words -> 4 words holding the 7 digit pairs: T[0] = {digit 0 | digit 1},
// T[1] = {2 | 3}, T[2] = {4 | 5}, T[3] = {6 | -}; each 16-bit half = (sample 0, sample 1).
// Digits 0..5 are the mantissa bytes XOR 0x80 (the bytes carry a +128 bias from the magic
// constant).  Digit 6 is left as the raw exponent-adjacent byte 0x38 + d6: the constant 56
// is taken out at flush time as 56 * (sum of counts), which the sampler knows exactly.
constexpr int I8_D6_BIAS = 0x38;
__device__ __forceinline__ void i8_slice2(double r0, double r1, uint32_t (&T)[4]) {
  const uint64_t b0 = (uint64_t)__double_as_longlong(r0), b1 = (uint64_t)__double_as_longlong(r1);
  const uint32_t l0 = (uint32_t)b0, l1 = (uint32_t)b1, h0 = (uint32_t)(b0 >> 32), h1 = (uint32_t)(b1 >> 32);
  T[0] = __builtin_amdgcn_perm(l1, l0, 0x05010400u) ^ 0x80808080u;
  T[1] = __builtin_amdgcn_perm(l1, l0, 0x07030602u) ^ 0x80808080u;
  T[2] = __builtin_amdgcn_perm(h1, h0, 0x05010400u) ^ 0x80808080u;
  T[3] = __builtin_amdgcn_perm(h1, h0, 0x07030602u);
}

// 1.5 * 2^52 + 0x80 in each of the six low mantissa bytes
constexpr double I8_MAGIC = 6755399441055744.0 + 141289400074368.0;

struct I8Chunk {
  double x[2];
};

// K = order + 1 fixes the layout of the partial sums; one launch slices the JN powers
// J0 .. J0 + JN - 1 (orders above 4 take two launches, five powers and the rest: 10 accumulator tiles per wave is what the
// 256-register budget leaves room for).
//
// B operand in LDS (per buffer):
//   pair row (jj, w), w = 0..2: the digits 2w and 2w+1 of power jj.  Each slicing lane stores the
//     ONE dword {d_2w(s0), d_2w(s1), d_2w+1(s0), d_2w+1(s1)} of its sample pair: [32 columns][16 pairs]
//     dwords = 2 KiB, the pair index rotated by 4 * ((column >> 1) & 3) so that the consumer's two
//     16-byte reads per row hit all banks.  The consumer de-interleaves with 8 v_perm_b32 into the
//     two MFMA operands.  (LDS writes cost 4.5 CU-cycles per wave instruction whatever their width
//     up to 32 bits, reads half of that: tools/lds_rate_probe.hip -- so the interleave is undone on
//     the read side.)
//   plain fragments [32 columns][32 k-bytes] = 1 KiB: digit 6 of every power (b16 stores), then the
//     packed u-row fragments (column 8 jj + i = digit i of the u-row power jj).
//
// PK > 1 (narrow states, C <= 32 / PK observables): the 32 B-operand columns carry PK powers per observable -- column
// slot n = jq * CP + c (CP = 32 / PK) holds w * du^(jq + PK * jj) * dx_c in row set jj -- so that a state with 8
// observables and order 4 needs 2 row sets instead of 5 and no column of an MFMA is wasted.  JN then counts row sets.
template <int K, int J0, int JN, bool WEIGHTED, int PK = 1>
__global__ __launch_bounds__(I8_BLOCK) void resample_i8_kernel(const I8Args a) {
  static_assert(JN >= 1 && JN <= 5 && (PK == 1 ? J0 + JN <= K : (J0 == 0 && (JN - 1) * PK < K)), "power range");
  static_assert(PK == 1 || PK == 2 || PK == 4, "powers per observable column");
  constexpr int CP = 32 / PK;                                     // observable columns per launch
  constexpr int KL = PK == 1 ? JN : (K < JN * PK ? K : JN * PK);  // powers of this launch (= u-row monomials)
#ifndef TXM_I8_XD
#define TXM_I8_XD 8
#endif
  constexpr int XD = JN <= 2 ? TXM_I8_XD : 1;  // x-chunk prefetch depth in k-steps (a power of two, even or 1)
  constexpr int NPAIR = 3 * JN;
  constexpr int UF = (8 * KL + 31) / 32;
  constexpr int NFRG = JN + UF;
  constexpr int PAIR_B = 2048, FRAG0 = NPAIR * PAIR_B, BUF = FRAG0 + NFRG * I8_FRAG;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *cnt = reinterpret_cast<uint32_t *>(lds);
  // One or two powers per launch: a k-step is all latency (operand reads after the barrier, stores before the next
  // one), so TWO 32-sample chunks go between barriers -- each B buffer holds a pair of chunks (NB = 2), chunk c in
  // buffer (c >> 1) & 1, half c & 1, and the step that contracts chunk c slices chunk c + 2 into the other buffer.
  constexpr int NB = JN <= 2 ? 2 : 1;
  unsigned char *bb0 = lds + I8_CNT_BYTES;
  unsigned char *bb1 = bb0 + NB * BUF;
  uint32_t *fsum = reinterpret_cast<uint32_t *>(bb1 + NB * BUF);  // [64] draws per replicate in the window
  // the tile's scaled u deviations (u - pu) / max|u - pu| and weights w / max|w|: loaded once per tile by the
  // whole workgroup instead of once per k-step by every lane
  uint32_t *cntlds = fsum + I8_REPS;  // [64] the next tile's draw counts, parked here while the k-steps run
  // unweighted: two u tiles (the next tile's u is staged while this one is contracted); weighted: one u + one w tile
  double *utile = reinterpret_cast<double *>(cntlds + I8_REPS);
  double *utile_nxt = utile + SM_T;
  double *wtile = utile + SM_T;

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int n32 = lane & 31, half = lane >> 5;
  // slicing role: column c, sample pair g2 of every chunk (a half-wave = 16 pairs x 2 columns = 32
  // consecutive dwords of a pair row)
  const int cslot = lane >> 4;
  const int c = wave * 4 + cslot;
  // the lane at position (lane & 15) of its column's 16-dword run slices the pair that the rotation puts there, so
  // that a wave's 64 dwords of a pair row are simply base + 4 * lane: ds_write_addtid_b32
  const int g2 = ((lane & 15) - 4 * ((c >> 1) & 3)) & 15;
  const int cq = c % CP;               // observable column of this slot
  const int jq = wave * 4 / CP;        // power offset of this slot: c / CP, the same for the whole wave
  const int64_t cc = cq < a.C ? cq : 0;  // columns >= C re-read column 0: their sums are never flushed
  const uint32_t poff = (uint32_t)(c * 64 + ((g2 + 4 * ((c >> 1) & 3)) & 15) * 4);
  const uint32_t foff = (uint32_t)(FRAG0 + c * 32 + g2 * 2);
  // u-row role (lanes of column slot 0): wave jj < 4 slices the power J0 + jj; a fifth power goes to
  // wave 7 (fewest MFMAs), not to wave 4, which shares its SIMD with wave 0
  // (PK > 1: the first wave of every power-offset group slices the u-row monomial jq + PK * jj in row set jj)
  const int jr = wave < 4 ? wave : 4;
  const bool urow = PK == 1 ? (jr < JN && (wave < 4 || wave == I8_WAVES - 1)) : (wave * 4 % CP == 0);  // wave-uniform
  // column slot k of the wave stores the digits 2k, 2k+1 of the u-row monomial (columns 8 jr + 2k, + 1)
  const uint32_t uoff = (uint32_t)(FRAG0 + (JN + (jr >> 2)) * I8_FRAG + ((jr & 3) * 8 + 2 * cslot) * 32 + g2 * 2);
  const uint32_t usel = (cslot & 1) ? 0x07030602u : 0x05010400u;   // which byte pair of the two source dwords
  const uint32_t uxor = cslot == 3 ? 0u : 0x80808080u;             // digit 6 keeps its bias (removed at flush time)
  // MFMA role, the same shape for every wave (no per-wave code paths): the pair rows 2 wave and
  // 2 wave + 1 (4 tiles each: two digits x two replicate halves) and the plain fragment `wave`
  // (2 tiles).  Indices past the end are clamped: those tiles compute a duplicate that is never flushed.
  // With one or two powers there are fewer rows than waves: they are dealt out one per wave (fragments from the
  // last wave down) and a wave without a row in a slot skips that slot's reads and MFMAs -- otherwise wave 0 would
  // carry 10 MFMAs per k-step and six waves would compute duplicates (orders 0 and 1 took as long as order 4).
  // (measured, N = 1e8, nrep = 1000: the dealt-out map also wins with four powers -- 203.0 -> 191.5 ms, no duplicates on
  // waves 6 and 7 -- is neutral with three and loses with five, 216 vs 203 ms, where every wave already owns two rows)
  constexpr bool SPREAD = JN <= 2 || JN == 4;
  const int p0i = SPREAD ? wave : 2 * wave, p1i = SPREAD ? wave + I8_WAVES : 2 * wave + 1,
            f2i = SPREAD ? I8_WAVES - 1 - wave : wave;
  // (from three powers on, the one or two waves without a first row compute a clamped duplicate as before: a branch
  // around the slot costs the other waves more than the duplicate does -- order 3: 209 vs 192 ms)
  const bool s0ok = !SPREAD || NPAIR >= I8_WAVES || p0i < NPAIR, p1ok = p1i < NPAIR, f2ok = f2i < NFRG;  // wave-uniform
  const int p0 = p0i < NPAIR ? p0i : NPAIR - 1, p1 = p1i < NPAIR ? p1i : NPAIR - 1, f2 = f2i < NFRG ? f2i : NFRG - 1;
  static_assert(2 * I8_WAVES >= NPAIR && I8_WAVES >= NFRG, "every row and fragment has an owner");
  const uint32_t prot = (uint32_t)((n32 >> 1) & 3);
  const uint32_t pr0 = (uint32_t)(n32 * 64) + ((2u * half + prot) & 3u) * 16u;       // pairs 8 half .. 8 half + 3
  const uint32_t pr1 = (uint32_t)(n32 * 64) + ((2u * half + 1u + prot) & 3u) * 16u;  // pairs 8 half + 4 .. + 7
  const uint32_t fr = (uint32_t)(FRAG0 + n32 * 32 + half * 16);
  const uint32_t aoff = (uint32_t)(n32 * I8_CNT_ROW + half * 4);

  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int chunk = (q / a.n_rbg) * 8 + xcd;
  const int rbg = q % a.n_rbg;
  const int64_t rep0 = (int64_t)rbg * I8_REPS;
  const int64_t t_begin = (int64_t)chunk * a.tiles_per_chunk;
  int64_t t_end = t_begin + a.tiles_per_chunk;
  if (t_end > a.ntiles) t_end = a.ntiles;

  const double pu = a.pivot[0];
  const double px = a.pivot[1 + a.col0 + cc];

  v16i acc[10];  // slot 0: 0..3 = {digit a, half 0}, {a, 1}, {b, 0}, {b, 1}; slot 1: 4..7; slot 2: 8, 9
#pragma unroll
  for (int e = 0; e < 10; ++e) acc[e] = (v16i)(0);

  // rows of the B buffers that are never written (columns >= C, unused u-row columns) stay zero
  for (int e = threadIdx.x; e < 2 * NB * BUF / 16; e += I8_BLOCK)
    reinterpret_cast<uint4 *>(bb0)[e] = make_uint4(0, 0, 0, 0);

  double inv_du = 0.0, inv_w = 1.0, sc = 0.0;
#ifdef TXM_I8_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t0 = clock64();
#define TXM_TICK(k) do { const long long t1_ = clock64(); tm[k] += t1_ - t0; t0 = t1_; } while (0)
#else
#define TXM_TICK(k) do {} while (0)
#endif

  // ---- flush the int32 accumulators of one window into the per-digit FP64 partial sums ----
  // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
  uint32_t fdraws = 0;  // lane rr < 8: draws of replicate rep0 + 8 wave + rr in the current window
  const double *wt = a.wtab;
  // `opq` is an opaque zero that `flush` re-creates for every window (asm volatile).  The flush addresses are built on
  // it so that they are computed where they are used: left to itself the compiler hoists all 160 of them out of the
  // window loop and spills them (227 VGPRs in round 1), and every scratch RELOAD in the per-tile code is a vector-memory
  // wait that also waits for whatever global loads are in flight (s_waitcnt vmcnt is in order) -- the prefetches.
  int64_t opq = 0;
  int64_t fwin = 0;  // the window being flushed
  // one tile: digit i of power J0 + jj; observable columns (urow_f < 0) or the packed u-row fragment urow_f
  auto flush_tile = [&](v16i &T, int h, int jj, int i, int urow_f) {
    bool valid;
    if (urow_f >= 0) {
      const int n = urow_f * 32 + n32;
      valid = (n & 7) < I8_NSL && (n >> 3) < KL;
      jj = valid ? n >> 3 : 0;
      i = valid ? n & 7 : 0;
    } else {
      valid = n32 < a.C;
    }
    int j = J0 + jj, col = n32;
    if constexpr (PK > 1) {
      if (urow_f < 0) {  // column slot n32 = (power offset, observable)
        col = n32 % CP;
        j = jj * PK + n32 / CP;
        valid = col < a.C && j < K;
        if (!valid) j = 0;
      }
    }
    double dsc = wt[I8_WT_DSP + j] * (urow_f >= 0 ? 0x1p-50 : wt[I8_WT_DSC + col]);
    dsc *= (double)((int64_t)1 << (8 * i));
    // one slot per scaling window: [window][replicate][power][digit slot][column] (u-row: [window][replicate][power]
    // [digit slot]; the 32 lanes of a tile row write 256 contiguous bytes), stored once and added up by the finalize kernel in window order -- a replicate's sums do not depend
    // on how the launch was cut into chunks (txm_resample_i8t.hip has the same layout)
    double *base;
    int64_t stride;
    if (urow_f < 0) {
      base = a.part_x + ((((size_t)fwin * a.nrep_pad + rep0 + 32 * h + 4 * half) * K + j) * 8 + i) * I8_CPAD + col + opq;
      stride = (int64_t)K * I8_CPAD * 8;
    } else {
      base = a.part_u + (((size_t)fwin * a.nrep_pad + rep0 + 32 * h + 4 * half) * K + j) * 8 + i + opq;
      stride = (int64_t)K * 8;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r >> 2) * 8 + (r & 3);
      int v = T[r];
      if (i == I8_NSL - 1) v -= I8_D6_BIAS * (int)fsum[32 * h + m + 4 * half];
      if (valid) base[(int64_t)m * stride] = (double)v * dsc;
    }
    T = (v16i)(0);
  };
  auto flush = [&](int64_t win) {
    {
      uint32_t z = 0;
      asm volatile("" : "+v"(z));
      opq = (int64_t)z;
    }
    wt = a.wtab + win * I8_WT_STRIDE;
    fwin = win;
    if (lane < I8_REPS_WAVE) fsum[wave * I8_REPS_WAVE + lane] = fdraws;
    fdraws = 0;
    __syncthreads();
    if (p0i < NPAIR) {
      const int jj = p0i / 3, w = p0i % 3;
      flush_tile(acc[0], 0, jj, 2 * w, -1);
      flush_tile(acc[1], 1, jj, 2 * w, -1);
      flush_tile(acc[2], 0, jj, 2 * w + 1, -1);
      flush_tile(acc[3], 1, jj, 2 * w + 1, -1);
    }
    if (p1i < NPAIR) {
      const int jj = p1i / 3, w = p1i % 3;
      flush_tile(acc[4], 0, jj, 2 * w, -1);
      flush_tile(acc[5], 1, jj, 2 * w, -1);
      flush_tile(acc[6], 0, jj, 2 * w + 1, -1);
      flush_tile(acc[7], 1, jj, 2 * w + 1, -1);
    }
    if (f2i < NFRG) {
      if (f2i < JN) {
        flush_tile(acc[8], 0, f2i, I8_NSL - 1, -1);
        flush_tile(acc[9], 1, f2i, I8_NSL - 1, -1);
      } else {
        flush_tile(acc[8], 0, 0, 0, f2i - JN);
        flush_tile(acc[9], 1, 0, 0, f2i - JN);
      }
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) acc[e] = (v16i)(0);  // duplicates of clamped slots included
  };

  auto load_chunk = [&](int64_t wbase, int s, I8Chunk &r) {
    const int64_t i = wbase + s * 32 + g2 * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#ifdef TXM_I8_NO_XLOAD
      r.x[e] = (double)(i + e) * 1e-9 + px;  // ablation: no memory access
#else
      r.x[e] = a.x[(i + e) * a.ldx_s + a.col0 + cc];
#endif
    }
  };

  // the two MFMA operands of a pair row: D[0..7] = the dwords of pairs 8 half .. 8 half + 7 of this
  // lane's column; the low halves are digit 2w, the high halves digit 2w + 1
  auto deinterleave = [](const v4i &lo, const v4i &hi, v4i &Xa, v4i &Xb) {
    Xa[0] = (int)__builtin_amdgcn_perm((uint32_t)lo[1], (uint32_t)lo[0], 0x05040100u);
    Xa[1] = (int)__builtin_amdgcn_perm((uint32_t)lo[3], (uint32_t)lo[2], 0x05040100u);
    Xa[2] = (int)__builtin_amdgcn_perm((uint32_t)hi[1], (uint32_t)hi[0], 0x05040100u);
    Xa[3] = (int)__builtin_amdgcn_perm((uint32_t)hi[3], (uint32_t)hi[2], 0x05040100u);
    Xb[0] = (int)__builtin_amdgcn_perm((uint32_t)lo[1], (uint32_t)lo[0], 0x07060302u);
    Xb[1] = (int)__builtin_amdgcn_perm((uint32_t)lo[3], (uint32_t)lo[2], 0x07060302u);
    Xb[2] = (int)__builtin_amdgcn_perm((uint32_t)hi[1], (uint32_t)hi[0], 0x07060302u);
    Xb[3] = (int)__builtin_amdgcn_perm((uint32_t)hi[3], (uint32_t)hi[2], 0x07060302u);
  };
#define TXM_I8_MFMA2(T0, T1, B_) \
  do { \
    (T0) = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, (B_), (T0), 0, 0, 0); \
    (T1) = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, (B_), (T1), 0, 0, 0); \
  } while (0)

  // ---- fused k-step: every LDS operand read of chunk s is issued first (the LDS queue is in order:
  // a read issued behind the slicing stores would wait for all of them), then the MFMAs go out in
  // pairs between the slicing of the powers ----
  // (sl = the chunk held in r, sliced now; snext = the chunk loaded into r for the next call)
  auto step = [&](const unsigned char *bcur, int s, unsigned char *bnxt, I8Chunk &r, int64_t wbase, int sl,
                  int snext, bool slice, bool mf = true) {
    const uint32_t *cw = cnt + s * 8 + aoff;
    const v4i A0 = *reinterpret_cast<const v4i *>(cw);
    const v4i A1 = *reinterpret_cast<const v4i *>(cw + 32 * I8_CNT_ROW);
    // the operands of the first pair row are read before anything is stored (the LDS queue is in
    // order); the MFMAs of a slot go out between the powers, the next slot's reads one power ahead
    v4i Ra = (v4i)(0), Rb = (v4i)(0), Rc = (v4i)(0), Rd = (v4i)(0);
#ifndef TXM_I8_NO_MFMA
    if (mf && s0ok) {
      Ra = *reinterpret_cast<const v4i *>(bcur + p0 * PAIR_B + pr0);
      Rb = *reinterpret_cast<const v4i *>(bcur + p0 * PAIR_B + pr1);
    }
#endif
    uint32_t wf = (uint32_t)(bnxt - lds) + foff;  // (pair rows go through M0 + immediates: no per-lane base)
    (void)poff;
    // opaque per-lane base + immediate offsets: the B buffers sit above 64 KiB, so constant-folded
    // absolute LDS addresses would take a register each
    asm volatile("" : "+v"(wf));
    double du[2], dx[2], p[2];
    {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        du[e] = utile[sl * 32 + g2 * 2 + e];
        dx[e] = (r.x[e] - px) * sc;
        if constexpr (WEIGHTED) p[e] = wtile[sl * 32 + g2 * 2 + e];
        else p[e] = 1.0;
      }
    }
    if (slice) load_chunk(wbase, snext, r);
    TXM_TICK(3);
#pragma unroll
    for (int q = 0; q < J0; ++q) {
      p[0] *= du[0];
      p[1] *= du[1];
    }
    double dstep[2] = {du[0], du[1]};  // what a row set multiplies by: du, or du^PK with PK powers per observable
    if constexpr (PK > 1) {
      // this wave's slots start at the power jq (wave-uniform): p = w * du^jq
      const double d2[2] = {du[0] * du[0], du[1] * du[1]};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (jq & 1) p[e] *= du[e];
        if (PK == 4 && (jq & 2)) p[e] *= d2[e];
        dstep[e] = PK == 2 ? d2[e] : d2[e] * d2[e];
      }
    }
    constexpr int NIT = JN > 3 ? JN : 3;
    // M0 = LDS byte address of this wave's 256-byte run in pair row 0 of the buffer being filled (ds_write_addtid_b32:
    // address = M0 + offset + 4 * lane, no address register, half the LDS-path cycles of ds_write_b32)
    const uint32_t m0base = __builtin_amdgcn_readfirstlane((uint32_t)(bnxt - lds) + (uint32_t)(wave * 256));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#ifdef TXM_I8_NO_PRODUCE
      const bool do_slice = false;
#else
      const bool do_slice = slice && it < JN;
#endif
      const int jj = it;
      uint32_t T[4] = {0, 0, 0, 0};
#ifndef TXM_I8_NO_MFMA
      if (mf) {
        if (it == 0) {
          if (SPREAD ? p1ok : true) {
            Rc = *reinterpret_cast<const v4i *>(bcur + p1 * PAIR_B + pr0);
            Rd = *reinterpret_cast<const v4i *>(bcur + p1 * PAIR_B + pr1);
          }
          if (s0ok) {
            v4i Xa, Xb;
            deinterleave(Ra, Rb, Xa, Xb);
            TXM_I8_MFMA2(acc[0], acc[1], Xa);
            TXM_I8_MFMA2(acc[2], acc[3], Xb);
          }
        }
        if (it == 1) {
          if (SPREAD ? f2ok : true) Ra = *reinterpret_cast<const v4i *>(bcur + f2 * I8_FRAG + fr);
          if (p1ok) {  // wave-uniform: a wave without a second row skips the slot
            v4i Xa, Xb;
            deinterleave(Rc, Rd, Xa, Xb);
            TXM_I8_MFMA2(acc[4], acc[5], Xa);
            TXM_I8_MFMA2(acc[6], acc[7], Xb);
          }
        }
        if (it == 2 && f2ok) TXM_I8_MFMA2(acc[8], acc[9], Ra);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (do_slice) {
        if (jj > 0) {
          p[0] *= dstep[0];
          p[1] *= dstep[1];
        }
        i8_slice2(fma(p[0], dx[0], I8_MAGIC), fma(p[1], dx[1], I8_MAGIC), T);
        // (s_nop: one wait state between an SALU write of M0 and an add-TID LDS instruction; the assembler pads
        // nothing inside an asm statement)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:%4\n\t"
                     "ds_write_addtid_b32 %1 offset:%5\n\t"
                     "ds_write_addtid_b32 %2 offset:%6"
                     :
                     : "v"(T[0]), "v"(T[1]), "v"(T[2]), "s"(m0base), "n"((jj * 3 + 0) * PAIR_B),
                       "n"((jj * 3 + 1) * PAIR_B), "n"((jj * 3 + 2) * PAIR_B)
                     : "memory");
        *reinterpret_cast<uint16_t *>(lds + wf + jj * I8_FRAG) = (uint16_t)T[3];
        // u-row: p is w * du^(J0 + jj) right now; the wave that owns this power slices it (dx = 1)
        // into the columns 8 jj + i of the packed u-row fragments
        bool ur = jj == jr && urow;
        uint32_t uo = uoff;
        if constexpr (PK > 1) {  // row set jj holds the monomial jq + PK * jj of this wave's power offset
          const int qm = jj * PK + jq;
          ur = urow && qm < KL;
          uo = (uint32_t)(FRAG0 + (JN + (qm >> 2)) * I8_FRAG + ((qm & 3) * 8 + 2 * cslot) * 32 + g2 * 2);
        }
        if (ur) {  // wave-uniform
          // all four column slots of the wave hold this sample pair's p (it does not depend on the column), so
          // slot k cuts and stores the digits 2k and 2k+1 only: 2 stores per wave instead of 7 by a quarter of it.
          // (slot 3's second halfword is the exponent byte pair: it lands in the unused column 8 jj + 7.)
          const uint64_t b0 = (uint64_t)__double_as_longlong(fma(p[0], 0x1p50, I8_MAGIC));
          const uint64_t b1 = (uint64_t)__double_as_longlong(fma(p[1], 0x1p50, I8_MAGIC));
          const uint32_t s0 = cslot < 2 ? (uint32_t)b0 : (uint32_t)(b0 >> 32);
          const uint32_t s1 = cslot < 2 ? (uint32_t)b1 : (uint32_t)(b1 >> 32);
          const uint32_t t = __builtin_amdgcn_perm(s1, s0, usel) ^ uxor;
          *reinterpret_cast<uint16_t *>(bnxt + uo) = (uint16_t)t;
          *reinterpret_cast<uint16_t *>(bnxt + uo + 32) = (uint16_t)(t >> 16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  uint32_t *pg = a.progress != nullptr ? a.progress + (size_t)chunk * 64 : nullptr;
  uint32_t tiles_done = 1;  // published value = tiles finished + 1 (0 means "not started")
  // chunks are made of whole windows (tiles_per_chunk is a multiple of win_tiles)
  const int64_t WT = a.win_tiles;
  for (int64_t win = t_begin / WT; win * WT < t_end; ++win) {
    if (a.wflag[win] != 0u) continue;  // precision guard: this window goes to the FP64 kernel (wave-uniform)
    {
      const double *wt = a.wtab + win * I8_WT_STRIDE;
      // (kept in vector registers on purpose: moving these two uniform factors to scalar registers changed the
      // register allocation so that the kernel fetched 37 instead of 15 GiB per launch -- measured, tools/ab_fetch.sh)
      inv_du = wt[I8_WT_INVDU];
      if constexpr (WEIGHTED) inv_w = wt[I8_WT_INVW];
      sc = wt[I8_WT_SC + cc];
    }
    int64_t tt_end = (win + 1) * WT;
    if (tt_end > t_end) tt_end = t_end;
    // What a tile needs that does not depend on the previous tile is fetched ONE TILE AHEAD: in-order waves park on a
    // load until it is back, and at the top of a tile every wave of the workgroup would (~7k cycles of HBM latency per
    // 89k-cycle tile in round 1).  The next tile's u samples and the wave's 8 tile counts are requested at the top of
    // this tile, are back by the end of the stage-3 fill and are parked in LDS there (second u tile, cntlds) -- not
    // kept in registers across the k-steps, which have none to spare; the next tile's first x chunk rides in the
    // load slot of this tile's last slicing call.  (Weighted launches have no LDS left for second u and w tiles:
    // they prefetch the counts and the x chunk only.)  Nothing is requested across windows.
    bool have_pref = false;  // uniform
    // x chunks in flight.  With one or two powers per launch a k-step is shorter than the latency of a global load
    // (the workgroup that leads its chunk group misses L2), so the chunk is requested XD = 8 k-steps ahead (order 0:
    // 150.8 / 146.6 / 138.1 / 176.6 ms for 4 / 4 with paired chunks / 8 / 16; the slicing temporaries of so few powers
    // leave the registers); from three powers on one step ahead is enough and anything deeper spills (depth 2 at
    // JN = 5: +3 %).  Ablation at order 0: without the x loads 111 ms -- what is left of their cost is the 16 cache
    // lines a wave's load touches (4 columns x 16 sample pairs), a property of the lane map.
    I8Chunk rq[XD];
    I8Chunk &r0 = rq[0];
    const int64_t rep0w = rep0 + wave * I8_REPS_WAVE;
    const uint32_t rl0 = (uint32_t)(wave * I8_REPS_WAVE);
    auto tile_base = [&](int64_t tt) {
      const int64_t b0 = tt * SM_T;
      return b0 > a.N - SM_T ? a.N - SM_T : b0;  // the last tile slides its window back
    };
#pragma unroll 1
    for (int64_t t = win * WT; t < tt_end; ++t) {
      const int64_t i_tile = t * SM_T;
      const uint32_t tsize = (t == a.ntiles - 1) ? a.last_tile_size : (uint32_t)SM_T;
      const int64_t wbase = tile_base(t);
      const uint32_t shift = (uint32_t)(i_tile - wbase);
      const bool has_next = t + 1 < tt_end;
      const int64_t wnext = has_next ? tile_base(t + 1) : wbase;

      if (pg != nullptr && wave == 0) {  // wave-uniform; the other waves are held by the next barrier
        if (lane == 0) __hip_atomic_store(&pg[rbg & 63], tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
        for (int spin = 0; spin < I8_THROTTLE_SPINS; ++spin) {
          uint32_t v = __hip_atomic_load(&pg[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v == 0u) v = 0xffffffffu;  // unused slot or a group that has not started
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const uint32_t w2 = (uint32_t)__shfl_xor((int)v, o);
            v = w2 < v ? w2 : v;
          }
          if (tiles_done <= v + I8_LEAD) break;
          __builtin_amdgcn_s_sleep(32);
        }
      }
      ++tiles_done;
      TXM_TICK(7);
      // The lane id is re-read from the hardware here (opaque to the compiler) and every per-thread address of this
      // per-tile code is built on it.  Otherwise those addresses are loop invariants that the register allocator keeps
      // in scratch, and each reload is a vector-memory wait that also waits for the global loads issued just before it
      // (vmcnt is in order): the requests below would be waited for one after the other, right here.
      uint32_t lane_f;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_f));
      const int i2 = 2 * (int)((uint32_t)wave * 64u + lane_f);  // 512 threads x 2 samples
      const bool cnt_lane = lane_f < (uint32_t)I8_REPS_WAVE && rep0w + (int64_t)lane_f < a.nrep;
      uint32_t wcnt;  // lane i < 8: draws of replicate rep0w + i in this tile
      if (!have_pref) {  // first tile of a window: nothing was requested ahead
#pragma unroll
        for (int q = 0; q < XD; ++q) load_chunk(wbase, q, rq[q]);
        wcnt = cnt_lane ? a.counts[(size_t)(rep0w + lane_f) * a.ntiles + t] : 0u;
      } else {
        wcnt = lane_f < (uint32_t)I8_REPS_WAVE ? cntlds[rl0 + lane_f] : 0u;
      }
      if (WEIGHTED || !have_pref) {
        const double u0 = a.u[wbase + i2], u1 = a.u[wbase + i2 + 1];
        utile[i2] = (u0 - pu) * inv_du;
        utile[i2 + 1] = (u1 - pu) * inv_du;
        if constexpr (WEIGHTED) {
          wtile[i2] = a.w[wbase + i2] * inv_w;
          wtile[i2 + 1] = a.w[wbase + i2 + 1] * inv_w;
        }
      }
      // requests for the next tile: in flight during the zeroing and the fill below
      double nu0 = 0.0, nu1 = 0.0;
      uint32_t ncnt = 0;
      if (has_next) {
        if constexpr (!WEIGHTED) {
          nu0 = a.u[wnext + i2];
          nu1 = a.u[wnext + i2 + 1];
        }
        if (cnt_lane) ncnt = a.counts[(size_t)(rep0w + lane_f) * a.ntiles + t + 1];
      }

      // ---- stage 3 of the sampler: the workgroup's 64 x 1024 count tile -------
      for (int e = i2 / 2; e < I8_CNT_BYTES / 16; e += I8_BLOCK)
        reinterpret_cast<uint4 *>(cnt)[e] = make_uint4(0, 0, 0, 0);
      TXM_TICK(0);
      __syncthreads();
      TXM_TICK(1);
      {
#ifdef TXM_I8_NO_FILL
        if (rep0w < 0) {
#else
        if (tsize == (uint32_t)SM_T) {
#endif
          i8_fill_full(a, cnt, rep0w, rl0, t, lane, wcnt);
        } else if (tsize != (uint32_t)SM_T) {
          for (int rr = 0; rr < I8_REPS_WAVE; ++rr) {
            const int64_t r = rep0w + rr;
            if (r >= a.nrep) break;  // wave-uniform
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)wcnt, rr);
            sampler_fine_tile(a.k0, a.k1, a.rep_base + (uint32_t)r, (uint32_t)t, n, tsize, lane, [&](uint32_t off0) {
              const uint32_t off = off0 + shift;
              atomicAdd(&cnt[(rl0 + (uint32_t)rr) * I8_CNT_ROW + (off >> 2)], 1u << ((off & 3u) << 3));
            });
          }
        }
      }
      if (cnt_lane) fdraws += wcnt;
      if (has_next) {  // park what was requested above (same window: same scale factors)
        if constexpr (!WEIGHTED) {
          utile_nxt[i2] = (nu0 - pu) * inv_du;
          utile_nxt[i2 + 1] = (nu1 - pu) * inv_du;
        }
        if (lane_f < (uint32_t)I8_REPS_WAVE) cntlds[rl0 + lane_f] = ncnt;
      }
      TXM_TICK(2);
      __syncthreads();
      TXM_TICK(1);

      // ---- contraction: chunk s on the matrix pipe, chunk s+1 through the slicer ----
      if constexpr (XD > 1) {
        // chunk c lives in rq[c % XD]; the step that slices it requests chunk c + XD into the same registers
        // (past the end of the tile: the next tile's chunk c + XD - 32, or a harmless re-load on the last tile)
        auto target = [&](int cl, int64_t &wb, int &sl) {
          if (cl < I8_STEPS) { wb = wbase; sl = cl; }
          else if (has_next) { wb = wnext; sl = cl - I8_STEPS; }
          else { wb = wbase; sl = I8_STEPS - 1; }
        };
        static_assert(NB == 2 && XD % 2 == 0, "paired chunks");
        auto bufof = [&](int c) { return (((c >> 1) & 1) ? bb1 : bb0) + (c & 1) * BUF; };
#pragma unroll
        for (int c = 0; c < 2; ++c) {  // prologue: slice chunks 0 and 1, no MFMAs
          int64_t wb; int sl;
          target(c + XD, wb, sl);
          step(bb1, 0, bufof(c), rq[c % XD], wb, c, sl, true, false);
        }
        __syncthreads();
#pragma unroll 1
        for (int s0 = 0; s0 < I8_STEPS; s0 += XD) {
#pragma unroll
          for (int e = 0; e < XD; ++e) {
            const int sq = s0 + e, cs = sq + 2;  // contract chunk sq, slice chunk sq + 2
            int64_t wb; int sl;
            target(cs + XD, wb, sl);
            step(bufof(e), sq, bufof(e + 2), rq[(e + 2) % XD], wb, cs < I8_STEPS ? cs : I8_STEPS - 1, sl, cs < I8_STEPS);
            TXM_TICK(4);
            if (e & 1) __syncthreads();
            TXM_TICK(5);
          }
        }
      } else {
      step(bb1, 0, bb0, r0, wbase, 0, 1, true, false);  // prologue: slice chunk 0, no MFMAs
      TXM_TICK(4);
      __syncthreads();
      TXM_TICK(5);
#pragma unroll 1
      for (int s = 0; s < I8_STEPS - 2; s += 2) {
        step(bb0, s, bb1, r0, wbase, s + 1, s + 2, true);
        TXM_TICK(4);
        __syncthreads();
        TXM_TICK(5);
        step(bb1, s + 1, bb0, r0, wbase, s + 2, s + 3 < I8_STEPS ? s + 3 : I8_STEPS - 1, true);
        TXM_TICK(4);
        __syncthreads();
        TXM_TICK(5);
      }
      // the last slicing call has no chunk of this tile left to load: it fetches the next tile's first chunk instead
      step(bb0, I8_STEPS - 2, bb1, r0, wnext, I8_STEPS - 1, has_next ? 0 : I8_STEPS - 1, true);
      __syncthreads();
      step(bb1, I8_STEPS - 1, bb0, r0, wbase, I8_STEPS - 1, I8_STEPS - 1, false);
      __syncthreads();
      }
      have_pref = has_next;
      if constexpr (!WEIGHTED) {
        if (has_next) {  // the parked u tile becomes the current one
          double *tmp = utile;
          utile = utile_nxt;
          utile_nxt = tmp;
        }
      }
    }
    flush(win);
    TXM_TICK(6);
  }
  if (pg != nullptr && threadIdx.x == 0)  // finished: never hold the others back
    __hip_atomic_store(&pg[rbg & 63], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TXM_I8_TIMING
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
    for (int k = 0; k < 8; ++k) a.wtab[a.nwin * I8_WT_STRIDE + ((blockIdx.x ? 1 : 0) * 8 + wave) * 8 + k] = (double)tm[k];
#endif
}

// ---------------------------------------------------------------------------
bool i8_supported(int64_t N, int64_t C, int64_t nrep, int K) {
  (void)nrep;
  return N >= SM_T && C >= 1 && C <= 64 * I8_CPAD && K >= 1 && K <= 8;  // C > 32: one launch per 32 columns
}

template <int K, int J0, int JN, int PK = 1>
static int launch_pass(const I8Args &a, bool weighted, size_t prog_bytes, hipStream_t st) {
  if (a.progress != nullptr) TXM_HIP(hipMemsetAsync(a.progress, 0, prog_bytes, st));  // every pass starts from "not started"
  const dim3 grid((unsigned)(a.n_chunks * a.n_rbg)), block(I8_BLOCK);
  constexpr int kl = PK == 1 ? JN : (K < JN * PK ? K : JN * PK);             // u-row monomials of the launch
  constexpr int buf = 3 * JN * 2048 + (JN + (8 * kl + 31) / 32) * I8_FRAG;  // pair rows + plain fragments
  constexpr int nb = JN <= 2 ? 2 : 1;  // chunks per B buffer (two k-steps per barrier with one or two powers)
  const size_t lds = (size_t)I8_CNT_BYTES + 2u * nb * (size_t)buf + 2u * I8_REPS * sizeof(uint32_t) +
                     2u * SM_T * sizeof(double);  // + window draws, parked counts; two u tiles, or one u + one w
  // the dynamic-LDS limit is a property of (function, device)
  if (weighted) TXM_SET_MAX_LDS((&resample_i8_kernel<K, J0, JN, true, PK>), lds);
  else TXM_SET_MAX_LDS((&resample_i8_kernel<K, J0, JN, false, PK>), lds);
  if (weighted) hipLaunchKernelGGL((resample_i8_kernel<K, J0, JN, true, PK>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((resample_i8_kernel<K, J0, JN, false, PK>), grid, block, lds, st, a);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// TXM_I8_PACK=0 keeps one power per column for narrow states too (A/B measurements)
static bool pack_i8_on() {
  static const bool on = [] {
    const char *e = getenv("TXM_I8_PACK");
    return !(e && e[0] == '0');
  }();
  return on;
}

// the pre-pass: per-window scale table, guard flags and the FP64 fallback list.  Depends on (x, u, w, pivot, shape)
// only -- a caller that bootstraps the same data again passes the tables back in (txm_resample_opts.prep)
__global__ void i8_or_flags_kernel(uint32_t *__restrict__ wflag, const uint32_t *__restrict__ yflag, int64_t nwin) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nwin) wflag[w] |= yflag[w];
}

int launch_i8_prepass(const I8Args &a, int K, hipStream_t st) {
  // sub-blocks of <= 16 tiles; a window is 1 sub-block (4- and 16-tile windows) or win_tiles / 16 of them
  const int64_t sub_tiles = a.win_tiles < I8_STAT_TILES ? a.win_tiles : I8_STAT_TILES;
  const int nsub = (int)(a.win_tiles / sub_tiles);
  const int64_t nsub_total = cdiv(a.ntiles, sub_tiles);
  const bool vec2 = ((reinterpret_cast<uintptr_t>(a.x + a.col0) & 15) == 0) && (a.ldx_s % 2 == 0);
  if (vec2)
    hipLaunchKernelGGL(i8_stats_kernel<true>, dim3((unsigned)nsub_total), dim3(256), 0, st, a.x, a.ldx_s, a.u, a.w,
                       a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.pivot, K - 1, a.stats);
  else
    hipLaunchKernelGGL(i8_stats_kernel<false>, dim3((unsigned)nsub_total), dim3(256), 0, st, a.x, a.ldx_s, a.u, a.w,
                       a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.pivot, K - 1, a.stats);
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL(i8_table_kernel, dim3((unsigned)a.nwin), dim3(64), 0, st, a.stats, nsub, nsub_total, a.N, a.C,
                     a.win_tiles * SM_T, a.w != nullptr, K - 1, a.wtab, a.wflag);
  TXM_LAUNCH_CHECK();
  if (a.y != nullptr) {
    // the second matrix: its own column scales (order-0 monomial w dy), its guard flags OR-ed into the call's -- a
    // window either runs on the int8 kernel for both matrices or on the FP64 kernel for both
    const bool vy = ((reinterpret_cast<uintptr_t>(a.y + a.col0) & 15) == 0) && (a.ldy_s % 2 == 0);
    if (vy)
      hipLaunchKernelGGL(i8_stats_kernel<true>, dim3((unsigned)nsub_total), dim3(256), 0, st, a.y, a.ldy_s, a.u, a.w,
                         a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.ypivot, 0, a.stats);
    else
      hipLaunchKernelGGL(i8_stats_kernel<false>, dim3((unsigned)nsub_total), dim3(256), 0, st, a.y, a.ldy_s, a.u, a.w,
                         a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.ypivot, 0, a.stats);
    TXM_LAUNCH_CHECK();
    hipLaunchKernelGGL(i8_table_kernel, dim3((unsigned)a.nwin), dim3(64), 0, st, a.stats, nsub, nsub_total, a.N, a.C,
                       a.win_tiles * SM_T, a.w != nullptr, 0, a.ywtab, a.yflag);
    TXM_LAUNCH_CHECK();
    hipLaunchKernelGGL(i8_or_flags_kernel, dim3((unsigned)cdiv(a.nwin, 256)), dim3(256), 0, st, a.wflag, a.yflag, a.nwin);
    TXM_LAUNCH_CHECK();
  }
  TXM_HIP(hipMemsetAsync(a.n_list, 0, 256, st));
  hipLaunchKernelGGL(i8_list_kernel, dim3(1), dim3(256), 0, st, a.wflag, a.nwin, a.win_tiles, a.sub_tiles, a.list,
                     a.n_list);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

static bool i8t_on() {
  static const bool on = [] {
    const char *e = getenv("TXM_I8T");
    return !(e && e[0] == '0');
  }();
  return on;
}

int launch_resample_i8(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st) {
  int rc = TXM_OK;
  // narrow state on the transposing-read kernel (the waves of a column quad split the powers)
  if (i8t_on() && i8t_narrow_nq(a.C_call, K) != 0) return launch_resample_i8t(a, K, weighted, prog_bytes, st);
  // narrow state: four powers per observable column (two row sets at most).  Not for the narrow tail group of a wide
  // state: its u-row sums would round differently from the other groups' (the monomials are formed in another order)
  if (a.C <= 8 && a.col0 == 0 && K >= 2 && pack_i8_on()) {
    switch (K) {
      case 2: return launch_pass<2, 0, 1, 4>(a, weighted, prog_bytes, st);
      case 3: return launch_pass<3, 0, 1, 4>(a, weighted, prog_bytes, st);
      case 4: return launch_pass<4, 0, 1, 4>(a, weighted, prog_bytes, st);
      case 5: return launch_pass<5, 0, 2, 4>(a, weighted, prog_bytes, st);
      case 6: return launch_pass<6, 0, 2, 4>(a, weighted, prog_bytes, st);
      case 7: return launch_pass<7, 0, 2, 4>(a, weighted, prog_bytes, st);
      case 8: return launch_pass<8, 0, 2, 4>(a, weighted, prog_bytes, st);
      default: break;
    }
  }
  if (a.C > 8 && a.C <= 16 && a.col0 == 0 && K >= 2 && pack_i8_on()) {  // two powers per observable column
    switch (K) {
      case 2: return launch_pass<2, 0, 1, 2>(a, weighted, prog_bytes, st);
      case 3: return launch_pass<3, 0, 2, 2>(a, weighted, prog_bytes, st);
      case 4: return launch_pass<4, 0, 2, 2>(a, weighted, prog_bytes, st);
      case 5: return launch_pass<5, 0, 3, 2>(a, weighted, prog_bytes, st);
      case 6: return launch_pass<6, 0, 3, 2>(a, weighted, prog_bytes, st);
      case 7: return launch_pass<7, 0, 4, 2>(a, weighted, prog_bytes, st);
      case 8: return launch_pass<8, 0, 4, 2>(a, weighted, prog_bytes, st);
      default: break;
    }
  }
  // one power per observable column: the transposing-read kernel (txm_resample_i8t.hip); TXM_I8T=0 keeps this
  // file's kernel for A/B timing
  if (i8t_on() && i8t_applicable(a.x, a.ldx_s, a.C_call)) return launch_resample_i8t(a, K, weighted, prog_bytes, st);
  switch (K) {
    case 1: rc = launch_pass<1, 0, 1>(a, weighted, prog_bytes, st); break;
    case 2: rc = launch_pass<2, 0, 2>(a, weighted, prog_bytes, st); break;
    case 3: rc = launch_pass<3, 0, 3>(a, weighted, prog_bytes, st); break;
    case 4: rc = launch_pass<4, 0, 4>(a, weighted, prog_bytes, st); break;
    case 5: rc = launch_pass<5, 0, 5>(a, weighted, prog_bytes, st); break;
    // orders 5..7: two passes over the sampler stream, each with its own powers
    // two passes: a full five-power pass, then the rest -- a pass costs ~150 / 157 / 181 / 205 / 202 ms for 1..5 powers
    // (N = 1e8, nrep = 1000), so 5 + 2 (359 ms) beats 4 + 3 (386 ms) at order 6 and 5 + 3 beats 4 + 4 at order 7
    case 6: rc = launch_pass<6, 0, 5>(a, weighted, prog_bytes, st); if (rc == TXM_OK) rc = launch_pass<6, 5, 1>(a, weighted, prog_bytes, st); break;
    case 7: rc = launch_pass<7, 0, 5>(a, weighted, prog_bytes, st); if (rc == TXM_OK) rc = launch_pass<7, 5, 2>(a, weighted, prog_bytes, st); break;
    case 8: rc = launch_pass<8, 0, 5>(a, weighted, prog_bytes, st); if (rc == TXM_OK) rc = launch_pass<8, 5, 3>(a, weighted, prog_bytes, st); break;
    default: set_error("resample_i8: order out of range"); return TXM_ERR_INVALID;
  }
  return rc;
}

}  // namespace txm
