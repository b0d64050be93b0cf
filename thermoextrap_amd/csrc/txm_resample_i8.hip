// txm_resample_i8.hip -- the pre-pass of the int8 bootstrap path: per-window scale tables, the data-dependent
// precision guard and its fallback list (the contraction itself is txm_resample_i8t.hip).
//
// The sums (cmomy.wrap_resample_vals as called from thermoextrap data.py:1803-1810, 1354-1366):
//        S1[r][c][j] = sum_i f[r][i] * w_i * du_i^j * dx_ic      S0[r][j] = sum_i f[r][i] * w_i * du_i^j
// On gfx950 the FP64 MFMA shares the VALU datapath (tools/mfma_f64_peak4.hip), which caps the FP64 formulation near
// 64 TFLOP/s.  The int8 MFMA runs at 4.4 POP/s beside the VALU (tools/mfma_i8_probe2.hip), and ONE operand is already an
// exact small integer: the bootstrap count f[r][i] (a u8 from the sampler).  Only the data operand needs slicing:
//
//   per scaling window (256 tiles = 262144 samples on long series; 64 / 16 / 4 tiles on short ones -- a function of N
//   alone) the pre-pass measures max|du|, max|w|, max|dx_c|; inside a window every monomial
//   m = (w/wmax)(du/dumax)^j * (dx_c/dxmax_c) lies in [-1, 1] and  X = rint(m * 2^50)  is a 52-bit signed integer obtained
//   with ONE v_fma_f64 against the magic constant 1.5*2^52 (+ a per-byte bias of 0x80), whose mantissa bytes -- after an
//   XOR with 0x80 -- are seven signed base-256 digits  X = sum_i d_i 256^i, d_i in [-128, 127].  Then
//        sum_k f_k X_k = sum_i 256^i * (sum_k f_k d_ik)      exactly, in int32 accumulators,
//   flushed per window into the window's own FP64 slot per digit (x 256^i x the window descale).
//   Rounding: one rint per monomial at 2^-51 of the WINDOW maximum (unbiased) -- which is only harmless while the
//   window's typical monomial is not dwarfed by its maximum: the guard below checks exactly that, per window and column,
//   and hands the windows that fail to the FP64 kernel (txm_resample.hip, listed mode) inside the same call.
#include "txm_resample_i8.h"

#include <stdlib.h>
#include "txm_sampler.h"

namespace txm {

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
                                                       double *__restrict__ stats, const I8State *__restrict__ states) {
  if (states != nullptr) {  // batched: state blockIdx.y
    const I8State e = states[blockIdx.y];
    x = e.x; u = e.u; w = e.w; pivot = e.pivot; stats = e.stats;
  }
  // thread = (column pair cp, row phase r): 16 lanes read one 256-byte row with 16-byte loads
  const int64_t sb = blockIdx.x;
  const int64_t i0 = sb * sub_samples;
  const int64_t i1 = (i0 + sub_samples < N) ? i0 + sub_samples : N;
  const int tid = threadIdx.x, cp = tid & 15, r = tid >> 4;
  // A narrow state fills only cpw = ceil(C / 2) (rounded up to a power of two) of the 16 column-pair lanes of a row.  The other
  // lanes take other QUARTERS of the sub-block for the same column pairs: lane cp = (column pair cpa = cp % cpw, sub = cp / cpw)
  // walks the quarters sub, sub + 16 / cpw, ...  Every (quarter, row phase, column pair) is still summed by ONE thread in row
  // order, so the statistics are bit for bit those of the plain mapping -- 4 x the lanes at C <= 8 (config 5's collection:
  // 2.99 -> see profiles/r04_experiments.md).
  const int cpw = C > 16 ? 16 : C > 8 ? 8 : C > 4 ? 4 : C > 2 ? 2 : 1, nsubq = 16 / cpw;
  const int cpa = cp % cpw, qsub = cp / cpw;
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
  const int c0 = 2 * cpa;
  const int slot = r * 16 + cpa;  // where the reductions below look for this (row phase, column pair)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    shq[g][0][tid] = shq[g][1][tid] = 0.0;
    shqn[g][tid] = 0;
  }
  __syncthreads();  // (a slot is written below by the lane that walked its quarter, not by the thread that zeroed it)
  if (c0 < C && qsub < 4) {
    const bool two = c0 + 1 < C;
    const double px0 = pivot[1 + col0 + c0], px1 = two ? pivot[1 + col0 + c0 + 1] : 0.0;
    const int64_t gs = sub_samples / 4;  // rows per quarter (sub_samples is a multiple of 4096)
    for (int g = qsub; g < 4; g += nsubq) {
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
      shq[g][0][slot] = s0;
      shq[g][1][slot] = s1;
      shqn[g][slot] = n;
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
  if (nsubq > 1) {  // the lanes that shared a column pair (maxima / minima: exact in any order)
    if (tid < cpw)
      for (int q = 1; q < nsubq; ++q)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          shx[v][tid] = fmax(shx[v][tid], shx[v][tid + q * cpw]);
          shg[v][tid] = fmin(shg[v][tid], shg[v][tid + q * cpw]);
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
                                                      uint32_t *__restrict__ wflag, const I8State *__restrict__ states) {
  if (states != nullptr) {
    const I8State e = states[blockIdx.y];
    stats = e.stats; wtab = e.wtab; wflag = e.wflag;
  }
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
                                                      uint32_t *__restrict__ n_list, const I8State *__restrict__ states) {
  if (states != nullptr) {  // one block per state
    const I8State e = states[blockIdx.x];
    wflag = e.wflag; list = e.list; n_list = e.n_list;
  }
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
bool i8_supported(int64_t N, int64_t C, int64_t nrep, int K) {
  (void)nrep;
  return N >= SM_T && C >= 1 && C <= 64 * I8_CPAD && K >= 1 && K <= 8;  // C > 32: one launch per 32 columns
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
  // (batched: a.x is the host's word for "16-byte loads are fine for every state")
  const bool vec2 = ((reinterpret_cast<uintptr_t>(a.x + a.col0) & 15) == 0) && (a.ldx_s % 2 == 0);
  const unsigned S = (unsigned)a.S;
  if (vec2)
    hipLaunchKernelGGL(i8_stats_kernel<true>, dim3((unsigned)nsub_total, S), dim3(256), 0, st, a.x, a.ldx_s, a.u, a.w,
                       a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.pivot, K - 1, a.stats, a.states);
  else
    hipLaunchKernelGGL(i8_stats_kernel<false>, dim3((unsigned)nsub_total, S), dim3(256), 0, st, a.x, a.ldx_s, a.u, a.w,
                       a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.pivot, K - 1, a.stats, a.states);
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL(i8_table_kernel, dim3((unsigned)a.nwin, S), dim3(64), 0, st, a.stats, nsub, nsub_total, a.N, a.C,
                     a.win_tiles * SM_T, a.w != nullptr, K - 1, a.wtab, a.wflag, a.states);
  TXM_LAUNCH_CHECK();
  if (a.y != nullptr) {
    // the second matrix: its own column scales (order-0 monomial w dy), its guard flags OR-ed into the call's -- a
    // window either runs on the int8 kernel for both matrices or on the FP64 kernel for both
    const bool vy = ((reinterpret_cast<uintptr_t>(a.y + a.col0) & 15) == 0) && (a.ldy_s % 2 == 0);
    if (vy)
      hipLaunchKernelGGL(i8_stats_kernel<true>, dim3((unsigned)nsub_total), dim3(256), 0, st, a.y, a.ldy_s, a.u, a.w,
                         a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.ypivot, 0, a.stats, nullptr);
    else
      hipLaunchKernelGGL(i8_stats_kernel<false>, dim3((unsigned)nsub_total), dim3(256), 0, st, a.y, a.ldy_s, a.u, a.w,
                         a.N, a.C, a.col0, sub_tiles * SM_T, nsub == 1 ? 1 : 0, a.ypivot, 0, a.stats, nullptr);
    TXM_LAUNCH_CHECK();
    hipLaunchKernelGGL(i8_table_kernel, dim3((unsigned)a.nwin), dim3(64), 0, st, a.stats, nsub, nsub_total, a.N, a.C,
                       a.win_tiles * SM_T, a.w != nullptr, 0, a.ywtab, a.yflag, nullptr);
    TXM_LAUNCH_CHECK();
    hipLaunchKernelGGL(i8_or_flags_kernel, dim3((unsigned)cdiv(a.nwin, 256)), dim3(256), 0, st, a.wflag, a.yflag, a.nwin);
    TXM_LAUNCH_CHECK();
  }
  if (a.states == nullptr) TXM_HIP(hipMemsetAsync(a.n_list, 0, 256, st));  // (batched: the caller clears the states' words)
  hipLaunchKernelGGL(i8_list_kernel, dim3(S), dim3(256), 0, st, a.wflag, a.nwin, a.win_tiles, a.sub_tiles, a.list,
                     a.n_list, a.states);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// every shape of the int8 path runs the transposing-read kernel (txm_resample_i8t.hip): one power per observable column
// for C > 16, the quad-sharing variant for narrow states.  (The round-1/2 kernel that sliced the digits on the VALU --
// resample_i8_kernel, DESIGN.md 4.2b "history" -- lived here until round 3.)
// (A sixteen-wave cut of the wide shape was measured 8 % slower in round 4: tools/experiments/.)
int launch_resample_i8(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st) {
  return launch_resample_i8t(a, K, weighted, prog_bytes, st);
}

}  // namespace txm
