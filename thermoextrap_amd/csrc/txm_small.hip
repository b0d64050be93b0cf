// txm_small.hip -- the small kernels of the path: raw<->central conversion
// (cmom()/rmom()/convert.moments_type, data.py:844-852, 1109-1115), block
// bootstrap / merge of pre-reduced states (resample_and_reduce / reduce,
// data.py:1048-1052, 996) and the table-driven derivative evaluator
// (models.py:317-383).  None of these is bandwidth- or flop-relevant
// (<= MBs); they exist so the whole path samples -> derivs stays on the device.
#include "txm_common.h"

namespace txm {

__device__ __forceinline__ double binom_d(int n, int k) {
  double c = 1.0;
  for (int i = 1; i <= k; ++i) c = c * (double)(n - k + i) / (double)i;
  return c;
}

__device__ __forceinline__ double ipow(double b, int e) {
  double r = 1.0;
  for (int i = 0; i < e; ++i) r *= b;
  return r;
}

// ---- conversion -------------------------------------------------------------
// cov states [2][K]; thread per state.  to_central: in = raw, out = central.
__global__ __launch_bounds__(256) void convert_cov_kernel(const double *__restrict__ in,
                                                          double *__restrict__ out, int64_t n, int K,
                                                          int to_central) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  double p[2 * TXM_MAXK], q[2 * TXM_MAXK];
  for (int e = 0; e < 2 * K; ++e) p[e] = in[s * 2 * K + e];
  const double xa = p[K], ua = (K > 1) ? p[1] : 0.0;
  const double sx = to_central ? -xa : xa, su = to_central ? -ua : ua;
  for (int a = 0; a <= 1; ++a)
    for (int b = 0; b < K; ++b) {
      if (a == 0 && b == 0) { q[0] = p[0]; continue; }
      if (to_central && a + b == 1) { q[a * K + b] = p[a * K + b]; continue; }
      double acc = 0.0;
      for (int i = 0; i <= a; ++i)
        for (int j = 0; j <= b; ++j) {
          double m;
          if (i == 0 && j == 0) m = 1.0;
          else if (to_central) m = p[i * K + j];
          else m = (i + j == 1) ? 0.0 : p[i * K + j];
          acc += binom_d(b, j) * m * ipow(sx, a - i) * ipow(su, b - j);
        }
      q[a * K + b] = acc;
    }
  for (int e = 0; e < 2 * K; ++e) out[s * 2 * K + e] = q[e];
}

__global__ __launch_bounds__(256) void convert_1d_kernel(const double *__restrict__ in,
                                                         double *__restrict__ out, int64_t n, int M,
                                                         int to_central) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  double p[TXM_MAXK + 1], q[TXM_MAXK + 1];
  for (int e = 0; e < M; ++e) p[e] = in[s * M + e];
  const double ua = (M > 1) ? p[1] : 0.0;
  const double su = to_central ? -ua : ua;
  for (int b = 0; b < M; ++b) {
    if (b == 0) { q[0] = p[0]; continue; }
    if (to_central && b == 1) { q[1] = p[1]; continue; }
    double acc = 0.0;
    for (int j = 0; j <= b; ++j) {
      double m;
      if (j == 0) m = 1.0;
      else if (to_central) m = p[j];
      else m = (j == 1) ? 0.0 : p[j];
      acc += binom_d(b, j) * m * ipow(su, b - j);
    }
    q[b] = acc;
  }
  for (int e = 0; e < M; ++e) out[s * M + e] = q[e];
}

// ---- merge of pre-reduced states -------------------------------------------
// Instead of cmomy's sequential pairwise Pebay merges, every state is
// re-expressed as weight-scaled power sums about one common pivot per column
// (mean of the block means), summed with the bootstrap counts, and shifted
// back once.  Same algebra as the sample kernels, so partials add exactly.

// pivot[c] = {mean over records of <u>, mean over records of <x>}
__global__ __launch_bounds__(256) void data_pivot_kernel(const double *__restrict__ data,
                                                         int64_t nrec, int64_t C, int K,
                                                         double *__restrict__ pivot) {
  const int c = blockIdx.x;
  double su = 0.0, sx = 0.0;
  for (int64_t i = threadIdx.x; i < nrec; i += blockDim.x) {
    const double *st = data + (i * C + c) * 2 * K;
    su += (K > 1) ? st[1] : 0.0;
    sx += st[K];
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = su;
  sh[1][threadIdx.x] = sx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double pu = sh[0][0] / (double)nrec, px = sh[1][0] / (double)nrec;
    if (!(pu - pu == 0.0)) pu = 0.0;
    if (!(px - px == 0.0)) px = 0.0;
    pivot[2 * c] = pu;
    pivot[2 * c + 1] = px;
  }
}

// P[i][c][2][K]: weight-scaled pivot sums of state i.
__global__ __launch_bounds__(256) void data_to_sums_kernel(const double *__restrict__ data,
                                                           int64_t nrec, int64_t C, int K,
                                                           const double *__restrict__ pivot,
                                                           double *__restrict__ P) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nrec * C) return;
  const int64_t c = e % C;
  const double *st = data + e * 2 * K;
  double *o = P + e * 2 * K;
  const double W = st[0];
  const double du = ((K > 1) ? st[1] : 0.0) - pivot[2 * c];
  const double dx = st[K] - pivot[2 * c + 1];
  for (int b = 0; b < K; ++b) {
    double a0 = 0.0, a1 = 0.0;
    for (int j = 0; j <= b; ++j) {
      const double m0 = (j == 0) ? 1.0 : (j == 1 ? 0.0 : st[j]);
      const double m1 = (j == 0) ? 0.0 : st[K + j];
      const double f = binom_d(b, j) * ipow(du, b - j);
      a0 += f * m0;
      a1 += f * (m1 + dx * m0);
    }
    o[b] = (W == 0.0) ? 0.0 : W * a0;
    o[K + b] = (W == 0.0) ? 0.0 : W * a1;
  }
}

// block per (c, r): out[r][c] = shift( sum_i f[r][i] * P[i][c] )
template <int K>
__global__ __launch_bounds__(256) void data_combine_kernel(const double *__restrict__ P,
                                                           const int64_t *__restrict__ freq,
                                                           int64_t nrec, int64_t C,
                                                           const double *__restrict__ pivot,
                                                           double *__restrict__ out) {
  const int64_t c = blockIdx.x, r = blockIdx.y;
  double acc[2 * K];
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) acc[q] = 0.0;
  for (int64_t i = threadIdx.x; i < nrec; i += blockDim.x) {
    const double f = freq ? (double)freq[r * nrec + i] : 1.0;
    if (f == 0.0) continue;
    const double *src = P + (i * C + c) * 2 * K;
#pragma unroll
    for (int q = 0; q < 2 * K; ++q) acc[q] = fma(f, src[q], acc[q]);
  }
  __shared__ double sh[256][2 * K];
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) sh[threadIdx.x][q] = acc[q];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
#pragma unroll
      for (int q = 0; q < 2 * K; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + off][q];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double S0[K], S1[K], st[2 * K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      S0[j] = sh[0][j];
      S1[j] = sh[0][K + j];
    }
    pivot_sums_to_state<K>(S0, S1, pivot[2 * c], pivot[2 * c + 1], st);
#pragma unroll
    for (int q = 0; q < 2 * K; ++q) out[(r * C + c) * 2 * K + q] = st[q];
  }
}

// ---- derivative polynomial evaluator ----------------------------------------
struct PolyDev {
  int32_t n_funcs, log_atom;
  const txm_atom *atoms;
  const int32_t *func_term0, *func_flags;
  const double *coef;
  const int32_t *term_fac0, *fac_atom, *fac_pow;
};

__global__ __launch_bounds__(256) void eval_poly_kernel(PolyDev t, const double *const *__restrict__ srcs,
                                                        int64_t nrep, int64_t nval,
                                                        double *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nrep * nval) return;
  const int64_t r = e / nval, v = e % nval;
  auto atom_val = [&](int a) {
    const txm_atom at = t.atoms[a];
    return srcs[at.src][r * at.s_rep + v * at.s_val + at.offset];
  };
  for (int f = 0; f < t.n_funcs; ++f) {
    double acc = 0.0;
    for (int tm = t.func_term0[f]; tm < t.func_term0[f + 1]; ++tm) {
      double prod = t.coef[tm];
      for (int k = t.term_fac0[tm]; k < t.term_fac0[tm + 1]; ++k) {
        const double b = atom_val(t.fac_atom[k]);
        const int p = t.fac_pow[k];
        double pw = 1.0;
        const int ap = p < 0 ? -p : p;
        for (int i = 0; i < ap; ++i) pw *= b;
        prod = p < 0 ? prod / pw : prod * pw;
      }
      acc += prod;
    }
    if (t.func_flags[f] & TXM_FUNC_MINUS_LOG) acc -= log(atom_val(t.log_atom));
    out[(size_t)f * nrep * nval + e] = acc;
  }
}

// ---- covariance over replicates ---------------------------------------------
// block per output value; two fixed-order passes (mean, then centred products).
__global__ __launch_bounds__(256) void cov_over_rep_kernel(const double *__restrict__ vals, int n_ord,
                                                           int64_t nrep, int64_t nval,
                                                           double *__restrict__ cov) {
  const int64_t v = blockIdx.x;
  __shared__ double red[256];
  __shared__ double mean[16];
  for (int a = 0; a < n_ord; ++a) {
    double acc = 0.0;
    for (int64_t r = threadIdx.x; r < nrep; r += blockDim.x) acc += vals[((size_t)a * nrep + r) * nval + v];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) mean[a] = red[0] / (double)nrep;
    __syncthreads();
  }
  for (int a = 0; a < n_ord; ++a)
    for (int b = a; b < n_ord; ++b) {
      double acc = 0.0;
      for (int64_t r = threadIdx.x; r < nrep; r += blockDim.x)
        acc += (vals[((size_t)a * nrep + r) * nval + v] - mean[a]) * (vals[((size_t)b * nrep + r) * nval + v] - mean[b]);
      red[threadIdx.x] = acc;
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        const double c = red[0] / (double)(nrep - 1);
        cov[((size_t)v * n_ord + a) * n_ord + b] = c;
        cov[((size_t)v * n_ord + b) * n_ord + a] = c;
      }
      __syncthreads();
    }
}

}  // namespace txm

using namespace txm;

extern "C" int txm_cov_over_rep(const double *vals, int32_t n_ord, int64_t nrep, int64_t nval, double *cov,
                                txm_stream stream) {
  TXM_REQUIRE(vals && cov, "cov_over_rep: null pointer");
  TXM_REQUIRE(n_ord >= 1 && n_ord <= 16 && nrep >= 2 && nval >= 1, "cov_over_rep: need 1 <= n_ord <= 16, nrep >= 2");
  TXM_REQUIRE(nval < ((int64_t)1 << 31), "cov_over_rep: nval too large");
  hipLaunchKernelGGL(cov_over_rep_kernel, dim3((unsigned)nval), dim3(256), 0, (hipStream_t)stream, vals,
                     (int)n_ord, nrep, nval, cov);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" int txm_convert_cov(const double *in, double *out, int64_t n, int order, int to_central,
                               txm_stream stream) {
  TXM_REQUIRE(in && out, "convert_cov: null pointer");
  TXM_REQUIRE(n >= 0 && order >= 0 && order <= TXM_MAX_ORDER, "convert_cov: bad n/order");
  if (n == 0) return TXM_OK;
  hipLaunchKernelGGL(convert_cov_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, n, order + 1, to_central);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" int txm_convert_1d(const double *in, double *out, int64_t n, int M, int to_central,
                              txm_stream stream) {
  TXM_REQUIRE(in && out, "convert_1d: null pointer");
  TXM_REQUIRE(n >= 0 && M >= 1 && M <= TXM_MAXK + 1, "convert_1d: bad n/M");
  if (n == 0) return TXM_OK;
  hipLaunchKernelGGL(convert_1d_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, n, M, to_central);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" size_t txm_resample_data_ws_bytes(int64_t nrec, int64_t C, int order) {
  if (nrec < 1 || C < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  return align_up((size_t)2 * C * sizeof(double), 256) +
         (size_t)nrec * C * 2 * (order + 1) * sizeof(double) + 256;
}

extern "C" int txm_resample_data(const double *data, const int64_t *freq, int64_t nrec, int64_t C,
                                 int64_t nrep, int order, double *out, void *ws, size_t ws_bytes,
                                 txm_stream stream) {
  TXM_REQUIRE(data && out && ws, "resample_data: null pointer");
  TXM_REQUIRE(nrec >= 1 && C >= 1 && nrep >= 1, "resample_data: need nrec, C, nrep >= 1");
  TXM_REQUIRE(C <= 65535 * 32 && nrep <= 65535, "resample_data: C or nrep too large");
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "resample_data: order out of range");
  TXM_REQUIRE(freq || nrep == 1, "resample_data: freq == NULL requires nrep == 1");
  if (ws_bytes < txm_resample_data_ws_bytes(nrec, C, order)) {
    set_error("resample_data: workspace too small");
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int K = order + 1;
  double *pivot = (double *)ws;
  double *P = (double *)((char *)ws + align_up((size_t)2 * C * sizeof(double), 256));
  hipLaunchKernelGGL(data_pivot_kernel, dim3((unsigned)C), dim3(256), 0, st, data, nrec, C, K, pivot);
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL(data_to_sums_kernel, dim3((unsigned)cdiv(nrec * C, 256)), dim3(256), 0, st, data,
                     nrec, C, K, pivot, P);
  TXM_LAUNCH_CHECK();
  dim3 grid((unsigned)C, (unsigned)nrep);
  switch (K) {
#define TXM_DC(KK)                                                                              \
  case KK:                                                                                      \
    hipLaunchKernelGGL((data_combine_kernel<KK>), grid, dim3(256), 0, st, P, freq, nrec, C, pivot, \
                       out);                                                                    \
    break;
    TXM_DC(1) TXM_DC(2) TXM_DC(3) TXM_DC(4) TXM_DC(5) TXM_DC(6) TXM_DC(7) TXM_DC(8) TXM_DC(9)
#undef TXM_DC
    default:
      set_error("resample_data: order out of range");
      return TXM_ERR_INVALID;
  }
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

namespace txm {
// thread per (alpha a, element m): Horner is avoided on purpose -- the terms are formed exactly as the host
// formula does (dalpha^k by repeated products, times the coefficient) so that cumsum / no_sum return them.
template <int MODE>
__global__ __launch_bounds__(256) void predict_taylor_kernel(const double *__restrict__ d, int n_ord, int64_t M,
                                                             const double *__restrict__ dalpha, int64_t na,
                                                             double *__restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t a = blockIdx.y;
  if (m >= M) return;
  const double da = dalpha[a];
  double p = 1.0, s = 0.0, fact = 1.0;
  for (int k = 0; k < n_ord; ++k) {
    if (k > 1) fact *= (double)k;  // exact: 15! < 2^53
    const double term = p * (d[(size_t)k * M + m] * (1.0 / fact));  // 1 / k! rounded once, as the host's taylor_series_norm
    s += term;
    if (MODE == TXM_TAYLOR_CUMSUM) out[((size_t)a * n_ord + k) * M + m] = s;
    if (MODE == TXM_TAYLOR_TERMS) out[((size_t)a * n_ord + k) * M + m] = term;
    p *= da;
  }
  if (MODE == TXM_TAYLOR_SUM) out[(size_t)a * M + m] = s;
}
}  // namespace txm

extern "C" int txm_predict_taylor(const double *derivs, int32_t n_ord, int64_t M, const double *dalpha,
                                  int64_t n_alpha, int32_t mode, double *out, txm_stream stream) {
  TXM_REQUIRE(derivs && dalpha && out, "predict_taylor: null pointer");
  TXM_REQUIRE(n_ord >= 1 && n_ord <= 16 && M >= 1 && n_alpha >= 1 && n_alpha <= 65535, "predict_taylor: bad sizes");
  dim3 grid((unsigned)cdiv(M, 256), (unsigned)n_alpha), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case TXM_TAYLOR_SUM:
      hipLaunchKernelGGL((predict_taylor_kernel<TXM_TAYLOR_SUM>), grid, block, 0, st, derivs, n_ord, M, dalpha, n_alpha, out);
      break;
    case TXM_TAYLOR_CUMSUM:
      hipLaunchKernelGGL((predict_taylor_kernel<TXM_TAYLOR_CUMSUM>), grid, block, 0, st, derivs, n_ord, M, dalpha, n_alpha, out);
      break;
    case TXM_TAYLOR_TERMS:
      hipLaunchKernelGGL((predict_taylor_kernel<TXM_TAYLOR_TERMS>), grid, block, 0, st, derivs, n_ord, M, dalpha, n_alpha, out);
      break;
    default:
      set_error("predict_taylor: unknown mode %d", mode);
      return TXM_ERR_INVALID;
  }
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" int txm_eval_poly(const txm_poly_table *tb, const double *const *srcs, int32_t n_srcs,
                             int64_t nrep, int64_t nval, double *out, txm_stream stream) {
  TXM_REQUIRE(tb && srcs && out, "eval_poly: null pointer");
  TXM_REQUIRE(tb->n_funcs >= 1 && tb->n_atoms >= 0 && tb->n_terms >= 0 && n_srcs >= 1,
              "eval_poly: bad table sizes");
  TXM_REQUIRE(nrep >= 1 && nval >= 1, "eval_poly: need nrep, nval >= 1");
  PolyDev t;
  t.n_funcs = tb->n_funcs;
  t.log_atom = tb->log_atom;
  t.atoms = tb->atoms;
  t.func_term0 = tb->func_term0;
  t.func_flags = tb->func_flags;
  t.coef = tb->coef;
  t.term_fac0 = tb->term_fac0;
  t.fac_atom = tb->fac_atom;
  t.fac_pow = tb->fac_pow;
  hipLaunchKernelGGL(eval_poly_kernel, dim3((unsigned)cdiv(nrep * nval, 256)), dim3(256), 0,
                     (hipStream_t)stream, t, srcs, nrep, nval, out);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}
