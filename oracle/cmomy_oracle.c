/*
 * oracle/cmomy_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, fp64) of the moment arithmetic that thermoextrap
 * delegates to the third-party package cmomy==0.24.0 (pinned in the reference's
 * uv.lock:454-455, floor in pyproject.toml:41).  cmomy's source is NOT under
 * /root/reference, so what is restated here is its *published algorithm*: the
 * one-pass weighted central (co)moment update of Pebay, "Formulas for robust,
 * one-pass parallel computation of covariances and arbitrary-order statistical
 * moments", SAND2008-6212 -- sequential in sample order, exactly the loop
 * structure the reference reaches through
 *
 *   cmomy.wrap_reduce_vals(xv, uv, weight, mom=(1, order))   src/thermoextrap/data.py:1632-1640, 1194-1203, 530-532
 *   cmomy.wrap_reduce_vals(uv, mom=order)                    src/thermoextrap/data.py:485, 528, 1183-1191
 *   cmomy.wrap_resample_vals(xv, uv, weight, sampler, ...)   src/thermoextrap/data.py:1354-1366, 1803-1810
 *   CentralMomentsData.resample_and_reduce(sampler)          src/thermoextrap/data.py:1048-1052
 *   CentralMomentsData.reduce(dim)                           src/thermoextrap/data.py:996
 *   CentralMomentsData.cmom() / .rmom()                      src/thermoextrap/data.py:844-852
 *   cmomy.convert.moments_type(raw, to="central")            src/thermoextrap/data.py:1109-1115
 *   IndexSampler.freq  (indices -> freq histogram)           src/thermoextrap/data.py:420-423 (sampler.indices is the only attribute read)
 *
 * Array convention (verified against the reference's notebook outputs, see
 * tests/golden/kat_notebooks.json): trailing dims (xmom=2, umom=K=order+1):
 *   [0,0] = sum of weights, [1,0] = <x>, [0,1] = <u>,
 *   [a,b] = < (x-<x>)^a (u-<u>)^b >   otherwise (weight-normalised).
 * 1-D (mom_ndim=1): [0] = sum w, [1] = <u>, [k>=2] = <du^k>.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (thermoextrap_amd/) never does.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXK 16 /* max order+1 supported by the oracle */

static double binom_tab[MAXK + 1][MAXK + 1];
static int binom_ready = 0;

static void binom_init(void) {
  if (binom_ready) return;
  for (int n = 0; n <= MAXK; ++n) {
    binom_tab[n][0] = 1.0;
    for (int k = 1; k <= n; ++k)
      binom_tab[n][k] = (k == n) ? 1.0 : binom_tab[n - 1][k - 1] + binom_tab[n - 1][k];
    for (int k = n + 1; k <= MAXK; ++k) binom_tab[n][k] = 0.0;
  }
  binom_ready = 1;
}

/* -------------------------------------------------------------------------
 * Co-moment state merge (Pebay 2008, eq. 2.9 generalised to weights).
 * `out` and `oth` are [A+1][K] states in the cmomy convention, A = x-order (1
 * for thermoextrap, data.py:1200 mom=(1, order)).
 * New state = union of the two weighted sets, `oth` weight scaled by `scale`.
 * ------------------------------------------------------------------------- */
static void push_data_scale_cov(double *out, const double *oth, double scale, int A, int K) {
  const double wB = oth[0] * scale;
  if (wB == 0.0) return;
  const double wA = out[0];
  if (wA == 0.0) {
    memcpy(out, oth, sizeof(double) * (size_t)(A + 1) * K);
    out[0] = wB;
    return;
  }
  const double W = wA + wB;
  const double fB = wB / W, fA = wA / W;
  const double dx = oth[1 * K + 0] - out[1 * K + 0];
  const double du = (K > 1) ? oth[0 * K + 1] - out[0 * K + 1] : 0.0;

  double nw[(2) * MAXK]; /* A <= 1 supported */
  /* helper accessors with the (0,0)=1, first-order=0 convention */
#define MU(p, a, b) (((a) == 0 && (b) == 0) ? 1.0 : (((a) + (b)) == 1 ? 0.0 : (p)[(a) * K + (b)]))
  for (int a = A; a >= 0; --a) {
    for (int b = K - 1; b >= 0; --b) {
      if (a + b < 2) continue;
      double acc = 0.0;
      double dxi = 1.0;
      for (int i = 0; i <= a; ++i) {
        double duj = 1.0;
        for (int j = 0; j <= b; ++j) {
          const int n = i + j;
          const double cA = fA * pow(-fB, n);
          const double cB = fB * pow(fA, n);
          acc += binom_tab[a][i] * binom_tab[b][j] * dxi * duj *
                 (cA * MU(out, a - i, b - j) + cB * MU(oth, a - i, b - j));
          duj *= du;
        }
        dxi *= dx;
      }
      nw[a * K + b] = acc;
    }
  }
#undef MU
  for (int a = 0; a <= A; ++a)
    for (int b = 0; b < K; ++b)
      if (a + b >= 2) out[a * K + b] = nw[a * K + b];
  out[0] = W;
  out[1 * K + 0] += fB * dx;
  if (K > 1) out[0 * K + 1] += fB * du;
}

/* Single weighted observation (x, u, w): the per-sample `push_val` loop body of
 * cmomy's reduce_vals / resample_vals.  A single point has all central moments
 * zero, so the merge collapses to
 *   mu'[a,b] = sum_{i,j} C(a,i)C(b,j) dx^i du^j fA (-fB)^(i+j) mu[a-i,b-j]
 *              + dx^a du^b fB fA^(a+b)
 * computed in descending (a,b) so it can be done in place.
 */
static void push_val_cov(double *out, double x, double u, double w, int K) {
  if (w == 0.0) return;
  const double wA = out[0];
  const double W = wA + w;
  const double fB = w / W, fA = 1.0 - fB;
  const double dx = x - out[K];
  const double du = (K > 1) ? u - out[1] : 0.0;
  out[0] = W;
  if (wA == 0.0) { /* first point: state is the point itself */
    out[K] = x;
    if (K > 1) out[1] = u;
    for (int b = 2; b < K; ++b) out[b] = 0.0;
    for (int b = 1; b < K; ++b) out[K + b] = 0.0;
    return;
  }
  /* powers */
  double dup[MAXK], mfB[MAXK + 1], pfA[MAXK + 1];
  dup[0] = 1.0;
  for (int j = 1; j < K; ++j) dup[j] = dup[j - 1] * du;
  mfB[0] = 1.0;
  pfA[0] = 1.0;
  for (int n = 1; n <= K; ++n) {
    mfB[n] = mfB[n - 1] * (-fB);
    pfA[n] = pfA[n - 1] * fA;
  }
#define MU1(a, b) (((a) == 0 && (b) == 0) ? 1.0 : (((a) + (b)) == 1 ? 0.0 : out[(a) * K + (b)]))
  /* x-row (a = 1) first (depends on a=0 row old values), b descending */
  for (int b = K - 1; b >= 1; --b) {
    double acc = 0.0;
    for (int j = 0; j <= b; ++j) {
      /* i = 0 */
      acc += binom_tab[b][j] * dup[j] * fA * mfB[j] * MU1(1, b - j);
      /* i = 1 */
      acc += binom_tab[b][j] * dx * dup[j] * fA * mfB[j + 1] * MU1(0, b - j);
    }
    acc += dx * dup[b] * fB * pfA[b + 1];
    out[K + b] = acc;
  }
  /* u-row (a = 0), b descending */
  for (int b = K - 1; b >= 2; --b) {
    double acc = 0.0;
    for (int j = 0; j <= b; ++j) acc += binom_tab[b][j] * dup[j] * fA * mfB[j] * MU1(0, b - j);
    acc += dup[b] * fB * pfA[b];
    out[b] = acc;
  }
#undef MU1
  out[K] += fB * dx;
  if (K > 1) out[1] += fB * du;
}

/* 1-D variant (mom_ndim = 1): state [M] = {W, <u>, <du^2>, ...} */
static void push_val_1d(double *out, double u, double w, int M) {
  if (w == 0.0) return;
  const double wA = out[0];
  const double W = wA + w;
  const double fB = w / W, fA = 1.0 - fB;
  out[0] = W;
  if (wA == 0.0) {
    if (M > 1) out[1] = u;
    for (int b = 2; b < M; ++b) out[b] = 0.0;
    return;
  }
  const double du = (M > 1) ? u - out[1] : 0.0;
  double dup[MAXK + 1], mfB[MAXK + 1], pfA[MAXK + 1];
  dup[0] = mfB[0] = pfA[0] = 1.0;
  for (int j = 1; j < M; ++j) {
    dup[j] = dup[j - 1] * du;
    mfB[j] = mfB[j - 1] * (-fB);
    pfA[j] = pfA[j - 1] * fA;
  }
  for (int b = M - 1; b >= 2; --b) {
    double acc = 0.0;
    for (int j = 0; j <= b; ++j) {
      const int c = b - j;
      const double mu = (c == 0) ? 1.0 : (c == 1 ? 0.0 : out[c]);
      acc += binom_tab[b][j] * dup[j] * fA * mfB[j] * mu;
    }
    acc += dup[b] * fB * pfA[b];
    out[b] = acc;
  }
  if (M > 1) out[1] += fB * du;
}

/* =========================================================================
 * Exported entry points.  x is addressed x[i*ldx_s + c*ldx_c] (element strides)
 * so both (rec, val) and (val, rec) layouts can be fed without copies.
 * out is [C][2][K] (reduce) or [nrep][C][2][K] (resample), C-contiguous.
 * ========================================================================= */

/* cmomy.wrap_reduce_vals(x, u, weight=w, mom=(1, order)) -- data.py:1632-1640 */
void orc_reduce_vals(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                     const double *w /* nullable */, int64_t N, int64_t C, int order,
                     double *out, int nthreads) {
  binom_init();
  const int K = order + 1;
  memset(out, 0, sizeof(double) * (size_t)C * 2 * K);
  if (nthreads < 1) nthreads = 1;
  /* cmomy's `parallel=True` spreads the broadcast (observable) dims over
   * threads, never the sample axis; same here. */
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int64_t c = 0; c < C; ++c) {
    double *o = out + c * 2 * K;
    for (int64_t i = 0; i < N; ++i)
      push_val_cov(o, x[i * ldx_s + c * ldx_c], u[i], w ? w[i] : 1.0, K);
  }
}

/* cmomy.wrap_reduce_vals(u, weight=w, mom=M-1) on R independent rows
 * (data.py:1183-1191: x_is_u path, mom = order + 1).  u[r*ldu_r + i*ldu_s]. */
void orc_reduce_vals_1d(const double *u, int64_t ldu_r, int64_t ldu_s, const double *w,
                        int64_t N, int64_t R, int M, double *out, int nthreads) {
  binom_init();
  memset(out, 0, sizeof(double) * (size_t)R * M);
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    double *o = out + r * M;
    for (int64_t i = 0; i < N; ++i) push_val_1d(o, u[r * ldu_r + i * ldu_s], w ? w[i] : 1.0, M);
  }
}

/* cmomy.wrap_resample_vals(x, u, weight=w, sampler(freq), mom=(1, order))
 * data.py:1803-1810.  freq is [nrep][N] int64; weight of sample i in replicate
 * r is w_i * freq[r,i].  out [nrep][C][2][K]  (the reference transposes rep to
 * the front afterwards, data.py:1812). */
void orc_resample_vals(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                       const double *w, const int64_t *freq, int64_t N, int64_t C,
                       int64_t nrep, int order, double *out, int nthreads) {
  binom_init();
  const int K = order + 1;
  memset(out, 0, sizeof(double) * (size_t)nrep * C * 2 * K);
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static) collapse(2)
  for (int64_t r = 0; r < nrep; ++r) {
    for (int64_t c = 0; c < C; ++c) {
      double *o = out + (r * C + c) * 2 * K;
      const int64_t *f = freq + r * N;
      for (int64_t i = 0; i < N; ++i) {
        if (f[i] == 0) continue;
        push_val_cov(o, x[i * ldx_s + c * ldx_c], u[i], (w ? w[i] : 1.0) * (double)f[i], K);
      }
    }
  }
}

/* CentralMomentsData.reduce(dim=rec) -- data.py:996.
 * data [nrec][C][2][K] -> out [C][2][K] */
void orc_reduce_data(const double *data, int64_t nrec, int64_t C, int order, double *out) {
  binom_init();
  const int K = order + 1;
  memset(out, 0, sizeof(double) * (size_t)C * 2 * K);
  for (int64_t c = 0; c < C; ++c)
    for (int64_t i = 0; i < nrec; ++i)
      push_data_scale_cov(out + c * 2 * K, data + (i * C + c) * 2 * K, 1.0, 1, K);
}

/* CentralMomentsData.resample_and_reduce(sampler) -- data.py:1048-1052.
 * data [nrec][C][2][K], freq [nrep][nrec] -> out [nrep][C][2][K] */
void orc_resample_data(const double *data, const int64_t *freq, int64_t nrec, int64_t C,
                       int64_t nrep, int order, double *out) {
  binom_init();
  const int K = order + 1;
  memset(out, 0, sizeof(double) * (size_t)nrep * C * 2 * K);
  for (int64_t r = 0; r < nrep; ++r)
    for (int64_t c = 0; c < C; ++c)
      for (int64_t i = 0; i < nrec; ++i)
        push_data_scale_cov(out + (r * C + c) * 2 * K, data + (i * C + c) * 2 * K,
                            (double)freq[r * nrec + i], 1, K);
}

/* central -> raw  (CentralMomentsData.rmom(), data.py:845-847) and
 * raw -> central  (cmomy.convert.moments_type(to="central"), data.py:1109-1115)
 * on n states [2][K].  In both forms [0,0] carries the weight and is copied.
 *   <x^a u^b> = sum_{i<=a, j<=b} C(a,i)C(b,j) <dx^i du^j> <x>^(a-i) <u>^(b-j)
 */
void orc_convert_cov(const double *in, double *out, int64_t n, int order, int to_central) {
  binom_init();
  const int K = order + 1;
  for (int64_t s = 0; s < n; ++s) {
    const double *p = in + s * 2 * K;
    double *q = out + s * 2 * K;
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
            else if (to_central) m = p[i * K + j];          /* raw moment */
            else m = (i + j == 1) ? 0.0 : p[i * K + j];     /* central moment */
            acc += binom_tab[a][i] * binom_tab[b][j] * m * pow(sx, a - i) * pow(su, b - j);
          }
        q[a * K + b] = acc;
      }
  }
}

/* 1-D version on n states [M] */
void orc_convert_1d(const double *in, double *out, int64_t n, int M, int to_central) {
  binom_init();
  for (int64_t s = 0; s < n; ++s) {
    const double *p = in + s * M;
    double *q = out + s * M;
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
        acc += binom_tab[b][j] * m * pow(su, b - j);
      }
      q[b] = acc;
    }
  }
}

/* indices [nrep][nsamp] -> freq [nrep][ndat]; returns -1 on out-of-range index
 * (cmomy's indices_to_freq; the sampler semantics are pinned by
 * tests/golden/kat_notebooks.json). */
int orc_indices_to_freq(const int64_t *idx, int64_t nrep, int64_t nsamp, int64_t ndat,
                        int64_t *freq) {
  memset(freq, 0, sizeof(int64_t) * (size_t)nrep * ndat);
  for (int64_t r = 0; r < nrep; ++r)
    for (int64_t k = 0; k < nsamp; ++k) {
      const int64_t j = idx[r * nsamp + k];
      if (j < 0 || j >= ndat) return -1;
      freq[r * ndat + j] += 1;
    }
  return 0;
}

/* -------------------------------------------------------------------------
 * Definitional "truth" in extended precision (x87 long double, 64-bit
 * mantissa): two-pass weighted central comoments.  Used to judge BOTH the
 * fp64 Pebay restatement above and the HIP kernels against the mathematical
 * definition (data.py:1233-1236 docstring of from_data).
 * freq may be NULL (plain reduce) or a [N] row of counts.
 * ------------------------------------------------------------------------- */
static void truth_cov_one(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w,
                          const int64_t *freq, int64_t N, int64_t c, int order, double *o) {
  const int K = order + 1;
  long double W = 0, sx = 0, su = 0;
  for (int64_t i = 0; i < N; ++i) {
    long double wi = (w ? w[i] : 1.0L) * (freq ? (long double)freq[i] : 1.0L);
    W += wi;
    sx += wi * x[i * ldx_s + c * ldx_c];
    su += wi * u[i];
  }
  long double mx = sx / W, mu = su / W;
  long double acc[2][MAXK];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < K; ++b) acc[a][b] = 0;
  for (int64_t i = 0; i < N; ++i) {
    long double wi = (w ? w[i] : 1.0L) * (freq ? (long double)freq[i] : 1.0L);
    if (wi == 0) continue;
    long double dx = x[i * ldx_s + c * ldx_c] - mx, du = u[i] - mu, p = wi;
    for (int b = 0; b < K; ++b) {
      acc[0][b] += p;
      acc[1][b] += p * dx;
      p *= du;
    }
  }
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < K; ++b) o[a * K + b] = (double)(acc[a][b] / W);
  o[0] = (double)W;
  o[K] = (double)mx;
  if (K > 1) o[1] = (double)mu;
}

void orc_truth_cov(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                   const double *w, const int64_t *freq, int64_t N, int64_t C, int order,
                   double *out) {
  const int K = order + 1;
  for (int64_t c = 0; c < C; ++c) truth_cov_one(x, ldx_s, ldx_c, u, w, freq, N, c, order, out + c * 2 * K);
}

/* R frequency rows at once, out [R][C][2][K]; the (row, column) pairs are independent: threads over them (the
   arithmetic of one pair is exactly orc_truth_cov's) */
void orc_truth_cov_multi(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w,
                         const int64_t *freq, int64_t R, int64_t N, int64_t C, int order, int nthreads,
                         double *out) {
  const int K = order + 1;
  const int64_t tasks = R * C;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
  for (int64_t t = 0; t < tasks; ++t) {
    const int64_t r = t / C, c = t % C;
    truth_cov_one(x, ldx_s, ldx_c, u, w, freq + r * N, N, c, order, out + (r * C + c) * 2 * K);
  }
}

void orc_truth_1d(const double *u, int64_t ldu_s, const double *w, const int64_t *freq,
                  int64_t N, int M, double *out) {
  long double W = 0, su = 0;
  for (int64_t i = 0; i < N; ++i) {
    long double wi = (w ? w[i] : 1.0L) * (freq ? (long double)freq[i] : 1.0L);
    W += wi;
    su += wi * u[i * ldu_s];
  }
  long double mu = su / W, acc[MAXK + 1];
  for (int b = 0; b < M; ++b) acc[b] = 0;
  for (int64_t i = 0; i < N; ++i) {
    long double wi = (w ? w[i] : 1.0L) * (freq ? (long double)freq[i] : 1.0L);
    long double du = u[i * ldu_s] - mu, p = wi;
    for (int b = 0; b < M; ++b) {
      acc[b] += p;
      p *= du;
    }
  }
  for (int b = 0; b < M; ++b) out[b] = (double)(acc[b] / W);
  out[0] = (double)W;
  if (M > 1) out[1] = (double)mu;
}
