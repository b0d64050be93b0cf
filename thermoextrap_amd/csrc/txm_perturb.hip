// txm_perturb.hip -- exponential-reweighting ("perturbation") averages
// (PerturbModel.predict, reference models.py:1019-1039): another one-pass,
// HBM-bound streaming reduction over the same (rec, val) sample layout.
//
//   out[a][c] = sum_i f_i x_ic e^{-da (u_i - uref_a)} / sum_i f_i e^{-da (u_i - uref_a)}
//
// All n_alpha perturbations are evaluated in one pass (the samples are read
// once: 8*(C+1) bytes per sample regardless of n_alpha); uref_a is the extreme
// of u that makes the largest exponent zero (the reference subtracts the max
// of -da*u for the same reason).  Thread layout as reduce_rowmajor_kernel: a lane
// owns VEC fixed columns, 2^L lanes span a row; sums live in registers.
#include "txm_common.h"

namespace txm {

constexpr int PB_BLOCK = 256;
constexpr int PB_MAXA = 8;

__global__ __launch_bounds__(PB_BLOCK) void minmax_kernel(const double *__restrict__ u, int64_t N,
                                                          double *__restrict__ part) {
  double lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * PB_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PB_BLOCK) {
    const double v = u[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  __shared__ double sl[PB_BLOCK], sh[PB_BLOCK];
  sl[threadIdx.x] = lo;
  sh[threadIdx.x] = hi;
  __syncthreads();
  for (int off = PB_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sl[threadIdx.x] = sl[threadIdx.x + off] < sl[threadIdx.x] ? sl[threadIdx.x + off] : sl[threadIdx.x];
      sh[threadIdx.x] = sh[threadIdx.x + off] > sh[threadIdx.x] ? sh[threadIdx.x + off] : sh[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = sl[0];
    part[2 * blockIdx.x + 1] = sh[0];
  }
}

__global__ void minmax_final_kernel(const double *__restrict__ part, int nblk, double *__restrict__ mm) {
  double lo = INFINITY, hi = -INFINITY;
  for (int b = 0; b < nblk; ++b) {
    lo = part[2 * b] < lo ? part[2 * b] : lo;
    hi = part[2 * b + 1] > hi ? part[2 * b + 1] : hi;
  }
  mm[0] = lo;
  mm[1] = hi;
}

struct PerturbArgs {
  double da[PB_MAXA];
};

// partial layout: [rep][gridDim.x][cols_pad][NA][2]  (num, den)
template <int NA, int VEC, int LPR_LOG2, bool FREQ>
__global__ __launch_bounds__(PB_BLOCK) void perturb_kernel(const double *__restrict__ x, int64_t ldx_s,
                                                           const double *__restrict__ u, int64_t N, int64_t C,
                                                           const PerturbArgs pa, const double *__restrict__ mm,
                                                           const int64_t *__restrict__ freq,
                                                           double *__restrict__ partial) {
  constexpr int LPR = 1 << LPR_LOG2;
  constexpr int ROWS = PB_BLOCK / LPR;
  const int tid = threadIdx.x;
  const int lir = tid & (LPR - 1), rib = tid >> LPR_LOG2;
  const int64_t col0 = (int64_t)blockIdx.y * (LPR * VEC) + (int64_t)lir * VEC;
  const bool col_ok = col0 < C;
  const int64_t rep = blockIdx.z;
  double uref[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) uref[a] = pa.da[a] >= 0.0 ? mm[0] : mm[1];
  double num[NA][VEC], den[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    den[a] = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) num[a][v] = 0.0;
  }
  const int64_t stride = (int64_t)gridDim.x * ROWS;
  // lanes of this column chunk that own real columns (block-uniform): the exp-sharing
  // path needs lanes 0..NA-1 of every row group to be active
  const int64_t cols_left = C - (int64_t)blockIdx.y * (LPR * VEC);
  const int64_t nvalid = (cols_left + VEC - 1) / VEC;
  const bool share = (LPR >= NA) && (LPR <= 64) && (NA > 1) && (nvalid >= NA);
  if (col_ok) {
    for (int64_t i = (int64_t)blockIdx.x * ROWS + rib; i < N; i += stride) {
      double xv[VEC];
      if constexpr (VEC == 2) {
        const double2 t2 = *reinterpret_cast<const double2 *>(x + i * ldx_s + col0);
        xv[0] = t2.x;
        xv[1] = t2.y;
      } else {
        xv[0] = x[i * ldx_s + col0];
      }
      const double ui = u[i];
      double fw = 1.0;
      if constexpr (FREQ) fw = (double)freq[rep * N + i];
      if (share) {
        // exp() is ~25 vector instructions per wave whatever the number of active lanes, and
        // the LPR lanes of a row would all compute the same NA values.  Instead lane `lir`
        // evaluates alpha number lir (one exp sequence covers all alphas of the row) and the
        // row's lanes fetch the NA weights with cross-lane reads.
        const int a_mine = lir < NA ? lir : 0;
        double da_mine = pa.da[0], ur_mine = uref[0];
#pragma unroll
        for (int a = 1; a < NA; ++a)
          if (a_mine == a) {
            da_mine = pa.da[a];
            ur_mine = uref[a];
          }
        const double w_mine = fw * exp(-da_mine * (ui - ur_mine));
        const int lane = tid & 63;
        const int row_lane0 = lane & ~(LPR - 1);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const double w = __shfl(w_mine, row_lane0 + a);
          den[a] += w;
#pragma unroll
          for (int v = 0; v < VEC; ++v) num[a][v] = fma(w, xv[v], num[a][v]);
        }
      } else {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const double w = fw * exp(-pa.da[a] * (ui - uref[a]));
          den[a] += w;
#pragma unroll
          for (int v = 0; v < VEC; ++v) num[a][v] = fma(w, xv[v], num[a][v]);
        }
      }
    }
  }
  // block reduction over the ROWS row slots of a column (fixed order)
  constexpr int NV = NA * (1 + VEC);
  __shared__ double sh[PB_BLOCK * NV];
  double *mine = sh + (size_t)tid * NV;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    mine[a * (1 + VEC)] = den[a];
#pragma unroll
    for (int v = 0; v < VEC; ++v) mine[a * (1 + VEC) + 1 + v] = num[a][v];
  }
  __syncthreads();
  for (int e = tid; e < LPR * NV; e += PB_BLOCK) {
    const int l = e / NV, q = e % NV;
    double acc = 0.0;
    for (int r = 0; r < ROWS; ++r) acc += sh[((size_t)r * LPR + l) * NV + q];
    const int a = q / (1 + VEC), k = q % (1 + VEC);
    const size_t cols_pad = (size_t)gridDim.y * LPR * VEC;
    double *dst = partial + (((size_t)rep * gridDim.x + blockIdx.x) * cols_pad +
                             (size_t)blockIdx.y * LPR * VEC + (size_t)l * VEC) * NA * 2;
    if (k == 0) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) dst[((size_t)v * NA + a) * 2 + 1] = acc;
    } else {
      dst[((size_t)(k - 1) * NA + a) * 2] = acc;
    }
  }
}

__global__ __launch_bounds__(PB_BLOCK) void perturb_final_kernel(const double *__restrict__ partial, int nblk,
                                                                 int64_t cols_pad, int64_t C, int NA,
                                                                 double *__restrict__ out) {
  const int64_t c = blockIdx.x, rep = blockIdx.y;
  const int a = blockIdx.z;
  double num = 0.0, den = 0.0;
  for (int b = threadIdx.x; b < nblk; b += PB_BLOCK) {
    const double *src = partial + ((((size_t)rep * nblk + b) * cols_pad + c) * NA + a) * 2;
    num += src[0];
    den += src[1];
  }
  __shared__ double sn[PB_BLOCK], sd[PB_BLOCK];
  sn[threadIdx.x] = num;
  sd[threadIdx.x] = den;
  __syncthreads();
  for (int off = PB_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sn[threadIdx.x] += sn[threadIdx.x + off];
      sd[threadIdx.x] += sd[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[((size_t)rep * NA + a) * C + c] = sn[0] / sd[0];
}

struct PbPlan {
  int vec, l2, chunks, gx;
  int64_t cols_pad;
};

static PbPlan pb_plan(const double *x, int64_t ldx_s, int64_t N, int64_t C, int64_t nrep) {
  PbPlan p;
  const bool v2 = (C % 2 == 0) && (ldx_s % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vec = v2 ? 2 : 1;
  int64_t lanes = cdiv(C, p.vec);
  p.l2 = 0;
  while ((1 << p.l2) < lanes && p.l2 < 8) ++p.l2;
  const int cpc = (1 << p.l2) * p.vec;
  p.chunks = (int)cdiv(C, cpc);
  p.cols_pad = (int64_t)p.chunks * cpc;
  int64_t want = cdiv(N, (int64_t)(PB_BLOCK >> p.l2) * 4);
  int64_t cap = (int64_t)num_cus() * 8 / (nrep > 8 ? 8 : nrep);
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  p.gx = (int)want;
  return p;
}

}  // namespace txm

using namespace txm;

extern "C" size_t txm_perturb_ws_bytes(int64_t N, int64_t C, int32_t n_alpha, int64_t nrep) {
  if (N < 1 || C < 1 || n_alpha < 1 || n_alpha > PB_MAXA || nrep < 1) return 0;
  int64_t cols_pad = 1;
  while (cols_pad < C) cols_pad <<= 1;
  if (cols_pad > 512) cols_pad = cdiv(C, 512) * 512;
  const size_t mm = 256 + (size_t)num_cus() * 8 * 2 * sizeof(double);
  int64_t gx_cap = (int64_t)num_cus() * 8 / (nrep > 8 ? 8 : nrep);  // as pb_plan()
  if (gx_cap < 1) gx_cap = 1;
  return align_up(mm, 256) + (size_t)nrep * gx_cap * cols_pad * n_alpha * 2 * sizeof(double) + 256;
}

extern "C" int txm_perturb(const double *x, int64_t ldx_s, const double *u, int64_t N, int64_t C,
                           const double *dalpha_host, int32_t n_alpha, const int64_t *freq, int64_t nrep,
                           double *out, void *ws, size_t ws_bytes, txm_stream stream) {
  TXM_REQUIRE(x && u && dalpha_host && out && ws, "perturb: null pointer");
  TXM_REQUIRE(N >= 1 && C >= 1 && C <= 65535 && ldx_s >= C, "perturb: bad N/C/ldx");
  TXM_REQUIRE(n_alpha >= 1 && n_alpha <= PB_MAXA, "perturb: n_alpha outside [1, %d]", PB_MAXA);
  TXM_REQUIRE(nrep >= 1 && nrep <= 65535 && (freq || nrep == 1), "perturb: freq == NULL requires nrep == 1");
  if (ws_bytes < txm_perturb_ws_bytes(N, C, n_alpha, nrep)) {
    set_error("perturb: workspace too small");
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double *mm = (double *)ws;
  double *mpart = mm + 2;
  const int mblk = num_cus() * 8;
  double *partial = (double *)((char *)ws + align_up(256 + (size_t)mblk * 2 * sizeof(double), 256));
  int gmm = (int)cdiv(N, PB_BLOCK * 8);
  if (gmm > mblk) gmm = mblk;
  hipLaunchKernelGGL(minmax_kernel, dim3(gmm), dim3(PB_BLOCK), 0, st, u, N, mpart);
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(1), 0, st, mpart, gmm, mm);
  TXM_LAUNCH_CHECK();
  const PbPlan p = pb_plan(x, ldx_s, N, C, nrep);
  PerturbArgs pa;
  for (int a = 0; a < PB_MAXA; ++a) pa.da[a] = a < n_alpha ? dalpha_host[a] : 0.0;
  dim3 grid(p.gx, p.chunks, (unsigned)nrep), block(PB_BLOCK);
  bool launched = false;
#define TXM_PB(NA_, VEC_, L2_)                                                                        \
  if (!launched && n_alpha == NA_ && p.vec == VEC_ && p.l2 == L2_) {                                  \
    if (freq)                                                                                         \
      hipLaunchKernelGGL((perturb_kernel<NA_, VEC_, L2_, true>), grid, block, 0, st, x, ldx_s, u, N, C, pa, \
                         mm, freq, partial);                                                          \
    else                                                                                              \
      hipLaunchKernelGGL((perturb_kernel<NA_, VEC_, L2_, false>), grid, block, 0, st, x, ldx_s, u, N, C, pa, \
                         mm, freq, partial);                                                          \
    launched = true;                                                                                  \
  }
#define TXM_PB_L(NA_, VEC_) \
  TXM_PB(NA_, VEC_, 0) TXM_PB(NA_, VEC_, 1) TXM_PB(NA_, VEC_, 2) TXM_PB(NA_, VEC_, 3) TXM_PB(NA_, VEC_, 4) \
  TXM_PB(NA_, VEC_, 5) TXM_PB(NA_, VEC_, 6) TXM_PB(NA_, VEC_, 7) TXM_PB(NA_, VEC_, 8)
#define TXM_PB_A(NA_) TXM_PB_L(NA_, 1) TXM_PB_L(NA_, 2)
  TXM_PB_A(1) TXM_PB_A(2) TXM_PB_A(3) TXM_PB_A(4) TXM_PB_A(5) TXM_PB_A(6) TXM_PB_A(7) TXM_PB_A(8)
#undef TXM_PB_A
#undef TXM_PB_L
#undef TXM_PB
  if (!launched) {
    set_error("perturb: no kernel variant");
    return TXM_ERR_UNSUPPORTED;
  }
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL(perturb_final_kernel, dim3((unsigned)C, (unsigned)nrep, (unsigned)n_alpha), dim3(PB_BLOCK), 0,
                     st, partial, p.gx, p.cols_pad, C, (int)n_alpha, out);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}
