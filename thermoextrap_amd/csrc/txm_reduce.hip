// txm_reduce.hip -- HBM-bound one-pass weighted central-comoment reduction
// (cmomy.wrap_reduce_vals as called from thermoextrap data.py:1632-1640,
// 1183-1203, 485-489, 528-532).
//
// Algorithm (NOT cmomy's per-sample Pebay push, which is ~300 flop per
// (sample, observable) and would make the pass ALU-bound): accumulate
// pivot-shifted power sums
//      S0[j]    = sum_i w_i (u_i - pu)^j
//      S1[c][j] = sum_i w_i (x_ic - px_c) (u_i - pu)^j          j = 0..order
// in registers (~K+1 FMA per 8-byte load), merge partials by plain addition
// (all workgroups share one pivot, estimated from a strided subsample by a
// tiny pre-kernel), and shift pivot-sums -> central moments once per column.
//
// Kernels
//   pivot_kernel          : pivot = mean of <=1024 strided samples per column
//   reduce_rowmajor_kernel: x is (rec, val) row-major; a wave reads whole rows
//                           with 16-B (or 8-B) loads per lane, fully coalesced;
//                           each lane owns VEC fixed columns -> register sums.
//   reduce_colmajor_kernel: each series contiguous along samples ((val, rec)
//                           layout, and the 1-D x_is_u path); lanes <-> samples.
//   finalize_*            : deterministic tree-sum of the per-workgroup
//                           partials + binomial shift to the cmomy layout.
#include "txm_common.h"
#include "txm_pivot.h"

namespace txm {

// Streaming loads of the sample matrix: every element is read exactly once, so the loads are marked non-temporal (no allocation in
// the caches; what IS re-read -- u, w, the pivots -- keeps its place).  Same box, N = 1e8, 32 observables, order 4: 4.46 - 4.55 ms
// (5.8 - 5.9 TB/s) -> 4.09 - 4.14 ms (6.4 - 6.5 TB/s = 0.80 of the 8 TB/s peak) -- profiles/r06_reduce_nt_ab.txt; -DTXM_RED_NO_NT is
// the A/B build; more rows in flight per lane (-DTXM_RED_UNR=8) or fewer (2) are slower (4.76 / 4.69 ms).  The 1-D reduction over
// (state, rec) series (BASELINE config 3): 0.222 -> 0.197 ms (5.8 -> 6.5 TB/s).
typedef double red_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 red_ld2(const double *p) {
#ifdef TXM_RED_NO_NT
  return *reinterpret_cast<const double2 *>(p);
#else
  const red_v2d t = __builtin_nontemporal_load(reinterpret_cast<const red_v2d *>(p));
  return make_double2(t.x, t.y);
#endif
}
// (the 8-byte path of odd or unaligned row pitches keeps plain loads: its rows share cache lines with their neighbours, which
// other wave instructions load -- non-temporal there re-fetches them: N = 1e8, 33 observables: 8.94 -> 9.23 ms)
__device__ __forceinline__ double red_ld1(const double *p) { return *p; }

// ---------------------------------------------------------------------------
// Row-major reduction.  Thread layout inside a 256-thread block:
//   lane_in_row = tid & (LPR-1)       owns columns  col0 + lane_in_row*VEC + {0..VEC-1}
//   row_in_blk  = tid >> LPR_LOG2     rows_per_blk = 256 / LPR
// blockIdx.y selects a chunk of LPR*VEC columns.  Rows are grid-strided with an
// unroll of UNR independent loads in flight per lane.
//
// partial layout (per (blockIdx.y, blockIdx.x)):  [LPR*VEC cols][2][K]
template <int K, int VEC, int LPR_LOG2, bool WEIGHTED>
__global__ __launch_bounds__(RED_BLOCK) void reduce_rowmajor_kernel(
    const double *__restrict__ x, int64_t ldx_s, const double *__restrict__ u,
    const double *__restrict__ w, int64_t N, int64_t C, const double *__restrict__ pivot,
    double *__restrict__ partial, const txm_state_ptrs *__restrict__ batch = nullptr) {
  constexpr int LPR = 1 << LPR_LOG2;
  if (batch != nullptr) {  // batched mode: blockIdx.z = state; per-state pivots and partial sums follow each other
    const txm_state_ptrs bs = batch[blockIdx.z];
    x = bs.x;
    u = bs.u;
    w = bs.w;
    pivot += (int64_t)blockIdx.z * (1 + C);
    partial += (size_t)blockIdx.z * gridDim.y * gridDim.x * (LPR * VEC) * 2 * K;
  }
  constexpr int ROWS = RED_BLOCK / LPR;
#ifndef TXM_RED_UNR  // (A/B builds: rows in flight per lane)
#define TXM_RED_UNR 4
#endif
  constexpr int UNR = TXM_RED_UNR;
  const int tid = threadIdx.x;
  const int lir = tid & (LPR - 1);
  const int rib = tid >> LPR_LOG2;
  const int64_t col0 = (int64_t)blockIdx.y * (LPR * VEC) + (int64_t)lir * VEC;
  const bool col_ok = col0 < C;  // VEC == 2 requires C even, so col0 + 1 < C too

  const double pu = pivot[0];
  double px[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) px[v] = col_ok ? pivot[1 + col0 + v] : 0.0;

  double s0[K], s1[VEC][K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    s0[j] = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) s1[v][j] = 0.0;
  }

  const int64_t stride = (int64_t)gridDim.x * ROWS;
  int64_t i = (int64_t)blockIdx.x * ROWS + rib;

  auto body = [&](double ui, double wi, const double (&xv)[VEC]) {
    const double du = ui - pu;
    double t = wi;  // w * du^j
#pragma unroll
    for (int j = 0; j < K; ++j) {
      s0[j] += t;
#pragma unroll
      for (int v = 0; v < VEC; ++v) s1[v][j] = fma(xv[v] - px[v], t, s1[v][j]);
      t *= du;
    }
  };

  if (col_ok) {
    // main loop, UNR rows in flight
    for (; i + (UNR - 1) * stride < N; i += UNR * stride) {
      double xv[UNR][VEC], ui[UNR], wi[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const int64_t r = i + q * stride;
        if constexpr (VEC == 2) {
          const double2 t2 = red_ld2(x + r * ldx_s + col0);
          xv[q][0] = t2.x;
          xv[q][1] = t2.y;
        } else {
          xv[q][0] = red_ld1(x + r * ldx_s + col0);
        }
        ui[q] = u[r];
        wi[q] = WEIGHTED ? w[r] : 1.0;
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) body(ui[q], wi[q], xv[q]);
    }
    for (; i < N; i += stride) {
      double xv[VEC];
      if constexpr (VEC == 2) {
        const double2 t2 = red_ld2(x + i * ldx_s + col0);
        xv[0] = t2.x;
        xv[1] = t2.y;
      } else {
        xv[0] = red_ld1(x + i * ldx_s + col0);
      }
      body(u[i], WEIGHTED ? w[i] : 1.0, xv);
    }
  }

  // ---- block reduction over the ROWS row-slots that share a column -------
  // LDS layout: [ROWS][LPR][NV] with NV = (1 + VEC) * K values per thread.
  constexpr int NV = (1 + VEC) * K;
  __shared__ double sh[RED_BLOCK * NV];
  double *mine = sh + (size_t)tid * NV;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    mine[j] = s0[j];
#pragma unroll
    for (int v = 0; v < VEC; ++v) mine[(1 + v) * K + j] = s1[v][j];
  }
  __syncthreads();
  // fixed-order (deterministic) sum over row slots, one thread per (lane, value)
  for (int e = tid; e < LPR * NV; e += RED_BLOCK) {
    const int l = e / NV, q = e % NV;
    double acc = 0.0;
#pragma unroll 4
    for (int r = 0; r < ROWS; ++r) acc += sh[((size_t)r * LPR + l) * NV + q];
    // q < K: u-row sums (identical for every lane of a row; keep per column)
    const int v = q / K - 1;  // -1 for the u-row
    const int j = q % K;
    double *dst = partial + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (LPR * VEC)) * 2 * K;
    if (v < 0) {
#pragma unroll
      for (int vv = 0; vv < VEC; ++vv) dst[((size_t)l * VEC + vv) * 2 * K + j] = acc;
    } else {
      dst[((size_t)l * VEC + v) * 2 * K + K + j] = acc;
    }
  }
}

// finalize: one block per column.  Sums partials over gridDim.x blocks in a
// fixed order, shifts to central moments, writes out[c][2][K].
template <int K>
__global__ __launch_bounds__(RED_BLOCK) void finalize_rowmajor_kernel(
    const double *__restrict__ partial, int nblk_x, int cols_per_chunk, int64_t C,
    const double *__restrict__ pivot, double *__restrict__ out, int n_colchunks = 0, int sums_only = 0) {
  const int64_t c = blockIdx.x;
  if (c >= C) return;
  // batched mode: blockIdx.y = state
  partial += (size_t)blockIdx.y * n_colchunks * nblk_x * cols_per_chunk * 2 * K;
  pivot += (int64_t)blockIdx.y * (1 + C);
  out += (size_t)blockIdx.y * C * 2 * K;
  const int chunk = (int)(c / cols_per_chunk), cc = (int)(c % cols_per_chunk);
  double acc[2 * K];
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) acc[q] = 0.0;
  for (int b = threadIdx.x; b < nblk_x; b += RED_BLOCK) {
    const double *src = partial + (((size_t)chunk * nblk_x + b) * cols_per_chunk + cc) * 2 * K;
#pragma unroll
    for (int q = 0; q < 2 * K; ++q) acc[q] += src[q];
  }
  __shared__ double sh[RED_BLOCK][2 * K];
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) sh[threadIdx.x][q] = acc[q];
  __syncthreads();
  for (int off = RED_BLOCK / 2; off > 0; off >>= 1) {
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
    if (sums_only) {  // the weight-scaled pivot power sums themselves (txm_reduce_vals_sums: sample-sharded reduce, push_vals)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        st[j] = S0[j];
        st[K + j] = S1[j];
      }
    } else {
      pivot_sums_to_state<K>(S0, S1, pivot[0], pivot[1 + c], st);
    }
#pragma unroll
    for (int q = 0; q < 2 * K; ++q) out[c * 2 * K + q] = st[q];
  }
}

// ---------------------------------------------------------------------------
// Column-major / 1-D reduction.  Series s (blockIdx.y) is contiguous along
// samples: xs = x + s*ld_series.  COV: also accumulate the x-row against u.
// For !COV the series itself is "u" and M = K moments are produced.
// partial: [series][gridDim.x][2 or 1][K]
template <int K, bool COV, bool WEIGHTED>
__global__ __launch_bounds__(RED_BLOCK) void reduce_colmajor_kernel(
    const double *__restrict__ x, int64_t ld_series, const double *__restrict__ u,
    const double *__restrict__ w, int64_t N, const double *__restrict__ pivot,
    double *__restrict__ partial) {
  const int s = blockIdx.y;
  const double *xs = x + (int64_t)s * ld_series;
  const double pu = COV ? pivot[0] : pivot[s];
  const double px = COV ? pivot[1 + s] : 0.0;
  double s0[K], s1[K];
#pragma unroll
  for (int j = 0; j < K; ++j) s0[j] = s1[j] = 0.0;

  auto body = [&](double xi, double ui, double wi) {
    const double du = (COV ? ui : xi) - pu;
    const double dx = xi - px;
    double t = wi;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      s0[j] += t;
      if (COV) s1[j] = fma(dx, t, s1[j]);
      t *= du;
    }
  };

  const int64_t nthreads = (int64_t)gridDim.x * RED_BLOCK;
  const int64_t gid = (int64_t)blockIdx.x * RED_BLOCK + threadIdx.x;
  // 16-byte loads when the series base is 16-B aligned
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(xs) & 15) == 0) &&
                      (!COV || (reinterpret_cast<uintptr_t>(u) & 15) == 0) &&
                      (!WEIGHTED || (reinterpret_cast<uintptr_t>(w) & 15) == 0);
  int64_t done = 0;
  if (vec_ok) {
    const int64_t npair = N / 2;
    constexpr int UNR = 4;
    int64_t p = gid;
    for (; p + (UNR - 1) * nthreads < npair; p += UNR * nthreads) {
      double2 xv[UNR], uv[UNR], wv[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const int64_t pp = p + q * nthreads;
        xv[q] = red_ld2(xs + 2 * pp);  // (the series itself: read once; u and w are shared by the series of a state and stay cached)
        if (COV) uv[q] = reinterpret_cast<const double2 *>(u)[pp];
        if (WEIGHTED) wv[q] = reinterpret_cast<const double2 *>(w)[pp];
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        body(xv[q].x, COV ? uv[q].x : 0.0, WEIGHTED ? wv[q].x : 1.0);
        body(xv[q].y, COV ? uv[q].y : 0.0, WEIGHTED ? wv[q].y : 1.0);
      }
    }
    for (; p < npair; p += nthreads) {
      const double2 xv = red_ld2(xs + 2 * p);
      double2 uv = {0, 0}, wv = {1, 1};
      if (COV) uv = reinterpret_cast<const double2 *>(u)[p];
      if (WEIGHTED) wv = reinterpret_cast<const double2 *>(w)[p];
      body(xv.x, uv.x, wv.x);
      body(xv.y, uv.y, wv.y);
    }
    done = npair * 2;
  }
  for (int64_t i = done + gid; i < N; i += nthreads)
    body(xs[i], COV ? u[i] : 0.0, WEIGHTED ? w[i] : 1.0);

  // block reduce (fixed order)
  constexpr int NV = COV ? 2 * K : K;
  __shared__ double sh[RED_BLOCK][NV];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    sh[threadIdx.x][j] = s0[j];
    if (COV) sh[threadIdx.x][K + j] = s1[j];
  }
  __syncthreads();
  for (int off = RED_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
#pragma unroll
      for (int q = 0; q < NV; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + off][q];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < NV)
    partial[((size_t)s * gridDim.x + blockIdx.x) * NV + threadIdx.x] = sh[0][threadIdx.x];
}

// finalize for colmajor partials: block per series.
template <int K, bool COV>
__global__ __launch_bounds__(RED_BLOCK) void finalize_colmajor_kernel(
    const double *__restrict__ partial, int nblk_x, const double *__restrict__ pivot,
    double *__restrict__ out, int sums_only = 0) {
  constexpr int NV = COV ? 2 * K : K;
  const int s = blockIdx.x;
  double acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = 0.0;
  for (int b = threadIdx.x; b < nblk_x; b += RED_BLOCK) {
    const double *src = partial + ((size_t)s * nblk_x + b) * NV;
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] += src[q];
  }
  __shared__ double sh[RED_BLOCK][NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) sh[threadIdx.x][q] = acc[q];
  __syncthreads();
  for (int off = RED_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
#pragma unroll
      for (int q = 0; q < NV; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + off][q];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (COV) {
      double S0[K], S1[K], st[2 * K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        S0[j] = sh[0][j];
        S1[j] = sh[0][K + j];
      }
      if (sums_only) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          st[j] = S0[j];
          st[K + j] = S1[j];
        }
      } else {
        pivot_sums_to_state<K>(S0, S1, pivot[0], pivot[1 + s], st);
      }
#pragma unroll
      for (int q = 0; q < 2 * K; ++q) out[(size_t)s * 2 * K + q] = st[q];
    } else {
      // 1-D: reuse the co-moment shift with a dummy x-row
      double S0[K], S1[K], st[2 * K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        S0[j] = sh[0][j];
        S1[j] = 0.0;
      }
      pivot_sums_to_state<K>(S0, S1, pivot[s], 0.0, st);
#pragma unroll
      for (int j = 0; j < K; ++j) out[(size_t)s * K + j] = st[j];
    }
  }
}

// ---------------------------------------------------------------------------
// host-side dispatch

static int grid_x_for(int64_t rows, int rows_per_block) {
  // memory-bound: ~8 blocks per CU, grid-stride the rest (guide Sec. 6 G11)
  int64_t want = cdiv(rows, (int64_t)rows_per_block * 4);  // >= 4 rows per slot
  int64_t cap = (int64_t)num_cus() * 8;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

struct RowPlan {
  int vec, lpr_log2, chunks, grid_x, cols_per_chunk;
};

static RowPlan plan_rowmajor(const double *x, int64_t ldx_s, int64_t N, int64_t C) {
  RowPlan p;
  const bool vec2 = (C % 2 == 0) && (ldx_s % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  p.vec = vec2 ? 2 : 1;
  int64_t lanes = cdiv(C, p.vec);
  int l2 = 0;
  while ((1 << l2) < lanes && l2 < 8) ++l2;
  p.lpr_log2 = l2;
  p.cols_per_chunk = (1 << l2) * p.vec;
  p.chunks = (int)cdiv(C, p.cols_per_chunk);
  p.grid_x = grid_x_for(N, RED_BLOCK >> l2);
  // keep total blocks bounded when there are many column chunks
  return p;
}

size_t reduce_vals_ws_bytes_impl(int64_t N, int64_t C, int order) {
  (void)order;
  const int K = TXM_MAXK;
  // worst case over both layouts
  size_t piv = align_up((size_t)(1 + C) * sizeof(double), 256);
  size_t rowm;
  // rowmajor: chunks * grid_x * cols_per_chunk * 2K  <= (C rounded up to pow2, x2) * grid cap
  int64_t cap = (int64_t)num_cus() * 8;
  int64_t cols_pad = 1;
  while (cols_pad < C) cols_pad <<= 1;
  if (cols_pad > 512) cols_pad = cdiv(C, 512) * 512;
  rowm = (size_t)cap * (size_t)cols_pad * 2 * K * sizeof(double);
  size_t colm = (size_t)C * (size_t)cap * 2 * K * sizeof(double);
  (void)N;
  return piv + (rowm > colm ? rowm : colm) + 256;
}

template <int K>
static int launch_rowmajor(const double *x, int64_t ldx_s, const double *u, const double *w,
                           int64_t N, int64_t C, const double *pivot, double *partial, double *out,
                           hipStream_t st, const txm_state_ptrs *batch = nullptr, int64_t S = 1,
                           bool aligned16 = true, int sums_only = 0) {
  // batched: `x` is only consulted for its alignment (aligned16 = every state's x is 16-byte aligned)
  RowPlan p = plan_rowmajor(aligned16 ? x : reinterpret_cast<const double *>(8), ldx_s, N, C);
  if (S > 1) {  // S states share the chip
    int gx = (int)cdiv(p.grid_x, S);
    p.grid_x = gx < 1 ? 1 : gx;
  }
  dim3 grid(p.grid_x, p.chunks, (unsigned)S), block(RED_BLOCK);
#define TXM_RM_CASE(VEC, L2)                                                                    \
  if (p.vec == VEC && p.lpr_log2 == L2) {                                                       \
    if (w)                                                                                      \
      hipLaunchKernelGGL((reduce_rowmajor_kernel<K, VEC, L2, true>), grid, block, 0, st, x,     \
                         ldx_s, u, w, N, C, pivot, partial, batch);                             \
    else                                                                                        \
      hipLaunchKernelGGL((reduce_rowmajor_kernel<K, VEC, L2, false>), grid, block, 0, st, x,    \
                         ldx_s, u, w, N, C, pivot, partial, batch);                             \
  } else
#define TXM_RM_VEC(VEC)                                                                         \
  TXM_RM_CASE(VEC, 0) TXM_RM_CASE(VEC, 1) TXM_RM_CASE(VEC, 2) TXM_RM_CASE(VEC, 3)               \
  TXM_RM_CASE(VEC, 4) TXM_RM_CASE(VEC, 5) TXM_RM_CASE(VEC, 6) TXM_RM_CASE(VEC, 7)               \
  TXM_RM_CASE(VEC, 8)
  TXM_RM_VEC(1) TXM_RM_VEC(2) {
    set_error("reduce_vals: no kernel for vec=%d lpr_log2=%d", p.vec, p.lpr_log2);
    return TXM_ERR_UNSUPPORTED;
  }
#undef TXM_RM_VEC
#undef TXM_RM_CASE
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL((finalize_rowmajor_kernel<K>), dim3((unsigned)C, (unsigned)S), block, 0, st, partial,
                     p.grid_x, p.cols_per_chunk, C, pivot, out, p.chunks, sums_only);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

template <int K>
static int launch_colmajor_cov(const double *x, int64_t ld_series, const double *u, const double *w,
                               int64_t N, int64_t C, const double *pivot, double *partial,
                               double *out, hipStream_t st, int sums_only = 0) {
  int gx = grid_x_for(N, RED_BLOCK * 2);
  // many series: fewer blocks per series is enough to fill the chip
  while (gx > 1 && (int64_t)gx * C > (int64_t)num_cus() * 16) gx = (gx + 1) / 2;
  dim3 grid(gx, (unsigned)C), block(RED_BLOCK);
  if (w)
    hipLaunchKernelGGL((reduce_colmajor_kernel<K, true, true>), grid, block, 0, st, x, ld_series, u,
                       w, N, pivot, partial);
  else
    hipLaunchKernelGGL((reduce_colmajor_kernel<K, true, false>), grid, block, 0, st, x, ld_series,
                       u, w, N, pivot, partial);
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL((finalize_colmajor_kernel<K, true>), dim3((unsigned)C), block, 0, st, partial,
                     gx, pivot, out, sums_only);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

template <int K>
static int launch_1d(const double *u, int64_t ldu_r, const double *w, int64_t N, int64_t R,
                     const double *pivot, double *partial, double *out, hipStream_t st) {
  int gx = grid_x_for(N, RED_BLOCK * 2);
  while (gx > 1 && (int64_t)gx * R > (int64_t)num_cus() * 16) gx = (gx + 1) / 2;
  dim3 grid(gx, (unsigned)R), block(RED_BLOCK);
  if (w)
    hipLaunchKernelGGL((reduce_colmajor_kernel<K, false, true>), grid, block, 0, st, u, ldu_r,
                       nullptr, w, N, pivot, partial);
  else
    hipLaunchKernelGGL((reduce_colmajor_kernel<K, false, false>), grid, block, 0, st, u, ldu_r,
                       nullptr, w, N, pivot, partial);
  TXM_LAUNCH_CHECK();
  hipLaunchKernelGGL((finalize_colmajor_kernel<K, false>), dim3((unsigned)R), block, 0, st, partial,
                     gx, pivot, out);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

}  // namespace txm

using namespace txm;

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

extern "C" size_t txm_reduce_vals_ws_bytes(int64_t N, int64_t C, int order) {
  if (N < 0 || C < 1) return 0;
  return reduce_vals_ws_bytes_impl(N, C, order);
}

// pivot_in == nullptr: the library's strided estimate; sums_only: out = the pivot power sums [C][2][K] (S0 | S1), not the state
static int reduce_vals_impl(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w, int64_t N,
                            int64_t C, int order, const double *pivot_in, int sums_only, double *out, void *ws,
                            size_t ws_bytes, hipStream_t st, const char *who) {
  TXM_REQUIRE(x && u && out && ws, "%s: null pointer", who);
  TXM_REQUIRE(N >= 1 && C >= 1, "%s: need N >= 1 and C >= 1 (N=%lld C=%lld)", who, (long long)N, (long long)C);
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "%s: order %d outside [0, %d]", who, order, TXM_MAX_ORDER);
  TXM_REQUIRE(ldx_c == 1 || ldx_s == 1, "%s: need ldx_c == 1 or ldx_s == 1", who);
  TXM_REQUIRE(C <= 65535, "%s: C > 65535 unsupported", who);
  if (ws_bytes < txm_reduce_vals_ws_bytes(N, C, order)) {
    set_error("%s: workspace too small", who);
    return TXM_ERR_WORKSPACE;
  }
  double *pivot = (double *)ws;
  double *partial = (double *)((char *)ws + align_up((size_t)(1 + C) * sizeof(double), 256));
  if (pivot_in != nullptr) {
    TXM_HIP(hipMemcpyAsync(pivot, pivot_in, sizeof(double) * (size_t)(1 + C), hipMemcpyDeviceToDevice, st));
  } else {
    hipLaunchKernelGGL(pivot_kernel, dim3((unsigned)(1 + C)), dim3(RED_BLOCK), 0, st, x, ldx_s, ldx_c, u, (int64_t)1, N, pivot);
    TXM_LAUNCH_CHECK();
  }
  const int K = order + 1;
  if (ldx_c == 1 && !(C == 1 && ldx_s == 1)) {
    TXM_REQUIRE(ldx_s >= C, "%s: row pitch ldx_s < C", who);
    TXM_K_SWITCH(K, return launch_rowmajor<KK>(x, ldx_s, u, w, N, C, pivot, partial, out, st, nullptr, 1, true, sums_only));
  } else {
    // (val, rec) layout, or a single contiguous series
    const int64_t ld_series = (C == 1) ? 0 : ldx_c;
    TXM_K_SWITCH(K, return launch_colmajor_cov<KK>(x, ld_series, u, w, N, C, pivot, partial, out, st, sums_only));
  }
  return TXM_OK;
}

extern "C" int txm_reduce_vals(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                               const double *w, int64_t N, int64_t C, int order, double *out,
                               void *ws, size_t ws_bytes, txm_stream stream) {
  return reduce_vals_impl(x, ldx_s, ldx_c, u, w, N, C, order, nullptr, 0, out, ws, ws_bytes, (hipStream_t)stream, "reduce_vals");
}

// ---- sample-sharded reduce / streaming accumulation (SURVEY 8(e) partition (4); cmomy push_vals) -------------------
// The reduce kernels accumulate weight-scaled power sums about a pivot; sums about ONE pivot add exactly (same algebra on
// every shard), so N samples split over ranks or over time are: a pivot everybody uses, per-shard sums, one fixed-order
// addition, one shift.
extern "C" int txm_reduce_vals_pivot(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, int64_t N, int64_t C,
                                     double *pivot, txm_stream stream) {
  TXM_REQUIRE(x && u && pivot, "reduce_vals_pivot: null pointer");
  TXM_REQUIRE(N >= 1 && C >= 1 && C <= 65535, "reduce_vals_pivot: need N >= 1 and 1 <= C <= 65535");
  TXM_REQUIRE(ldx_c == 1 || ldx_s == 1, "reduce_vals_pivot: need ldx_c == 1 or ldx_s == 1");
  hipLaunchKernelGGL(pivot_kernel, dim3((unsigned)(1 + C)), dim3(RED_BLOCK), 0, (hipStream_t)stream, x, ldx_s, ldx_c, u,
                     (int64_t)1, N, pivot);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" int txm_reduce_vals_sums(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w,
                                    int64_t N, int64_t C, int order, const double *pivot, double *sums, void *ws,
                                    size_t ws_bytes, txm_stream stream) {
  TXM_REQUIRE(pivot, "reduce_vals_sums: null pivot");
  return reduce_vals_impl(x, ldx_s, ldx_c, u, w, N, C, order, pivot, 1, sums, ws, ws_bytes, (hipStream_t)stream,
                          "reduce_vals_sums");
}

namespace txm {
// out[c] = shift( sums[0][c] + sums[1][c] + ... + sums[n - 1][c] ), added in that order; thread per column
template <int K>
__global__ __launch_bounds__(256) void sums_to_state_kernel(const double *__restrict__ sums, int64_t n, int64_t C,
                                                            const double *__restrict__ pivot, double *__restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double S0[K], S1[K], st[2 * K];
#pragma unroll
  for (int j = 0; j < K; ++j) S0[j] = S1[j] = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double *src = sums + (i * C + c) * 2 * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      S0[j] += src[j];
      S1[j] += src[K + j];
    }
  }
  pivot_sums_to_state<K>(S0, S1, pivot[0], pivot[1 + c], st);
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) out[c * 2 * K + q] = st[q];
}

// a cmomy state [2][K] as weight-scaled power sums about (pu, px) -- the inverse of pivot_sums_to_state
template <int K>
__device__ inline void state_to_pivot_sums(const double *st, double pu, double px, double *S0, double *S1) {
  const double W = st[0];
  const double du = ((K > 1) ? st[1] : 0.0) - pu, dx = st[K] - px;
#pragma unroll
  for (int b = 0; b < K; ++b) {
    double a0 = 0.0, a1 = 0.0, c = 1.0, p = 1.0;  // c = C(b, j), p = du^(b - j), j descending
    for (int j = b; j >= 0; --j) {
      const double m0 = (j == 0) ? 1.0 : (j == 1 ? 0.0 : st[j]);
      const double m1 = (j == 0) ? 0.0 : st[K + j];
      a0 += c * p * m0;
      a1 += c * p * (m1 + dx * m0);
      p *= du;
      c = c * (double)j / (double)(b - j + 1);
    }
    S0[b] = (W == 0.0) ? 0.0 : W * a0;
    S1[b] = (W == 0.0) ? 0.0 : W * a1;
  }
}

// state[c] <- shift( sums(state[c]) + chunk[c] ): the old state re-expressed about the chunk's pivot, added, shifted back
template <int K>
__global__ __launch_bounds__(256) void push_merge_kernel(double *__restrict__ state, const double *__restrict__ chunk,
                                                         int64_t C, const double *__restrict__ pivot) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double old[2 * K], S0[K], S1[K], st[2 * K];
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) old[q] = state[c * 2 * K + q];
  state_to_pivot_sums<K>(old, pivot[0], pivot[1 + c], S0, S1);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    S0[j] += chunk[c * 2 * K + j];
    S1[j] += chunk[c * 2 * K + K + j];
  }
  if (S0[0] == 0.0) return;  // nothing pushed into an empty accumulator: it stays empty (zeros)
  pivot_sums_to_state<K>(S0, S1, pivot[0], pivot[1 + c], st);
#pragma unroll
  for (int q = 0; q < 2 * K; ++q) state[c * 2 * K + q] = st[q];
}
}  // namespace txm

extern "C" int txm_sums_to_state(const double *sums, int64_t n, const double *pivot, int64_t C, int order, double *out,
                                 txm_stream stream) {
  TXM_REQUIRE(sums && pivot && out, "sums_to_state: null pointer");
  TXM_REQUIRE(n >= 1 && C >= 1, "sums_to_state: need n >= 1 and C >= 1");
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "sums_to_state: order %d outside [0, %d]", order, TXM_MAX_ORDER);
  TXM_K_SWITCH(order + 1, hipLaunchKernelGGL((sums_to_state_kernel<KK>), dim3((unsigned)cdiv(C, 256)), dim3(256), 0,
                                             (hipStream_t)stream, sums, n, C, pivot, out));
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" size_t txm_push_vals_ws_bytes(int64_t N, int64_t C, int order) {
  if (N < 1 || C < 1 || order < 0 || order > TXM_MAX_ORDER) return 0;
  return align_up((size_t)(1 + C) * sizeof(double), 256) + align_up((size_t)C * 2 * (order + 1) * sizeof(double), 256) +
         txm_reduce_vals_ws_bytes(N, C, order);
}

extern "C" int txm_push_vals(double *state, const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w,
                             int64_t N, int64_t C, int order, void *ws, size_t ws_bytes, txm_stream stream) {
  TXM_REQUIRE(state && x && u && ws, "push_vals: null pointer");
  TXM_REQUIRE(N >= 1 && C >= 1 && C <= 65535, "push_vals: need N >= 1 and 1 <= C <= 65535");
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "push_vals: order %d outside [0, %d]", order, TXM_MAX_ORDER);
  if (ws_bytes < txm_push_vals_ws_bytes(N, C, order)) {
    set_error("push_vals: workspace too small");
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double *pivot = (double *)ws;
  double *chunk = (double *)((char *)ws + align_up((size_t)(1 + C) * sizeof(double), 256));
  void *rws = (char *)chunk + align_up((size_t)C * 2 * (order + 1) * sizeof(double), 256);
  int rc = txm_reduce_vals_pivot(x, ldx_s, ldx_c, u, N, C, pivot, stream);
  if (rc != TXM_OK) return rc;
  rc = reduce_vals_impl(x, ldx_s, ldx_c, u, w, N, C, order, pivot, 1, chunk, rws, txm_reduce_vals_ws_bytes(N, C, order), st,
                        "push_vals");
  if (rc != TXM_OK) return rc;
  TXM_K_SWITCH(order + 1, hipLaunchKernelGGL((push_merge_kernel<KK>), dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, state,
                                             chunk, C, pivot));
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" size_t txm_reduce_vals_batched_ws_bytes(int64_t S, int64_t N, int64_t C, int order) {
  if (S < 1 || N < 0 || C < 1) return 0;
  // pointer table + per-state pivots + per-state partial sums (the per-state grid is never larger than one state's)
  return align_up((size_t)S * sizeof(txm_state_ptrs), 256) + align_up((size_t)S * (1 + C) * sizeof(double), 256) +
         (size_t)S * reduce_vals_ws_bytes_impl(N, C, order);
}

extern "C" int txm_reduce_vals_batched(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N,
                                       int64_t C, int order, double *out, void *ws, size_t ws_bytes,
                                       txm_stream stream) {
  TXM_REQUIRE(states_host && out && ws, "reduce_vals_batched: null pointer");
  TXM_REQUIRE(S >= 1 && S <= 65535 && N >= 1 && C >= 1 && C <= 65535, "reduce_vals_batched: bad S/N/C");
  TXM_REQUIRE(order >= 0 && order <= TXM_MAX_ORDER, "reduce_vals_batched: order %d outside [0, %d]", order, TXM_MAX_ORDER);
  TXM_REQUIRE(ldx_s >= C, "reduce_vals_batched: row pitch ldx_s < C");
  const bool weighted = states_host[0].w != nullptr;
  bool aligned16 = true;
  for (int64_t s = 0; s < S; ++s) {
    TXM_REQUIRE(states_host[s].x && states_host[s].u, "reduce_vals_batched: state %lld has a null pointer", (long long)s);
    TXM_REQUIRE((states_host[s].w != nullptr) == weighted, "reduce_vals_batched: weights for all states or for none");
    aligned16 = aligned16 && (reinterpret_cast<uintptr_t>(states_host[s].x) & 15) == 0;
  }
  if (ws_bytes < txm_reduce_vals_batched_ws_bytes(S, N, C, order)) {
    set_error("reduce_vals_batched: workspace too small");
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  txm_state_ptrs *tab = (txm_state_ptrs *)ws;
  TXM_HIP(hipMemcpyAsync(tab, states_host, (size_t)S * sizeof(txm_state_ptrs), hipMemcpyHostToDevice, st));
  double *pivot = (double *)((char *)ws + align_up((size_t)S * sizeof(txm_state_ptrs), 256));
  double *partial = (double *)((char *)pivot + align_up((size_t)S * (1 + C) * sizeof(double), 256));
  hipLaunchKernelGGL(pivot_batch_kernel, dim3((unsigned)(1 + C), (unsigned)S), dim3(RED_BLOCK), 0, st, tab, ldx_s, N, C, pivot);
  TXM_LAUNCH_CHECK();
  const int K = order + 1;
  const double *w0 = weighted ? states_host[0].w : nullptr;  // only its null-ness selects the kernel
  TXM_K_SWITCH(K, return launch_rowmajor<KK>(states_host[0].x, ldx_s, states_host[0].u, w0, N, C, pivot, partial, out,
                                             st, tab, S, aligned16));
  return TXM_OK;
}

extern "C" size_t txm_reduce_vals_1d_ws_bytes(int64_t N, int64_t R, int M) {
  (void)N; (void)M;
  if (R < 1) return 0;
  return align_up((size_t)R * sizeof(double), 256) +
         (size_t)R * (size_t)num_cus() * 8 * TXM_MAXK * sizeof(double) + 256;
}

extern "C" int txm_reduce_vals_1d(const double *u, int64_t ldu_r, int64_t ldu_s, const double *w,
                                  int64_t N, int64_t R, int M, double *out, void *ws,
                                  size_t ws_bytes, txm_stream stream) {
  TXM_REQUIRE(u && out && ws, "reduce_vals_1d: null pointer");
  TXM_REQUIRE(N >= 1 && R >= 1 && R <= 65535, "reduce_vals_1d: bad N/R");
  TXM_REQUIRE(M >= 1 && M <= TXM_MAXK, "reduce_vals_1d: M=%d outside [1, %d]", M, TXM_MAXK);
  TXM_REQUIRE(ldu_s == 1, "reduce_vals_1d: series must be contiguous (ldu_s == 1)");
  if (ws_bytes < txm_reduce_vals_1d_ws_bytes(N, R, M)) {
    set_error("reduce_vals_1d: workspace too small");
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double *pivot = (double *)ws;
  double *partial = (double *)((char *)ws + align_up((size_t)R * sizeof(double), 256));
  hipLaunchKernelGGL(pivot_rows_kernel, dim3((unsigned)R), dim3(RED_BLOCK), 0, st, u, ldu_r, N,
                     pivot);
  TXM_LAUNCH_CHECK();
  TXM_K_SWITCH(M, return launch_1d<KK>(u, ldu_r, w, N, R, pivot, partial, out, st));
  return TXM_OK;
}
