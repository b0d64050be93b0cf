// txm_pivot.h -- pivot estimation kernels shared by the reduce and resample
// translation units (static: one copy per TU, no relocatable device code).
#pragma once
#include "txm_common.h"

namespace txm {

constexpr int RED_BLOCK = 256;
constexpr int PIVOT_SAMPLES = 1024;

// ---------------------------------------------------------------------------
// pivot: block b = 0 -> u, b >= 1 -> column b-1.  pivot[b] = mean of a strided
// subsample.  Any value near the mean works; exactness is irrelevant.
static __global__ void pivot_kernel(const double *__restrict__ x, int64_t ldx_s, int64_t ldx_c,
                             const double *__restrict__ u, int64_t ldu_s, int64_t N,
                             double *__restrict__ pivot) {
  const int b = blockIdx.x;
  const int64_t ns = N < PIVOT_SAMPLES ? N : PIVOT_SAMPLES;
  const int64_t step = N / ns;
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < ns; k += blockDim.x) {
    const int64_t i = k * step;
    acc += (b == 0) ? u[i * ldu_s] : x[i * ldx_s + (int64_t)(b - 1) * ldx_c];
  }
  __shared__ double sh[RED_BLOCK];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = RED_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double p = sh[0] / (double)ns;
    // a non-finite pivot (inf/nan in the subsample) would poison every sum;
    // fall back to 0 and let the data speak for itself.
    if (!(p - p == 0.0)) p = 0.0;
    pivot[b] = p;
  }
}

// batched variant (S state points, blockIdx.y = state): pivot[s][1 + C]
static __global__ void pivot_batch_kernel(const txm_state_ptrs *__restrict__ batch, int64_t ldx_s, int64_t N,
                                          int64_t C, double *__restrict__ pivot) {
  const int b = blockIdx.x;
  const txm_state_ptrs bs = batch[blockIdx.y];
  const int64_t ns = N < PIVOT_SAMPLES ? N : PIVOT_SAMPLES;
  const int64_t step = N / ns;
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < ns; k += blockDim.x) {
    const int64_t i = k * step;
    acc += (b == 0) ? bs.u[i] : bs.x[i * ldx_s + (int64_t)(b - 1)];
  }
  __shared__ double sh[RED_BLOCK];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = RED_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double p = sh[0] / (double)ns;
    if (!(p - p == 0.0)) p = 0.0;
    pivot[(int64_t)blockIdx.y * (1 + C) + b] = p;
  }
}

// 1-D series variant: pivot[r] for row r of u2d.
static __global__ void pivot_rows_kernel(const double *__restrict__ u, int64_t ldu_r, int64_t N,
                                  double *__restrict__ pivot) {
  const int r = blockIdx.x;
  const int64_t ns = N < PIVOT_SAMPLES ? N : PIVOT_SAMPLES;
  const int64_t step = N / ns;
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < ns; k += blockDim.x) acc += u[(int64_t)r * ldu_r + k * step];
  __shared__ double sh[RED_BLOCK];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = RED_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double p = sh[0] / (double)ns;
    if (!(p - p == 0.0)) p = 0.0;
    pivot[r] = p;
  }
}


}  // namespace txm
