// Micro-benchmark 2: v_mfma_f64_16x16x4_f64 issue rate with distinct A/B registers
// per MFMA (as in the bootstrap kernel), A/B updated by VALU between groups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NA, int NB, bool UPD>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  v4f64 acc[NA][NB];
  double a[NA], b[NB];
  for (int i = 0; i < NA; ++i) { a[i] = a0 + i + threadIdx.x * 1e-6; for (int j = 0; j < NB; ++j) acc[i][j] = (v4f64){0, 0, 0, 0}; }
  for (int j = 0; j < NB; ++j) b[j] = b0 + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    if (UPD) {
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = a[i] * 1.0000001;
    }
  }
  double r = 0;
  for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NA, int NB, bool UPD>
void run(int wps, const char *name) {
  int blocks = 256 * wps, iters = 20000;
  double *out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NA, NB, UPD><<<blocks, 256>>>(out, 100, 1.0, 1e-3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NA, NB, UPD><<<blocks, 256>>>(out, iters, 1.0, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2048.0 * NA * NB * (double)iters * blocks * 4;
  printf("%-34s waves/SIMD=%d %8.2f ms %6.1f TF  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", name, wps, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)NA * NB * iters * wps));
  (void)hipFree(out);
}

int main() {
  run<5, 2, false>(1, "5x2 distinct A/B, static");
  run<5, 2, false>(2, "5x2 distinct A/B, static");
  run<5, 2, true>(1, "5x2 distinct A/B, A updated");
  run<5, 2, true>(2, "5x2 distinct A/B, A updated");
  run<10, 1, false>(2, "10x1");
  run<1, 10, false>(2, "1x10");
  run<5, 4, false>(1, "5x4 (20 acc)");
  run<5, 4, false>(2, "5x4 (20 acc)");
  return 0;
}
