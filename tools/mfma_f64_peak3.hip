// Micro-benchmark 3: issue rate of v_mfma_f64_16x16x4_f64 with inline asm
// (accumulators pinned in VGPRs, no compiler-inserted AGPR shuffling).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-6, b = b0, t = 1.0, s = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (i < NVALU) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_add_f64 %1, %1, %0" : "+v"(t), "+v"(s) : "v"(a), "v"(b));
    }
  }
  double r = s;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int NVALU>
void run(int wps, const char *name) {
  int blocks = 256 * wps, iters = 20000;
  double *out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, NVALU><<<blocks, 256>>>(out, 100, 1.0, 1e-3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC, NVALU><<<blocks, 256>>>(out, iters, 1.0, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2048.0 * NACC * (double)iters * blocks * 4;
  printf("%-26s waves/SIMD=%d %8.2f ms %6.1f TF  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", name, wps, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps));
  (void)hipFree(out);
}

int main() {
  run<10, 0>(1, "10 acc");
  run<10, 0>(2, "10 acc");
  run<10, 0>(4, "10 acc");
  run<4, 0>(1, "4 acc");
  run<2, 0>(1, "2 acc");
  run<1, 0>(1, "1 acc (dependent chain)");
  run<10, 7>(1, "10 acc + 14 DP VALU");
  run<10, 7>(2, "10 acc + 14 DP VALU");
  return 0;
}
