// Micro-benchmark 4: do non-DP VALU instructions (int / f32) steal FP64-MFMA throughput on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// MODE 0: none, 1: v_xor_b32, 2: v_mul_lo_u32, 3: v_fma_f32, 4: v_fma_f64
template <int NACC, int NOPS, int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-6, b = b0;
  unsigned r0 = threadIdx.x, r1 = 12345u;
  float f0 = 1.0f, f1 = 0.999f;
  double d0 = 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NOPS; ++q) {
        if (MODE == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r0) : "v"(r1));
        if (MODE == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r0) : "v"(r1));
        if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));
        if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(a));
      }
    }
  }
  double r = (double)r0 + f0 + d0;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int NOPS, int MODE>
void run(int wps, const char *name) {
  int blocks = 256 * wps, iters = 10000;
  double *out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, NOPS, MODE><<<blocks, 256>>>(out, 100, 1.0, 1e-3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC, NOPS, MODE><<<blocks, 256>>>(out, iters, 1.0, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps);
  printf("%-34s waves/SIMD=%d %7.2f ms  %.1f cyc per MFMA slot  => %.2f cyc per extra op\n", name, wps, ms, cyc,
         NOPS ? (cyc - 64.8) / NOPS : 0.0);
  (void)hipFree(out);
}

int main() {
  run<10, 0, 0>(2, "MFMA only");
  run<10, 4, 1>(1, "+4 v_xor_b32 per MFMA");
  run<10, 4, 1>(2, "+4 v_xor_b32 per MFMA");
  run<10, 8, 1>(2, "+8 v_xor_b32 per MFMA");
  run<10, 4, 2>(2, "+4 v_mul_lo_u32 per MFMA");
  run<10, 4, 3>(2, "+4 v_fma_f32 per MFMA");
  run<10, 2, 4>(2, "+2 v_fma_f64 per MFMA");
  run<10, 4, 4>(2, "+4 v_fma_f64 per MFMA");
  return 0;
}
