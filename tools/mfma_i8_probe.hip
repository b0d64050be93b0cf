// Probe: v_mfma_i32_32x32x32_i8 issue rate on gfx950 and whether VALU work co-executes
// with it (it does not with the FP64 MFMA, see mfma_f64_peak4.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC, int NOPS, int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters) {
  v16i acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  v4i a = {0x01010101, 0x02020202, 0x01010101, 0x03030303}, b = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  unsigned r0 = threadIdx.x, r1 = 12345u;
  double d0 = 1.0, d1 = 1.0000001;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NOPS; ++q) {
        if (MODE == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r0) : "v"(r1));
        if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(d1));
      }
    }
  }
  int r = (int)r0 + (int)d0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int NOPS, int MODE>
void run(int wps, const char *name) {
  int blocks = 256 * wps, iters = 20000;
  int *out; (void)hipMalloc(&out, sizeof(int) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, NOPS, MODE><<<blocks, 256>>>(out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC, NOPS, MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double macs = 32768.0 * NACC * (double)iters * blocks * 4;
  double cyc = ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps);
  printf("%-30s waves/SIMD=%d %7.2f ms  %7.1f TOP/s  %.1f cyc per MFMA slot\n", name, wps, ms, 2 * macs / ms / 1e9, cyc);
  (void)hipFree(out);
}

int main() {
  run<4, 0, 0>(1, "i8 MFMA only");
  run<4, 0, 0>(2, "i8 MFMA only");
  run<4, 4, 1>(1, "+4 v_xor per MFMA");
  run<4, 4, 1>(2, "+4 v_xor per MFMA");
  run<4, 8, 1>(2, "+8 v_xor per MFMA");
  run<4, 4, 2>(2, "+4 v_fma_f64 per MFMA");
  run<4, 8, 2>(2, "+8 v_fma_f64 per MFMA");
  return 0;
}
