// Probe 3 (round 5, verdict item 8): does v_mfma_i32_16x16x64_i8 hide more vector work than v_mfma_i32_32x32x32_i8?
// slot = one MFMA + n independent vector instructions of one wave; two 16x16x64 do the work of one 32x32x32.
// Prints SIMD cycles per slot and per 65536 int8 operations for n = 0 .. 16, one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC, int NOPS, int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters) {
  v16i acc32[SHAPE == 32 ? NACC : 1];
  v4i acc16[SHAPE == 16 ? NACC : 1];
  for (auto &a : acc32) for (int j = 0; j < 16; ++j) a[j] = 0;
  for (auto &a : acc16) for (int j = 0; j < 4; ++j) a[j] = 0;
  v4i a = {0x01010101, 0x02020202, 0x01010101, 0x03030303}, b = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  unsigned rr[8], r1 = 12345u;
  double dd[8], d1 = 1.0000001;
  for (int i = 0; i < 8; ++i) { rr[i] = threadIdx.x * 2654435761u + i; dd[i] = 1.0 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (SHAPE == 32) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc32[SHAPE == 32 ? i : 0]) : "v"(a), "v"(b));
      if (SHAPE == 16) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc16[SHAPE == 16 ? i : 0]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NOPS; ++q) {
        if (MODE == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(rr[q & 7]) : "v"(r1));
        if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[q & 7]) : "v"(d1));
      }
    }
  }
  int r = 0;
  for (int i = 0; i < 8; ++i) r += (int)rr[i] + (int)dd[i];
  for (auto &x : acc32) for (int j = 0; j < 16; ++j) r += x[j];
  for (auto &x : acc16) for (int j = 0; j < 4; ++j) r += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int SHAPE, int NOPS, int MODE>
void run(int wps, const char *name) {
  constexpr int NACC = 8;
  int blocks = 256 * wps, iters = 2000;
  int *out; (void)hipMalloc(&out, sizeof(int) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<SHAPE, NACC, NOPS, MODE><<<blocks, 256>>>(out, 50);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<SHAPE, NACC, NOPS, MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps);  // SIMD cycles per slot at a nominal 2.4 GHz
  const double per64k = cyc * (SHAPE == 32 ? 1.0 : 2.0);
  printf("%-10s shape=%dx  n=%2d waves/SIMD=%d %8.2f ms %7.1f cyc/slot %7.1f cyc per 65536 ops (with %d vector ops)\n", name, SHAPE, NOPS, wps,
         ms, cyc, per64k, SHAPE == 32 ? NOPS : 2 * NOPS);
  (void)hipFree(out);
}

#define SWEEP(SHAPE, MODE, NAME) \
  run<SHAPE, 0, MODE>(1, NAME); run<SHAPE, 2, MODE>(1, NAME); run<SHAPE, 4, MODE>(1, NAME); run<SHAPE, 8, MODE>(1, NAME); run<SHAPE, 16, MODE>(1, NAME); \
  run<SHAPE, 0, MODE>(2, NAME); run<SHAPE, 2, MODE>(2, NAME); run<SHAPE, 4, MODE>(2, NAME); run<SHAPE, 8, MODE>(2, NAME); run<SHAPE, 16, MODE>(2, NAME);

int main() {
  SWEEP(32, 1, "v_xor_b32")
  SWEEP(16, 1, "v_xor_b32")
  SWEEP(32, 2, "v_fma_f64")
  SWEEP(16, 2, "v_fma_f64")
  return 0;
}
