// Probe 4 (round 5): accumulators in AccVGPRs against accumulators in VGPRs (-amdgpu-mfma-vgpr-form, what both int8 kernels
// use).  slot = one v_mfma_i32_32x32x32_i8 + n vector instructions of the same wave.  If the VGPR form's C/D traffic (16
// registers in, 16 out per MFMA) takes the vector register ports, vector instructions beside it cannot overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <bool AGPR, int NOPS, int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters) {
  constexpr int NACC = 8;
  v16i acc[NACC];
  for (auto &a : acc) for (int j = 0; j < 16; ++j) a[j] = 0;
  v4i a = {0x01010101, 0x02020202, 0x01010101, 0x03030303}, b = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  unsigned rr[8], r1 = 12345u;
  double dd[8], d1 = 1.0000001;
  for (int i = 0; i < 8; ++i) { rr[i] = threadIdx.x * 2654435761u + i; dd[i] = 1.0 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (AGPR) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NOPS; ++q) {
        if (MODE == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(rr[q & 7]) : "v"(r1));
        if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[q & 7]) : "v"(d1));
        if (MODE == 3) {  // a dependent chain, as the slicing: fma -> xor -> xor
          if (q % 3 == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[(q / 3) & 7]) : "v"(d1));
          else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(rr[q & 7]) : "v"(r1), "v"(dd[(q / 3) & 7]));
        }
      }
    }
  }
  int r = 0;
  for (int i = 0; i < 8; ++i) r += (int)rr[i] + (int)dd[i];
  for (auto &x : acc) for (int j = 0; j < 16; ++j) r += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool AGPR, int NOPS, int MODE>
void run(int wps, const char *name) {
  int blocks = 256 * wps, iters = 2000;
  int *out; (void)hipMalloc(&out, sizeof(int) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<AGPR, NOPS, MODE><<<blocks, 256>>>(out, 50);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<AGPR, NOPS, MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * iters * wps);
  printf("%-10s acc in %s  n=%2d waves/SIMD=%d %8.2f ms %7.1f cycles per slot\n", name, AGPR ? "AccVGPRs" : "VGPRs   ", NOPS, wps, ms, cyc);
  (void)hipFree(out);
}
#define SWEEP(AG, MODE, NAME) \
  run<AG, 0, MODE>(1, NAME); run<AG, 2, MODE>(1, NAME); run<AG, 4, MODE>(1, NAME); run<AG, 6, MODE>(1, NAME); run<AG, 8, MODE>(1, NAME); run<AG, 12, MODE>(1, NAME); \
  run<AG, 0, MODE>(2, NAME); run<AG, 4, MODE>(2, NAME); run<AG, 6, MODE>(2, NAME); run<AG, 8, MODE>(2, NAME); run<AG, 12, MODE>(2, NAME);
int main() {
  SWEEP(false, 1, "v_xor_b32") SWEEP(true, 1, "v_xor_b32")
  SWEEP(false, 2, "v_fma_f64") SWEEP(true, 2, "v_fma_f64")
  SWEEP(false, 3, "chain") SWEEP(true, 3, "chain")
  return 0;
}
