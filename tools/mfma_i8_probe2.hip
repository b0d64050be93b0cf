// Probe 2: VALU instruction cost (cycles per wave instruction) alone and in the shadow of
// v_mfma_i32_32x32x32_i8, for the instruction mix of an int8 (Ozaki-sliced) bootstrap kernel:
// v_fma_f64, v_xor_b32, v_perm_b32, v_mad_u64_u32, v_bfe_u32, ds_add_u32 (random LDS address).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC, int NOPS, int MODE, bool MFMA>
__global__ __launch_bounds__(256) void k(int *out, int iters) {
  __shared__ unsigned lds[8192];
  v16i acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  v4i a = {0x01010101, 0x02020202, 0x01010101, 0x03030303}, b = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  unsigned rr[8], r1 = 12345u, r2 = 0x07020500u;
  unsigned long long mm[8];
  double dd[8], d1 = 1.0000001;
  for (int i = 0; i < 8; ++i) { rr[i] = threadIdx.x * 2654435761u + i; mm[i] = i; dd[i] = 1.0 + i; }
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MFMA) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < NOPS; ++q) {
        unsigned &r0 = rr[q & 7];
        unsigned long long &m = mm[q & 7];
        double &d0 = dd[q & 7];
        if (MODE == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r0) : "v"(r1));
        if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(d1));
        if (MODE == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r0) : "v"(r1), "v"(r2));
        if (MODE == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m) : "v"(r0), "v"(r1) : "vcc");
        if (MODE == 5) asm volatile("v_bfe_u32 %0, %0, 3, 10" : "+v"(r0));
        if (MODE == 6) { r0 = r0 * 1664525u + 1013904223u; atomicAdd(&lds[(r0 >> 8) & 8191u], 1u); }
        if (MODE == 7) { asm volatile("ds_add_u32 %0, %1" :: "v"((r0 & 8191u) << 2), "v"(r1) : "memory"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(r0) : "v"(r2)); }
        if (MODE == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r0) : "v"(r1));
        if (MODE == 9) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d0) : "v"(r1));
      }
    }
  }
  int r = (int)lds[threadIdx.x];
  for (int i = 0; i < 8; ++i) r += (int)rr[i] + (int)dd[i] + (int)mm[i];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int NOPS, int MODE, bool MFMA>
void run(int wps, const char *name) {
  int blocks = 256 * wps, iters = 4000;
  int *out; (void)hipMalloc(&out, sizeof(int) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC, NOPS, MODE, MFMA><<<blocks, 256>>>(out, 50);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC, NOPS, MODE, MFMA><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps);   // SIMD cycles per slot (1 MFMA + NOPS ops)
  printf("%-34s mfma=%d ops/slot=%2d waves/SIMD=%d %8.2f ms  %7.1f cyc/slot  %6.2f cyc/op(excess over 36)\n", name, (int)MFMA, NOPS, wps, ms,
         cyc, NOPS ? (cyc - (MFMA ? 36.0 : 0.0)) / NOPS : 0.0);
  (void)hipFree(out);
}

#define SWEEP(MODE, NAME)                                   \
  run<4, 16, MODE, false>(1, NAME " alone");                \
  run<4, 16, MODE, false>(2, NAME " alone");                \
  run<4, 4, MODE, true>(1, NAME);                           \
  run<4, 8, MODE, true>(1, NAME);                           \
  run<4, 16, MODE, true>(1, NAME);                          \
  run<4, 4, MODE, true>(2, NAME);                           \
  run<4, 8, MODE, true>(2, NAME);                           \
  run<4, 16, MODE, true>(2, NAME);

int main() {
  run<4, 0, 0, true>(1, "i8 MFMA only");
  run<4, 0, 0, true>(2, "i8 MFMA only");
  SWEEP(1, "v_xor_b32")
  SWEEP(2, "v_fma_f64")
  SWEEP(3, "v_perm_b32")
  SWEEP(4, "v_mad_u64_u32")
  SWEEP(5, "v_bfe_u32")
  SWEEP(7, "ds_add_u32+v_add")
  SWEEP(8, "v_mul_lo_u32")
  SWEEP(9, "v_cvt_f64_i32")
  return 0;
}
