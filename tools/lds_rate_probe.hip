// LDS instruction throughput per CU on gfx950: cycles per wave64 instruction for ds_write_b16/b32/b64/b128
// and ds_read_b32/b128 with conflict-free addresses, 8 waves per CU (512-thread blocks, 1 block per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(int *out, int iters) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned v0 = threadIdx.x, v1 = 2, v2 = 3, v3 = 4;
  unsigned acc = 0;
  // per-wave 8 KiB region, lane-contiguous addresses
  const unsigned base = wave * 8192;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (MODE == 0) asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(base + lane * 2), "v"(v0), "n"(q * 128) : "memory");
      if (MODE == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base + lane * 4), "v"(v0), "n"(q * 256) : "memory");
      if (MODE == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(base + lane * 8), "v"((unsigned long long)v0), "n"(q * 512) : "memory");
      if (MODE == 3) { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 d = {v0, v1, v2, v3};
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base + lane * 16 + (q & 7) * 1024), "v"(d), "n"(0) : "memory"); }
      if (MODE == 4) { unsigned r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(base + lane * 4), "n"(q * 256) : "memory"); acc += r; }
      if (MODE == 5) { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(base + lane * 16 + (q & 7) * 1024), "n"(0) : "memory"); acc += r[0]; }
      if (MODE == 6) asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(base + (lane & 15) * 2 + (lane >> 4) * 32), "v"(v0), "n"(q * 128) : "memory");
      if (MODE == 7) { typedef unsigned u2 __attribute__((ext_vector_type(2))); u2 r;  // transposing read, 8-bit: 128 contiguous bytes per 16-lane group
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r) : "v"(base + lane * 8), "n"(q * 512) : "memory"); acc += r[0]; }
      if (MODE == 8) { typedef unsigned u2 __attribute__((ext_vector_type(2))); u2 r;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(base + lane * 8), "n"(q * 512) : "memory"); acc += r[0]; }
      if (MODE == 9) { typedef unsigned u2 __attribute__((ext_vector_type(2))); u2 r;  // the i8t kernel's B-operand address pattern
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r) : "v"(base + (16 * (lane >> 5) + ((lane & 15) >> 1)) * 32 + ((lane >> 4) & 1) * 16 + (lane & 1) * 8), "n"((q & 7) * 1024) : "memory"); acc += r[0]; }
      if (MODE == 10) { typedef unsigned u2 __attribute__((ext_vector_type(2))); u2 r;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(base + lane * 8), "n"(q * 512) : "memory"); acc += r[0]; }
      if (MODE == 11) { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 d = {v0, v1, v2, v3};  // the i8t kernel's X-table store pattern
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base + (lane >> 3) * 1024 + ((lane & 7) >> 1) * 32 + (lane & 1) * 16), "v"(d), "n"((q & 7) * 128) : "memory"); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc + lds[threadIdx.x];
}
template <int MODE> void run(const char *name) {
  int *out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256, 512, 65536>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<256, 512, 65536>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * 8);  // CU cycles per wave instruction
  printf("%-28s %8.3f ms  %6.2f CU-cycles per wave64 instruction\n", name, ms, cyc);
  (void)hipFree(out);
}
int main() {
  run<0>("ds_write_b16 (contiguous)");
  run<6>("ds_write_b16 (16 x 4 rows)");
  run<1>("ds_write_b32");
  run<2>("ds_write_b64");
  run<3>("ds_write_b128");
  run<4>("ds_read_b32");
  run<5>("ds_read_b128");
  run<8>("ds_read_b64");
  run<10>("ds_read_b64_tr_b16");
  run<7>("ds_read_b64_tr_b8 (contiguous)");
  run<9>("ds_read_b64_tr_b8 (i8t B operand)");
  run<11>("ds_write_b128 (i8t X table)");
  return 0;
}
