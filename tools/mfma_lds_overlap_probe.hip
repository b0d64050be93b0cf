// Probe (round 5): do int8 MFMAs and LDS traffic overlap on a CU at two waves per SIMD?  The count-table kernel's time is its
// LDS time PLUS its MFMA time (profiles/r05_experiments.md); is that the chip or the kernel?
// Per iteration and wave: NM independent v_mfma_i32_32x32x32_i8 (own accumulators), NR ds_read_b128 + NW ds_write_b32 whose
// results nobody waits for until the end of the iteration (one s_waitcnt lgkmcnt(0) per iteration), either GROUPED (all MFMAs,
// then all LDS ops -- what the compiler made of the kernel) or INTERLEAVED (an LDS op after every MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NM, int NR, int NW, int MODE>  // MODE 0: MFMA only, 1: LDS only, 2: grouped, 3: interleaved
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(int *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  v16i acc[NM];
  for (int i = 0; i < NM; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  v4i a = {0x01010101, 0x02020202, 0x01010101, 0x03030303}, b = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned rbase = wave * 8192 + lane * 16, wbase = 65536 + wave * 4096 + lane * 4;
  v4i r[4] = {a, a, a, a};
  for (int i = threadIdx.x; i < 40000; i += 512) ((int *)lds)[i] = i;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    constexpr int NL = NR + NW;
    int li = 0;
#pragma unroll
    for (int m = 0; m < (MODE == 1 ? 0 : NM); ++m) {
      asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
      if (MODE == 3) {
#pragma unroll
        for (int q = 0; q < (NL + NM - 1) / NM; ++q)
          if (li < NL) {
            if (li < NR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[li & 3]) : "v"(rbase), "n"((li % 4) * 1024) : "memory");
            else asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(wbase), "v"(lane), "n"(((li - NR) % 8) * 256) : "memory");
            ++li;
          }
      }
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        if (q < NR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[q & 3]) : "v"(rbase), "n"((q % 4) * 1024) : "memory");
        else asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(wbase), "v"(lane), "n"(((q - NR) % 8) * 256) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a[0] ^= r[0][0] & 0;  // the reads feed the next iteration's operand (value unchanged)
  }
  int s = 0;
  for (int i = 0; i < NM; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 4; ++i) s += r[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NR, int NW, int MODE>
double run(const char *name) {
  int *out; (void)hipMalloc(&out, 4 * 256 * 512);
  (void)hipFuncSetAttribute((const void *)k<NM, NR, NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  const int iters = 20000;
  k<NM, NR, NW, MODE><<<256, 512, 160000>>>(out, 100);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<NM, NR, NW, MODE><<<256, 512, 160000>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us_it = ms * 1e3 / iters;
  printf("%-40s NM=%2d NR=%2d NW=%2d  %8.3f ms  %7.3f us per iteration (8 waves per CU, 2 per SIMD)\n", name, NM, NR, NW, ms, us_it);
  (void)hipFree(out);
  return us_it;
}

int main() {
  // the table kernel's k-step per wave: 12 MFMAs; 4 b128 count reads + 6 operand reads + 7 factor / x reads ~ 12 reads of 1 KiB each, 12 stores
  run<12, 12, 12, 0>("MFMA only");
  run<12, 12, 12, 1>("LDS only (12 b128 reads + 12 b32 writes)");
  run<12, 12, 12, 2>("grouped: 12 MFMA then 24 LDS ops");
  run<12, 12, 12, 3>("interleaved: 2 LDS ops behind every MFMA");
  run<12, 24, 12, 1>("LDS only (24 reads + 12 writes)");
  run<12, 24, 12, 2>("grouped");
  run<12, 24, 12, 3>("interleaved");
  run<12, 6, 6, 1>("LDS only (6 reads + 6 writes)");
  run<12, 6, 6, 2>("grouped");
  run<12, 6, 6, 3>("interleaved");
  return 0;
}
