// Micro-benchmark: practical ceiling of v_mfma_f64_16x16x4_f64 on gfx950, alone
// and with double-precision VALU work interleaved (what the bootstrap kernel does).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0, s = 0.0, t = 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (i < NVALU) { t = t * a + b; s += t; }  // 2 DP VALU ops per slot
    }
  }
  double r = s;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int NVALU>
void run(int waves_per_simd, const char *name) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
  int iters = 20000;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, NVALU><<<blocks, 256>>>(out, 100, 1.0, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, NVALU><<<blocks, 256>>>(out, iters, 1.0, 1e-3);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2048.0 * NACC * (double)iters * blocks * 4;
  printf("%-28s waves/SIMD=%d  %8.2f ms  %6.1f TF (mfma only)\n", name, waves_per_simd, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 8; w *= 2) run<10, 0>(w, "10 acc, no VALU");
  for (int w = 1; w <= 8; w *= 2) run<10, 7>(w, "10 acc + 14 DP VALU");
  run<4, 0>(4, "4 acc, no VALU");
  run<2, 0>(8, "2 acc, no VALU");
  return 0;
}
