// Probe: can the int8 bootstrap contraction stream a PRE-SLICED B operand from HBM/L2?
// 256 workgroups = 16 replicate groups x 16 chunks (same XCD-aware map as txm_resample_i8.hip);
// every workgroup of a chunk streams the same B data: [chunk][k-step][40 fragments][1 KiB].
// Per k-step and wave: 5 x global_load_dwordx4 (its fragments), 2 LDS reads (A), 10 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int NFR = 40, NPW = 5;

template <int PF, int MAP>
__global__ __launch_bounds__(512) void k(const unsigned char *__restrict__ B, int64_t steps_per_chunk, int n_rbg, int *out) {
  __shared__ __attribute__((aligned(16))) unsigned char cnt[64 * 1040];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  for (int e = threadIdx.x; e < 64 * 1040 / 4; e += 512) reinterpret_cast<unsigned *>(cnt)[e] = 0x01010101u;
  __syncthreads();
  const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
  int chunk = (q / n_rbg) * 8 + xcd;   // MAP 0: the 16 workgroups of a chunk on one XCD
  if (MAP == 1) chunk = 0;              // everyone reads the same chunk
  if (MAP == 2) chunk = b & 15;         // a chunk's workgroups spread over 2 XCDs, interleaved
  if (MAP == 3) chunk = b >> 4;         // consecutive blocks share a chunk (8 XCDs each)
  v16i acc[NPW][2];
  for (int e = 0; e < NPW; ++e) { acc[e][0] = (v16i)(0); acc[e][1] = (v16i)(0); }
  const unsigned char *base = B + (size_t)chunk * steps_per_chunk * NFR * 1024 + (size_t)wave * NPW * 1024 + lane * 16;
  const unsigned *cw = reinterpret_cast<const unsigned *>(cnt) + (lane & 31) * 260 + (lane >> 5) * 4;
  v4i Bn[PF][NPW];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int e = 0; e < NPW; ++e) Bn[p][e] = *reinterpret_cast<const v4i *>(base + (size_t)p * NFR * 1024 + e * 1024);
  for (int64_t s = 0; s < steps_per_chunk; s += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      v4i Bc[NPW];
#pragma unroll
      for (int e = 0; e < NPW; ++e) Bc[e] = Bn[p][e];
      const int64_t sn = s + p + PF < steps_per_chunk ? s + p + PF : s + p;
#pragma unroll
      for (int e = 0; e < NPW; ++e) Bn[p][e] = *reinterpret_cast<const v4i *>(base + (size_t)sn * NFR * 1024 + e * 1024);
      const int ss = (int)((s + p) & 31);
      const v4i A0 = *reinterpret_cast<const v4i *>(cw + ss * 8);
      const v4i A1 = *reinterpret_cast<const v4i *>(cw + ss * 8 + 32 * 260);
#pragma unroll
      for (int e = 0; e < NPW; ++e) {
        acc[e][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, Bc[e], acc[e][0], 0, 0, 0);
        acc[e][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, Bc[e], acc[e][1], 0, 0, 0);
      }
    }
  }
  int r = 0;
  for (int e = 0; e < NPW; ++e) for (int i = 0; i < 16; ++i) r += acc[e][0][i] + acc[e][1][i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int PF, int MAP> void run(const unsigned char *B, int64_t steps_per_chunk, int *out, const char *name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<PF, MAP><<<256, 512>>>(B, steps_per_chunk, 16, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<PF, MAP><<<256, 512>>>(B, steps_per_chunk, 16, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 16.0 * steps_per_chunk * NFR * 1024;
  printf("%-12s %8.2f ms  B = %.1f GB read by 16 rep-groups: %.2f TB/s L2-side, %.2f TB/s if HBM-once; %.0f cycles per k-step\n", name, ms,
         bytes / 1e9, 16 * bytes / ms / 1e9, bytes / ms / 1e9, ms * 1e-3 * 2.4e9 / steps_per_chunk);
}

int main(int argc, char **argv) {
  const int64_t N = argc > 1 ? (int64_t)atof(argv[1]) : 20000000;
  const int64_t steps_per_chunk = N / 32 / 16;
  const size_t bytes = (size_t)16 * steps_per_chunk * NFR * 1024;
  unsigned char *B; int *out;
  if (hipMalloc(&B, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(B, 1, bytes);
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<2, 0>(B, steps_per_chunk, out, "map xcd");
  run<2, 1>(B, steps_per_chunk, out, "map same");
  run<2, 2>(B, steps_per_chunk, out, "map b&15");
  run<2, 3>(B, steps_per_chunk, out, "map b>>4");
  return 0;
}
