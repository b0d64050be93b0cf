// Probe: semantics of ds_read_b64_tr_b8 on gfx950 (which LDS byte lands in which lane/byte), and
// whether v_mfma_i32_32x32x32_i8 fed from transposed reads contracts what we expect.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_sem(uint32_t *out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
  const int lane = threadIdx.x;
  // byte at offset o of every 256-byte page holds o; page p's bytes get p in a parallel array
  for (int i = lane; i < 4096; i += 64) lds[i] = (unsigned char)(i & 255);
  __syncthreads();
  const uint32_t lb = (uint32_t)(uintptr_t)lds;  // LDS byte offset (low half of the flat address); also makes the array escape
  uint32_t addr;
  if (mode == 0) addr = (uint32_t)(lane * 8);                                  // contiguous 8 B per lane
  else if (mode == 1) addr = (uint32_t)((lane & 15) * 8 + (lane >> 4) * 1024); // groups of 16 lanes on separate KiB
  else addr = (uint32_t)((lane & 7) * 16 + ((lane >> 3) & 1) * 8 + (lane >> 4) * 1024);
  addr += lb;
  v2i r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[lane * 2] = (uint32_t)r[0];
  out[lane * 2 + 1] = (uint32_t)r[1];
}

// MFMA check: A[m][k] = counts (bytes), B via tr reads of X table [k = sample][cl][8 bytes]; D[m][n]
__global__ void k_mfma(const signed char *Ag /*[32][32]*/, const signed char *Xg /*[32 samples][4 cl][8]*/, int *Dg /*[32][32]*/) {
  __shared__ __attribute__((aligned(16))) unsigned char xt[1024];
  const int lane = threadIdx.x, n32 = lane & 31, half = lane >> 5;
  for (int i = lane; i < 1024; i += 64) xt[i] = (unsigned char)Xg[i];
  __syncthreads();
  // A operand: lane (m = n32, half): bytes k = 16 half + 0..15
  v4i A;
  for (int q = 0; q < 4; ++q) {
    uint32_t w = 0;
    for (int e = 0; e < 4; ++e) w |= (uint32_t)(unsigned char)Ag[n32 * 32 + 16 * half + 4 * q + e] << (8 * e);
    A[q] = (int)w;
  }
  // B operand: two transposed reads: rows 16 half + 0..7 and + 8..15; lane in its 16-group supplies 8 bytes
  const int g = (lane >> 4) & 1, l16 = lane & 15;
  const uint32_t a0 = (uint32_t)(uintptr_t)xt + (uint32_t)((16 * half + (l16 >> 1)) * 32 + g * 16 + (l16 & 1) * 8);
  v2i b0, b1;
  asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(b0), "=&v"(b1) : "v"(a0) : "memory");
  v4i B = {b0[0], b0[1], b1[0], b1[1]};
  v16i acc = (v16i)(0);
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int m = 8 * (r / 4) + 4 * half + (r % 4);
    Dg[m * 32 + n32] = acc[r];
  }
}

int main() {
  uint32_t *d; (void)hipMalloc(&d, 128 * 4);
  uint32_t h[128];
  for (int mode = 0; mode < 3; ++mode) {
    k_sem<<<1, 64>>>(d, mode);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf(" lane %2d:", l);
      for (int b = 0; b < 8; ++b) printf(" %3u", (h[l * 2 + b / 4] >> (8 * (b % 4))) & 255u);
      printf("\n");
    }
  }
  // MFMA check
  std::vector<signed char> A(1024), X(1024);
  std::vector<int> D(1024), R(1024, 0);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)(s >> 24); };
  for (auto &v : A) v = (signed char)(rnd() & 15);
  for (auto &v : X) v = (signed char)(rnd() - 128);
  // expected: D[m][n] with n = cl * 8 + digit: sum_k A[m][k] * X[k][cl][digit]
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      int acc = 0;
      for (int k = 0; k < 32; ++k) acc += (int)A[m * 32 + k] * (int)X[k * 32 + (n >> 3) * 8 + (n & 7)];
      R[m * 32 + n] = acc;
    }
  signed char *dA, *dX; int *dD;
  (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dX, 1024); (void)hipMalloc(&dD, 4096);
  (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
  (void)hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dX, dD);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
  printf("mfma via tr8: %d mismatches of 1024\n", bad);
  if (bad) for (int i = 0; i < 8; ++i) printf("  D[0][%d] = %d expected %d\n", i, D[i], R[i]);
  return 0;
}
