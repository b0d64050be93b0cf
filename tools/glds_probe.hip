// Probe (round 5): global_load_lds_dwordx4 on gfx950 -- where does lane L's 16 bytes land (M0 base + 16 L?), does the
// destination reach LDS addresses above 64 KiB (M0 wider than 16 bits?), what does a half-masked wave do, and does
// ds_write_addtid_b32 reach above 64 KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const void *g, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void glds16_s(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 offset:0" ::"s"(sbase), "v"(voff), "s"(lds_dst) : "memory", "m0");
}

__global__ __launch_bounds__(64) void k(const uint32_t *src, uint32_t *out, int mode, uint32_t base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *l32 = reinterpret_cast<uint32_t *>(lds);
  const int lane = threadIdx.x;
  for (int i = lane; i < 40960; i += 64) l32[i] = 0xdeadbeefu;
  __syncthreads();
  if (mode == 0) glds16(src + 4 * lane, base);                       // per-lane 64-bit address
  if (mode == 1) { if (lane < 32) glds16(src + 4 * lane, base); }     // half wave
  if (mode == 2) glds16_s(src, (uint32_t)(16 * (63 - lane)), base);  // saddr + voffset, reversed lanes
  if (mode == 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" ::"v"(0x1000 + lane), "s"(base) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // report the first 4 dwords that changed and how many changed in total
  int changed = 0;
  for (int i = 0; i < 40960; ++i) if (l32[i] != 0xdeadbeefu) ++changed;
  if (lane == 0) {
    out[0] = changed;
    int n = 1;
    for (int i = 0; i < 40960 && n < 16; ++i) if (l32[i] != 0xdeadbeefu) { out[n++] = i * 4; out[n++] = l32[i]; i += 63; }
  }
}

int main() {
  std::vector<uint32_t> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 0xA0000000u + i;
  uint32_t *src, *out;
  (void)hipMalloc(&src, 1024); (void)hipMalloc(&out, 256);
  (void)hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int mode = 0; mode < 4; ++mode)
    for (uint32_t base : {0u, 0x8000u, 0xfff0u, 0x10000u, 0x18000u, 0x27000u}) {
      (void)hipMemset(out, 0, 256);
      k<<<1, 64, 163840>>>(src, out, mode, base);
      uint32_t r[16];
      (void)hipMemcpy(r, out, 64, hipMemcpyDeviceToHost);
      printf("mode %d base 0x%05x: %u dwords changed; first at byte 0x%05x = %08x, next 0x%05x = %08x, 0x%05x = %08x\n", mode, base, r[0],
             r[1], r[2], r[3], r[4], r[5], r[6]);
    }
  return 0;
}
