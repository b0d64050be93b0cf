// Probe (round 5): does a wave's DS read wait for an LDS-DMA (global_load_lds) the same wave issued just before?
// Cycles (s_memtime) from after the DMA issue to the completion of: (0) nothing but `s_waitcnt lgkmcnt(0)`, (1) a ds_read_b32 of an
// unrelated LDS address + lgkmcnt(0), (2) the same with NO DMA before, (3) s_waitcnt vmcnt(0) (the DMA's own landing time),
// (4) ds_write_b32 + lgkmcnt(0) after the DMA, (5) a ds_read issued by ANOTHER wave of the workgroup while wave 0 has a DMA in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16(const void *g, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(128) void k(const uint32_t *src, long long *out, int mode, int stride) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long acc = 0;
  uint32_t sink = 0;
  for (int it = 0; it < 64; ++it) {
    const uint32_t *g = src + ((size_t)(blockIdx.x * 64 + it) * stride + lane) * 4;  // fresh lines: HBM misses
    __syncthreads();
    long long t0 = 0, t1 = 0;
    if (wave == 0) {
      t0 = now();  // BEFORE the DMA: the stamp's own lgkmcnt(0) cannot see it
      if (mode != 2 && mode != 6) glds16(g, 4096 * 4);
      if (mode == 0 || mode == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (mode == 1 || mode == 2) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane * 4) : "memory"); sink += v; }
      if (mode == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (mode == 4) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(lane * 4 + 1024), "v"(sink) : "memory");
      if (mode == 7) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(1)" : "=v"(v) : "v"(lane * 4) : "memory"); asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); sink += v; }
      // the closing stamp: s_memtime alone, its value waited for with lgkmcnt(0) only AFTER reading the counter would be
      // circular -- so take it with a vmcnt-free, lgkm-counted read and accept that modes 0/1/4 include that wait by design
      asm volatile("s_memtime %0" : "=s"(t1)::"memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode == 5) {
      __builtin_amdgcn_s_sleep(2);
      t0 = now();
      uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane * 4) : "memory"); sink += v;
      t1 = now();
    }
    if (mode == 5 && wave == 0) { t0 = t1 = 0; }
    acc += t1 - t0;
  }
  if (lane == 0 && ((mode == 5) == (wave == 1))) out[blockIdx.x] = acc + (sink == 12345678u);
}

int main() {
  uint32_t *src; long long *out;
  const size_t bytes = (size_t)256 * 64 * 257 * 64 * 16;
  (void)hipMalloc(&src, bytes); (void)hipMemset(src, 1, bytes); (void)hipMalloc(&out, 256 * 8);
  const char *names[] = {"DMA; lgkmcnt(0)", "DMA; ds_read; lgkmcnt(0)", "no DMA; ds_read; lgkmcnt(0)", "DMA; vmcnt(0)", "DMA; ds_write; lgkmcnt(0)",
                         "other wave's ds_read while wave 0 has a DMA in flight", "no DMA; lgkmcnt(0)", "DMA; ds_read; lgkmcnt(1)"};
  for (int mode = 0; mode < 8; ++mode) {
    k<<<256, 128>>>(src, out, mode, 257);
    long long h[256];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    printf("mode %d  %-56s %8.1f cycles (s_memtime ticks) per iteration, stamp cost included\n", mode, names[mode], s / 256 / 64);
  }
  return 0;
}
