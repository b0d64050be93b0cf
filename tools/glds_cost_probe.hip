// Probe (round 5): what does a global_load_lds piece cost the wave that ISSUES it, and does a DS instruction of that wave wait
// behind its outstanding pieces?  (The table kernel loses ~20 ms of 75 to its LDS-DMA although the bytes are few.)
//   t1 - t0: issue of NP pieces (no wait)          t2 - t1: one ds_read_b32 + lgkmcnt(0) right behind them
//   t3 - t0: until vmcnt(0)
// modes: 0 = s_mov m0 per piece (what the kernel does), 1 = one m0, instruction offsets 0, 1024, ... (offset moves BOTH addresses),
//        2 = plain global_load_dwordx4 into registers (no LDS), 3 = no pieces at all (the stamps' own cost),
//        4 = as 0, but every piece gathers 32 B out of each of 32 rows of 256 B (what the kernel's x pieces do)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t stamp() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int MODE, int NP>
__global__ __launch_bounds__(512) void k(const unsigned char *src, uint64_t *out, int active_waves, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (wave >= active_waves) return;
  const unsigned char *s = src + (size_t)(blockIdx.x & 63) * 65536 + wave * 8192;  // 4 MiB in all: L2 resident
  const uint32_t dst = 16384 + wave * 8192;
  uint64_t a1 = 0, a2 = 0, a3 = 0;
  v4i r[4] = {};
  for (int it = 0; it < iters; ++it) {
    const uint64_t t0 = stamp();
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(s + (i & 7) * 1024), "v"(lane * 16), "s"(dst + (i & 7) * 1024) : "memory", "m0");
    } else if constexpr (MODE == 1) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory", "m0");
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if ((i & 3) == 0) asm volatile("global_load_lds_dwordx4 %1, %0" ::"s"(s), "v"(lane * 16) : "memory");
        if ((i & 3) == 1) asm volatile("global_load_lds_dwordx4 %1, %0 offset:1024" ::"s"(s), "v"(lane * 16) : "memory");
        if ((i & 3) == 2) asm volatile("global_load_lds_dwordx4 %1, %0 offset:2048" ::"s"(s), "v"(lane * 16) : "memory");
        if ((i & 3) == 3) asm volatile("global_load_lds_dwordx4 %1, %0 offset:3072" ::"s"(s), "v"(lane * 16) : "memory");
      }
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(s + (i & 7) * 32), "v"((lane >> 1) * 256 + (lane & 1) * 16), "s"(dst + (i & 7) * 1024) : "memory", "m0");
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NP; ++i) asm volatile("global_load_dwordx4 %0, %2, %1" : "=v"(r[i & 3]) : "s"(s + (i & 7) * 1024), "v"(lane * 16) : "memory");
    }
    const uint64_t t1 = stamp();
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lane * 4) : "memory");
    const uint64_t t2 = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t3 = stamp();
    r[0][0] += v;
    if (it > 0) { a1 += t1 - t0; a2 += t2 - t1; a3 += t3 - t0; }
  }
  if (lane == 0 && blockIdx.x == 7) {
    out[wave * 4 + 0] = a1 / (iters - 1); out[wave * 4 + 1] = a2 / (iters - 1); out[wave * 4 + 2] = a3 / (iters - 1);
    out[wave * 4 + 3] = r[0][0] + r[1][1] + r[2][2] + r[3][3];
  }
  if (MODE == 1 && blockIdx.x == 7 && wave == 0) {  // functional: did piece i land at dst + 1024 i ?
    __builtin_amdgcn_s_waitcnt(0);
    const uint32_t *l = reinterpret_cast<const uint32_t *>(lds + dst);
    if (lane == 0) out[60] = ((uint64_t)l[0] << 32) | l[256], out[61] = ((uint64_t)l[512] << 32) | l[768];
  }
}

template <int MODE, int NP>
void run(const unsigned char *src, uint64_t *out, int active, const char *name) {
  (void)hipFuncSetAttribute((const void *)k<MODE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  (void)hipMemset(out, 0, 512);
  k<MODE, NP><<<256, 512, 160000>>>(src, out, active, 200);
  (void)hipDeviceSynchronize();
  uint64_t h[64];
  (void)hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz on gfx9?  report raw ticks
  printf("%-34s NP=%2d waves=%d: issue %6llu  ds_read behind %6llu  until landed %6llu   (wave %d: %llu %llu %llu)\n", name, NP, active,
         (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[2], active - 1, (unsigned long long)h[(active - 1) * 4],
         (unsigned long long)h[(active - 1) * 4 + 1], (unsigned long long)h[(active - 1) * 4 + 2]);
  if (MODE == 1) printf("    landed words: %08llx %08llx %08llx %08llx\n", (unsigned long long)(h[60] >> 32), (unsigned long long)(h[60] & 0xffffffffu),
                        (unsigned long long)(h[61] >> 32), (unsigned long long)(h[61] & 0xffffffffu));
}

int main() {
  unsigned char *src; uint64_t *out;
  (void)hipMalloc(&src, 8 << 20); (void)hipMalloc(&out, 512);
  uint32_t *h = new uint32_t[2 << 20];
  for (int i = 0; i < (2 << 20); ++i) h[i] = i * 4;  // every dword holds its own byte offset
  (void)hipMemcpy(src, h, 8 << 20, hipMemcpyHostToDevice);
  for (int active : {1, 4, 8}) {
    run<3, 0>(src, out, active, "no pieces (stamps only)");
    run<0, 4>(src, out, active, "m0 per piece");
    run<0, 14>(src, out, active, "m0 per piece");
    run<1, 4>(src, out, active, "one m0 + instruction offsets");
    run<1, 12>(src, out, active, "one m0 + instruction offsets");
    run<2, 4>(src, out, active, "global_load_dwordx4 (registers)");
    run<2, 14>(src, out, active, "global_load_dwordx4 (registers)");
    run<4, 4>(src, out, active, "m0 per piece, 32 rows x 32 B");
    run<4, 8>(src, out, active, "m0 per piece, 32 rows x 32 B");
    run<4, 14>(src, out, active, "m0 per piece, 32 rows x 32 B");
  }
  return 0;
}
