// Probe (round 5): what does ONE DS instruction of each kind cost a CU when all eight waves issue them back to back?
// (The overlap probe fitted ~4 cycles for a ds_read_b128 AND for a ds_write_b32: cost per instruction, not per byte?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

enum { RD_B32, RD_B64, RD_B128, RD2_B64, RD2ST64_B64, RD_TR_B8, WR_B32, WR_ADDTID, WR2ST64_B32, WR_B64, WR2_B64, WR_B128, RD_B64_BCAST, RD_B64_Q16, NKIND };
static const char *names[] = {"ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read2_b64", "ds_read2st64_b64", "ds_read_b64_tr_b8",
                              "ds_write_b32", "ds_write_addtid_b32", "ds_write2st64_b32", "ds_write_b64", "ds_write2_b64", "ds_write_b128",
                              "ds_read_b64 (4 lanes share an address)", "ds_read_b64 (16 of 64 lanes active)"};

template <int KIND>
__global__ __launch_bounds__(512) void k(int *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 40000; i += 512) ((int *)lds)[i] = i;
  __syncthreads();
  const unsigned base = wave * 8192;
  unsigned a16 = base + lane * 16, a8 = base + lane * 8, a4 = base + lane * 4, ab = base + (lane >> 2) * 8;
  v4i r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
  v4i d = {(int)lane, 1, 2, 3};
  v2i p0 = {0, 0}, p1 = p0, e0 = {(int)lane, 5}, e1 = {7, (int)lane};
  asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(base)));
  for (int it = 0; it < iters; ++it) {
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
    if constexpr (KIND == RD_B32) {
#define S(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0[i & 3]) : "v"(a4), "n"(i * 256) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == RD_B64) {
#define S(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(i & 1 ? p1 : p0) : "v"(a8), "n"(i * 512) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == RD_B64_BCAST) {
#define S(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(i & 1 ? p1 : p0) : "v"(ab), "n"(i * 512) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == RD_B64_Q16) {
      if ((lane & 3) == 0) {
#define S(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(i & 1 ? p1 : p0) : "v"(ab), "n"(i * 512) : "memory");
        REP8(S)
#undef S
      }
    } else if constexpr (KIND == RD_B128) {
#define S(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(i & 1 ? r1 : r0) : "v"(a16), "n"((i & 3) * 1024) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == RD2_B64) {
#define S(i) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(i & 1 ? r1 : r0) : "v"(a8), "n"((i & 1) * 64), "n"((i & 1) * 64 + 128) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == RD2ST64_B64) {
#define S(i) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(i & 1 ? r1 : r0) : "v"(a8), "n"(i & 3), "n"((i & 3) + 4) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == RD_TR_B8) {
#define S(i) asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(i & 1 ? p1 : p0) : "v"(a8), "n"(i * 512) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == WR_B32) {
#define S(i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a4), "v"(d[i & 3]), "n"(i * 256) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == WR_ADDTID) {
#define S(i) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d[i & 3]), "n"(i * 256) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == WR2ST64_B32) {
#define S(i) asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a4), "v"(d[i & 3]), "v"(d[(i + 1) & 3]), "n"(2 * (i & 3)), "n"(2 * (i & 3) + 1) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == WR_B64) {
#define S(i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(i & 1 ? e1 : e0), "n"(i * 512) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == WR2_B64) {
#define S(i) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a8), "v"(e0), "v"(e1), "n"((i & 1) * 64), "n"((i & 1) * 64 + 128) : "memory");
      REP8(S)
#undef S
    } else if constexpr (KIND == WR_B128) {
#define S(i) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(d), "n"((i & 3) * 1024) : "memory");
      REP8(S)
#undef S
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * 512 + threadIdx.x] = r0[0] + r1[1] + r2[2] + r3[3] + r0[2] + r1[3] + p0[0] + p0[1] + p1[0] + p1[1];
}

template <int KIND>
void run() {
  int *out; (void)hipMalloc(&out, 4 * 256 * 512);
  const int iters = 20000;
  k<KIND><<<256, 512, 160000>>>(out, 100);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<KIND><<<256, 512, 160000>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // 8 waves x 8 instructions per iteration on one CU's LDS
  printf("%-44s %8.3f ms   %6.2f ns per instruction and CU  (= %5.2f cycles at 2.4 GHz)\n", names[KIND], ms, ms * 1e6 / iters / 64, ms * 1e6 / iters / 64 * 2.4);
  (void)hipFree(out);
}
template <int... KS> void run_all(std::integer_sequence<int, KS...>) {
  ((hipFuncSetAttribute((const void *)k<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000), run<KS>()), ...);
}
int main() { run_all(std::make_integer_sequence<int, NKIND>{}); return 0; }
