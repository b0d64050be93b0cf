// txm_i8t_common.h -- pieces shared by the transposing-read int8 bootstrap kernels (txm_resample_i8t.hip: the sampler fill
// inside the kernel, 64 replicates per workgroup; txm_resample_i8g.hip: counts from a table in HBM, 128 replicates) and the
// count-table generator (txm_count_table.hip).
#pragma once
#include "txm_resample_i8.h"
#include "txm_sampler.h"

#include <type_traits>
#include <utility>

namespace txm {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) v2i *lds_v2i;
#ifdef TXM_T_NO_TR  // ablation build: plain 8-byte reads in place of the transposing ones
#define T_TRREAD(p) (*(p))
#else
#define T_TRREAD(p) __builtin_amdgcn_ds_read_tr8_b64_v2i32(p)
#endif

constexpr int T_BLOCK = 512;  // 8 waves, two per SIMD (256 registers each: 11 accumulator tiles in AGPRs + 80 VGPRs)
constexpr int T_WAVES = T_BLOCK / 64;
constexpr int T_CNT_BYTES = (SM_T / 4) * I8_REPS * 4;  // 65536: [256 words][64 replicates]
constexpr int T_FRAG = 1024;                           // [32 samples][4 columns][8 bytes]
constexpr int T_STEPS = SM_T / 32;
#ifndef TXM_T_XD
#define TXM_T_XD 2
#endif
constexpr int T_XD = TXM_T_XD;  // k-steps between the request of an x chunk and its use (the step loop is unrolled by max(4, depth))
#ifndef TXM_T_XDN
#define TXM_T_XDN 8
#endif
// ... of the narrow-state variant: its k-steps are short (a few MFMAs per wave), so the same memory latency is more
// k-steps (measured at BASELINE config 2: depth 2 left the k-steps waiting for x -- 850 cycles each for 3 MFMAs)
constexpr int T_XDN = TXM_T_XDN;
#ifndef TXM_T_XDN1
#define TXM_T_XDN1 4
#endif
constexpr int T_XDN1 = TXM_T_XDN1;  // ... of one-quad states (C <= 4)
static_assert(T_XDN1 == 1 || T_XDN1 == 2 || T_XDN1 == 4 || T_XDN1 == 8, "ring depth");
static_assert((T_XD == 1 || T_XD == 2 || T_XD == 4 || T_XD == 8) && (T_XDN == 1 || T_XDN == 2 || T_XDN == 4 || T_XDN == 8), "ring depth");
// 1.5 * 2^52 + 0x80 in each of the six low mantissa bytes (the digits come out biased by 128; byte 6 holds
// 0x38 + digit 6, taken out at flush time as 56 * draws; byte 7 is the sign/exponent byte: the dead slot)
constexpr int T_PLANE = 512, T_PB = 2 * T_PLANE + 128;  // X region of a wave: bytes per plane, per (wave, row set)
constexpr double T_MAGIC = 6755399441055744.0 + 141289400074368.0;
constexpr int T_D6_BIAS = 0x38;

// Registers.  A wave holds up to 11 accumulator tiles = 176 registers of its 256.  The compiler's default splits a
// 256-register budget 128 : 128 between VGPRs and AGPRs as soon as a function uses AGPRs, which leaves room for 8 tiles
// only: the rest would migrate between the two files around every MFMA (v_accvgpr moves by the hundred per k-step) or
// spill.  This file is therefore compiled with  -mllvm -amdgpu-mfma-vgpr-form  (thermoextrap_amd/_build.py): every
// MFMA takes its accumulator in VGPRs, the kernel uses no AGPR at all and all 256 registers are one file.
// (Pinning register classes with inline-asm MFMAs was tried first and is WRONG under register pressure: the compiler
// does not know an asm's output is an MFMA result, so a spill store placed right behind it reads the registers before
// the matrix pipe has written them -- silent wrong sums, measured.)
template <int... I, class F>
__device__ __forceinline__ void t_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void t_static_for(F &&f) {
  t_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <bool AG>
__device__ __forceinline__ void t_mfma(v16i &c, const v4i &a, const v4i &b) {
#ifdef TXM_T_NO_MFMA  // ablation build
  asm volatile("" : "+v"(const_cast<v4i &>(a)), "+v"(const_cast<v4i &>(b)));
  return;
#endif
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

template <bool ALL_VALID>
__device__ __forceinline__ void t_fill_call(uint32_t *cntw, uint32_t k0, uint32_t k1, uint32_t rs, uint32_t t, uint32_t c,
                                            uint32_t n, uint32_t lane4) {
  // lane = replicate: counter word 2 differs per lane, words 0, 1, 3 are wave-uniform
  const uint32_t first = c * 12u;
  // (XOR3: the two XORs of a Philox round as one v_bitop3_b32 -- 3 % of the kernel here; it lost in the old kernel's
  // fused fill, txm_sampler.h)
  const Philox4 o = philox4x32_10<true>(t, c, rs, 3u, k0, k1);
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
    const uint32_t word = o.w[wi];
    // byte lane of every field at once: keep bits {0, 1} of the three fields, so that a plain shift leaves 8 * (f & 3) in
    // the five bits the shifter reads and zeros below them.  The opaque values keep this selection (mask once; bfe +
    // lshl_add for the address): 4 instead of 6 vector instructions per draw -- left alone the optimiser re-splits the
    // shared mask into shift, and, and per field
    uint32_t lo2 = word & 0x00300C03u;
    asm volatile("" : "+v"(lo2));
#ifdef TXM_FILL_BYTE_TILE  // (timing stand-in, round 6: a byte-per-(sample, replicate) tile [sample][64 replicates] -- the increment is
    // a per-lane constant, the address one bfe + one lshl_add: 2 instead of 4 vector instructions per draw.  Four replicates share
    // a count word, so the 32 lanes of a DS pass hit 8 + 8 banks instead of 32: what the ds_add costs then is what this measures)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint32_t fld = __builtin_amdgcn_ubfe(word, 10 * k, 10);
      asm volatile("" : "+v"(fld));
      uint32_t incb = 1u << ((lane4 & 12u) << 1);
      if (!ALL_VALID) incb = (first + (uint32_t)(wi * 3 + k) < n) ? incb : 0u;
      atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(cntw) + (fld << 6) + ((lane4 >> 2) & ~3u)), incb);
    }
    continue;
#endif
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint32_t q = __builtin_amdgcn_ubfe(word, 10 * k + 2, 8);  // field >> 2: the count word of the sample
      asm volatile("" : "+v"(q));
      uint32_t inc = 1u << (((k == 0) ? (lo2 << 3) : (lo2 >> (10 * k - 3))) & 31u);
      if (!ALL_VALID) inc = (first + (uint32_t)(wi * 3 + k) < n) ? inc : 0u;
#ifdef TXM_FILL_NO_ATOMIC  // (timing build: the fill's vector work without its LDS atomics)
      uint32_t ad_ = (q << 8) + lane4;
      asm volatile("" ::"v"(ad_), "v"(inc));
#else
      atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(cntw) + (q << 8) + lane4), inc);
#endif
    }
  }
}

}  // namespace txm
