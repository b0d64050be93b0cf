// txm_resample_i8t.hip -- the int8 bootstrap contraction with the data operand built by the LDS TRANSPOSING READ of
// gfx950 (ds_read_b64_tr_b8).  Same sums, same fixed-point slicing and same sampler stream as txm_resample_i8.hip
// (cmomy.wrap_resample_vals as called from thermoextrap data.py:1803-1810, 1354-1366):
//        S1[r][c][j] = sum_i f[r][i] w_i du_i^j dx_ic        S0[r][j] = sum_i f[r][i] w_i du_i^j
//
// Why another kernel.  In txm_resample_i8.hip every monomial X = rint(m 2^50) is cut into its seven int8 digits by
// the VALU (v_perm, v_xor), stored digit by digit into MFMA-ready fragments in LDS (22 stores per wave and k-step)
// and de-interleaved again by the consumer: the k-step is bound by the LDS write path and by the vector issue of that
// byte shuffling, the matrix pipe idles at 24 %.  Here the 8-byte fixed-point word is stored AS IT IS -- one
// ds_write_b128 per lane, power and pair of columns: [fragment][sample][4 columns][8 bytes] -- and the byte transpose
// is done by the LDS hardware on the way back: one ds_read_b64_tr_b8 hands every lane of a 16-lane group one BYTE
// COLUMN of 8 rows x 16 bytes, i.e. digit (lane & 7) of column (lane >> 3) for 8 consecutive samples -- exactly one
// half of an MFMA B operand whose 32 columns are (4 observable columns) x (8 digit slots).  Slot 7 (the exponent byte
// of the magic-constant double) is a dead column: 12.5 % more MFMAs, no vector instruction at all between the LDS and
// the matrix pipe.  Per monomial the VALU work is one v_fma_f64 and two v_xor_b32 (bias removal), against ~10
// instructions before; the kernel is now paced by the matrix pipe.
//
// Workgroup = 8 waves (two per SIMD, 256 registers: up to 11 int32 accumulator tiles of 32 x 32 per wave) x 64
// replicates x one group of 32 observable columns.  (One wave per SIMD with 21 tiles was built first: an in-order wave
// alone adds its LDS-store stalls, its vector work and its MFMAs up -- 2350 cycles per k-step against 840 of matrix
// work; two waves per SIMD fill each other's stalls.)
//   fragment f = (row set rs, column quad cq): rs = power J0 + rs of the launch; 32 tile columns = columns 4 cq ..
//   4 cq + 3 x digit slots 0..7.  u-row fragments (dx = 1): tile column = (monomial, digit slot), 4 monomials each.
//   wave w owns column quad w of every row set (both replicate halves); waves 0..3 one u-row tile each; it slices
//   the samples 4 w .. 4 w + 3 of every 32-sample chunk.
// Count tile: cnt[word g = sample / 4][replicate], one u32 = the u8 counts of 4 samples.  Stage 3 of the sampler runs
// with ONE LANE PER REPLICATE (lane = replicate, the eight waves split the Philox calls): all 64 lanes of a ds_add hit
// 64 consecutive words -- no bank conflict by construction (the old layout lost 11 cycles per ds_add to conflicts).
// Partial sums: one slot per SCALING WINDOW (a fixed block of samples: the window size depends on N only),
// part[window][replicate][power][column][digit slot], stored once -- no read-modify-write, no zeroing -- and added up
// by the finalize kernel in window order.  A replicate's result therefore does not depend on how many replicates,
// chunks or workgroups the launch had: rows [a, b) of a bootstrap equal the (b - a)-replicate call with rep0 = a
// bit for bit (multi-GPU slabs, txm_sampler_spec.rep0).
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
#define TXM_T_XD 4
#endif
constexpr int T_XD = TXM_T_XD;  // k-steps between the request of an x chunk and its use (1, 2 or 4: the step loop is unrolled by 4)
static_assert(T_XD == 1 || T_XD == 2 || T_XD == 4, "ring depth");
// 1.5 * 2^52 + 0x80 in each of the six low mantissa bytes (the digits come out biased by 128; byte 6 holds
// 0x38 + digit 6, taken out at flush time as 56 * draws; byte 7 is the sign/exponent byte: the dead slot)
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
  const Philox4 o = philox4x32_10(t, c, rs, 3u, k0, k1);
#pragma unroll
  for (int wi = 0; wi < 4; ++wi) {
    const uint32_t word = o.w[wi];
    const uint32_t lo2 = word & 0x00300C03u;  // bits {0, 1} of the three fields: the byte lane
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t q = __builtin_amdgcn_ubfe(word, 10 * k + 2, 8);  // field >> 2: the count word of the sample
      uint32_t inc = 1u << (((k == 0) ? (lo2 << 3) : (lo2 >> (10 * k - 3))) & 31u);
      if (!ALL_VALID) inc = (first + (uint32_t)(wi * 3 + k) < n) ? inc : 0u;
      atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(cntw) + (q << 8) + lane4), inc);
    }
  }
}

// K = order + 1 is a run-time argument (it only enters the flush addresses); one launch slices the JN powers
// J0 .. J0 + JN - 1.
template <int J0, int JN, bool WEIGHTED>
__global__ __launch_bounds__(T_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void resample_i8t_kernel(const I8Args a, const int K) {
  static_assert(JN >= 1 && JN <= 5 && J0 + JN <= 8, "power range");
  static_assert(!WEIGHTED || JN <= 4, "weighted launches stage a second 8 KiB tile: four row sets at most");
  constexpr int NS = JN;             // row sets of the launch = x fragments per wave
  constexpr int UF = (JN + 3) / 4;   // u-row fragments (4 monomials each)
  constexpr int NF = NS * 8 + UF;
  constexpr int XBUF = NF * T_FRAG;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *cntw = reinterpret_cast<uint32_t *>(lds);
  unsigned char *xt0 = lds + T_CNT_BYTES;
  unsigned char *xt1 = xt0 + XBUF;
  uint32_t *fsum = reinterpret_cast<uint32_t *>(xt1 + XBUF);  // [64] draws per replicate in the window
  uint32_t *cnt_a = fsum + I8_REPS;                           // [64] tile draw counts, double buffered
  uint32_t *cnt_b = cnt_a + I8_REPS;
  // the scaled u deviations (u - pu) / max|u - pu| (and weights w / max|w|) of the 1024 samples whose X words this
  // tile's k-steps produce: chunks 1 .. 31 of the tile and chunk 0 of the next one -- staged once per tile by the whole
  // workgroup, so the k-steps load nothing but x
  double *utile = reinterpret_cast<double *>(cnt_b + I8_REPS);
  double *wtile = utile + SM_T;  // WEIGHTED only

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int n32 = lane & 31, half = lane >> 5;

  // ---- producer role: wave w slices unit w of every chunk = 4 samples x 32 columns;
  // lane = (column quad, sample, column pair)
  const int cqd = lane >> 3, s4 = (lane & 7) >> 1, hf = lane & 1;
  // (the launcher takes this kernel only for full groups of 32 columns whose rows are 16-byte aligned:
  // i8t_applicable; everything else stays on txm_resample_i8.hip)
  const int c0 = 4 * cqd + 2 * hf;
  const int cc0 = c0, cc1 = c0 + 1;
  const uint32_t wr_off = (uint32_t)(cqd * T_FRAG + (wave * 4 + s4) * 32 + hf * 16);  // + row set * 8 KiB
  // per-lane BYTE offsets from a wave-uniform row base (saddr + zext(voffset) form of global_load: no 64-bit
  // vector address arithmetic in the k-step)
  const uint32_t xo0 = (uint32_t)((s4 * a.ldx_s + cc0) * 8);
  // u-row monomials of this wave's 4 samples: lane = (monomial, sample)
  const int us = lane & 3, um = lane >> 2;
  const uint32_t uw_off = (uint32_t)((NS * 8 + ((um >> 2) < UF ? (um >> 2) : 0)) * T_FRAG + (wave * 4 + us) * 32 + (um & 3) * 8);
  // ---- consumer role: wave w owns column quad w of every row set (both replicate halves) and, waves 0 .. 2 UF - 1,
  // one u-row tile
  const uint32_t rd_off = (uint32_t)(wave * T_FRAG + (16 * half + ((lane & 15) >> 1)) * 32 + ((lane >> 4) & 1) * 16 +
                                     (lane & 1) * 8);  // + row set * 8 KiB; second read + 256
  const uint32_t a_off = (uint32_t)((4 * half) * I8_REPS + n32);  // words; + 8 s * 64 + q * 64 (+ 32: second half)
  constexpr int NUT = 2 * UF;
  const bool has_ut = wave < NUT;  // wave-uniform
  const int fu = has_ut ? (wave >> 1) : 0, uh = wave & 1;
  const int urd_delta = __builtin_amdgcn_readfirstlane((NS * 8 + fu - wave) * T_FRAG);  // u-row fragment relative to rd_off

  const int b = blockIdx.x;
  const int xcd = b & 7, qq = b >> 3;
  const int chunk = (qq / a.n_rbg) * 8 + xcd;
  const int rbg = qq % a.n_rbg;
  const int64_t rep0 = (int64_t)rbg * I8_REPS;
  const int64_t t_begin = (int64_t)chunk * a.tiles_per_chunk;
  int64_t t_end = t_begin + a.tiles_per_chunk;
  if (t_end > a.ntiles) t_end = a.ntiles;

  const double pu = a.pivot[0];
  const double px0 = a.pivot[1 + a.col0 + cc0], px1 = a.pivot[1 + a.col0 + cc1];

  v16i acc[NS][2];
  v16i accu;
#pragma unroll
  for (int e = 0; e < NS; ++e) acc[e][0] = acc[e][1] = (v16i)(0);
  accu = (v16i)(0);

  // table slots that are never written (columns >= C re-read column 0 and ARE written; the unused monomial slots of a
  // short u-row fragment are not) hold integers that are never flushed: zero them for determinism
  for (int e = threadIdx.x; e < 2 * XBUF / 16; e += T_BLOCK) reinterpret_cast<uint4 *>(xt0)[e] = make_uint4(0, 0, 0, 0);

  // stage-3 role: lane = replicate
  const int64_t my_rep = rep0 + lane;
  const bool rep_live = my_rep < a.nrep;
  const uint32_t rstream = a.rep_base + (uint32_t)my_rep;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  uint32_t fdraws = 0;

  struct XIn {
    double x0, x1;
  };
  // the wave's unit of one chunk: i0 = its first sample (wave-uniform)
  auto load_x = [&](int64_t i0, XIn &r) {
#ifdef TXM_T_NO_LOAD  // ablation build: no memory access
    r.x0 = (double)i0 * 1e-9 + px0;
    r.x1 = (double)i0 * 2e-9 + px1;
    return;
#endif
    const double *xr = a.x + i0 * a.ldx_s + a.col0;
    const double2 t2 = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(xr) + xo0);
    r.x0 = t2.x;
    r.x1 = t2.y;
  };

#ifdef TXM_I8T_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk0 = clock64();
#define T_TICK(k) do { const long long t1_ = clock64(); tm[k] += t1_ - tk0; tk0 = t1_; } while (0)
#else
#define T_TICK(k) do {} while (0)
#endif
  uint32_t *pg = a.progress != nullptr ? a.progress + (size_t)chunk * 64 : nullptr;
  uint32_t tiles_done = 1;
  const int64_t WT = a.win_tiles;
  auto tile_base = [&](int64_t tt) {
    const int64_t b0 = tt * SM_T;
    return b0 > a.N - SM_T ? a.N - SM_T : b0;  // the last tile slides its window back (zero counts for foreign samples)
  };

  for (int64_t win = t_begin / WT; win * WT < t_end; ++win) {
    if (a.wflag[win] != 0u) continue;  // precision guard: this window goes to the FP64 kernel (uniform)
    const double *wt = a.wtab + win * I8_WT_STRIDE;
    const double inv_du = wt[I8_WT_INVDU];
    const double inv_w = WEIGHTED ? wt[I8_WT_INVW] : 1.0;
    const double sc0 = wt[I8_WT_SC + cc0], sc1 = wt[I8_WT_SC + cc1];
    int64_t tt_end = (win + 1) * WT;
    if (tt_end > t_end) tt_end = t_end;

    // ---- the unit of one chunk: 4 samples x 32 columns x NS powers -> the X table of buffer `nxt`
    // du / w: `dup` points at the entry of this lane's sample (the staged tiles, or two scalars of the direct path)
    auto produce_head = [&](const XIn &r, double duv, double wv, double &du, double &p, double &dx0, double &dx1) {
      du = duv;
      dx0 = (r.x0 - px0) * sc0;
      dx1 = (r.x1 - px1) * sc1;
      p = WEIGHTED ? wv : 1.0;
#pragma unroll
      for (int q = 0; q < J0; ++q) p *= du;
    };
    auto produce_power = [&](unsigned char *nxt, int jj, double du, double &p, double dx0, double dx1) {
      if (jj > 0) p *= du;
      const uint64_t b0 = (uint64_t)__double_as_longlong(fma(p, dx0, T_MAGIC));
      const uint64_t b1 = (uint64_t)__double_as_longlong(fma(p, dx1, T_MAGIC));
      uint4 v;
      v.x = (uint32_t)b0 ^ 0x80808080u;
      v.y = (uint32_t)(b0 >> 32) ^ 0x00008080u;
      v.z = (uint32_t)b1 ^ 0x80808080u;
      v.w = (uint32_t)(b1 >> 32) ^ 0x00008080u;
#ifdef TXM_T_NO_WRITE  // ablation build: the values stay live, nothing is stored
      asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
      *reinterpret_cast<uint4 *>(nxt + wr_off + jj * 8 * T_FRAG) = v;
#endif
    };
    // the u-row monomials w du^(J0 + m) of this wave's 4 samples (dx = 1): lane = (monomial m, sample)
    auto produce_urow = [&](unsigned char *nxt, double du, double wv) {
      double p = WEIGHTED ? wv : 1.0;
#pragma unroll
      for (int q = 0; q < J0; ++q) p *= du;
      if constexpr (JN > 1) {
        const double d2 = du * du;
        p *= (um & 1) ? du : 1.0;
        if constexpr (JN > 2) p *= (um & 2) ? d2 : 1.0;
        if constexpr (JN > 4) p *= (um & 4) ? d2 * d2 : 1.0;
      }
      // (ldexp + add: exact, and no second 64-bit literal for the register allocator to park in scratch -- a scratch
      // reload in the k-step is a vmcnt(0) wait on the x requests just issued)
      const uint64_t b0 = (uint64_t)__double_as_longlong(__builtin_ldexp(p, 50) + T_MAGIC);
      uint2 v;
      v.x = (uint32_t)b0 ^ 0x80808080u;
      v.y = (uint32_t)(b0 >> 32) ^ 0x00008080u;
      if (um < JN) *reinterpret_cast<uint2 *>(nxt + uw_off) = v;
    };

    // ---- one k-step: chunk s of the current tile on the matrix pipe out of `cur`, the wave's unit of chunk s + 1
    // produced into `nxt` from the samples in R (loaded one step earlier)
    // (e0 = entry of the wave's first sample of the produced chunk in the staged tiles; e0 < 0: the direct path of a
    // window's first chunk -- du / w come in the four scalars)
    auto kstep = [&](auto produce_c, auto consume_c, const unsigned char *cur, unsigned char *nxt, int s, const XIn &R,
                     int e0, double d_du = 0.0, double d_w = 1.0, double d_duu = 0.0, double d_wu = 1.0) {
#ifdef TXM_T_NO_PRODUCE  // ablation build
      constexpr bool produce = false;
#else
      constexpr bool produce = decltype(produce_c)::value;
#endif
      constexpr bool consume = decltype(consume_c)::value;
      v4i A0 = (v4i)(0), A1 = (v4i)(0);
      v2i Ba = (v2i)(0), Bb = (v2i)(0);
      if constexpr (consume) {
        const uint32_t *cw = cntw + s * (8 * I8_REPS) + a_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          A0[q] = (int)cw[q * I8_REPS];
          A1[q] = (int)cw[q * I8_REPS + 32];
        }
        Ba = T_TRREAD((lds_v2i)(cur + rd_off));
        Bb = T_TRREAD((lds_v2i)(cur + rd_off + 256));
      }
      double du = 0.0, p = 0.0, dx0 = 0.0, dx1 = 0.0, duu = d_duu, wu = d_wu;
      if constexpr (produce) {
        double duv = d_du, wv = d_w;
        if (e0 >= 0) {  // uniform
          duv = utile[e0 + s4];
          duu = utile[e0 + us];
          if constexpr (WEIGHTED) {
            wv = wtile[e0 + s4];
            wu = wtile[e0 + us];
          }
        }
        produce_head(R, duv, wv, du, p, dx0, dx1);
      }
      t_static_for<NS>([&](auto fic) {
        constexpr int fi = decltype(fic)::value;
        v2i Na = (v2i)(0), Nb = (v2i)(0);
        if constexpr (consume) {
          if constexpr (fi + 1 < NS) {
            Na = T_TRREAD((lds_v2i)(cur + rd_off + (fi + 1) * 8 * T_FRAG));
            Nb = T_TRREAD((lds_v2i)(cur + rd_off + (fi + 1) * 8 * T_FRAG + 256));
          }
          const v4i B = {Ba[0], Ba[1], Bb[0], Bb[1]};
          t_mfma<(fi < 4)>(acc[fi][0], A0, B);  // 8 tiles in the 128 AGPRs, the fifth row set and the u-row tile in VGPRs
          t_mfma<(fi < 4)>(acc[fi][1], A1, B);
        }
        if constexpr (produce) produce_power(nxt, fi, du, p, dx0, dx1);
        Ba = Na;
        Bb = Nb;
      });
      if constexpr (produce) produce_urow(nxt, duu, wu);
      if constexpr (consume) {
        if (has_ut) {  // wave-uniform
          Ba = T_TRREAD((lds_v2i)(cur + urd_delta + rd_off));
          Bb = T_TRREAD((lds_v2i)(cur + urd_delta + rd_off + 256));
          const v4i B = {Ba[0], Ba[1], Bb[0], Bb[1]};
          const v4i Au = {uh ? A1[0] : A0[0], uh ? A1[1] : A0[1], uh ? A1[2] : A0[2], uh ? A1[3] : A0[3]};
          t_mfma<false>(accu, Au, B);
        }
      }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};

    // ---- flush: int32 accumulators of one window -> its slot of the partial sums (stored, never re-read here)
    // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
    auto flush_tile = [&](v16i &T, int h, int rs, int cq, int ufrag) {
      // `opq` is an opaque zero re-created per tile: the addresses below are then computed where they are used; left
      // to itself the compiler hoists them all out of the window loop and spills them
      uint32_t z = 0;
      asm volatile("" : "+v"(z));
      const int64_t opq = (int64_t)z;
      const int cl = n32 >> 3, dg = n32 & 7;
      bool valid = dg < I8_NSL;
      int j;
      double dsc;
      double *base;
      if (ufrag < 0) {
        const int col = 4 * cq + cl;
        valid = valid && col < a.C;
        j = J0 + rs;
        dsc = wt[I8_WT_DSP + j] * wt[I8_WT_DSC + (col < a.C ? col : 0)];
        // [window][replicate][power][column][digit slot]: the 32 lanes of a row write 256 contiguous bytes
        base = a.part_x + ((((size_t)win * a.nrep_pad + rep0 + 32 * h + 4 * half) * K + j) * I8_CPAD + col) * 8 + dg + opq;
      } else {
        const int m = 4 * ufrag + cl;
        valid = valid && m < JN;
        j = J0 + (m < JN ? m : 0);
        dsc = wt[I8_WT_DSP + j] * 0x1p-50;
        base = a.part_u + (((size_t)win * a.nrep_pad + rep0 + 32 * h + 4 * half) * K + j) * 8 + dg + opq;
      }
      dsc *= (double)((int64_t)1 << (8 * (dg < I8_NSL ? dg : 0)));
      const size_t stride = (size_t)K * (ufrag < 0 ? I8_CPAD : 1) * 8;  // one replicate
      const int bias = dg == I8_NSL - 1 ? T_D6_BIAS : 0;
      if (valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r >> 2) * 8 + (r & 3);
          const int v = T[r] - bias * (int)fsum[32 * h + m + 4 * half];
          base[(size_t)m * stride] = (double)v * dsc;
        }
      }
      T = (v16i)(0);
    };

    bool first_tile = true;
    uint32_t *cnt_cur = cnt_a, *cnt_nxt = cnt_b;
    XIn XR[T_XD];  // x of the wave's unit, requested T_XD k-steps ahead: chunk c lives in slot c % T_XD
    auto load_chunk = [&](int64_t wb, int c, XIn &R) { load_x(wb + c * 32 + wave * 4, R); };

#pragma unroll 1
    for (int64_t t = win * WT; t < tt_end; ++t) {
      const int64_t i_tile = t * SM_T;
      const uint32_t tsize = (t == a.ntiles - 1) ? a.last_tile_size : (uint32_t)SM_T;
      const int64_t wbase = tile_base(t);
      const uint32_t shift = (uint32_t)(i_tile - wbase);
      const bool has_next = t + 1 < tt_end;
      const int64_t wnext = has_next ? tile_base(t + 1) : wbase;

      if (pg != nullptr && wave == 0) {  // L2-sharing hint (bounded; no result depends on it)
        if (lane == 0) __hip_atomic_store(&pg[rbg & 63], tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
        for (int spin = 0; spin < I8_THROTTLE_SPINS; ++spin) {
          uint32_t v = __hip_atomic_load(&pg[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v == 0u) v = 0xffffffffu;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const uint32_t w2 = (uint32_t)__shfl_xor((int)v, o);
            v = w2 < v ? w2 : v;
          }
          if (tiles_done <= v + I8_LEAD) break;
          __builtin_amdgcn_s_sleep(32);
        }
      }
      ++tiles_done;
      T_TICK(0);

      // ---- tile prologue: counts of this tile (first tile of a window: loaded here; otherwise parked by the
      // previous tile), zero the count tile, chunk 0 of the first tile
      if (first_tile) {
        if (wave == 0) cnt_cur[lane] = rep_live ? a.counts[(size_t)my_rep * a.ntiles + t] : 0u;
        load_chunk(wbase, 0, XR[0]);
      }
      // staging requests (in flight during the zeroing and the fill): entries 0 .. 991 = samples wbase + 32 ...,
      // entries 992 .. 1023 = the next tile's first chunk
      double su[2], sw[2] = {1.0, 1.0};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (int)threadIdx.x + q * T_BLOCK;
        const int64_t i = e < SM_T - 32 ? wbase + 32 + e : wnext + (e - (SM_T - 32));
        su[q] = a.u[i];
        if constexpr (WEIGHTED) sw[q] = a.w[i];
      }
      uint32_t ncnt = 0;
      if (wave == 0 && has_next && rep_live) ncnt = a.counts[(size_t)my_rep * a.ntiles + t + 1];
      for (int e = threadIdx.x; e < T_CNT_BYTES / 16; e += T_BLOCK) reinterpret_cast<uint4 *>(cntw)[e] = make_uint4(0, 0, 0, 0);
      T_TICK(1);
      __syncthreads();
      T_TICK(2);

      // ---- stage 3 of the sampler: the 64 x 1024 count tile, lane = replicate, the waves split the Philox calls ----
      {
        uint32_t n = cnt_cur[lane];
        if (wave == 0) fdraws += n;
#ifdef TXM_T_NO_FILL  // ablation build
        if (tsize == 0u) {
#else
        if (tsize == (uint32_t)SM_T) {
#endif
          // dead lanes (replicates past nrep) draw like the smallest live lane: their columns are never flushed
          uint32_t nmin = rep_live ? n : 0xffffffffu, nmax = rep_live ? n : 0u;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)nmin, o), hi = (uint32_t)__shfl_xor((int)nmax, o);
            nmin = lo < nmin ? lo : nmin;
            nmax = hi > nmax ? hi : nmax;
          }
          nmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmin);
          nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
          if (!rep_live) n = nmin;
          const uint32_t call_all = nmin / 12u;  // calls below this index are complete for every lane
          uint32_t c = (uint32_t)wave;
#pragma unroll 1
          for (; c + T_WAVES < call_all; c += 2 * T_WAVES) {  // two calls per trip: two Philox chains in flight
            t_fill_call<true>(cntw, a.k0, a.k1, rstream, (uint32_t)t, c, n, lane4);
            t_fill_call<true>(cntw, a.k0, a.k1, rstream, (uint32_t)t, c + T_WAVES, n, lane4);
          }
#pragma unroll 1
          for (; c < call_all; c += T_WAVES) t_fill_call<true>(cntw, a.k0, a.k1, rstream, (uint32_t)t, c, n, lane4);
#pragma unroll 1
          for (; c * 12u < nmax; c += T_WAVES) t_fill_call<false>(cntw, a.k0, a.k1, rstream, (uint32_t)t, c, n, lane4);
        } else if (tsize != (uint32_t)SM_T) {
          // the partial last tile: the stream is defined over 64 virtual lanes per replicate (txm_sampler.h)
          for (int rr = wave * (I8_REPS / T_WAVES); rr < (wave + 1) * (I8_REPS / T_WAVES); ++rr) {
            const int64_t r = rep0 + rr;
            if (r >= a.nrep) break;  // wave-uniform
            const uint32_t nr = cnt_cur[rr];
            sampler_fine_tile(a.k0, a.k1, a.rep_base + (uint32_t)r, (uint32_t)t, nr, tsize, lane, [&](uint32_t off0) {
              const uint32_t off = off0 + shift;
              atomicAdd(&cntw[(off >> 2) * I8_REPS + (uint32_t)rr], 1u << ((off & 3u) << 3));
            });
          }
        }
      }
      if (first_tile) {
        // the X table of chunk 0 (no matrix work yet; its u / w straight from memory), chunks 1 .. 4 requested
        const int64_t i0 = wbase + wave * 4;
        const double d_du = (a.u[i0 + s4] - pu) * inv_du, d_duu = (a.u[i0 + us] - pu) * inv_du;
        double d_w = 1.0, d_wu = 1.0;
        if constexpr (WEIGHTED) {
          d_w = a.w[i0 + s4] * inv_w;
          d_wu = a.w[i0 + us] * inv_w;
        }
        kstep(YES, NO, xt1, xt0, 0, XR[0], -1, d_du, d_w, d_duu, d_wu);
#pragma unroll
        for (int c = 1; c <= T_XD; ++c) load_chunk(wbase, c, XR[c % T_XD]);
        first_tile = false;
      }
      // park the staged tiles (the previous tile's last k-step was their last reader)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (int)threadIdx.x + q * T_BLOCK;
        utile[e] = (su[q] - pu) * inv_du;
        if constexpr (WEIGHTED) wtile[e] = sw[q] * inv_w;
      }
      if (wave == 0) cnt_nxt[lane] = ncnt;
      T_TICK(3);
      __syncthreads();
      T_TICK(4);

      // ---- 32 k-steps, four per trip.  Step s contracts chunk s, produces chunk s + 1 from the ring slot (s + 1) % T_XD
      // (x requested T_XD steps ago: an HBM miss is ~2 k-steps long) and requests chunk s + 1 + T_XD into that slot.
      // Chunk 32 is the next tile's chunk 0 (a buffer nobody reads when there is no next tile).
      auto target = [&](int c, int64_t &wb, int &cl) {
        if (c < T_STEPS) { wb = wbase; cl = c; }
        else { wb = wnext; cl = has_next ? c - T_STEPS : T_STEPS - 1; }
      };
#pragma unroll 1
      for (int s = 0; s < T_STEPS; s += 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int sq = s + e;
          XIn &R = XR[(e + 1) % T_XD];
          const XIn cur_x = R;
          int64_t wb;
          int cl;
          target(sq + 1 + T_XD, wb, cl);
          load_chunk(wb, cl, R);
          kstep(YES, YES, (e & 1) ? xt1 : xt0, (e & 1) ? xt0 : xt1, sq, cur_x, sq * 32 + wave * 4);
          T_TICK(5);
          __syncthreads();
          T_TICK(6);
        }
      }
      {
        uint32_t *tmp = cnt_cur;
        cnt_cur = cnt_nxt;
        cnt_nxt = tmp;
      }
    }

    // ---- end of the window: flush ----
    if (wave == 0) fsum[lane] = fdraws;
    fdraws = 0;
    __syncthreads();
#pragma unroll
    for (int fi = 0; fi < NS; ++fi) {
      flush_tile(acc[fi][0], 0, fi, wave, -1);
      flush_tile(acc[fi][1], 1, fi, wave, -1);
    }
    if (has_ut) flush_tile(accu, uh, 0, 0, fu);  // wave-uniform
    accu = (v16i)(0);
    __syncthreads();  // fsum is rewritten by the next window
    T_TICK(7);
  }
  if (pg != nullptr && threadIdx.x == 0)
    __hip_atomic_store(&pg[rbg & 63], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TXM_I8T_TIMING
  // diagnostic build only: phase cycles of two workgroups into the slack behind the window table
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
    for (int k = 0; k < 8; ++k) a.wtab[a.nwin * I8_WT_STRIDE + ((blockIdx.x ? 1 : 0) * T_WAVES + wave) * 8 + k] = (double)tm[k];
#endif
}

// ---------------------------------------------------------------------------
template <int J0, int JN, bool WEIGHTED>
static int launch_pass_t(const I8Args &a, int K, size_t prog_bytes, hipStream_t st) {
  if (a.progress != nullptr) TXM_HIP(hipMemsetAsync(a.progress, 0, prog_bytes, st));
  const dim3 grid((unsigned)(a.n_chunks * a.n_rbg)), block(T_BLOCK);
  constexpr int nf = JN * 8 + (JN + 3) / 4;
  const size_t lds = (size_t)T_CNT_BYTES + 2u * nf * T_FRAG + 3u * I8_REPS * sizeof(uint32_t) +
                     (WEIGHTED ? 2u : 1u) * SM_T * sizeof(double);
  TXM_SET_MAX_LDS((&resample_i8t_kernel<J0, JN, WEIGHTED>), lds);
  hipLaunchKernelGGL((resample_i8t_kernel<J0, JN, WEIGHTED>), grid, block, lds, st, a, K);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// what this kernel takes: full groups of 32 columns (C a multiple of 32), rows 16-byte aligned -- its lanes load
// two columns as one 16-byte word.  Everything else stays on txm_resample_i8.hip (one kernel family per call, so that
// all column groups of a state round their u-row sums the same way).
bool i8t_applicable(const double *x, int64_t ldx_s, int64_t C) {
  return C % I8_CPAD == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ldx_s % 2 == 0;
}

// one power per observable column (C > 16, or order 0): every order 0..7.  Unweighted: five row sets per pass, orders
// 5..7 in two passes over the sampler stream (the matrix pipe paces a pass, so the split is by fragments: 3 + 3, 4 + 3,
// 4 + 4 row sets).  Weighted launches stage a second 8 KiB tile (the weights) in LDS and hold four row sets at most:
// order 4 takes 3 + 2.
int launch_resample_i8t(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st) {
  int rc = TXM_OK;
#define T_PASS(J0_, JN_) (weighted ? launch_pass_t<J0_, JN_, true>(a, K, prog_bytes, st) : launch_pass_t<J0_, JN_, false>(a, K, prog_bytes, st))
  switch (K) {
    case 1: rc = T_PASS(0, 1); break;
    case 2: rc = T_PASS(0, 2); break;
    case 3: rc = T_PASS(0, 3); break;
    case 4: rc = T_PASS(0, 4); break;
    case 5:
      if (weighted) { rc = launch_pass_t<0, 3, true>(a, K, prog_bytes, st); if (rc == TXM_OK) rc = launch_pass_t<3, 2, true>(a, K, prog_bytes, st); }
      else rc = launch_pass_t<0, 5, false>(a, K, prog_bytes, st);
      break;
    case 6: rc = T_PASS(0, 3); if (rc == TXM_OK) rc = T_PASS(3, 3); break;
    case 7: rc = T_PASS(0, 4); if (rc == TXM_OK) rc = T_PASS(4, 3); break;
    case 8: rc = T_PASS(0, 4); if (rc == TXM_OK) rc = T_PASS(4, 4); break;
    default: set_error("resample_i8t: order out of range"); return TXM_ERR_INVALID;
  }
#undef T_PASS
  return rc;
}

}  // namespace txm
