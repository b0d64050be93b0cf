// txm_resample_i8t.hip -- the bootstrap contraction on the int8 matrix pipe, the data operand built by the LDS
// TRANSPOSING READ of gfx950 (ds_read_b64_tr_b8).  The sums (cmomy.wrap_resample_vals as called from thermoextrap
// data.py:1803-1810, 1354-1366; fixed-point slicing, scaling windows and the precision guard: txm_resample_i8.hip):
//        S1[r][c][j] = sum_i f[r][i] w_i du_i^j dx_ic        S0[r][j] = sum_i f[r][i] w_i du_i^j
//
// The design in one paragraph.  The round-1/2 kernel cut every monomial X = rint(m 2^50) into its seven int8 digits on
// the VALU (v_perm, v_xor), stored them digit by digit into MFMA-ready fragments in LDS (22 stores per wave and k-step)
// and de-interleaved them again on the read side: its k-step was bound by the LDS write path and by the vector issue of
// that byte shuffling, the matrix pipe idled at 24 %.  Here the 8-byte fixed-point word is stored AS IT IS -- its low
// and high dwords into two planes [sample][4 columns] -- and the byte transpose is done by the LDS hardware on the way
// back: one ds_read_b64_tr_b8 hands every lane of a 16-lane group one BYTE COLUMN of 8 rows x 16 bytes, i.e. digit
// (i & 3) of column (i >> 2) of a plane for 8 consecutive samples -- one half of an MFMA B operand whose 32 tile columns
// are (4 observable columns) x (8 digit slots).  Slot 7 (the exponent byte of the magic-constant double) is a dead
// column: 12.5 % more MFMAs, no vector instruction at all between the LDS and the matrix pipe.  Per monomial the VALU work
// is one v_fma_f64 and two v_xor_b32 (bias removal), against ~10 instructions before.
//
// Workgroup = 8 waves (two per SIMD, 256 registers: up to 11 int32 accumulator tiles of 32 x 32 per wave) x 64
// replicates x one group of 32 observable columns.  (One wave per SIMD with 21 tiles was built first: an in-order wave
// alone adds its LDS-store stalls, its vector work and its MFMAs up -- 2350 cycles per k-step against 840 of matrix
// work; two waves per SIMD fill each other's stalls.)
//   fragment f = (row set rs, column quad cq): rs = power J0 + rs of the launch; 32 tile columns = columns 4 cq ..
//   4 cq + 3 x digit slots 0..7.  u-row fragments (dx = 1): tile column = (monomial, digit slot), 4 monomials each.
//   wave w owns column quad w of every row set (both replicate halves) from the global load to the accumulators;
//   2 ceil(JN / 4) of the waves also one u-row tile each.  Narrow states (C <= 16): the waves that share a column quad
//   split the powers (template parameter NQ, see the kernel).
//   (A first cut shared one X table between the waves with a barrier per k-step: every wave was in the same phase at
//   the same time and the k-step behaved like the SUM of its LDS, vector and matrix time.)
// Count tile: cnt[word g = sample / 4][replicate], one u32 = the u8 counts of 4 samples.  Stage 3 of the sampler runs
// with ONE LANE PER REPLICATE (lane = replicate, the eight waves split the Philox calls): all 64 lanes of a ds_add hit
// 64 consecutive words -- no bank conflict by construction (a [replicate][word] layout lost 11 cycles per ds_add).
// Partial sums: one slot per SCALING WINDOW (a fixed block of samples: the window size depends on N only),
// part[window][replicate][power][column] -- an element's seven digit sums added up in the flush, in the fixed order the
// finalize kernel used to apply to per-digit slots part[window][replicate][power][digit slot][column] (round 6; calls that
// carry a second matrix still store those) -- stored once -- no read-modify-write, no zeroing -- and added up by the finalize
// kernel in window order.  A replicate's result therefore does not depend on how many replicates,
// chunks or workgroups the launch had: rows [a, b) of a bootstrap equal the (b - a)-replicate call with rep0 = a
// bit for bit (multi-GPU slabs, txm_sampler_spec.rep0).
#include "txm_i8t_common.h"

namespace txm {

// staged factor tiles: entry e (sample e of the 1024 the tile's k-steps slice) -> line of the tile.  A lane reads the
// factors of its two samples (e and e + 16).  The second unit's lines sit 513 lines further on -- out of the reach of a
// ds_read2_b64 pair and not a multiple of 64 lines (ds_read2st64_b64 would pair them again): two ds_read_b64 (2 LDS
// cycles each) instead of one paired read (8).  Same-box A/B, round 4: 36.4 -> 35.8 ms at N = 2e7, 165.8 -> 163.2 ms at
// the north star (-1.6 %); TXM_T_PAIRREAD restores the adjacent layout.
// (PU = the unit stride in lines: 513, or 545 / 577 where a tile stages 33 / 35 chunks -- chunk groups, see the kernel)
#ifndef TXM_T_PAIRREAD
#define T_PIDX(e) (((((e) >> 4) & 1) * T_PUNIT) + (((e) >> 5) * 16) + ((e) & 15))
#define T_PUNIT_OF(cg) ((cg) > 2 ? 577 : (cg) > 1 ? 545 : 513)
#define T_PLINES_OF(cg) (T_PUNIT_OF(cg) + (31 + (cg)) * 16)
#else
#define T_PIDX(e) (e)
#define T_PUNIT_OF(cg) 16
#define T_PLINES_OF(cg) (SM_T + 32 * ((cg) - 1))
#endif
// chunk groups of a launch: 2 for one-quad states, for two-quad states with at most six powers and for four-quad states with at
// most four (see the kernel; beyond that the row sets of a wave and the staged tiles of 33 chunks do not fit the LDS -- with four
// quads a wave of a group holds every power of its quad), else 1.  One-quad states with at most four powers (orders 1-3; BASELINE
// config 5's states): FOUR groups of two waves (round 6) -- a wave then holds two row sets and walks 8 k-steps per tile instead of
// one row set over 16: the same work per wave in half as many latency chains, each with two independent row sets in flight
// (five and more powers need four u-row waves per group, and from seven on the staged tiles do not fit next to the regions)
#ifndef TXM_T_CG1
#define TXM_T_CG1 4
#endif
#ifdef TXM_T_FLUSH_SERIAL  // (A/B build: the chunk groups' first flush, see the end of the window loop)
constexpr bool T_FLUSH_SERIAL = true;
#else
constexpr bool T_FLUSH_SERIAL = false;
#endif
#define T_CG_OF(nq, jn) (((nq) == 1 && (jn) <= 4) ? TXM_T_CG1 : ((nq) == 1 || ((nq) == 2 && (jn) <= 6) || ((nq) == 4 && (jn) <= 4)) ? 2 : 1)

// K = order + 1 is a run-time argument (it only enters the flush addresses); one launch slices the JN powers
// J0 .. J0 + JN - 1.
//
// Wave w owns COLUMN QUAD w (columns 4 w .. 4 w + 3) from the global load to the accumulators: it loads those four
// columns of every sample, slices them, writes the fixed-point words into ITS OWN LDS region and reads them back
// transposed.  Nothing but the count tile (and the staged u tile) is shared between waves, so the k-steps need no
// barrier: the eight waves drift apart and one wave's LDS / vector phases fall into another's matrix phases.
//   region of a wave: per power two planes [sample 32][column 4] of 4-byte halves (low dword / high dword of the
//   8-byte word), the planes 640 bytes apart (128 bytes of padding: the two 16-lane groups of a half-wave read the
//   two planes at once, on disjoint banks).  A lane slices (sample l >> 2, column l & 3) of a 16-sample unit, so a
//   wave's 64 low dwords are 256 contiguous bytes: ds_write_addtid_b32 (no address register, 128 B/clk -- twice the
//   rate of ds_write_b128).  The transposing read of a plane hands lane i of a 16-lane group byte column i =
//   (column i >> 2, digit i & 3): tile column n = 16 plane + 4 column + digit-in-plane.
//   A region is single-buffered: within a wave LDS operations execute in order, and the words of chunk s + 1 of a power
//   are written after the MFMAs of chunk s of that power have taken their operands.
//   (Measured alternative, same box: lane = (sample, column PAIR), one ds_write_b128 per power and a plain
//   [sample][column][8 B] region -- 17 % fewer instructions per k-step, 10 % MORE time: 40.4 vs 36.8 ms at N = 2e7.  The
//   store path of the wide writes costs more than the issue slots they save.)
// (T_PLANE, T_PB: txm_i8t_common.h)
// YS: the launch carries one more row set, the order-0 monomial w * dy of a SECOND sample matrix y (I8Args::y: the
// volume callback's dx/dq, txm_resample_opts.y) -- its per-replicate sums ride on the same count tile and k-steps.
//
// NQ < 8: a NARROW state (C <= 4 NQ observables, NQ = 1, 2 or 4 column quads).  The 8 / NQ waves that share a column
// quad split the powers between them: wave w owns quad w % NQ and the powers g, g + GS, g + 2 GS ... (g = w / NQ,
// GS = 8 / NQ) -- ceil(JN / GS) row sets per wave instead of JN, every MFMA column in use, all orders in one pass.
// The u-row tiles go to the last waves (the ones with the fewest power rows).
// BATCHED: state blockIdx.y of a batched launch.  A template parameter and not a run-time test: a first version patched a local
// copy of the arguments from the I8State table behind `if (a.states)`, and that copy cost the SINGLE-state narrow kernel 18 %
// (config 2: 3.70 vs 3.09 ms per step, same box, gpurun_out/r4_c2_ab3.log) -- its fields live in scalar registers the kernel
// does not have, where kernel arguments are re-loaded on demand.  The batched instances read a per-state argument block instead.
// PD (wide instance only): per-digit slots although the launch has no second matrix -- the FIRST pass of a two-pass call that carries one
// in its last pass (the call's finalize reads one layout).
template <int J0, int JN, bool WEIGHTED, bool YS = false, int NQ = 8, bool BATCHED = false, bool PD = false>
__global__ __launch_bounds__(T_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void resample_i8t_kernel(const I8Args a_in, const int K) {
  static_assert(NQ == 8 || NQ == 4 || NQ == 2 || NQ == 1, "column quads");
  static_assert(!BATCHED || NQ < 8, "batched launches: narrow states");
  // batched launch (narrow states only): state blockIdx.y runs on ITS copy of the arguments in device memory
  // (I8Args::batch_args: uniform address, read like kernel arguments -- on demand, by scalar loads)
  // (the constant address space: nothing writes the block while the kernel runs, and that is what lets the compiler take the
  // fields with scalar loads; as plain global memory they became per-lane vector loads: 9.3 instead of 8.1 ms at config 5)
  typedef const __attribute__((address_space(4))) I8Args *const_args_p;
  auto pick_args = [&]() -> decltype(auto) {
    if constexpr (BATCHED) return (*(const_args_p)(uintptr_t)(a_in.batch_args + blockIdx.y));
    else return (a_in);
  };
  const auto &a = pick_args();
  // CG = 2 (narrow states): a narrow k-step is a latency chain per chunk, not work -- a wave has one or two row sets, and with
  // one column quad and four powers (BASELINE config 5's states) four of the eight waves had none at all.  The waves form two
  // CHUNK GROUPS instead: group c = wave / 4 contracts the chunks 16 c .. 16 c + 15 of every tile (its own stream of 512-sample
  // half tiles: the same pipeline from base + 512 c samples), its four waves share the quads and the powers as the eight did --
  // 16 k-steps per tile and wave with twice the row sets instead of 32.  The two groups' int32 accumulators are added through the
  // idle count tile at the flush (exact, so the sums are those of one group).
  constexpr int CG = T_CG_OF(NQ, JN);
  constexpr int T_PUNIT = T_PUNIT_OF(CG);
  constexpr int STEPS = T_STEPS / CG;            // k-steps (chunks) of a wave per tile
  constexpr int WPG = T_WAVES / CG;              // waves of a chunk group
  static_assert(CG == 1 || CG == 2 || CG == 4, "chunk groups");
  constexpr int GS = 8 / NQ / CG;             // waves per column quad and chunk group = stride of a wave's powers
  constexpr int NSW = (JN + GS - 1) / GS;     // power row sets per wave
  static_assert(JN >= 1 && NSW + (YS ? 1 : 0) <= 5 && J0 + JN <= 8, "power range");
  static_assert(NQ == 8 || !YS, "narrow states: no second matrix");
  constexpr int NS = NSW + (YS ? 1 : 0);  // row sets of the launch = x fragments per wave
  constexpr int NPT = JN + ((YS && WEIGHTED && J0 > 0) ? 1 : 0);  // staged factor tiles (the y row set needs plain w)
  constexpr int UF = (JN + 3) / 4;   // u-row fragments (4 monomials each)
  constexpr int WREG = (NS + 1) * T_PB;  // a wave's region: NS powers + one u-row fragment
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // the X regions come first: ds_write_addtid takes its base from M0[15:0]
  static_assert(T_WAVES * WREG <= 65536, "X regions within the first 64 KiB");
  uint32_t *cntw = reinterpret_cast<uint32_t *>(lds + T_WAVES * WREG);
  uint32_t *fsum = cntw + T_CNT_BYTES / 4;  // [64] draws per replicate in the window
  uint32_t *cnt_a = fsum + I8_REPS;         // [64] tile draw counts, double buffered
  uint32_t *cnt_b = cnt_a + I8_REPS;
  // the sample factors w du^(J0 + jj), jj < JN, of the 1024 samples whose X words this tile's k-steps produce (chunks
  // 1 .. 31 of the tile and chunk 0 of the next one): staged once per tile by the whole workgroup (JN x 8 KiB), so that
  // a k-step loads nothing but x and spends one v_fma_f64 + two v_xor_b32 per word -- the powers are LDS reads
  // (broadcast: four lanes per sample), not vector multiplies.  Entry-major: the factors of a lane's two samples for one
  // row set are one ds_read2_b64 off a common base (power-major they were two reads and two address adds per row set)
  double *ptile = reinterpret_cast<double *>(cnt_b + I8_REPS);  // [1024 entries][NPT factors]: a lane's factors of all row sets in one line

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int n32 = lane & 31, half = lane >> 5;
  const uint32_t wreg = (uint32_t)(wave * WREG);  // LDS byte offset of this wave's region (uniform)

  // ---- producer role: lane = (sample l >> 2 of a 16-sample unit, column l & 3 of the wave's quad)
  const int ps = lane >> 2, cl = lane & 3;
  const int quad = NQ == 8 ? wave : wave % NQ;  // the wave's column quad
  const int g = NQ == 8 ? 0 : (wave / NQ) % GS;  // ... and its first power (relative to J0)
  const int cgrp = CG == 1 ? 0 : wave / (T_WAVES / CG);  // chunk group
  const int coff = cgrp * STEPS;                         // its first chunk of a tile
  const int col = 4 * quad + cl;
  // row set fi of this wave = power J0 + g + fi GS; past the launch's powers: an idle row set (wave-uniform)
  auto row_live = [&](int fi) { return NQ == 8 || g + fi * GS < JN; };
  const int ccol = col < a.C ? col : 0;  // columns >= C re-read column 0: their sums are never flushed
  const uint32_t xo = (uint32_t)((ps * a.ldx_s + ccol) * 8);  // byte offset from the unit's (uniform) row base
  const uint32_t yo = YS ? (uint32_t)((ps * a.ldy_s + ccol) * 8) : 0u;
  // ---- consumer role
  // transposing read of (plane g, rows 16 half + 0..7): lane 2 q + p of the 16-lane group supplies row q, bytes 8 p ..
  const uint32_t rd_off = wreg + (uint32_t)(((lane >> 4) & 1) * (T_PLANE + 128) + (16 * half + ((lane & 15) >> 1)) * 16 +
                                            (lane & 1) * 8);  // + power * T_PB; second read + 128
  // tile column n32 -> (column, digit slot)
  const int tcl = (n32 >> 2) & 3, tdg = 4 * (n32 >> 4) + (n32 & 3);
  // u-row tile of waves 0 .. 2 UF - 1: fragment fu (monomials 4 fu .. 4 fu + 3), replicate half uh
  constexpr int NUT = 2 * UF;
  // (narrow states: the last NUT waves -- of every chunk group)
  const int uw = NQ == 8 ? wave : (wave % (T_WAVES / CG)) - (T_WAVES / CG - NUT);
  const bool has_ut = uw >= 0 && uw < NUT;  // wave-uniform
  const int fu = has_ut ? (uw >> 1) : 0, uh = uw & 1;
  // A wave with a u-row tile for replicate half 1 takes the halves in swapped order (operand A0 = ITS u-row half, flushed
  // as such): the u-row MFMA then needs no per-step selection between the two count operands (four v_cndmask)
  const int hswap = (has_ut && uh) ? 1 : 0;  // wave-uniform
  const uint32_t a_off = (uint32_t)((4 * half) * I8_REPS + n32 + 32 * hswap);  // words of operand A0; + 8 s * 64 + q * 64
  const uint32_t a_off1 = hswap ? (uint32_t)(-32 * 4) : (uint32_t)(32 * 4);   // bytes from A0's words to A1's
  const int um = 4 * fu + cl;      // this lane's u-row monomial
  const int umc = um < JN ? um : 0;

  const int b = blockIdx.x;
  const int xcd = b & 7, qq = b >> 3;
  const int chunk = (qq / a.n_rbg) * 8 + xcd;
  const int rbg = qq % a.n_rbg;
  const int64_t rep0 = (int64_t)rbg * I8_REPS;
  const int64_t t_begin = (int64_t)chunk * a.tiles_per_chunk;
  int64_t t_end = t_begin + a.tiles_per_chunk;
  if (t_end > a.ntiles) t_end = a.ntiles;

  const double pu = a.pivot[0];
  const double px = a.pivot[1 + a.col0 + ccol];
  const double py = YS ? a.ypivot[1 + a.col0 + ccol] : 0.0;

  v16i acc[NS][2];
  v16i accu;
#pragma unroll
  for (int e = 0; e < NS; ++e) acc[e][0] = acc[e][1] = (v16i)(0);
  accu = (v16i)(0);

  // zero the regions once (the padding between the planes is never written)
  for (int e = threadIdx.x; e < T_WAVES * WREG / 16; e += T_BLOCK) reinterpret_cast<uint4 *>(lds)[e] = make_uint4(0, 0, 0, 0);

  // stage-3 role: lane = replicate
  // (static priority for waves 4..7 was measured: +3 % time; nothing to arbitrate without a barrier per k-step)
  const int64_t my_rep = rep0 + lane;
  const bool rep_live = my_rep < a.nrep;
  const uint32_t rstream = a.rep_base + (uint32_t)my_rep;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  uint32_t fdraws = 0;

  struct XIn {
    double x[2];             // the two 16-sample units of a chunk
    double y[YS ? 2 : 1];    // ... of the second matrix
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
    const double sc = wt[I8_WT_SC + ccol];
    const double *wty = YS ? a.ywtab + win * I8_WT_STRIDE : wt;
    const double scy = YS ? wty[I8_WT_SC + ccol] : 0.0;
    int64_t tt_end = (win + 1) * WT;
    if (tt_end > t_end) tt_end = t_end;

    // ---- store the fixed-point words of the wave's two units of one power: low dwords and high dwords as four
    // 256-byte runs (M0 = the wave's region)
    auto store_x2 = [&](uint64_t bits0, uint64_t bits1, int off) {
      const uint32_t lo0 = (uint32_t)bits0 ^ 0x80808080u, hi0 = (uint32_t)(bits0 >> 32) ^ 0x00008080u;
      const uint32_t lo1 = (uint32_t)bits1 ^ 0x80808080u, hi1 = (uint32_t)(bits1 >> 32) ^ 0x00008080u;
#ifdef TXM_T_NO_WRITE  // ablation build: the values stay live, nothing is stored
      asm volatile("" ::"v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1));
#else
      // (s_nop: one wait state between an SALU write of M0 and an add-TID LDS instruction.  M0 as a tracked "{m0}" operand,
      // set once per window, saves the twelve scalar instructions per k-step and buys nothing: 37.2 vs 36.9 ms, same box --
      // the scalar unit is not what the k-steps wait for)
      asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "ds_write_addtid_b32 %0 offset:%5\n\t"
                   "ds_write_addtid_b32 %1 offset:%6\n\t"
                   "ds_write_addtid_b32 %2 offset:%7\n\t"
                   "ds_write_addtid_b32 %3 offset:%8"
                   :
                   : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "s"(wreg), "n"(off), "n"(off + T_PLANE + 128), "n"(off + 256),
                     "n"(off + 256 + T_PLANE + 128)
                   : "memory", "m0");  // (M0 is written: the compiler must not keep a value of its own there)
#endif
    };

    // ---- one k-step of a wave: chunk s of the tile on the matrix pipe, the X words of chunk s + 1 sliced in between.
    // e0 = entry of the produced chunk's first sample in the staged tiles (e0 < 0: the direct path of a window's first
    // chunk -- du / w of this lane's two samples come in d_du / d_w)
    auto kstep = [&](auto produce_c, auto consume_c, int s, const XIn &R, int e0, const double (&d_du)[2],
                     const double (&d_w)[2]) {
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
        // the second replicate half through a base of its own (opaque): the load merger then pairs the words of ONE
        // half -- ds_read2_b32 (q, q + 1) lands in adjacent registers of the operand -- instead of (half 0, half 1) of one
        // word, which took eight v_mov per k-step to sort into the two operands
        uint32_t off1 = a_off1;
        asm volatile("" : "+v"(off1));
        const uint32_t *cw1 = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(cw) + off1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          A0[q] = (int)cw[q * I8_REPS];
          A1[q] = (int)cw1[q * I8_REPS];
        }
        Ba = T_TRREAD((lds_v2i)(lds + rd_off));
        Bb = T_TRREAD((lds_v2i)(lds + rd_off + 128));
      }
      double dx[2] = {0.0, 0.0}, dy[2] = {0.0, 0.0};
      const bool staged = e0 >= 0;  // uniform
      if constexpr (produce) {
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          dx[uu] = (R.x[uu] - px) * sc;
          if constexpr (YS) dy[uu] = (R.y[uu] - py) * scy;
        }
      }
      // factor jj of unit uu: a broadcast LDS read (staged) or du / w of the direct path multiplied up
      // (the base is an opaque byte offset: the staged tiles sit above 64 KiB, and with the constant folded in every read
      // got a v_add_u32 of its own for "lane part + 0x1fb00"; an opaque base takes the 16-bit immediate offsets)
      const uint32_t pt_lds = (uint32_t)(reinterpret_cast<const unsigned char *>(ptile) - lds);  // compile-time constant
      uint32_t pt_b = pt_lds + (uint32_t)((T_PIDX((staged ? e0 : 0) + ps) * NPT + g) * 8);
      uint32_t pu_b = pt_lds + (uint32_t)((T_PIDX((staged ? e0 : 0) + ps) * NPT + umc) * 8);
      asm volatile("" : "+v"(pt_b), "+v"(pu_b));
      typedef __attribute__((address_space(3))) const double *lds_cd;
      auto pt_at = [&](uint32_t base, int idx) { return ((lds_cd)(lds + base))[idx]; };  // (cast first: the index is then a 32-bit LDS offset)
      auto factor = [&](int fi, int uu) {  // row set fi = power J0 + g + fi GS
        if (NQ == 8 && !WEIGHTED && J0 == 0 && fi == 0) return 1.0;
        if (staged) return pt_at(pt_b, fi * GS + T_PUNIT * NPT * uu);
        // (narrow states: the 1.0 of an unweighted power 0 is kept OPAQUE on this direct path of a window's first chunk.  Where a
        // wave's first power is a compile-time 0 -- four quads in two chunk groups: one wave per quad and group -- the compiler
        // otherwise folds fma(1, dx, magic) into dx + magic and contracts that with dx's own multiply: ONE rounding for these 32
        // samples where every staged chunk has two, and the words then depend on which chunk a chunk group starts a window with.
        // With it the direct path is the staged arithmetic, and resample_i8gn_kernel -- one chunk group -- agrees bit for bit)
        double pw = WEIGHTED ? d_w[uu] : 1.0;
        if constexpr (NQ < 8 && !WEIGHTED) asm volatile("" : "+v"(pw));
        for (int q = 0; q < J0 + g + fi * GS; ++q) pw *= d_du[uu];
        return pw;
      };
      // the y row set's factor: the plain weight (order 0)
      auto factor_y = [&](int uu) {
        if (!WEIGHTED) return 1.0;
        if (!staged) return d_w[uu];
        return J0 == 0 ? pt_at(pt_b, T_PUNIT * NPT * uu) : pt_at(pt_b, JN + T_PUNIT * NPT * uu);  // tile 0 is w du^0 when J0 == 0, else the extra tile (NQ = 8: g = 0)
      };
      t_static_for<NS>([&](auto fic) {
        constexpr int fi = decltype(fic)::value;
        v2i Na = (v2i)(0), Nb = (v2i)(0);
        if constexpr (consume) {
          if constexpr (fi + 1 < NS) {
            if (row_live(fi + 1)) {  // wave-uniform (always for NQ = 8)
              Na = T_TRREAD((lds_v2i)(lds + rd_off + (fi + 1) * T_PB));
              Nb = T_TRREAD((lds_v2i)(lds + rd_off + (fi + 1) * T_PB + 128));
            }
          }
        }
        if (row_live(fi)) {
          if constexpr (consume) {
            const v4i B = {Ba[0], Ba[1], Bb[0], Bb[1]};
            t_mfma<true>(acc[fi][0], A0, B);
            t_mfma<true>(acc[fi][1], A1, B);
          }
          if constexpr (produce) {
            // the words of chunk s + 1, row set fi: behind the MFMAs that took chunk s's (the region is single-buffered)
            if constexpr (YS && fi == JN)
              store_x2((uint64_t)__double_as_longlong(fma(factor_y(0), dy[0], T_MAGIC)),
                       (uint64_t)__double_as_longlong(fma(factor_y(1), dy[1], T_MAGIC)), fi * T_PB);
            else
              store_x2((uint64_t)__double_as_longlong(fma(factor(fi, 0), dx[0], T_MAGIC)),
                       (uint64_t)__double_as_longlong(fma(factor(fi, 1), dx[1], T_MAGIC)), fi * T_PB);
          }
        }
        Ba = Na;
        Bb = Nb;
      });
      if (has_ut) {  // wave-uniform: the u-row tile
        if constexpr (consume) {
          Ba = T_TRREAD((lds_v2i)(lds + rd_off + NS * T_PB));
          Bb = T_TRREAD((lds_v2i)(lds + rd_off + NS * T_PB + 128));
          const v4i B = {Ba[0], Ba[1], Bb[0], Bb[1]};
          t_mfma<false>(accu, A0, B);  // A0 = the wave's u-row half (hswap)
        }
        if constexpr (produce) {
          // monomial um of this lane's sample (dx = 1); the unused monomial slots of a short fragment repeat monomial 0:
          // their sums are never flushed
          double pw[2];
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) {
            if (staged) pw[uu] = pt_at(pu_b, T_PUNIT * NPT * uu);
            else {
              pw[uu] = WEIGHTED ? d_w[uu] : 1.0;
              for (int q = 0; q < J0 + umc; ++q) pw[uu] *= d_du[uu];
            }
          }
          // (ldexp + add: exact, and no second 64-bit literal for the register allocator to park in scratch)
          store_x2((uint64_t)__double_as_longlong(__builtin_ldexp(pw[0], 50) + T_MAGIC),
                   (uint64_t)__double_as_longlong(__builtin_ldexp(pw[1], 50) + T_MAGIC), NS * T_PB);
        }
      }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};

    // ---- flush: int32 accumulators of one window -> its slot of the partial sums (stored, never re-read here)
    // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
    auto flush_tile = [&](v16i &T, int h, int rs, int ufrag) {
      // `opq` is an opaque zero re-created per tile: the addresses below are then computed where they are used; left
      // to itself the compiler hoists them all out of the window loop and spills them
      uint32_t z = 0;
      asm volatile("" : "+v"(z));
      const int64_t opq = (int64_t)z;
      bool valid = tdg < I8_NSL;
      int j;
      double dsc;
      double *base;
      size_t stride = (size_t)K * (ufrag < 0 ? a.cpad : 1) * 8;  // one replicate
      const int64_t cpad = a.cpad;  // columns of a partial-sum row: 32, or 4 NQ for a narrow state
      if (ufrag < 0 && YS && rs == JN) {
        // the second matrix: [window][replicate][digit slot][column], scale = max|w| x its own column scale
        const int c = 4 * quad + tcl;
        valid = valid && c < a.C;
        j = 0;
        dsc = wty[I8_WT_DSP + 0] * wty[I8_WT_DSC + (c < a.C ? c : 0)];
        base = a.part_y + (((size_t)win * a.nrep_pad + rep0 + 32 * h + 4 * half) * 8 + tdg) * cpad + c + opq;
        stride = (size_t)8 * cpad;
      } else if (ufrag < 0) {
        const int c = 4 * quad + tcl;
        valid = valid && c < a.C && row_live(rs);
        j = J0 + (row_live(rs) ? g + rs * GS : 0);
        dsc = wt[I8_WT_DSP + j] * wt[I8_WT_DSC + (c < a.C ? c : 0)];
        // [window][replicate][power][digit slot][column]
        base = a.part_x + ((((size_t)win * a.nrep_pad + rep0 + 32 * h + 4 * half) * K + j) * 8 + tdg) * cpad + c + opq;
      } else {
        const int m = 4 * ufrag + tcl;
        valid = valid && m < JN;
        j = J0 + (m < JN ? m : 0);
        dsc = wt[I8_WT_DSP + j] * 0x1p-50;
        base = a.part_u + (((size_t)win * a.nrep_pad + rep0 + 32 * h + 4 * half) * K + j) * 8 + tdg + opq;
      }
      dsc *= (double)((int64_t)1 << (8 * (tdg < I8_NSL ? tdg : 0)));
      const int bias = tdg == I8_NSL - 1 ? T_D6_BIAS : 0;
#ifdef TXM_T_NO_FLUSH_STORE  // (ablation build: one row of a tile is written)
      if (valid && tdg == 0) {
#else
      if (valid) {
#endif
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
    // (one-quad states: depth 4 -- config 5's batched launch 6.76 -> 6.57 ms, same box, gpurun_out/r6_exp1.log; two quads: 8 and 4 tie)
    constexpr int XD = NQ == 8 ? T_XD : NQ == 1 ? T_XDN1 : T_XDN, UNR = XD > 4 ? XD : 4;
    XIn XR[XD];  // the wave's columns of a chunk, requested XD k-steps ahead: chunk c lives in slot c % XD
    // The rows of the next chunk to request, as RUNNING wave-uniform pointers: chunks follow each other in memory inside a
    // tile and from tile to tile (all but the slid last tile of the series), so a request is two scalar adds -- the row
    // base "(tile base + 32 c) * pitch + column" from scratch was a chain of some fifteen dependent scalar instructions
    // (64-bit multiplies) at the head of every k-step of an in-order wave.
    const char *xq = nullptr, *yq = nullptr;
    int64_t xq_step = 0, yq_step = 0;
    const int64_t xq_u1 = 16 * a.ldx_s * 8, yq_u1 = YS ? 16 * a.ldy_s * 8 : 0;  // second 16-sample unit of a chunk
    auto load_q = [&](XIn &r) {
#ifdef TXM_T_NO_LOAD
      r.x[0] = px + 1e-3; r.x[1] = px - 1e-3;
      if constexpr (YS) { r.y[0] = py + 1e-3; r.y[1] = py - 1e-3; }
#else
      r.x[0] = *reinterpret_cast<const double *>(xq + xo);
      r.x[1] = *reinterpret_cast<const double *>(xq + xq_u1 + xo);
      if constexpr (YS) {
        r.y[0] = *reinterpret_cast<const double *>(yq + yo);
        r.y[1] = *reinterpret_cast<const double *>(yq + yq_u1 + yo);
      }
#endif
    };
    auto set_q = [&](int64_t i0) {  // the running pointers at sample i0 (a chunk's first)
      xq = reinterpret_cast<const char *>(a.x + i0 * a.ldx_s + a.col0);
      if constexpr (YS) yq = reinterpret_cast<const char *>(a.y + i0 * a.ldy_s + a.col0);
    };
    auto step_q = [&]() {
      xq += xq_step;
      if constexpr (YS) yq += yq_step;
    };
    const double no_d[2] = {0.0, 0.0}, no_w[2] = {1.0, 1.0};

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
        set_q(wbase + 32 * coff);  // (chunk groups: this group's first chunk of the tile)
        xq_step = 32 * a.ldx_s * 8;
        if constexpr (YS) yq_step = 32 * a.ldy_s * 8;
        load_q(XR[0]);
        step_q();
      }
      // staging requests (in flight during the zeroing and the fill): entries 0 .. 991 = samples wbase + 32 ...,
      // entries 992 .. 1023 = the next tile's first chunk
      // (chunk groups: entries 1024 .. 1055 = the next tile's chunk 16, which the second group slices at its last k-step)
      constexpr int NSTG = CG > 1 ? 3 : 2;
      static_assert(SM_T + 32 * (CG - 1) <= NSTG * T_BLOCK, "staging rounds");
      double su[NSTG], sw[NSTG];
#pragma unroll
      for (int q = 0; q < NSTG; ++q) {
        const int e = (int)threadIdx.x + q * T_BLOCK;
        // (entries past 1024: the next tile's first chunk of chunk group 1, 2, ... -- chunk STEPS, 2 STEPS, ... of that tile)
        const int64_t i = e < SM_T - 32 ? wbase + 32 + e : e < SM_T ? wnext + (e - (SM_T - 32))
                                                         : wnext + 32 * STEPS * (1 + (((e - SM_T) >> 5) % (CG > 1 ? CG - 1 : 1))) + ((e - SM_T) & 31);
        su[q] = a.u[i];
        sw[q] = 1.0;
        if constexpr (WEIGHTED) sw[q] = a.w[i];
      }
      uint32_t ncnt = 0;
      if (wave == 0 && has_next && rep_live) ncnt = a.counts[(size_t)my_rep * a.ntiles + t + 1];
      // (the last k-step of the previous tile was the count tile's and the staged tiles' last reader: barrier below)
      __syncthreads();
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
        // live replicates of this group (re-derived per tile: two scalar registers fewer across the k-steps); <= 32 -- a short
        // last group -- and the fill packs several calls of a replicate into one wave instruction
        // (narrow-state instances only: in the wide ones the extra code costs the register allocator 6 more spilled
        // registers and the north star 0.5 %, same-box A/B -- gpurun_out/r4_pack_ab2.log)
        const int64_t live_reps = a.nrep - rep0;
        const int fill_pack = NQ == 8 ? 1 : live_reps <= 8 ? 8 : live_reps <= 16 ? 4 : live_reps <= 32 ? 2 : 1;  // wave-uniform
        if (tsize == (uint32_t)SM_T && fill_pack > 1) {
          // a short LAST replicate group (live <= 32 of its 64 lanes: 8 of 64 at nrep = 200): FP = fill_pack lanes share a
          // replicate and take FP consecutive Philox calls of it -- 1 / FP of the wave instructions of the lane-per-replicate
          // fill below for the same draws (which lane runs a call of the stream is free; the count words are atomics)
          const int rp = lane & (I8_REPS / fill_pack - 1), slot = lane / (I8_REPS / fill_pack);
          const bool plive = rep0 + rp < a.nrep;
          const uint32_t np = plive ? cnt_cur[rp] : 0u;
          uint32_t nmx = np;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const uint32_t hi = (uint32_t)__shfl_xor((int)nmx, o);
            nmx = hi > nmx ? hi : nmx;
          }
          nmx = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmx);
          const uint32_t rsp = a.rep_base + (uint32_t)(rep0 + rp);
#pragma unroll 1
          for (uint32_t c0 = (uint32_t)wave * (uint32_t)fill_pack; c0 * 12u < nmx; c0 += (uint32_t)(T_WAVES * fill_pack))
            t_fill_call<false>(cntw, a.k0, a.k1, rsp, (uint32_t)t, c0 + (uint32_t)slot, np, (uint32_t)rp * 4u);
        } else if (tsize == (uint32_t)SM_T) {
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
        // the X words of chunk 0 (no matrix work yet; its u / w straight from memory), chunks 1 .. XD requested
        double d_du[2], d_w[2] = {1.0, 1.0};
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          d_du[uu] = (a.u[wbase + 32 * coff + 16 * uu + ps] - pu) * inv_du;
          if constexpr (WEIGHTED) d_w[uu] = a.w[wbase + 32 * coff + 16 * uu + ps] * inv_w;
        }
        kstep(YES, NO, 0, XR[0], -1, d_du, d_w);
#pragma unroll
        for (int c = 1; c <= XD; ++c) {
          load_q(XR[c % XD]);
          step_q();
        }
        first_tile = false;
      }
      // park the staged tiles
#pragma unroll
      for (int q = 0; q < NSTG; ++q) {
        const int e = (int)threadIdx.x + q * T_BLOCK;
        if (q == 2 && e >= SM_T + 32 * (CG - 1)) continue;  // (only 32 entries per further chunk group exist in the third round)
        const double du = (su[q] - pu) * inv_du;
        double pw = WEIGHTED ? sw[q] * inv_w : 1.0;
        if constexpr (NPT > JN) ptile[T_PIDX(e) * NPT + JN] = pw;  // plain w for the y row set
#pragma unroll
        for (int k = 0; k < J0; ++k) pw *= du;
#pragma unroll
        for (int jj = 0; jj < JN; ++jj) {
          ptile[T_PIDX(e) * NPT + jj] = pw;
          pw *= du;
        }
      }
      if (wave == 0) cnt_nxt[lane] = ncnt;
      T_TICK(3);
      __syncthreads();
      T_TICK(4);

      // ---- 32 k-steps, four per trip, NO barrier between them.  Step s contracts chunk s, slices chunk s + 1 from
      // the ring slot (s + 1) % XD (x requested XD steps ago) and requests chunk s + 1 + XD into that slot.
      // Chunk 32 is the next tile's chunk 0 (words nobody reads when there is no next tile).
#pragma unroll 1
      for (int s = 0; s < STEPS; s += UNR) {
#pragma unroll
        for (int e = 0; e < UNR; ++e) {
          const int sq = s + e;   // k-step of this wave; its chunk of the tile: coff + sq
          XIn &R = XR[(e + 1) % XD];
          const XIn cur_x = R;
          load_q(R);  // chunk sq + 1 + XD of the wave's stream
          // the request after the group's last chunk is the next tile's first chunk of the group (not adjacent when that tile
          // is the slid last one); no next tile: the last chunk again, words nobody reads
          if ((STEPS - 2 - XD - e) % UNR == 0 && s == STEPS - 2 - XD - e) {
            if (has_next) set_q(wnext + 32 * coff);
            else { xq_step = 0; yq_step = 0; }
          } else step_q();
          // entry of the sliced chunk in the staged tiles: the next chunk of the tile, or -- last k-step -- the next tile's
          // first chunk of the group (entries 992 .. for chunk 0, 1024 .. for chunk 16)
          const int e0 = (CG > 1 && sq == STEPS - 1) ? (cgrp ? SM_T + 32 * (cgrp - 1) : SM_T - 32) : (coff + sq) * 32;
          kstep(YES, YES, coff + sq, cur_x, e0, no_d, no_w);
          T_TICK(5);
        }
      }
      T_TICK(6);
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
    constexpr int XS = 65;        // words per register row of a tile in the LDS (64 lanes + 1: see the reads of flush_summed)
    constexpr int XT = 16 * XS;   // words per tile
    // element (replicate row m, column / monomial q4) of a summed tile: its digit d sits in register 4 (m >> 3) + (m & 3) of
    // lane 16 (d >> 2) + 4 q4 + (d & 3) + 32 ((m >> 2) & 1) (the D layout of v_mfma_i32_32x32x32_i8; tile column -> (column,
    // digit) as tcl / tdg above).  Lane L takes the elements L and L + 64 of the tile's 32 x 4.  With rows of 65 words the
    // lanes of one read spread over rr + 4 q4 (+ const): four-way bank conflicts; rows of 64 put all sixteen replicate rows of
    // a lane quartet on one bank.
    auto flush_summed = [&](const uint32_t *tl, int h, int rs, int ufrag) {
      uint32_t z = 0;
      asm volatile("" : "+v"(z));  // (addresses formed here, not hoisted out of the window loop and spilled: see flush_tile)
      const int64_t opq = (int64_t)z;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = lane + 64 * i, m = o >> 2, q4 = o & 3;
        const int rr = ((m >> 3) << 2) | (m & 3), hh = (m >> 2) & 1;
        bool valid;
        int j;
        double dsc0;
        double *dst;
        if (ufrag < 0) {
          const int c = 4 * quad + q4;
          valid = c < a.C && row_live(rs);
          j = J0 + (row_live(rs) ? g + rs * GS : 0);
          dsc0 = wt[I8_WT_DSP + j] * wt[I8_WT_DSC + (c < a.C ? c : 0)];
          dst = a.part_x + (((size_t)win * a.nrep_pad + rep0 + 32 * h + m) * K + j) * a.cpad + c + opq;
        } else {
          const int mm = 4 * ufrag + q4;
          valid = mm < JN;
          j = J0 + (mm < JN ? mm : 0);
          dsc0 = wt[I8_WT_DSP + j] * 0x1p-50;
          dst = a.part_u + ((size_t)win * a.nrep_pad + rep0 + 32 * h + m) * K + j + opq;
        }
        if (valid) {
          const uint32_t *src = tl + rr * XS + 32 * hh + 4 * q4;
          const int fs = (int)fsum[32 * h + m];
          int v[I8_NSL];
#pragma unroll
          for (int d = 0; d < I8_NSL; ++d) v[d] = (int)src[16 * (d >> 2) + (d & 3)];
          v[I8_NSL - 1] -= T_D6_BIAS * fs;
          double sum = 0.0;
#pragma unroll
          for (int d = 0; d < I8_NSL; ++d) {
            double pd = (double)v[d] * (dsc0 * (double)((int64_t)1 << (8 * d)));
            asm volatile("" : "+v"(pd));  // (a product rounded on its own, as the stored slot was: no fused multiply-add with the sum)
            sum = d == 0 ? pd : sum + pd;
          }
          *dst = sum;
        }
      }
    };
    if constexpr (CG > 1 && !T_FLUSH_SERIAL) {
      // chunk groups: the groups' int32 accumulators are added up in the (now idle) count tile -- group 0 stores its tiles, the
      // others add theirs with LDS atomics (exact, whatever the order: what is flushed is what one group contracting all 32 chunks
      // would have flushed) -- and the CG waves that share a role (wave % WPG: same quad, same powers, same u-row half) then SPLIT
      // the summed tiles between them for the write-out.  Three barriers a round whatever the number of groups, and all eight
      // waves store.  (Round 6, phase clocks of config 5's state: the first cut -- group after group through group 0's registers,
      // two barriers per source and round, group 0's two waves alone converting and storing five tiles each -- took 19 000 cycles
      // per window next to 24 000 per TILE, and a window of a short series is four tiles.)
      // The write-out adds the seven digit sums of an element up itself -- ((((((d0 + d1) + d2) + d3) + d4) + d5) + d6), each
      // (double)(int32 sum) x its scale: the expression and the order resample_finalize_i8_kernel applies to the per-digit slots
      // the other kernels store -- so the window's slot holds ONE double per (replicate, power, column): an eighth of the bytes
      // written here and read by the finalize (a short series has a window every four tiles: config 5's call wrote and re-read
      // 2.5 GB of slots, as much as its samples; finalize 0.46 ms of a 5.8 ms call).  Same bits (I8Args::cpad rows without the
      // digit dimension: i8t_partials_summed() tells the host which layout a launch writes).
      constexpr int NT = 2 * NS + 1;                          // tiles of a wave: NS row sets x 2 replicate halves + the u-row tile
      constexpr int RT0 = T_CNT_BYTES / 4 / (WPG * XT);       // tiles per wave and round that fit the 64 KiB count tile
      constexpr int RT = RT0 < NT ? RT0 : NT;
      static_assert(RT >= 1 && WPG * RT * XT * 4 <= T_CNT_BYTES, "exchange slots");
      uint32_t *xrole = cntw + (size_t)((wave % WPG) * RT) * XT;  // this role's tiles of the round
      uint32_t *xch = xrole + lane;
      auto tile_of = [&](int tt) -> v16i & { return tt < 2 * NS ? acc[tt >> 1][tt & 1] : accu; };
      t_static_for<(NT + RT - 1) / RT>([&](auto rc) {
        constexpr int t0 = decltype(rc)::value * RT;
        if (cgrp == 0) {
#pragma unroll
          for (int k = 0; k < RT; ++k)
            if (t0 + k < NT) {
              v16i &T = tile_of(t0 + k);
#pragma unroll
              for (int r = 0; r < 16; ++r) xch[k * XT + r * XS] = (uint32_t)T[r];
            }
        }
        __syncthreads();
        if (cgrp != 0) {
#pragma unroll
          for (int k = 0; k < RT; ++k)
            if (t0 + k < NT) {
              v16i &T = tile_of(t0 + k);
#pragma unroll
              for (int r = 0; r < 16; ++r)
                __hip_atomic_fetch_add(&xch[k * XT + r * XS], (uint32_t)T[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        t_static_for<RT>([&](auto kc) {
          constexpr int k = decltype(kc)::value, tt = t0 + k;
          if constexpr (tt < NT) {
            if (cgrp == tt % CG) {  // wave-uniform: this group's share of the role's tiles
              if constexpr (tt < 2 * NS) flush_summed(xrole + k * XT, (tt & 1) ? 1 - hswap : hswap, tt >> 1, -1);
              else if (has_ut) flush_summed(xrole + k * XT, uh, 0, fu);  // wave-uniform
            }
          }
        });
        if constexpr (t0 + RT < NT) __syncthreads();  // (the next round overwrites the slots)
      });
#pragma unroll
      for (int fi = 0; fi < NS; ++fi) acc[fi][0] = acc[fi][1] = (v16i)(0);
    } else {
      if constexpr (CG > 1) {
        // (A/B build -DTXM_T_FLUSH_SERIAL: the first cut -- group after group into group 0's registers, which flushes alone)
        constexpr int NT = 2 * NS + 1;
        uint32_t *xch = cntw + (size_t)((wave % WPG) * 4) * 16 * 64 + lane;
        auto tile_of = [&](int tt) -> v16i & { return tt < 2 * NS ? acc[tt >> 1][tt & 1] : accu; };
#pragma unroll 1
        for (int src = 1; src < CG; ++src) {  // group src -> group 0, wave for wave (same quad, same powers)
#pragma unroll
          for (int t0 = 0; t0 < NT; t0 += 4) {
            if (cgrp == src) {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (t0 + k < NT) {
                  v16i &T = tile_of(t0 + k);
#pragma unroll
                  for (int r = 0; r < 16; ++r) xch[(k * 16 + r) * 64] = (uint32_t)T[r];
                  T = (v16i)(0);
                }
            }
            __syncthreads();
            if (cgrp == 0) {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (t0 + k < NT) {
                  v16i &T = tile_of(t0 + k);
#pragma unroll
                  for (int r = 0; r < 16; ++r) T[r] += (int)xch[(k * 16 + r) * 64];
                }
            }
            if (t0 + 4 < NT || src + 1 < CG) __syncthreads();  // (the next round overwrites the slots)
          }
        }
      }
      // instances without chunk groups: every wave writes its own tiles out -- digit-summed as above, a tile at a time through a
      // wave-private scratch in the idle count tile (narrow instances always; the wide instance unless the CALL carries a second
      // matrix, whose finalize reads per-digit u slots: YS launches, and PD = the first pass of such a call)
      constexpr bool summed1 = CG == 1 && !T_FLUSH_SERIAL && (NQ < 8 || (!YS && !PD));
      if constexpr (summed1) {
        uint32_t *xw = cntw + (size_t)wave * XT;
        auto via_lds = [&](v16i &T, int h, int rs, int ufrag) {
#pragma unroll
          for (int r = 0; r < 16; ++r) xw[r * XS + lane] = (uint32_t)T[r];
          flush_summed(xw, h, rs, ufrag);  // (the wave's own LDS operations execute in order)
          T = (v16i)(0);
        };
#pragma unroll
        for (int fi = 0; fi < NS; ++fi) {
          via_lds(acc[fi][0], hswap, fi, -1);
          via_lds(acc[fi][1], 1 - hswap, fi, -1);
        }
        if (has_ut) via_lds(accu, uh, 0, fu);  // wave-uniform
      } else if (CG == 1 || cgrp == 0) {  // wave-uniform
#pragma unroll
        for (int fi = 0; fi < NS; ++fi) {
          flush_tile(acc[fi][0], hswap, fi, -1);
          flush_tile(acc[fi][1], 1 - hswap, fi, -1);
        }
        if (has_ut) flush_tile(accu, uh, 0, fu);  // wave-uniform
      }
    }
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
template <int J0, int JN, bool WEIGHTED, bool YS = false, int NQ = 8, bool PD = false>
static int launch_pass_t(const I8Args &a, int K, size_t prog_bytes, hipStream_t st) {
  if (a.progress != nullptr) TXM_HIP(hipMemsetAsync(a.progress, 0, prog_bytes, st));
  constexpr int gs = 8 / NQ / T_CG_OF(NQ, JN), ns = (JN + gs - 1) / gs + (YS ? 1 : 0), npt = JN + ((YS && WEIGHTED && J0 > 0) ? 1 : 0);
  const size_t lds = (size_t)T_WAVES * (ns + 1) * T_PB + T_CNT_BYTES + 3u * I8_REPS * sizeof(uint32_t) +
                     (size_t)npt * T_PLINES_OF(T_CG_OF(NQ, JN)) * sizeof(double);
  const dim3 block(T_BLOCK);
  if constexpr (NQ < 8) {
    if (a.states != nullptr) {
      const dim3 grid((unsigned)(a.n_chunks * a.n_rbg), (unsigned)a.S);
      TXM_SET_MAX_LDS((&resample_i8t_kernel<J0, JN, WEIGHTED, YS, NQ, true>), lds);
      hipLaunchKernelGGL((resample_i8t_kernel<J0, JN, WEIGHTED, YS, NQ, true>), grid, block, lds, st, a, K);
      TXM_LAUNCH_CHECK();
      return TXM_OK;
    }
  }
  if (a.states != nullptr) {
    set_error("resample_i8t: a batched launch needs narrow states");
    return TXM_ERR_INVALID;
  }
  const dim3 grid((unsigned)(a.n_chunks * a.n_rbg));
  TXM_SET_MAX_LDS((&resample_i8t_kernel<J0, JN, WEIGHTED, YS, NQ, false, PD>), lds);
  hipLaunchKernelGGL((resample_i8t_kernel<J0, JN, WEIGHTED, YS, NQ, false, PD>), grid, block, lds, st, a, K);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// what this kernel takes: everything the one-power-per-column int8 path serves (8-byte loads: no alignment demand;
// columns past C re-read column 0 and are never flushed)
bool i8t_applicable(const double *x, int64_t ldx_s, int64_t C) {
  (void)x; (void)ldx_s; (void)C;
  return true;
}

// one power per observable column (C > 16, or order 0): every order 0..7; five row sets per pass, orders 5..7 in two
// passes over the sampler stream (the matrix pipe paces a pass, so the split is by fragments: 3 + 3, 4 + 3, 4 + 4).
// A second sample matrix (a.y) rides as one more row set of the LAST pass wherever that pass has four power row sets at
// most -- every order but 4 (i8t_carries_y).
bool i8t_carries_y(int64_t C, int K) {
  const bool narrow = C <= 16 && K >= 2;  // narrow states: the quad-sharing variant / txm_resample_i8.hip's power-packed kernel
  return !narrow && K != 5 && K >= 1 && K <= 8;
}

// narrow states (C <= 16 observables in the whole call, orders 1..7): the quad-sharing variant -- one pass for every order
// with one or two column quads, orders 6 and 7 in two passes with four (LDS: the factor tiles of 7 or 8 powers do not fit
// next to four row sets per wave).  0: not a narrow-state shape.
int i8t_narrow_nq(int64_t C_call, int K) {
  if (C_call > 16 || K < 2 || K > 8) return 0;
  return C_call <= 4 ? 1 : C_call <= 8 ? 2 : 4;
}
int i8_cpad(int64_t C_call, int K) {
  const int nq = i8t_narrow_nq(C_call, K);
  return nq ? 4 * nq : I8_CPAD;
}

// Which layout the launches of launch_resample_i8t leave in I8Args::part_x / part_u for a NARROW shape: every narrow instance adds
// the digit sums up in its flush and stores [window][replicate][power][column] (u-row: [window][replicate][power]) -- the per-digit
// slots [...][power][8 digit slots][column] only in the A/B build of the serial flush.  The wide instance does what
// I8Args::part_summed says (0 when the call carries a second matrix).  The finalize is told which (its `summed` argument).
bool i8t_partials_summed(int64_t C_call, int K) {
  return i8t_narrow_nq(C_call, K) != 0 && !T_FLUSH_SERIAL;   // (wide states: i8t_wide_summed)
}
// ... and the wide instance: digit-summed slots unless the CALL carries a second matrix (whose finalize reads per-digit u slots)
bool i8t_wide_summed(bool call_carries_y) { return !call_carries_y && !T_FLUSH_SERIAL; }

template <int NQ>
static int launch_narrow_t(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st) {
#define T_NARROW(JN_) (weighted ? launch_pass_t<0, JN_, true, false, NQ>(a, K, prog_bytes, st) : launch_pass_t<0, JN_, false, false, NQ>(a, K, prog_bytes, st))
  switch (K) {
    case 2: return T_NARROW(2);
    case 3: return T_NARROW(3);
    case 4: return T_NARROW(4);
    case 5: return T_NARROW(5);
    case 6: return T_NARROW(6);
    case 7: return T_NARROW(7);
    case 8: return T_NARROW(8);
    default: set_error("resample_i8t: order out of range"); return TXM_ERR_INVALID;
  }
#undef T_NARROW
}

int launch_resample_i8t(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st) {
  int rc = TXM_OK;
  const bool ys = a.y != nullptr;
  if (const int nq = i8t_narrow_nq(a.C_call, K)) {
    if (ys) {
      set_error("resample_i8t: no second matrix on the narrow-state kernel");
      return TXM_ERR_INVALID;
    }
    if (nq == 1) return launch_narrow_t<1>(a, K, weighted, prog_bytes, st);
    if (nq == 2) return launch_narrow_t<2>(a, K, weighted, prog_bytes, st);
#define T_N4(J0_, JN_) (weighted ? launch_pass_t<J0_, JN_, true, false, 4>(a, K, prog_bytes, st) : launch_pass_t<J0_, JN_, false, false, 4>(a, K, prog_bytes, st))
    switch (K) {
      case 2: return T_N4(0, 2);
      case 3: return T_N4(0, 3);
      case 4: return T_N4(0, 4);
      case 5: return T_N4(0, 5);
      case 6: return T_N4(0, 6);
      case 7: rc = T_N4(0, 4); return rc != TXM_OK ? rc : T_N4(4, 3);
      default: rc = T_N4(0, 4); return rc != TXM_OK ? rc : T_N4(4, 4);
    }
#undef T_N4
  }
  if (ys && K == 5) {
    set_error("resample_i8t: a second matrix cannot ride on a five-power pass");
    return TXM_ERR_INVALID;
  }
#define T_PASS(J0_, JN_) (weighted ? launch_pass_t<J0_, JN_, true>(a, K, prog_bytes, st) : launch_pass_t<J0_, JN_, false>(a, K, prog_bytes, st))
  // the first pass of a two-pass call whose LAST pass carries a second matrix: per-digit slots like that pass (one layout per call)
#define T_FIRST(J0_, JN_)                                                                                                      \
  (ys ? (weighted ? launch_pass_t<J0_, JN_, true, false, 8, true>(a, K, prog_bytes, st)                                        \
                  : launch_pass_t<J0_, JN_, false, false, 8, true>(a, K, prog_bytes, st))                                      \
      : T_PASS(J0_, JN_))
#define T_LAST(J0_, JN_)                                                                                                   \
  (ys ? (weighted ? launch_pass_t<J0_, JN_, true, true>(a, K, prog_bytes, st) : launch_pass_t<J0_, JN_, false, true>(a, K, prog_bytes, st)) \
      : T_PASS(J0_, JN_))
  switch (K) {
    case 1: rc = T_LAST(0, 1); break;
    case 2: rc = T_LAST(0, 2); break;
    case 3: rc = T_LAST(0, 3); break;
    case 4: rc = T_LAST(0, 4); break;
    case 5: rc = T_PASS(0, 5); break;
    case 6: rc = T_FIRST(0, 3); if (rc == TXM_OK) rc = T_LAST(3, 3); break;
    case 7: rc = T_FIRST(0, 4); if (rc == TXM_OK) rc = T_LAST(4, 3); break;
    case 8: rc = T_FIRST(0, 4); if (rc == TXM_OK) rc = T_LAST(4, 4); break;
    default: set_error("resample_i8t: order out of range"); return TXM_ERR_INVALID;
  }
#undef T_PASS
#undef T_FIRST
#undef T_LAST
  return rc;
}

}  // namespace txm
