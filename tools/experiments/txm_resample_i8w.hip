// txm_resample_i8w.hip -- the int8 bootstrap contraction with SIXTEEN waves per workgroup (four per SIMD, 128 registers
// each): the wide shape (C > 16 per column group, one power per observable column) of txm_resample_i8t.hip re-cut so
// that a wave's serial instruction stream per k-step is half as long.  Same sums, same sampler stream, same slicing,
// same partial-sum slots (cmomy.wrap_resample_vals as called from thermoextrap data.py:1803-1810, 1354-1366):
//        S1[r][c][j] = sum_i f[r][i] w_i du_i^j dx_ic        S0[r][j] = sum_i f[r][i] w_i du_i^j
// and the output is BIT FOR BIT that of the eight-wave kernel (tests/test_i8w_gpu.py), because every accumulation is an
// exact int32 sum of the same digits and the conversion at the flush is the same expression.
//
// Why (round-3 evidence): in the eight-wave kernel a k-step is the in-order stream of ONE wave (~140
// instructions for 11 MFMAs; a single wave alone needs 80 % of the time all eight need) and two waves per SIMD are
// too few to fill each other's stalls -- no unit is more than half busy.  A wave cannot be given fewer tiles there:
// its 10-11 accumulator tiles are what lets it own a column quad from the load to the accumulators without talking to
// anybody.  Here TWO waves share a column quad:
//   wave (q, h), q = column quad 0..7, h = replicate half 0..1: five accumulator tiles (80 registers) -- the five
//   power row sets of quad q for the replicates 32 h .. 32 h + 31;
//   it PRODUCES the fixed-point words of the 16-sample unit h of every chunk (one sample x one column per lane: one x
//   load, one v_fma_f64 + two v_xor per power) and CONSUMES both units (two transposing reads + ONE MFMA per power).
// The pair hands chunks over through a double-buffered region and one progress word per wave in LDS (no s_barrier in
// the k-steps): before k-step g a wave waits until its partner has finished k-step g - 1, which says both "the
// partner's unit of chunk g is written" and "the partner has read chunk g - 1, whose buffer I overwrite now".  LDS
// executes a wave's instructions in order, so the progress store behind the data stores (and behind the transposing
// reads) is the release, and the transposing reads behind the progress load are the acquire.
// What makes it fit:
//   registers -- the u-row (the S0 monomials w du^j, dx = 1) has no tiles of its own: slot 7 of every 8-byte word
//   (the exponent byte of the magic-constant double) is a dead MFMA column in the eight-wave kernel; here the dead
//   bytes of the column quads 0 and 1 carry the seven digits of the row set's own u monomial (one v_fma_f64 + one
//   v_perm_b32 per word on those four waves), so the sums S0 ride on MFMA columns that were computing garbage.
//   LDS -- the staged factor tiles w du^j (5 x 8 KiB) are gone: a lane has ONE sample per k-step and multiplies the
//   powers up itself (same product chain as the staging loop of the eight-wave kernel, hence the same bits); only du
//   (and w) are staged.  Regions: [quad][buffer][power] planes of 512 B, low dwords and high dwords in two areas 128 B
//   out of phase (the two 16-lane groups of a half-wave read the two planes on disjoint banks): 82 KiB at five powers.
#include "txm_i8t_common.h"

namespace txm {

constexpr int W_BLOCK = 1024;
constexpr int W_WAVES = W_BLOCK / 64;
#ifndef TXM_W_XD
#define TXM_W_XD 2
#endif
constexpr int W_XD = TXM_W_XD;  // k-steps between the request of an x unit and its use

__device__ __forceinline__ uint32_t w_lds_load(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void w_lds_store(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

// LDS plan of a launch with JN power row sets
template <int JN, bool WEIGHTED>
struct WPlan {
  static constexpr int LO = 0;                              // [quad 8][buffer 2][power JN][512]: low dwords
  static constexpr int QB = JN * 512;                       // one (quad, buffer)
  static constexpr int HI = 16 * QB + 128;                  // the same shape, high dwords
  static constexpr int CNT = HI + 16 * QB;                  // [256 words][64 replicates]
  static constexpr int SMALL = CNT + T_CNT_BYTES;           // fsum[64] cnt_a[64] cnt_b[64] prog[16] (u32)
  static constexpr int DU = SMALL + (3 * I8_REPS + 16) * 4; // [1024] du (+ [1024] w)
  static constexpr int TOTAL = DU + SM_T * 8 * (WEIGHTED ? 2 : 1);
  static_assert(HI + 512 * JN < 65536, "store offsets are 16-bit immediates");
  static_assert(15 * QB + 256 < 65536, "ds_write_addtid takes its base from M0[15:0]");
  static_assert(CNT % 16 == 0 && DU % 8 == 0, "alignment");
};
constexpr int W_LDS_MAX = 160 * 1024;

template <int J0, int JN, bool WEIGHTED>
__global__ __launch_bounds__(W_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void resample_i8w_kernel(const I8Args a, const int K) {
  using P = WPlan<JN, WEIGHTED>;
  static_assert(JN >= 1 && JN <= 5 && J0 + JN <= 8, "power range");
  static_assert(P::TOTAL <= W_LDS_MAX, "LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *cntw = reinterpret_cast<uint32_t *>(lds + P::CNT);
  uint32_t *fsum = reinterpret_cast<uint32_t *>(lds + P::SMALL);  // [64] draws per replicate in the window
  uint32_t *cnt_a = fsum + I8_REPS;                                // [64] tile draw counts, double buffered
  uint32_t *cnt_b = cnt_a + I8_REPS;
  double *dtile = reinterpret_cast<double *>(lds + P::DU);         // du of the 1024 samples the tile's k-steps slice
  double *wtile = dtile + SM_T;                                    // ... and w / max|w| (WEIGHTED)

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int quad = wave >> 1, h = wave & 1;  // column quad, replicate half = produced unit
  const int n32 = lane & 31, khalf = lane >> 5;
  const uint32_t prog_me = (uint32_t)(P::SMALL + (3 * I8_REPS + wave) * 4);
  const uint32_t prog_pt = (uint32_t)(P::SMALL + (3 * I8_REPS + (wave ^ 1)) * 4);

  // ---- producer role: lane = (sample 16 h + (l >> 2) of the chunk, column l & 3 of the quad)
  const int ps = lane >> 2, cl = lane & 3;
  const int col = 4 * quad + cl;
  const int ccol = col < a.C ? col : 0;  // columns >= C re-read column 0: their sums are never flushed
  const uint32_t xo = (uint32_t)(((16 * h + ps) * a.ldx_s + ccol) * 8);  // byte offset from the chunk's (uniform) row base
  // M0 of the stores into buffer b: the low plane of (quad, b, power 0) + this unit's 256 bytes
  const uint32_t st_m0 = (uint32_t)(P::LO + (2 * quad) * P::QB + 256 * h);
  // the u-row digit this lane's dead byte carries: digit 4 quad + cl of the row set's monomial (quads 0 and 1; digit 7
  // does not exist)
  const bool carry_u = quad < 2;  // wave-uniform
  const int udig = 4 * quad + cl;
  const uint32_t usel = udig < I8_NSL ? (uint32_t)udig : 0x0cu;  // v_perm selector: byte udig of the u word (or a zero) into byte 0
  const uint32_t uxor = udig < I8_NSL - 1 ? 0x80u : 0u;         // digits 0..5 are biased by 128, digit 6 by T_D6_BIAS (flush)
  const uint32_t ub_addr = st_m0 + lane * 4u;                   // + buffer, high area, power, byte 3: immediates
  // ---- consumer role
  // transposing read of (plane g, rows 16 khalf + 0..7): lane 2 r + p of the 16-lane group supplies row r, bytes 8 p ..
  const uint32_t rd_off = (uint32_t)(P::LO + (2 * quad) * P::QB + ((lane >> 4) & 1) * P::HI + (16 * khalf + ((lane & 15) >> 1)) * 16 +
                                     (lane & 1) * 8);  // + buffer * QB + power * 512; second read + 128
  // tile column n32 -> (column, digit slot)
  const int tcl = (n32 >> 2) & 3, tdg = 4 * (n32 >> 4) + (n32 & 3);
  const uint32_t a_off = (uint32_t)((4 * khalf) * I8_REPS + 32 * h + n32);  // words of the A operand; + 8 s * 64 + q * 64

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

  v16i acc[JN];
#pragma unroll
  for (int e = 0; e < JN; ++e) acc[e] = (v16i)(0);

  // stage-3 role: lane = replicate
  const int64_t my_rep = rep0 + lane;
  const bool rep_live = my_rep < a.nrep;
  const uint32_t rstream = a.rep_base + (uint32_t)my_rep;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  uint32_t fdraws = 0;

#ifdef TXM_I8T_TIMING
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk0 = clock64();
#define W_TICK(k) do { const long long t1_ = clock64(); tm[k] += t1_ - tk0; tk0 = t1_; } while (0)
#else
#define W_TICK(k) do {} while (0)
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
    int64_t tt_end = (win + 1) * WT;
    if (tt_end > t_end) tt_end = t_end;
    double c50 = 0x1p50;
    asm volatile("" : "+v"(c50));  // (a register operand: 2^50 and the magic constant cannot both sit on the constant bus)
    uint32_t gstep = 0;  // k-steps this wave has finished in the window = the value of its progress word

    // ---- one k-step of a wave.  consume: chunk s of the tile (buffer PB) on the matrix pipe; produce: this wave's unit
    // of chunk s + 1 into buffer 1 - PB.  e0 = entry of the produced chunk's first sample in the staged tile (e0 < 0:
    // the direct path of a window's first chunk -- du / w of this lane's sample come in d_du / d_w)
    auto kstep = [&](auto produce_c, auto consume_c, auto pb_c, int s, double xin, int e0, double d_du, double d_w) {
#ifdef TXM_T_NO_PRODUCE  // ablation build
      constexpr bool produce = false;
#else
      constexpr bool produce = decltype(produce_c)::value;
#endif
      constexpr bool consume = decltype(consume_c)::value;
      constexpr int PB = decltype(pb_c)::value;
      v4i A = (v4i)(0);
      v2i Ba = (v2i)(0), Bb = (v2i)(0);
      if constexpr (consume) {
        // acquire: the partner has finished k-step gstep - 1
#ifndef TXM_W_NO_SYNC
        while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)w_lds_load(prog_pt)) - gstep) < 0) __builtin_amdgcn_s_sleep(1);
#endif
        const uint32_t *cw = cntw + s * (8 * I8_REPS) + a_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = (int)cw[q * I8_REPS];
        Ba = T_TRREAD((lds_v2i)(lds + rd_off + PB * P::QB));
        Bb = T_TRREAD((lds_v2i)(lds + rd_off + PB * P::QB + 128));
      }
      double dx = 0.0, du = d_du, pw = 1.0;
      if constexpr (produce) {
        dx = (xin - px) * sc;
        if (e0 >= 0) {  // uniform
          du = dtile[e0 + 16 * h + ps];
          if constexpr (WEIGHTED) pw = wtile[e0 + 16 * h + ps];
        } else if constexpr (WEIGHTED) pw = d_w;
#pragma unroll
        for (int q = 0; q < J0; ++q) pw *= du;
#ifndef TXM_T_NO_WRITE
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(st_m0 + (uint32_t)((1 - PB) * P::QB)) : "memory");
#endif
      }
      const double pw0 = pw;
      t_static_for<JN>([&](auto fic) {
        constexpr int fi = decltype(fic)::value;
        v2i Na = (v2i)(0), Nb = (v2i)(0);
        if constexpr (consume && fi + 1 < JN) {
          Na = T_TRREAD((lds_v2i)(lds + rd_off + PB * P::QB + (fi + 1) * 512));
          Nb = T_TRREAD((lds_v2i)(lds + rd_off + PB * P::QB + (fi + 1) * 512 + 128));
        }
        if constexpr (consume) {
          const v4i B = {Ba[0], Ba[1], Bb[0], Bb[1]};
          t_mfma<true>(acc[fi], A, B);
        }
        if constexpr (produce) {
          // the word of (sample, column, power J0 + fi) of chunk s + 1
          // (the fused multiply-adds as asm: left to itself the compiler takes the two-operand form, v_mov_b64 + v_fmac_f64)
          double wd;
          if constexpr (!WEIGHTED && J0 + fi == 0) wd = dx + T_MAGIC;
          else asm("v_fma_f64 %0, %1, %2, %3" : "=v"(wd) : "v"(pw), "v"(dx), "s"(T_MAGIC));
          const uint64_t bits = (uint64_t)__double_as_longlong(wd);
          const uint32_t lo = (uint32_t)bits ^ 0x80808080u;
          const uint32_t hi = (uint32_t)(bits >> 32) ^ 0x00008080u;
#ifdef TXM_T_NO_WRITE  // ablation build
          asm volatile("" ::"v"(lo), "v"(hi));
#else
          // (M0 = this k-step's store base, set once above; nothing the compiler emits in the k-steps writes M0 --
          // tools/check_m0.py scans a --save-temps assembly of this file for any other write of M0)
          asm volatile("ds_write_addtid_b32 %0 offset:%2\n\t"
                       "ds_write_addtid_b32 %1 offset:%3"
                       :
                       : "v"(lo), "v"(hi), "n"(fi * 512), "n"(P::HI + fi * 512)
                       : "memory");
#endif
          if constexpr (fi + 1 < JN) pw *= du;
        }
        Ba = Na;
        Bb = Nb;
      });
      if constexpr (produce) {
        if (carry_u) {  // wave-uniform, quads 0 and 1: the dead bytes of this unit's words take the digits of the u monomials
          double pc = pw0;
          t_static_for<JN>([&](auto fic) {
            constexpr int fi = decltype(fic)::value;
            double ud;
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ud) : "v"(pc), "v"(c50), "s"(T_MAGIC));
            const uint64_t ub = (uint64_t)__double_as_longlong(ud);
            const uint32_t dg = __builtin_amdgcn_perm((uint32_t)(ub >> 32), (uint32_t)ub, usel) ^ uxor;
#ifndef TXM_T_NO_WRITE
            asm volatile("ds_write_b8 %0, %1 offset:%2" : : "v"(ub_addr), "v"(dg), "n"((1 - PB) * P::QB + P::HI + fi * 512 + 3) : "memory");
#endif
            if constexpr (fi + 1 < JN) pc *= du;
          });
        }
      }
      if constexpr (consume) {
        // release: data stores and transposing reads of this k-step are ahead of this store in the wave's LDS queue
        ++gstep;
#ifndef TXM_W_NO_SYNC
        w_lds_store(prog_me, gstep);
#endif
      }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    constexpr std::integral_constant<int, 0> B0{};
    constexpr std::integral_constant<int, 1> B1{};

    // ---- flush: int32 accumulators of one window -> its slot of the partial sums (stored, never re-read here)
    // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
    auto flush_tile = [&](v16i &T, int rs) {
      // an opaque zero re-created per tile: the addresses below are then computed where they are used
      uint32_t z = 0;
      asm volatile("" : "+v"(z));
      const int64_t opq = (int64_t)z;
      const int c = 4 * quad + tcl;
      const int j = J0 + rs;
      const int64_t row0 = (int64_t)win * a.nrep_pad + rep0 + 32 * h + 4 * khalf;
      bool valid;
      double dsc;
      double *base;
      size_t stride;
      int dig;
      if (tdg < I8_NSL) {  // a digit of column c: [window][replicate][power][digit slot][column]
        valid = c < a.C;
        dig = tdg;
        dsc = wt[I8_WT_DSP + j] * wt[I8_WT_DSC + (c < a.C ? c : 0)];
        base = a.part_x + (((size_t)row0 * K + j) * 8 + tdg) * a.cpad + c + opq;
        stride = (size_t)K * a.cpad * 8;
      } else {  // the dead slot: digit c of the u monomial: [window][replicate][power][digit slot]
        valid = carry_u && c < I8_NSL;
        dig = c < I8_NSL ? c : 0;
        dsc = wt[I8_WT_DSP + j] * 0x1p-50;
        base = a.part_u + ((size_t)row0 * K + j) * 8 + dig + opq;
        stride = (size_t)K * 8;
      }
      dsc *= (double)((int64_t)1 << (8 * dig));
      const int bias = dig == I8_NSL - 1 ? T_D6_BIAS : 0;
      if (valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r >> 2) * 8 + (r & 3);
          const int v = T[r] - bias * (int)fsum[32 * h + m + 4 * khalf];
          base[(size_t)m * stride] = (double)v * dsc;
        }
      }
      T = (v16i)(0);
    };

    bool first_tile = true;
    uint32_t *cnt_cur = cnt_a, *cnt_nxt = cnt_b;
    constexpr int XD = W_XD, UNR = XD > 4 ? XD : 4;
    double XR[XD];  // the lane's sample of a chunk, requested XD k-steps ahead: chunk c lives in slot c % XD
    // running wave-uniform row pointer of the next chunk to request (chunks are adjacent in memory inside a tile and
    // from tile to tile, all but the slid last tile of the series)
    const char *xq = nullptr;
    int64_t xq_step = 0;
    auto load_q = [&](double &r) {
#ifdef TXM_T_NO_LOAD
      r = px + 1e-3;
#else
      r = *reinterpret_cast<const double *>(xq + xo);
#endif
    };
    auto set_q = [&](int64_t i0) { xq = reinterpret_cast<const char *>(a.x + i0 * a.ldx_s + a.col0); };

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
      W_TICK(0);

      // ---- tile prologue: counts of this tile (first tile of a window: loaded here; otherwise parked by the previous
      // tile), zero the count tile, chunk 0 of the first tile
      if (first_tile) {
        if (wave == 0) cnt_cur[lane] = rep_live ? a.counts[(size_t)my_rep * a.ntiles + t] : 0u;
        if (lane == 0) w_lds_store(prog_me, 0u);  // (the barriers below are ahead of the first acquire)
        set_q(wbase);
        xq_step = 32 * a.ldx_s * 8;
        load_q(XR[0]);
        xq += xq_step;
      }
      // staging request (in flight during the zeroing and the fill): entries 0 .. 991 = samples wbase + 32 ...,
      // entries 992 .. 1023 = the next tile's first chunk
      double su, sw = 1.0;
      {
        const int e = (int)threadIdx.x;
        const int64_t i = e < SM_T - 32 ? wbase + 32 + e : wnext + (e - (SM_T - 32));
        su = a.u[i];
        if constexpr (WEIGHTED) sw = a.w[i];
      }
      uint32_t ncnt = 0;
      if (wave == 0 && has_next && rep_live) ncnt = a.counts[(size_t)my_rep * a.ntiles + t + 1];
      // (the last k-step of the previous tile was the count tile's and the staged tile's last reader: barrier below)
      __syncthreads();
      for (int e = threadIdx.x; e < T_CNT_BYTES / 16; e += W_BLOCK) reinterpret_cast<uint4 *>(cntw)[e] = make_uint4(0, 0, 0, 0);
      W_TICK(1);
      __syncthreads();
      W_TICK(2);

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
          for (; c < call_all; c += W_WAVES) t_fill_call<true>(cntw, a.k0, a.k1, rstream, (uint32_t)t, c, n, lane4);
#pragma unroll 1
          for (; c * 12u < nmax; c += W_WAVES) t_fill_call<false>(cntw, a.k0, a.k1, rstream, (uint32_t)t, c, n, lane4);
        } else if (tsize != (uint32_t)SM_T) {
          // the partial last tile: the stream is defined over 64 virtual lanes per replicate (txm_sampler.h)
          for (int rr = wave * (I8_REPS / W_WAVES); rr < (wave + 1) * (I8_REPS / W_WAVES); ++rr) {
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
        // the words of this wave's unit of chunk 0 (no matrix work yet; its u / w straight from memory) into buffer 0,
        // chunks 1 .. XD requested
        const double d_du = (a.u[wbase + 16 * h + ps] - pu) * inv_du;
        double d_w = 1.0;
        if constexpr (WEIGHTED) d_w = a.w[wbase + 16 * h + ps] * inv_w;
        kstep(YES, NO, B1, 0, XR[0], -1, d_du, d_w);  // (consumed buffer "1" = produced buffer 0)
#pragma unroll
        for (int c = 1; c <= XD; ++c) {
          load_q(XR[c % XD]);
          xq += xq_step;
        }
        first_tile = false;
      }
      // park the staged tile
      {
        const int e = (int)threadIdx.x;
        dtile[e] = (su - pu) * inv_du;
        if constexpr (WEIGHTED) wtile[e] = sw * inv_w;
      }
      if (wave == 0) cnt_nxt[lane] = ncnt;
      W_TICK(3);
      __syncthreads();
      W_TICK(4);

      // from the rows of chunk 31 to the rows of the next tile's chunk 0
      const int64_t xq_jump = has_next ? (wnext - (wbase + SM_T - 32)) * a.ldx_s * 8 : 0;
      // ---- 32 k-steps, four per trip, NO barrier between them.  Step s contracts chunk s (buffer s & 1), slices this
      // wave's unit of chunk s + 1 from the ring slot (s + 1) % XD (x requested XD steps ago) and requests chunk
      // s + 1 + XD into that slot.  Chunk 32 is the next tile's chunk 0 (words nobody reads when there is no next tile).
#pragma unroll 1
      for (int s = 0; s < T_STEPS; s += UNR) {
#pragma unroll
        for (int e = 0; e < UNR; ++e) {
          const int sq = s + e;
          double &R = XR[(e + 1) % XD];
          const double cur_x = R;
          load_q(R);  // chunk sq + 1 + XD
          // the request after chunk 31 is the next tile's chunk 0 (not adjacent when that tile is the slid last one: a jump
          // of its own); no next tile: chunk 31 again, words nobody reads
          if ((T_STEPS - 2 - XD - e) % UNR == 0) {
            const bool at31 = s == T_STEPS - 2 - XD - e;
            xq += at31 ? xq_jump : xq_step;
            xq_step = (at31 && !has_next) ? 0 : xq_step;
          } else xq += xq_step;
          if (e % 2 == 0) kstep(YES, YES, B0, sq, cur_x, sq * 32, 0.0, 1.0);
          else kstep(YES, YES, B1, sq, cur_x, sq * 32, 0.0, 1.0);
          W_TICK(5);
        }
      }
      W_TICK(6);
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
    for (int fi = 0; fi < JN; ++fi) flush_tile(acc[fi], fi);
    __syncthreads();  // fsum and the progress words are rewritten by the next window
    W_TICK(7);
  }
  if (pg != nullptr && threadIdx.x == 0)
    __hip_atomic_store(&pg[rbg & 63], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TXM_I8T_TIMING
  // diagnostic build only: phase cycles of two workgroups into the slack behind the window table (even waves)
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133) && (wave & 1) == 0)
    for (int k = 0; k < 8; ++k) a.wtab[a.nwin * I8_WT_STRIDE + ((blockIdx.x ? 1 : 0) * 8 + (wave >> 1)) * 8 + k] = (double)tm[k];
#endif
}

// ---------------------------------------------------------------------------
template <int J0, int JN, bool WEIGHTED>
static int launch_pass_w(const I8Args &a, int K, size_t prog_bytes, hipStream_t st) {
  if (a.progress != nullptr) TXM_HIP(hipMemsetAsync(a.progress, 0, prog_bytes, st));
  const dim3 grid((unsigned)(a.n_chunks * a.n_rbg)), block(W_BLOCK);
  constexpr size_t lds = (size_t)WPlan<JN, WEIGHTED>::TOTAL;
  TXM_SET_MAX_LDS((&resample_i8w_kernel<J0, JN, WEIGHTED>), lds);
  hipLaunchKernelGGL((resample_i8w_kernel<J0, JN, WEIGHTED>), grid, block, lds, st, a, K);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// what the sixteen-wave kernel takes: the wide shape (one power per observable column: C > 16 in the call, or order 0),
// no second matrix, and a weighted launch only where du AND w fit next to the regions (four power row sets at most:
// every weighted order but 4)
bool i8w_takes(const I8Args &a, int K, bool weighted) {
  if (i8t_narrow_nq(a.C_call, K) != 0 || a.y != nullptr || K < 1 || K > 8) return false;
  if (weighted && K == 5) return false;
  return true;
}

int launch_resample_i8w(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st) {
  int rc = TXM_OK;
#define W_PASS(J0_, JN_) (weighted ? launch_pass_w<J0_, JN_, true>(a, K, prog_bytes, st) : launch_pass_w<J0_, JN_, false>(a, K, prog_bytes, st))
  switch (K) {
    case 1: rc = W_PASS(0, 1); break;
    case 2: rc = W_PASS(0, 2); break;
    case 3: rc = W_PASS(0, 3); break;
    case 4: rc = W_PASS(0, 4); break;
    case 5: rc = weighted ? TXM_ERR_INVALID : launch_pass_w<0, 5, false>(a, K, prog_bytes, st); break;
    case 6: rc = W_PASS(0, 3); if (rc == TXM_OK) rc = W_PASS(3, 3); break;
    case 7: rc = W_PASS(0, 4); if (rc == TXM_OK) rc = W_PASS(4, 3); break;
    case 8: rc = W_PASS(0, 4); if (rc == TXM_OK) rc = W_PASS(4, 4); break;
    default: set_error("resample_i8w: order out of range"); return TXM_ERR_INVALID;
  }
#undef W_PASS
  return rc;
}

}  // namespace txm
