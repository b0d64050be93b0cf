// txm_resample_i8gn.hip -- the table-fed int8 bootstrap contraction for NARROW states (C <= 16 observables; round 6).
//
// Same sums, same fixed-point words, same partial-sum slots as the quad-sharing variant of resample_i8t_kernel
// (cmomy.wrap_resample_vals as called from thermoextrap data.py:1803-1810, 1354-1366; StateCollection.resample,
// models.py:614-641):
//        S1[r][c][j] = sum_i f[r][i] w_i du_i^j dx_ic        S0[r][j] = sum_i f[r][i] w_i du_i^j
// with the counts f read from the table count_table_kernel (txm_count_table.hip) wrote in MFMA-A-operand order.
//
// Why.  The narrow-state launches of the kernel that draws in place are bound by what ONE workgroup per CU can overlap between
// its barriers (profiles/r06_experiments.md section 2: fill 31-35 % of the call, slicing 25-28 %, the parts add), and the
// 64 KiB count tile of a 1024-sample sampler tile rules out both a 128-replicate workgroup (128 KiB of the CU's 160) and a
// second resident workgroup.  Fed from the table a workgroup needs no count tile at all -- a ring of count words -- so it takes
// 128 replicates (every sliced word feeds four MFMAs instead of two, every chore of a k-step serves twice the replicates), has no
// fill phase, no zeroing, no per-tile barriers; the generator runs at the Philox-bound rate of the fill it replaces.
// What it costs: the table crosses HBM twice (one byte per replicate and sample, written by the generator, read here), which for a
// narrow state is 10-30 x the samples themselves -- measured against the fused kernel (tools/narrow_table_sweep.py,
// profiles/r06_narrow_table_sweep.txt) the pair generator + this kernel is 1.0-1.27 x faster from two replicate groups on, even at
// one group, slower at 64 replicates; the dispatch rule (txm_resample.hip narrow_table_pays) follows those numbers.
//
// Workgroup = 8 waves x 128 replicates x NCQ column quads (1, 2 or 4) x the powers J0 .. J0 + JN - 1.  The GS = 8 / NCQ waves
// that share a quad split the powers (wave w: quad w % NCQ, powers g, g + GS, ... with g = w / NCQ); the S0 monomials w du^j
// (dx = 1) are u-row fragments of four monomials x eight digit slots on the LAST waves (the ones with the fewest powers).  Each
// of these is a SLOT of four accumulator tiles (GnGeom below); waves with at most two slots take their k-steps in pairs.
// Counts, x, u and w arrive by LDS-DMA as in resample_i8g_kernel: the loader waves (GnRoles) request a block of four k-steps at
// a time, two or three blocks ahead, one s_barrier per block; the x ring is per QUAD (the waves of a quad read the same chunk).
// The int32 sums of a window are exact and the flush is the expression of resample_i8t_kernel: the two kernels agree BIT FOR
// BIT (tests/test_i8gn_gpu.py).
#include "txm_i8g.h"

namespace txm {

// Who does the block's chores.  A wave's matrix work is its live power row sets + its u-row fragment; the DMA requests (four
// loader waves, a block step each) and the staging of the factors (two waves) go to the waves with the LEAST of it -- with one quad
// and four powers the waves 4..6 have none at all, and the first cut had the four working waves issue every DMA piece and the
// u-row wave stage the factors in front of its k-steps (890 cycles a k-step for 20 MFMAs).
template <int JN, int NCQ>
struct GnRoles {
  static constexpr int GS = T_WAVES / NCQ, UF = (JN + 3) / 4;
  static constexpr int work(int w) {  // (a u-row step costs more than a power row's: ldexp + add per word, phase clocks 2190 against 1840 cycles a block)
    int n = 0;
    for (int j = w / NCQ; j < JN; j += GS) n += 2;
    return n + (w >= T_WAVES - UF ? 3 : 0);
  }
  static constexpr int idle() {
    int n = 0;
    for (int w = 0; w < T_WAVES; ++w) n += work(w) == 0 ? 1 : 0;
    return n;
  }
  static constexpr int first_idle(int skip) {
    for (int w = 0; w < T_WAVES; ++w)
      if (work(w) == 0 && skip-- == 0) return w;
    return 0;
  }
  // first wave of the run of `len` consecutive waves with the smallest total work (ties: the later run -- away from wave 0's polling)
  static constexpr int best_run(int len) {
    int best = 0, bw = 1 << 30;
    for (int w0 = 0; w0 + len <= T_WAVES; ++w0) {
      int t = 0;
      for (int k = 0; k < len; ++k) t += work(w0 + k);
      if (t <= bw) { bw = t; best = w0; }
    }
    return best;
  }
  // two waves without matrix work take all the requests (two block steps each) and the staging; otherwise the requests are spread
  // over the four least loaded consecutive waves (a block step each) and the staging goes to the two least loaded
  static constexpr int NLD = idle() >= 2 ? 2 : 4;
  static constexpr int LOAD0 = idle() >= 2 ? first_idle(0) : best_run(4);
  static constexpr int STAGE0 = idle() >= 3 ? first_idle(1) : idle() >= 2 ? first_idle(0) : best_run(2);
  static_assert(idle() < 2 || first_idle(1) == first_idle(0) + 1, "idle waves are consecutive");
};

// The shape of one instance: what a wave holds, the depth of the rings, the LDS map (shared by the kernel and its launcher).
// A wave's matrix work is a list of SLOTS, each four accumulator tiles (one per replicate quarter): its power rows g, g + GS, ...
// first, then -- on the last UF waves -- one u-row fragment.  A slot's fixed-point word is fma(factor, m, magic) with m = dx for a
// power row and 2^50 for a u-row monomial (= the ldexp + add of resample_i8t_kernel, bit for bit: the product is exact either way).
template <int JN, int NCQ, bool WEIGHTED>
struct GnGeom {
  static constexpr int GS = T_WAVES / NCQ, UF = (JN + 3) / 4, NQ = 4, NPT = JN;
  static constexpr int rows(int w) { return w / NCQ < JN ? (JN - 1 - w / NCQ) / GS + 1 : 0; }
  static constexpr int slots(int w) { return rows(w) + (w >= T_WAVES - UF ? 1 : 0); }
  static constexpr int max_slots() {
    int m = 0;
    for (int w = 0; w < T_WAVES; ++w) m = slots(w) > m ? slots(w) : m;
    return m;
  }
  static constexpr int MAXS = max_slots();
  // k-steps a wave takes TOGETHER.  A narrow k-step is a handful of MFMAs between two LDS round trips (read the words stored a step
  // ago -- contract -- store the next ones): with one workgroup on the CU and one or two working waves on a SIMD nothing hides them
  // (phase clocks of the first cut, one power row a wave: 460 cycles a k-step for 4 MFMAs).  With two X regions a wave the words are
  // sliced TWO chunks ahead: a pair of k-steps reads both regions' words, both steps' count words, factors and x in one go, issues
  // the MFMAs of both, then stores the two chunks after next -- one round trip per pair.  (Twelve tiles -- three slots -- leave no
  // registers for the second set of operands: those instances take their k-steps one at a time.)
  static constexpr int W = MAXS <= 2 ? 2 : 1;
  static constexpr int WREG = W * MAXS * T_PB;       // a wave's X regions: [step of the pair][slot]
  static constexpr int A_STEP = NQ * 1024;           // count words of one k-step: [quarter][1024]
  static constexpr int FU = G_BS * 16 * NPT * 8 + 136;  // bytes between the factor lines of a lane's two 16-sample units
  static constexpr int lds_bytes(int d) {
    return T_WAVES * WREG + (d + 1) * G_BS * A_STEP + NCQ * (d + 1) * G_BS * 1024 + (d + 2) * G_RAW + 3 * 2 * FU + G_REPS * 4;
  }
  // DMA lead: a narrow k-step is short (200-600 cycles), and a request to HBM takes 2-3 us to land -- the ONE block of lead
  // resample_i8g_kernel lives on (its k-steps take 1800 cycles) left every block's barrier waiting for the DMA (first cut: 1200
  // cycles a k-step for 20 MFMAs).  Block B + D is requested at the top of block B, into rings of D + 1 blocks: three blocks
  // where the CU's 160 KiB hold them.
  static constexpr int D = NCQ < 4 && lds_bytes(3) <= 160 * 1024 ? 3 : 2;
  static constexpr int NB = D + 1, XSLOTS = NB * G_BS, NR = D + 2;
  static constexpr int OFF_A = T_WAVES * WREG;                 // [NB][G_BS][A_STEP]
  static constexpr int OFF_X = OFF_A + NB * G_BS * A_STEP;     // [quad][XSLOTS][32 samples][4 columns] doubles
  static constexpr int OFF_RAW = OFF_X + NCQ * XSLOTS * 1024;  // [NR][u | w][G_BS * 32] doubles
  static constexpr int OFF_F = OFF_RAW + NR * G_RAW;           // [3][unit][FU]: line (chunk-in-block, sample) x NPT factors
  static constexpr int OFF_FS = OFF_F + 3 * 2 * FU;            // [128] draws per replicate in the window
  static constexpr int LDS = OFF_FS + G_REPS * 4;
  static_assert(LDS == lds_bytes(D) && LDS <= 160 * 1024, "the instance fits the CU's LDS");
  static_assert(T_WAVES * WREG <= 65536, "X regions within the first 64 KiB (ds_write_addtid takes its base from M0[15:0])");
  static_assert(MAXS >= 1 && MAXS <= 3 && UF <= 2, "at most twelve accumulator tiles a wave");
};

template <int J0, int JN, bool WEIGHTED, int NCQ, bool BATCHED = false>
__global__ __launch_bounds__(T_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void resample_i8gn_kernel(
    const I8Args a_in, const int K, const unsigned char *__restrict__ table_in, const int64_t rep_begin, const int n_grp,
    const size_t table_state_stride) {
  static_assert(NCQ == 1 || NCQ == 2 || NCQ == 4, "column quads of a narrow state");
  static_assert(JN >= 1 && J0 + JN <= 8, "power range");
  using Geo = GnGeom<JN, NCQ, WEIGHTED>;
  constexpr int GS = Geo::GS, UF = Geo::UF, NQ = Geo::NQ, NPT = Geo::NPT, MAXS = Geo::MAXS, W = Geo::W;
  constexpr int D = Geo::D, NB = Geo::NB, XSLOTS = Geo::XSLOTS, NR = Geo::NR, A_STEP = Geo::A_STEP, FU = Geo::FU;
  constexpr int OFF_A = Geo::OFF_A, OFF_X = Geo::OFF_X, OFF_RAW = Geo::OFF_RAW, OFF_F = Geo::OFF_F, OFF_FS = Geo::OFF_FS;
  // DMA pieces a loader wave issues at the top of a block: the count words, the x of every quad and the raw u / w of its block steps
  constexpr int PIECES = (G_BS / GnRoles<JN, NCQ>::NLD) * (NQ + NCQ + (WEIGHTED ? 2 : 1));
  static_assert((D - 1) * PIECES < 64, "the block barrier's vmcnt");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *fsum = reinterpret_cast<uint32_t *>(lds + OFF_FS);

  typedef const __attribute__((address_space(4))) I8Args *const_args_p;
  auto pick_args = [&]() -> decltype(auto) {
    if constexpr (BATCHED) return (*(const_args_p)(uintptr_t)(a_in.batch_args + blockIdx.y));
    else return (a_in);
  };
  const auto &a = pick_args();
  const unsigned char *table = table_in + (BATCHED ? (size_t)blockIdx.y * table_state_stride : (size_t)0);

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int n32 = lane & 31, half = lane >> 5;
  const uint32_t wreg = (uint32_t)(wave * Geo::WREG);
  const int quad = wave % NCQ, g = wave / NCQ;  // the wave's column quad and its first power (relative to J0)
  // the wave's slots (all wave-uniform): nrows power rows, then the u-row fragment fu on wave 7 - fu
  const int nrows = g < JN ? (JN - 1 - g) / GS + 1 : 0;
  const bool has_ut = wave >= T_WAVES - UF;
  const int fu = has_ut ? T_WAVES - 1 - wave : 0;
  const int nslots = nrows + (has_ut ? 1 : 0);

  // ---- which window, which replicate group (the groups of a window share an XCD: b and b + 8 land on the same one)
  const int b = blockIdx.x;
  const int64_t win = (int64_t)((b >> 3) / n_grp) * 8 + (b & 7);
  const int grp = (b >> 3) % n_grp;
  if (win >= a.nwin) return;
  if (a.wflag[win] != 0u) return;  // precision guard: this window goes to the FP64 kernel
  const int64_t rep0 = rep_begin + (int64_t)grp * G_REPS;
  const int64_t WT = a.win_tiles;
  const int64_t t0 = win * WT;
  const int64_t t1 = t0 + WT < a.ntiles ? t0 + WT : a.ntiles;
  const int nsteps = (int)(t1 - t0) * T_STEPS;  // k-steps (32-sample chunks) of the window
  const int nblk = nsteps / G_BS;
  const unsigned char *tab = table + ((size_t)grp * (size_t)a.ntiles + (size_t)t0) * G_TILE_BYTES;  // the window in this group's table

  // ---- producer role: lane = (sample l >> 2 of a 16-sample unit, column l & 3 of the wave's quad)
  const int ps = lane >> 2, cl = lane & 3;
  const int col = 4 * quad + cl;
  const int ccol = col < a.C ? col : 0;  // columns >= C re-read column 0: their sums are never flushed
  // ---- consumer role (as resample_i8t_kernel)
  const uint32_t rd_off = wreg + (uint32_t)(((lane >> 4) & 1) * (T_PLANE + 128) + (16 * half + ((lane & 15) >> 1)) * 16 + (lane & 1) * 8);
  const int tcl = (n32 >> 2) & 3, tdg = 4 * (n32 >> 4) + (n32 & 3);
  const int um = 4 * fu + cl;             // this lane's u-row monomial
  const int umc = um < JN ? um : 0;       // (the unused slots of a short fragment repeat monomial 0: never flushed)
  // the factor a slot's word takes (index into a sample's line of NPT staged factors)
  auto slot_power = [&](int s) { return s < nrows ? g + s * GS : umc; };

  const double *wt = a.wtab + win * I8_WT_STRIDE;
  const double pu = a.pivot[0];
  const double inv_du = wt[I8_WT_INVDU];
  const double inv_w = WEIGHTED ? wt[I8_WT_INVW] : 1.0;
  const double sc = wt[I8_WT_SC + ccol];
  const double px = a.pivot[1 + a.col0 + ccol];

  v16i acc[MAXS][NQ];
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[s][q] = (v16i)(0);

  // first sample of chunk c of the window (the last tile of the series slides its window back)
  auto chunk_sample = [&](int c) -> int64_t {
    int64_t b0 = (t0 + (c >> 5)) * SM_T;
    if (b0 > a.N - SM_T) b0 = a.N - SM_T;
    return b0 + 32 * (c & 31);
  };
  // ---- requests
  constexpr int LOAD0 = GnRoles<JN, NCQ>::LOAD0, STAGE0 = GnRoles<JN, NCQ>::STAGE0, NLD = GnRoles<JN, NCQ>::NLD;
  constexpr int SPL = G_BS / NLD;                         // block steps per loader wave
  const bool loader = wave >= LOAD0 && wave < LOAD0 + NLD;  // uniform
  const int lw = wave - LOAD0;                            // this loader requests the pieces of block steps lw SPL .. lw SPL + SPL - 1
  int cq = 0;                    // chunk of the next x request (uniform)
  const char *xq = reinterpret_cast<const char *>(a.x + chunk_sample(0) * a.ldx_s);
  const int64_t xstep = 32 * a.ldx_s * 8;
  // chunk cq -> slot `slot` of every quad's ring (lane L fetches the 16-byte half L & 1 of row L >> 1 of the quad's four columns)
  // (every loader wave walks the chunks; loader lw requests the lw-th chunk of a block -- `mine`)
  auto x_request = [&](int slot, bool mine) {
    if (loader && mine) {
      const uint32_t ln = g_lane_now();
      const uint32_t row = (ln >> 1) * (uint32_t)(a.ldx_s * 8) + (ln & 1) * 16u + (uint32_t)(a.col0 * 8);
#pragma unroll
      for (int qd = 0; qd < NCQ; ++qd) {
        const int qsrc = 4 * qd < a.C ? qd : 0;  // quads past C re-read quad 0 (never flushed)
        g_dma16(xq, row + (uint32_t)(32 * qsrc), (uint32_t)(OFF_X + (qd * XSLOTS + slot) * 1024));
      }
    }
    ++cq;
    if (cq < nsteps) {
      if ((cq & 31) == 0) xq = reinterpret_cast<const char *>(a.x + chunk_sample(cq) * a.ldx_s);  // a new tile
      else xq += xstep;
    }
  };
  // count words of block B -> ring buffer B % NB: a loader wave the four 1-KiB pieces of each of its block steps
  auto a_request = [&](int B) {
    if (!loader) return;
    const int Bc = B < nblk ? B : nblk - 1;
    const uint32_t l16 = g_lane_now() * 16u;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int st = lw * SPL + k;
      const unsigned char *src = tab + (size_t)(Bc * G_BS + st) * G_KSTEP_BYTES;
      const uint32_t dst = (uint32_t)(OFF_A + (B % NB) * (G_BS * A_STEP) + st * A_STEP);
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q) g_dma16_stream(src + (size_t)Q * 1024, l16, dst + (uint32_t)(Q * 1024));  // (count words: read once)
    }
  };
  // raw u / w of factor block B (the chunks B * G_BS + W .. B * G_BS + W + G_BS - 1: what block B's k-steps slice)
  auto raw_request = [&](int B) {
    if (!loader) return;
    const uint32_t l4 = g_lane_now() * 4u;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int st = lw * SPL + k;
      int c = B * G_BS + W + st;
      if (c > nsteps - 1) c = nsteps - 1;
      const int64_t i0 = chunk_sample(c);
      g_dma4(a.u + i0, l4, (uint32_t)(OFF_RAW + (B % NR) * G_RAW + st * 256));
      if constexpr (WEIGHTED) g_dma4(a.w + i0, l4, (uint32_t)(OFF_RAW + (B % NR) * G_RAW + G_BS * 256 + st * 256));
    }
  };
  // factors of block B from raw buffer B % NR into factor buffer B % 3 (two waves, one sample per lane)
  auto stage_factors = [&](int B) {
    if (wave < STAGE0 || wave >= STAGE0 + 2) return;  // uniform
    const int e = (int)g_lane_now() + (wave - STAGE0) * 64;  // entry: chunk-in-block e >> 5, sample e & 31
    const double *raw = reinterpret_cast<const double *>(lds + OFF_RAW + (B % NR) * G_RAW);
    const double du = (raw[e] - pu) * inv_du;
    double pw = WEIGHTED ? raw[G_BS * 32 + e] * inv_w : 1.0;
    double *f = reinterpret_cast<double *>(lds + OFF_F + (B % 3) * (2 * FU) + ((e >> 4) & 1) * FU) + (((e >> 5) * 16 + (e & 15)) * NPT);
#pragma unroll
    for (int k = 0; k < J0; ++k) pw *= du;
#pragma unroll
    for (int jj = 0; jj < JN; ++jj) {
      f[jj] = pw;
      pw *= du;
    }
  };
  // ---- store the fixed-point words of the wave's two units of one slot (as resample_i8t_kernel: four 256-byte runs)
  auto store_x2 = [&](uint64_t bits0, uint64_t bits1, int off) {
    const uint32_t lo0 = (uint32_t)bits0 ^ 0x80808080u, hi0 = (uint32_t)(bits0 >> 32) ^ 0x00008080u;
    const uint32_t lo1 = (uint32_t)bits1 ^ 0x80808080u, hi1 = (uint32_t)(bits1 >> 32) ^ 0x00008080u;
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:%5\n\t"
                 "ds_write_addtid_b32 %1 offset:%6\n\t"
                 "ds_write_addtid_b32 %2 offset:%7\n\t"
                 "ds_write_addtid_b32 %3 offset:%8"
                 :
                 : "v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "s"(wreg), "n"(off), "n"(off + T_PLANE + 128), "n"(off + 256),
                   "n"(off + 256 + T_PLANE + 128)
                 : "memory", "m0");
  };
  typedef __attribute__((address_space(3))) const double *lds_cd;
  typedef __attribute__((address_space(3))) const v4i *lds_cv4;
  // the words of one chunk into region `R` (a step of the pair) of the wave's slots: dx of the lane's two samples, the factors
  // `fac(slot, unit)`
  auto produce = [&](auto rc, const double (&dx)[2], auto &&fac) {
    constexpr int R = decltype(rc)::value;
    t_static_for<MAXS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (s < nslots) {  // wave-uniform
        const bool row = s < nrows;
        const double m0 = row ? dx[0] : 0x1p50, m1 = row ? dx[1] : 0x1p50;
        store_x2((uint64_t)__double_as_longlong(fma(fac(s, 0), m0, T_MAGIC)),
                 (uint64_t)__double_as_longlong(fma(fac(s, 1), m1, T_MAGIC)), (R * MAXS + s) * T_PB);
      }
    });
  };

  // ================= prologue =================
  // draws per replicate in the window (the top digit's bias is removed with them at the flush)
  if (threadIdx.x < G_REPS) {
    const int64_t r = rep0 + threadIdx.x;
    uint32_t s = 0;
    if (r < a.nrep)
      for (int64_t t = t0; t < t1; ++t) s += a.counts[(size_t)r * a.ntiles + t];
    fsum[threadIdx.x] = s;
  }
  // zero the X regions once (the padding between the planes is never written)
  for (int e = threadIdx.x; e < T_WAVES * Geo::WREG / 16; e += T_BLOCK) reinterpret_cast<uint4 *>(lds)[e] = make_uint4(0, 0, 0, 0);
  // the first W chunks' u / w straight from memory (the direct path of resample_i8t_kernel)
  double d_du[W][2], d_w[W][2];
#pragma unroll
  for (int r = 0; r < W; ++r) {
    const int64_t i0 = chunk_sample(r);
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      d_du[r][uu] = (a.u[i0 + 16 * uu + ps] - pu) * inv_du;
      d_w[r][uu] = 1.0;
      if constexpr (WEIGHTED) d_w[r][uu] = a.w[i0 + 16 * uu + ps] * inv_w;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing of the compiler's is in flight behind the DMAs below)
  // blocks 0 .. D - 1 (and the raw u / w of blocks 0 .. D); chunk c lives in slot (c - W) mod XSLOTS: the first W chunks in the
  // last slots, block B's chunks 4 B + W .. 4 B + W + 3 in slots 4 (B mod NB) ...
#pragma unroll
  for (int r = 0; r < W; ++r) x_request(XSLOTS - W + r, lw == 0);  // (W more pieces per quad on loader 0: the prologue waits for everything)
#pragma unroll
  for (int Bq = 0; Bq < D; ++Bq) {
    raw_request(Bq);
    a_request(Bq);
#pragma unroll
    for (int c = 0; c < G_BS; ++c) x_request(Bq * G_BS + c, lw == c / SPL);
  }
  raw_request(D);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  stage_factors(0);
  const uint32_t x_va0 = (uint32_t)(OFF_X + quad * XSLOTS * 1024 + ps * 32 + cl * 8);
  // the X words of the first W chunks (no matrix work yet): powers multiplied up from du / w of the direct path
  t_static_for<W>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    double dx[2];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) dx[uu] = (*(lds_cd)(lds + x_va0 + (XSLOTS - W + r) * 1024 + uu * 512) - px) * sc;
    produce(rc, dx, [&](int s, int uu) {
      double pw = d_w[r][uu];
      if constexpr (!WEIGHTED) asm volatile("" : "+v"(pw));  // (opaque: the staged arithmetic, two roundings -- see resample_i8t_kernel)
      const int n = J0 + slot_power(s);
      for (int q = 0; q < n; ++q) pw *= d_du[r][uu];
      return pw;
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // factors of block 0 visible; the first chunks' x slots free

  // ================= the blocks =================
  // Step s = B * 4 + p contracts chunk s (count words: ring buffer B % NB, X words: region p % W of the wave) and slices chunk s + W
  // into the same region (x from ring slot (B % NB) * 4 + p, factors from buffer B % 3 line p).  The loader waves request block
  // B + D's count words and x, and block B + D + 1's raw u / w, at the top of block B; two waves stage block B + 1's factors there.
  // ONE barrier per block, behind its last step: what was requested D - 1 block tops ago has landed (every loader waited), the
  // staged factors are visible, and nobody reads this block's ring parts any more.
  uint32_t *pg = a.progress != nullptr ? a.progress + (size_t)win * 16 : nullptr;
#ifdef TXM_GN_TIMING  // diagnostic build: cycles of one workgroup's waves in the block tops, the k-steps and at the barrier
  long long tm_top = 0, tm_steps = 0, tm_bar = 0, tk = clock64();
#define GN_TICK(v) do { const long long t1_ = clock64(); v += t1_ - tk; tk = t1_; } while (0)
#else
#define GN_TICK(v) do {} while (0)
#endif
  // per-lane byte offset of a slot's factor inside a sample's line
  uint32_t f_slot[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) f_slot[s] = (uint32_t)(slot_power(s) * 8);
#pragma unroll 1
  for (int B = 0; B < nblk; ++B) {
    if (pg != nullptr && wave == 0 && (B & 7) == 0) {  // L2-sharing hint (uniform; bounded, no result depends on it)
      const uint32_t done = (uint32_t)(B >> 3) + 1u;
      if (lane == 0) __hip_atomic_store(&pg[grp & 15], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
      for (int spin = 0; spin < I8_THROTTLE_SPINS; ++spin) {
        if (done <= g_min_progress(pg) + G_LEAD) break;
        __builtin_amdgcn_s_sleep(32);
      }
    }
    stage_factors(B + 1);
    raw_request(B + D + 1);
    a_request(B + D);
#pragma unroll
    for (int i = 0; i < G_BS; ++i) x_request(((B + D) % NB) * G_BS + i, lw == i / SPL);
    GN_TICK(tm_top);
    const uint32_t a_va = (uint32_t)(OFF_A + (B % NB) * (G_BS * A_STEP)) + (uint32_t)lane * 16u;
    const uint32_t f_va = (uint32_t)(OFF_F + (B % 3) * (2 * FU) + ps * NPT * 8);
    const uint32_t x_va = x_va0 + (uint32_t)((B % NB) * G_BS * 1024);
    if (nslots > 0) {  // wave-uniform: a wave without a slot only does the block's chores
      if constexpr (W == 1) {
        // twelve tiles: a step's count words and x at its top, each slot's words and factors where the slot takes them
        t_static_for<G_BS>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          v4i A[NQ];
          double dx[2];
#pragma unroll
          for (int q = 0; q < NQ; ++q) A[q] = *(lds_cv4)(lds + a_va + p * A_STEP + q * 1024);
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) dx[uu] = (*(lds_cd)(lds + x_va + p * 1024 + uu * 512) - px) * sc;
          t_static_for<MAXS>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            if (s < nslots) {  // wave-uniform
              const v2i Ba = T_TRREAD((lds_v2i)(lds + rd_off + s * T_PB)), Bb = T_TRREAD((lds_v2i)(lds + rd_off + s * T_PB + 128));
              const double f0 = *(lds_cd)(lds + f_va + f_slot[s] + p * 16 * NPT * 8), f1 = *(lds_cd)(lds + f_va + f_slot[s] + FU + p * 16 * NPT * 8);
              const v4i Bv = {Ba[0], Ba[1], Bb[0], Bb[1]};
#pragma unroll
              for (int q = 0; q < NQ; ++q) t_mfma<true>(acc[s][q], A[q], Bv);
              const bool row = s < nrows;
              store_x2((uint64_t)__double_as_longlong(fma(f0, row ? dx[0] : 0x1p50, T_MAGIC)),
                       (uint64_t)__double_as_longlong(fma(f1, row ? dx[1] : 0x1p50, T_MAGIC)), s * T_PB);
            }
          });
        });
      } else
      t_static_for<G_BS / W>([&](auto gc) {
        constexpr int p0 = decltype(gc)::value * W;
        // every operand of the W steps up front: the X words stored a pair ago, the count words, and what the next slicing takes
        v2i Ba[W][MAXS], Bb[W][MAXS];
        v4i A[W][NQ];
        double fr[W][MAXS][2], xr[W][2];
#pragma unroll
        for (int r = 0; r < W; ++r)
#pragma unroll
          for (int s = 0; s < MAXS; ++s)
            if (s < nslots) {
              Ba[r][s] = T_TRREAD((lds_v2i)(lds + rd_off + (r * MAXS + s) * T_PB));
              Bb[r][s] = T_TRREAD((lds_v2i)(lds + rd_off + (r * MAXS + s) * T_PB + 128));
            }
#pragma unroll
        for (int r = 0; r < W; ++r)
#pragma unroll
          for (int q = 0; q < NQ; ++q) A[r][q] = *(lds_cv4)(lds + a_va + (p0 + r) * A_STEP + q * 1024);
        // (always the staged factor, also the 1.0 of an unweighted power 0 -- as the quad-sharing variant of resample_i8t_kernel
        // does: with the constant visible the compiler turns fma(1, dx, magic) into dx + magic and contracts it with dx's own
        // multiply -- ONE rounding where the other kernel has two, and the words of a few per cent of the samples differ by a unit)
#pragma unroll
        for (int r = 0; r < W; ++r) {
#pragma unroll
          for (int s = 0; s < MAXS; ++s)
            if (s < nslots) {
#pragma unroll
              for (int uu = 0; uu < 2; ++uu) fr[r][s][uu] = *(lds_cd)(lds + f_va + f_slot[s] + uu * FU + (p0 + r) * 16 * NPT * 8);
            }
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) xr[r][uu] = *(lds_cd)(lds + x_va + (p0 + r) * 1024 + uu * 512);
        }
        t_static_for<W>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          t_static_for<MAXS>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            if (s < nslots) {  // wave-uniform
              const v4i Bv = {Ba[r][s][0], Ba[r][s][1], Bb[r][s][0], Bb[r][s][1]};
#pragma unroll
              for (int q = 0; q < NQ; ++q) t_mfma<true>(acc[s][q], A[r][q], Bv);
            }
          });
        });
        // the words of chunks s + W: behind the MFMAs that took chunk s's (their B operands are in registers by then)
        t_static_for<W>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          double dx[2];
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) dx[uu] = (xr[r][uu] - px) * sc;
          produce(rc, dx, [&](int s, int uu) { return fr[r][s][uu]; });
        });
      });
    }
    // the block's barrier: block B + 1's count words, x and block B + 2's raw u / w -- requested at the top of block B + 1 - D -- have
    // landed when at most the pieces of the D - 1 block tops since then are in flight (a wave's DMAs complete in order)
    GN_TICK(tm_steps);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * PIECES) : "memory");
    GN_TICK(tm_bar);
  }

#ifdef TXM_GN_TIMING
  if (lane == 0 && blockIdx.x == 16 && blockIdx.y == 0)
    printf("wave %d  blocks %d  top %lld  steps %lld  barrier %lld cycles per block\n", wave, nblk, tm_top / nblk, tm_steps / nblk, tm_bar / nblk);
#endif
  // ================= flush: int32 accumulators of the window -> its slot of the partial sums =================
  // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
  if (pg != nullptr && threadIdx.x == 0) __hip_atomic_store(&pg[grp & 15], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // The slot of a window holds ONE double per (replicate, power, column) -- the seven digit sums of an element added up here,
  // ((((((d0 + d1) + d2) + d3) + d4) + d5) + d6) of (double)(int32 sum) x scale: the expression and the order
  // resample_finalize_i8_kernel applies to per-digit slots, and what the chunk-group instances of resample_i8t_kernel store (same
  // bits; an eighth of the bytes written here and read by the finalize: config 2's finalize 0.10 -> 0.03 ms).  A tile goes through a
  // wave-private scratch in the (now idle) count ring -- rows of 65 words, see resample_i8t_kernel -- because an element's digits sit
  // in seven different lanes: digit d of (replicate row m, column q4) in register 4 (m >> 3) + (m & 3) of lane
  // 16 (d >> 2) + 4 q4 + (d & 3) + 32 ((m >> 2) & 1).
  constexpr int XS = 65, XT = 16 * XS;
  static_assert(T_WAVES * XT * 4 <= Geo::NB * G_BS * Geo::A_STEP, "flush scratch inside the count ring");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // no DMA piece in flight, no wave still reading the rings
  uint32_t *xw = reinterpret_cast<uint32_t *>(lds + Geo::OFF_A) + wave * XT;
  auto flush_tile = [&](v16i &T, int q, int s) {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));  // opaque zero: the addresses are formed where they are used, not hoisted and spilled
    const int64_t opq = (int64_t)z;
#pragma unroll
    for (int r = 0; r < 16; ++r) xw[r * XS + lane] = (uint32_t)T[r];
    // (the wave's own LDS operations execute in order: the reads below see the stores above)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = lane + 64 * i, m = o >> 2, q4 = o & 3;
      const int rr = ((m >> 3) << 2) | (m & 3), hh = (m >> 2) & 1;
      const int64_t rrow = rep0 + 32 * q + m;
      bool valid = rrow < a.nrep;
      int j;
      double dsc0;
      double *dst;
      if (s < nrows) {  // a power row
        const int c = 4 * quad + q4;
        valid = valid && c < a.C;
        j = J0 + g + s * GS;
        dsc0 = wt[I8_WT_DSP + j] * wt[I8_WT_DSC + (c < a.C ? c : 0)];
        dst = a.part_x + (((size_t)win * a.nrep_pad + rrow) * K + j) * a.cpad + c + opq;
      } else {  // the u-row fragment
        const int mm = 4 * fu + q4;
        valid = valid && mm < JN;
        j = J0 + (mm < JN ? mm : 0);
        dsc0 = wt[I8_WT_DSP + j] * 0x1p-50;
        dst = a.part_u + ((size_t)win * a.nrep_pad + rrow) * K + j + opq;
      }
      if (valid) {
        const uint32_t *src = xw + rr * XS + 32 * hh + 4 * q4;
        int v[I8_NSL];
#pragma unroll
        for (int d = 0; d < I8_NSL; ++d) v[d] = (int)src[16 * (d >> 2) + (d & 3)];
        v[I8_NSL - 1] -= T_D6_BIAS * (int)fsum[32 * q + m];
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
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < nslots) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < NQ; ++q) flush_tile(acc[s][q], q, s);
    }
}

// ---------------------------------------------------------------------------
template <int J0, int JN, bool WEIGHTED, int NCQ>
static int launch_pass_gn(const I8Args &a, int K, const unsigned char *table, int64_t rep_begin, int n_grp, size_t table_state_stride,
                          hipStream_t st) {
  const size_t lds = (size_t)GnGeom<JN, NCQ, WEIGHTED>::LDS;
  const int64_t S = a.states != nullptr ? a.S : 1;
  const dim3 grid((unsigned)(cdiv(a.nwin, 8) * 8 * n_grp), (unsigned)S);
  if (a.progress != nullptr) TXM_HIP(hipMemsetAsync(a.progress, 0, (size_t)cdiv(a.nwin, 8) * 8 * 16 * sizeof(uint32_t), st));
  if (a.states != nullptr) {
    TXM_SET_MAX_LDS((&resample_i8gn_kernel<J0, JN, WEIGHTED, NCQ, true>), lds);
    hipLaunchKernelGGL((resample_i8gn_kernel<J0, JN, WEIGHTED, NCQ, true>), grid, dim3(T_BLOCK), lds, st, a, K, table, rep_begin, n_grp,
                       table_state_stride);
  } else {
    TXM_SET_MAX_LDS((&resample_i8gn_kernel<J0, JN, WEIGHTED, NCQ, false>), lds);
    hipLaunchKernelGGL((resample_i8gn_kernel<J0, JN, WEIGHTED, NCQ, false>), grid, dim3(T_BLOCK), lds, st, a, K, table, rep_begin, n_grp,
                       table_state_stride);
  }
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// what the narrow table kernel takes: whole column quads readable inside a row through 16-byte DMA pieces, orders 1 .. 7
bool i8gn_applicable(const double *x, int64_t ldx_s, int64_t C, int K) {
  if (C < 1 || C > 16 || K < 2 || K > 8) return false;
  const int64_t cq = (C + 3) / 4 * 4;
  if (((uintptr_t)x & 15) != 0 || (ldx_s & 1) != 0 || cq > ldx_s) return false;
  return true;
}

// the passes of a narrow state over the count table of replicate groups [rep_begin, rep_begin + 128 n_grp): a wave holds at most
// three power row sets (two where it also carries a u-row fragment), so one quad takes every order in one pass, two quads
// orders 1..7 in one pass (order 7: two), four quads orders 1..3 in one pass and 4..7 in two
template <int NCQ>
static int launch_gn_t(const I8Args &a, int K, bool weighted, const unsigned char *table, int64_t rep_begin, int n_grp,
                       size_t table_state_stride, hipStream_t st) {
#define GN_PASS(J0_, JN_) (weighted ? launch_pass_gn<J0_, JN_, true, NCQ>(a, K, table, rep_begin, n_grp, table_state_stride, st) \
                                    : launch_pass_gn<J0_, JN_, false, NCQ>(a, K, table, rep_begin, n_grp, table_state_stride, st))
  // powers a pass can take: the u-row waves (the last ceil(JN / 4)) hold NSW + 1 row sets <= 3
  constexpr int GS = T_WAVES / NCQ;
  constexpr int JMAX = 2 * GS < 8 ? 2 * GS : 8;  // NSW <= 2 everywhere: every wave may carry a u-row fragment
  int rc = TXM_OK;
  if (K <= JMAX) {
    switch (K) {
      case 2: return GN_PASS(0, 2);
      case 3: return GN_PASS(0, 3);
      case 4: return GN_PASS(0, 4);
      case 5: if constexpr (JMAX >= 5) return GN_PASS(0, 5); break;
      case 6: if constexpr (JMAX >= 6) return GN_PASS(0, 6); break;
      case 7: if constexpr (JMAX >= 7) return GN_PASS(0, 7); break;
      case 8: if constexpr (JMAX >= 8) return GN_PASS(0, 8); break;
      default: break;
    }
  } else if constexpr (JMAX == 4) {  // four quads: 4 + (K - 4) powers
    rc = GN_PASS(0, 4);
    if (rc != TXM_OK) return rc;
    switch (K) {
      case 5: return GN_PASS(4, 1);
      case 6: return GN_PASS(4, 2);
      case 7: return GN_PASS(4, 3);
      case 8: return GN_PASS(4, 4);
      default: break;
    }
  }
#undef GN_PASS
  set_error("resample_i8gn: order out of range");
  return TXM_ERR_INVALID;
}

int launch_resample_i8gn(const I8Args &a, int K, bool weighted, const unsigned char *table, int64_t rep_begin, int n_grp,
                         size_t table_state_stride, hipStream_t st) {
  const int nq = i8t_narrow_nq(a.C_call, K);
  if (nq == 1) return launch_gn_t<1>(a, K, weighted, table, rep_begin, n_grp, table_state_stride, st);
  if (nq == 2) return launch_gn_t<2>(a, K, weighted, table, rep_begin, n_grp, table_state_stride, st);
  if (nq == 4) return launch_gn_t<4>(a, K, weighted, table, rep_begin, n_grp, table_state_stride, st);
  set_error("resample_i8gn: not a narrow state");
  return TXM_ERR_INVALID;
}

}  // namespace txm
