// txm_resample_i8g.hip -- the int8 bootstrap contraction WITHOUT a sampler inside (round 5).  Same sums, same fixed-point
// words, same partial-sum slots as txm_resample_i8t.hip (cmomy.wrap_resample_vals as called from thermoextrap
// data.py:1803-1810, 1354-1366):
//        S1[r][c][j] = sum_i f[r][i] w_i du_i^j dx_ic        S0[r][j] = sum_i f[r][i] w_i du_i^j
// but the counts f come from a table in HBM that count_table_kernel (txm_count_table.hip) wrote in MFMA-A-operand order.
//
// Why.  In resample_i8t_kernel 36 % of all vector instructions were the stage-3 sampler fill (Philox + ds_add), the
// 64 KiB count tile held the workgroup at 64 replicates, and every sliced B operand fed two MFMAs.  Measured on gfx950
// (tools/mfma_i8_probe3.hip, profiles/r05_experiments.md): a v_mfma_i32_32x32x32_i8 hides about four vector
// instructions issued beside it (38.5 -> 41 cycles per slot) and charges ~5 cycles for each one beyond that -- the old
// kernel ran at 9.1 vector instructions per MFMA.  Here:
//   * workgroup = 8 waves x 128 replicates x 32 columns x at most THREE row sets (powers J0 .. J0 + JN - 1, plus the
//     second matrix's row set): wave w owns column quad w, 3 x 4 accumulator tiles (192 registers); every sliced operand
//     feeds FOUR MFMAs; orders above 2 take several passes, all over ONE count table (a pass re-reads x and the table).
//   * the u-row (S0) rides in the dead eighth byte of the fixed-point words: digit d of w du^j sits in byte 7 of column
//     (d & 3) of quad 2 fi + (d >> 2) of row set fi -- no u-row tiles, no u-row MFMAs.
//   * no global load lands in a register inside the k-steps: counts, x, y, u and w arrive by LDS-DMA
//     (global_load_lds) with explicit s_waitcnt vmcnt counts -- the count words of a block of 4 k-steps into a
//     double-buffered ring shared by the eight waves (one s_barrier per block), a wave's own x columns into a private
//     ring three k-steps ahead, u and w two blocks ahead; the staged factors w du^j of a block are computed from them by
//     two waves while the block before runs.
//   * a pass of ONE row set (order 0) has registers to spare and takes TWO replicate groups per workgroup (256 replicates,
//     eight tiles per wave): 74.4 -> 62.2 ms per call at N = 1e8, nrep = 1000.
//   * grid = scaling windows x replicate groups; the int32 sums of a window are exact, the flush writes the same
//     doubles into the same slots as resample_i8t_kernel: the two kernels agree BIT FOR BIT.
#include "txm_i8g.h"

namespace txm {

constexpr int G_XR = 4;  // chunks in a wave's x ring when x is requested step by step (second-matrix passes; = G_BS)
static_assert(G_XR == G_BS && G_BS == 4, "the wait counts below are written for blocks of four k-steps");
// ... and when a block's four x chunks are requested together with its count words (passes without a second matrix): eight.
// Why: on gfx950 a wave's DS instructions queue BEHIND its outstanding LDS-DMA pieces -- a ds_read issued after a
// global_load_lds waited ~800-1100 cycles for it (phase clocks of the first cut, profiles/r05_experiments.md) -- so one x piece
// per k-step stalled every step's LDS reads for an L2 round trip.  One DMA event per block instead of five.
constexpr int G_XRB = 8;
constexpr int G_FU = 2176;           // bytes between the factor lines of a lane's two 16-sample units (> 2040: no ds_read2 pairing)

// replicate QUARTERS (32 replicates = one MFMA row block = one 1-KiB piece of a table k-step) per workgroup: four (128 replicates),
// and eight in a pass of ONE row set (order 0), which has the registers for eight accumulator tiles per wave: every x chunk, sliced
// word and B operand then serves twice the replicates (74.4 -> 62.2 ms per call at the north-star size).  A workgroup's quarters
// are addressed piece by piece and may straddle the table's 128-replicate groups -- six quarters in passes of two row sets (twelve
// tiles, like three row sets at four) were built and measured: 7 % faster per replicate, but 1000 replicates are 32 quarters =
// 5.33 workgroups of six, and the padding costs more: order 1 88.7 against 86.3 ms, order 3 146.9 / 141.9 (-DTXM_G_SIX_QUARTERS).
#ifdef TXM_G_FOUR_QUARTERS  // (A/B build: 128 replicates per workgroup in every pass)
template <int NS> constexpr int G_QUARTERS = 4;
#elif defined(TXM_G_SIX_QUARTERS)  // (A/B build)
template <int NS> constexpr int G_QUARTERS = NS == 1 ? 8 : NS == 2 ? 6 : 4;
#else
template <int NS> constexpr int G_QUARTERS = NS == 1 ? 8 : 4;
#endif

// (in the six-quarter A/B build one instance does not fit its registers -- weighted order 0 with a second matrix -- and stays at four)
template <int J0, int JN, bool WEIGHTED, bool YS>
constexpr int G_NQ = (J0 == 0 && JN == 1 && WEIGHTED && YS) ? 4 : G_QUARTERS<JN + (YS ? 1 : 0)>;

template <int J0, int JN, bool WEIGHTED, bool YS>
__global__ __launch_bounds__(T_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void resample_i8g_kernel(
    const I8Args a, const int K, const unsigned char *__restrict__ table, const int64_t rep_begin, const int n_grp) {
  constexpr int NS = JN + (YS ? 1 : 0);  // row sets of the pass
  static_assert(JN >= 1 && NS <= 3 && J0 + JN <= 8, "row sets");
  constexpr int NPT = JN + ((YS && WEIGHTED && J0 > 0) ? 1 : 0);  // staged factors per sample (the y row set needs plain w)
  constexpr int NX = YS ? 2 : 1;                                 // x-ring DMAs per k-step
  constexpr int WREG = NS * T_PB;
  constexpr int NQ = G_NQ<J0, JN, WEIGHTED, YS>;                  // replicate quarters (A operands, accumulator tiles per row set)
  constexpr int A_STEP = NQ * 1024;                               // count words of one k-step in the ring: [quarter][1024]
  constexpr int PPW = G_BS * NQ / T_WAVES;                        // count pieces per wave and block when every wave requests its own
  static_assert(G_BS * NQ % T_WAVES == 0, "count pieces per wave");
  constexpr int OFF_A = T_WAVES * WREG;                           // [2][G_BS][A_STEP] count words
  constexpr bool XBLK = !YS;                                      // x requested per block (8-slot ring) / per step (4 slots, + y)
#ifdef TXM_G_ADIR
  // (experiment, round 6) the count words of a two-row-set pass straight from global memory into registers: the table is in
  // MFMA-A-operand order already, a pass of two row sets has 64 registers to spare -- A[step of the block][quarter], reloaded for
  // the NEXT block right behind the operand's last MFMA (one block = four k-steps of lead).  No count ring in the LDS (32 of the
  // 100 KiB a k-step moves through it), no count pieces among the loader waves' DMA (16 of 50 a block).
  constexpr bool ADIR = XBLK && NS == 2;
#else
  constexpr bool ADIR = false;
#endif
  constexpr int XRN = XBLK ? G_XRB : G_XR;
  constexpr int OFF_X = OFF_A + 2 * G_BS * A_STEP;                // [wave][XRN][32 samples][4 columns] doubles
  constexpr int OFF_Y = OFF_X + T_WAVES * XRN * 1024;
  constexpr int OFF_RAW = OFF_Y + (YS ? T_WAVES * XRN * 1024 : 0);  // [3][u | w][G_BS * 32] doubles
  constexpr int OFF_F = OFF_RAW + 3 * G_RAW;                      // [3][unit][G_FU]: factors, line (chunk-in-block, sample) x NPT
  constexpr int OFF_FS = OFF_F + 3 * 2 * G_FU;                    // [32 NQ] draws per replicate in the window
  static_assert(G_BS * 16 * NPT * 8 <= G_FU, "factor lines");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *fsum = reinterpret_cast<uint32_t *>(lds + OFF_FS);

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int n32 = lane & 31, half = lane >> 5;
  const uint32_t wreg = (uint32_t)(wave * WREG);

  // ---- which window, which replicate group (the groups of a window share an XCD: b and b + 8 land on the same one)
  const int b = blockIdx.x;
  const int n_wgg = (4 * n_grp + NQ - 1) / NQ;  // replicate groups of the grid (NQ quarters each; the table's groups hold 4)
  const int64_t win = (int64_t)((b >> 3) / n_wgg) * 8 + (b & 7);
  const int grp = (b >> 3) % n_wgg;
  if (win >= a.nwin) return;
  if (a.wflag[win] != 0u) return;  // precision guard: this window goes to the FP64 kernel
  const int64_t rep0 = rep_begin + (int64_t)grp * (32 * NQ);
  const int64_t WT = a.win_tiles;
  const int64_t t0 = win * WT;
  const int64_t t1 = t0 + WT < a.ntiles ? t0 + WT : a.ntiles;
  const int nsteps = (int)(t1 - t0) * T_STEPS;  // k-steps (32-sample chunks) of the window
  const int nblk = nsteps / G_BS;
  const unsigned char *tab = table + (size_t)t0 * G_TILE_BYTES;  // the window in table group 0
  // quarter Q of the workgroup = quarter (grp NQ + Q) of the call: piece (that & 3) of table group (that >> 2).  Past the call's
  // last quarter (a last workgroup that is not full) the first quarter's words are read again; those rows -- all past nrep --
  // are not flushed
  auto q_off = [&](int Q) -> size_t {
    int gq = grp * NQ + Q;
    if (gq >= 4 * n_grp) gq = grp * NQ;
    return (size_t)(gq >> 2) * (size_t)a.ntiles * G_TILE_BYTES + (size_t)(gq & 3) * 1024;
  };

  // ---- producer role: lane = (sample l >> 2 of a 16-sample unit, column l & 3 of the wave's quad)
  const int ps = lane >> 2, cl = lane & 3;
  const int col = 4 * wave + cl;
  const int ccol = col < a.C ? col : 0;
  const int qsrc = 4 * wave < a.C ? wave : 0;  // quads past C re-read quad 0 (their sums are never flushed)
  // ---- consumer role (as resample_i8t_kernel)
  const uint32_t rd_off = wreg + (uint32_t)(((lane >> 4) & 1) * (T_PLANE + 128) + (16 * half + ((lane & 15) >> 1)) * 16 + (lane & 1) * 8);
  const int tcl = (n32 >> 2) & 3, tdg = 4 * (n32 >> 4) + (n32 & 3);
  // ---- the u-row overlay: waves 2 fi, 2 fi + 1 carry the digits of row set fi's monomial w du^j in byte 7 of their words
  const int ofi = wave >> 1;                        // the row set this wave overlays (if < JN)
  const int odig = 4 * (wave & 1) + cl;             // this lane's digit (7: none)
  // v_perm selector: byte odig of {hi, lo} into byte 3, zeros below; digits 0..5 come biased by 0x80
  const uint32_t osel = odig < 7 ? (((uint32_t)odig << 24) | 0x000c0c0cu) : 0x0c0c0c0cu;
  const uint32_t oxor = odig < 6 ? 0x80000000u : 0u;

  const double *wt = a.wtab + win * I8_WT_STRIDE;
  const double pu = a.pivot[0];
  const double inv_du = wt[I8_WT_INVDU];
  const double inv_w = WEIGHTED ? wt[I8_WT_INVW] : 1.0;
  const double sc = wt[I8_WT_SC + ccol];
  const double px = a.pivot[1 + a.col0 + ccol];
  const double *wty = YS ? a.ywtab + win * I8_WT_STRIDE : wt;
  const double scy = YS ? wty[I8_WT_SC + ccol] : 0.0;
  const double py = YS ? a.ypivot[1 + a.col0 + ccol] : 0.0;

  v16i acc[NS][NQ];
#pragma unroll
  for (int e = 0; e < NS; ++e)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[e][q] = (v16i)(0);

#ifdef TXM_G_CLOCKS  // diagnostic build: shader cycles (s_memtime) against the 100 MHz reference clock (s_memrealtime) over the kernel
  const long long gc0 = clock64(), gr0 = wall_clock64();
#endif
#ifdef TXM_G_TIMING  // diagnostic build: cycles per phase of two workgroups (tools/i8g_timing.py)
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk0 = clock64();
#define G_TICK(k) do { const long long t1_ = clock64(); tm[k] += t1_ - tk0; tk0 = t1_; } while (0)
#else
#define G_TICK(k) do {} while (0)
#endif
  // first sample of chunk c of the window (the last tile of the series slides its window back)
  auto chunk_sample = [&](int c) -> int64_t {
    int64_t b0 = (t0 + (c >> 5)) * SM_T;
    if (b0 > a.N - SM_T) b0 = a.N - SM_T;
    return b0 + 32 * (c & 31);
  };
  // ---- x ring: the wave's own four columns of a chunk, [32 samples][4 columns] doubles, lane L fetches the 16-byte half
  // (L & 1) of row L >> 1
  const uint32_t xring = (uint32_t)(OFF_X + wave * XRN * 1024), yring = (uint32_t)(OFF_Y + wave * XRN * 1024);
  int cq = 0;  // chunk of the next x request (uniform)
  const char *xq = reinterpret_cast<const char *>(a.x + chunk_sample(0) * a.ldx_s);
  const char *yq = YS ? reinterpret_cast<const char *>(a.y + chunk_sample(0) * a.ldy_s) : nullptr;
  const int64_t xstep = 32 * a.ldx_s * 8, ystep = YS ? 32 * a.ldy_s * 8 : 0;
  // Who issues the DMA pieces (passes without a second matrix): the OLDER wave of every SIMD -- waves 0..3; wave w + 4 shares
  // wave w's SIMD and loses the issue arbitration to it, so the younger waves set the pace of a block (phase clocks: step
  // bodies 917 against 713 cycles) while the older ones wait a quarter of their time at the barrier.  The older waves therefore
  // request everything (a piece costs its issuer ~100 cycles: 56 pieces per block and CU), also their partner's x columns, and
  // stage the factors; landing is published by the block's barrier, behind every issuer's vmcnt(0).
  const bool loader = !XBLK || wave < 4;  // uniform
  const int qsrc2 = 4 * (wave + 4) < a.C ? wave + 4 : 0;
  const int64_t xcol1 = (a.col0 + 4 * qsrc) * 8, xcol2 = (a.col0 + 4 * qsrc2) * 8;  // byte offsets of the two column quads in a row
  size_t qoff[NQ];  // (uniform: scalar registers)
#pragma unroll
  for (int Q = 0; Q < NQ; ++Q) qoff[Q] = q_off(Q);
  auto x_request = [&](int slot) {  // chunk cq -> ring slot; chunks past the window re-read its last one
#ifdef TXM_G_NO_XDMA  // ablation build
    if (slot >= 0) { ++cq; return; }
#endif
    if constexpr (XBLK) {
      if (loader) {
        const uint32_t ln = g_lane_now();
        const uint32_t row = (ln >> 1) * (uint32_t)(a.ldx_s * 8) + (ln & 1) * 16u + (uint32_t)(a.col0 * 8);
        g_dma16(xq, row + (uint32_t)(32 * qsrc), xring + (uint32_t)slot * 1024u);
        g_dma16(xq, row + (uint32_t)(32 * qsrc2), xring + (uint32_t)(4 * XRN * 1024) + (uint32_t)slot * 1024u);  // wave + 4's ring
      }
    } else {
      // (offsets recomputed from the lane id: held across the k-steps they were spilled in the second-matrix instances, and a
      // scratch reload's s_waitcnt vmcnt -- the wave's DMA pieces complete in order with it -- drained the x / y prefetch every step)
      const uint32_t ln = g_lane_now();
      const uint32_t cb = (ln & 1) * 16u + (uint32_t)((a.col0 + 4 * qsrc) * 8);
      g_dma16(xq, (ln >> 1) * (uint32_t)(a.ldx_s * 8) + cb, xring + (uint32_t)slot * 1024u);
      if constexpr (YS) g_dma16(yq, (ln >> 1) * (uint32_t)(a.ldy_s * 8) + cb, yring + (uint32_t)slot * 1024u);
    }
    ++cq;
    if (cq < nsteps) {
      if ((cq & 31) == 0) {  // a new tile (the slid last one does not follow its predecessor in memory)
        const int64_t i0 = chunk_sample(cq);
        xq = reinterpret_cast<const char *>(a.x + i0 * a.ldx_s);
        if constexpr (YS) yq = reinterpret_cast<const char *>(a.y + i0 * a.ldy_s);
      } else {
        xq += xstep;
        if constexpr (YS) yq += ystep;
      }
    }
  };
  // ---- count words of block B -> ring buffer B & 1: the wave's two 1-KiB pieces
  auto a_request = [&](int B, int piece = -1) {
    const int Bc = B < nblk ? B : nblk - 1;
#ifdef TXM_G_NO_ADMA  // ablation build
    if (Bc >= 0) return;
#endif
    if constexpr (XBLK) {  // loader wave w: the NQ 1-KiB pieces of the block's k-step w (piece >= 0: that one only)
      if (!loader) return;
      const unsigned char *src = tab + (size_t)(Bc * G_BS + wave) * G_KSTEP_BYTES;
      const uint32_t dst = (uint32_t)(OFF_A + (B & 1) * (G_BS * A_STEP) + wave * A_STEP);
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q)
        if (piece < 0 || piece == Q) g_dma16_stream(src + q_off(Q), g_lane_now() * 16u, dst + (uint32_t)(Q * 1024));
    } else {  // every wave PPW of the block's G_BS x NQ pieces
      const uint32_t l16 = g_lane_now() * 16u;
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int i = wave * PPW + k, st = i / NQ, Q = i % NQ;
        g_dma16_stream(tab + (size_t)(Bc * G_BS + st) * G_KSTEP_BYTES + q_off(Q), l16,
                (uint32_t)(OFF_A + (B & 1) * (G_BS * A_STEP) + st * A_STEP + Q * 1024));
      }
    }
  };
  // ---- raw u / w of factor block B (the chunks B * G_BS + 1 .. B * G_BS + G_BS, i.e. what block B's k-steps slice):
  // waves 0..3 one chunk of u each, waves 4..7 one chunk of w (u again when unweighted: never read)
  auto raw_request = [&](int B) {
    int c = B * G_BS + 1 + (wave & 3);
    if (c > nsteps - 1) c = nsteps - 1;
    if constexpr (XBLK) {  // loader wave w: chunk w of u, and of w
      if (!loader) return;
      const int64_t i0 = chunk_sample(c);
      const uint32_t l4 = g_lane_now() * 4u;
      g_dma4(a.u + i0, l4, (uint32_t)(OFF_RAW + (B % 3) * G_RAW + wave * 256));
      if constexpr (WEIGHTED) g_dma4(a.w + i0, l4, (uint32_t)(OFF_RAW + (B % 3) * G_RAW + G_BS * 256 + wave * 256));
    } else {
      const double *src = ((WEIGHTED && wave >= 4) ? a.w : a.u) + chunk_sample(c);
      g_dma4(src, g_lane_now() * 4u, (uint32_t)(OFF_RAW + (B % 3) * G_RAW + (wave >> 2) * (G_BS * 256) + (wave & 3) * 256));
    }
  };
  // ---- factors of block B from raw buffer B % 3 into factor buffer B % 3 (two waves, one sample per lane: waves 0 and 1 where
  // the older waves load, else waves 6 and 7)
  auto stage_factors = [&](int B) {
#ifndef TXM_G_SW
#define TXM_G_SW 6
#endif
    constexpr int SW = TXM_G_SW;  // first staging wave
    if (wave < SW || wave >= SW + 2) return;  // uniform
    const int e = (int)g_lane_now() + (wave - SW) * 64;  // entry: chunk-in-block e >> 5, sample e & 31
    const double *raw = reinterpret_cast<const double *>(lds + OFF_RAW + (B % 3) * G_RAW);
#ifdef TXM_G_TIMING
    G_TICK(1);
    double r0 = raw[e];
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0));
    G_TICK(5);
    const double du = (r0 - pu) * inv_du;
#else
    const double du = (raw[e] - pu) * inv_du;
#endif
    double pw = WEIGHTED ? raw[G_BS * 32 + e] * inv_w : 1.0;
    double *f = reinterpret_cast<double *>(lds + OFF_F + (B % 3) * (2 * G_FU) + ((e >> 4) & 1) * G_FU) + (((e >> 5) * 16 + (e & 15)) * NPT);
    if constexpr (NPT > JN) f[JN] = pw;  // plain w for the y row set
#pragma unroll
    for (int k = 0; k < J0; ++k) pw *= du;
#pragma unroll
    for (int jj = 0; jj < JN; ++jj) {
      if (WEIGHTED || J0 + jj > 0) f[jj] = pw;  // (the constant 1 of an unweighted power 0 is never read)
      pw *= du;
    }
  };

  // ---- store the fixed-point words of the wave's two units of one row set (as resample_i8t_kernel: four 256-byte runs)
  auto store_x2 = [&](uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1, int off) {
#ifdef TXM_G_PLAIN_STORES  // (experiment: stores the compiler can see and count in its s_waitcnt lgkmcnt values -- the four
    // ds_write_addtid_b32 of the asm below are invisible to it, so every wait it derives is up to four operations too strict)
    typedef __attribute__((address_space(3))) uint32_t *lds_u32;
    const uint32_t base = wreg + g_lane_now() * 4u;
    *(lds_u32)(lds + base + off) = lo0;
    *(lds_u32)(lds + base + off + T_PLANE + 128) = hi0;
    *(lds_u32)(lds + base + off + 256) = lo1;
    *(lds_u32)(lds + base + off + 256 + T_PLANE + 128) = hi1;
    return;
#endif
#ifdef TXM_G_PLAIN_STORES_ASM  // (experiment, round 6: the SAME two paired stores the compiler makes of TXM_G_PLAIN_STORES, but inside an
    // asm it cannot count -- the two builds differ only in how strict the s_waitcnt lgkmcnt values behind them are)
    {
      const uint32_t base = wreg + g_lane_now() * 4u + (uint32_t)off;
      asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:1\n\t"
                   "ds_write2st64_b32 %3, %4, %5 offset0:0 offset1:1"
                   :
                   : "v"(base), "v"(lo0), "v"(lo1), "v"(base + (uint32_t)(T_PLANE + 128)), "v"(hi0), "v"(hi1)
                   : "memory");
      return;
    }
#endif
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
  // the words of row set fi for this lane's two samples: f[uu] = the sample factor, d[uu] = dx (or dy).  `ahead()` runs
  // between the last use of f / d and the stores: the LDS reads the NEXT slot consumes are issued there, so that they sit in
  // front of this slot's four stores and two operand reads in the wave's (in-order) LDS queue
  auto produce_row = [&](auto fic, double (&f)[2], const double (&d)[2], auto &&ahead) {
    constexpr int fi = decltype(fic)::value;
    uint32_t lo[2], hi[2];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const uint64_t bits = (uint64_t)__double_as_longlong(fma(f[uu], d[uu], T_MAGIC));
      lo[uu] = (uint32_t)bits ^ 0x80808080u;
      hi[uu] = (uint32_t)(bits >> 32) ^ 0x00008080u;
    }
#ifdef TXM_G_NO_OVERLAY  // ablation build
    if (false) {
#else
    if (fi < JN && ofi == fi) {  // wave-uniform: this wave carries digits of the row set's u-row monomial in byte 7
#endif
      // (the permute as a volatile asm: the compiler must keep the branch -- if-converted, all eight waves executed the
      // overlay of all three row sets, 30 vector instructions per k-step instead of 10 on six waves)
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        const uint64_t ub = (uint64_t)__double_as_longlong(__builtin_ldexp(f[uu], 50) + T_MAGIC);
        uint32_t dig;
        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(dig) : "v"((uint32_t)(ub >> 32)), "v"((uint32_t)ub), "v"(osel));
        hi[uu] = (hi[uu] & 0x00ffffffu) | (dig ^ oxor);
      }
    }
    ahead();
    store_x2(lo[0], hi[0], lo[1], hi[1], fi * T_PB);
  };

  typedef __attribute__((address_space(3))) const double *lds_cd;
  typedef __attribute__((address_space(3))) const v4i *lds_cv4;

  // ================= prologue =================
  // draws per replicate in the window (the top digit's bias is removed with them at the flush)
  if (threadIdx.x < 32 * NQ) {
    const int64_t r = rep0 + threadIdx.x;
    uint32_t s = 0;
    if (r < a.nrep)
      for (int64_t t = t0; t < t1; ++t) s += a.counts[(size_t)r * a.ntiles + t];
    fsum[threadIdx.x] = s;
  }
  // zero the X regions once (the padding between the planes is never written)
  for (int e = threadIdx.x; e < T_WAVES * WREG / 16; e += T_BLOCK) reinterpret_cast<uint4 *>(lds)[e] = make_uint4(0, 0, 0, 0);
  // chunk 0's u / w straight from memory (the direct path of resample_i8t_kernel)
  double d_du[2], d_w[2] = {1.0, 1.0};
  {
    const int64_t i0 = chunk_sample(0);
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      d_du[uu] = (a.u[i0 + 16 * uu + ps] - pu) * inv_du;
      if constexpr (WEIGHTED) d_w[uu] = a.w[i0 + 16 * uu + ps] * inv_w;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing of the compiler's is in flight behind the DMAs below)
  raw_request(0);
  raw_request(1);
  if constexpr (!ADIR) a_request(0);
  if constexpr (XBLK) {  // chunk c lives in slot (c - 1) & 7: chunk 0 in slot 7, block 0's chunks 1..4 in slots 0..3
    x_request(7);
#pragma unroll
    for (int c = 0; c < G_BS; ++c) x_request(c);
  } else {
#pragma unroll
    for (int c = 0; c < G_XR; ++c) x_request(c);
  }
  v4i AD[
#ifdef TXM_G_ADIR
      (XBLK && NS == 2) ? G_BS : 1
#else
      1
#endif
  ][NQ];  // (ADIR) the count operands of the block's four steps
#ifdef TXM_G_ADIR
  if constexpr (XBLK && NS == 2) {
    const uint32_t l16 = g_lane_now() * 16u;
#pragma unroll
    for (int pp = 0; pp < G_BS; ++pp)
#pragma unroll
      for (int q = 0; q < NQ; ++q) g_load16(AD[pp][q], tab + (size_t)pp * G_KSTEP_BYTES + qoff[q], l16);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  stage_factors(0);
  {
    // the X words of chunk 0 (no matrix work yet)
    const uint32_t xa = xring + (uint32_t)(ps * 32 + cl * 8) + (XBLK ? 7u * 1024u : 0u);
    double dx[2], dy[2] = {0.0, 0.0};
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      dx[uu] = (*(lds_cd)(lds + xa + uu * 512) - px) * sc;
      if constexpr (YS) dy[uu] = (*(lds_cd)(lds + xa + (yring - xring) + uu * 512) - py) * scy;
    }
    t_static_for<NS>([&](auto fic) {
      constexpr int fi = decltype(fic)::value;
      double f[2];
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        if constexpr (YS && fi == JN) {
          f[uu] = WEIGHTED ? d_w[uu] : 1.0;
        } else {
          double pw = WEIGHTED ? d_w[uu] : 1.0;
          for (int q = 0; q < J0 + fi; ++q) pw *= d_du[uu];
          f[uu] = pw;
        }
      }
      if constexpr (YS && fi == JN) produce_row(fic, f, dy, [] {});
      else produce_row(fic, f, dx, [] {});
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // factors of block 0 visible; chunk 0's x slot free
  if constexpr (!XBLK) x_request(0);                               // chunk G_XR into slot 0
  G_TICK(0);

  // ================= the blocks =================
  // Step s = B * 4 + p contracts chunk s (count words: ring buffer B & 1, X words: the wave's regions), slices chunk
  // s + 1 (x from ring slot (p + 1) & 3, factors from buffer B % 3 line p) and requests chunk s + 5 into that slot.
  //
  // Software pipeline, no extra registers: the operands of step s + 1 are read INSIDE step s into the registers step s has
  // just finished with -- count operand q behind the last row set's MFMA on quarter q, the B operand of row set fi behind
  // that row set's stores (a wave's LDS operations execute in order) -- so a step starts with its operands in registers
  // instead of a wait for eleven LDS reads.  (First cut: reads at the top of the step; the two waves of a SIMD leave every
  // barrier in phase, so their waits coincided: matrix pipe 55 % busy, profiles/r05a_pmc_i8g.json.)  The block's barrier
  // therefore sits INSIDE its last step, before the last row set: behind it the count words of the next block (landed:
  // every wave waited for its pieces) are read for that block's first step.
  //
  // vmcnt: a wave's DMAs complete in order.  Issue order: [block start: 2 count pieces, 1 raw piece], then per step NX x
  // pieces at its end.  The x of chunk s + 1 was requested at the end of step s - 4; newer than it at the top of step s
  // are the x pieces of steps s - 3 .. s - 1 (3 NX) and the three pieces of the one block start among the last four
  // steps (this step's own when p = 0): 3 NX + 3 at every p.  At the barrier the block-start pieces are older than the
  // 3 NX x pieces of the block's first three steps.
  // L2-sharing hint (as in resample_i8t_kernel; bounded, no result depends on it): the replicate groups of a window run on
  // one XCD and read the same x -- every tile (8 blocks) wave 0 publishes the tiles this group has finished and sleeps while
  // it is more than G_LEAD tiles ahead of the slowest started group, so that the window's x is streamed from HBM once
  uint32_t *pg = a.progress != nullptr ? a.progress + (size_t)win * 16 : nullptr;
#ifdef TXM_G_PRIO  // experiment: the younger wave of every SIMD at a higher issue priority
  if (wave >= 4) __builtin_amdgcn_s_setprio(TXM_G_PRIO);
#endif
  // ---- what a slot reads one slot AHEAD (see below): the factors of the next row set, the raw x of the next chunk
  auto read_factors = [&](auto pnc, auto finc, uint32_t fb, double (&f)[2]) {
    constexpr int pn = decltype(pnc)::value, fin = decltype(finc)::value;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      if constexpr (YS && fin == JN) {
        if constexpr (WEIGHTED) f[uu] = *(lds_cd)(lds + fb + uu * G_FU + (pn * 16 * NPT + (J0 == 0 ? 0 : JN)) * 8);
        else f[uu] = 1.0;
      } else if constexpr (!WEIGHTED && J0 == 0 && fin == 0) {
        f[uu] = 1.0;
      } else {
        f[uu] = *(lds_cd)(lds + fb + uu * G_FU + (pn * 16 * NPT + fin) * 8);
      }
    }
  };
  v4i A[NQ];
  // B operands: passes of >= 2 row sets hold TWO (this slot's and the next one's, read a slot ahead), not one per row set
  constexpr bool BT2 = NS >= 2;
  v2i Bt[BT2 ? 2 : 1][2];
  double f[2], xr[2], yr[2] = {0.0, 0.0};
  // loop-carried bases (opaque below: the reads take 16-bit immediate offsets): the factor buffer of the block, and the x ring
  // (half of the block when a block's chunks are requested together)
  uint32_t f_va = (uint32_t)(OFF_F + ps * NPT * 8);
  uint32_t x_va = xring + (uint32_t)(ps * 32 + cl * 8);
  {
    const uint32_t a_va0 = (uint32_t)OFF_A + (uint32_t)lane * 16u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if constexpr (!ADIR) A[q] = *(lds_cv4)(lds + a_va0 + q * 1024);
    }
    Bt[0][0] = T_TRREAD((lds_v2i)(lds + rd_off));
    Bt[0][1] = T_TRREAD((lds_v2i)(lds + rd_off + 128));
    read_factors(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, f_va, f);
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) xr[uu] = *(lds_cd)(lds + x_va + (XBLK ? 0 : 1024) + uu * 512);  // chunk 1
  }
#pragma unroll 1
  for (int B = 0; B < nblk; ++B) {
    if (pg != nullptr && wave == 0 && (B & 7) == 0) {  // uniform
      const uint32_t done = (uint32_t)(B >> 3) + 1u;
      if (lane == 0) __hip_atomic_store(&pg[grp & 15], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
      for (int spin = 0; spin < I8_THROTTLE_SPINS; ++spin) {
        if (done <= g_min_progress(pg) + G_LEAD) break;
        __builtin_amdgcn_s_sleep(32);
      }
    }
    G_TICK(1);
    // (the staging reads first: a DS read issued right behind the block's DMA pieces waited ~1000 cycles for them on the two
    // staging waves -- phase clocks, profiles/r05_experiments.md)
    stage_factors(B + 1);
    G_TICK(7);
    raw_request(B + 2);
    // the loader waves' requests for the next block, lean: what does not change inside a block -- the count pieces' source and
    // destination, the lanes' offsets -- is formed here once (the general a_request / x_request did it per piece: 57
    // instructions an item, a third of a loader wave's instruction stream -- this kernel is paced by what a wave has to issue)
    const unsigned char *blk_asrc = nullptr;
    uint32_t blk_adst = 0, blk_l16 = 0, blk_xrow = 0;
    if constexpr (XBLK) {
      if (loader) {
        const int Bn = B + 1 < nblk ? B + 1 : nblk - 1;
        blk_asrc = tab + (size_t)(Bn * G_BS + wave) * G_KSTEP_BYTES;
        blk_adst = (uint32_t)(OFF_A + ((B + 1) & 1) * (G_BS * A_STEP) + wave * A_STEP);
        const uint32_t ln = g_lane_now();
        blk_l16 = ln * 16u;
        blk_xrow = (ln >> 1) * (uint32_t)(a.ldx_s * 8) + (ln & 1) * 16u;
      }
    }
    auto a_piece = [&](int Q) {
#ifndef TXM_G_NO_ADMA
      if constexpr (!ADIR)
        if (loader) g_dma16_stream(blk_asrc + qoff[Q], blk_l16, blk_adst + (uint32_t)(Q * 1024));
#endif
    };
    // (ADIR) the next block's count words: this lane's 16 bytes of every (step, quarter) piece
    const unsigned char *nb_src = nullptr;
    uint32_t nb_l16 = 0;
    if constexpr (ADIR) {
      const int Bn = B + 1 < nblk ? B + 1 : nblk - 1;
      nb_src = tab + (size_t)(Bn * G_BS) * G_KSTEP_BYTES;
      nb_l16 = g_lane_now() * 16u;
    }
    auto x_chunk = [&](int i) {  // chunk cq -> slot i of the ring half the next block reads: the wave's own columns and its partner's
#ifndef TXM_G_NO_XDMA
      if (loader) {
        const uint32_t dst = xring + (uint32_t)((4 * ((B + 1) & 1) + i) * 1024);
        g_dma16(xq + xcol1, blk_xrow, dst);
        g_dma16(xq + xcol2, blk_xrow, dst + (uint32_t)(4 * XRN * 1024));
      }
#endif
      ++cq;
      if (cq < nsteps) {
        if ((cq & 31) == 0) xq = reinterpret_cast<const char *>(a.x + chunk_sample(cq) * a.ldx_s);  // a new tile
        else xq += xstep;
      }
    };
#ifdef TXM_G_BURST  // (experiment build: the next block's count words and x requested here, 14 pieces per loader wave at once)
    constexpr bool SPREAD = false;
#else
    constexpr bool SPREAD = XBLK && NS >= 2;  // (one row set: a step is short, two bursts of 6 cost more than one of 12: 77.7 -> 75.6 ms at order 0)
#endif
    if constexpr (!SPREAD) {
      if constexpr (XBLK) {  // the next block's count words and x (chunks 4 (B + 1) + 1 .. + 4 into the ring half this block does not read)
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q) a_piece(Q);
#pragma unroll
        for (int i = 0; i < G_BS; ++i) x_chunk(i);
      } else {
        a_request(B + 1);
      }
    }
    // SPREAD: the eight requests of the next block (four count pieces, four x chunks = 12 pieces per loader wave) go out one or
    // two per slot over the slots of steps 0 and 1 -- a burst of 56 KiB per CU filled the vector memory pipeline and held
    // its issuers for ~2000 cycles a block (the DMA cost 20 of the pass's 75 ms; additive ablation, profiles/r05_experiments.md)
    constexpr int ISSUE_SLOTS = 2 * NS;
    auto issue_items = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (SPREAD && k < ISSUE_SLOTS) {
        if constexpr (ADIR) {  // four x chunks over the four slots of steps 0 and 1
          static_assert(!ADIR || ISSUE_SLOTS == G_BS, "one x chunk a slot");
          x_chunk(k);
        } else {
#pragma unroll
          for (int j = 0; j < NQ + 4; ++j)
            if (j * ISSUE_SLOTS / (NQ + 4) == k) {  // interleaved: count piece, x chunk, count piece, ... then the remaining count pieces
              if (j < 8 && (j & 1) == 0) a_piece(j >> 1);
              else if (j < 8) x_chunk(j >> 1);
              else a_piece(j - 4);
            }
        }
      }
    };
    G_TICK(2);
    // count words: this block's steps 1..3 from buffer B & 1, the next block's step 0 from the other buffer
    uint32_t a_va = (uint32_t)(OFF_A + (B & 1) * (G_BS * A_STEP)) + (uint32_t)lane * 16u;
    asm volatile("" : "+v"(f_va), "+v"(x_va), "+v"(a_va));
    // SLOT = one row set of one step: its four MFMAs, then the words of chunk s + 1 for that row set.  A wave issues in order,
    // and its LDS operations complete in order: a read consumed right where it was issued -- the first cuts read a slot's factors
    // behind the stores and operand reads of the slot before and waited for them three MFMAs later, and read x at the top of the
    // step -- DRAINS the wave's whole LDS queue under the load of eight waves, four times a step, with the matrix pipe idle:
    // the kernel's time was its LDS time PLUS its MFMA time (ablations, profiles/r05_experiments.md), although the two do
    // overlap on this chip when nothing waits (tools/mfma_lds_overlap_probe.hip: 12 MFMAs + 24 LDS operations per wave take
    // 0.39 us interleaved against 0.37 + 0.34 alone).  So every read is issued ONE SLOT AHEAD of its use and IN FRONT of that
    // slot's stores: the next slot's factors between the last use of this slot's and its stores (same registers), the raw x
    // of the next step behind the last slot's operand reads (into the registers of dx, dead by then), y in the slot before its
    // row set; the waits the compiler derives are lgkmcnt(6) and up -- nothing drains but the block's barrier.
    // (An experiment that let the younger wave of a SIMD take the barrier behind the last row set's MFMAs -- the waves of a
    // SIMD out of phase -- was slower: 114.4 against 107.9 ms at order 2.)
#ifdef TXM_G_NO_PIN
#define G_PIN() do {} while (0)
#else
#define G_PIN() __builtin_amdgcn_sched_barrier(0)
#endif
    t_static_for<G_BS>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      double dx[2], dy[2] = {0.0, 0.0};
      if constexpr (ADIR) {
        // AD[p][*] were requested behind step p's last MFMAs ONE BLOCK AGO.  Vector-memory operations complete in order; issued
        // behind AD[p][3] since then: the twelve count loads of the three other steps, and on a loader wave a block's DMA --
        // R raw pieces at the block's top and the x pieces of steps 0 and 1 (two per slot): of those, four x pieces lie behind
        // AD[0][3] / AD[1][3] and eight behind AD[2][3] / AD[3][3]
        constexpr int R = WEIGHTED ? 2 : 1;
        constexpr int NL = 12 + (p < 2 ? 4 : 8) + R;
        // (no register operands on the waits: tied through the two branches they made the compiler copy the operands -- in one
        // branch in FRONT of the wait; the scheduling barriers keep the step's MFMAs behind them instead)
        G_PIN();
        if (loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        G_PIN();
      }
      t_static_for<NS>([&](auto fic) {
        constexpr int fi = decltype(fic)::value;
        constexpr bool last = fi == NS - 1;
        if constexpr (p == G_BS - 1 && last) {
          G_TICK(3);
          if constexpr (ADIR) {
            // the block's DMA (raw pieces at its top, the x pieces in the slots of steps 0 and 1) has landed when at most the
            // seven count loads issued behind the last x piece are in flight: A[1] quarters 1..3 and A[2]; waves 4..7 issue no DMA
            if (loader) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XBLK ? 0 : 3 * NX) : "memory");
          }
#ifndef TXM_G_NO_BARRIER  // (ablation build: no barrier)
          asm volatile("s_barrier" ::: "memory");  // the block's barrier
#endif
          G_TICK(6);
        }
        auto ahead = [&]() {
          if constexpr (!last) {
            read_factors(pc, std::integral_constant<int, fi + 1>{}, f_va, f);
          } else if constexpr (p < G_BS - 1) {
            read_factors(std::integral_constant<int, p + 1>{}, std::integral_constant<int, 0>{}, f_va, f);
          } else {  // behind the barrier: the next block's buffer
            f_va = (uint32_t)(OFF_F + ((B + 1) % 3) * (2 * G_FU) + ps * NPT * 8);
            asm volatile("" : "+v"(f_va));
            read_factors(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, f_va, f);
          }
          if constexpr (YS && fi == JN - 1) {  // y of the chunk being sliced (same ring slot as its x)
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) yr[uu] = *(lds_cd)(lds + x_va + (OFF_Y - OFF_X) + ((p + 1) & 3) * 1024 + uu * 512);
          }
          if constexpr (last) {  // raw x of chunk s + 2, which the next step slices (dx is dead: into its registers)
            if constexpr (XBLK) {
              if constexpr (p == G_BS - 1) {  // the other half of the ring (landed: every loader waited before the barrier)
                x_va = xring + (uint32_t)(ps * 32 + cl * 8) + (uint32_t)(((B + 1) & 1) * 4096);
                asm volatile("" : "+v"(x_va));
              }
#pragma unroll
              for (int uu = 0; uu < 2; ++uu) xr[uu] = *(lds_cd)(lds + x_va + (p == G_BS - 1 ? 0 : p + 1) * 1024 + uu * 512);
            } else {
              // vmcnt: chunk s + 2 was requested at the end of step s - 3; newer than it are the x (+ y) pieces of steps s - 2
              // and s - 1 and the three pieces of a block start at the top of step s - 2, s - 1 or s (none when p = 3)
              g_wait_vm<2 * NX + (p == G_BS - 1 ? 0 : PPW + 1)>();
#pragma unroll
              for (int uu = 0; uu < 2; ++uu) xr[uu] = *(lds_cd)(lds + x_va + ((p + 2) & 3) * 1024 + uu * 512);
            }
          }
        };
        constexpr int bcur = BT2 ? ((p * NS + fi) & 1) : 0;  // (a block has an even number of slots: the parity restarts with it)
        const v4i Bv = {Bt[bcur][0][0], Bt[bcur][0][1], Bt[bcur][1][0], Bt[bcur][1][1]};
        // The slot, INTERLEAVED: a piece of the slicing behind every MFMA (pinned: the scheduler otherwise issues the four
        // MFMAs back to back, and a wave that waits for the matrix pipe between two of its own MFMAs issues nothing else).
        auto mfma_q = [&](auto qc) {
          constexpr int q = decltype(qc)::value;
          if constexpr (ADIR) t_mfma<true>(acc[fi][q], AD[p][q], Bv);
          else t_mfma<true>(acc[fi][q], A[q], Bv);
          // quarter q's count operand of the NEXT step into the registers just used for the last time
#ifndef TXM_G_NO_AREAD  // (ablation build)
          if constexpr (last && ADIR) {  // step p of the NEXT block into the registers this block's step p has just finished with
            g_load16(AD[p][q], nb_src + (size_t)p * G_KSTEP_BYTES + qoff[q], nb_l16);
          } else if constexpr (last) {
            if constexpr (p == G_BS - 1) {  // the other ring buffer: the next block's step 0 (address formed here, not held)
              const uint32_t a_vn = (uint32_t)(OFF_A + ((B + 1) & 1) * (G_BS * A_STEP)) + g_lane_now() * 16u;
              A[q] = *(lds_cv4)(lds + a_vn + q * 1024);
            } else {
              A[q] = *(lds_cv4)(lds + a_va + (p + 1) * A_STEP + q * 1024);
            }
          }
#endif
        };
        G_PIN();
        t_static_for<(0 + 1) * NQ / 4 - 0 * NQ / 4>([&](auto jc) { mfma_q(std::integral_constant<int, 0 * NQ / 4 + decltype(jc)::value>{}); });
        G_PIN();
        issue_items(std::integral_constant<int, p * NS + fi>{});
        if constexpr (BT2) {
          // the NEXT slot's B operand, a slot ahead: row set fi + 1's words of this chunk were stored a step ago, row set 0's words
          // of the next chunk in this step's slot 0 (earlier in this wave's LDS queue either way)
          constexpr int fn = (fi + 1) % NS;
          Bt[bcur ^ 1][0] = T_TRREAD((lds_v2i)(lds + rd_off + fn * T_PB));
          Bt[bcur ^ 1][1] = T_TRREAD((lds_v2i)(lds + rd_off + fn * T_PB + 128));
        }
        G_PIN();
        // (a) the fixed-point words of chunk s + 1, row set fi
        if constexpr (fi == 0) {
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) dx[uu] = (xr[uu] - px) * sc;
        }
        if constexpr (YS && fi == JN) {
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) dy[uu] = (yr[uu] - py) * scy;
        }
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
#ifdef TXM_G_NO_SLICE  // (ablation build: constants are stored)
          lo[uu] = 0x01020304u + uu;
          hi[uu] = 0x04030201u + uu;
          asm volatile("" : "+v"(lo[uu]), "+v"(hi[uu]));
#else
          const double dd = (YS && fi == JN) ? dy[uu] : dx[uu];
          const uint64_t bits = (uint64_t)__double_as_longlong(fma(f[uu], dd, T_MAGIC));
          lo[uu] = (uint32_t)bits;
          hi[uu] = (uint32_t)(bits >> 32);
#endif
        }
        G_PIN();
        t_static_for<(1 + 1) * NQ / 4 - 1 * NQ / 4>([&](auto jc) { mfma_q(std::integral_constant<int, 1 * NQ / 4 + decltype(jc)::value>{}); });
        G_PIN();
#ifndef TXM_G_NO_SLICE
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
          lo[uu] ^= 0x80808080u;
          hi[uu] ^= 0x00008080u;
        }
#endif
        G_PIN();
        // (b) the u-row overlay (wave-uniform branch, kept by the volatile permute), then the reads of the NEXT slot
#if !defined(TXM_G_NO_OVERLAY) && !defined(TXM_G_NO_SLICE)
        if (fi < JN && ofi == fi) {
#ifdef TXM_G_OVERLAY_RECOMPUTE  // (selector and bias from the lane id, here: two registers fewer held across the k-steps, nine vector
          // instructions more per overlay -- what the six-quarter A/B build needs to fit)
          const uint32_t od_l = 4u * (uint32_t)(wave & 1) + (g_lane_now() & 3u);
          const uint32_t osel_l = od_l < 7u ? ((od_l << 24) | 0x000c0c0cu) : 0x0c0c0c0cu;
          const uint32_t oxor_l = od_l < 6u ? 0x80000000u : 0u;
#else
          const uint32_t osel_l = osel, oxor_l = oxor;
#endif
#pragma unroll
          for (int uu = 0; uu < 2; ++uu) {
            const uint64_t ub = (uint64_t)__double_as_longlong(__builtin_ldexp(f[uu], 50) + T_MAGIC);
            uint32_t dig;
            asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(dig) : "v"((uint32_t)(ub >> 32)), "v"((uint32_t)ub), "v"(osel_l));
            hi[uu] = (hi[uu] & 0x00ffffffu) | (dig ^ oxor_l);
          }
        }
#endif
        G_PIN();
#ifndef TXM_G_NO_AHEAD  // (ablation build: no factor / x / y reads)
        ahead();
#endif
        G_PIN();
        t_static_for<(2 + 1) * NQ / 4 - 2 * NQ / 4>([&](auto jc) { mfma_q(std::integral_constant<int, 2 * NQ / 4 + decltype(jc)::value>{}); });
        G_PIN();
        // (c) the stores
#ifndef TXM_G_NO_PRODUCE  // (ablation build: no stores)
        store_x2(lo[0], hi[0], lo[1], hi[1], fi * T_PB);
#else
        asm volatile("" ::"v"(lo[0]), "v"(hi[0]), "v"(lo[1]), "v"(hi[1]));
#endif
        G_PIN();
        t_static_for<(3 + 1) * NQ / 4 - 3 * NQ / 4>([&](auto jc) { mfma_q(std::integral_constant<int, 3 * NQ / 4 + decltype(jc)::value>{}); });
        G_PIN();
        // (d) behind the stores, behind the MFMAs that read the old ones: the next step's B operand of the row set
#ifndef TXM_G_NO_TRREAD  // (ablation build)
        if constexpr (!BT2) {  // one row set: the next step's operand, behind this step's stores
          Bt[0][0] = T_TRREAD((lds_v2i)(lds + rd_off));
          Bt[0][1] = T_TRREAD((lds_v2i)(lds + rd_off + 128));
        }
#endif
        G_PIN();
      });
      if constexpr (!XBLK) x_request((p + 1) & 3);  // chunk s + 5 (slot of chunk s + 1: x read a step ago, y in this one)
    });
  }

  // ================= flush: int32 accumulators of the window -> its slot of the partial sums =================
  // D layout of v_mfma_i32_32x32x32_i8: column = lane & 31, row = 8 * (reg / 4) + 4 * (lane >> 5) + reg % 4
  // x slots: ONE double per (replicate, power, column) -- the seven digit sums of an element added up here,
  // ((((((d0 + d1) + d2) + d3) + d4) + d5) + d6) of (double)(int32 sum) x scale: the expression and the order
  // resample_finalize_i8_kernel applies to per-digit slots, and what the narrow kernels store (same bits; an eighth of the x slots'
  // bytes written here and read by the finalize: on a short wide series -- N = 1e6, 32 observables, 1000 replicates: 245 windows --
  // the slots were 2.6 GB around a 2.9 ms call, the finalize 0.44 ms of it).  A tile goes through a wave-private scratch in the (now
  // idle) count / x rings, rows of 65 words (see resample_i8t_kernel): digit d of (replicate row m, column q4) sits in register
  // 4 (m >> 3) + (m & 3) of lane 16 (d >> 2) + 4 q4 + (d & 3) + 32 ((m >> 2) & 1).  The u-row digits that ride in the dead eighth
  // byte of OTHER waves' columns stay per-digit slots (part_u), and so does the second matrix's row set (part_y): the finalize is told
  // (mode 2: x summed, u per digit).
  constexpr int XS = 65, XT = 16 * XS;
  static_assert(OFF_A + T_WAVES * XT * 4 <= OFF_RAW, "flush scratch inside the count and x rings");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // no DMA piece in flight, no wave still reading the rings
  if (pg != nullptr && threadIdx.x == 0) __hip_atomic_store(&pg[grp & 15], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto odig_of = [](int w, int c) { return 4 * (w & 1) + c; };  // the u-row digit column c of wave w carries in its dead byte
  uint32_t *xw = reinterpret_cast<uint32_t *>(lds + OFF_A) + wave * XT;
  auto flush_tile = [&](v16i &T, int q, int rs) {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));  // opaque zero: the addresses are formed where they are used, not hoisted and spilled
    const int64_t opq = (int64_t)z;
    const bool is_y = YS && rs == JN;
    if (is_y) {  // the second matrix's row set: per-digit slots [window][replicate][digit slot][column], as before
      const int64_t rrow = rep0 + 32 * q + 4 * half;
      const int c = 4 * wave + tcl;
      const bool valid = tdg < I8_NSL && c < a.C;
      const int dgt = tdg < I8_NSL ? tdg : 0;
      double dsc = wty[I8_WT_DSP + 0] * wty[I8_WT_DSC + (c < a.C ? c : 0)];
      double *base = a.part_y + (((size_t)win * a.nrep_pad + rrow) * 8 + dgt) * a.cpad + c + opq;
      const size_t stride = (size_t)8 * a.cpad;
      dsc *= (double)((int64_t)1 << (8 * dgt));
      const int bias = dgt == I8_NSL - 1 ? T_D6_BIAS : 0;
      if (valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r >> 2) * 8 + (r & 3);
          if (rrow + m < a.nrep) {
            const int v = T[r] - bias * (int)fsum[32 * q + m + 4 * half];
            base[(size_t)m * stride] = (double)v * dsc;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) xw[r * XS + lane] = (uint32_t)T[r];
    // (the wave's own LDS operations execute in order: the reads below see the stores above)
    const int j = J0 + rs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = lane + 64 * i, m = o >> 2, q4 = o & 3;
      const int rr = ((m >> 3) << 2) | (m & 3), hh = (m >> 2) & 1;
      const int64_t rrow = rep0 + 32 * q + m;
      if (rrow < a.nrep) {
        const uint32_t *src = xw + rr * XS + 32 * hh + 4 * q4;
        const int fs = (int)fsum[32 * q + m];
        const int c = 4 * wave + q4;
        if (c < a.C) {
          int v[I8_NSL];
#pragma unroll
          for (int d = 0; d < I8_NSL; ++d) v[d] = (int)src[16 * (d >> 2) + (d & 3)];
          v[I8_NSL - 1] -= T_D6_BIAS * fs;
          const double dsc0 = wt[I8_WT_DSP + j] * wt[I8_WT_DSC + c];
          double sum = 0.0;
#pragma unroll
          for (int d = 0; d < I8_NSL; ++d) {
            double pd = (double)v[d] * (dsc0 * (double)((int64_t)1 << (8 * d)));
            asm volatile("" : "+v"(pd));  // (a product rounded on its own, as the stored slot was: no fused multiply-add with the sum)
            sum = d == 0 ? pd : sum + pd;
          }
          a.part_x[(((size_t)win * a.nrep_pad + rrow) * K + j) * a.cpad + c + opq] = sum;
        }
        const int dgt = odig_of(wave, q4);
        if (ofi == rs && dgt < 7) {  // the u-row digit this column's dead byte carried: [window][replicate][power][digit slot]
          const int v7 = (int)src[16 + 3] - (dgt == I8_NSL - 1 ? T_D6_BIAS * fs : 0);
          const double dsc = (wt[I8_WT_DSP + j] * 0x1p-50) * (double)((int64_t)1 << (8 * dgt));
          a.part_u[((size_t)win * a.nrep_pad + rrow) * K * 8 + (size_t)j * 8 + dgt + opq] = (double)v7 * dsc;
        }
      }
    }
  };
#pragma unroll
  for (int fi = 0; fi < NS; ++fi)
#pragma unroll
    for (int q = 0; q < NQ; ++q) flush_tile(acc[fi][q], q, fi);
#ifdef TXM_G_CLOCKS
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 1064)) {
    double *o = a.wtab + a.nwin * I8_WT_STRIDE + ((blockIdx.x ? 1 : 0) * T_WAVES + wave) * 8;
    o[0] = (double)(clock64() - gc0);
    o[1] = (double)(wall_clock64() - gr0);
  }
#endif
#ifdef TXM_G_TIMING
  G_TICK(0);
  if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 1064))
    for (int k = 0; k < 8; ++k) a.wtab[a.nwin * I8_WT_STRIDE + ((blockIdx.x ? 1 : 0) * T_WAVES + wave) * 8 + k] = (double)tm[k];
#endif
}

// ---------------------------------------------------------------------------
template <int J0, int JN, bool WEIGHTED, bool YS>
static int launch_pass_g(const I8Args &a, int K, const unsigned char *table, int64_t rep_begin, int n_grp, hipStream_t st) {
  constexpr int NS = JN + (YS ? 1 : 0);
  constexpr int NQ = G_NQ<J0, JN, WEIGHTED, YS>;
  const size_t lds = (size_t)T_WAVES * NS * T_PB + 2 * G_BS * NQ * 1024 + (size_t)(YS ? 2 * G_XR : G_XRB) * T_WAVES * 1024 + 3 * G_RAW +
                     3 * 2 * G_FU + 32 * NQ * sizeof(uint32_t);
  const dim3 grid((unsigned)(cdiv(a.nwin, 8) * 8 * cdiv(4 * (int64_t)n_grp, NQ)));
  if (a.progress != nullptr) TXM_HIP(hipMemsetAsync(a.progress, 0, (size_t)cdiv(a.nwin, 8) * 8 * 16 * sizeof(uint32_t), st));
  TXM_SET_MAX_LDS((&resample_i8g_kernel<J0, JN, WEIGHTED, YS>), lds);
  hipLaunchKernelGGL((resample_i8g_kernel<J0, JN, WEIGHTED, YS>), grid, dim3(T_BLOCK), lds, st, a, K, table, rep_begin, n_grp);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

// what the kernel asks of a call beyond the shape (LDS-DMA moves 16 bytes per lane): x (and y) 16-byte aligned with an even
// row pitch, and whole column quads readable inside a row
bool i8g_applicable(const double *x, int64_t ldx_s, int64_t C, const double *y, int64_t ldy_s) {
  if (C <= 16) return false;  // narrow states: the quad-sharing variant of resample_i8t_kernel
  const int64_t cq = (C + 3) / 4 * 4;
  if (((uintptr_t)x & 15) != 0 || (ldx_s & 1) != 0 || cq > ldx_s) return false;
  if (y != nullptr && (((uintptr_t)y & 15) != 0 || (ldy_s & 1) != 0 || cq > ldy_s)) return false;
  return true;
}

// the passes of one 32-column group over the count table of replicate groups [rep_begin, rep_begin + 128 n_grp):
// K power row sets (+ the second matrix's) over the fewest passes of at most three.  Measured pass times at the north-star size:
// 70.5 / 54.5 / 30.8 ms for 3 / 2 / 1 row sets (the single-row-set pass takes 256 replicates per workgroup), so without a second
// matrix a remainder of one goes into a pass of its own -- 3 + 1 (101 ms) beats 2 + 2 (109), 3 + 3 + 1 beats 3 + 2 + 2 -- and
// every other count is dealt out evenly; with a second matrix (it rides on the last pass, which needs a power beside it) evenly.
int launch_resample_i8g(const I8Args &a, int K, bool weighted, const unsigned char *table, int64_t rep_begin, int n_grp,
                        hipStream_t st) {
  const bool ys = a.y != nullptr;
  const int rows = K + (ys ? 1 : 0), np = (rows + 2) / 3;
#ifdef TXM_G_EVEN_SPLIT  // (A/B build: sizes within one of each other, always)
  const bool greedy = false;
#else
  const bool greedy = !ys && rows > 3 && rows % 3 == 1;
#endif
  int j0 = 0;
  for (int i = 0; i < np; ++i) {
    const int n = greedy ? (i < np - 1 ? 3 : 1) : rows / np + (i < rows % np ? 1 : 0);
    const bool last = i == np - 1;
    const int jn = n - ((last && ys) ? 1 : 0);
    const bool y_here = last && ys;
    int rc = TXM_ERR_INVALID;
#define G_CASE(J0_, JN_)                                                                                         \
  if (j0 == J0_ && jn == JN_) {                                                                                  \
    if (y_here) {                                                                                                \
      if constexpr (JN_ <= 2) rc = weighted ? launch_pass_g<J0_, JN_, true, true>(a, K, table, rep_begin, n_grp, st) \
                                            : launch_pass_g<J0_, JN_, false, true>(a, K, table, rep_begin, n_grp, st); \
    } else rc = weighted ? launch_pass_g<J0_, JN_, true, false>(a, K, table, rep_begin, n_grp, st)                \
                       : launch_pass_g<J0_, JN_, false, false>(a, K, table, rep_begin, n_grp, st);              \
  }
#ifdef TXM_G_ONLY03  // (ablation builds: one instance, seconds to compile)
    if (j0 == 0 && jn == 3 && !y_here && !weighted) rc = launch_pass_g<0, 3, false, false>(a, K, table, rep_begin, n_grp, st);
#elif defined(TXM_G_ONLY32)
    if (j0 == 3 && jn == 2 && !y_here && !weighted) rc = launch_pass_g<3, 2, false, false>(a, K, table, rep_begin, n_grp, st);
#elif defined(TXM_G_ONLY02)
    if (j0 == 0 && jn == 2 && !y_here && !weighted) rc = launch_pass_g<0, 2, false, false>(a, K, table, rep_begin, n_grp, st);
#elif defined(TXM_G_ONLY01)
    if (j0 == 0 && jn == 1 && !y_here && !weighted) rc = launch_pass_g<0, 1, false, false>(a, K, table, rep_begin, n_grp, st);
#else
    G_CASE(0, 1) G_CASE(0, 2) G_CASE(0, 3) G_CASE(2, 1) G_CASE(2, 2) G_CASE(3, 1) G_CASE(3, 2) G_CASE(3, 3)
    G_CASE(5, 1) G_CASE(5, 2) G_CASE(6, 1) G_CASE(6, 2)
#endif
#undef G_CASE
    if (rc != TXM_OK) {
      if (rc == TXM_ERR_INVALID) set_error("resample_i8g: no pass for powers %d..%d", j0, j0 + jn - 1);
      return rc;
    }
    j0 += jn;
  }
  return TXM_OK;
}

}  // namespace txm
