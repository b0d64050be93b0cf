// txm_resample_i8.h -- internal interface of the int8-sliced bootstrap kernel
// (txm_resample_i8.hip), called from txm_resample_vals (txm_resample.hip).
#pragma once
#include "txm_common.h"

namespace txm {

constexpr int I8_REPS = 64;        // replicates per workgroup (2 MFMA row blocks of 32)
constexpr int I8_NSL = 7;          // signed 8-bit slices of the 51-bit fixed-point operand
constexpr int I8_WIN_TILES = 256;  // sampler tiles per scaling window (262144 samples); short chunks use 64, 16 or 4
constexpr int I8_CPAD = 32;        // columns of one MFMA column block
constexpr int I8_WT_STRIDE = 80;   // doubles per window-table entry
constexpr int I8_SUB_TILES = 8;    // tiles per entry of the FP64 fallback list (a flagged window = win_tiles / 8 runs)

// Precision guard.  One rint per monomial at 2^-50 of the window's scale M = max|w| max|du|^j max|dx_c|: a window of n
// samples contributes an error of ~0.41 * 2^-50 * M * sqrt(n) to a replicate's sum (sum f^2 ~ 2n, uniform rounding).
// Held against the window's TYPICAL content n * typ, typ = a robust mean of |w du^j dx_c| (the smallest of 64 group
// means: a handful of outliers cannot inflate it), that is within 1e-13 as long as  M <= I8_GUARD * sqrt(n) * typ.
// M / typ grows with j, so only the top power is tested.  Windows that fail are contracted by the FP64 kernel.
constexpr double I8_GUARD = 275.0;  // 1e-13 / (0.41 * 2^-50)

// Window table entry w (doubles):  [0] 1/max|u-pu|   [1] 1/max|w| (1 when unweighted)
//   [2 + j]      descale of power j      = max|w| * max|u-pu|^j          (j < 9)
//   [12 + c]     scale of column c       = 2^50 / max|x_c - px_c|        (c < 32)
//   [44 + c]     descale of column c     = max|x_c - px_c| * 2^-50
constexpr int I8_WT_INVDU = 0, I8_WT_INVW = 1, I8_WT_DSP = 2, I8_WT_SC = 12, I8_WT_DSC = 44;

// S state points of one shape in one set of launches (txm_resample_vals_batched on the int8 path): every kernel of the
// path takes the state from a grid axis and its operands from this table (device memory, filled by the host per call).
// Everything a single-state call derives from its arguments sits here per state, so a batched call and S single calls
// run the same arithmetic on the same partial-sum slots: bit for bit the same moments.
struct I8State {
  const double *x, *u, *w;
  const double *pivot;     // [1 + C]
  double *stats;           // pre-pass scratch of this state
  double *wtab;            // [nwin][I8_WT_STRIDE]
  uint32_t *wflag;         // [nwin]
  uint32_t *list, *n_list; // FP64 fallback runs of this state
  double *part_x, *part_u; // int8 partial sums (per window)
  double *fb_x, *fb_u;     // FP64 fallback partial sums (per chunk)
  const uint32_t *counts;  // [nrep][ntiles] rows of this state
  double *out;             // [nrep][C][2][K]
  uint32_t rep_base;       // stream replicate of the state's replicate 0
  uint32_t pad_;
};

struct I8Args;
struct I8Args {
  const I8State *states = nullptr;  // batched launch: state blockIdx.y (narrow-state kernels only); nullptr: one state
  const I8Args *batch_args = nullptr;  // ... and the bootstrap kernel's own arguments per state (device array [S], built per call)
  int64_t S = 1;
  const double *x;
  int64_t ldx_s;
  const double *u;
  const double *w;
  int64_t N, C, nrep;      // C <= 32: the columns col0 .. col0 + C - 1 of x (one column group per launch)
  int64_t col0;
  int64_t C_call;          // all columns of the call (decides the kernel family: i8t_applicable)
  const uint32_t *counts;  // [nrep][ntiles]
  uint32_t k0, k1;
  uint32_t rep_base;       // replicate r of the call draws stream replicate rep_base + r (txm_sampler_spec.rep0)
  int64_t ntiles;
  uint32_t last_tile_size;
  const double *pivot;     // [1 + all columns]: {pivot_u, pivot_x[...]}, indexed with col0
  double *wtab;            // [nwin][I8_WT_STRIDE]
  double *stats;           // [ceil(ntiles / min(16, win_tiles))][100] per-sub-block statistics of the pre-pass
  int64_t nwin;
  // partial sums: ONE SLOT PER SCALING WINDOW, written once (no read-modify-write, no zeroing; the finalize kernel
  // skips the windows the guard flagged and adds the rest in window order, so a replicate's sums do not depend on the
  // launch geometry)
  double *part_x;          // [nwin][nrep_pad][K][8 digit slots][cpad columns], or digit-summed [nwin][nrep_pad][K][cpad] (part_summed)
  int cpad;                // columns of a row of part_x: 32, or 4 / 8 where the narrow-state kernel runs (i8_cpad)
  double *part_u;          // [nwin][nrep_pad][K][8 digit slots]
  int part_summed;         // wide fused kernel: 1 = store digit-summed slots [nwin][nrep_pad][K][cpad] / [nwin][nrep_pad][K] (the narrow
                           // kernels always do, the wide table-fed kernel for x only: see resample_finalize_i8_kernel's `summed`)
  // optional second sample matrix (txm_resample_opts.y): order-0 sums sum_i f w (y_c - py_c) of its 32 columns, carried
  // as one more row set of the LAST pass of the transposing-read kernel (nullptr: none)
  const double *y;
  int64_t ldy_s;
  const double *ypivot;    // [1 + all columns]: {pivot_u, pivot_y[...]}
  double *ywtab;           // [nwin][I8_WT_STRIDE]: the window table of y (column scales of y)
  uint32_t *yflag;         // [nwin] guard flags of y, OR-ed into wflag by the pre-pass
  double *part_y;          // [nwin][nrep_pad][8 digit slots][32 columns]
  int n_chunks, n_rbg;
  int64_t tiles_per_chunk; // multiple of win_tiles
  int64_t win_tiles;       // sampler tiles per scaling window: 256, 64, 16 or 4
  int64_t nrep_pad;
  // precision guard: flag[w] != 0 -> window w is left to the FP64 kernel (run list built by i8_list_kernel)
  uint32_t *wflag;         // [nwin]
  uint32_t *list;          // [nwin * (win_tiles / sub_tiles)] first tile of every run; n_list[0] = runs, n_list[1] = flagged windows
  uint32_t *n_list;
  int sub_tiles;           // tiles per run: min(I8_SUB_TILES, win_tiles)
  // L2-sharing hint (see ResampleArgs::progress in txm_resample.hip): the replicate groups of a chunk publish the
  // tiles they have finished and stay within I8_LEAD tiles of each other, so that the chunk is streamed from HBM once
  uint32_t *progress;      // [n_chunks][64], zeroed by the launcher; nullptr: off
};
#ifndef TXM_I8_LEAD
#define TXM_I8_LEAD 2
#endif
constexpr uint32_t I8_LEAD = TXM_I8_LEAD;
constexpr int I8_THROTTLE_SPINS = 48;

// true when the int8 path can take this problem (device-sampler mode only)
bool i8_supported(int64_t N, int64_t C, int64_t nrep, int K);
// the pre-pass (window table, guard flags, fallback list) and the bootstrap kernel; partial sums land in part_x/part_u
int launch_i8_prepass(const I8Args &a, int K, hipStream_t st);  // (a.states: S states, the pointers of a ignored)
int launch_resample_i8(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st);
// the same contraction with the B operands built by the LDS transposing read (txm_resample_i8t.hip): one power per
// observable column, every order the int8 path serves
int launch_resample_i8t(const I8Args &a, int K, bool weighted, size_t prog_bytes, hipStream_t st);
// true when a call of this shape carries a second sample matrix inside its last pass (else the caller bootstraps it on its own)
bool i8t_carries_y(int64_t C, int K);
bool i8t_applicable(const double *x, int64_t ldx_s, int64_t C);  // C = all columns of the call
// narrow states (C <= 16 observables): column quads of the transposing-read kernel's quad-sharing variant (1, 2 or 4), 0 =
// the shape is not served by it; i8_cpad = the columns of a row of I8Args::part_x for the shape (4, 8, 16 or 32)
int i8t_narrow_nq(int64_t C_call, int K);
int i8_cpad(int64_t C_call, int K);
bool i8t_partials_summed(int64_t C_call, int K);  // the fused narrow kernel stores digit-summed slots for this shape
bool i8t_wide_summed(bool call_carries_y);         // ... and the fused wide kernel for a call with / without a second matrix

}  // namespace txm
