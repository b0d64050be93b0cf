/*
 * txmom.h -- C ABI of libtxmom.so: the MI355X (gfx950) moment engine behind
 * thermoextrap's central-(co)moment / bootstrap / derivative hot path.
 *
 * The reference (usnistgov/thermoextrap @ v0.6.0) has NO FFI for this path: its
 * boundary is the Python API of the third-party package cmomy 0.24.0 as called
 * from src/thermoextrap/data.py.  Each entry point below names the cmomy call
 * (and the reference call site, file:line under /root/reference) it replaces;
 * INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - return 0 on success, negative txm_status on failure; never throws;
 *     txm_last_error() gives a thread-local message.
 *   - every array pointer is a DEVICE pointer (HBM resident) unless the
 *     parameter name ends in `_host`; the caller owns all buffers; outputs
 *     are caller-allocated and fully overwritten.
 *   - all floating point is IEEE binary64, all indices/counts int64_t,
 *     strides are in ELEMENTS.
 *   - moment-state layout (cmomy convention, verified against the reference's
 *     notebook outputs): trailing dims [xmom = 2][umom = K = order + 1]:
 *        [0][0] = sum of weights   [1][0] = <x>   [0][1] = <u>
 *        [a][b] = <(x-<x>)^a (u-<u>)^b>  otherwise.
 *     1-D states: [0] = sum w, [1] = <u>, [k>=2] = <du^k>.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it and nothing synchronises unless stated.
 *   - `ws` is scratch in HBM of at least the size the matching *_ws_bytes()
 *     returns; no entry point allocates or frees device memory, so every call
 *     is legal inside hipGraph stream capture.
 */
#ifndef TXMOM_H
#define TXMOM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TXM_ABI_VERSION 2 /* 2: txm_sampler_spec.rep0 */
#define TXM_MAX_ORDER 8 /* K = order + 1 <= 9 */
#define TXM_SAMPLER_STREAM_VERSION 3 /* 3: one BTRS binomial per tree node (round 4); txm_sampler_stream_version() */

typedef enum txm_status {
  TXM_OK = 0,
  TXM_ERR_INVALID = -1,     /* bad argument (shape, order, null pointer, alignment) */
  TXM_ERR_HIP = -2,         /* a HIP runtime call failed; see txm_last_error() */
  TXM_ERR_WORKSPACE = -3,   /* ws too small */
  TXM_ERR_UNSUPPORTED = -4, /* valid request the library cannot serve */
  TXM_ERR_NO_DEVICE = -5    /* no gfx950 device visible */
} txm_status;

typedef void *txm_stream;

/* ---- runtime ------------------------------------------------------------ */
int txm_abi_version(void);
int txm_sampler_stream_version(void); /* the device sampler stream this build draws (tables are a function of it) */
const char *txm_csrc_sha(void); /* sha256 (16 hex digits) over thermoextrap_amd/csrc/*.hip|*.h this library was compiled from */
const char *txm_last_error(void);
/* Select the device for the calling thread and check it is a gfx950 part. */
int txm_init(int device);
int txm_device_count(int *count_host);
int txm_malloc(void **ptr_host, size_t bytes);
int txm_free(void *ptr);
int txm_memcpy_h2d(void *dst, const void *src_host, size_t bytes, txm_stream stream);
int txm_memcpy_d2h(void *dst_host, const void *src, size_t bytes, txm_stream stream);
int txm_memset(void *dst, int value, size_t bytes, txm_stream stream);
int txm_stream_sync(txm_stream stream);

/* ---- a1/a2: cmomy.wrap_reduce_vals ------------------------------------- */
/* Weighted central comoments of C observables against u, mom = (1, order).
 *   replaces cmomy.wrap_reduce_vals(xv, uv, weight=w, dim=rec, mom=(1, order))
 *   reference call sites: data.py:1632-1640 (DataCentralMomentsVals),
 *   data.py:1194-1203 (DataCentralMoments.from_vals), data.py:530-532,
 *   data.py:487-489 (DataValues*).
 * x[i*ldx_s + c*ldx_c], i < N, c < C.  Either ldx_c == 1 ((rec, val) layout,
 * core/xrutils.py:73-116) or ldx_s == 1 ((val, rec) layout).  w may be NULL.
 * out: [C][2][K].
 */
size_t txm_reduce_vals_ws_bytes(int64_t N, int64_t C, int order);
int txm_reduce_vals(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                    const double *w, int64_t N, int64_t C, int order, double *out, void *ws,
                    size_t ws_bytes, txm_stream stream);

/* ---- the same reduction in pieces: sample shards (SURVEY 8(e) partition (4)) and streaming accumulation ----------
 * The reduce kernels accumulate WEIGHT-SCALED POWER SUMS about a pivot {pivot_u, pivot_x[0..C)}:
 *     sums[c][0][j] = sum_i w_i (u_i - pivot_u)^j          sums[c][1][j] = sum_i w_i (x_ic - pivot_x[c]) (u_i - pivot_u)^j
 * Sums about ONE pivot add exactly like the samples they stand for, so N samples split over ranks (or over time) are:
 * one pivot everybody uses, per-shard sums, one addition in a fixed order, one shift to the cmomy state.
 *   txm_reduce_vals_pivot   the library's own estimate (strided means of at most 1024 samples) for a shard: [1 + C]
 *   txm_reduce_vals_sums    the sums of a shard about a GIVEN pivot: [C][2][K]            (ws: txm_reduce_vals_ws_bytes)
 *   txm_sums_to_state       out[C][2][K] = shift(sums[0] + sums[1] + ... + sums[n - 1]), sums [n][C][2][K], added in index
 *                           order -- an all-gathered stack of per-rank sums gives every rank the same bits
 *   replaces, for a sample-sharded array, the one cmomy.wrap_reduce_vals call of data.py:1632-1640 / 1194-1203.
 * txm_push_vals: cmomy's CentralMomentsData.push_vals(x, u, weight=w) -- accumulate a new chunk of samples into an
 * existing state [C][2][K] in place (a state of zeros is the empty accumulator): the chunk's sums about its own pivot,
 * the old state re-expressed about that pivot, added and shifted back.  (thermoextrap itself never calls push_vals --
 * SURVEY 0.7 -- but it is the streaming form of the reduction above.)  ws: txm_push_vals_ws_bytes. */
int txm_reduce_vals_pivot(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, int64_t N, int64_t C,
                          double *pivot, txm_stream stream);
int txm_reduce_vals_sums(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w,
                         int64_t N, int64_t C, int order, const double *pivot, double *sums, void *ws,
                         size_t ws_bytes, txm_stream stream);
int txm_sums_to_state(const double *sums, int64_t n, const double *pivot, int64_t C, int order, double *out,
                      txm_stream stream);
size_t txm_push_vals_ws_bytes(int64_t N, int64_t C, int order);
int txm_push_vals(double *state, const double *x, int64_t ldx_s, int64_t ldx_c, const double *u, const double *w,
                  int64_t N, int64_t C, int order, void *ws, size_t ws_bytes, txm_stream stream);

/* 1-D central moments of R independent series, mom = M - 1.
 *   replaces cmomy.wrap_reduce_vals(uv, weight=w, mom=order[+1])
 *   reference call sites: data.py:1183-1191 (x_is_u, mom = order + 1),
 *   data.py:485, 528; lnpi.py:278-282.
 * u[r*ldu_r + i*ldu_s]; ldu_s must be 1.  w (nullable) is shared by all rows.
 * out: [R][M].
 */
size_t txm_reduce_vals_1d_ws_bytes(int64_t N, int64_t R, int M);
int txm_reduce_vals_1d(const double *u, int64_t ldu_r, int64_t ldu_s, const double *w,
                       int64_t N, int64_t R, int M, double *out, void *ws, size_t ws_bytes,
                       txm_stream stream);

/* ---- sampler: cmomy.factory_sampler / IndexSampler ---------------------- */
/* indices[nrep][nsamp] -> freq[nrep][ndat] (histogram; cmomy indices_to_freq).
 *   reference: sampler built at data.py:1028-1037, 1344-1352, 1782-1789;
 *   tests/test_data.py:94-112 feeds explicit indices.
 * Returns TXM_ERR_INVALID if any index is outside [0, ndat) (checked on device,
 * reported after a stream sync).  ws: txm_indices_to_freq_ws_bytes() bytes; it holds the call's
 * error word, so concurrent calls on different streams share no state. */
size_t txm_indices_to_freq_ws_bytes(void);
int txm_indices_to_freq(const int64_t *indices, int64_t nrep, int64_t nsamp, int64_t ndat,
                        int64_t *freq, void *ws, size_t ws_bytes, txm_stream stream);

/* Device multinomial sampler ("scale mode").  The reference draws
 *   indices = rng.choice(ndat, (nrep, ndat), replace=True)   
 * and histograms them, which needs 16*nrep*ndat bytes of tables.  This
 * generates the SAME DISTRIBUTION of frequency tables (multinomial:
 * nsamp draws with replacement per replicate) from a
 * counter-based Philox4x32-10 stream keyed by (seed, stage, replicate, tile)
 * without ever materialising indices; oracle/philox_oracle.c restates the
 * stream bit for bit.  The sampler object is a plain struct (no hidden state).
 * Stream version 3 (round 4; TXM_SAMPLER_STREAM_VERSION): the per-tile draw counts come from recursive binomial
 * splitting over a count-balanced binary tree of tile ranges -- ONE binomial variate per node, Hormann's BTRS
 * (transformed rejection with squeeze, O(1) uniforms) restated with IEEE double + - * / floor in a fixed order so
 * that a CPU and the device produce the same integers, and the version-2 bitwise comparison for nodes with n p < 10;
 * the per-sample counts inside a tile from 10-bit fields (unchanged).  Tables of a given seed differ from versions
 * 1 and 2; the distribution is the same multinomial (BTRS: exact in real arithmetic; here to double rounding, 52-bit
 * uniforms and a < 1e-10 truncation of Stirling's series in the acceptance bound -- the class of
 * numpy.random.Generator.binomial; versions 1 and 2 were exact in integers at 4 x the cost).
 */
typedef struct txm_sampler_spec {
  uint64_t seed;
  int64_t nrep;  /* replicates of this call */
  int64_t ndat;  /* samples being resampled */
  int64_t nsamp; /* draws per replicate; 0 means ndat */
  int64_t rep0;  /* replicate r of this call draws STREAM replicate rep0 + r (rep0 + nrep <= 2^32).  The stream
                    is keyed by (seed, stream replicate, tile) and a replicate's draws depend on nothing else, so
                    the rows [a, b) of an (seed, nrep) table equal the (seed, b - a, rep0 = a) table bit for bit:
                    a rank that owns replicates [a, b) -- or the states [s0, s1) of a collection whose state s owns
                    replicates s * nrep ... -- reproduces its slab of the one-GPU result exactly (the reference
                    runs those loops serially: models.py:614-641).  0 for a whole table. */
} txm_sampler_spec;

/* Per-(replicate, tile) draw counts, tile = 1024 consecutive samples.
 * counts: [nrep][txm_sampler_ntiles(ndat)] uint32.  ndat <= 2^30, nrep <= 2^24; the workspace
 * is no longer used (the tile tree lives in LDS) but the query still returns a
 * valid size and the arguments are accepted. */
int64_t txm_sampler_ntiles(int64_t ndat);
size_t txm_sampler_counts_ws_bytes(const txm_sampler_spec *spec_host);
int txm_sampler_tile_counts(const txm_sampler_spec *spec_host, uint32_t *counts, void *ws,
                            size_t ws_bytes, txm_stream stream);
/* Materialise freq[nrep][ndat] int64 from the stream (testing / small sizes). */
int txm_sampler_freq(const txm_sampler_spec *spec_host, const uint32_t *counts, int64_t *freq,
                     txm_stream stream);

/* Per-SAMPLE draw counts of a slab of replicates as a table in HBM (round 5): what cmomy's freq table holds
 * (indices_to_freq, reached from data.py:1782-1789), at one byte per (replicate, sample), for the replicates
 * [rep_begin, rep_begin + nreps) of the spec only, laid out for the int8 bootstrap kernel's matrix operands:
 *   table[((g * ntiles + t) * 32 + s) * 4096 + q * 1024 + L * 16 + b]
 *     = draws of sample min(1024 t, ndat - 1024) + 32 s + 16 (L >> 5) + b   (0 for the samples a slid last tile
 *       shares with its predecessor)   in replicate rep_begin + 128 g + 32 q + (L & 31)   (0 past the spec's nrep)
 * -- the same Philox calls as txm_sampler_freq, bit for bit.  rep_begin is a multiple of 128; ndat >= 1024;
 * txm_sampler_count_table_bytes(ndat, nreps) = ceil(nreps / 128) * ntiles * 131072.  txm_resample_vals builds and
 * consumes such tables inside its workspace; the entry point exists for tests and for callers that keep a table. */
size_t txm_sampler_count_table_bytes(int64_t ndat, int64_t nreps);
int txm_sampler_count_table(const txm_sampler_spec *spec_host, const uint32_t *counts, int64_t rep_begin,
                            int64_t nreps, uint8_t *table, txm_stream stream);

/* ---- a3/a6: cmomy.wrap_resample_vals ------------------------------------ */
/* Sample-level bootstrap of central comoments: replicate r is the weighted
 * comoment state of the data with weights w_i * freq[r][i].
 *   replaces cmomy.wrap_resample_vals(xv, uv, weight=w, sampler=..., mom=(1, order))
 *   reference call sites: data.py:1803-1810 (DataCentralMomentsVals.resample),
 *   data.py:1354-1366 (DataCentralMoments.from_resample_vals).
 * Exactly one of `freq` (explicit table, [nrep][N] int64: "parity mode") and
 * `spec_host` + `counts` (device sampler: "scale mode") must be given.
 * pivot (nullable): [1 + C] doubles {pivot_u, pivot_x[0..C)} near the means;
 * NULL lets the library estimate one.  Requires ldx_c == 1.
 * out: [nrep][C][2][K]  (rep-major, i.e. already `.transpose(rep_dim, ...)`,
 * data.py:1812).
 *
 * Scale mode has two kernel families behind it and the library picks one per call:
 *   TXM_PATH_FP64  the contraction on the FP64 matrix pipe (any order, any C);
 *   TXM_PATH_INT8  the same sums on the int8 matrix pipe (orders 0..7, N >= 1024): per SCALING WINDOW -- 262144 samples on
 *                  long series (N >= 2^26), 65536 / 16384 / 4096 on shorter ones, a function of N alone -- every monomial is
 *                  scaled by the window maximum, rounded ONCE to a 51-bit fixed-point integer (error <= 2^-51 of the window
 *                  maximum, unbiased), split into seven signed 8-bit digits and accumulated exactly in int32; 32 columns
 *                  per launch.
 *   The rule of TXM_PATH_AUTO (txm_resample_path; tests/test_abi_cpu.py calls it on both sides of every threshold named here):
 *     - needs N >= 262144 and order <= 7;
 *     - NARROW states, C <= 16 observables in the call: order >= 1, and N >= 786432 at ANY replicate count or, below that,
 *       nrep >= 128 (order 0 stays FP64);
 *     - WIDE states, C > 16 (a last group of 1..16 columns behind full 32-column groups is allowed from order 1; at order 0
 *       it keeps the call on FP64): long series (N >= 786432) from nrep >= 32 at order >= 1 and nrep >= 100 at order 0;
 *       shorter series from nrep >= 64 at order >= 3, 128 at orders 1-2, 384 at order 0.
 *   Inside the int8 path two contraction kernels serve every state and agree BIT FOR BIT (same int32 sums, same flush):
 *     - the kernel that draws the per-sample counts in place (64 replicates per workgroup; orders 0-4 one pass over the
 *       sampler stream, 5-7 two): wide states at orders 3 and 4 without a second matrix at nrep <= 128, replicate counts that
 *       pad badly to 128 (4 * ceil128(nrep) > 5 * ceil64(nrep)), misaligned operands (x not 16-byte aligned or an odd row
 *       pitch); narrow states at nrep <= 64, where 128s pad worse than 64s (ceil128(nrep) > ceil64(nrep)), N < 786432, 13-16
 *       observables at order 4 (one fused pass against two table passes), a row shorter than a whole column quad (C = 1..3 in a
 *       tight array) (txm_resample.hip narrow_table_pays: the sweeps it is read from);
 *     - the count-table kernels (128 replicates per workgroup over ONE table of per-sample counts, txm_sampler_count_table --
 *       wide states: at most three row sets per pass, every call with a second matrix rides it; narrow states: the column
 *       quads and powers of the state in one pass, two for four quads from order 4): all other calls whose workspace holds the
 *       table (txm_resample_vals_ws_bytes_opts).  TXM_PATH_INT8_FUSED / TXM_PATH_INT8_TABLE force one of them;
 *       txm_resample_kernel reports the choice for a shape.
 *                  PRECISION GUARD (data dependent, automatic): the pre-pass also takes, per window, a robust
 *                  typical magnitude of the top-power monomial (the smallest of 64 group means of
 *                  |w du^order dx_c|); a window whose scale exceeds 275 sqrt(n) times it -- a heavy tail, an
 *                  outlier, weights spanning decades: the rounding would no longer stay within 1e-13 of what
 *                  the window contributes -- is contracted by the FP64 kernel instead, in the same call, and
 *                  the two sets of partial sums are added.  Ordinary data flags nothing.
 * txm_resample_path reports the shape-based choice.  The kernel of ONE CALL is chosen by txm_resample_opts.path
 * (TXM_PATH_AUTO = that rule); txm_set_resample_path is a process-wide default for TXM_PATH_AUTO calls kept for tests
 * and A/B timing (initially TXM_PATH_AUTO; the library reads NO environment variable) -- calls that pass a path
 * share no mutable state and are re-entrant per (workspace, stream).
 *
 * txm_resample_opts (HOST struct, NULL = all defaults):
 *   path        TXM_PATH_AUTO / TXM_PATH_FP64 / TXM_PATH_INT8 (/ _INT8_FUSED / _INT8_TABLE: which int8 kernel) for this call.
 *   info        DEVICE pointer to 4 int64 (nullable), written on the stream, no synchronisation:
 *               [0] path taken, [1] scaling windows x column groups, [2] how many of them the precision guard sent
 *               to the FP64 kernel, [3] bit 0: the pre-pass tables came from `prep` (below) instead of being computed; bit 1: the
 *               column groups ran a count-table kernel (else the kernel that draws in place).
 *   prep, prep_bytes, prep_valid
 *               The int8 path's pre-pass (pivot + per-window scale table, guard flags and fallback list) depends on
 *               (x, u, w, pivot, N, C, nrep, order) only -- not on the sampler -- and costs one more read of the samples.
 *               A caller that bootstraps the SAME data repeatedly (the reference caches per data object: data.py:285,
 *               844-942) passes a device buffer of txm_resample_prep_bytes() that it keeps next to the data: a call with
 *               prep_valid == 0 computes the tables into it, later calls with prep_valid == 1 reuse them (one HBM pass
 *               and three launches fewer).  The caller invalidates by passing 0 again; the library keeps no state.
 *   y, ldy_s, out_y
 *               optional second sample matrix y[N][C] (row pitch ldy_s) resampled with the SAME replicate weights:
 *               out_y[nrep][C] = sum_i f_ri w_i y_ic / sum_i f_ri w_i   (the per-replicate mean; DEVICE pointers).
 *               This is VolumeDataCallback's <dx/dq> over the bootstrap sample (reference volume.py:121-134,
 *               dxdqv[sampler.indices].mean): carried by the same pass over the sampler stream instead of a
 *               separate order-0 bootstrap.
 * txm_resample_vals_info is the older synchronising read-back of [0..2] from the workspace of the last call.
 */
#define TXM_PATH_AUTO (-1)
#define TXM_PATH_FP64 0
#define TXM_PATH_INT8 1
#define TXM_PATH_INT8_FUSED 2 /* int8 path, wide states on the kernel that draws the per-sample counts in place (txm_resample_i8t.hip) */
#define TXM_PATH_INT8_TABLE 3 /* int8 path, wide states on the count-table kernel wherever it applies (txm_resample_i8g.hip) */
typedef struct txm_resample_opts {
  int32_t path;
  int32_t prep_valid;
  void *prep;
  size_t prep_bytes;
  int64_t *info;
  const double *y;
  int64_t ldy_s;
  double *out_y;
} txm_resample_opts;
int txm_resample_path(int64_t N, int64_t C, int64_t nrep, int order);
/* The contraction kernel ONE scale-mode call with these options runs when it is handed txm_resample_vals_ws_bytes_opts(N, C,
 * nrep, order, path, has_y) bytes: TXM_PATH_FP64, TXM_PATH_INT8_FUSED or TXM_PATH_INT8_TABLE, OR-ed with TXM_KERNEL_WITH_Y
 * when that kernel carries opts.y as a row set of its own (otherwise y is bootstrapped by a separate order-0 pass behind the
 * call).  `aligned` != 0: x (and y) are 16-byte aligned with an even row pitch >= C rounded up to 4 (the count-table kernel's
 * operand requirement; anything else runs the fused kernel).  What a caller-held pre-pass block (opts.prep) contains depends
 * on exactly this word: WITH_Y blocks hold y's pivot, scales and guard flags, the others do not -- reuse a block
 * (prep_valid = 1) only for calls on the same tensors and (N, C, order) that return the same word here.  Replicate slabs of
 * one bootstrap (spec.rep0) should pass the whole call's kernel as their opts.path, so that a short last slab does not fall on
 * the other side of a replicate-count threshold of the rule. */
#define TXM_KERNEL_WITH_Y 0x100
int txm_resample_kernel(int64_t N, int64_t C, int64_t nrep, int order, int path, int has_y, int aligned);
/* `aligned` for a pair of operands as txm_resample_vals would see them (device or host pointer values: only the address bits and
 * the row pitches are looked at; y may be NULL): 1 when the count-table kernel can take them */
int txm_resample_operands_aligned(const double *x, int64_t ldx_s, int64_t C, const double *y, int64_t ldy_s);
int txm_set_resample_path(int path);
int txm_resample_vals_info(const void *ws, int64_t N, int64_t C, int64_t nrep, int order,
                           int64_t *info_host, txm_stream stream);
size_t txm_resample_prep_bytes(int64_t N, int64_t C, int64_t nrep, int order);
size_t txm_resample_vals_ws_bytes(int64_t N, int64_t C, int64_t nrep, int order);
/* the same for a call that will pass txm_resample_opts.path = `path` and a second matrix iff has_y: the int8 path's count-table
 * kernel (wide states; txm_resample_vals_ws_bytes = path TXM_PATH_AUTO, has_y 0) keeps one byte per (replicate padded to 128,
 * sample) in the workspace -- 12.8 GB per 128 replicates at N = 1e8.  A call given less workspace than this, but at least
 * what TXM_PATH_INT8_FUSED needs, runs the kernel that draws the counts in place: same bits, other speed.  Callers bound the
 * workspace by bootstrapping replicate slabs (spec.rep0): rows [a, b) of a call equal the (b - a)-replicate call at rep0 = a. */
size_t txm_resample_vals_ws_bytes_opts(int64_t N, int64_t C, int64_t nrep, int order, int path, int has_y);
/* extra workspace, BEHIND txm_resample_vals_ws_bytes(), that a call with opts.y needs (ws_bytes >= the sum); a call
 * without opts.y needs none of it (~2 GB at N = 1e8, C = 32, nrep = 1000) */
size_t txm_resample_y_ws_bytes(int64_t N, int64_t C, int64_t nrep);
/* 1 when the int8 kernel can take the shape at all (N >= one sampler tile, order <= 7, ...): what TXM_PATH_INT8 needs;
 * txm_resample_path() is the rule TXM_PATH_AUTO applies on top of it */
int txm_resample_i8_supported(int64_t N, int64_t C, int64_t nrep, int order);
int txm_resample_vals(const double *x, int64_t ldx_s, int64_t ldx_c, const double *u,
                      const double *w, int64_t N, int64_t C, int order, int64_t nrep,
                      const int64_t *freq, const txm_sampler_spec *spec_host,
                      const uint32_t *counts, const double *pivot, double *out,
                      const txm_resample_opts *opts_host, void *ws, size_t ws_bytes, txm_stream stream);

/* ---- (f-1) S state points of one shape in one set of launches ------------- */
/* The reference handles a collection of states with a serial Python loop
 * (StateCollection.resample / map_concat, models.py:614-671; one cmomy call per state).
 * These entry points take S state points that share (N, C, order[, nrep]) -- the 16 lnPi states of
 * BASELINE config 3, the 64 states of the GPR loop of config 5 -- and run ONE launch per kernel with
 * the state on a grid axis.  `states_host` is a HOST array of S device-pointer triples (w NULL for all
 * states or for none); it is copied into the workspace by the call (not stream-capturable).
 *   txm_reduce_vals_batched:   out [S][C][2][K]            (= S x txm_reduce_vals, x row-major, pitch ldx_s)
 *   txm_resample_vals_batched: out [S][nrep][C][2][K]      (= S x txm_resample_vals; FP64 kernel, or the int8 path below)
 *     sampler: ONE spec over S * nrep replicates of ndat = N (state s owns replicates s*nrep ..
 *     (s+1)*nrep - 1 of the stream, so the states' bootstrap samples are independent), with its
 *     counts [S * nrep][ntiles]; or an explicit freq table [S * nrep][N]. */
typedef struct txm_state_ptrs {
  const double *x; /* [N][ldx_s] */
  const double *u; /* [N] */
  const double *w; /* [N] or NULL */
} txm_state_ptrs;
size_t txm_reduce_vals_batched_ws_bytes(int64_t S, int64_t N, int64_t C, int order);
int txm_reduce_vals_batched(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N,
                            int64_t C, int order, double *out, void *ws, size_t ws_bytes,
                            txm_stream stream);
size_t txm_resample_vals_batched_ws_bytes(int64_t S, int64_t N, int64_t C, int64_t nrep, int order);
int txm_resample_vals_batched(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N,
                              int64_t C, int order, int64_t nrep, const int64_t *freq,
                              const txm_sampler_spec *spec_host, const uint32_t *counts, double *out,
                              void *ws, size_t ws_bytes, txm_stream stream);
/* The same with per-call options (round 4; txm_resample_vals_batched == opts_host NULL).  Narrow states (C <= 16
 * observables, order >= 1, and N >= 786432 at any replicate count or N >= 262144 from 128 replicates per state -- the rule
 * of the single call, txm_resample_path; or opts.path = TXM_PATH_INT8 wherever txm_resample_i8_supported) run the int8 path of txm_resample_vals with the state on a grid axis of EVERY kernel of it
 * (pre-pass, bootstrap kernel, the precision guard's FP64 fallback, finalize): state s of the batch is, bit for bit, the
 * single call on state s with spec.rep0 + s * nrep.  Fields of txm_resample_opts used: path, info, prep / prep_bytes /
 * prep_valid (ONE block of txm_resample_batched_prep_bytes() for the S states -- their pivots, window tables, guard flags
 * and fallback lists; 0 bytes = the shape never takes the int8 path); y / out_y must be NULL.
 * Reference loops replaced: models.py:635-641, gpr_active/active_utils.py:896-925. */
size_t txm_resample_batched_prep_bytes(int64_t S, int64_t N, int64_t C, int64_t nrep, int order);
/* the kernel a TXM_PATH_AUTO batched call takes (TXM_PATH_FP64 / TXM_PATH_INT8): bind a prep block only for INT8 */
int txm_resample_batched_path(int64_t S, int64_t N, int64_t C, int64_t nrep, int order);
int txm_resample_vals_batched_opts(const txm_state_ptrs *states_host, int64_t S, int64_t ldx_s, int64_t N,
                                   int64_t C, int order, int64_t nrep, const int64_t *freq,
                                   const txm_sampler_spec *spec_host, const uint32_t *counts, double *out,
                                   const txm_resample_opts *opts_host, void *ws, size_t ws_bytes, txm_stream stream);

/* ---- a4/a5: CentralMomentsData.reduce / resample_and_reduce ------------- */
/* Block bootstrap of pre-reduced states: replicate r merges freq[r][i] copies
 * of state i.   data [nrec][C][2][K], freq [nrep][nrec] -> out [nrep][C][2][K]
 *   replaces CentralMomentsData.resample_and_reduce(sampler, dim=rec)
 *   reference call site: data.py:1048-1052 (DataCentralMoments.resample).
 * freq == NULL means nrep = 1 with every count 1, i.e.
 *   CentralMomentsData.reduce(dim=rec)  (data.py:996, DataCentralMoments.reduce). */
size_t txm_resample_data_ws_bytes(int64_t nrec, int64_t C, int order);
int txm_resample_data(const double *data, const int64_t *freq, int64_t nrec, int64_t C,
                      int64_t nrep, int order, double *out, void *ws, size_t ws_bytes,
                      txm_stream stream);

/* ---- a8/a9: cmom()/rmom()/convert.moments_type -------------------------- */
/* n states [2][K]; to_central = 0: central -> raw (CentralMomentsData.rmom(),
 * data.py:845-847); 1: raw -> central (convert.moments_type(to="central"),
 * data.py:1109-1115).  [0][0] (weight) is carried over.  in == out allowed. */
int txm_convert_cov(const double *in, double *out, int64_t n, int order, int to_central,
                    txm_stream stream);
/* n states [M], 1-D moments. */
int txm_convert_1d(const double *in, double *out, int64_t n, int M, int to_central,
                   txm_stream stream);

/* ---- a10/a11: Derivatives.derivs ---------------------------------------- */
/* Table-driven evaluation of the derivative polynomials that the reference
 * builds with sympy + lambdify and evaluates with one xarray isel per symbol
 * (models.py:317-383, beta.py:532-573).  A function is
 *     f = sum_t coef[t] * prod_{k < nfac[t]} atom(fac[t][k])^pow[t][k]   (* optional -log, see flags)
 * where an atom addresses one scalar per output element e = (rep, val):
 *     value = src[atom.src][ rep*atom.s_rep + val*atom.s_val + atom.offset ].
 * Tables are built on the host from the symbolic recursion (thermoextrap_amd/beta.py).
 */
typedef struct txm_atom {
  int32_t src;    /* index into srcs[] */
  int32_t pad;
  int64_t offset; /* element offset of the scalar inside one (rep, val) cell */
  int64_t s_rep;  /* element stride per replicate */
  int64_t s_val;  /* element stride per value column */
} txm_atom;

#define TXM_FUNC_PLAIN 0
#define TXM_FUNC_MINUS_LOG 1 /* f = -log(atoms[log_atom]) + polynomial part */

typedef struct txm_poly_table {
  int32_t n_funcs;           /* number of functions (orders 0..n_funcs-1) */
  int32_t n_atoms;
  int32_t n_terms;           /* total terms over all functions */
  int32_t n_factors;         /* total factors over all terms */
  int32_t log_atom;          /* atom used by TXM_FUNC_MINUS_LOG functions */
  int32_t pad;
  const txm_atom *atoms;     /* [n_atoms]                          (device) */
  const int32_t *func_term0; /* [n_funcs + 1] term range per func  (device) */
  const int32_t *func_flags; /* [n_funcs] TXM_FUNC_*               (device) */
  const double *coef;        /* [n_terms]                          (device) */
  const int32_t *term_fac0;  /* [n_terms + 1] factor range         (device) */
  const int32_t *fac_atom;   /* [n_factors] atom id                (device) */
  const int32_t *fac_pow;    /* [n_factors] integer power, may be negative (device) */
} txm_poly_table;

/* srcs: device array of n_srcs device pointers (const double*).
 * out: [n_funcs][nrep][nval].  The struct itself is read on the host. */
int txm_eval_poly(const txm_poly_table *table_host, const double *const *srcs, int32_t n_srcs,
                  int64_t nrep, int64_t nval, double *out, txm_stream stream);

/* ---- (f-1) ExtrapModel.predict: Taylor series of the derivative table -------- */
/* derivs [n_ord][M] (the output of txm_eval_poly, M = nrep * nval), dalpha [n_alpha] (device):
 *   term[a][k][m] = dalpha[a]^k * (derivs[k][m] * (1 / k!))
 *   mode TXM_TAYLOR_SUM    : out[a][m]    = sum_k term[a][k][m]           (k ascending)
 *   mode TXM_TAYLOR_CUMSUM : out[a][k][m] = sum_{k' <= k} term[a][k'][m]
 *   mode TXM_TAYLOR_TERMS  : out[a][k][m] = term[a][k][m]
 *   replaces ExtrapModel.predict = coefs * dalpha**p summed / cumsummed over the order dim
 *   reference: src/thermoextrap/models.py:479-565 (coefs: models.py:449-477, taylor_series_norm :55-70).
 * One launch over (alpha, rep, val); n_ord <= 16. */
#define TXM_TAYLOR_SUM 0
#define TXM_TAYLOR_CUMSUM 1
#define TXM_TAYLOR_TERMS 2
int txm_predict_taylor(const double *derivs, int32_t n_ord, int64_t M, const double *dalpha,
                       int64_t n_alpha, int32_t mode, double *out, txm_stream stream);

/* ---- (f-2) covariance of the derivatives over bootstrap replicates --------- */
/* vals [n_ord][nrep][nval] -> cov [nval][n_ord][n_ord] with ddof = 1 (numpy.cov),
 * the per-output covariance that gpr_active.input_GP_from_state feeds to the GP
 *   replaces np.cov(resamp_derivs.values[..., k]) for every output k
 *   reference: src/thermoextrap/gpr_active/active_utils.py:134-140.  n_ord <= 16. */
int txm_cov_over_rep(const double *vals, int32_t n_ord, int64_t nrep, int64_t nval, double *cov,
                     txm_stream stream);

/* ---- (f-4) PerturbModel.predict: exponential reweighting ------------------- */
/* out[a][c] = sum_i x[i][c] e^{-dalpha[a] (u_i - uref[a])} / sum_i e^{-dalpha[a] (u_i - uref[a])}
 * for n_alpha perturbations in ONE pass over the samples; uref[a] is the u extreme that
 * makes the largest exponent 0 (min u for dalpha > 0, max u for dalpha < 0), computed by
 * the library.   replaces PerturbModel.predict (reference models.py:1019-1039:
 * exp(dalpha_uv - max) weights, xr.dot / mean).  x row-major (ldx_c == 1); freq (nullable,
 * [nrep][N] int64) adds bootstrap weights -> out [nrep][n_alpha][C]; n_alpha <= 8.
 * dalpha_host: host array.  ws: txm_perturb_ws_bytes. */
size_t txm_perturb_ws_bytes(int64_t N, int64_t C, int32_t n_alpha, int64_t nrep);
int txm_perturb(const double *x, int64_t ldx_s, const double *u, int64_t N, int64_t C,
                const double *dalpha_host, int32_t n_alpha, const int64_t *freq, int64_t nrep,
                double *out, void *ws, size_t ws_bytes, txm_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* TXMOM_H */
