// txm_sampler.hip -- sampler kernels: indices -> freq histogram (parity mode,
// cmomy's indices_to_freq) and the counter-based exact multinomial sampler
// (scale mode) that replaces `rng.choice(ndat, (nrep, ndat))` + histogram
// (cmomy factory_sampler as reached from data.py:1782-1789) without ever
// materialising the (nrep, ndat) index/freq tables.
//
// Tile counts: recursive binomial splitting over the tree of tile ranges, one
//          workgroup per replicate (sampler_tree_kernel).
// Per-sample counts inside a tile ("stage 3"): txm_sampler.h, executed inside the
//          bootstrap kernels (txm_resample*.hip), tile by tile, straight into LDS.
#include "txm_sampler.h"

namespace txm {

// `err` is a word of the CALLER's workspace: concurrent calls on different streams do not share it
__global__ __launch_bounds__(256) void indices_to_freq_kernel(const int64_t *__restrict__ idx,
                                                              int64_t nrep, int64_t nsamp,
                                                              int64_t ndat,
                                                              int64_t *__restrict__ freq,
                                                              int *__restrict__ err) {
  for (int64_t r = blockIdx.y; r < nrep; r += gridDim.y) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nsamp;
         k += (int64_t)gridDim.x * blockDim.x) {
      const int64_t j = idx[r * nsamp + k];
      if (j < 0 || j >= ndat) {
        *err = 1;
      } else {
        atomicAdd(reinterpret_cast<unsigned long long *>(freq + r * ndat + j), 1ULL);
      }
    }
  }
}

// ---- tile counts: recursive binomial splitting (stream v2, oracle/philox_oracle.c) -------------------
// One workgroup per replicate walks the binary tree of tile ranges.  A node's n draws are split between its
// children by counting 1 bits in n bits of the node's Philox bit stream: one v_bcnt per 32 draws and level,
// no histogram, no atomics, no rejection.  Levels with few nodes are split by the whole workgroup (block
// reduction per node), levels with >= 4 nodes per wave by one wave per node, and the subtrees below level
// ka = k - ST_DEPTH by one wave each in wave-private LDS with g <= 64 lanes per node, g chosen so that every
// lane has a few Philox calls per node.
constexpr int ST_DEPTH = 7;        // levels of a wave-private subtree (2^7 leaves)
constexpr int ST_BLOCK_NODES = 4;  // nodes per wave below which a level is split by the whole workgroup

__device__ __forceinline__ uint32_t popc4(const Philox4 &o) {
  return (uint32_t)(__popc(o.w[0]) + __popc(o.w[1]) + __popc(o.w[2]) + __popc(o.w[3]));
}

// this lane's share of the 1 bits among the first `nbits` bits of stream (h, tagj): calls sub, sub + g, ...
// The call index is the SECOND counter word: rounds 1-3 then need 2 instead of 6 multiplies per call (the rest is
// the same for all calls of a node and hoisted out of the loop).
// (g a power of two); the partial last call belongs to the lane whose turn it is.
__device__ __forceinline__ uint32_t stream_ones_partial(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r,
                                                        uint32_t tagj, uint32_t nbits, uint32_t sub, uint32_t g) {
  const uint32_t full = nbits >> 7, tail = nbits & 127u;
  uint32_t acc = 0;
  for (uint32_t c = sub; c < full; c += g) acc += popc4(philox4x32_10<true>(h, c, r, tagj, k0, k1));
  if (tail != 0u && (full & (g - 1u)) == sub) {
    const Philox4 o = philox4x32_10<true>(h, full, r, tagj, k0, k1);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int nb = (int)tail - 32 * w;
      const uint32_t m = nb >= 32 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << nb) - 1u));
      acc += (uint32_t)__popc(o.w[w] & m);
    }
  }
  return acc;
}

__device__ __forceinline__ int64_t tree_node_size(int64_t ndat, int k, int l, int64_t i) {
  const int64_t span = (int64_t)SM_T << (k - l);
  const int64_t lo = i * span;
  if (lo >= ndat) return 0;
  return (ndat - lo < span) ? ndat - lo : span;
}

// left ~ Binomial(n, A / (A + B)): all n uniforms are compared with p bit by bit at once.  SUM(v) adds v over the
// lanes that share the node (and returns the total on each of them).
template <class Sum>
__device__ __forceinline__ uint32_t split_left(uint32_t k0, uint32_t k1, uint32_t h, uint32_t r, uint32_t n,
                                               int64_t A, int64_t B, uint32_t sub, uint32_t g, Sum sum) {
  if (B == 0) return n;
  uint64_t rem = (uint64_t)A;
  const uint64_t den = (uint64_t)A + (uint64_t)B;
  uint32_t tie = n, left = 0;
  for (uint32_t j = 0; tie > 0u && j < 255u; ++j) {
    rem <<= 1;
    const bool pj = rem >= den;
    if (pj) rem -= den;
    const uint32_t K = sum(stream_ones_partial(k0, k1, h, r, 1u + 256u * j, tie, sub, g));
    if (pj) {
      left += tie - K;
      tie = K;
    } else {
      tie -= K;
    }
    if (rem == 0) break;
  }
  return left;
}

// grid nrep, block 64 * nwaves (a power of two), dynamic LDS: heap[2^(ka+1)] + nwaves * sub[2^(depth+1)] + red[nwaves]
__global__ __launch_bounds__(1024) void sampler_tree_kernel(uint32_t k0, uint32_t k1key, uint32_t nsamp,
                                                            SamplerGeom g, uint32_t rep0,
                                                            uint32_t *__restrict__ counts) {
  extern __shared__ uint32_t tree_lds[];
  const int k = g.k;
  const int depth = k < ST_DEPTH ? k : ST_DEPTH;
  const int ka = k - depth;
  const int nwaves = (int)(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint32_t *heap = tree_lds;                                   // levels 0 .. ka, heap indexed
  uint32_t *sub_all = heap + ((size_t)2 << ka);                // per wave: subtree levels 0 .. depth
  uint32_t *red = sub_all + (size_t)nwaves * ((size_t)2 << depth);
  const uint32_t r = rep0 + blockIdx.x;  // replicate of the STREAM; row blockIdx.x of this call's table

  for (uint32_t q = threadIdx.x; q < ((uint32_t)2 << ka); q += blockDim.x) heap[q] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) heap[1] = nsamp;
  __syncthreads();

  auto block_sum = [&](uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    uint32_t s = 0;
    for (int w = 0; w < nwaves; ++w) s += red[w];
    return s;
  };
  auto wave_sum = [&](uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
  };

  // ---- levels 0 .. ka-1: counts in the workgroup's heap ----
  for (int l = 0; l < ka; ++l) {
    const uint32_t nn = 1u << l;
    const int64_t span = (int64_t)1 << (k - l);                      // tiles per node
    const uint32_t nreal = (uint32_t)((g.ntiles + span - 1) / span);  // nodes that cover samples
    if (nreal < (uint32_t)(ST_BLOCK_NODES * nwaves)) {
      for (uint32_t i = 0; i < nreal; ++i) {
        const uint32_t n = heap[nn + i];
        const uint32_t left = split_left(k0, k1key, nn + i, r, n, tree_node_size(g.ndat, k, l + 1, 2 * (int64_t)i),
                                         tree_node_size(g.ndat, k, l + 1, 2 * (int64_t)i + 1), threadIdx.x,
                                         blockDim.x, block_sum);
        if (threadIdx.x == 0) {
          heap[2 * nn + 2 * i] = left;
          heap[2 * nn + 2 * i + 1] = n - left;
        }
      }
    } else {
      for (uint32_t i = (uint32_t)wave; i < nreal; i += (uint32_t)nwaves) {
        const uint32_t n = heap[nn + i];
        const uint32_t left = split_left(k0, k1key, nn + i, r, n, tree_node_size(g.ndat, k, l + 1, 2 * (int64_t)i),
                                         tree_node_size(g.ndat, k, l + 1, 2 * (int64_t)i + 1), (uint32_t)lane, 64u,
                                         wave_sum);
        if (lane == 0) {
          heap[2 * nn + 2 * i] = left;
          heap[2 * nn + 2 * i + 1] = n - left;
        }
      }
    }
    __syncthreads();
  }

  // ---- subtrees rooted at level ka: one wave each, counts in wave-private LDS ----
  uint32_t *sh = sub_all + (size_t)wave * ((size_t)2 << depth);
  const uint32_t nroots = (uint32_t)((g.ntiles + ((int64_t)1 << depth) - 1) >> depth);
  for (uint32_t s = (uint32_t)wave; s < nroots; s += (uint32_t)nwaves) {
    if (lane == 0) sh[1] = heap[((uint32_t)1 << ka) + s];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int d = 0; d < depth; ++d) {
      const int l = ka + d;
      const uint32_t nn = 1u << d;
      // lanes per node: the largest power of two <= (expected Philox calls per node) / 3, within [1, 64]
      const uint32_t ecalls = (uint32_t)(((uint64_t)nsamp >> l) >> 7);
      uint32_t gsz = 64u;
      while (gsz > 1u && gsz * 3u > ecalls) gsz >>= 1;
      const uint32_t per = 64u / gsz;  // nodes per pass of the wave
      const uint32_t sub = (uint32_t)lane & (gsz - 1u);
      for (uint32_t base = 0; base < nn; base += per) {
        const uint32_t jn = base + (uint32_t)lane / gsz;
        const bool act = jn < nn;
        const uint32_t n = act ? sh[nn + jn] : 0u;
        const int64_t gi = ((int64_t)s << d) + jn;  // node index within level l
        const int64_t A = tree_node_size(g.ndat, k, l + 1, 2 * gi), B = tree_node_size(g.ndat, k, l + 1, 2 * gi + 1);
        const uint32_t left = split_left(k0, k1key, (1u << l) + (uint32_t)gi, r, n, A, B, sub, gsz, [&](uint32_t v) {
          for (uint32_t o = gsz >> 1; o > 0u; o >>= 1) v += (uint32_t)__shfl_xor((int)v, (int)o);
          return v;
        });
        if (act && sub == 0u) {
          sh[2 * nn + 2 * jn] = left;
          sh[2 * nn + 2 * jn + 1] = n - left;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t j = (uint32_t)lane; j < (1u << depth); j += 64u) {
      const int64_t t = ((int64_t)s << depth) + j;
      if (t < g.ntiles) counts[(size_t)blockIdx.x * g.ntiles + t] = sh[(1u << depth) + j];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- materialise freq (testing / small sizes) -------------------------------
// one wave per (r, t); block 256 = 4 waves, each with a private 1024-bin tile.
__global__ __launch_bounds__(256) void sampler_freq_kernel(uint32_t k0, uint32_t k1key,
                                                           int64_t nrep, int64_t ndat,
                                                           SamplerGeom g, uint32_t rep0,
                                                           const uint32_t *__restrict__ counts,
                                                           int64_t *__restrict__ freq) {
  __shared__ uint32_t tile[4][SM_T];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t task = (int64_t)blockIdx.x * 4 + wave;
  const bool active = task < nrep * g.ntiles;
  const uint32_t r = active ? (uint32_t)(task / g.ntiles) : 0u;
  const uint32_t t = active ? (uint32_t)(task % g.ntiles) : 0u;
  for (int i = lane; i < SM_T; i += 64) tile[wave][i] = 0u;
  __syncthreads();
  const uint32_t tsize = (t == (uint32_t)g.ntiles - 1u) ? (uint32_t)g.last_tile_size : (uint32_t)SM_T;
  if (active) {
    const uint32_t n = counts[(size_t)r * g.ntiles + t];
    uint32_t *tl = tile[wave];
    sampler_fine_tile(k0, k1key, rep0 + r, t, n, tsize, lane, [&](uint32_t off) { atomicAdd(&tl[off], 1u); });
  }
  __syncthreads();
  if (active) {
    for (uint32_t i = lane; i < tsize; i += 64)
      freq[(size_t)r * ndat + (size_t)t * SM_T + i] = (int64_t)tile[wave][i];
  }
}

}  // namespace txm

using namespace txm;

extern "C" size_t txm_indices_to_freq_ws_bytes(void) { return 256; }

extern "C" int txm_indices_to_freq(const int64_t *indices, int64_t nrep, int64_t nsamp,
                                   int64_t ndat, int64_t *freq, void *ws, size_t ws_bytes,
                                   txm_stream stream) {
  TXM_REQUIRE(indices && freq && ws, "indices_to_freq: null pointer");
  TXM_REQUIRE(nrep >= 1 && nsamp >= 1 && ndat >= 1, "indices_to_freq: bad sizes");
  if (ws_bytes < txm_indices_to_freq_ws_bytes()) {
    set_error("indices_to_freq: workspace too small (%zu < 256)", ws_bytes);
    return TXM_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int *err = reinterpret_cast<int *>(ws);
  TXM_HIP(hipMemsetAsync(freq, 0, sizeof(int64_t) * (size_t)nrep * ndat, st));
  TXM_HIP(hipMemsetAsync(err, 0, sizeof(int), st));
  int gx = (int)cdiv(nsamp, 256 * 4);
  if (gx > 4096) gx = 4096;
  const unsigned gy = (unsigned)(nrep < 65535 ? nrep : 65535);
  hipLaunchKernelGGL(indices_to_freq_kernel, dim3(gx, gy), dim3(256), 0, st, indices, nrep, nsamp, ndat, freq, err);
  TXM_LAUNCH_CHECK();
  int flag = 0;
  TXM_HIP(hipMemcpyAsync(&flag, err, sizeof(int), hipMemcpyDeviceToHost, st));
  TXM_HIP(hipStreamSynchronize(st));
  if (flag) {
    set_error("indices_to_freq: index outside [0, %lld)", (long long)ndat);
    return TXM_ERR_INVALID;
  }
  return TXM_OK;
}

extern "C" int64_t txm_sampler_ntiles(int64_t ndat) { return ndat < 1 ? 0 : (ndat + SM_T - 1) / SM_T; }

static int check_spec(const txm_sampler_spec *sp, SamplerGeom *g, int64_t *nsamp) {
  TXM_REQUIRE(sp, "sampler: null spec");
  TXM_REQUIRE(sp->nrep >= 1 && sp->nrep <= ((int64_t)1 << 24), "sampler: nrep=%lld outside [1, 2^24]",
              (long long)sp->nrep);
  TXM_REQUIRE(sp->rep0 >= 0 && sp->rep0 + sp->nrep <= ((int64_t)1 << 32),
              "sampler: stream replicates [%lld, %lld) outside [0, 2^32)", (long long)sp->rep0,
              (long long)(sp->rep0 + sp->nrep));
  TXM_REQUIRE(sp->ndat >= 1, "sampler: ndat < 1");
  *nsamp = sp->nsamp > 0 ? sp->nsamp : sp->ndat;
  TXM_REQUIRE(*nsamp < ((int64_t)1 << 32), "sampler: nsamp >= 2^32 unsupported");
  // per-sample counts live in 8-bit LDS counters inside the bootstrap kernel: keep
  // the mean count <= 16 so that an overflow (count >= 256) is beyond any reachable tail
  TXM_REQUIRE(*nsamp <= 16 * sp->ndat, "sampler: nsamp > 16 * ndat unsupported (8-bit per-sample counters)");
  if (sampler_geometry(sp->ndat, g) != 0) {
    set_error("sampler: ndat=%lld too large (max 2^30)", (long long)sp->ndat);
    return TXM_ERR_UNSUPPORTED;
  }
  return TXM_OK;
}

extern "C" size_t txm_sampler_counts_ws_bytes(const txm_sampler_spec *sp) {
  SamplerGeom g;
  int64_t nsamp;
  if (check_spec(sp, &g, &nsamp) != TXM_OK) return 0;
  return 256;  // the tile tree lives in LDS; kept so that callers' workspace plumbing stays valid
}

extern "C" int txm_sampler_tile_counts(const txm_sampler_spec *sp, uint32_t *counts, void *ws,
                                       size_t ws_bytes, txm_stream stream) {
  SamplerGeom g;
  int64_t nsamp;
  int rc = check_spec(sp, &g, &nsamp);
  if (rc != TXM_OK) return rc;
  TXM_REQUIRE(counts, "sampler: null counts");
  (void)ws;
  (void)ws_bytes;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t k0 = (uint32_t)sp->seed, k1 = (uint32_t)(sp->seed >> 32);
  // 16 waves per replicate when a replicate has enough draws to feed them, 4 otherwise
  const int nwaves = nsamp >= ((int64_t)1 << 22) ? 16 : 4;
  const int depth = g.k < ST_DEPTH ? g.k : ST_DEPTH, ka = g.k - depth;
  const size_t lds = (((size_t)2 << ka) + (size_t)nwaves * ((size_t)2 << depth) + (size_t)nwaves) * sizeof(uint32_t);
  TXM_SET_MAX_LDS(sampler_tree_kernel, 160 * 1024);
  hipLaunchKernelGGL(sampler_tree_kernel, dim3((unsigned)sp->nrep), dim3(64 * nwaves), lds, st, k0, k1,
                     (uint32_t)nsamp, g, (uint32_t)sp->rep0, counts);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

extern "C" int txm_sampler_freq(const txm_sampler_spec *sp, const uint32_t *counts, int64_t *freq,
                                txm_stream stream) {
  SamplerGeom g;
  int64_t nsamp;
  int rc = check_spec(sp, &g, &nsamp);
  if (rc != TXM_OK) return rc;
  TXM_REQUIRE(counts && freq, "sampler_freq: null pointer");
  const int64_t tasks = sp->nrep * g.ntiles;
  TXM_REQUIRE(cdiv(tasks, 4) < ((int64_t)1 << 31), "sampler_freq: too many tiles");
  const uint32_t k0 = (uint32_t)sp->seed, k1 = (uint32_t)(sp->seed >> 32);
  hipLaunchKernelGGL(sampler_freq_kernel, dim3((unsigned)cdiv(tasks, 4)), dim3(256), 0,
                     (hipStream_t)stream, k0, k1, sp->nrep, sp->ndat, g, (uint32_t)sp->rep0, counts, freq);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}
