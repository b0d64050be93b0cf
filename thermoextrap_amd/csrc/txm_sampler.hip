// txm_sampler.hip -- sampler kernels: indices -> freq histogram (parity mode,
// cmomy's indices_to_freq) and the counter-based exact multinomial sampler
// (scale mode) that replaces `rng.choice(ndat, (nrep, ndat))` + histogram
// (cmomy factory_sampler as reached from data.py:1782-1789) without ever
// materialising the (nrep, ndat) index/freq tables.
//
// Stage 1: per replicate, nsamp draws of a coarse bin id into LDS bins (u32),
//          16 workgroups of 1024 threads per replicate, flushed with one global
//          atomic per non-empty bin.
// Stage 2: one wave per (replicate, coarse bin) splits its count over the 2^s
//          tiles of the bin.
// Stage 3: lives in txm_sampler.h and is executed inside the bootstrap kernel
//          (txm_resample.hip), tile by tile, straight into LDS.
#include "txm_sampler.h"

namespace txm {

__device__ int g_index_error;

__global__ void clear_index_error_kernel() { g_index_error = 0; }

__global__ __launch_bounds__(256) void indices_to_freq_kernel(const int64_t *__restrict__ idx,
                                                              int64_t nrep, int64_t nsamp,
                                                              int64_t ndat,
                                                              int64_t *__restrict__ freq) {
  const int64_t r = blockIdx.y;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nsamp;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx[r * nsamp + k];
    if (j < 0 || j >= ndat) {
      g_index_error = 1;
    } else {
      atomicAdd(reinterpret_cast<unsigned long long *>(freq + r * ndat + j), 1ULL);
    }
  }
  (void)nrep;
}

// ---- stage 1 ---------------------------------------------------------------
// grid (SM_V1 / 1024, nrep), block 1024, dynamic LDS = nb1 * 4 bytes.
__global__ __launch_bounds__(1024) void sampler_stage1_kernel(uint32_t k0, uint32_t k1key,
                                                              int64_t nsamp, SamplerGeom g,
                                                              uint32_t *__restrict__ n1) {
  extern __shared__ uint32_t bins[];
  const uint32_t r = blockIdx.y;
  const uint32_t v = blockIdx.x * 1024u + threadIdx.x;
  const uint32_t nb1 = (uint32_t)g.nb1;
  for (uint32_t b = threadIdx.x; b < nb1; b += 1024u) bins[b] = 0u;
  __syncthreads();
  // nsamp <= 16 * 2^30 (check_spec), so a lane's share of SM_V1 = 16384 lanes fits 32 bits
  uint32_t quota = (uint32_t)(nsamp / SM_V1) + ((int64_t)v < (nsamp % SM_V1) ? 1u : 0u);
  const uint32_t mask = (1u << g.k1) - 1u;
  const bool last_partial = g.last_bin_size < g.BS;
  const uint32_t bsmask = (uint32_t)(g.BS - 1);
  const uint32_t last_size = (uint32_t)g.last_bin_size;
  uint32_t j = 0, m = 0;
  while (quota) {
    const Philox4 o = philox4x32_10<true>(j++, v, r, 1u, k0, k1key);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t c = slot16(o, e) & mask;
      bool ok = quota && (c < nb1);
      if (ok && last_partial && c == nb1 - 1u) {
        const Philox4 o2 = philox4x32_10<true>(m++, v, r, 4u, k0, k1key);
        ok = (o2.w[0] & bsmask) < last_size;
      }
      if (ok) {
        atomicAdd(&bins[c], 1u);
        --quota;
      }
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb1; b += 1024u) {
    const uint32_t cnt = bins[b];
    if (cnt) atomicAdd(&n1[(size_t)r * nb1 + b], cnt);
  }
}

__global__ void sampler_fill_single_bin_kernel(uint32_t *__restrict__ counts, int64_t nrep,
                                               uint32_t nsamp) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrep) counts[r] = nsamp;
}

// ---- stage 2 ---------------------------------------------------------------
// One wave per run of S2_BINS consecutive coarse bins of one replicate; the
// wave's 64-entry LDS table is private, so no workgroup barrier is needed.
// Full bins use s-bit fields (F = 4 * floor(32/s) draws per Philox call, call c
// on lane c % 64); only the last, partial bin needs rejection.
constexpr int S2_BINS = 8;

// Fast path for s <= 3 (ndat <= 1.3e8).  Per-tile counts come from popcounts of
// bit-plane products: with B_b = bit b of every field, P(m) = popc(AND_{b in m} B_b)
// for the 2^S - 1 non-empty bit subsets m, and the number of fields equal to v is
// the Moebius sum  c(v) = sum_{m >= v} (-1)^{|m|-|v|} P(m)  (P(0) = #fields).
// 16 ALU ops per 32-bit word for S = 3 instead of one match-and-count per value.
template <int S>
__global__ __launch_bounds__(256) void sampler_stage2_popc_kernel(
    uint32_t k0, uint32_t k1key, int64_t nrep, SamplerGeom g, const uint32_t *__restrict__ n1,
    uint32_t *__restrict__ counts) {
  constexpr int NS = 1 << S;
  constexpr int FPW = 32 / S;
  constexpr uint32_t F = 4u * FPW;
  uint32_t lsb = 0;  // LSB of every field of a word
#pragma unroll
  for (int k = 0; k < FPW; ++k) lsb |= 1u << (k * S);
  const int lane = threadIdx.x & 63;
  const int64_t runs_per_rep = (g.nb1 + S2_BINS - 1) / S2_BINS;
  const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= nrep * runs_per_rep) return;
  const uint32_t r = (uint32_t)(task / runs_per_rep);
  const int64_t b_begin = (task % runs_per_rep) * S2_BINS;
  const int64_t b_end = (b_begin + S2_BINS < g.nb1) ? b_begin + S2_BINS : g.nb1;
  // all counts of the run are loaded up front (one latency, not one per bin)
  uint32_t nb[S2_BINS];
#pragma unroll
  for (int i = 0; i < S2_BINS; ++i)
    nb[i] = (b_begin + i < b_end) ? n1[(size_t)r * g.nb1 + b_begin + i] : 0u;
#pragma unroll
  for (int i = 0; i < S2_BINS; ++i) {
    const int64_t bb = b_begin + i;
    if (bb >= b_end) break;
    const uint32_t b = (uint32_t)bb;
    const uint32_t n = nb[i];
    uint32_t pc[NS];   // pc[m] = P(m); pc[0] = number of fields seen by this lane
    uint32_t cnt[NS];  // exact per-value counts of the (rare) rejection path
#pragma unroll
    for (int v = 0; v < NS; ++v) pc[v] = cnt[v] = 0u;
    const bool full_bin = !(bb == g.nb1 - 1 && g.last_bin_size < g.BS);
    if (full_bin) {
      for (uint32_t c0 = 0; (uint64_t)c0 * F < n; c0 += 64u) {
        const uint32_t c = c0 + (uint32_t)lane;
        const uint64_t first = (uint64_t)c * F;
        if (first < n) {
          const Philox4 o = philox4x32_10<true>(c, b, r, 2u, k0, k1key);
          const uint32_t nd = (n - first < F) ? (uint32_t)(n - first) : F;
#pragma unroll
          for (int wi = 0; wi < 4; ++wi) {
            const int nv = (int)nd - wi * FPW;  // fields of this word that are real draws
            uint32_t valid = lsb;
            if (nv <= 0) valid = 0u;
            else if (nv < FPW) valid = lsb & ((1u << (nv * S)) - 1u);
            const uint32_t word = o.w[wi];
            uint32_t B[S];
#pragma unroll
            for (int bit = 0; bit < S; ++bit) B[bit] = (word >> bit) & valid;
            pc[0] += __popc(valid);
#pragma unroll
            for (int m = 1; m < NS; ++m) {
              uint32_t t = 0xffffffffu;
#pragma unroll
              for (int bit = 0; bit < S; ++bit)
                if (m & (1 << bit)) t &= B[bit];
              pc[m] += __popc(t);
            }
          }
        }
      }
    } else {
      const uint32_t size_b = (uint32_t)g.last_bin_size;
      const uint32_t bsmask = (uint32_t)(g.BS - 1);
      uint32_t quota = n / 64u + ((uint32_t)lane < (n % 64u) ? 1u : 0u);
      uint32_t j = 0;
      const uint32_t c1 = b * 64u + (uint32_t)lane;
      while (quota) {
        const Philox4 o = philox4x32_10<true>(j++, c1, r, 5u, k0, k1key);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t off = slot16(o, e) & bsmask;
          if (quota && off < size_b) {
            const uint32_t sub = off >> SM_LT;
#pragma unroll
            for (int v = 0; v < NS; ++v) cnt[v] += (sub == (uint32_t)v) ? 1u : 0u;
            --quota;
          }
        }
      }
    }
    // wave reduction of the subset popcounts, Moebius inversion, lane v stores tile v
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      uint32_t x = pc[m];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
      pc[m] = x;
      uint32_t y = cnt[m];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) y += __shfl_xor(y, off);
      cnt[m] = y;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      int32_t c = 0;
#pragma unroll
      for (int m = 0; m < NS; ++m) {
        if ((m & v) == v) {  // m is a superset of v
          const int extra = __builtin_popcount(m ^ v);
          c += (extra & 1) ? -(int32_t)pc[m] : (int32_t)pc[m];
        }
      }
      const uint32_t tot = (uint32_t)c + cnt[v];
      if (lane == v) mine = tot;
    }
    if (lane < NS) {
      const int64_t t = (bb << S) + lane;
      if (t < g.ntiles) counts[(size_t)r * g.ntiles + t] = mine;
    }
  }
}

constexpr int S2_WAVES = 2;  // waves per workgroup

// Every lane keeps PRIVATE counters in LDS (row pitch 2^s + 1 words, so lanes
// sit on different banks): a shared 2^s-entry table would make all 64 lanes of
// a ds_add hit the same few addresses and serialise.
__global__ __launch_bounds__(64 * S2_WAVES) void sampler_stage2_kernel(
    uint32_t k0, uint32_t k1key, int64_t nrep, SamplerGeom g, const uint32_t *__restrict__ n1,
    uint32_t *__restrict__ counts) {
  extern __shared__ uint32_t sub_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = g.s;
  const int nsub = 1 << s, pitch = nsub + 1;
  uint32_t *sub = sub_all + (size_t)wave * 64 * pitch;
  uint32_t *mine = sub + lane * pitch;
  const int64_t runs_per_rep = (g.nb1 + S2_BINS - 1) / S2_BINS;
  const int64_t task = (int64_t)blockIdx.x * S2_WAVES + wave;
  if (task >= nrep * runs_per_rep) return;  // whole wave exits together
  const uint32_t r = (uint32_t)(task / runs_per_rep);
  const int64_t b_begin = (task % runs_per_rep) * S2_BINS;
  const int64_t b_end = (b_begin + S2_BINS < g.nb1) ? b_begin + S2_BINS : g.nb1;
  const int fpw = 32 / s;
  const uint32_t F = 4u * (uint32_t)fpw;
  const uint32_t fmask = (1u << s) - 1u;
  for (int64_t bb = b_begin; bb < b_end; ++bb) {
    const uint32_t b = (uint32_t)bb;
    for (int v = 0; v < nsub; ++v) mine[v] = 0u;
    const uint32_t n = n1[(size_t)r * g.nb1 + b];
    const bool full_bin = !(bb == g.nb1 - 1 && g.last_bin_size < g.BS);
    if (full_bin) {
      for (uint32_t c0 = 0; (uint64_t)c0 * F < n; c0 += 64u) {
        const uint32_t c = c0 + (uint32_t)lane;
        const uint64_t first = (uint64_t)c * F;
        if (first < n) {
          const Philox4 o = philox4x32_10<true>(c, b, r, 2u, k0, k1key);
          const uint32_t nd = (n - first < F) ? (uint32_t)(n - first) : F;
          uint32_t q = 0;
#pragma unroll
          for (int wi = 0; wi < 4; ++wi) {  // static word index: keeps `o` in registers
            uint32_t word = o.w[wi];
            for (int k = 0; k < fpw; ++k, ++q) {
              if (q < nd) mine[word & fmask] += 1u;
              word >>= s;
            }
          }
        }
      }
    } else {
      const uint32_t size_b = (uint32_t)g.last_bin_size;
      const uint32_t bsmask = (uint32_t)(g.BS - 1);
      uint32_t quota = n / 64u + ((uint32_t)lane < (n % 64u) ? 1u : 0u);
      uint32_t j = 0;
      const uint32_t c1 = b * 64u + (uint32_t)lane;
      while (quota) {
        const Philox4 o = philox4x32_10<true>(j++, c1, r, 5u, k0, k1key);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t off = slot16(o, e) & bsmask;
          if (quota && off < size_b) {
            mine[off >> SM_LT] += 1u;
            --quota;
          }
        }
      }
    }
    // wave-private region: DS ops of one wave execute in order; fence the compiler
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < nsub) {
      uint32_t tot = 0;
      for (int l = 0; l < 64; ++l) tot += sub[l * pitch + lane];
      const int64_t t = (bb << s) + lane;
      if (t < g.ntiles) counts[(size_t)r * g.ntiles + t] = tot;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- materialise freq (testing / small sizes) -------------------------------
// one wave per (r, t); block 256 = 4 waves, each with a private 1024-bin tile.
__global__ __launch_bounds__(256) void sampler_freq_kernel(uint32_t k0, uint32_t k1key,
                                                           int64_t nrep, int64_t ndat,
                                                           SamplerGeom g,
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
    sampler_fine_tile(k0, k1key, r, t, n, tsize, lane, [&](uint32_t off) { atomicAdd(&tl[off], 1u); });
  }
  __syncthreads();
  if (active) {
    for (uint32_t i = lane; i < tsize; i += 64)
      freq[(size_t)r * ndat + (size_t)t * SM_T + i] = (int64_t)tile[wave][i];
  }
}

}  // namespace txm

using namespace txm;

extern "C" int txm_indices_to_freq(const int64_t *indices, int64_t nrep, int64_t nsamp,
                                   int64_t ndat, int64_t *freq, txm_stream stream) {
  TXM_REQUIRE(indices && freq, "indices_to_freq: null pointer");
  TXM_REQUIRE(nrep >= 1 && nsamp >= 1 && ndat >= 1 && nrep <= 65535, "indices_to_freq: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  TXM_HIP(hipMemsetAsync(freq, 0, sizeof(int64_t) * (size_t)nrep * ndat, st));
  hipLaunchKernelGGL(clear_index_error_kernel, dim3(1), dim3(1), 0, st);
  TXM_LAUNCH_CHECK();
  int gx = (int)cdiv(nsamp, 256 * 4);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(indices_to_freq_kernel, dim3(gx, (unsigned)nrep), dim3(256), 0, st, indices,
                     nrep, nsamp, ndat, freq);
  TXM_LAUNCH_CHECK();
  int flag = 0;
  TXM_HIP(hipMemcpyFromSymbolAsync(&flag, HIP_SYMBOL(g_index_error), sizeof(int), 0,
                                   hipMemcpyDeviceToHost, st));
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
  TXM_REQUIRE(sp->nrep >= 1 && sp->nrep <= 65535, "sampler: nrep=%lld outside [1, 65535]",
              (long long)sp->nrep);
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
  return (size_t)sp->nrep * (size_t)g.nb1 * sizeof(uint32_t) + 256;
}

extern "C" int txm_sampler_tile_counts(const txm_sampler_spec *sp, uint32_t *counts, void *ws,
                                       size_t ws_bytes, txm_stream stream) {
  SamplerGeom g;
  int64_t nsamp;
  int rc = check_spec(sp, &g, &nsamp);
  if (rc != TXM_OK) return rc;
  TXM_REQUIRE(counts, "sampler: null counts");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t k0 = (uint32_t)sp->seed, k1 = (uint32_t)(sp->seed >> 32);
  if (g.nb1 == 1) {
    hipLaunchKernelGGL(sampler_fill_single_bin_kernel, dim3((unsigned)cdiv(sp->nrep, 256)),
                       dim3(256), 0, st, counts, sp->nrep, (uint32_t)nsamp);
    TXM_LAUNCH_CHECK();
    return TXM_OK;
  }
  uint32_t *n1 = counts;
  if (g.s > 0) {
    if (!ws || ws_bytes < txm_sampler_counts_ws_bytes(sp)) {
      set_error("sampler: workspace too small");
      return TXM_ERR_WORKSPACE;
    }
    n1 = (uint32_t *)ws;
  }
  TXM_HIP(hipMemsetAsync(n1, 0, sizeof(uint32_t) * (size_t)sp->nrep * g.nb1, st));
  hipLaunchKernelGGL(sampler_stage1_kernel, dim3(SM_V1 / 1024, (unsigned)sp->nrep), dim3(1024),
                     (size_t)g.nb1 * sizeof(uint32_t), st, k0, k1, nsamp, g, n1);
  TXM_LAUNCH_CHECK();
  if (g.s > 0) {
    const int64_t tasks = sp->nrep * cdiv(g.nb1, S2_BINS);
    if (g.s <= 3) {
      dim3 grid((unsigned)cdiv(tasks, 4)), block(256);
      if (g.s == 1)
        hipLaunchKernelGGL((sampler_stage2_popc_kernel<1>), grid, block, 0, st, k0, k1, sp->nrep, g, n1, counts);
      else if (g.s == 2)
        hipLaunchKernelGGL((sampler_stage2_popc_kernel<2>), grid, block, 0, st, k0, k1, sp->nrep, g, n1, counts);
      else
        hipLaunchKernelGGL((sampler_stage2_popc_kernel<3>), grid, block, 0, st, k0, k1, sp->nrep, g, n1, counts);
    } else {
      const size_t lds = (size_t)S2_WAVES * 64 * ((1 << g.s) + 1) * sizeof(uint32_t);
      hipLaunchKernelGGL(sampler_stage2_kernel, dim3((unsigned)cdiv(tasks, S2_WAVES)),
                         dim3(64 * S2_WAVES), lds, st, k0, k1, sp->nrep, g, n1, counts);
    }
    TXM_LAUNCH_CHECK();
  }
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
                     (hipStream_t)stream, k0, k1, sp->nrep, sp->ndat, g, counts, freq);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}
