// txm_count_table.hip -- stage 3 of the device sampler as a kernel of its own: the per-sample u8 draw counts of a slab of
// replicates, written to HBM in the order the int8 contraction kernel (txm_resample_i8g.hip) reads its MFMA A operands.
//
// The stream is the one every bootstrap kernel draws (txm_sampler.h; normative statement oracle/philox_oracle.c): draw d of
// (replicate r, tile t) is field d % 12 of Philox call d / 12 with counter (t, d / 12, r, 3) -- the same calls, the same
// integers as the fill phase inside resample_i8t_kernel, which this kernel takes out of the contraction (reference op:
// cmomy.factory_sampler + indices_to_freq as reached from src/thermoextrap/data.py:1782-1789; the table replaces the
// (nrep, ndat) int64 freq table cmomy materialises, at one byte per entry and only for one slab of replicates at a time).
//
// Workgroup = 8 waves x 64 replicates (lane = replicate, the waves split the Philox calls of a tile: every ds_add of a wave
// hits 64 consecutive words, no bank conflict) x a run of tiles.  Per tile: zero the [256 words][64 replicates] count tile
// in LDS, fill it, write it out transposed -- a lane's 16 bytes are the counts of 16 consecutive samples of one replicate,
// a wave's store 1 KiB contiguous.  64 KiB of LDS: two workgroups per CU, four waves per SIMD.
#include "txm_i8g.h"

namespace txm {

__global__ __launch_bounds__(T_BLOCK) void count_table_kernel(const uint32_t *__restrict__ counts, const int64_t nrep,
                                                              const int64_t ntiles, const int64_t N,
                                                              const uint32_t last_tile_size, const uint32_t k0,
                                                              const uint32_t k1, const uint32_t rep_base,
                                                              const int64_t rep_begin, const int tiles_per_block,
                                                              unsigned char *__restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint32_t *cntw = reinterpret_cast<uint32_t *>(lds);  // [256 words][64 replicates]
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int h64 = blockIdx.y;                          // 64-replicate half of replicate group h64 >> 1
  const int64_t rep0 = rep_begin + (int64_t)h64 * I8_REPS;  // first replicate (of the call) of this workgroup
  const int64_t my_rep = rep0 + lane;
  const bool rep_live = my_rep < nrep;
  const uint32_t rstream = rep_base + (uint32_t)my_rep;
  const uint32_t lane4 = (uint32_t)lane * 4u;
  const bool any_live = rep0 < nrep;  // uniform
  int64_t t_begin = (int64_t)blockIdx.x * tiles_per_block, t_end = t_begin + tiles_per_block;
  if (t_end > ntiles) t_end = ntiles;
  unsigned char *out_g = table + ((size_t)(h64 >> 1) * (size_t)ntiles) * G_TILE_BYTES + (size_t)(2 * (h64 & 1)) * 1024;

#pragma unroll 1
  for (int64_t t = t_begin; t < t_end; ++t) {
    const uint32_t tsize = (t == ntiles - 1) ? last_tile_size : (uint32_t)SM_T;
    const int64_t i_tile = t * SM_T;
    const int64_t wbase = i_tile > N - SM_T ? N - SM_T : i_tile;  // the last tile slides its window back
    const uint32_t shift = (uint32_t)(i_tile - wbase);
    uint32_t n = (rep_live && any_live) ? counts[(size_t)my_rep * ntiles + t] : 0u;
    __syncthreads();  // the previous tile's write-out has read the count tile
    for (int e = threadIdx.x; e < T_CNT_BYTES / 16; e += T_BLOCK) reinterpret_cast<uint4 *>(cntw)[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
#ifdef TXM_CT_NO_FILL  // (timing build: zeroed tiles written out -- what the write-out path alone costs)
    if (false) {
#else
    if (any_live) {
#endif
      // live replicates of this half group; <= 32 -- the call's last one: 8 of 64 at nrep = 200 -- and FP = fill_pack lanes share
      // a replicate and take FP consecutive Philox calls of it: 1 / FP of the wave instructions of the lane-per-replicate fill for
      // the same draws (which lane runs a call of the stream is free: the count words are atomics).  The fill phase of
      // resample_i8t_kernel has had this since round 4; BASELINE config 2 (200 replicates) lost it when its call moved to the
      // table-fed kernel (round 6: generator 0.90 ms for 256 replicates' worth of Philox calls)
      const int64_t live_reps = nrep - rep0;
      const int fill_pack = live_reps <= 8 ? 8 : live_reps <= 16 ? 4 : live_reps <= 32 ? 2 : 1;  // uniform
      if (tsize == (uint32_t)SM_T && fill_pack > 1) {
        const int rp = lane & (I8_REPS / fill_pack - 1), slot = lane / (I8_REPS / fill_pack);
        const uint32_t np = rep0 + rp < nrep ? counts[(size_t)(rep0 + rp) * ntiles + t] : 0u;
        uint32_t nmx = np;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t hi = (uint32_t)__shfl_xor((int)nmx, o);
          nmx = hi > nmx ? hi : nmx;
        }
        nmx = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmx);
        const uint32_t rsp = rep_base + (uint32_t)(rep0 + rp);
#pragma unroll 1
        for (uint32_t c0 = (uint32_t)wave * (uint32_t)fill_pack; c0 * 12u < nmx; c0 += (uint32_t)(T_WAVES * fill_pack))
          t_fill_call<false>(cntw, k0, k1, rsp, (uint32_t)t, c0 + (uint32_t)slot, np, (uint32_t)rp * 4u);
      } else if (tsize == (uint32_t)SM_T) {
        // dead lanes (replicates past nrep) draw like the smallest live lane; their rows are written as zeros below
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
        for (; c + T_WAVES < call_all; c += 2 * T_WAVES) {  // two Philox chains in flight
          t_fill_call<true>(cntw, k0, k1, rstream, (uint32_t)t, c, n, lane4);
          t_fill_call<true>(cntw, k0, k1, rstream, (uint32_t)t, c + T_WAVES, n, lane4);
        }
#pragma unroll 1
        for (; c < call_all; c += T_WAVES) t_fill_call<true>(cntw, k0, k1, rstream, (uint32_t)t, c, n, lane4);
#pragma unroll 1
        for (; c * 12u < nmax; c += T_WAVES) t_fill_call<false>(cntw, k0, k1, rstream, (uint32_t)t, c, n, lane4);
      } else {
        // the partial last tile: the stream is defined over 64 virtual lanes per replicate (txm_sampler.h)
        for (int rr = wave * (I8_REPS / T_WAVES); rr < (wave + 1) * (I8_REPS / T_WAVES); ++rr) {
          const int64_t r = rep0 + rr;
          if (r >= nrep) break;  // wave-uniform
          const uint32_t nr = counts[(size_t)r * ntiles + t];
          sampler_fine_tile(k0, k1, rep_base + (uint32_t)r, (uint32_t)t, nr, tsize, lane, [&](uint32_t off0) {
            const uint32_t off = off0 + shift;
            atomicAdd(&cntw[(off >> 2) * I8_REPS + (uint32_t)rr], 1u << ((off & 3u) << 3));
          });
        }
      }
    }
    __syncthreads();
    // write-out: element e = (k-step s, local quarter ql, lane L): words 8 s + 4 (L >> 5) + 0..3 of replicate 32 ql + (L & 31)
    unsigned char *out_t = out_g + (size_t)t * G_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = (int)threadIdx.x + i * T_BLOCK;
      const int s = e >> 7, ql = (e >> 6) & 1, L = e & 63;
      const int rl = 32 * ql + (L & 31);
      const uint32_t *src = cntw + (8 * s + 4 * (L >> 5)) * I8_REPS + rl;
      uint4 v = make_uint4(src[0], src[I8_REPS], src[2 * I8_REPS], src[3 * I8_REPS]);
      if (rep0 + rl >= nrep) v = make_uint4(0, 0, 0, 0);
#ifdef TXM_CT_NO_STORE  // (timing build: the fill alone)
      if (v.x == 0xdeadbeefu) *reinterpret_cast<uint4 *>(out_t + (size_t)L * 16) = v;
#elif defined(TXM_CT_PLAIN_STORE)
      *reinterpret_cast<uint4 *>(out_t + (size_t)s * G_KSTEP_BYTES + (size_t)ql * 1024 + (size_t)L * 16) = v;
#else
      typedef uint32_t ct_v4u __attribute__((ext_vector_type(4)));
      const ct_v4u vv = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(vv, reinterpret_cast<ct_v4u *>(out_t + (size_t)s * G_KSTEP_BYTES + (size_t)ql * 1024 + (size_t)L * 16));
#endif
    }
  }
}

int launch_count_table(const uint32_t *counts, int64_t nrep, int64_t N, uint32_t k0, uint32_t k1, uint32_t rep_base,
                       int64_t rep_begin, int64_t n_groups, unsigned char *table, hipStream_t st) {
  const int64_t ntiles = cdiv(N, SM_T);
  if (N < SM_T || n_groups < 1 || rep_begin < 0) {
    set_error("count_table: needs N >= %d samples, n_groups >= 1", SM_T);
    return TXM_ERR_INVALID;
  }
  // runs of tiles per workgroup: enough workgroups to fill the chip several times over, long enough runs to stream
  // the tile counts (32 per cache line)
#ifndef TXM_CT_ROUNDS  // (A/B builds)
#define TXM_CT_ROUNDS 4
#endif
  int tpb = 32;
  while (tpb > 1 && cdiv(ntiles, tpb) * 2 * n_groups < TXM_CT_ROUNDS * (int64_t)num_cus()) tpb /= 2;
  const dim3 grid((unsigned)cdiv(ntiles, tpb), (unsigned)(2 * n_groups));
  TXM_SET_MAX_LDS(count_table_kernel, T_CNT_BYTES);
  hipLaunchKernelGGL(count_table_kernel, grid, dim3(T_BLOCK), T_CNT_BYTES, st, counts, nrep, ntiles, N,
                     (uint32_t)(N - (ntiles - 1) * SM_T), k0, k1, rep_base, rep_begin, tpb, table);
  TXM_LAUNCH_CHECK();
  return TXM_OK;
}

}  // namespace txm

using namespace txm;

extern "C" size_t txm_sampler_count_table_bytes(int64_t ndat, int64_t nreps) {
  if (ndat < SM_T || nreps < 1) return 0;
  return count_table_bytes(cdiv(ndat, SM_T), nreps);
}

extern "C" int txm_sampler_count_table(const txm_sampler_spec *sp, const uint32_t *counts, int64_t rep_begin,
                                       int64_t nreps, uint8_t *table, txm_stream stream) {
  TXM_REQUIRE(sp && counts && table, "sampler_count_table: null pointer");
  TXM_REQUIRE(sp->nrep >= 1 && sp->ndat >= SM_T && sp->ndat <= ((int64_t)1 << 30), "sampler_count_table: needs %d <= ndat <= 2^30", SM_T);
  TXM_REQUIRE(sp->rep0 >= 0 && sp->rep0 + sp->nrep <= ((int64_t)1 << 32), "sampler_count_table: stream replicates out of range");
  TXM_REQUIRE(rep_begin >= 0 && nreps >= 1 && rep_begin % G_REPS == 0 && rep_begin < sp->nrep,
              "sampler_count_table: the slab starts at a multiple of %d replicates inside the table", G_REPS);
  return launch_count_table(counts, sp->nrep, sp->ndat, (uint32_t)sp->seed, (uint32_t)(sp->seed >> 32), (uint32_t)sp->rep0,
                            rep_begin, cdiv(nreps, G_REPS), table, (hipStream_t)stream);
}
