// Pieces shared by the kernels that take known answers back from the rank counters of the x3 mode (kernels_tail_bf16.hip:
// the fused tail kernel; kernels_score3_bf16.hip: k_filter_excess_bf16x3):
// one 32 x 32 tile of (entity rows gathered from the row-major twins) x (the resident query fragments of a 32-query block),
// in the k-step order of every x3 kernel (bf16x3_chain.h), and the CSR walk around it.
#pragma once
#include <hip/hip_runtime.h>

#include "bf16x3_chain.h"
#include "conv_fold.h"

namespace coper {

// one 32 x 32 tile: A rows = entity rows erow[i] (gathered from the row-major twins; erow < 0: a zero row is not needed,
// its result is discarded), B = the resident query fragments; accumulators start from pred_bias of the row
template <int KS>
__device__ __forceinline__ f32x16 tail_tile(const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo, const float* __restrict__ bias_pad,
                                            const int64_t* s_e, const int64_t my_erow, const uint4 (&bh)[KS], const uint4 (&bl)[KS],
                                            const int half, const int sexp) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t er = s_e[(r & 3) + 8 * (r >> 2) + 4 * half];
    acc[r] = er >= 0 ? x3_scale(bias_pad[er], sexp) : 0.f;     // accumulators carry 2^(e_E + e_h) (split16.h)
  }
  const int64_t ea = my_erow >= 0 ? my_erow : 0;
  const uint4* pa_h = Ehi + ea * (2 * KS) + half;
  const uint4* pa_l = Elo + ea * (2 * KS) + half;
  constexpr int PB = KS;  // one batch: the compiler keeps as many of the tile's gathered 16-byte loads in flight as fit beside
                          // the resident fragments (162 registers: three workgroups per CU.  Forcing all 2 KS loads into
                          // registers first -- 252 registers, two workgroups per CU -- measured 52 us against 45)
#pragma unroll
  for (int k0 = 0; k0 < KS; k0 += PB) {
    uint4 ah[PB + 1], al[PB + 1];   // (+1: the pair loop names element u + 1 in a branch that is never taken for the last odd step)
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int k = k0 + u < KS ? k0 + u : KS - 1;
      ah[u] = pa_h[k * 2];
      al[u] = pa_l[k * 2];
    }
#pragma unroll
    for (int u = 0; u < PB; u += 2) {
      if (k0 + u + 1 < KS) { BX3_PAIR(ah[u], al[u], bh[k0 + u], bl[k0 + u], ah[u + 1], al[u + 1], bh[k0 + u + 1], bl[k0 + u + 1], acc); }
      else if (k0 + u < KS) { BX3_LAST(ah[u], al[u], bh[k0 + u], bl[k0 + u], acc); }
    }
  }
  return acc;
}

// one tile of CSR entries [pb, pb + 32) of a 32-query block whose fragments the wave holds: the score of entry i against its
// own query (sc), that query's index in the block (qi_out), and the entry's row -- or -1 for what the dense mask of
// metrics.py:40-46 does not change (the target, an adjacent duplicate, a row of another shard, a lane past p_end).
// my_lo / my_e2: lane i (both halves) holds the first entry and the target of query i.  s_e: 32 slots of the wave.
template <int KS>
__device__ __forceinline__ int64_t tail_filter_tile(const int64_t pb, const int64_t p_end, const int64_t my_lo, const int64_t my_e2,
                                                    const int64_t* __restrict__ idx, const int64_t n_local, int64_t* s_e,
                                                    const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo,
                                                    const float* __restrict__ bias_pad, const uint4 (&bh)[KS], const uint4 (&bl)[KS],
                                                    const int i, const int half, const int sexp, float& sc, int& qi_out) {
  const int64_t p = pb + i;
  // every lane runs the same cross-lane reads (a shuffle must not sit in divergent code: inactive lanes supply nothing);
  // lanes past the last entry carry frow = -1
  const bool valid = p < p_end;
  // local query of entry p: the last j with indptr[q0 + j] <= p (binary lifting over the lanes' first entries)
  int qi = 0;
#pragma unroll
  for (int step = 16; step >= 1; step >>= 1) {
    const int cand = qi + step;
    const int64_t first = __shfl(my_lo, cand < 32 ? cand : 31);
    qi = (valid && cand < 32 && first <= p) ? cand : qi;
  }
  const int64_t qfirst = __shfl(my_lo, qi);
  const int64_t qe2 = __shfl(my_e2, qi);
  int64_t frow = -1;
  if (valid) {
    const int64_t f = idx[p];
    frow = f;
    if (p > qfirst && idx[p - 1] == f) frow = -1;          // adjacent duplicate: the dense mask is idempotent
    if (f == qe2) frow = -1;                                // the target is restored after masking (metrics.py:46)
    if (frow < 0 || frow >= n_local) frow = -1;
  }
  __builtin_amdgcn_wave_barrier();
  if (half == 0) s_e[i] = frow;
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): wave-local LDS exchange
  __builtin_amdgcn_wave_barrier();
  const f32x16 acc = tail_tile<KS>(Ehi, Elo, bias_pad, s_e, frow, bh, bl, half, sexp);
  // entry i wants D[i][qi]: register (i & 3) + 4 * (i >> 3) of lane qi + 32 * ((i >> 2) & 1)
  const int src = qi + 32 * ((i >> 2) & 1);
  const int reg = (i & 3) + 4 * (i >> 3);
  sc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = __shfl(acc[r], src);
    sc = (r == reg) ? v : sc;
  }
  qi_out = qi;
  return frow;
}

#ifndef COPER_TL_WAVES
#define COPER_TL_WAVES 4
#endif
#ifndef COPER_TL_OWN_TILES
#define COPER_TL_OWN_TILES (3 * COPER_TL_WAVES - 1)
#endif
// CSR entries of its 32 queries a workgroup of the tail kernel takes back itself: round 0 (the waves beside the target wave)
// and two more rounds of all waves -- 352 entries; the synthetic filters of the BASELINE configs hold 160 +- 25 per block, at
// most 237.  A block with 1,024 kept its workgroup for eight rounds: +48 us on the launch, the chip idle behind it.
constexpr int64_t TL_OWN_ENTRIES = 32 * (int64_t)(COPER_TL_OWN_TILES);
constexpr int FX_WAVES = 4;                                // waves per workgroup of the excess kernel
constexpr int FX_GRID = 512;                               // its workgroups: 2,048 waves, two per SIMD
constexpr int TL_WAVES = COPER_TL_WAVES;   // waves per 32-query block: they share the finalize and deal the filter tiles among themselves


// The query fragments of 32-query block q0 .. q0 + 31 rebuilt from the fp32 rows the finalize wrote: lane (i, half) holds
// piece (ks, half) of query q0 + i -- same values, same power of two (eh = e_h of the batch), same split, so the bits the
// finalize held.
template <int KS>
__device__ __forceinline__ void tail_fragments_from_rows(const float* __restrict__ hvec, const int64_t q, const bool live, const int d,
                                                         const int half, const int eh, uint4 (&bh)[KS], uint4 (&bl)[KS]) {
  const bool vec_ok = (d & 3) == 0 && (((uintptr_t)hvec) & 15) == 0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k0 = 16 * ks + 8 * half;
    float y[8];
    if (live && k0 + 8 <= d && vec_ok) {
      const float4 a = *(const float4*)(hvec + q * d + k0), b = *(const float4*)(hvec + q * d + k0 + 4);
      y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w; y[4] = b.x; y[5] = b.y; y[6] = b.z; y[7] = b.w;
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) y[c] = (live && k0 + c < d) ? hvec[q * d + k0 + c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) y[c] = x3_scale(y[c], eh);
    split8_bf16(y, bh[ks], bl[ks]);
  }
}

// One tile's known answers above the band taken back from `ranks`: entries of a tile are in CSR order (runs of equal qi), the
// first lane of a run subtracts the run's hits at once.  Every lane runs the cross-lane steps.
__device__ __forceinline__ void tail_take_back(const bool hit, const int qi, const int i, const int half, const int64_t q0,
                                               int32_t* __restrict__ ranks) {
  const int q_prev = __shfl_up(qi, 1);
  const bool head = half == 0 && (i == 0 || q_prev != qi);
  const unsigned heads = (unsigned)(__ballot(head) & 0xFFFFFFFFull);
  const unsigned m_hit = (unsigned)(__ballot(hit) & 0xFFFFFFFFull);
  if (head) {
    const unsigned later = i < 31 ? (heads >> (i + 1)) : 0u;
    const int end = later ? i + 1 + __builtin_ctz(later) : 32;
    const unsigned run = (end >= 32 ? 0xFFFFFFFFu : ((1u << end) - 1u)) & ~((1u << i) - 1u);
    const int c = __builtin_popcount(m_hit & run);
    if (c) atomicSub(&ranks[q0 + qi], c);
  }
}

struct FilterArgs {          // the filter role's view of a pass (queries of one count launch: all pointers at its first query)
  const float* hvec; const uint4* Ehi; const uint4* Elo; const float* bias_pad; const int64_t* e2; const int64_t* indptr;
  const int64_t* idx; const float2* tband; int32_t* ranks; int32_t* heavy; const int32_t* x3s; int64_t B, n_local; int d;
};

}  // namespace coper
