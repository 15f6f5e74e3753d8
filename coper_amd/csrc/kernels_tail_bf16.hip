// k_finalize_targets_filter_bf16x3 -- everything of a ranking pass between the encoder and the count kernel, and the sparse
// filter correction that used to follow it, in ONE launch (coper_encode_rank, bf16x3, ranks only).
//
// Before (round 2): k_dense_finalize_pack_q (h -> 16-bit planes, 22 us) -> k_pair_bf16x3 mode 0 (targets, 15 us) -> count ->
// k_pair_bf16x3 mode 2 (filter correction, 22 us): three small launches and their gaps were 12 % of an FB15k-237-shaped
// pass.  Now one workgroup (4 waves) owns a block of 32 queries from their fp32 h rows to their final rank offsets:
//   1. fragments: the fp32 rows (written by the fused encoder's epilogue, or by k_finalize_h_publish below when the dense
//      layer ran in several K slices) are multiplied by 2^e_h -- the batch's exponent, reduced here from the per-block maxima
//      the producer left in x3m (bf16x3_chain.h) -- and split into hi / lo fp16.  Lane l handles piece (k-step ks, half l >> 5)
//      of query l & 31, which IS lane l's part of the block's B-operand fragment of k-step ks: the fragments go to LDS (the
//      tiles read them from there) and, once, to the f3 image the count kernel reads.
//   2. targets: logit(q_i, e2[q_i]) = the diagonal of the 32 x 32 tile whose A rows are the gathered entity rows of e2 and
//      whose B operand are those fragments -- the MFMA sequence of every other bf16x3 kernel, so the same bits.  tau_q and the
//      band tband[q] = {t - tau, t + tau} (x3_band_tau), in the accumulators' units.
//   3. filter correction: the CSR entries of these 32 queries are contiguous; 32 entries at a time their entity rows are
//      the A rows, the tile against the SAME B fragments holds logit(f_i, q_j) for every local query j, and entry i reads
//      column j = its own query (one cross-lane read per accumulator register).  What the count kernel will count for a
//      known answer (logit ABOVE the band) is subtracted in advance: ranks[q] = 1 - #(such entries); the count kernel then
//      adds to it (the reference: pred[e2_multi == 1] = -inf; pred[e2] = target, metrics.py:45-46).
// Round 4: every wave requests the rows of its first tile (targets or CSR entries) BEFORE step 1 -- they do not depend on the
// fragments -- and the ids of its next tile a tile ahead (tl_prep / tl_gather / tl_mma below).
#include "bf16x3_chain.h"
#include "coper_internal.h"
#include "conv_fold.h"
#include "tail_tile.h"
#include <algorithm>
#include <vector>

namespace coper {

#ifdef COPER_DBG_TL_CLOCK
// diagnostic build (tools/ab_build.py): s_memrealtime (100 MHz) at the phase boundaries of wave 0 of every workgroup;
// g_tl_clk[blk] = {kernel-relative start, after finalize, after the fragment exchange, after the targets, after the filter
// tiles, end}.  No output depends on them.
constexpr int TL_NSTAMP = 6;
__device__ unsigned long long g_tl_clk[TL_NSTAMP * 1024];
__device__ unsigned long long g_tl_t0;
#define TL_STAMP(i_)                                                                                         \
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_tl_clk[TL_NSTAMP * blockIdx.x + (i_)] = __builtin_amdgcn_s_memrealtime();
#else
#define TL_STAMP(i_)
#endif

// h rows of a pass; the arithmetic of k_dense_finalize (kernels_encode.hip), bit for bit.  A thread owns piece p = 8 features
// (32 lanes per row, np = d_pad16 / 8 of them in use: folded FCBN scale / shift of the piece stay in registers) and walks the
// rows 8 apart; per row two 16-byte loads of partial sums per K slice, two of the dense bias, two 16-byte stores.  (The first
// form -- a thread per piece, every operand a scalar load, 30 loads per piece -- took 43 us for 33 MB.)  Every block folds its
// largest value into its slot of x3m (bf16x3_chain.h: the tail kernel reduces the slots to e_h of the batch).
__global__ __launch_bounds__(256) void k_finalize_h_publish(const float* __restrict__ z_part, int ksplit, int64_t Bcap, int64_t B, int d,
                                                            int d_pad16, const int32_t* __restrict__ inv_perm,
                                                            const int32_t* __restrict__ sorted_rid, const float* __restrict__ fc_b,
                                                            int per_rel_bias, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const int32_t* __restrict__ w_exp, int x_exp,
                                                            float* __restrict__ h_out, float* __restrict__ x3m) {
  const int np = d_pad16 >> 3;
  const int p = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int k0 = 8 * p;
  float m = 0.f;
  for (int p0 = 0; p0 < np; p0 += 32) {       // (d <= 256: one round)
    const int kp = k0 + 8 * p0;
    const bool mine = p + p0 < np && kp < d;
    const bool vec = mine && kp + 8 <= d && (d & 3) == 0 &&
                     ((((uintptr_t)fc_b) | ((uintptr_t)scale) | ((uintptr_t)shift) | ((uintptr_t)h_out)) & 15) == 0;
    float sc8[8], sh8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int k = kp + c < d ? kp + c : d - 1;
      sc8[c] = mine ? scale[k] : 0.f;
      sh8[c] = mine ? shift[k] : 0.f;
    }
    for (int64_t q = (int64_t)blockIdx.x * 8 + r; q < B; q += (int64_t)gridDim.x * 8) {
      if (!mine) continue;
      const int64_t pos = inv_perm[q];
      const int rid = sorted_rid[pos];
      const float* bsrc = per_rel_bias ? fc_b + (int64_t)rid * d : fc_b;
      const int zexp = -((w_exp ? w_exp[per_rel_bias ? rid : 0] : 0) + x_exp);   // the partial sums carry 2^(e_W + e_x) (split16.h)
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < ksplit; ++s) {
        const float4* pp = (const float4*)(z_part + ((int64_t)s * Bcap + pos) * d_pad16 + kp);
        const float4 a = pp[0], b = pp[1];
        z[0] += a.x; z[1] += a.y; z[2] += a.z; z[3] += a.w;
        z[4] += b.x; z[5] += b.y; z[6] += b.z; z[7] += b.w;
      }
      float bb[8];
      if (vec) {
        const float4 b0 = *(const float4*)(bsrc + kp), b1 = *(const float4*)(bsrc + kp + 4);
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) bb[c] = bsrc[kp + c < d ? kp + c : d - 1];
      }
      float y[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = x3_scale(z[c], zexp) + bb[c];
        v = fmaf(v, sc8[c], sh8[c]);
        v = fmaxf(v, 0.f);
        y[c] = v;
        if (kp + c < d) m = fmaxf(m, v);
      }
      if (vec) {
        float4* ho = (float4*)(h_out + q * d + kp);
        ho[0] = make_float4(y[0], y[1], y[2], y[3]);
        ho[1] = make_float4(y[4], y[5], y[6], y[7]);
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (kp + c < d) h_out[q * d + kp + c] = y[c];
      }
    }
  }
  x3_block_store_max(m, x3m);
}

// ---- the tile of this kernel in three pieces (tail_tile.h holds the one-piece form the excess role keeps): the rows of a tile are
// requested as soon as their ids are known -- for a wave's first tile that is BEFORE the block's fragments exist -- and the
// fragments are read from LDS step by step instead of sitting in 104 registers, which is what leaves room for a whole tile of
// gathers in flight.  Same instructions in the same order as tail_tile: the same bits.
//   tl_prep:   CSR entries [pb, pb + 32) -> the entry's row (or -1: see tail_filter_tile) and its query; rows published in s_e
__device__ __forceinline__ int64_t tl_prep(const int64_t pb, const int64_t p_end, const int64_t my_lo, const int64_t my_e2,
                                           const int64_t f_cur, const int64_t f_prev, const int64_t n_local, int64_t* s_e, const int i,
                                           const int half, int& qi_out) {
  const int64_t p = pb + i;
  const bool valid = p < p_end;
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
    frow = f_cur;
    if (p > qfirst && f_prev == f_cur) frow = -1;          // adjacent duplicate: the dense mask is idempotent
    if (f_cur == qe2) frow = -1;                            // the target is restored after masking (metrics.py:46)
    if (frow < 0 || frow >= n_local) frow = -1;
  }
  __builtin_amdgcn_wave_barrier();
  if (half == 0) s_e[i] = frow;
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): wave-local LDS exchange
  __builtin_amdgcn_wave_barrier();
  qi_out = qi;
  return frow;
}
//   tl_gather: the 2 KS 16-byte pieces of the lane's row, all in flight
template <int KS>
__device__ __forceinline__ void tl_gather(const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo, const int64_t my_erow, const int half,
                                          uint4 (&ah)[KS + 1], uint4 (&al)[KS + 1]) {
#ifdef COPER_DBG_TL_NOGATHER   /* ablation (wrong results): every lane reads row 0 */
  const int64_t ea = 0;
#else
  const int64_t ea = my_erow >= 0 ? my_erow : 0;
#endif
  const uint4* pa_h = Ehi + ea * (2 * KS) + half;
  const uint4* pa_l = Elo + ea * (2 * KS) + half;
#pragma unroll
  for (int k = 0; k < KS; ++k) { ah[k] = pa_h[k * 2]; al[k] = pa_l[k * 2]; }
  ah[KS] = ah[KS - 1]; al[KS] = al[KS - 1];     // (never used: the pair loop names element u + 1 in a branch it does not take)
}
//   tl_mma:    the tile, query fragments from LDS
template <int KS>
__device__ __forceinline__ f32x16 tl_mma(const float* __restrict__ bias_pad, const int64_t* s_e, const uint4 (&ah)[KS + 1], const uint4 (&al)[KS + 1],
                                         const uint4 (*s_bh)[64], const uint4 (*s_bl)[64], const int lane, const int half, const int sexp) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t er = s_e[(r & 3) + 8 * (r >> 2) + 4 * half];
    acc[r] = er >= 0 ? x3_scale(bias_pad[er], sexp) : 0.f;
  }
#pragma unroll
  for (int u = 0; u < KS; u += 2) {
    if (u + 1 < KS) {
      const uint4 bh0 = s_bh[u][lane], bl0 = s_bl[u][lane], bh1 = s_bh[u + 1][lane], bl1 = s_bl[u + 1][lane];
      BX3_PAIR(ah[u], al[u], bh0, bl0, ah[u + 1], al[u + 1], bh1, bl1, acc);
    } else {
      const uint4 bh0 = s_bh[u][lane], bl0 = s_bl[u][lane];
      BX3_LAST(ah[u], al[u], bh0, bl0, acc);
    }
  }
  return acc;
}
//   entry i's score: D[i][qi] = register (i & 3) + 4 (i >> 3) of lane qi + 32 ((i >> 2) & 1)
__device__ __forceinline__ float tl_entry_score(const f32x16& acc, const int qi, const int i) {
  const int src = qi + 32 * ((i >> 2) & 1);
  const int reg = (i & 3) + 4 * (i >> 3);
  float sc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = __shfl(acc[r], src);
    sc = (r == reg) ? v : sc;
  }
  return sc;
}

template <int KS>
__global__ __launch_bounds__(64 * TL_WAVES, KS <= 13 ? 3 : 2) void k_finalize_targets_filter_bf16x3(
    int64_t B, int d, const float* __restrict__ h_rows, uint4* __restrict__ hf3,
    const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo, const float* __restrict__ bias_pad, int64_t n_local,
    const int64_t* __restrict__ e2, const int64_t* __restrict__ indptr, const int64_t* __restrict__ idx, float* __restrict__ tgt,
    float kappa, const unsigned* __restrict__ band_consts, const float* __restrict__ x3m, int ent_exp, int32_t* __restrict__ x3s,
    float2* __restrict__ tband, int32_t* __restrict__ ranks, int32_t* __restrict__ heavy, const int32_t* __restrict__ stale,
    int32_t* __restrict__ stale_count) {
  __shared__ uint4 s_bh[KS][64], s_bl[KS][64];   // the block's B-operand fragments (hi / lo), shared by the waves
  __shared__ int64_t s_e[TL_WAVES][32];
  __shared__ int s_corr[32];
  __shared__ float s_t[32];       // t_hi of the block's queries: the upper edge of the exact band
  __shared__ float s_n2[TL_WAVES][32];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
  const int64_t blk = blockIdx.x, q0 = blk * 32, q = q0 + i;
  const bool live = q < B;
  if (threadIdx.x < 32) s_corr[threadIdx.x] = 0;
  TL_STAMP(0);
  // ids of the later phases, requested before the finalize so that their latency lies under it
  const int64_t my_e2 = live ? e2[q] : -1;
  const int64_t qe = q0 + 32 < B ? q0 + 32 : B;
  const int64_t p_begin = q0 < B ? indptr[q0] : 0, p_all = q0 < B ? indptr[qe] : 0;
  // the block's own share of its CSR entries; what lies beyond (a query with thousands of known answers) is listed for
  // k_filter_excess_bf16x3, which deals those entries over the whole chip
  const int64_t p_end = p_all - p_begin > TL_OWN_ENTRIES ? p_begin + TL_OWN_ENTRIES : p_all;
  if (threadIdx.x == 0 && p_all > p_end) heavy[2 + atomicAdd(&heavy[0], 1)] = (int32_t)blk;
  const int64_t my_lo = live ? indptr[q] : p_end;   // first entry of query i (lane i), for the search in filter_tile

  // ---- 0. the wave's first item, requested before anything else: item 0 = the targets (pair i = (query i, e2[query i]), the
  // diagonal of the tile), item 1 + n = the n-th tile of 32 CSR entries of these queries ([indptr[q0], indptr[min(q0 + 32, B)])
  // is contiguous); wave 0 takes the targets, waves 1.. the first tiles.  Their rows travel while the fragments are built.
  int64_t erow = my_e2;
  if (erow < 0 || erow >= n_local) erow = -1;
  int qi0 = 0;
  int64_t row0 = -1;
  bool have0 = true;   // (wave-uniform) the wave has a first item
  if (wave == 0) {   // (wave-uniform branches: the shuffles inside run with every lane)
    row0 = erow;
    if (half == 0) s_e[0][i] = erow;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  } else {
    const int64_t pb = p_begin + 32 * (int64_t)(wave - 1);
    if (pb < p_end) {
      const int64_t p = pb + i;
      const int64_t f_cur = p < p_end ? idx[p] : -1, f_prev = (p < p_end && p > p_begin) ? idx[p - 1] : -1;
      row0 = tl_prep(pb, p_end, my_lo, my_e2, f_cur, f_prev, n_local, s_e[wave], i, half, qi0);
    } else {
      have0 = false;       // (s_e[wave] holds nothing: no tile, no gather)
    }
  }
  uint4 ah[KS + 1], al[KS + 1];
  if (have0) tl_gather<KS>(Ehi, Elo, row0, half, ah, al);

  // ---- 1. the block's fragments: wave w takes k-steps w, w + 4, ...; a lane scales and splits piece (ks, half) of its query's
  // fp32 row (the encoder's epilogue or k_finalize_h_publish wrote it), with the batch's exponent reduced from x3m
  static_assert(TL_WAVES == 4, "x3_batch_exp: 256 threads");
  int eh, sexp;
  int32_t x3l[2];                         // (what x3s holds once block 0 has published: this kernel must not read it back)
  {
    const bool vec_ok = (d & 3) == 0 && (((uintptr_t)h_rows) & 15) == 0;
    constexpr int NKW = (KS + TL_WAVES - 1) / TL_WAVES;      // k-steps of a wave
    float yv[NKW][8];
    // the wave's pieces of the fp32 rows are requested BEFORE the exponent is reduced (they do not depend on it): one memory
    // latency of the block's start lies under the other
#pragma unroll
    for (int u = 0; u < NKW; ++u) {
      const int ks = wave + TL_WAVES * u;
      const int k0 = 16 * ks + 8 * half;
      if (ks < KS) {
        if (live && k0 + 8 <= d && vec_ok) {
          const float4 a = *(const float4*)(h_rows + q * d + k0), b = *(const float4*)(h_rows + q * d + k0 + 4);
          yv[u][0] = a.x; yv[u][1] = a.y; yv[u][2] = a.z; yv[u][3] = a.w; yv[u][4] = b.x; yv[u][5] = b.y; yv[u][6] = b.z; yv[u][7] = b.w;
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) yv[u][c] = (live && k0 + c < d) ? h_rows[q * d + k0 + c] : 0.f;
        }
      }
    }
    eh = x3_batch_exp(x3m, ent_exp, x3s, &s_n2[0][0]); sexp = eh + ent_exp; x3l[0] = eh; x3l[1] = sexp;
    __syncthreads();                        // (s_n2 served as the reduction's scratch)
    float n2 = 0.f;
#pragma unroll
    for (int u = 0; u < NKW; ++u) {
      const int ks = wave + TL_WAVES * u;
      if (ks < KS) {
        float y[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { n2 = fmaf(yv[u][c], yv[u][c], n2); y[c] = x3_scale(yv[u][c], eh); }
        uint4 h4, l4;
        split8_bf16(y, h4, l4);
        f3_store_piece(hf3, KS, q, ks, half, h4, l4, true);   // rows past B: zero pieces (the count kernel's tile is whole)
        s_bh[ks][lane] = h4;
        s_bl[ks][lane] = l4;
      }
    }
    n2 += __shfl_xor(n2, 32);
    if (half == 0) s_n2[wave][i] = n2;
  }
  TL_STAMP(1);
  __syncthreads();
  TL_STAMP(2);

  // ---- 2. round 0: every wave's first tile against the fragments in LDS; the filter waves hold their comparisons back until
  // the targets are published.  (Phases of a workgroup, COPER_DBG_TL_CLOCK build, tools/ab_tail.py -- all 640 workgroups of an
  // FB15k-237-shaped pass are resident at once and move in step, a sequence of chip-wide bursts of 16-byte gathers.  Round 3:
  // finalize 10.7 us, round 0 10.7, remaining tiles 7 - 13.  Round 4 before this form: fragments 5.1, round 0 12.2, rest 10 - 20.)
  float sc0 = 0.f;
  if (have0) {
    const f32x16 acc = tl_mma<KS>(bias_pad, s_e[wave], ah, al, s_bh, s_bl, lane, half, sexp);
    if (wave == 0) {
      // D[i][i] sits in lane i + 32 * ((i >> 2) & 1), register (i & 3) + 4 * (i >> 3)
      float diag = 0.f;
      const int reg = (i & 3) + 4 * (i >> 3);
#pragma unroll
      for (int r = 0; r < 16; ++r) diag = (r == reg) ? acc[r] : diag;
      float t0 = __shfl(diag, i + 32 * ((i >> 2) & 1));   // lane i (both halves): the target of query i
      t0 = erow >= 0 ? t0 : 0.f;
      if (half == 0) {
        float n2 = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < TL_WAVES; ++w2) n2 += s_n2[w2][i];
        // t0, the band and everything compared with them stay in the accumulators' units (x 2^(e_E + e_h)); tgt leaves descaled
        const float tau = x3_scale(x3_band_tau(n2, kappa, band_consts, d, x3l), sexp);
        s_t[i] = t0 + tau;
        if (live) { tgt[q] = x3_scale(t0, -sexp); tband[q] = make_float2(t0 - tau, t0 + tau); }
      }
    } else {
      sc0 = tl_entry_score(acc, qi0, i);
    }
  }
  // ---- 3. the remaining tiles; the ids of a wave's next tile are requested before the block waits for the targets
  int64_t pb = p_begin + 32 * (int64_t)(TL_WAVES - 1 + wave);
  int64_t f_cur = -1, f_prev = -1;
  if (pb < p_end) {
    const int64_t p = pb + i;
    f_cur = p < p_end ? idx[p] : -1;
    f_prev = (p < p_end && p > p_begin) ? idx[p - 1] : -1;
  }
  TL_STAMP(3);
  __syncthreads();
  const float t = s_t[i];
  if (wave != 0) {
    const float tq = __shfl(t, qi0);
    if (half == 0 && row0 >= 0 && sc0 > tq) atomicAdd(&s_corr[qi0], 1);
  }
  for (; pb < p_end; pb += 32 * TL_WAVES) {
    int qi;
    const int64_t frow = tl_prep(pb, p_end, my_lo, my_e2, f_cur, f_prev, n_local, s_e[wave], i, half, qi);
    tl_gather<KS>(Ehi, Elo, frow, half, ah, al);
    const int64_t pn = pb + 32 * TL_WAVES + i;      // the ids of the tile after this one
    f_cur = pn < p_end ? idx[pn] : -1;
    f_prev = (pn < p_end && pn > p_begin) ? idx[pn - 1] : -1;
    const f32x16 acc = tl_mma<KS>(bias_pad, s_e[wave], ah, al, s_bh, s_bl, lane, half, sexp);
    const float sc = tl_entry_score(acc, qi, i);
    const float tq = __shfl(t, qi);
    if (half == 0 && frow >= 0 && sc > tq) atomicAdd(&s_corr[qi], 1);
  }
  TL_STAMP(4);
  __syncthreads();
  // a pass that ran on a grouping prepared ahead whose ids had changed since (group_body.h: the guard; `stale` is the verdict of the
  // encoder launch in front of this one): every rank is written as COPER_RANK_STALE -- what the count and band launches add keeps it
  // negative -- and the pass is counted (coper_stale_passes)
  const bool is_stale = stale != nullptr && *stale != 0;
  if (wave == 0 && half == 0 && live) ranks[q] = is_stale ? COPER_RANK_STALE : 1 - s_corr[i];
  if (is_stale && blk == 0 && threadIdx.x == 0) atomicAdd(stale_count, 1);
  TL_STAMP(5);
}

#ifdef COPER_DBG_TL_CLOCK
extern "C" __attribute__((visibility("default"))) int coper_dbg_tl_clock(int n_wg, double* out /* [TL_NSTAMP] medians, us from the earliest start */) {
  static unsigned long long hbuf[TL_NSTAMP * 1024];
  if (hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_tl_clk), sizeof hbuf) != hipSuccess) return 1;
  if (n_wg > 1024) n_wg = 1024;
  unsigned long long t0 = ~0ull;
  for (int i = 0; i < n_wg; ++i) if (hbuf[TL_NSTAMP * i] && hbuf[TL_NSTAMP * i] < t0) t0 = hbuf[TL_NSTAMP * i];
  for (int j = 0; j < TL_NSTAMP; ++j) {
    std::vector<double> v;
    for (int i = 0; i < n_wg; ++i) if (hbuf[TL_NSTAMP * i]) v.push_back((double)(hbuf[TL_NSTAMP * i + j] - t0) * 0.01);
    if (v.empty()) return 2;
    std::sort(v.begin(), v.end());
    out[j] = v[v.size() / 2];
    out[TL_NSTAMP + j] = v.back();
  }
  return 0;
}
#endif

bool tail_fused_supported(const coper_handle* h) {
  static const bool off = getenv("COPER_TAIL_UNFUSED") != nullptr;   // A/B switch, read once
  if (off) return false;
  return (h->dm.KS16 == 13 || h->dm.KS16 == 16) && h->dm.n_local == h->dm.E;   // (the resident fragments: KS16 registers x 2 per lane)
}

// (Tried in round 3: the filter tiles moved out of this launch into extra workgroups of the band launch that follows the count
// kernel -- this launch 44 -> 25 us, but the band launch 16 -> 51: at ~190 registers a wave only two workgroups fit a CU and
// the 1,280 filter workgroups ran in three rounds.  Pass 0.536 against 0.516 ms; not kept.)
int launch_finalize_h_publish(coper_handle* h, int64_t B, int ksplit, float* h_out, hipStream_t s) {
  if (ksplit == 0) return COPER_OK;      // the fused encoder finalized in its own epilogue (kernels_dense_fused_bf16.hip: FusedFin)
  const Dims& dm = h->dm;
  const float* fcb = dm.gen_fc ? h->fc_b_rel : h->params["fc_bias"].ptr;
  int64_t blocks = (B + 7) / 8;                    // eight rows per block and round
  if (blocks > X3M_SLOTS) blocks = X3M_SLOTS;
  hipLaunchKernelGGL(k_finalize_h_publish, dim3((unsigned)blocks), dim3(256), 0, s, h->z_part, ksplit, h->ws_queries, B, dm.d,
                     dm.d_pad16, h->inv_perm, h->sorted_rid, fcb, dm.gen_fc ? 1 : 0, h->fc_scale, h->fc_shift, h->w_exp, h->x_exp, h_out,
                     h->x3m);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_finalize_targets_filter_bf16x3(coper_handle* h, int64_t B, int ksplit, float* h_out, const int64_t* e2, const int64_t* indptr,
                                          const int64_t* idx, int64_t nnz, float* tgt, int32_t* ranks, hipStream_t s) {
  const Dims& dm = h->dm;
  int rc0 = launch_finalize_h_publish(h, B, ksplit, h_out, s);
  if (rc0) return rc0;
  const int64_t rows_pad = (B + 127) / 128 * 128;
  const unsigned grid = (unsigned)(rows_pad / 32);
#define TL_GO(KS_)                                                                                                                 \
  hipLaunchKernelGGL(k_finalize_targets_filter_bf16x3<KS_>, dim3(grid), dim3(64 * TL_WAVES), 0, s, B, dm.d, (const float*)h_out,   \
                     (uint4*)h->hf3_ws, (const uint4*)h->Erm16_hi, (const uint4*)h->Erm16_lo,                                       \
                     h->bias_pad, dm.n_local, e2, indptr, idx, tgt, band_kappa(h), h->band_consts, h->x3m, h->x3_ent_exp, h->x3s,   \
                     (float2*)h->tband_ws, ranks, h->heavy_ws, h->pass_chk ? (const int32_t*)(h->pass_chk + GROUP_CHK_STALE) : nullptr, \
                     h->group_done + 2)
  if (dm.KS16 == 13) { TL_GO(13); } else { TL_GO(16); }
#undef TL_GO
  COPER_HIP_TRY(h, hipGetLastError());
  // blocks beyond a workgroup's own share: worked off beside the band walk of the count pass the caller launches next
  return launch_filter_excess_bf16x3(h, h_out, e2, indptr, idx, nnz, B, ranks, true, s);
}

}  // namespace coper
