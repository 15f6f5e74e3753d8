// COPER_SCORE_BF16X3: the 1-vs-all scorer / ranker on the bf16 matrix cores at fp32-class accuracy.
//
// gfx950 has no TF32/xf32 path and its exact-f32 MFMA runs at 1/16 of the bf16 rate, so the fp32 mode
// of kernels_score.hip is MFMA-bound at 157 TFLOP/s.  Here every fp32 operand is split once into two
// bf16 terms,  x = hi + lo  (hi = rne_bf16(x), lo = rne_bf16(x - hi), |x - hi - lo| <= 2^-16 |x|), and
// each product is formed as  lo*hi + hi*lo + hi*hi  on the bf16 matrix cores with fp32 accumulation
// (the dropped lo*lo term is <= 2^-16 relative): 3 MFMAs at the bf16 rate instead of 1 at the f32 rate,
// 16/3 = 5.3x the matrix throughput, logit error ~1e-5 relative to |h||E| -- inside the 1e-3 gate of BASELINE.json.
// Storage is two bf16 planes (same bytes as fp32).
//
// Every logit of this mode -- the count kernel (kernels_score3_bf16.hip, 16x16x32), the tiles of score_all and the
// (query, entity) pair kernel used for targets, filter correction and the sampled scorer (32x32x16) -- is the SAME
// sequence of K = 16 accumulation steps in the SAME order (bf16x3_chain.h), accumulators started from pred_bias, so they
// agree bit for bit (tests/test_gpu_parity.py checks it); an MFMA's result for element (i, j) does not depend on where
// the row / column sits in the tile.  RANKS of the mode do not rest on these bits alone: comparisons closer than the
// mode's error are decided by the fp32 chain (the exact band, kernels_score3_bf16.hip).
#include "bf16x3_chain.h"
#include "coper_internal.h"

namespace coper {

__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) { split8_s16(v, hi, lo); }

#define MFMA_BF16(a, b, c) BX3_MFMA32(a, b, c)

// ------------------------------------------------------------------------------------------------
// prepare / pack: fragment image  X16[(blk*KS16 + ks)*64 + l] = 8 bf16 { X[32blk + (l&31)][16ks + 8(l>>5) + j] }
// for the hi and the lo plane (the A/B operand map of v_mfma_f32_32x32x16_bf16), the row-major twins the pair kernels
// gather from, and the f3 image of the count kernel (bf16x3_chain.h): the same 16-byte pieces, placed three ways.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_to_frag_bf16(const float* __restrict__ src, int64_t n_rows, int d,
                                                           int KS16, uint4* __restrict__ hi, uint4* __restrict__ lo,
                                                           uint4* __restrict__ rm_hi, uint4* __restrict__ rm_lo,
                                                           uint4* __restrict__ f3, int query_side,
                                                           int64_t total, int32_t* __restrict__ cnt, int32_t cnt_base,
                                                           int32_t* __restrict__ cnt_eq, int fixed_exp,
                                                           const float* __restrict__ x3m, int32_t* __restrict__ x3s,
                                                           const int32_t* __restrict__ stale, int32_t* __restrict__ stale_count) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (blk*KS16 + ks)*64 + l
  // the power of two of this operand class (split16.h): the table's (prepare), or the packed batch's -- reduced here from the
  // block maxima k_absmax_publish / k_finalize_h_publish left, and published by block 0 for the kernels that follow
  __shared__ float s_red[4];
  const int e2x = x3m ? x3_batch_exp(x3m, fixed_exp, x3s, s_red) : fixed_exp;
  // optional: preset the rank counters of the pass this packing opens (saves a launch on the ranking path)
  // (a pass on a grouping prepared ahead whose ids had changed since -- group_body.h: the guard -- gets COPER_RANK_STALE in every rank)
  if (cnt) {
    const bool is_stale = stale != nullptr && *stale != 0;
    if (j < n_rows) { cnt[j] = is_stale ? COPER_RANK_STALE : cnt_base; if (cnt_eq) cnt_eq[j] = 0; }
    if (is_stale && j == 0) atomicAdd(stale_count, 1);
  }
  if (j >= total) return;
  int l = (int)(j & 63);
  int64_t rest = j >> 6;
  int ks = (int)(rest % KS16);
  int64_t blk = rest / KS16;
  int64_t row = blk * 32 + (l & 31);
  int k = 16 * ks + 8 * (l >> 5);
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = (row < n_rows && k + t < d) ? x3_scale(src[row * d + k + t], e2x) : 0.f;
  uint4 h4, l4;
  split8(v, h4, l4);
  hi[j] = h4;
  lo[j] = l4;
  if (rm_hi) {  // row-major twin of the planes: row r = 2*KS16 pieces of 16 B (the pair kernel gathers whole rows)
    int64_t o = row * (2 * KS16) + 2 * ks + (l >> 5);
    rm_hi[o] = h4;
    rm_lo[o] = l4;
  }
  if (f3) f3_store_piece(f3, KS16, row, ks, l >> 5, h4, l4, query_side != 0);
}

int launch_rows_to_frag_bf16(coper_handle* h, const float* src, int64_t n_rows, int64_t n_blk, uint4* hi, uint4* lo,
                             uint4* rm_hi, uint4* rm_lo, uint4* f3, bool query_side, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t total = n_blk * dm.KS16 * 64;
  hipLaunchKernelGGL(k_rows_to_frag_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n_rows, dm.d,
                     dm.KS16, hi, lo, rm_hi, rm_lo, f3, query_side ? 1 : 0, total, h->preset_cnt, h->count_base, h->preset_eq,
                     h->x3_ent_exp, query_side ? h->x3m : nullptr, h->x3s,
                     h->preset_cnt && h->pass_chk ? (const int32_t*)(h->pass_chk + GROUP_CHK_STALE) : nullptr, h->group_done ? h->group_done + 2 : nullptr);
  if (h->preset_cnt) h->counts_preset = h->preset_cnt;
  h->preset_cnt = nullptr;
  h->preset_eq = nullptr;
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

__global__ void k_zero_counts(int64_t B, int32_t* __restrict__ ng, int32_t* __restrict__ ne, int32_t base) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < B) { ng[j] = base; if (ne) ne[j] = 0; }
}

constexpr int BX_NQ = 4;    // 32-query blocks of a 128-query tile (the packing granule of the query planes)

// The largest |h| element of a batch of query rows, one maximum per block (bf16x3_chain.h: x3_block_store_max; the packing
// launch that follows reduces them to the batch's exponent e_h).
__global__ __launch_bounds__(256) void k_absmax_publish(const float* __restrict__ src, int64_t n, float* __restrict__ x3m) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
  x3_block_store_max(m, x3m);
}

int launch_absmax_publish(coper_handle* h, const float* src, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > X3M_SLOTS) blocks = X3M_SLOTS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_absmax_publish, dim3((unsigned)blocks), dim3(256), 0, s, src, n, h->x3m);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_pack_h_bf16(coper_handle* h, const float* hvec, int64_t B, hipStream_t s) {
  int rc0 = launch_absmax_publish(h, hvec, B * h->dm.d, s);
  if (rc0) return rc0;
  int64_t n_blk = (B + 32 * BX_NQ - 1) / (32 * BX_NQ) * BX_NQ;
  return launch_rows_to_frag_bf16(h, hvec, B, n_blk, (uint4*)h->hfrag16_hi, (uint4*)h->hfrag16_lo, (uint4*)h->hrm16_hi,
                                  (uint4*)h->hrm16_lo, (uint4*)h->hf3_ws, true, s);
}

// counters start from count_base unless the caller's pack launch preset them (coper_rank: n_greater accumulates
// straight into `ranks`, started from 1)
void score_count_begin_bf16x3(coper_handle* h, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s) {
  if (h->counts_preset != ng)
    hipLaunchKernelGGL(k_zero_counts, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, ng, ne, h->count_base);
  h->counts_preset = nullptr;
}

// the whole packed batch in one count launch (+ the exact decision of its band)
int launch_score_count_bf16x3(coper_handle* h, const float* hvec, const float* tgt_x, const int64_t* e2, const int64_t* indptr,
                              const int64_t* idx, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s) {
  score_count_begin_bf16x3(h, B, ng, ne, s);
  // chunks of queries bound the band mask (one bit per logit of a launch): 20,480 queries against 10 M entities would be 25 GB
  const int64_t qc = topk_chunk_queries(h->dm.n_eblk, B, h->gmax_max_floats);
  for (int64_t q0 = 0; q0 < B; q0 += qc) {
    const int rc = score_count3_chunk_bf16x3(h, q0, B - q0 < qc ? B - q0 : qc, hvec, tgt_x, e2, indptr, idx, ng, ne, nullptr, 0, s);
    if (rc) return rc;
  }
  return COPER_OK;
}

int score_bf16_kernels_init(coper_handle* h) { (void)h; return COPER_OK; }

// ------------------------------------------------------------------------------------------------
// logits out (predictions_all): operand roles swapped relative to score_count (queries = A rows, entities
// = B columns) so that a register row is 32 consecutive entities of one query -> 128-B contiguous stores.
// The three MFMAs of a k-step multiply the same (entity, query) terms in the same order, and a product
// does not depend on which side of the matrix unit its factors enter: the bits equal score_count's.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_score_all_bf16x3(const uint4* __restrict__ Ehi,
                                                             const uint4* __restrict__ Elo,
                                                             const float* __restrict__ bias_pad,
                                                             const uint4* __restrict__ Hhi,
                                                             const uint4* __restrict__ Hlo, int64_t B, int KS,
                                                             int64_t n_eblk, int64_t n_local,
                                                             float* __restrict__ logits, int64_t ld,
                                                             const int32_t* __restrict__ x3s) {
  constexpr int NQ = 2, ME = 2;
  const int sexp = x3s[1];   // accumulators carry 2^(e_E + e_h) (split16.h)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t qblk0 = (int64_t)blockIdx.x * NQ;  // 32-query blocks (hfrag is packed per 32-query block)
  const int64_t eb0 = ((int64_t)blockIdx.y * 4 + wave) * ME;
  if (eb0 >= n_eblk) return;
  f32x16 acc[NQ][ME];
#pragma unroll
  for (int a = 0; a < ME; ++a) {
    float bv = x3_scale(bias_pad[(eb0 + a) * 32 + (lane & 31)], sexp);
#pragma unroll
    for (int b = 0; b < NQ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][a][r] = bv;
  }
  uint4 eh0[ME], el0[ME], qh0[NQ], ql0[NQ], eh1[ME], el1[ME], qh1[NQ], ql1[NQ];
  uint4 eh2[ME], el2[ME], qh2[NQ], ql2[NQ], eh3[ME], el3[ME], qh3[NQ], ql3[NQ];
#define KCL(k_) ((k_) < KS ? (k_) : KS - 1)
#define LOAD_EQ(eh, el, qh, ql, ks_)                                          \
  {                                                                           \
    _Pragma("unroll") for (int a = 0; a < ME; ++a) {                          \
      int64_t o = ((eb0 + a) * KS + (ks_)) * 64 + lane;                       \
      eh[a] = Ehi[o];                                                         \
      el[a] = Elo[o];                                                         \
    }                                                                         \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) {                          \
      int64_t o = ((qblk0 + b) * KS + (ks_)) * 64 + lane;                     \
      qh[b] = Hhi[o];                                                         \
      ql[b] = Hlo[o];                                                         \
    }                                                                         \
  }
  // the virtual ops of bf16x3_chain.h with the query as A: pairs of k-steps, then the last one of an odd count
#define STEP_PAIR(ehA, elA, qhA, qlA, ehB, elB, qhB, qlB)                                            \
  {                                                                                                  \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) _Pragma("unroll") for (int a = 0; a < ME; ++a)    \
      BX3_PAIR_QA(ehA[a], elA[a], qhA[b], qlA[b], ehB[a], elB[a], qhB[b], qlB[b], acc[b][a]);        \
  }
#define STEP_LAST(eh, el, qh, ql)                                                                    \
  {                                                                                                  \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) _Pragma("unroll") for (int a = 0; a < ME; ++a)    \
      BX3_LAST_QA(eh[a], el[a], qh[b], ql[b], acc[b][a]);                                            \
  }
  LOAD_EQ(eh0, el0, qh0, ql0, 0);
  LOAD_EQ(eh1, el1, qh1, ql1, KCL(1));
  int ks = 0;
  for (; ks + 4 <= KS; ks += 4) {
    LOAD_EQ(eh2, el2, qh2, ql2, KCL(ks + 2));
    LOAD_EQ(eh3, el3, qh3, ql3, KCL(ks + 3));
    __builtin_amdgcn_sched_barrier(0);
    STEP_PAIR(eh0, el0, qh0, ql0, eh1, el1, qh1, ql1);
    LOAD_EQ(eh0, el0, qh0, ql0, KCL(ks + 4));
    LOAD_EQ(eh1, el1, qh1, ql1, KCL(ks + 5));
    __builtin_amdgcn_sched_barrier(0);
    STEP_PAIR(eh2, el2, qh2, ql2, eh3, el3, qh3, ql3);
  }
  if (ks + 2 <= KS) {   // two or three k-steps left; buffers 0 / 1 hold the first two
    LOAD_EQ(eh2, el2, qh2, ql2, KCL(ks + 2));
    __builtin_amdgcn_sched_barrier(0);
    STEP_PAIR(eh0, el0, qh0, ql0, eh1, el1, qh1, ql1);
    if (ks + 2 < KS) STEP_LAST(eh2, el2, qh2, ql2);
  } else if (ks < KS) {
    STEP_LAST(eh0, el0, qh0, ql0);
  }
#undef KCL
#undef LOAD_EQ
#undef STEP_PAIR
#undef STEP_LAST
#pragma unroll
  for (int b = 0; b < NQ; ++b)
#pragma unroll
    for (int a = 0; a < ME; ++a) {
      int64_t e = (eb0 + a) * 32 + (lane & 31);
      if (e >= n_local) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t q = (qblk0 + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (q < B) logits[q * ld + e] = x3_scale(acc[b][a][r], -sexp);
      }
    }
}

int launch_score_all_bf16x3(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, hipStream_t s) {
  const Dims& dm = h->dm;
  int rc = launch_pack_h_bf16(h, hvec, B, s);
  if (rc) return rc;
  int64_t q_groups = (B + 63) / 64;
  int64_t e_groups = (dm.n_eblk + 7) / 8;
  if (e_groups > 65535) return fail(h, COPER_EUNSUPPORTED, "score_all: shard too large to materialise logits");
  ScopedKernelTimer t(h, "score_all", s);
  hipLaunchKernelGGL(k_score_all_bf16x3, dim3((unsigned)q_groups, (unsigned)e_groups), dim3(256), 0, s,
                     (const uint4*)h->Ef16_hi, (const uint4*)h->Ef16_lo, h->bias_pad, (const uint4*)h->hfrag16_hi,
                     (const uint4*)h->hfrag16_lo, B, dm.KS16, dm.n_eblk, dm.n_local, logits, ld, h->x3s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// (query, entity) pairs through the same MFMA sequence: a wave takes 32 pairs, gathers their entity and
// query fragments straight from the two fragment images, and reads the diagonal of the 32x32 tile.
//   mode 0  targets:  pair p = query p with e2[p]          -> out[p] = logit (0 outside the shard)
//   mode 1  lookup:   pair p = (b = p / L, lookup[p])      -> out[p] = logit (0 outside the shard)
//   mode 2  filter:   pair p = CSR entry p of query row[p] -> subtract from ng / ne what score_count counted
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pair_bf16x3(const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo,
                                                     const float* __restrict__ bias_pad,
                                                     const uint4* __restrict__ Hhi, const uint4* __restrict__ Hlo,
                                                     int KS, int mode, int64_t n_pairs, int64_t B, int64_t L,
                                                     const int64_t* __restrict__ e2, const int32_t* __restrict__ lookup,
                                                     const int64_t* __restrict__ indptr, const int64_t* __restrict__ idx,
                                                     const int32_t* __restrict__ row_of, const float2* __restrict__ tband,
                                                     int64_t lo, int64_t n_local, float* __restrict__ out,
                                                     int32_t* __restrict__ ng, const int32_t* __restrict__ x3s) {
  __shared__ int64_t s_e[4][32];
  const int sexp = x3s[1];   // accumulators (and tband) carry 2^(e_E + e_h); logits leave descaled
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 31, half = lane >> 5;
  const int64_t p = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 32 + i;
  int64_t q = 0, erow = -1;  // erow < 0: nothing to score for this pair
  if (p < n_pairs) {
    if (mode == 0) {
      q = p;
      erow = e2[p] - lo;
      // the ranking flows hand the CSR filter to the target pass too: it writes the row id of every filter entry
      // (consumed by the filter pass that follows on the same stream) -- no separate expansion launch, no bisection
      // (a lane writes the first 32 entries of its row; longer rows are finished by the whole wave below)
      if (indptr && half == 0) {
        const int64_t j0 = indptr[p], j1 = indptr[p + 1];
        for (int64_t j = j0; j < j1 && j < j0 + 32; ++j) const_cast<int32_t*>(row_of)[j] = (int32_t)p;
      }
    } else if (mode == 1) {
      q = p / L;
      erow = (int64_t)lookup[p] - lo;
    } else if (p >= indptr[B]) {
      erow = -1;  // the launch may be sized by a capacity larger than the CSR (hipGraph replay): no query owns p
    } else {
      if (row_of) {
        q = row_of[p];
      } else {   // CSR row of entry p: the last b with indptr[b] <= p
        int64_t lo_b = 0, hi_b = B;
        while (hi_b - lo_b > 1) {
          const int64_t mid = (lo_b + hi_b) >> 1;
          if (indptr[mid] <= p) lo_b = mid; else hi_b = mid;
        }
        q = lo_b;
      }
      int64_t f = idx[p];
      erow = f - lo;
      if (p > indptr[q] && idx[p - 1] == f) erow = -1;  // adjacent duplicate: the dense mask is idempotent
      if (f == e2[q]) erow = -1;                        // the target is restored after masking (metrics.py:46)
    }
    if (erow >= n_local) erow = -1;
  }
  if (mode == 0 && indptr) {
    // rows with more than 32 known answers (real KGs have rows with thousands): the wave fills the rest of row_of together,
    // one such row at a time -- a single lane walking 5,000 entries held the whole launch back by ~50 us
    const bool is_long = half == 0 && p < n_pairs && indptr[p + 1] - indptr[p] > 32;
    unsigned long long todo = __ballot(is_long);
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int64_t pp = __shfl(p, src);
      const int64_t j0 = indptr[pp] + 32, j1 = indptr[pp + 1];
      for (int64_t j = j0 + lane; j < j1; j += 64) const_cast<int32_t*>(row_of)[j] = (int32_t)pp;
    }
  }
  if (half == 0) s_e[wave][i] = erow;
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): wave-local LDS exchange
  __builtin_amdgcn_wave_barrier();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int64_t er = s_e[wave][(r & 3) + 8 * (r >> 2) + 4 * half];
    acc[r] = er >= 0 ? x3_scale(bias_pad[er], sexp) : 0.f;
  }
  const int64_t ea = erow >= 0 ? erow : 0;
  // row-major twins of the planes: a lane walks its own row (32 B per k-step), so every fetched line is used whole
  const uint4* pa_h = Ehi + ea * (2 * KS) + half;
  const uint4* pa_l = Elo + ea * (2 * KS) + half;
  const uint4* pb_h = Hhi + q * (2 * KS) + half;
  const uint4* pb_l = Hlo + q * (2 * KS) + half;
  // gathered 16-B loads, batched PAIR_BATCH k-steps deep (few waves per SIMD here: nothing else hides their latency;
  // d = 200 is 13 k-steps); a short last batch re-reads the final k-step and skips its MFMAs
#ifndef COPER_PAIR_BATCH
#define COPER_PAIR_BATCH 4
#endif
  constexpr int PB = COPER_PAIR_BATCH;
  static_assert(PB % 2 == 0, "batches hold whole pairs of k-steps (bf16x3_chain.h)");
  for (int ks = 0; ks < KS; ks += PB) {
    uint4 ah[PB], al[PB], bh[PB], bl[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int k = ks + u < KS ? ks + u : KS - 1;
      ah[u] = pa_h[k * 2]; al[u] = pa_l[k * 2];
      bh[u] = pb_h[k * 2]; bl[u] = pb_l[k * 2];
    }
#pragma unroll
    for (int u = 0; u < PB; u += 2) {   // wave-uniform
      if (ks + u + 1 < KS) { BX3_PAIR(ah[u], al[u], bh[u], bl[u], ah[u + 1], al[u + 1], bh[u + 1], bl[u + 1], acc); }
      else if (ks + u < KS) { BX3_LAST(ah[u], al[u], bh[u], bl[u], acc); }
    }
  }
  // D[i][i] sits in lane i + 32*((i>>2)&1), register (i&3) + 4*(i>>3)
  const bool diag_lane = ((i >> 2) & 1) == half && p < n_pairs;
  const int reg = (i & 3) + 4 * (i >> 3);
  float sc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sc = (r == reg) ? acc[r] : sc;
  if (mode != 2) {
    if (diag_lane) out[p] = erow >= 0 ? x3_scale(sc, -sexp) : 0.f;
    return;
  }
  // filter correction: what the count kernel counted for a known answer -- its logit ABOVE THE BAND of the query's target
  // (sc > t_hi; entries inside the band are not counted by the count kernel and are skipped by k_band_exact) -- is taken
  // back.  Consecutive entries belong to the same query (CSR order), so the wave adds up each run of equal query ids and
  // issues ONE atomic per run -- a row with 5,000 known answers was 5,000 atomics on one address (~100 us), now 157.
  // Every lane runs the cross-lane steps; lanes 0..31 stand for entries 0..31.
  const bool valid = diag_lane && erow >= 0;
  const float t_hi = valid ? tband[q].y : 0.f;
  const int src = i + 32 * ((i >> 2) & 1);                       // the lane that holds entry i's diagonal value
  const int hit_g = __shfl((valid && sc > t_hi) ? 1 : 0, src);
  const int64_t q_prev = __shfl_up(q, 1);
  const bool in_tile = p < n_pairs && p < indptr[B];
  const bool head = half == 0 && in_tile && (i == 0 || q_prev != q);
  const unsigned heads = (unsigned)(__ballot(head) & 0xFFFFFFFFull);
  const unsigned m_g = (unsigned)(__ballot(half == 0 && hit_g) & 0xFFFFFFFFull);
  if (head) {
    const unsigned later = i < 31 ? (heads >> (i + 1)) : 0u;
    const int end = later ? i + 1 + __builtin_ctz(later) : 32;   // entries [i, end) share this lane's query
    const unsigned run = (end >= 32 ? 0xFFFFFFFFu : ((1u << end) - 1u)) & ~((1u << i) - 1u);
    const int cg = __builtin_popcount(m_g & run);
    if (cg) atomicSub(&ng[q], cg);
  }
}

// CSR -> row id per entry (callers of coper_rank_counts that did not come through a target pass with the CSR)
__global__ void k_expand_rows(const int64_t* __restrict__ indptr, int64_t B, int32_t* __restrict__ row_of) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = gid >> 4;
  int sub = (int)(gid & 15);
  if (b >= B) return;
  for (int64_t i = indptr[b] + sub; i < indptr[b + 1]; i += 16) row_of[i] = (int32_t)b;
}

static void pair_launch(coper_handle* h, int mode, int64_t n_pairs, int64_t B, int64_t L, const int64_t* e2,
                        const int32_t* lookup, const int64_t* indptr, const int64_t* idx, const int32_t* row_of,
                        const float2* tband, float* out, int32_t* ng, hipStream_t s) {
  const Dims& dm = h->dm;
  if (n_pairs <= 0) return;
  // few pairs (the target pass: one per query): one wave per workgroup spreads them over all CUs
  const int wpb = n_pairs <= (1 << 22) ? 1 : 4;
  hipLaunchKernelGGL(k_pair_bf16x3, dim3((unsigned)((n_pairs + 32 * wpb - 1) / (32 * wpb))), dim3(64 * wpb), 0, s, (const uint4*)h->Erm16_hi,
                     (const uint4*)h->Erm16_lo, h->bias_pad, (const uint4*)h->hrm16_hi, (const uint4*)h->hrm16_lo,
                     dm.KS16, mode, n_pairs, B, L, e2, lookup, indptr, idx, row_of, tband, (int64_t)h->cfg.shard_lo,
                     dm.n_local, out, ng, h->x3s);
}

int launch_pair_targets_bf16x3(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt,
                               hipStream_t s) {
  int rc = launch_pack_h_bf16(h, hvec, B, s);
  if (rc) return rc;
  h->packed_hvec = hvec;  // coper_rank reuses this packing for the count pass on the same stream
  h->packed_B = B;
  pair_launch(h, 0, B, B, 1, e2, nullptr, h->expand_indptr, nullptr, h->expand_indptr ? h->row_of_ws : nullptr, nullptr, tgt, nullptr, s);
  if (h->expand_indptr) h->rows_expanded_for = h->expand_indptr;
  h->expand_indptr = nullptr;
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// targets of queries whose planes are already in place (coper_encode_rank: written by k_dense_finalize_pack)
int launch_pair_targets_packed_bf16x3(coper_handle* h, const int64_t* e2, int64_t B, float* tgt, hipStream_t s) {
  pair_launch(h, 0, B, B, 1, e2, nullptr, h->expand_indptr, nullptr, h->expand_indptr ? h->row_of_ws : nullptr, nullptr, tgt, nullptr, s);
  if (h->expand_indptr) h->rows_expanded_for = h->expand_indptr;
  h->expand_indptr = nullptr;
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_score_lookup_bf16x3(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L,
                               float* out, hipStream_t s) {
  int rc = launch_pack_h_bf16(h, hvec, B, s);
  if (rc) return rc;
  pair_launch(h, 1, B * L, B, L, nullptr, lookup, nullptr, nullptr, nullptr, nullptr, out, nullptr, s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// takes back, for every known answer, what the count kernel counted for it (logit above the band: tband_ws)
int launch_filter_correct_bf16x3(coper_handle* h, const int64_t* e2, const int64_t* indptr, const int64_t* idx, int64_t nnz,
                                 int64_t B, int32_t* ng, hipStream_t s) {
  // row ids were written by the target pass of coper_rank / coper_encode_rank; other callers of coper_rank_counts get them
  // from an expansion launch (long rows) -- the pair kernel's bisection of indptr serves short CSRs
  const int32_t* rows = h->rows_expanded_for == indptr ? h->row_of_ws : nullptr;
  h->rows_expanded_for = nullptr;
  if (!rows && nnz > 4096) {
    int64_t threads = B * 16;
    hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, indptr, B, h->row_of_ws);
    rows = h->row_of_ws;
  }
  pair_launch(h, 2, nnz, B, 1, e2, nullptr, indptr, idx, rows, (const float2*)h->tband_ws, nullptr, ng, s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
