// Scoring / ranking kernels: ConvE._compute_likelihoods (models.py:428-446) and the filtered ranker
// of ranking_and_hits (metrics.py:44-50), fp32-exact mode.
//
// Every logit in this file is the SAME fp32 fma chain
//     s = pred_bias[e];  for ks: for t in 0..3:  s = fma(E[e][8ks+t],   h[q][8ks+t],   s)
//                                                s = fma(E[e][8ks+4+t], h[q][8ks+4+t], s)
// (k-pairs (k, k+4) because lanes 0-31 / 32-63 of v_mfma_f32_32x32x2_f32 carry the two k of one
// instruction), whether it is produced by the MFMA tiles (score_all, score_count) or by the VALU
// pair kernels (targets, filter correction, score_lookup).  The f32 MFMA is a k-ordered fmaf chain
// with one rounding per product, so all five agree bit for bit; tests/test_gpu_score.py checks it.
//
//   score_all     logits[B, n_local] (predictions_all).  A = h tile (LDS, fragment-major), B = entity
//                 fragments streamed from the prepare-time image, 1 KiB per wave-instruction.
//   score_count   fused 1-vs-all ranker: same tiles with the roles swapped (entity rows on the
//                 accumulator registers, one query per lane) so that counting
//                 #{logit > target}, #{logit == target} is lane-local; logits never leave registers.
//   filter_correct  subtracts, per known answer in the CSR filter (and for the target itself), what
//                 score_count counted for it -- the sparse form of `pred[e2_multi == 1] = -inf`
//                 followed by restoring the target (metrics.py:45-46).
#include "coper_internal.h"

namespace coper {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define F4C(v, t) ((t) == 0 ? (v).x : (t) == 1 ? (v).y : (t) == 2 ? (v).z : (v).w)

// h[q0 .. q0+32*NQ) rows -> LDS in MFMA-fragment order: hl[(qb*KS + ks)*64 + l] =
//   float4{ h[q0 + 32qb + (l&31)][8ks + 4(l>>5) + t] }, zero padded.
template <int NQ>
__device__ __forceinline__ void stage_h_frag(float4* hl, const float* __restrict__ hvec, int64_t q0, int64_t B,
                                             int d, int KS) {
  for (int j = threadIdx.x; j < NQ * KS * 64; j += 256) {
    int l = j & 63;
    int ks = (j >> 6) % KS;
    int qb = (j >> 6) / KS;
    int64_t q = q0 + qb * 32 + (l & 31);
    int k = 8 * ks + 4 * (l >> 5);
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = (q < B && k + t < d) ? hvec[q * d + k + t] : 0.f;
    hl[j] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// fused score + count (the dominant kernel).
//
// Persistent, statically balanced: the work is U = q_tiles x iters units, one unit = 128 queries x
// 256 entities (8 waves x one 32-row entity block each); workgroup w of G = #CUs owns the contiguous
// unit range [U*w/G, U*(w+1)/G) in query-tile-major order, so every workgroup does the same number of
// MFMAs (no tail) and re-stages its query tile at most twice.  512 threads = 2 waves per SIMD sharing
// one 128-query h tile in LDS (pre-packed in fragment order by k_pack_h: staging is a straight copy);
// each wave streams its own entity block from Ef straight into VGPRs, software-pipelined one k-step
// ahead.  D[i][j]: i = entity row (A operand, on the accumulator registers), j = query (B operand, one
// per lane): counting #{s > t}, #{s == t} is lane-local.
// ------------------------------------------------------------------------------------------------
constexpr int SC_NQ = 4;      // 32-query blocks per workgroup tile
constexpr int SC_WAVES = 8;   // entity blocks per unit

__global__ void k_pack_h(const float* __restrict__ hvec, int64_t B, int d, int KS, float4* __restrict__ hfrag,
                         int64_t total, int32_t* __restrict__ ng, int32_t* __restrict__ ne) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // ((qtile*NQ + qb)*KS + ks)*64 + l
  if (ng && j < B) { ng[j] = 0; if (ne) ne[j] = 0; }  // the count buffers start from zero (saves two memset nodes)
  if (j >= total) return;
  int l = (int)(j & 63);
  int64_t rest = j >> 6;
  int ks = (int)(rest % KS);
  int64_t qblk = rest / KS;
  int64_t q = qblk * 32 + (l & 31);
  int k = 8 * ks + 4 * (l >> 5);
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = (q < B && k + t < d) ? hvec[q * d + k + t] : 0.f;
  hfrag[j] = make_float4(v[0], v[1], v[2], v[3]);
}

// EQ = false: the caller does not want n_equal (ties are a diagnostic the reference never computes): the
// epilogue is one compare per score instead of two.
// GM: also write the largest logit of every (32-entity block, query) for the pruned top-k (kernels_topk_bf16.hip).
template <bool EQ, bool GM>
__global__ __launch_bounds__(512, 2) void k_score_count_f32(const float4* __restrict__ Ef,
                                                            const float* __restrict__ bias_pad,
                                                            const float4* __restrict__ hfrag,
                                                            const float* __restrict__ tgt, int64_t B, int KS,
                                                            int64_t iters, int64_t units,
                                                            int32_t* __restrict__ ng, int32_t* __restrict__ ne,
                                                            float* __restrict__ gmax, int64_t gm_stride) {
  constexpr int NQ = SC_NQ;
  extern __shared__ float4 hl[];  // [NQ][KS][64]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t u_begin = units * blockIdx.x / gridDim.x;
  const int64_t u_end = units * (blockIdx.x + 1) / gridDim.x;
  float t[NQ];
  int cg[NQ], ce[NQ];
  int64_t cur_tile = -1;

  float4 a0, a1, b0[NQ], b1[NQ], bq[4];
#ifdef COPER_DBG_NO_GLOADS
#define EF_AT(ebx, ks_) Ef[lane]
#else
#define EF_AT(ebx, ks_) Ef[((ebx)*KS + (ks_)) * 64 + lane]
#endif
#ifdef COPER_DBG_NO_LDS
#define LOAD_B(bv, ks_) \
  { _Pragma("unroll") for (int b = 0; b < NQ; ++b) bv[b] = hl[b * 64 + lane]; }
#else
#define LOAD_B(bv, ks_) \
  { _Pragma("unroll") for (int b = 0; b < NQ; ++b) bv[b] = hl[(b * KS + (ks_)) * 64 + lane]; }
#endif
#define LOAD_BIAS(ebx)                                                                 \
  {                                                                                    \
    const float4* bp = (const float4*)(bias_pad + (ebx)*32 + 4 * (lane >> 5));         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bq[j] = bp[2 * j];                   \
  }
#define MFMA_STEP(av, bv)                                                                                  \
  {                                                                                                        \
    _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) _Pragma("unroll") for (int b = 0; b < NQ; ++b) acc[b] = \
        __builtin_amdgcn_mfma_f32_32x32x2f32(F4C(av, tt), F4C(bv[b], tt), acc[b], 0, 0, 0);                 \
  }
#define FLUSH_COUNTS()                                                        \
  {                                                                           \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) {                          \
      int g = cg[b] + __shfl_xor(cg[b], 32);                                  \
      int e = EQ ? ce[b] + __shfl_xor(ce[b], 32) : 0;                         \
      int64_t q = cur_tile * (32 * NQ) + b * 32 + (lane & 31);                \
      if (lane < 32 && q < B) {                                               \
        if (g) atomicAdd(&ng[q], g);                                          \
        if (EQ && e) atomicAdd(&ne[q], e);                                    \
      }                                                                       \
    }                                                                         \
  }

  if (u_begin < u_end) {
    int64_t eb = (u_begin % iters) * SC_WAVES + wave;
    a0 = EF_AT(eb, 0);
    LOAD_BIAS(eb);
  }
  for (int64_t u = u_begin; u < u_end; ++u) {
    const int64_t tile = u / iters;
    const int64_t eb = (u % iters) * SC_WAVES + wave;
    if (tile != cur_tile) {  // workgroup-uniform
      if (cur_tile >= 0) FLUSH_COUNTS();
      __syncthreads();
      const float4* src = hfrag + tile * (NQ * KS * 64);
      for (int j = threadIdx.x; j < NQ * KS * 64; j += 512) hl[j] = src[j];
      cur_tile = tile;
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        int64_t q = tile * (32 * NQ) + b * 32 + (lane & 31);
        t[b] = q < B ? tgt[q] : INFINITY;
        cg[b] = 0;
        ce[b] = 0;
      }
      __syncthreads();
    }
    f32x16 acc[NQ];
    // accumulator row of reg r: (r&3) + 8(r>>2) + 4(lane>>5): the chain starts from pred_bias
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        acc[b][4 * j + 0] = bq[j].x; acc[b][4 * j + 1] = bq[j].y;
        acc[b][4 * j + 2] = bq[j].z; acc[b][4 * j + 3] = bq[j].w;
      }
    LOAD_B(b0, 0);
    int ks = 0;
    for (; ks + 2 <= KS; ks += 2) {
      a1 = EF_AT(eb, ks + 1);
      LOAD_B(b1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs it overlaps
      MFMA_STEP(a0, b0);
      const int kn = ks + 2 < KS ? ks + 2 : KS - 1;
      a0 = EF_AT(eb, kn);
      LOAD_B(b0, kn);
      __builtin_amdgcn_sched_barrier(0);
      MFMA_STEP(a1, b1);
    }
    if (ks < KS) MFMA_STEP(a0, b0);  // odd KS: a0/b0 hold k-step KS-1
    if (u + 1 < u_end) {             // next unit's first fragment + bias rows, ahead of the compares
      const int64_t ebn = ((u + 1) % iters) * SC_WAVES + wave;
      a0 = EF_AT(ebn, 0);
      LOAD_BIAS(ebn);
    }
#ifdef COPER_DBG_NO_EPILOGUE
#pragma unroll
    for (int b = 0; b < NQ; ++b) cg[b] += (acc[b][0] + acc[b][5] + acc[b][10] + acc[b][15] > t[b]) ? 1 : 0;
#else
#pragma unroll
    for (int b = 0; b < NQ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sc = acc[b][r];
        cg[b] += (sc > t[b]) ? 1 : 0;
        if (EQ) ce[b] += (sc == t[b]) ? 1 : 0;
      }
#endif
    if (GM) {
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        float mx = acc[b][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[b][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (lane < 32) gmax[eb * gm_stride + cur_tile * (32 * NQ) + b * 32 + lane] = mx;
      }
    }
  }
  if (cur_tile >= 0) FLUSH_COUNTS();
#undef EF_AT
#undef LOAD_B
#undef LOAD_BIAS
#undef MFMA_STEP
#undef FLUSH_COUNTS
}

// packs h for the whole batch and zeroes the counters
void score_count_begin_f32(coper_handle* h, const float* hvec, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t q_tiles = (B + 32 * SC_NQ - 1) / (32 * SC_NQ);
  int64_t total = q_tiles * SC_NQ * dm.KS * 64;
  hipLaunchKernelGGL(k_pack_h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hvec, B, dm.d, dm.KS,
                     (float4*)h->hfrag_ws, total, ng, ne);
}

// queries [q0, q0 + Bc) of the packed batch (q0 a multiple of the 128-query tile); gmax != NULL: block maxima too
int score_count_chunk_f32(coper_handle* h, int64_t q0, int64_t Bc, const float* tgt, int32_t* ng, int32_t* ne, float* gmax,
                          int64_t gm_stride, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t q_tiles = (Bc + 32 * SC_NQ - 1) / (32 * SC_NQ);
  int64_t iters = dm.n_eblk / SC_WAVES;  // n_eblk is padded to EBLK_ALIGN = SC_WAVES
  int64_t units = q_tiles * iters;
  int64_t grid = h->num_cus;
  if (grid > units) grid = units;
  size_t lds = (size_t)SC_NQ * dm.KS * 64 * sizeof(float4);
  const float4* hf = (const float4*)h->hfrag_ws + (q0 / 32) * dm.KS * 64;
  ScopedKernelTimer t(h, "score_count", s);
#define SC_LAUNCH(EQ_, GM_)                                                                                                    \
  hipLaunchKernelGGL((k_score_count_f32<EQ_, GM_>), dim3((unsigned)grid), dim3(512), lds, s, (const float4*)h->Ef, h->bias_pad, \
                     hf, tgt + q0, Bc, dm.KS, iters, units, ng + q0, ne ? ne + q0 : nullptr, gmax, gm_stride)
  if (gmax) { if (ne) SC_LAUNCH(true, true); else SC_LAUNCH(false, true); }
  else      { if (ne) SC_LAUNCH(true, false); else SC_LAUNCH(false, false); }
#undef SC_LAUNCH
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_score_count(coper_handle* h, const float* hvec, const float* tgt, int64_t B, int32_t* ng, int32_t* ne,
                       hipStream_t s) {
  score_count_begin_f32(h, hvec, B, ng, ne, s);
  return score_count_chunk_f32(h, 0, B, tgt, ng, ne, nullptr, 0, s);
}

int score_kernels_init(coper_handle* h) {
  const Dims& dm = h->dm;
  int lds = (int)((size_t)SC_NQ * dm.KS * 64 * sizeof(float4));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_f32<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_f32<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_f32<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_f32<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return COPER_OK;
}


// ------------------------------------------------------------------------------------------------
// logits out.  D[i][j]: i = query (A operand from LDS), j = entity (B operand, one entity per lane):
// a register row is 32 consecutive entities of one query -> 128-B contiguous stores.
// ------------------------------------------------------------------------------------------------
template <int NQ, int ME>
__global__ __launch_bounds__(256, 2) void k_score_all_f32(const float4* __restrict__ Ef,
                                                          const float* __restrict__ bias_pad,
                                                          const float4* __restrict__ hfrag, int64_t B, int KS,
                                                          int64_t n_eblk, int64_t n_local,
                                                          float* __restrict__ logits, int64_t ld) {
  // both operands stream from their fragment images (h pre-packed by k_pack_h), one k-step ahead; no LDS
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t qb0 = (int64_t)blockIdx.x * NQ;  // 32-query blocks
  const int64_t q0 = qb0 * 32;
  const int64_t eb0 = ((int64_t)blockIdx.y * 4 + wave) * ME;
  if (eb0 >= n_eblk) return;
  f32x16 acc[NQ][ME];
#pragma unroll
  for (int a = 0; a < ME; ++a) {
    float bv = bias_pad[(eb0 + a) * 32 + (lane & 31)];
#pragma unroll
    for (int b = 0; b < NQ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][a][r] = bv;
  }
  const float4* ep = Ef + (eb0 * KS) * 64 + lane;
  const float4* hp = hfrag + (qb0 * KS) * 64 + lane;
  float4 e0[ME], h0[NQ], e1[ME], h1[NQ];
#define LOAD_EH(ev, hv, ks_)                                                              \
  {                                                                                       \
    _Pragma("unroll") for (int a = 0; a < ME; ++a) ev[a] = ep[((int64_t)a * KS + (ks_)) * 64]; \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) hv[b] = hp[((int64_t)b * KS + (ks_)) * 64]; \
  }
#define MFMA_EH(ev, hv)                                                                                     \
  {                                                                                                         \
    _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) _Pragma("unroll") for (int b = 0; b < NQ; ++b)          \
        _Pragma("unroll") for (int a = 0; a < ME; ++a) acc[b][a] =                                           \
            __builtin_amdgcn_mfma_f32_32x32x2f32(F4C(hv[b], tt), F4C(ev[a], tt), acc[b][a], 0, 0, 0);        \
  }
  LOAD_EH(e0, h0, 0);
  int ks = 0;
  for (; ks + 2 <= KS; ks += 2) {
    LOAD_EH(e1, h1, ks + 1);
    __builtin_amdgcn_sched_barrier(0);
    MFMA_EH(e0, h0);
    const int kn = ks + 2 < KS ? ks + 2 : KS - 1;
    LOAD_EH(e0, h0, kn);
    __builtin_amdgcn_sched_barrier(0);
    MFMA_EH(e1, h1);
  }
  if (ks < KS) MFMA_EH(e0, h0);
#undef LOAD_EH
#undef MFMA_EH
#pragma unroll
  for (int b = 0; b < NQ; ++b)
#pragma unroll
    for (int a = 0; a < ME; ++a) {
      int64_t e = (eb0 + a) * 32 + (lane & 31);
      if (e >= n_local) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t q = q0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (q < B) logits[q * ld + e] = acc[b][a][r];
      }
    }
}

int launch_score_all(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, hipStream_t s) {
  const Dims& dm = h->dm;
  constexpr int NQ = 2, ME = 2;
  int64_t q_tiles = (B + 32 * NQ - 1) / (32 * NQ);
  int64_t e_groups = (dm.n_eblk + 4 * ME - 1) / (4 * ME);
  if (e_groups > 65535) return fail(h, COPER_EUNSUPPORTED, "score_all: shard too large to materialise logits");
  int64_t q_tiles4 = (B + 32 * SC_NQ - 1) / (32 * SC_NQ);
  int64_t total = q_tiles4 * SC_NQ * dm.KS * 64;
  hipLaunchKernelGGL(k_pack_h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hvec, B, dm.d, dm.KS,
                     (float4*)h->hfrag_ws, total, (int32_t*)nullptr, (int32_t*)nullptr);
  ScopedKernelTimer t(h, "score_all", s);
  hipLaunchKernelGGL((k_score_all_f32<NQ, ME>), dim3((unsigned)q_tiles, (unsigned)e_groups), dim3(256), 0, s,
                     (const float4*)h->Ef, h->bias_pad, (const float4*)h->hfrag_ws, B, dm.KS, dm.n_eblk, dm.n_local,
                     logits, ld);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// VALU pair scores: the same chain on (query, entity) pairs picked by index.
// ------------------------------------------------------------------------------------------------
// The chain of the file header on one (entity row, query row) pair.  d % 8 == 0 (every BASELINE config):
// two 16-B loads per operand per k-step, issued a few k-steps ahead of the dependent fma chain.
__device__ __forceinline__ float chain_score(const float* __restrict__ erow, const float* __restrict__ hrow,
                                             float bias, int d) {
  float s = bias;
  if ((d & 7) == 0) {
    const float4* e4 = (const float4*)erow;
    const float4* h4 = (const float4*)hrow;
    const int KS = d >> 3;
#pragma unroll 5
    for (int ks = 0; ks < KS; ++ks) {
      float4 a0 = e4[2 * ks], a1 = e4[2 * ks + 1], b0 = h4[2 * ks], b1 = h4[2 * ks + 1];
      s = fmaf(a0.x, b0.x, s); s = fmaf(a1.x, b1.x, s);
      s = fmaf(a0.y, b0.y, s); s = fmaf(a1.y, b1.y, s);
      s = fmaf(a0.z, b0.z, s); s = fmaf(a1.z, b1.z, s);
      s = fmaf(a0.w, b0.w, s); s = fmaf(a1.w, b1.w, s);
    }
    return s;
  }
  int KS = (d + 7) >> 3;
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int k0 = 8 * ks + t, k1 = k0 + 4;
      if (k0 < d) s = fmaf(erow[k0], hrow[k0], s);
      if (k1 < d) s = fmaf(erow[k1], hrow[k1], s);
    }
  }
  return s;
}

__global__ void k_pair_targets(const float* __restrict__ ent, const float* __restrict__ bias,
                               const float* __restrict__ hvec, const int64_t* __restrict__ e2, int64_t B, int d,
                               int64_t lo, int64_t n_local, float* __restrict__ tgt) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t row = e2[b] - lo;
  tgt[b] = (row >= 0 && row < n_local) ? chain_score(ent + row * d, hvec + b * d, bias[row], d) : 0.f;
}

int launch_pair_targets(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt,
                        hipStream_t s) {
  const Dims& dm = h->dm;
  hipLaunchKernelGGL(k_pair_targets, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->params["ent_emb"].ptr,
                     h->params["pred_bias"].ptr, hvec, e2, B, dm.d, (int64_t)h->cfg.shard_lo, dm.n_local, tgt);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// 16 lanes per query walk its CSR filter list (coalesced id reads, no row search); lane 0 of the group
// also retires the target itself, which scored == tgt by construction and was counted as "equal".
constexpr int FC_GROUP_ENTRIES = 256;   // entries of a row its 16-lane group walks; the rest of a longer row: the whole workgroup

__global__ __launch_bounds__(256) void k_filter_correct(const float* __restrict__ ent, const float* __restrict__ bias,
                                 const float* __restrict__ hvec, const float* __restrict__ tgt,
                                 const int64_t* __restrict__ e2, const int64_t* __restrict__ indptr,
                                 const int64_t* __restrict__ idx, int64_t B, int d, int64_t lo, int64_t n_local,
                                 int32_t* __restrict__ ng, int32_t* __restrict__ ne) {
  __shared__ int s_long[16];       // queries of this workgroup (16 per workgroup) with more than FC_GROUP_ENTRIES entries
  __shared__ int s_nlong;
  __shared__ int s_dg, s_de;
  if (threadIdx.x == 0) s_nlong = 0;
  __syncthreads();
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = gid >> 4;
  int sub = (int)(gid & 15);
  // one entry of the filter list against the query's target: (greater, equal) contributions
  auto entry = [&](int64_t bq, int64_t i, int64_t beg, float t, int64_t target, int& dg, int& de) {
    int64_t f = idx[i];
    if (i > beg && idx[i - 1] == f) return;    // adjacent duplicate: the dense mask is idempotent
    if (f == target) return;                   // the target is restored after masking (metrics.py:46)
    int64_t row = f - lo;
    if (row < 0 || row >= n_local) return;
    float sc = chain_score(ent + row * d, hvec + bq * d, bias[row], d);
    dg += sc > t ? 1 : 0;
    de += sc == t ? 1 : 0;
  };
  if (b < B) {
    const float t = tgt[b];
    const int64_t target = e2[b];
    if (sub == 0) {
      int64_t row = target - lo;
      if (ne && row >= 0 && row < n_local && t == t) atomicSub(&ne[b], 1);
    }
    const int64_t beg = indptr[b], end = indptr[b + 1];
    const int64_t stop = end - beg > FC_GROUP_ENTRIES ? beg + FC_GROUP_ENTRIES : end;
    if (sub == 0 && end > stop) s_long[atomicAdd(&s_nlong, 1)] = (int)(b - (int64_t)blockIdx.x * 16);
    int dg = 0, de = 0;
    for (int64_t i = beg + sub; i < stop; i += 16) entry(b, i, beg, t, target, dg, de);
    // reduce over the 16-lane group
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      dg += __shfl_xor(dg, m);
      de += __shfl_xor(de, m);
    }
    if (sub == 0) {
      if (dg) atomicSub(&ng[b], dg);
      if (ne && de) atomicSub(&ne[b], de);
    }
  }
  __syncthreads();
  // rows with thousands of known answers (real KGs have them): all 256 threads take the rest of such a row together -- 16
  // lanes walking 5,000 entries one fma chain at a time held the launch back by hundreds of microseconds
  const int nlong = s_nlong;
  for (int r = 0; r < nlong; ++r) {
    if (threadIdx.x == 0) { s_dg = 0; s_de = 0; }
    __syncthreads();
    const int64_t bq = (int64_t)blockIdx.x * 16 + s_long[r];
    const float t = tgt[bq];
    const int64_t target = e2[bq];
    const int64_t beg = indptr[bq], end = indptr[bq + 1];
    int dg = 0, de = 0;
    for (int64_t i = beg + FC_GROUP_ENTRIES + threadIdx.x; i < end; i += 256) entry(bq, i, beg, t, target, dg, de);
    if (dg) atomicAdd(&s_dg, dg);
    if (de) atomicAdd(&s_de, de);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (s_dg) atomicSub(&ng[bq], s_dg);
      if (ne && s_de) atomicSub(&ne[bq], s_de);
    }
    __syncthreads();
  }
}

int launch_filter_correct(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2,
                          const int64_t* indptr, const int64_t* idx, int64_t nnz, int64_t B, int32_t* ng,
                          int32_t* ne, hipStream_t s) {
  const Dims& dm = h->dm;
  (void)nnz;
  int64_t threads = B * 16;
  hipLaunchKernelGGL(k_filter_correct, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                     h->params["ent_emb"].ptr, h->params["pred_bias"].ptr, hvec, tgt, e2, indptr, idx, B, dm.d,
                     (int64_t)h->cfg.shard_lo, dm.n_local, ng, ne);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

__global__ void k_score_lookup(const float* __restrict__ ent, const float* __restrict__ bias,
                               const float* __restrict__ hvec, const int32_t* __restrict__ lookup, int64_t B,
                               int64_t L, int d, int64_t lo, int64_t n_local, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  int64_t b = i / L;
  int64_t row = (int64_t)lookup[i] - lo;
  out[i] = (row >= 0 && row < n_local) ? chain_score(ent + row * d, hvec + b * d, bias[row], d) : 0.f;
}

int launch_score_lookup(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L, float* out,
                        hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t total = B * L;
  hipLaunchKernelGGL(k_score_lookup, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     h->params["ent_emb"].ptr, h->params["pred_bias"].ptr, hvec, lookup, B, L, dm.d,
                     (int64_t)h->cfg.shard_lo, dm.n_local, out);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// top-k of the filtered row (optional output of coper_rank_counts, k > 0): logits of a chunk of queries
// are materialised by k_score_all_f32 into a workspace, known answers other than the target are set to
// -inf (metrics.py:45-46 in sparse form), then k rounds of a block-wide arg-max per row pick the
// candidates in (score desc, id asc) order.  Not on the ranking hot path: ranks never need it.
// ------------------------------------------------------------------------------------------------
__global__ void k_mask_filtered(float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ e2,
                                const int64_t* __restrict__ indptr, const int64_t* __restrict__ idx, int64_t b0,
                                int64_t nb, int64_t lo, int64_t n_local) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t bl = gid >> 4;
  int sub = (int)(gid & 15);
  if (bl >= nb) return;
  int64_t b = b0 + bl;
  const int64_t target = e2[b];
  for (int64_t i = indptr[b] + sub; i < indptr[b + 1]; i += 16) {
    int64_t f = idx[i];
    int64_t row = f - lo;
    if (f == target || row < 0 || row >= n_local) continue;
    logits[bl * ld + row] = -INFINITY;
  }
}

__global__ __launch_bounds__(256) void k_topk_select(float* __restrict__ logits, int64_t ld, int64_t n_local,
                                                     int64_t lo, int k, float* __restrict__ out_val,
                                                     int64_t* __restrict__ out_idx, int64_t b0) {
  __shared__ float s_val[256];
  __shared__ int s_idx[256];
  float* row = logits + (int64_t)blockIdx.x * ld;
  int64_t b = b0 + blockIdx.x;
  for (int round = 0; round < k; ++round) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t j = threadIdx.x; j < n_local; j += 256) {
      float v = row[j];
      if (v > best || (v == best && v > -INFINITY && (int)j < bi)) { best = v; bi = (int)j; }
    }
    s_val[threadIdx.x] = best;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) {
        float v2 = s_val[threadIdx.x + off];
        int i2 = s_idx[threadIdx.x + off];
        if (v2 > s_val[threadIdx.x] || (v2 == s_val[threadIdx.x] && i2 < s_idx[threadIdx.x])) {
          s_val[threadIdx.x] = v2;
          s_idx[threadIdx.x] = i2;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      float v = s_val[0];
      int i = s_idx[0];
      bool ok = i != 0x7fffffff && v > -INFINITY;
      out_val[b * k + round] = ok ? v : -INFINITY;
      out_idx[b * k + round] = ok ? lo + i : -1;
      if (ok) row[i] = -INFINITY;  // taken
    }
    __syncthreads();
  }
}


// Single-pass variant for k <= 32: every thread keeps the top-k of its strided share of the row in LDS (sorted,
// score descending then index ascending), then k rounds of a 256-way arg-max over the heads of those lists pick the
// row's top-k in order -- the row is read once instead of k times.  Same results as k_topk_select (ties by index).
constexpr int TOPK_SMALL_MAX = 32;
__global__ __launch_bounds__(256) void k_topk_select_small(const float* __restrict__ logits, int64_t ld, int64_t n_local,
                                                           int64_t lo, int k, float* __restrict__ out_val,
                                                           int64_t* __restrict__ out_idx, int64_t b0) {
  extern __shared__ float tk_lds[];          // val[k][256] | idx[k][256] | red_val[256] | red_idx[256]
  float* lv = tk_lds;
  int* li = (int*)(tk_lds + k * 256);
  float* rv = tk_lds + 2 * k * 256;
  int* ri = (int*)(rv + 256);
  const int tid = threadIdx.x;
  const float* row = logits + (int64_t)blockIdx.x * ld;
  const int64_t b = b0 + blockIdx.x;
  for (int s = 0; s < k; ++s) { lv[s * 256 + tid] = -INFINITY; li[s * 256 + tid] = 0x7fffffff; }
  float worst = -INFINITY;                   // value of the thread's k-th entry
  int worst_i = 0x7fffffff;
  for (int64_t j0 = tid; j0 < n_local; j0 += 256 * 8) {
    float vb[8];   // eight independent loads in flight, then the (data-dependent) insertions
#pragma unroll
    for (int u = 0; u < 8; ++u) vb[u] = j0 + 256 * u < n_local ? row[j0 + 256 * u] : -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
    const float v = vb[u];
    const int64_t j = j0 + 256 * u;
    if (!(v > -INFINITY)) continue;
    if (v < worst || (v == worst && (int)j > worst_i)) continue;
    // insert (v, j) into the sorted list
    int pos = k - 1;
    while (pos > 0) {
      const float pv = lv[(pos - 1) * 256 + tid];
      const int pi = li[(pos - 1) * 256 + tid];
      if (pv > v || (pv == v && pi < (int)j)) break;
      lv[pos * 256 + tid] = pv;
      li[pos * 256 + tid] = pi;
      --pos;
    }
    lv[pos * 256 + tid] = v;
    li[pos * 256 + tid] = (int)j;
    worst = lv[(k - 1) * 256 + tid];
    worst_i = li[(k - 1) * 256 + tid];
    }
  }
  int head = 0;
  for (int round = 0; round < k; ++round) {
    rv[tid] = head < k ? lv[head * 256 + tid] : -INFINITY;
    ri[tid] = head < k ? li[head * 256 + tid] : 0x7fffffff;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if (tid < off) {
        const float v2 = rv[tid + off];
        const int i2 = ri[tid + off];
        if (v2 > rv[tid] || (v2 == rv[tid] && i2 < ri[tid])) { rv[tid] = v2; ri[tid] = i2; }
      }
      __syncthreads();
    }
    const float bv = rv[0];
    const int bi = ri[0];
    const bool ok = bi != 0x7fffffff && bv > -INFINITY;
    if (tid == 0) {
      out_val[b * k + round] = ok ? bv : -INFINITY;
      out_idx[b * k + round] = ok ? lo + bi : -1;
    }
    if (ok && head < k && li[head * 256 + tid] == bi) ++head;   // the owner of the winner advances
    __syncthreads();
  }
}

int launch_topk(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* indptr, const int64_t* idx,
                int64_t B, int k, float* topk_val, int64_t* topk_idx, float* logits_ws, int64_t chunk_rows,
                hipStream_t s) {
  const Dims& dm = h->dm;
  for (int64_t b0 = 0; b0 < B; b0 += chunk_rows) {
    int64_t nb = B - b0 < chunk_rows ? B - b0 : chunk_rows;
    int rc = score_all_dispatch(h, hvec + b0 * dm.d, nb, logits_ws, dm.n_local, s);
    if (rc) return rc;
    int64_t threads = nb * 16;
    hipLaunchKernelGGL(k_mask_filtered, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, logits_ws, dm.n_local,
                       e2, indptr, idx, b0, nb, (int64_t)h->cfg.shard_lo, dm.n_local);
    if (k <= TOPK_SMALL_MAX)
      hipLaunchKernelGGL(k_topk_select_small, dim3((unsigned)nb), dim3(256), sizeof(float) * (size_t)(2 * k + 2) * 256, s, logits_ws,
                         dm.n_local, dm.n_local, (int64_t)h->cfg.shard_lo, k, topk_val, topk_idx, b0);
    else
      hipLaunchKernelGGL(k_topk_select, dim3((unsigned)nb), dim3(256), 0, s, logits_ws, dm.n_local, dm.n_local,
                         (int64_t)h->cfg.shard_lo, k, topk_val, topk_idx, b0);
    COPER_HIP_TRY(h, hipGetLastError());
  }
  return COPER_OK;
}

__global__ void k_finish_ranks(const int32_t* __restrict__ ng, int64_t B, int32_t* __restrict__ ranks) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) ranks[b] = 1 + ng[b];
}

int launch_finish_ranks(coper_handle* h, const int32_t* ng, int64_t B, int32_t* ranks, hipStream_t s) {
  hipLaunchKernelGGL(k_finish_ranks, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, ng, B, ranks);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// pruned top-k, fp32-exact mode: the 32 logits of every candidate (query, block) slot by the VALU chain (the bits
// of the MFMA tiles), known answers except the target masked -- no grouping by block needed, a thread per logit
__global__ __launch_bounds__(256) void k_topk_score_blocks_f32(const float* __restrict__ ent, const float* __restrict__ bias,
                                                               const float* __restrict__ hvec, int d, int64_t T,
                                                               const int64_t* __restrict__ e2, const int64_t* __restrict__ indptr,
                                                               const int64_t* __restrict__ idx, const int32_t* __restrict__ cand_blk,
                                                               const int32_t* __restrict__ cand_q, int64_t lo, int64_t n_local,
                                                               float* __restrict__ cand_val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * 32) return;
  const int64_t w = i >> 5;
  const int32_t g = cand_blk[w];
  if (g < 0) { cand_val[i] = -INFINITY; return; }
  const int64_t q = cand_q[w];
  const int64_t row = (int64_t)g * 32 + (i & 31);
  float v = -INFINITY;
  if (row < n_local) {
    const int64_t ent_id = row + lo;
    bool known = false;
    if (ent_id != e2[q])
      for (int64_t j = indptr[q]; j < indptr[q + 1]; ++j) known |= idx[j] == ent_id;
    if (!known) v = chain_score(ent + row * d, hvec + q * d, bias[row], d);
  }
  cand_val[i] = v;
}

int launch_topk_score_blocks_f32(coper_handle* h, const float* hvec, int64_t T, const int64_t* e2, const int64_t* indptr,
                                 const int64_t* idx, hipStream_t s) {
  const Dims& dm = h->dm;
  hipLaunchKernelGGL(k_topk_score_blocks_f32, dim3((unsigned)((T * 32 + 255) / 256)), dim3(256), 0, s, h->params["ent_emb"].ptr,
                     h->params["pred_bias"].ptr, hvec, dm.d, T, e2, indptr, idx, h->cand_blk_ws, h->cand_q_ws, (int64_t)h->cfg.shard_lo,
                     dm.n_local, h->cand_val_ws);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
