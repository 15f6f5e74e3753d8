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
// fused score + count.  grid = (query tiles of 32*NQ, entity splits); 4 waves, each wave ME entity
// blocks per iteration.  D[i][j]: i = entity row (A operand), j = query (B operand, one per lane).
// ------------------------------------------------------------------------------------------------
template <int NQ, int ME>
__global__ __launch_bounds__(256, 2) void k_score_count_f32(const float4* __restrict__ Ef,
                                                            const float* __restrict__ bias_pad,
                                                            const float* __restrict__ hvec,
                                                            const float* __restrict__ tgt, int64_t B, int d, int KS,
                                                            int64_t n_eblk, int64_t eblk_per_split,
                                                            int32_t* __restrict__ ng, int32_t* __restrict__ ne) {
  extern __shared__ float4 hl[];
  const int64_t q0 = (int64_t)blockIdx.x * (32 * NQ);
  stage_h_frag<NQ>(hl, hvec, q0, B, d, KS);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float t[NQ];
  int cg[NQ], ce[NQ];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    int64_t q = q0 + b * 32 + (lane & 31);
    t[b] = q < B ? tgt[q] : INFINITY;
    cg[b] = 0;
    ce[b] = 0;
  }
  const int64_t e_begin = (int64_t)blockIdx.y * eblk_per_split;
  int64_t e_end = e_begin + eblk_per_split;
  if (e_end > n_eblk) e_end = n_eblk;
  for (int64_t eb = e_begin + wave * ME; eb < e_end; eb += 4 * ME) {
    f32x16 acc[ME][NQ];
#pragma unroll
    for (int a = 0; a < ME; ++a) {
      // accumulator row of reg r: (r&3) + 8(r>>2) + 4(lane>>5): start the chain from pred_bias
      const float4* bp = (const float4*)(bias_pad + (eb + a) * 32 + 4 * (lane >> 5));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 b4 = bp[2 * j];
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          acc[a][b][4 * j + 0] = b4.x; acc[a][b][4 * j + 1] = b4.y;
          acc[a][b][4 * j + 2] = b4.z; acc[a][b][4 * j + 3] = b4.w;
        }
      }
    }
    const float4* ep = Ef + (eb * KS) * 64 + lane;
    for (int ks = 0; ks < KS; ++ks) {
      float4 av[ME], bv[NQ];
#pragma unroll
      for (int a = 0; a < ME; ++a) av[a] = ep[((int64_t)a * KS + ks) * 64];
#pragma unroll
      for (int b = 0; b < NQ; ++b) bv[b] = hl[(b * KS + ks) * 64 + lane];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int a = 0; a < ME; ++a)
#pragma unroll
          for (int b = 0; b < NQ; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(F4C(av[a], tt), F4C(bv[b], tt), acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < ME; ++a)
#pragma unroll
      for (int b = 0; b < NQ; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s = acc[a][b][r];
          cg[b] += (s > t[b]) ? 1 : 0;
          ce[b] += (s == t[b]) ? 1 : 0;
        }
  }
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    cg[b] += __shfl_xor(cg[b], 32);
    ce[b] += __shfl_xor(ce[b], 32);
    int64_t q = q0 + b * 32 + (lane & 31);
    if (lane < 32 && q < B) {
      if (cg[b]) atomicAdd(&ng[q], cg[b]);
      if (ce[b]) atomicAdd(&ne[q], ce[b]);
    }
  }
}

int launch_score_count(coper_handle* h, const float* hvec, const float* tgt, int64_t B, int32_t* ng, int32_t* ne,
                       hipStream_t s) {
  const Dims& dm = h->dm;
  constexpr int NQ = 2, ME = 2;
  int64_t q_tiles = (B + 32 * NQ - 1) / (32 * NQ);
  int64_t iters = dm.n_eblk / (4 * ME);  // n_eblk is padded to EBLK_ALIGN = 4*ME
  int64_t splits = (2048 + q_tiles - 1) / q_tiles;
  if (splits > iters) splits = iters;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  int64_t iters_per_split = (iters + splits - 1) / splits;
  splits = (iters + iters_per_split - 1) / iters_per_split;
  size_t lds = (size_t)NQ * dm.KS * 64 * sizeof(float4);
  ScopedKernelTimer t(h, "score_count", s);
  hipLaunchKernelGGL((k_score_count_f32<NQ, ME>), dim3((unsigned)q_tiles, (unsigned)splits), dim3(256), lds, s,
                     (const float4*)h->Ef, h->bias_pad, hvec, tgt, B, dm.d, dm.KS, dm.n_eblk,
                     iters_per_split * 4 * ME, ng, ne);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// logits out.  D[i][j]: i = query (A operand from LDS), j = entity (B operand, one entity per lane):
// a register row is 32 consecutive entities of one query -> 128-B contiguous stores.
// ------------------------------------------------------------------------------------------------
template <int NQ, int ME>
__global__ __launch_bounds__(256, 2) void k_score_all_f32(const float4* __restrict__ Ef,
                                                          const float* __restrict__ bias_pad,
                                                          const float* __restrict__ hvec, int64_t B, int d, int KS,
                                                          int64_t n_eblk, int64_t n_local,
                                                          float* __restrict__ logits, int64_t ld) {
  extern __shared__ float4 hl[];
  const int64_t q0 = (int64_t)blockIdx.x * (32 * NQ);
  stage_h_frag<NQ>(hl, hvec, q0, B, d, KS);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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
  for (int ks = 0; ks < KS; ++ks) {
    float4 ev[ME], hv[NQ];
#pragma unroll
    for (int a = 0; a < ME; ++a) ev[a] = ep[((int64_t)a * KS + ks) * 64];
#pragma unroll
    for (int b = 0; b < NQ; ++b) hv[b] = hl[(b * KS + ks) * 64 + lane];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int b = 0; b < NQ; ++b)
#pragma unroll
        for (int a = 0; a < ME; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(F4C(hv[b], tt), F4C(ev[a], tt), acc[b][a], 0, 0, 0);
  }
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
  size_t lds = (size_t)NQ * dm.KS * 64 * sizeof(float4);
  ScopedKernelTimer t(h, "score_all", s);
  hipLaunchKernelGGL((k_score_all_f32<NQ, ME>), dim3((unsigned)q_tiles, (unsigned)e_groups), dim3(256), lds, s,
                     (const float4*)h->Ef, h->bias_pad, hvec, B, dm.d, dm.KS, dm.n_eblk, dm.n_local, logits, ld);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// VALU pair scores: the same chain on (query, entity) pairs picked by index.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float chain_score(const float* __restrict__ erow, const float* __restrict__ hrow,
                                             float bias, int d) {
  float s = bias;
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

// thread i < nnz: filter entry i;  thread nnz + b: the target of query b.
__global__ void k_filter_correct(const float* __restrict__ ent, const float* __restrict__ bias,
                                 const float* __restrict__ hvec, const float* __restrict__ tgt,
                                 const int64_t* __restrict__ e2, const int64_t* __restrict__ indptr,
                                 const int64_t* __restrict__ idx, int64_t B, int d, int64_t lo, int64_t n_local,
                                 int32_t* __restrict__ ng, int32_t* __restrict__ ne) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t nnz = indptr[B];
  if (i >= nnz + B) return;
  if (i >= nnz) {  // the target scores == tgt by construction and was counted as "equal"
    int64_t b = i - nnz;
    int64_t row = e2[b] - lo;
    float t = tgt[b];
    if (row >= 0 && row < n_local && t == t) atomicSub(&ne[b], 1);
    return;
  }
  // largest b with indptr[b] <= i
  int64_t lo_b = 0, hi_b = B;
  while (hi_b - lo_b > 1) {
    int64_t mid = (lo_b + hi_b) >> 1;
    if (indptr[mid] <= i) lo_b = mid; else hi_b = mid;
  }
  int64_t b = lo_b;
  int64_t f = idx[i];
  if (i > indptr[b] && idx[i - 1] == f) return;  // adjacent duplicate: the dense mask is idempotent
  if (f == e2[b]) return;                        // the target is restored after masking (metrics.py:46)
  int64_t row = f - lo;
  if (row < 0 || row >= n_local) return;
  float s = chain_score(ent + row * d, hvec + b * d, bias[row], d);
  float t = tgt[b];
  if (s > t) atomicSub(&ng[b], 1);
  else if (s == t) atomicSub(&ne[b], 1);
}

int launch_filter_correct(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2,
                          const int64_t* indptr, const int64_t* idx, int64_t nnz, int64_t B, int32_t* ng,
                          int32_t* ne, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t bound = nnz + B;  // the kernel re-reads nnz = indptr[B] on the device and trusts that
  hipLaunchKernelGGL(k_filter_correct, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, s,
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

__global__ void k_finish_ranks(const int32_t* __restrict__ ng, int64_t B, int32_t* __restrict__ ranks) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) ranks[b] = 1 + ng[b];
}

int launch_finish_ranks(coper_handle* h, const int32_t* ng, int64_t B, int32_t* ranks, hipStream_t s) {
  hipLaunchKernelGGL(k_finish_ranks, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, ng, B, ranks);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
