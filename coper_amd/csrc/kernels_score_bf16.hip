// COPER_SCORE_BF16X3: the 1-vs-all scorer / ranker on the bf16 matrix cores at fp32-class accuracy.
//
// gfx950 has no TF32/xf32 path and its exact-f32 MFMA runs at 1/16 of the bf16 rate, so the fp32 mode
// of kernels_score.hip is MFMA-bound at 157 TFLOP/s.  Here every fp32 operand is split once into two
// bf16 terms,  x = hi + lo  (hi = rne_bf16(x), lo = rne_bf16(x - hi), |x - hi - lo| <= 2^-17 |x|), and
// each product is formed as  lo*hi + hi*lo + hi*hi  on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (the dropped lo*lo term is <= 2^-16 relative): 3 MFMAs at the bf16 rate instead of 1 at the f32 rate,
// 16/3 = 5.3x the matrix throughput, logit error ~1e-5 -- inside the 1e-3 gate of BASELINE.json.
// Storage is two bf16 planes (same bytes as fp32).
//
// Every logit of this mode -- tiles of score_count / score_all and the (query, entity) pair kernel used
// for targets, filter correction and the sampled scorer -- is produced by the SAME instruction sequence
// with the SAME operand roles (entity rows = A, queries = B), accumulators started from pred_bias, so
// they agree bit for bit (tests/test_gpu_parity.py checks it); an MFMA's result for element (i, j) does
// not depend on where the row / column sits in the tile.
#include "coper_internal.h"

namespace coper {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
  // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  unsigned hw[4], lw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf16x2_t hp = {(__bf16)v[2 * j], (__bf16)v[2 * j + 1]};
    float r0 = v[2 * j] - (float)hp[0], r1 = v[2 * j + 1] - (float)hp[1];
    bf16x2_t lp = {(__bf16)r0, (__bf16)r1};
    hw[j] = __builtin_bit_cast(unsigned, hp);
    lw[j] = __builtin_bit_cast(unsigned, lp);
  }
  hi = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  lo = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&(a), *(const bf16x8*)&(b), (c), 0, 0, 0)
// one k-step (16 k) of the split product, smallest terms first
#define MFMA_X3(ahi, alo, bhi, blo, c) \
  { (c) = MFMA_BF16(alo, bhi, c); (c) = MFMA_BF16(ahi, blo, c); (c) = MFMA_BF16(ahi, bhi, c); }

// ------------------------------------------------------------------------------------------------
// prepare: fragment image  X16[(blk*KS16 + ks)*64 + l] = 8 bf16 { X[32blk + (l&31)][16ks + 8(l>>5) + j] }
// for the hi and the lo plane (the A/B operand map of v_mfma_f32_32x32x16_bf16).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_to_frag_bf16(const float* __restrict__ src, int64_t n_rows, int d,
                                                           int KS16, uint4* __restrict__ hi, uint4* __restrict__ lo,
                                                           uint4* __restrict__ rm_hi, uint4* __restrict__ rm_lo,
                                                           int64_t total, int32_t* __restrict__ cnt, int32_t cnt_base,
                                                           int32_t* __restrict__ cnt_eq) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (blk*KS16 + ks)*64 + l
  // optional: preset the rank counters of the pass this packing opens (saves a launch on the ranking path)
  if (cnt && j < n_rows) { cnt[j] = cnt_base; if (cnt_eq) cnt_eq[j] = 0; }
  if (j >= total) return;
  int l = (int)(j & 63);
  int64_t rest = j >> 6;
  int ks = (int)(rest % KS16);
  int64_t blk = rest / KS16;
  int64_t row = blk * 32 + (l & 31);
  int k = 16 * ks + 8 * (l >> 5);
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = (row < n_rows && k + t < d) ? src[row * d + k + t] : 0.f;
  uint4 h4, l4;
  split8(v, h4, l4);
  hi[j] = h4;
  lo[j] = l4;
  if (rm_hi) {  // row-major twin of the planes: row r = 2*KS16 pieces of 16 B (the pair kernel gathers whole rows)
    int64_t o = row * (2 * KS16) + 2 * ks + (l >> 5);
    rm_hi[o] = h4;
    rm_lo[o] = l4;
  }
}

int launch_rows_to_frag_bf16(coper_handle* h, const float* src, int64_t n_rows, int64_t n_blk, uint4* hi, uint4* lo,
                             uint4* rm_hi, uint4* rm_lo, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t total = n_blk * dm.KS16 * 64;
  hipLaunchKernelGGL(k_rows_to_frag_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n_rows, dm.d,
                     dm.KS16, hi, lo, rm_hi, rm_lo, total, h->preset_cnt, h->count_base, h->preset_eq);
  if (h->preset_cnt) h->counts_preset = h->preset_cnt;
  h->preset_cnt = nullptr;
  h->preset_eq = nullptr;
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// fused score + count: same persistent, statically balanced structure as k_score_count_f32
// (kernels_score.hip): unit = 128 queries x 256 entities, 8 waves x one 32-row entity block.
// ------------------------------------------------------------------------------------------------
#ifndef COPER_BX_NQ
#define COPER_BX_NQ 4
#endif
#ifndef COPER_BX_WGS_PER_CU
#define COPER_BX_WGS_PER_CU 1
#endif
constexpr int BX_NQ = COPER_BX_NQ;
#ifndef COPER_BX_WAVES
#define COPER_BX_WAVES 8
#endif
constexpr int BX_WAVES = COPER_BX_WAVES;          // waves per workgroup (8: two per SIMD; 16: four per SIMD, needs ME = 1)
constexpr int BX_THREADS = 64 * BX_WAVES;
#ifndef COPER_BX_ME
#define COPER_BX_ME 2
#endif
#ifndef COPER_BX_PD
#define COPER_BX_PD 3
#endif
constexpr int BX_ME = COPER_BX_ME;
#ifdef COPER_BX_NT
typedef unsigned bx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 bx_nt_load(const uint4* p) {
  bx_u32x4 v = __builtin_nontemporal_load((const bx_u32x4*)p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
#define BX_LOADQ(p) bx_nt_load(p)
#else
#define BX_LOADQ(p) (*(p))
#endif
// ablation switches (tools/ab_build.py): fixed addresses instead of the streams, no compare epilogue
#ifdef COPER_DBG_BX_NO_GLOADS
#define BX_DBG_GL(x) (((x)*0) + m)
#else
#define BX_DBG_GL(x) (x)
#endif
#ifdef COPER_DBG_BX_NO_LDS
#define BX_DBG_LDS(x) (((x)*0) + b)
#else
#define BX_DBG_LDS(x) (x)
#endif
static_assert(BX_WAVES * BX_ME <= EBLK_ALIGN, "entity blocks are padded to EBLK_ALIGN");

__global__ void k_zero_counts(int64_t B, int32_t* __restrict__ ng, int32_t* __restrict__ ne, int32_t base) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < B) { ng[j] = base; if (ne) ne[j] = 0; }
}

// GM: also write, per (32-entity block, query), the largest logit of the block -- what the pruned top-k
// (kernels_topk_bf16.hip) selects its candidate blocks from; gmax[block * gm_stride + query].
template <bool EQ, bool GM>
__global__ __launch_bounds__(BX_THREADS, BX_WAVES / 4) void k_score_count_bf16x3(const uint4* __restrict__ Ehi,
                                                               const uint4* __restrict__ Elo,
                                                               const float* __restrict__ bias_pad,
                                                               const uint4* __restrict__ Hhi,
                                                               const uint4* __restrict__ Hlo,
                                                               const float* __restrict__ tgt, int64_t B, int KS,
                                                               int64_t iters, int64_t units,
                                                               int32_t* __restrict__ ng, int32_t* __restrict__ ne,
                                                               float* __restrict__ gmax, int64_t gm_stride) {
  constexpr int NQ = BX_NQ;
  extern __shared__ uint4 hl16[];  // [2 planes][NQ][KS][64]
  uint4* hl_hi = hl16;
  uint4* hl_lo = hl16 + NQ * KS * 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t u_begin = units * blockIdx.x / gridDim.x;
  const int64_t u_end = units * (blockIdx.x + 1) / gridDim.x;
  constexpr int ME = BX_ME;   // entity blocks per wave per unit: one LDS read of a query fragment feeds 3*ME MFMAs
  float t[NQ];
  int cg[NQ], ce[NQ];
  int64_t cur_tile = -1;
  float4 bq[ME][4];

#define LOAD_A_R(ah, al, ebx, ks_)                                \
  {                                                               \
    _Pragma("unroll") for (int m = 0; m < ME; ++m) {              \
      int64_t o_ = BX_DBG_GL(((ebx) + m) * KS + (ks_)) * 64 + lane; \
      ah[m] = BX_LOADQ(Ehi + o_);                                 \
      al[m] = BX_LOADQ(Elo + o_);                                 \
    }                                                             \
  }
#define LOAD_B_R(bh, bl, ks_)                                                 \
  {                                                                           \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) {                          \
      bh[b] = hl_hi[BX_DBG_LDS(b * KS + (ks_)) * 64 + lane];                  \
      bl[b] = hl_lo[BX_DBG_LDS(b * KS + (ks_)) * 64 + lane];                  \
    }                                                                         \
  }
  /* ablations: the loop without its global loads / without its LDS reads (operands loaded once, outside) */
#ifdef COPER_DBG_BX_SKIP_GL
#define LOAD_A(ah, al, ebx, ks_) {}
#else
#define LOAD_A(ah, al, ebx, ks_) LOAD_A_R(ah, al, ebx, ks_)
#endif
#ifdef COPER_DBG_BX_SKIP_LDS
#define LOAD_B(bh, bl, ks_) {}
#define BX_DECL_B
#else
#define LOAD_B(bh, bl, ks_) LOAD_B_R(bh, bl, ks_)
#define BX_DECL_B uint4 bh[NQ], bl[NQ];
#endif
#define LOAD_BIAS(ebx)                                                                \
  {                                                                                   \
    _Pragma("unroll") for (int m = 0; m < ME; ++m) {                                  \
      const float4* bp = (const float4*)(bias_pad + ((ebx) + m) * 32 + 4 * (lane >> 5)); \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) bq[m][j] = bp[2 * j];             \
    }                                                                                 \
  }
#ifdef COPER_DBG_BX_NO_MFMA   /* ablation: the fragment stream without the matrix work (loads stay live) */
#define STEP(ah, al, bh, bl)                                                  \
  {                                                                           \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b)                            \
      _Pragma("unroll") for (int m = 0; m < ME; ++m)                          \
        acc[m][b][0] += __uint_as_float((ah[m].x ^ al[m].y ^ ah[m].z ^ al[m].w ^ bh[b].x ^ bl[b].y) & 0x3fffffffu); \
  }
#else
#define STEP(ah, al, bh, bl)                                                  \
  {                                                                           \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b)                            \
      _Pragma("unroll") for (int m = 0; m < ME; ++m) MFMA_X3(ah[m], al[m], bh[b], bl[b], acc[m][b]); \
  }
#endif
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

  // entity fragments are fetched PD = 3 k-steps ahead (a k-step is only 12*ME MFMAs here, shorter than an L2
  // round trip under load): four rotating register buffers
  constexpr int PD = COPER_BX_PD, NBF = PD + 1;   // prefetch distance in k-steps, rotating register buffers
  uint4 ah[NBF][ME], al[NBF][ME];
#define KCL(k_) ((k_) < KS ? (k_) : KS - 1)
#ifdef COPER_BX_STAGGER
  // the two waves of a SIMD (w, w + 4) would otherwise run their compare epilogues at the same time, matrix pipe idle:
  // start the second one late so that one wave's epilogue falls into the other's k-loop
  if (wave >= 4) __builtin_amdgcn_s_sleep(COPER_BX_STAGGER);
#endif
  if (u_begin < u_end) {
    int64_t eb = ((u_begin % iters) * BX_WAVES + wave) * ME;
#pragma unroll
    for (int i = 0; i < PD; ++i) LOAD_A(ah[i], al[i], eb, KCL(i));
#ifdef COPER_DBG_BX_SKIP_GL
#pragma unroll
    for (int i = 0; i < NBF; ++i) LOAD_A_R(ah[i], al[i], eb, KCL(i));
#endif
    LOAD_BIAS(eb);
  }
  for (int64_t u = u_begin; u < u_end; ++u) {
    const int64_t tile = u / iters;
    const int64_t eb = ((u % iters) * BX_WAVES + wave) * ME;
    if (tile != cur_tile) {  // workgroup-uniform
      if (cur_tile >= 0) FLUSH_COUNTS();
      __syncthreads();
      const uint4* sh = Hhi + tile * (NQ * KS * 64);
      const uint4* sl = Hlo + tile * (NQ * KS * 64);
      for (int j = threadIdx.x; j < NQ * KS * 64; j += BX_THREADS) { hl_hi[j] = sh[j]; hl_lo[j] = sl[j]; }
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
    f32x16 acc[ME][NQ];
#pragma unroll
    for (int m = 0; m < ME; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          acc[m][b][4 * j + 0] = bq[m][j].x; acc[m][b][4 * j + 1] = bq[m][j].y;
          acc[m][b][4 * j + 2] = bq[m][j].z; acc[m][b][4 * j + 3] = bq[m][j].w;
        }
    // No conditional code around a k-step or its prefetch in the main loop: hipcc's s_waitcnt insertion takes
    // the minimum over paths, and a branch there turns the 3-step-ahead prefetch into "everything but the
    // loads just issued must have landed" -- one step ahead.  Fetches past the last k-step re-read it (KCL),
    // the prefetch for the unit after the last one re-reads this unit's blocks.
    int ks = 0;
#ifdef COPER_DBG_BX_SKIP_LDS
    uint4 bh[NQ], bl[NQ];
    LOAD_B_R(bh, bl, 0);
#endif
    for (; ks + NBF <= KS; ks += NBF) {
#pragma unroll
      for (int j = 0; j < NBF; ++j) {
        LOAD_A(ah[(j + PD) % NBF], al[(j + PD) % NBF], eb, KCL(ks + j + PD));
        __builtin_amdgcn_sched_barrier(0);
        BX_DECL_B
        LOAD_B(bh, bl, ks + j);
        STEP(ah[j], al[j], bh, bl);
      }
    }
#pragma unroll
    for (int j = 0; j < NBF - 1; ++j) {
      if (ks + j < KS) {  // wave-uniform; KS % NBF trailing steps
        LOAD_A(ah[(j + PD) % NBF], al[(j + PD) % NBF], eb, KCL(ks + j + PD));
        __builtin_amdgcn_sched_barrier(0);
        BX_DECL_B
        LOAD_B(bh, bl, ks + j);
        STEP(ah[j], al[j], bh, bl);
      }
    }
    {
      const int64_t ebn = u + 1 < u_end ? (((u + 1) % iters) * BX_WAVES + wave) * ME : eb;
#pragma unroll
      for (int i = 0; i < PD; ++i) LOAD_A(ah[i], al[i], ebn, KCL(i));
      LOAD_BIAS(ebn);
    }
#ifdef COPER_DBG_BX_NO_EPILOGUE
#pragma unroll
    for (int m = 0; m < ME; ++m)
#pragma unroll
      for (int b = 0; b < NQ; ++b) cg[b] += (acc[m][b][0] + acc[m][b][5] + acc[m][b][10] + acc[m][b][15] > t[b]) ? 1 : 0;
#else
#pragma unroll
    for (int m = 0; m < ME; ++m)
#pragma unroll
      for (int b = 0; b < NQ; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sc = acc[m][b][r];
          cg[b] += (sc > t[b]) ? 1 : 0;
          if (EQ) ce[b] += (sc == t[b]) ? 1 : 0;
        }
#endif
    if (GM) {
#pragma unroll
      for (int m = 0; m < ME; ++m)
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
          float mx = acc[m][b][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[m][b][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32));   // the other 16 rows of the block
          if (lane < 32) gmax[(eb + m) * gm_stride + cur_tile * (32 * NQ) + b * 32 + lane] = mx;
        }
    }
  }
#undef KCL
  if (cur_tile >= 0) FLUSH_COUNTS();
#undef LOAD_A
#undef LOAD_B
#undef LOAD_A_R
#undef LOAD_B_R
#undef BX_DECL_B
#undef LOAD_BIAS
#undef STEP
#undef FLUSH_COUNTS
}

int launch_pack_h_bf16(coper_handle* h, const float* hvec, int64_t B, hipStream_t s) {
  int64_t n_blk = (B + 32 * BX_NQ - 1) / (32 * BX_NQ) * BX_NQ;
  return launch_rows_to_frag_bf16(h, hvec, B, n_blk, (uint4*)h->hfrag16_hi, (uint4*)h->hfrag16_lo, (uint4*)h->hrm16_hi,
                                  (uint4*)h->hrm16_lo, s);
}

// queries [q0, q0 + Bc) of the packed batch (q0 a multiple of the 128-query tile); gmax != NULL: block maxima too
int score_count_chunk_bf16x3(coper_handle* h, int64_t q0, int64_t Bc, const float* tgt, int32_t* ng, int32_t* ne, float* gmax,
                             int64_t gm_stride, hipStream_t s) {
  const Dims& dm = h->dm;
  if (score_count2_supported(h)) {   // the pipelined kernel (kernels_score2_bf16.hip): same counts, same block maxima
    ScopedKernelTimer t(h, "score_count", s);
    return score_count2_chunk_bf16x3(h, q0, Bc, tgt, ng, ne, gmax, gm_stride, s);
  }
  int64_t q_tiles = (Bc + 32 * BX_NQ - 1) / (32 * BX_NQ);
  int64_t iters = dm.n_eblk / (BX_WAVES * BX_ME);
  int64_t units = q_tiles * iters;
  int64_t grid = (int64_t)h->num_cus * COPER_BX_WGS_PER_CU;
  if (grid > units) grid = units;
  size_t lds = (size_t)2 * BX_NQ * dm.KS16 * 64 * sizeof(uint4);
  const uint4* hhi = (const uint4*)h->hfrag16_hi + (q0 / 32) * dm.KS16 * 64;
  const uint4* hlo = (const uint4*)h->hfrag16_lo + (q0 / 32) * dm.KS16 * 64;
  ScopedKernelTimer t(h, "score_count", s);
#define BX_LAUNCH(EQ_, GM_)                                                                                                       \
  hipLaunchKernelGGL((k_score_count_bf16x3<EQ_, GM_>), dim3((unsigned)grid), dim3(BX_THREADS), lds, s, (const uint4*)h->Ef16_hi,  \
                     (const uint4*)h->Ef16_lo, h->bias_pad, hhi, hlo, tgt + q0, Bc, dm.KS16, iters, units, ng + q0,                \
                     ne ? ne + q0 : nullptr, gmax, gm_stride)
  if (gmax) { if (ne) BX_LAUNCH(true, true); else BX_LAUNCH(false, true); }
  else      { if (ne) BX_LAUNCH(true, false); else BX_LAUNCH(false, false); }
#undef BX_LAUNCH
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// counters start from count_base unless the caller's pack launch preset them (coper_rank: n_greater accumulates
// straight into `ranks`, started from 1)
void score_count_begin_bf16x3(coper_handle* h, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s) {
  if (h->counts_preset != ng)
    hipLaunchKernelGGL(k_zero_counts, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, ng, ne, h->count_base);
  h->counts_preset = nullptr;
}

int launch_score_count_bf16x3(coper_handle* h, const float* hvec, const float* tgt, int64_t B, int32_t* ng,
                              int32_t* ne, hipStream_t s) {
  (void)hvec;  // already packed by launch_pack_h_bf16 (coper_rank_counts packs once per call)
  score_count_begin_bf16x3(h, B, ng, ne, s);
  return score_count_chunk_bf16x3(h, 0, B, tgt, ng, ne, nullptr, 0, s);
}

int score_bf16_kernels_init(coper_handle* h) {
  int lds = (int)((size_t)2 * BX_NQ * h->dm.KS16 * 64 * sizeof(uint4));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_bf16x3<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_bf16x3<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_bf16x3<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count_bf16x3<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return COPER_OK;
}

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
                                                             float* __restrict__ logits, int64_t ld) {
  constexpr int NQ = 2, ME = 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t qblk0 = (int64_t)blockIdx.x * NQ;  // 32-query blocks (hfrag is packed per 32-query block)
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
  uint4 eh0[ME], el0[ME], qh0[NQ], ql0[NQ], eh1[ME], el1[ME], qh1[NQ], ql1[NQ];
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
  // same term order as MFMA_X3(e_hi, e_lo, q_hi, q_lo): e_lo*q_hi, e_hi*q_lo, e_hi*q_hi -- with the query as A
#define STEP_EQ(eh, el, qh, ql)                                                                      \
  {                                                                                                  \
    _Pragma("unroll") for (int b = 0; b < NQ; ++b) _Pragma("unroll") for (int a = 0; a < ME; ++a) {  \
      acc[b][a] = MFMA_BF16(qh[b], el[a], acc[b][a]);                                                \
      acc[b][a] = MFMA_BF16(ql[b], eh[a], acc[b][a]);                                                \
      acc[b][a] = MFMA_BF16(qh[b], eh[a], acc[b][a]);                                                \
    }                                                                                                \
  }
  LOAD_EQ(eh0, el0, qh0, ql0, 0);
  int ks = 0;
  for (; ks + 2 <= KS; ks += 2) {
    LOAD_EQ(eh1, el1, qh1, ql1, ks + 1);
    __builtin_amdgcn_sched_barrier(0);
    STEP_EQ(eh0, el0, qh0, ql0);
    const int kn = ks + 2 < KS ? ks + 2 : KS - 1;
    LOAD_EQ(eh0, el0, qh0, ql0, kn);
    __builtin_amdgcn_sched_barrier(0);
    STEP_EQ(eh1, el1, qh1, ql1);
  }
  if (ks < KS) STEP_EQ(eh0, el0, qh0, ql0);
#undef LOAD_EQ
#undef STEP_EQ
#pragma unroll
  for (int b = 0; b < NQ; ++b)
#pragma unroll
    for (int a = 0; a < ME; ++a) {
      int64_t e = (eb0 + a) * 32 + (lane & 31);
      if (e >= n_local) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t q = (qblk0 + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (q < B) logits[q * ld + e] = acc[b][a][r];
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
                     (const uint4*)h->hfrag16_lo, B, dm.KS16, dm.n_eblk, dm.n_local, logits, ld);
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
                                                     const int32_t* __restrict__ row_of, const float* __restrict__ tgt,
                                                     int64_t lo, int64_t n_local, float* __restrict__ out,
                                                     int32_t* __restrict__ ng, int32_t* __restrict__ ne) {
  __shared__ int64_t s_e[4][32];
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
    acc[r] = er >= 0 ? bias_pad[er] : 0.f;
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
  for (int ks = 0; ks < KS; ks += PB) {
    uint4 ah[PB], al[PB], bh[PB], bl[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int k = ks + u < KS ? ks + u : KS - 1;
      ah[u] = pa_h[k * 2]; al[u] = pa_l[k * 2];
      bh[u] = pb_h[k * 2]; bl[u] = pb_l[k * 2];
    }
#pragma unroll
    for (int u = 0; u < PB; ++u)
      if (ks + u < KS) MFMA_X3(ah[u], al[u], bh[u], bl[u], acc);   // wave-uniform
  }
  // D[i][i] sits in lane i + 32*((i>>2)&1), register (i&3) + 4*(i>>3)
  const bool diag_lane = ((i >> 2) & 1) == half && p < n_pairs;
  const int reg = (i & 3) + 4 * (i >> 3);
  float sc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sc = (r == reg) ? acc[r] : sc;
  if (mode != 2) {
    if (diag_lane) out[p] = erow >= 0 ? sc : 0.f;
    return;
  }
  // filter correction: consecutive entries belong to the same query (CSR order), so the wave adds up each run of equal
  // query ids and issues ONE atomic per run and counter -- a row with 5,000 known answers was 5,000 atomics on one address
  // (~100 us), now 157.  Every lane runs the cross-lane steps; lanes 0..31 stand for entries 0..31.
  const bool valid = diag_lane && erow >= 0;
  const float t = valid ? tgt[q] : 0.f;
  const int src = i + 32 * ((i >> 2) & 1);                       // the lane that holds entry i's diagonal value
  const int hit_g = __shfl((valid && sc > t) ? 1 : 0, src);
  const int hit_e = __shfl((valid && sc == t) ? 1 : 0, src);
  const int64_t q_prev = __shfl_up(q, 1);
  const bool in_tile = p < n_pairs && p < indptr[B];
  const bool head = half == 0 && in_tile && (i == 0 || q_prev != q);
  const unsigned heads = (unsigned)(__ballot(head) & 0xFFFFFFFFull);
  const unsigned m_g = (unsigned)(__ballot(half == 0 && hit_g) & 0xFFFFFFFFull);
  const unsigned m_e = (unsigned)(__ballot(half == 0 && hit_e) & 0xFFFFFFFFull);
  if (head) {
    const unsigned later = i < 31 ? (heads >> (i + 1)) : 0u;
    const int end = later ? i + 1 + __builtin_ctz(later) : 32;   // entries [i, end) share this lane's query
    const unsigned run = (end >= 32 ? 0xFFFFFFFFu : ((1u << end) - 1u)) & ~((1u << i) - 1u);
    const int cg = __builtin_popcount(m_g & run), ce = __builtin_popcount(m_e & run);
    if (cg) atomicSub(&ng[q], cg);
    if (ne && ce) atomicSub(&ne[q], ce);
  }
}

// CSR -> row id per entry, and the retirement of the target itself (scored == tgt, counted as "equal")
__global__ void k_expand_rows_retire_target(const int64_t* __restrict__ indptr, int64_t B, const int64_t* __restrict__ e2,
                                            const float* __restrict__ tgt, int64_t lo, int64_t n_local,
                                            int32_t* __restrict__ row_of, int32_t* __restrict__ ne) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = gid >> 4;
  int sub = (int)(gid & 15);
  if (b >= B) return;
  for (int64_t i = indptr[b] + sub; i < indptr[b + 1]; i += 16) row_of[i] = (int32_t)b;
  if (sub == 0) {
    int64_t row = e2[b] - lo;
    float t = tgt[b];
    if (ne && row >= 0 && row < n_local && t == t) atomicSub(&ne[b], 1);
  }
}

static void pair_launch(coper_handle* h, int mode, int64_t n_pairs, int64_t B, int64_t L, const int64_t* e2,
                        const int32_t* lookup, const int64_t* indptr, const int64_t* idx, const int32_t* row_of,
                        const float* tgt, float* out, int32_t* ng, int32_t* ne, hipStream_t s) {
  const Dims& dm = h->dm;
  if (n_pairs <= 0) return;
  // few pairs (the target pass: one per query): one wave per workgroup spreads them over all CUs
  const int wpb = n_pairs <= (1 << 22) ? 1 : 4;
  hipLaunchKernelGGL(k_pair_bf16x3, dim3((unsigned)((n_pairs + 32 * wpb - 1) / (32 * wpb))), dim3(64 * wpb), 0, s, (const uint4*)h->Erm16_hi,
                     (const uint4*)h->Erm16_lo, h->bias_pad, (const uint4*)h->hrm16_hi, (const uint4*)h->hrm16_lo,
                     dm.KS16, mode, n_pairs, B, L, e2, lookup, indptr, idx, row_of, tgt, (int64_t)h->cfg.shard_lo,
                     dm.n_local, out, ng, ne);
}

int launch_pair_targets_bf16x3(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt,
                               hipStream_t s) {
  int rc = launch_pack_h_bf16(h, hvec, B, s);
  if (rc) return rc;
  h->packed_hvec = hvec;  // coper_rank reuses this packing for the count pass on the same stream
  h->packed_B = B;
  pair_launch(h, 0, B, B, 1, e2, nullptr, h->expand_indptr, nullptr, h->expand_indptr ? h->row_of_ws : nullptr, nullptr, tgt, nullptr,
              nullptr, s);
  if (h->expand_indptr) h->rows_expanded_for = h->expand_indptr;
  h->expand_indptr = nullptr;
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// targets of queries whose planes are already in place (coper_encode_rank: written by k_dense_finalize_pack)
int launch_pair_targets_packed_bf16x3(coper_handle* h, const int64_t* e2, int64_t B, float* tgt, hipStream_t s) {
  pair_launch(h, 0, B, B, 1, e2, nullptr, h->expand_indptr, nullptr, h->expand_indptr ? h->row_of_ws : nullptr, nullptr, tgt, nullptr,
              nullptr, s);
  if (h->expand_indptr) h->rows_expanded_for = h->expand_indptr;
  h->expand_indptr = nullptr;
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_score_lookup_bf16x3(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L,
                               float* out, hipStream_t s) {
  int rc = launch_pack_h_bf16(h, hvec, B, s);
  if (rc) return rc;
  pair_launch(h, 1, B * L, B, L, nullptr, lookup, nullptr, nullptr, nullptr, nullptr, out, nullptr, nullptr, s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_filter_correct_bf16x3(coper_handle* h, const float* tgt, const int64_t* e2, const int64_t* indptr,
                                 const int64_t* idx, int64_t nnz, int64_t B, int32_t* ng, int32_t* ne,
                                 hipStream_t s) {
  const Dims& dm = h->dm;
  if (ne) {
    // tie counts requested: the target itself was counted as "equal" and is retired here; the row expansion rides along
    int64_t threads = B * 16;
    hipLaunchKernelGGL(k_expand_rows_retire_target, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, indptr, B, e2,
                       tgt, (int64_t)h->cfg.shard_lo, dm.n_local, h->row_of_ws, ne);
    pair_launch(h, 2, nnz, B, 1, e2, nullptr, indptr, idx, h->row_of_ws, tgt, nullptr, ng, ne, s);
  } else {
    // ranks only (what the reference computes): row ids were written by the target pass of coper_rank /
    // coper_encode_rank; other callers of coper_rank_counts get them by bisection of indptr inside the pair kernel
    const int32_t* rows = h->rows_expanded_for == indptr ? h->row_of_ws : nullptr;
    h->rows_expanded_for = nullptr;
    pair_launch(h, 2, nnz, B, 1, e2, nullptr, indptr, idx, rows, tgt, nullptr, ng, ne, s);
  }
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
