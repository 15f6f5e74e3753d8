// k_score_count2_bf16x3 -- the fused score + count pass of the bf16x3 mode, software-pipelined for ONE wave per SIMD.
//
// Same arithmetic as k_score_count_bf16x3 (kernels_score_bf16.hip): every logit is the accumulator of
//   acc = pred_bias;  for ks: acc = mfma(E_lo, h_hi, acc); acc = mfma(E_hi, h_lo, acc); acc = mfma(E_hi, h_hi, acc)
// on v_mfma_f32_32x32x16_bf16 with entity rows as A and queries as B, so the counts are the same integers
// (tests/test_gpu_parity.py compares them with the reference ranker applied to the materialised logits).
//
// What is different is the schedule.  The first kernel runs two waves per SIMD with 256 registers each and lets the
// partner wave cover every stall: its LDS reads are waited for right after they are issued, the compare epilogue and
// the accumulator initialisation run with the matrix pipe idle whenever both waves reach them together (measured:
// matrix pipes busy 65 % of the cycles).  Here a workgroup is 4 waves, one per SIMD:
//   * a wave alternates between its two entity blocks (m = 0, 1): while the 12 MFMAs per k-step of block m run, the
//     compare / count epilogue of block 1-m (finished a half-row earlier) is issued between them, a value or two per
//     accumulator chain -- VALU beside MFMA on one SIMD goes in the MFMA's shadow;
//   * the first MFMA of every accumulator chain takes pred_bias as its C operand: no initialisation moves;
//   * the query fragments of the next k-step are read from LDS during this one (two register buffers), the entity
//     fragments come straight from the fragment image into registers PD k-steps ahead (two sets of PD buffers), across
//     block and row boundaries -- every wait the compiler inserts is a counted vmcnt / lgkmcnt for data requested a
//     k-step or more earlier;
//   * work is dealt in rows of 8 entity blocks (a wave owns 2) instead of units of 16: 2 % imbalance instead of 6 %.
// A row = (128-query tile, 8 consecutive 32-entity blocks); a workgroup walks a contiguous range of rows.
// All register state is indexed with template constants (integer sequences, no loops to unroll).
#include <algorithm>
#include <utility>

#include "coper_internal.h"

namespace coper {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA2_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&(a), *(const bf16x8*)&(b), (c), 0, 0, 0)
#define SC2_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int KS, int PD, bool EQ, bool GM>
struct SC2 {
  static constexpr int NQ = 4;
  static constexpr int G = (KS + PD - 1) / PD;            // prefetch groups per half-row
  static constexpr int NV = NQ * 16;                      // accumulator values per lane per half-row
  static constexpr int CH = (NV + KS - 1) / KS;           // epilogue values handled per k-step
  f32x16 acc[2][NQ];                                      // [entity block m][query block b]
  uint4 ah[2][PD], al[2][PD];                             // entity fragments: two sets of PD k-steps
  uint4 bh[2][NQ], bl[2][NQ];                             // query fragments: this k-step / the next
  f32x16 biasv[2];
  float t[NQ];
  int cg[NQ], ce[NQ];
  float mx[NQ];
};

struct SC2Ptrs {
  const uint4 *a_hi[2], *a_lo[2];           // fragment (m, ks = 0) of this row's two blocks, lane included
  const uint4 *n_hi, *n_lo;                 // ... of the next row's block 0
  const uint4 *hl_hi, *hl_lo;               // the query tile in LDS, lane included
};

// one value of the other block's accumulators: compare with the query's target, count; block maximum for the top-k
template <int KS, int PD, bool EQ, bool GM, int M, int V>
__device__ __forceinline__ void sc2_value(SC2<KS, PD, EQ, GM>& S, const int lane, const bool store_gm, float* __restrict__ gm_row,
                                          const int64_t gm_col) {
  constexpr int b = V / 16, r = V % 16;
#ifdef COPER_DBG_SC2_EPI_R0   /* ablation: one value per accumulator keeps the MFMA chains alive, the epilogue nearly free */
  if constexpr (r != 0) return;
#endif
  // The accumulator is read out of its AGPR here, at the point of use: left to the compiler the whole accumulator is
  // copied to VGPRs right behind its last MFMA (s_nop + 16 reads that wait for the matrix pipe to drain).
  float sc;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sc) : "a"(S.acc[M][b][r]));
  // count += (sc > t): compare into VCC, add with carry
  asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(S.cg[b]) : "v"(sc), "v"(S.t[b]) : "vcc");
  if constexpr (EQ)
    asm volatile("v_cmp_eq_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(S.ce[b]) : "v"(sc), "v"(S.t[b]) : "vcc");
  if constexpr (GM) {
    if constexpr (r == 0) S.mx[b] = sc; else S.mx[b] = fmaxf(S.mx[b], sc);
    if constexpr (r == 15) {
      const float mxx = fmaxf(S.mx[b], __shfl_xor(S.mx[b], 32));   // the other 16 rows of the block
      if (store_gm && lane < 32) gm_row[gm_col + b * 32 + lane] = mxx;
    }
  }
}

template <int KS, int PD, bool EQ, bool GM, int M, int... V>
__device__ __forceinline__ void sc2_values(SC2<KS, PD, EQ, GM>& S, const int lane, float* __restrict__ gm_row, const int64_t gm_col,
                                           std::integer_sequence<int, V...>) {
  (sc2_value<KS, PD, EQ, GM, M, V>(S, lane, true, gm_row, gm_col), ...);
}

template <int KS, int PD, bool EQ, bool GM, int M>
__device__ __forceinline__ void sc2_load_bias(SC2<KS, PD, EQ, GM>& S, const float* __restrict__ bias_pad, const int64_t blk, const int lane) {
  const float4* bp = (const float4*)(bias_pad + blk * 32 + 4 * (lane >> 5));
  const float4 q0 = bp[0], q1 = bp[2], q2 = bp[4], q3 = bp[6];
  S.biasv[M][0] = q0.x; S.biasv[M][1] = q0.y; S.biasv[M][2] = q0.z; S.biasv[M][3] = q0.w;
  S.biasv[M][4] = q1.x; S.biasv[M][5] = q1.y; S.biasv[M][6] = q1.z; S.biasv[M][7] = q1.w;
  S.biasv[M][8] = q2.x; S.biasv[M][9] = q2.y; S.biasv[M][10] = q2.z; S.biasv[M][11] = q2.w;
  S.biasv[M][12] = q3.x; S.biasv[M][13] = q3.y; S.biasv[M][14] = q3.z; S.biasv[M][15] = q3.w;
}

// Region b (0..3) of k-step ks of block M: one accumulator chain (3 MFMAs) with, in front of it, one entity-fragment
// load PD k-steps ahead (regions 0, 1), two LDS reads of the next k-step's query fragments, and behind it this region's
// share of the other block's epilogue.
template <int KS, int PD, bool EQ, bool GM, int M, int ks, int b>
__device__ __forceinline__ void sc2_region(SC2<KS, PD, EQ, GM>& S, const SC2Ptrs& X, const int lane, const bool prev_valid,
                                           float* __restrict__ gm_row, const int64_t gm_col) {
  typedef SC2<KS, PD, EQ, GM> ST;
  constexpr int G = ST::G, CH = ST::CH, NV = ST::NV;
  constexpr int PA = (G & 1) ? M : 0;        // entity-fragment set of this block's k-step 0
  constexpr int PA_NEXT = (PA + G) & 1;      // ... of the next block's k-step 0
  constexpr int PB = (KS & 1) ? M : 0;       // query-fragment buffer of this block's k-step 0
  constexpr int sa = (PA + ks / PD) & 1, sl = ks % PD, sb = (PB + ks) & 1, tk = ks + PD;
#ifndef COPER_DBG_SC2_SKIP_GL
  if constexpr (b < 2) {
    if constexpr (tk < KS) {
      constexpr int ta = (PA + tk / PD) & 1, tl = tk % PD;
      if constexpr (b == 0) S.ah[ta][tl] = X.a_hi[M][tk * 64]; else S.al[ta][tl] = X.a_lo[M][tk * 64];
    } else if constexpr (M == 0) {           // k-step tk - KS of this row's block 1
      constexpr int j = tk - KS;
      if constexpr (b == 0) S.ah[PA_NEXT][j] = X.a_hi[1][j * 64]; else S.al[PA_NEXT][j] = X.a_lo[1][j * 64];
    } else {                                 // ... of the next row's block 0
      constexpr int j = tk - KS;
      if constexpr (b == 0) S.ah[PA_NEXT][j] = X.n_hi[j * 64]; else S.al[PA_NEXT][j] = X.n_lo[j * 64];
    }
  }
#endif
#ifndef COPER_DBG_SC2_SKIP_LDS
  {
    constexpr int nk = ks + 1 < KS ? ks + 1 : 0;
    S.bh[sb ^ 1][b] = X.hl_hi[(b * KS + nk) * 64];
    S.bl[sb ^ 1][b] = X.hl_lo[(b * KS + nk) * 64];
  }
#endif
  // the chain: smallest terms first; k-step 0 starts from pred_bias
  if constexpr (ks == 0) S.acc[M][b] = MFMA2_BF16(S.al[sa][sl], S.bh[sb][b], S.biasv[M]);
  else S.acc[M][b] = MFMA2_BF16(S.al[sa][sl], S.bh[sb][b], S.acc[M][b]);
  S.acc[M][b] = MFMA2_BF16(S.ah[sa][sl], S.bl[sb][b], S.acc[M][b]);
  S.acc[M][b] = MFMA2_BF16(S.ah[sa][sl], S.bh[sb][b], S.acc[M][b]);
  // epilogue of the other block: this k-step's chunk of CH values is dealt to the four regions in order (the running
  // block maximum needs value r = 0 of an accumulator first and r = 15 last)
  constexpr int c0 = b * CH / 4, c1 = (b + 1) * CH / 4, v0 = ks * CH + c0;
#ifndef COPER_DBG_SC2_NO_EPI
  if constexpr (c1 - c0 > 0 && v0 < NV) sc2_value<KS, PD, EQ, GM, 1 - M, v0>(S, lane, prev_valid, gm_row, gm_col);
  if constexpr (c1 - c0 > 1 && v0 + 1 < NV) sc2_value<KS, PD, EQ, GM, 1 - M, v0 + 1>(S, lane, prev_valid, gm_row, gm_col);
  if constexpr (c1 - c0 > 2 && v0 + 2 < NV) sc2_value<KS, PD, EQ, GM, 1 - M, v0 + 2>(S, lane, prev_valid, gm_row, gm_col);
#endif
  static_assert(CH <= 12, "three epilogue values per region at most");
  SC2_FENCE();
}

template <int KS, int PD, bool EQ, bool GM, int M, int ks>
__device__ __forceinline__ void sc2_step(SC2<KS, PD, EQ, GM>& S, const SC2Ptrs& X, const float* __restrict__ bias_pad,
                                         const int64_t bias_blk_next, const int lane, const bool prev_valid, float* __restrict__ gm_row,
                                         const int64_t gm_col) {
  sc2_region<KS, PD, EQ, GM, M, ks, 0>(S, X, lane, prev_valid, gm_row, gm_col);
  sc2_region<KS, PD, EQ, GM, M, ks, 1>(S, X, lane, prev_valid, gm_row, gm_col);
  sc2_region<KS, PD, EQ, GM, M, ks, 2>(S, X, lane, prev_valid, gm_row, gm_col);
  sc2_region<KS, PD, EQ, GM, M, ks, 3>(S, X, lane, prev_valid, gm_row, gm_col);
  if constexpr (ks == 0) {
    // pred_bias of this block in the NEXT row: the four chains have consumed biasv[M] (program order)
    sc2_load_bias<KS, PD, EQ, GM, M>(S, bias_pad, bias_blk_next, lane);
    SC2_FENCE();
  }
}

// half a row: the KS k-steps of block M
template <int KS, int PD, bool EQ, bool GM, int M, int... K>
__device__ __forceinline__ void sc2_half(SC2<KS, PD, EQ, GM>& S, const SC2Ptrs& X, const float* __restrict__ bias_pad,
                                         const int64_t bias_blk_next, const int lane, const bool prev_valid, float* __restrict__ gm_row,
                                         const int64_t gm_col, std::integer_sequence<int, K...>) {
  (sc2_step<KS, PD, EQ, GM, M, K>(S, X, bias_pad, bias_blk_next, lane, prev_valid, gm_row, gm_col), ...);
}

template <int KS, int PD, bool EQ, bool GM, int... J>
__device__ __forceinline__ void sc2_prologue_a(SC2<KS, PD, EQ, GM>& S, const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo,
                                               const int64_t eb, const int lane, std::integer_sequence<int, J...>) {
  ((S.ah[0][J] = Ehi[(eb * KS + (J < KS ? J : KS - 1)) * 64 + lane], S.al[0][J] = Elo[(eb * KS + (J < KS ? J : KS - 1)) * 64 + lane]), ...);
}

#ifdef COPER_DBG_CLOCK
// diagnostic build (tools/ab_build.py): shader clock held inside the kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6); the stamps go to a buffer of their own, no output depends on them
__device__ unsigned long long g_sc2_clk[2 * 1024];
#endif

template <int KS, int PD, bool EQ, bool GM, int... J>
__device__ __forceinline__ void sc2_prologue_a1(SC2<KS, PD, EQ, GM>& S, std::integer_sequence<int, J...>) {   // ablation builds only
  ((S.ah[1][J] = S.ah[0][J], S.al[1][J] = S.al[0][J]), ...);
}

template <int KS, int PD, bool EQ, bool GM>
__global__ __launch_bounds__(256, 1) void k_score_count2_bf16x3(const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo,
                                                                 const float* __restrict__ bias_pad,
                                                                 const uint4* __restrict__ Hhi, const uint4* __restrict__ Hlo,
                                                                 const float* __restrict__ tgt, int64_t B, int64_t rows_per_tile,
                                                                 int64_t total_rows, int32_t* __restrict__ ng,
                                                                 int32_t* __restrict__ ne, float* __restrict__ gmax,
                                                                 int64_t gm_stride) {
  typedef SC2<KS, PD, EQ, GM> ST;
  constexpr int NQ = ST::NQ, NV = ST::NV;
  static_assert(PD <= KS, "the prefetch reaches at most one half-row ahead");
  extern __shared__ uint4 hl16[];  // [2 planes][NQ][KS][64]
  uint4* hl_hi = hl16;
  uint4* hl_lo = hl16 + NQ * KS * 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t r_begin = total_rows * blockIdx.x / gridDim.x, r_end = total_rows * (blockIdx.x + 1) / gridDim.x;
  if (r_begin >= r_end) return;
#ifdef COPER_DBG_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  ST S;
  int64_t cur_tile = -1;
  int64_t eb_prev = 0;
  bool prev_valid = false;
  const std::make_integer_sequence<int, KS> KSEQ{};
  const std::make_integer_sequence<int, NV> VSEQ{};

  for (int64_t r = r_begin; r < r_end; ++r) {
    const int64_t tile = r / rows_per_tile;
    const int64_t eb = ((r % rows_per_tile) * 4 + wave) * 2;
    if (tile != cur_tile) {   // workgroup-uniform: (re)start of the pipeline
      __syncthreads();
      const uint4* sh = Hhi + tile * (NQ * KS * 64);
      const uint4* sl = Hlo + tile * (NQ * KS * 64);
      // the query tile (2 x NQ x KS KiB) into LDS: NQ * KS / 4 pieces per thread and plane, all loads of a plane's half in
      // flight before the first LDS store (a load-store-load-store loop pays one L2 round trip per piece)
      {
        constexpr int NPC = NQ * KS * 64 / 256;   // pieces per thread per plane (13 or 16)
        constexpr int HB = (NPC + 1) / 2;
        uint4 th[HB], tl[HB];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
          for (int u = 0; u < HB; ++u) {
            const int j = (part * HB + u) * 256 + threadIdx.x;
            if (part * HB + u < NPC) { th[u] = sh[j]; tl[u] = sl[j]; }
          }
#pragma unroll
          for (int u = 0; u < HB; ++u) {
            const int j = (part * HB + u) * 256 + threadIdx.x;
            if (part * HB + u < NPC) { hl_hi[j] = th[u]; hl_lo[j] = tl[u]; }
          }
        }
      }
      cur_tile = tile;
      {
        const int64_t q = tile * (32 * NQ) + (lane & 31);
        S.t[0] = q < B ? tgt[q] : INFINITY; S.t[1] = q + 32 < B ? tgt[q + 32] : INFINITY;
        S.t[2] = q + 64 < B ? tgt[q + 64] : INFINITY; S.t[3] = q + 96 < B ? tgt[q + 96] : INFINITY;
        S.cg[0] = S.cg[1] = S.cg[2] = S.cg[3] = 0;
        S.ce[0] = S.ce[1] = S.ce[2] = S.ce[3] = 0;
      }
      // entity fragments of block 0's first PD k-steps, pred_bias of both blocks, "previous block" accumulators that
      // count nothing
      sc2_prologue_a<KS, PD, EQ, GM>(S, Ehi, Elo, eb, lane, std::make_integer_sequence<int, PD>{});
      sc2_load_bias<KS, PD, EQ, GM, 0>(S, bias_pad, eb, lane);
      sc2_load_bias<KS, PD, EQ, GM, 1>(S, bias_pad, eb + 1, lane);
      S.acc[1][0] = f32x16(-INFINITY); S.acc[1][1] = f32x16(-INFINITY); S.acc[1][2] = f32x16(-INFINITY); S.acc[1][3] = f32x16(-INFINITY);
      prev_valid = false;
      __syncthreads();
      S.bh[0][0] = hl_hi[(0 * KS) * 64 + lane]; S.bl[0][0] = hl_lo[(0 * KS) * 64 + lane];
      S.bh[0][1] = hl_hi[(1 * KS) * 64 + lane]; S.bl[0][1] = hl_lo[(1 * KS) * 64 + lane];
      S.bh[0][2] = hl_hi[(2 * KS) * 64 + lane]; S.bl[0][2] = hl_lo[(2 * KS) * 64 + lane];
      S.bh[0][3] = hl_hi[(3 * KS) * 64 + lane]; S.bl[0][3] = hl_lo[(3 * KS) * 64 + lane];
#if defined(COPER_DBG_SC2_SKIP_LDS) || defined(COPER_DBG_SC2_SKIP_GL)
      S.bh[1][0] = S.bh[0][0]; S.bh[1][1] = S.bh[0][1]; S.bh[1][2] = S.bh[0][2]; S.bh[1][3] = S.bh[0][3];
      S.bl[1][0] = S.bl[0][0]; S.bl[1][1] = S.bl[0][1]; S.bl[1][2] = S.bl[0][2]; S.bl[1][3] = S.bl[0][3];
      sc2_prologue_a1<KS, PD, EQ, GM>(S, std::make_integer_sequence<int, PD>{});
#endif
    }
    const bool has_next = r + 1 < r_end;
    const bool last_of_tile = !has_next || (r + 1) / rows_per_tile != tile;
    const int64_t eb_next = has_next ? (((r + 1) % rows_per_tile) * 4 + wave) * 2 : eb;     // past the end: re-read this row's blocks
    const int64_t gm_col = cur_tile * (32 * NQ);
    SC2Ptrs X;
    X.a_hi[0] = Ehi + eb * KS * 64 + lane;       // fragment (blk, ks) at ((blk * KS) + ks) * 64 + lane
    X.a_lo[0] = Elo + eb * KS * 64 + lane;
    X.a_hi[1] = X.a_hi[0] + KS * 64;
    X.a_lo[1] = X.a_lo[0] + KS * 64;
    X.n_hi = Ehi + eb_next * KS * 64 + lane;
    X.n_lo = Elo + eb_next * KS * 64 + lane;
    X.hl_hi = hl_hi + lane;
    X.hl_lo = hl_lo + lane;
    // block 0 (epilogue of the previous row's block 1 beside it), then block 1 (epilogue of this row's block 0)
    sc2_half<KS, PD, EQ, GM, 0>(S, X, bias_pad, eb_next, lane, prev_valid, GM ? gmax + (eb_prev + 1) * gm_stride : nullptr, gm_col, KSEQ);
    sc2_half<KS, PD, EQ, GM, 1>(S, X, bias_pad, eb_next + 1, lane, true, GM ? gmax + eb * gm_stride : nullptr, gm_col, KSEQ);
    eb_prev = eb;
    prev_valid = true;
    if (last_of_tile) {
      // drain: block 1's accumulators have no next row of the same tile to hide behind
      sc2_values<KS, PD, EQ, GM, 1>(S, lane, GM ? gmax + (eb + 1) * gm_stride : nullptr, gm_col, VSEQ);
      S.acc[1][0] = f32x16(-INFINITY); S.acc[1][1] = f32x16(-INFINITY); S.acc[1][2] = f32x16(-INFINITY); S.acc[1][3] = f32x16(-INFINITY);
      {
        const int g0 = S.cg[0] + __shfl_xor(S.cg[0], 32), g1 = S.cg[1] + __shfl_xor(S.cg[1], 32);
        const int g2 = S.cg[2] + __shfl_xor(S.cg[2], 32), g3 = S.cg[3] + __shfl_xor(S.cg[3], 32);
        const int64_t q = cur_tile * (32 * NQ) + (lane & 31);
        if (lane < 32) {
          if (q < B && g0) atomicAdd(&ng[q], g0);
          if (q + 32 < B && g1) atomicAdd(&ng[q + 32], g1);
          if (q + 64 < B && g2) atomicAdd(&ng[q + 64], g2);
          if (q + 96 < B && g3) atomicAdd(&ng[q + 96], g3);
        }
        if constexpr (EQ) {
          const int e0 = S.ce[0] + __shfl_xor(S.ce[0], 32), e1 = S.ce[1] + __shfl_xor(S.ce[1], 32);
          const int e2 = S.ce[2] + __shfl_xor(S.ce[2], 32), e3 = S.ce[3] + __shfl_xor(S.ce[3], 32);
          if (lane < 32) {
            if (q < B && e0) atomicAdd(&ne[q], e0);
            if (q + 32 < B && e1) atomicAdd(&ne[q + 32], e1);
            if (q + 64 < B && e2) atomicAdd(&ne[q + 64], e2);
            if (q + 96 < B && e3) atomicAdd(&ne[q + 96], e3);
          }
        }
        S.cg[0] = S.cg[1] = S.cg[2] = S.cg[3] = 0;
        S.ce[0] = S.ce[1] = S.ce[2] = S.ce[3] = 0;
      }
      prev_valid = false;
    }
  }
#ifdef COPER_DBG_CLOCK
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_sc2_clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_t0;
    g_sc2_clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
}

#ifdef COPER_DBG_CLOCK
extern "C" __attribute__((visibility("default"))) int coper_dbg_clock(int n_wg, double* ghz_median, double* us_median) {
  static unsigned long long hbuf[2 * 1024];
  if (hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_sc2_clk), sizeof hbuf) != hipSuccess) return 1;
  std::vector<double> g, u;
  for (int i = 0; i < n_wg && i < 1024; ++i)
    if (hbuf[2 * i + 1]) { g.push_back((double)hbuf[2 * i] / (double)hbuf[2 * i + 1] * 0.1); u.push_back((double)hbuf[2 * i + 1] * 0.01); }
  if (g.empty()) return 2;
  std::sort(g.begin(), g.end());
  std::sort(u.begin(), u.end());
  *ghz_median = g[g.size() / 2];
  *us_median = u[u.size() / 2];
  return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int KS, int PD>
static int sc2_launch(coper_handle* h, int64_t q0, int64_t Bc, const float* tgt, int32_t* ng, int32_t* ne, float* gmax,
                      int64_t gm_stride, hipStream_t s) {
  const Dims& dm = h->dm;
  const int64_t q_tiles = (Bc + 127) / 128;
  const int64_t rows_per_tile = dm.n_eblk / 8;
  const int64_t total_rows = q_tiles * rows_per_tile;
  int64_t grid = h->num_cus;
  if (grid > total_rows) grid = total_rows;
  const size_t lds = (size_t)2 * 4 * KS * 64 * sizeof(uint4);
  const uint4* hhi = (const uint4*)h->hfrag16_hi + (q0 / 32) * KS * 64;
  const uint4* hlo = (const uint4*)h->hfrag16_lo + (q0 / 32) * KS * 64;
#define SC2_LAUNCH(EQ_, GM_)                                                                                                   \
  {                                                                                                                            \
    static bool attr_done[16] = {};                                                                                            \
    const int dev = h->cfg.device & 15;                                                                                        \
    if (!attr_done[dev]) {                                                                                                     \
      COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count2_bf16x3<KS, PD, EQ_, GM_>,                               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
      attr_done[dev] = true;                                                                                                   \
    }                                                                                                                          \
    hipLaunchKernelGGL((k_score_count2_bf16x3<KS, PD, EQ_, GM_>), dim3((unsigned)grid), dim3(256), lds, s,                     \
                       (const uint4*)h->Ef16_hi, (const uint4*)h->Ef16_lo, h->bias_pad, hhi, hlo, tgt + q0, Bc, rows_per_tile,  \
                       total_rows, ng + q0, ne ? ne + q0 : nullptr, gmax, gm_stride);                                          \
  }
  if (gmax) { if (ne) SC2_LAUNCH(true, true) else SC2_LAUNCH(false, true) }
  else      { if (ne) SC2_LAUNCH(true, false) else SC2_LAUNCH(false, false) }
#undef SC2_LAUNCH
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// true when the pipelined kernel serves this configuration (k-step counts it is instantiated for)
bool score_count2_supported(const coper_handle* h) {
  static const bool off = getenv("COPER_SCORE_V1") != nullptr;   // A/B switch, read once
  if (off) return false;
  return h->dm.KS16 == 13 || h->dm.KS16 == 16;
}

int score_count2_chunk_bf16x3(coper_handle* h, int64_t q0, int64_t Bc, const float* tgt, int32_t* ng, int32_t* ne, float* gmax,
                              int64_t gm_stride, hipStream_t s) {
  switch (h->dm.KS16) {
#ifndef COPER_SC2_PD
#define COPER_SC2_PD 6
#endif
    case 13: return sc2_launch<13, COPER_SC2_PD>(h, q0, Bc, tgt, ng, ne, gmax, gm_stride, s);
    case 16: return sc2_launch<16, COPER_SC2_PD>(h, q0, Bc, tgt, ng, ne, gmax, gm_stride, s);
  }
  return fail(h, COPER_EUNSUPPORTED, "score_count2: k-step count not instantiated");
}

}  // namespace coper
