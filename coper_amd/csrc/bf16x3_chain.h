// The ONE summation order of every COPER_SCORE_BF16X3 logit, and the operand images of the count kernel that follow from it.
//
// A logit is  s = pred_bias[e] + sum_k E[e][k] h[q][k]  with every factor split into two bf16 terms (x = hi + lo) and the
// product formed as  E_lo h_hi + E_hi h_lo + E_hi h_hi  on the bf16 matrix cores with fp32 accumulation.  Per k-step of
// 16 that is three "virtual ops" of one K = 16 matrix instruction each:
//     T1(ks) = E_lo(ks) . h_hi(ks)      T2(ks) = E_hi(ks) . h_lo(ks)      T3(ks) = E_hi(ks) . h_hi(ks)
// Round 3 measured (tools/microbench/mfma_shape.hip, 1 M outputs, bit for bit): ONE v_mfma_f32_16x16x32_bf16 over 32 k is
// the same function of (A, B, C) as TWO chained v_mfma_f32_32x32x16_bf16 over k 0..15 then 16..31 -- the K = 32 instruction
// is two K = 16 accumulation steps (lanes 0..31 carry the first, lanes 32..63 the second).  The count kernel
// (kernels_score3_bf16.hip) runs on the 16x16x32 shape (1.12x the FLOP/s of 32x32x16 at the chip's power limit, same
// microbenchmark) and packs two virtual ops into each instruction; the order that makes this possible with two registers
// per pair of k-steps -- and that every other kernel of the mode (tiles of score_all, the gathered-pair kernels, the fused
// tail, the top-k rescoring; all on 32x32x16) follows so that all logits of the mode agree bit for bit -- is
//
//     acc = pred_bias 2^(e_E + e_h)      [round 4: operands enter as E 2^e_E and h 2^e_h -- exact powers of two per table and
//                                          per packed batch (split16.h) -- and whatever leaves is multiplied by 2^-(e_E + e_h)]
//     for j in 0 .. NP-1   (NP = KS16 / 2 pairs of k-steps):
//         acc += T1(2j); acc += T1(2j+1); acc += T2(2j); acc += T2(2j+1); acc += T3(2j); acc += T3(2j+1)
//     if KS16 is odd (last k-step t = KS16 - 1):
//         acc += T1(t); acc += T2(t); acc += T3(t)            [the K = 32 form adds an all-zero second half to T3: acc + 0]
//
// Count-kernel images ("f3"): per block of 16 rows (entities) or 16 columns (queries), per step s (NP pair steps, then
// the tail step when KS16 is odd), two 64-lane registers of 16 bytes; lane l holds row (l & 15), 8 consecutive k of one
// plane: k = 16 ks + 8 ((l >> 4) & 1), with (plane, ks) chosen per half-wave (l >> 5):
//     pair step s:   reg 0 = [lo(2s) | lo(2s+1)]      reg 1 = [hi(2s) | hi(2s+1)]                  (both sides)
//     tail step:     entities: reg 0 = [lo(t) | hi(t)], reg 1 = [hi(t) | 0]
//                    queries:  reg 0 = [hi(t) | 0],     reg 1 = [hi(t) | lo(t)]
// and the instructions of a step are  (e.reg0, q.reg1), (e.reg1, q.reg0), and for pair steps (e.reg1, q.reg1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split16.h"

namespace coper {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (the 16-bit type of the split -- fp16 -- is split16.h's business)
#define BX3_MFMA32(a, b, c) S16_MFMA32(a, b, c)
#define BX3_MFMA16(a, b, c) S16_MFMA16(a, b, c)

// 32x32x16 kernels, entity rows = A, queries = B: two consecutive k-steps (fragments *0 of the first, *1 of the second)
#define BX3_PAIR(eh0, el0, qh0, ql0, eh1, el1, qh1, ql1, c)                     \
  {                                                                             \
    (c) = BX3_MFMA32(el0, qh0, c); (c) = BX3_MFMA32(el1, qh1, c);               \
    (c) = BX3_MFMA32(eh0, ql0, c); (c) = BX3_MFMA32(eh1, ql1, c);               \
    (c) = BX3_MFMA32(eh0, qh0, c); (c) = BX3_MFMA32(eh1, qh1, c);               \
  }
// ... the last k-step of an odd count
#define BX3_LAST(eh, el, qh, ql, c) \
  { (c) = BX3_MFMA32(el, qh, c); (c) = BX3_MFMA32(eh, ql, c); (c) = BX3_MFMA32(eh, qh, c); }
// the same virtual ops with the roles swapped (queries = A rows, entities = B columns: k_score_all_bf16x3): a product does
// not depend on which side of the matrix unit its factors enter
#define BX3_PAIR_QA(eh0, el0, qh0, ql0, eh1, el1, qh1, ql1, c)                  \
  {                                                                             \
    (c) = BX3_MFMA32(qh0, el0, c); (c) = BX3_MFMA32(qh1, el1, c);               \
    (c) = BX3_MFMA32(ql0, eh0, c); (c) = BX3_MFMA32(ql1, eh1, c);               \
    (c) = BX3_MFMA32(qh0, eh0, c); (c) = BX3_MFMA32(qh1, eh1, c);               \
  }
#define BX3_LAST_QA(eh, el, qh, ql, c) \
  { (c) = BX3_MFMA32(qh, el, c); (c) = BX3_MFMA32(ql, eh, c); (c) = BX3_MFMA32(qh, eh, c); }

// ---- f3 images ------------------------------------------------------------------------------------------------------
__host__ __device__ inline int f3_steps(int KS16) { return (KS16 + 1) / 2; }          // NS = NP + TAIL
// uint4 index of register (blk16, step s, which) lane l
__host__ __device__ inline int64_t f3_at(int64_t blk16, int NS, int s, int which, int l) { return ((blk16 * NS + s) * 2 + which) * 64 + l; }

// Writes the 16-byte pieces (hi, lo) of (row, k-step ks, half) into an f3 image.  `query_side`: the queries' tail layout.
// Zero halves of the tail registers are never written: the image is zero-filled when it is allocated.
__device__ __forceinline__ void f3_store_piece(uint4* __restrict__ img, int KS16, int64_t row, int ks, int half, const uint4& hi,
                                               const uint4& lo, bool query_side) {
  const int NS = f3_steps(KS16), NP = KS16 / 2;
  const int64_t blk = row >> 4;
  const int r = (int)(row & 15) + 16 * half;
  if (ks < 2 * NP) {
    const int s = ks >> 1, l = r + 32 * (ks & 1);
    img[f3_at(blk, NS, s, 0, l)] = lo;
    img[f3_at(blk, NS, s, 1, l)] = hi;
  } else if (!query_side) {   // entities: reg 0 = [lo | hi], reg 1 = [hi | 0]
    img[f3_at(blk, NS, NP, 0, r)] = lo;
    img[f3_at(blk, NS, NP, 0, r + 32)] = hi;
    img[f3_at(blk, NS, NP, 1, r)] = hi;
  } else {                    // queries: reg 0 = [hi | 0], reg 1 = [hi | lo]
    img[f3_at(blk, NS, NP, 0, r)] = hi;
    img[f3_at(blk, NS, NP, 1, r)] = hi;
    img[f3_at(blk, NS, NP, 1, r + 32)] = lo;
  }
}

// ---- the exponent of a packed batch (split16.h) ---------------------------------------------------------------------------
// e_h comes from the largest |h| element of the batch.  Two steps, no atomics and no fences (the first form -- blocks folding
// their maxima into one word with atomicMax and a ticket -- cost 40 - 60 us per pass: thousands of device-scope atomics on one
// line are served one after the other at the memory side):
//   1. the kernel that PRODUCES the h rows (k_finalize_h_publish, k_absmax_publish) stores one maximum per block into
//      x3m[blockIdx] (X3M_SLOTS floats; block 0 zeroes the slots beyond the grid);
//   2. the kernel that SPLITS them (the tail kernel, k_rows_to_frag_bf16) reduces the X3M_SLOTS floats in every block (4 KB out
//      of L2) -> e_h; its block 0 publishes x3s[0] = e_h, x3s[1] = e_E + e_h for the kernels that follow it on the stream
//      (count, band walk, pair kernels, score_all, top-k).
// Plain stores ordered by kernel boundaries: deterministic, no host-side state, hipGraph-replayable.
constexpr int X3M_SLOTS = 1024;
__device__ __forceinline__ void x3_block_store_max(float m, float* __restrict__ x3m) {     // blocks of <= 1024 threads, grid <= X3M_SLOTS
  __shared__ float s_m[16];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) s_m[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w) m = fmaxf(m, s_m[w]);
    x3m[blockIdx.x] = m == m ? m : 0.f;            // (NaN: no contribution)
  }
  if (blockIdx.x == 0)
    for (int i = gridDim.x + threadIdx.x; i < X3M_SLOTS; i += blockDim.x) x3m[i] = 0.f;
}
// every thread of a 256-thread block returns e_h; block 0 publishes (s_red: 4 floats of the caller's LDS)
__device__ __forceinline__ int x3_batch_exp(const float* __restrict__ x3m, const int ent_exp, int32_t* __restrict__ x3s, float* s_red) {
  static_assert(X3M_SLOTS == 1024, "one float4 per thread of a 256-thread block");
  const float4 v = ((const float4*)x3m)[threadIdx.x & 255];
  float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  const int e = x3_exp_for_bits(__float_as_uint(m));
  if (blockIdx.x == 0 && threadIdx.x == 0) { x3s[0] = e; x3s[1] = e + ent_exp; }
  return e;
}

// ---- the half-width of the exact band (kernels_score3_bf16.hip) ------------------------------------------------------------
// tau_q = 2 (kappa (|h_q| Emax + 8 Bmax) + abs_q): both logits of a comparison carry an error of at most that half each.
//   kappa |h_q| Emax   the split's RELATIVE error and the rounding of the fp32 accumulation, for logits that are sums of products;
//   8 kappa Bmax       the accumulation's rounding when pred_bias DOMINATES the logit: every step of either arithmetic rounds
//                      to an ulp of the running sum ~ |bias| (the chain 200 times, the matrix cores ~40 times at d = 200), and
//                      the two random walks differ by up to 1.2e-6 |bias| over 6e6 logits (tests/test_gpu_scale.py, tables of
//                      N(0, 0.01^2) beside biases of N(0, 0.1^2)) -- where the products' term measures <= 2.1e-7 |h_q| Emax;
//                      both terms keep a factor 5 above the largest error seen.  (Empirical, like kappa: the band audit
//                      checks every count launch against them; the proven bound is 75x wider, coper_internal.h.)
//   abs_q              the split's ABSOLUTE floor, rigorous: an element below 2^-3 after scaling is off by at most 2^-25
//                      (split16.h), so a logit by at most 2^-25 (|E'|_1 + |h'|_1) in scaled units = 2^-25 sqrt(d) (Emax 2^-e_h +
//                      |h_q| 2^-e_E).  ~2^-35 of |h_q| Emax for a query as large as its batch's largest; it takes over for
//                      queries 2^15 smaller (the batch exponent serves its largest query) and keeps their ranks exact at the
//                      price of more band pairs.
constexpr float X3_BAND_BIAS_WEIGHT = 8.f;
__device__ __forceinline__ float x3_band_tau(float h_norm2, float kappa, const unsigned* __restrict__ consts, int d,
                                             const int32_t* __restrict__ x3s) {
  const float emax = __uint_as_float(consts[0]), bmax = __uint_as_float(consts[1]);
  const float hn = sqrtf(h_norm2) * 1.000001f;
  const int eh = x3s[0], ee = x3s[1] - eh;
  const float abs_q = 2.98023224e-8f * sqrtf((float)d) * (x3_scale(emax, -eh) + x3_scale(hn, -ee));
  return 2.f * (kappa * (hn * emax + X3_BAND_BIAS_WEIGHT * bmax) + abs_q);
}

// ---- the exact chain (the logit of COPER_SCORE_F32: chain_score of kernels_score.hip) on fp32 rows ---------------------
// s = bias; for ks: for t in 0..3: s = fma(E[8ks+t], h[8ks+t], s); s = fma(E[8ks+4+t], h[8ks+4+t], s)
__device__ __forceinline__ float exact_chain(const float* __restrict__ er, const float* __restrict__ hr, float bias, int d) {
  float s = bias;
  const int KS = (d + 7) >> 3;
  if ((d & 7) == 0 && ((((uintptr_t)er) | ((uintptr_t)hr)) & 15) == 0) {
    for (int ks = 0; ks < KS; ++ks) {
      const float4 e0 = *(const float4*)(er + 8 * ks), e1 = *(const float4*)(er + 8 * ks + 4);
      const float4 h0 = *(const float4*)(hr + 8 * ks), h1 = *(const float4*)(hr + 8 * ks + 4);
      s = __builtin_fmaf(e0.x, h0.x, s); s = __builtin_fmaf(e1.x, h1.x, s);
      s = __builtin_fmaf(e0.y, h0.y, s); s = __builtin_fmaf(e1.y, h1.y, s);
      s = __builtin_fmaf(e0.z, h0.z, s); s = __builtin_fmaf(e1.z, h1.z, s);
      s = __builtin_fmaf(e0.w, h0.w, s); s = __builtin_fmaf(e1.w, h1.w, s);
    }
    return s;
  }
  for (int ks = 0; ks < KS; ++ks)
    for (int t = 0; t < 4; ++t) {
      const int k0 = 8 * ks + t, k1 = k0 + 4;
      if (k0 < d) s = __builtin_fmaf(er[k0], hr[k0], s);
      if (k1 < d) s = __builtin_fmaf(er[k1], hr[k1], s);
    }
  return s;
}


// The same chain for the two logits of a comparison -- entity row `er` and target row `tr` (may be NULL: t_out untouched)
// against one query row -- with the loads of CB k-steps of 8 issued before their fmas: a lane that walks a chain alone is
// latency-bound on its row loads (25 dependent round trips to L2 at d = 200 took 25 us per chain; in batches of five: 5).
#ifndef COPER_CHAIN_CB
#define COPER_CHAIN_CB 5
#endif
// CB: k-steps of 8 values requested per round trip (5: the band walk of small tables, the tail kernels; 3: the band walk of a large
// table or shard, which decides hundreds of thousands of pairs, one per lane -- there residency counts for more than round trips:
// 10 M x 256, 4,096 queries: 44.1 -> 43.85 ms, a rank of eight 6.43 -> 6.37 ms; 8: slower everywhere).  The order of the chain's
// fused multiply-adds does not depend on it.
template <int CB = COPER_CHAIN_CB>
__device__ __forceinline__ void exact_chain_pair(const float* __restrict__ er, const float* __restrict__ tr, const float* __restrict__ hr,
                                                 float bias_e, float bias_t, int d, float& s_out, float& t_out) {
  const int KS = (d + 7) >> 3;
  uintptr_t al = ((uintptr_t)er) | ((uintptr_t)hr);
  if (tr) al |= (uintptr_t)tr;
  if ((d & 7) != 0 || (al & 15) != 0) {
    s_out = exact_chain(er, hr, bias_e, d);
    if (tr) t_out = exact_chain(tr, hr, bias_t, d);
    return;
  }
  float s = bias_e, t = bias_t;
  const float4* e4 = (const float4*)er;
  const float4* t4 = (const float4*)(tr ? tr : er);
  const float4* h4 = (const float4*)hr;
  for (int k0 = 0; k0 < KS; k0 += CB) {
    float4 ev[CB][2], tv[CB][2], hv[CB][2];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int k = k0 + u < KS ? k0 + u : KS - 1;
      ev[u][0] = e4[2 * k]; ev[u][1] = e4[2 * k + 1];
      hv[u][0] = h4[2 * k]; hv[u][1] = h4[2 * k + 1];
      if (tr) { tv[u][0] = t4[2 * k]; tv[u][1] = t4[2 * k + 1]; }
    }
#pragma unroll
    for (int u = 0; u < CB; ++u)
      if (k0 + u < KS) {
        s = __builtin_fmaf(ev[u][0].x, hv[u][0].x, s); s = __builtin_fmaf(ev[u][1].x, hv[u][1].x, s);
        s = __builtin_fmaf(ev[u][0].y, hv[u][0].y, s); s = __builtin_fmaf(ev[u][1].y, hv[u][1].y, s);
        s = __builtin_fmaf(ev[u][0].z, hv[u][0].z, s); s = __builtin_fmaf(ev[u][1].z, hv[u][1].z, s);
        s = __builtin_fmaf(ev[u][0].w, hv[u][0].w, s); s = __builtin_fmaf(ev[u][1].w, hv[u][1].w, s);
        if (tr) {
          t = __builtin_fmaf(tv[u][0].x, hv[u][0].x, t); t = __builtin_fmaf(tv[u][1].x, hv[u][1].x, t);
          t = __builtin_fmaf(tv[u][0].y, hv[u][0].y, t); t = __builtin_fmaf(tv[u][1].y, hv[u][1].y, t);
          t = __builtin_fmaf(tv[u][0].z, hv[u][0].z, t); t = __builtin_fmaf(tv[u][1].z, hv[u][1].z, t);
          t = __builtin_fmaf(tv[u][0].w, hv[u][0].w, t); t = __builtin_fmaf(tv[u][1].w, hv[u][1].w, t);
        }
      }
  }
  s_out = s;
  if (tr) t_out = t;
}

}  // namespace coper
