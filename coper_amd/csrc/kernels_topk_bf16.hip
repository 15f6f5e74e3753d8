// Pruned top-k of the filtered rows (coper_rank_counts with 0 < k <= 128): the logits are never materialised, also
// not for the top-k.  Kernels of the bf16x3 mode; the fp32-exact mode shares the threshold and selection kernels
// (launch_topk_pruned_f32 at the end; its block maxima and VALU rescoring live in kernels_score.hip).
//
// The per-shard top-k is what the entity-sharded ranker exchanges (the masked row of
// metrics.py:44-46 is the thing it is the top of).  At 1.25 M entities per shard a [B, |E_shard|] logit
// matrix is 20 GB per 4096 queries, so the selection works on block maxima instead:
//
//   1. the count pass itself (k_score_count_bf16x3<.., GM>) writes, next to the rank counters, the largest logit
//      of every (32-entity block, query): gmax[block][query], 1/32 of the logits, coalesced 128-B rows;
//   2. k_topk_threshold_emit: per query, the m-th largest block maximum tau, m = k + (filter entries of the
//      query), by radix select on the float bits.  m distinct blocks have their maximum >= tau and at most
//      `filter entries` of those maxima are masked, so at least k unmasked logits >= tau exist: every entity of
//      the row's top-k sits in a block whose maximum is >= tau.  Exactly m blocks are emitted (all above tau, and
//      the lowest-numbered ones equal to tau), so the candidate list has a fixed place k*q + indptr[q] and no
//      size exchange with the host is needed;
//   3. the candidate slots are grouped by entity block (k_topk_blk_scan / k_topk_blk_scatter, device-side counters)
//      and k_topk_score_blocks re-computes 32 logits x 32 slots per wave with the instruction sequence of every
//      other bf16x3 kernel (bit-identical values), masking the known answers except the target;
//   4. k_topk_select_cand, one wave per query: candidates >= tau are compacted into LDS and placed by counting the
//      survivors ahead of each, (score desc, id asc).
//
// Work beyond the count pass: (k*B + nnz) blocks of 32 logits instead of B*|E|.
#include "bf16x3_chain.h"
#include "coper_internal.h"

namespace coper {

// order-preserving map float -> uint32
__device__ __forceinline__ uint32_t tk_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int TK_THREADS = 512;
#ifndef COPER_TK_DEPTH
#define COPER_TK_DEPTH 4
#endif
constexpr int TK_DEPTH = COPER_TK_DEPTH;   // 16-byte loads a thread keeps in flight per batch (two batches overlap).  Measured on
// the 10M-entity table (threshold kernel, 4,096 queries): 4 -> 3.49 ms, 2 -> 3.60, 8 -> 4.03, 16 -> 4.60

// One sweep of a thread's share of the block axis: gmax[block][query] read as float4 (4 queries of one block), TK_DEPTH
// loads per batch and the next batch in flight while the current one is consumed (the sweeps are latency-bound).
template <typename F>
__device__ __forceinline__ void tk_sweep(const float4* __restrict__ col, int64_t qs4, int64_t g_lo, int64_t g_hi, F&& f) {
  constexpr int D = TK_DEPTH;
  int64_t g = g_lo;
  if (g + D <= g_hi) {
    float4 cur[D];
#pragma unroll
    for (int u = 0; u < D; ++u) cur[u] = col[(g + u) * qs4];
    for (; g + 2 * D <= g_hi; g += D) {
      float4 nxt[D];
#pragma unroll
      for (int u = 0; u < D; ++u) nxt[u] = col[(g + D + u) * qs4];
#pragma unroll
      for (int u = 0; u < D; ++u) f(g + u, cur[u]);
#pragma unroll
      for (int u = 0; u < D; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int u = 0; u < D; ++u) f(g + u, cur[u]);
    g += D;
  }
  for (; g < g_hi; ++g) f(g, col[g * qs4]);
}

// The same, for the blocks g = first, first + step, ...: at any moment the threads of a workgroup -- and the workgroups of
// the other strips, which move through the block axis at the same pace -- read neighbouring rows of gmax.  For the sweeps
// whose result does not depend on the visiting order: histograms, and the fast path's emission (slots by counter, ties by id).
template <typename F>
__device__ __forceinline__ void tk_sweep_strided(const float4* __restrict__ col, int64_t qs4, int64_t first, int64_t step, int64_t G, F&& f) {
  constexpr int D = TK_DEPTH;
  int64_t g = first;
  if (g + (D - 1) * step < G) {
    float4 cur[D];
#pragma unroll
    for (int u = 0; u < D; ++u) cur[u] = col[(g + u * step) * qs4];
    for (; g + (2 * D - 1) * step < G; g += D * step) {
      float4 nxt[D];
#pragma unroll
      for (int u = 0; u < D; ++u) nxt[u] = col[(g + (D + u) * step) * qs4];
#pragma unroll
      for (int u = 0; u < D; ++u) f(g + u * step, cur[u]);
#pragma unroll
      for (int u = 0; u < D; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int u = 0; u < D; ++u) f(g + u * step, cur[u]);
    g += D * step;
  }
  for (; g < G; g += step) f(g, col[g * qs4]);
}

constexpr int TK_BIN = 64;   // block maxima per query that may share the threshold's upper 16 bits on the fast path
// Round 5 -- the coarse seed (long block axes: the 10M-entity table and its shards).  The three sweeps above read every block
// maximum three times (2.56 GB each for 4,096 queries against 10 M entities).  Instead: ONE sweep that only keeps the largest key
// of every TK_GRP consecutive visits of a thread (1/16 of the data, written to `coarse`), an exact radix select of the m-th
// largest COARSE key tau_c on that level (four passes over 1/16), and then only the groups whose coarse key reaches tau_c are
// read again: the m-th largest maximum tau is >= tau_c (the m largest coarse keys are m maxima >= tau_c), so every block with a
// maximum >= tau lies in such a group -- about m groups per query.  Their maxima >= tau_c go to an LDS list (TK_CL entries per
// query) that is ranked (key desc, block asc) exactly as the bin list of the fast path was; a list that overflows (queries with
// hundreds of known answers: m = k + their number) sends the strip down the three-sweep route.
constexpr int TK_GRP = 16;   // visits per coarse key
constexpr int TK_CL = 128;   // candidate-list entries per query on the coarse route (>= TK_BIN: the two routes share the lists)

// Threshold + candidate blocks of a strip of NQS = 4 * QV queries (QV = 8: 32 queries; QV = 4: 16).
// thread = (sub-range of the block axis, 4 queries); HCOPY histogram copies (sub-range % HCOPY) thin out same-bank
// LDS atomics.  Three sweeps over the strip's block maxima in the common case:
//   1, 2  radix-select the upper 16 bits of the m-th largest (two 8-bit digits);
//   3     blocks above that 16-bit bin are candidates (slot by an LDS counter), blocks inside it go to a short LDS
//         list; the list is ranked in LDS ((key desc, block asc): exact ties go to the lowest-numbered blocks) and
//         its first `rem` entries complete the m candidates.
// A bin with more than TK_BIN blocks (heavy ties, clustered maxima) takes the general route instead: two more radix
// digits, a counting sweep and an ordered emission sweep, rewriting the strip's slots.
#ifdef COPER_DBG_TK_OVER
// diagnostic build: strips that left the fast path (a query of the strip has more than TK_BIN maxima in its threshold bin)
__device__ int g_tk_over;
extern "C" __attribute__((visibility("default"))) int coper_dbg_tk_over() {
  int v = -1, z = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_tk_over), sizeof v) != hipSuccess) return -1;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tk_over), &z, sizeof z);
  return v;
}
#endif

// A thread's walk over the coarse keys it wrote (TK_GRP): eight 16-byte loads in flight per batch -- the walk is a handful of
// groups per thread (10 on a 1.25 M-row shard, 77 on the 10M table), one load at a time it was a round trip per group and pass.
template <typename F>
__device__ __forceinline__ void tk_coarse_walk(const uint4* __restrict__ my, int64_t n_grp, F&& f) {
  constexpr int D = 8;
  int64_t jg = 0;
  for (; jg + D <= n_grp; jg += D) {
    uint4 v[D];
#pragma unroll
    for (int u = 0; u < D; ++u) v[u] = my[(jg + u) * TK_THREADS];
#pragma unroll
    for (int u = 0; u < D; ++u) f(jg + u, v[u]);
  }
  if (jg < n_grp) {
    uint4 v[D];
#pragma unroll
    for (int u = 0; u < D; ++u) v[u] = my[(jg + u < n_grp ? jg + u : n_grp - 1) * TK_THREADS];
#pragma unroll
    for (int u = 0; u < D; ++u)
      if (jg + u < n_grp) f(jg + u, v[u]);
  }
}

template <int QV, int HCOPY>
__global__ __launch_bounds__(TK_THREADS) void k_topk_threshold_emit(const float* __restrict__ gmax, int64_t G, int64_t Qs, int64_t q0,
                                                                    int64_t Bc, int k, const int64_t* __restrict__ indptr,
                                                                    int32_t* __restrict__ cand_blk, int32_t* __restrict__ cand_q,
                                                                    int32_t* __restrict__ blk_cnt, int nseg,
                                                                    uint32_t* __restrict__ cand_tau, int pair_xcd,
                                                                    uint4* __restrict__ coarse, int64_t NG) {
  constexpr int NQS = 4 * QV, SUB = TK_THREADS / QV;
  extern __shared__ uint32_t tk_lds[];
  uint32_t* hist = tk_lds;                               // [HCOPY][256 digits][NQS]
  uint32_t* s_prefix = hist + HCOPY * 256 * NQS;         // [NQS]
  uint32_t* s_rem = s_prefix + NQS;                      // [NQS]
  uint32_t* s_slot = s_rem + NQS;                        // [NQS] fast path: candidates emitted so far
  uint32_t* s_bn = s_slot + NQS;                         // [NQS] fast path: entries of the bin list
  uint32_t* s_cgt = s_bn + NQS;                          // [SUB][NQS]   (general route)
  uint32_t* s_ceq = s_cgt + SUB * NQS;                   // [SUB][NQS]
  uint32_t* s_bkey = s_ceq + SUB * NQS;                  // [NQS][TK_CL]   (fast path: TK_BIN of them)
  int32_t* s_bg = (int32_t*)(s_bkey + NQS * TK_CL);      // [NQS][TK_CL]
  const int qv = threadIdx.x % QV, sr = threadIdx.x / QV;
  // 16-query strips are 64 B of a 128-B line of gmax: the two strips of a line go to workgroups 8 apart, i.e. to the same
  // XCD (workgroup w runs on XCD w % 8) -- one L2 then fetches the line once, where neighbouring workgroups (two XCDs)
  // fetched it twice
  int64_t strip = blockIdx.x;
  if (QV == 4 && pair_xcd && (gridDim.x & 15) == 0) strip = (((int64_t)blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) * 2 + ((blockIdx.x >> 3) & 1);
  const int64_t qs0 = strip * NQS;                       // first query of the strip within the chunk
  const int64_t gs = (G + SUB - 1) / SUB;
  const int64_t g_lo = sr * gs < G ? sr * gs : G;
  const int64_t g_hi = g_lo + gs < G ? g_lo + gs : G;
  const float4* col = (const float4*)(gmax + qs0) + qv;  // + g * (Qs / 4)
  const int64_t qs4 = Qs >> 2;
  const uint32_t kinf = tk_key(-INFINITY);
  if (threadIdx.x < NQS) {
    const int64_t ql = qs0 + threadIdx.x;
    int64_t m64 = 0;
    if (ql < Bc) {
      m64 = (int64_t)k + (indptr[q0 + ql + 1] - indptr[q0 + ql]);
      if (m64 > G) m64 = G;
    }
    s_prefix[threadIdx.x] = 0;
    s_rem[threadIdx.x] = (uint32_t)m64;
    s_slot[threadIdx.x] = 0;
    s_bn[threadIdx.x] = 0;
  }
  // this thread's four queries: slot range and validity
  int64_t off[4], slots[4];
  bool valid[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t ql = qs0 + 4 * qv + c;
    valid[c] = ql < Bc;
    off[c] = slots[c] = 0;
    if (valid[c]) {
      const int64_t qg = q0 + ql;
      const int64_t beg = indptr[qg] - indptr[0];
      off[c] = (int64_t)k * qg + beg;
      slots[c] = (int64_t)k + (indptr[qg + 1] - indptr[0] - beg);
    }
  }
  // a sweep whose result does not depend on the visiting order (contiguous ranges were the A/B form of round 4)
  auto sweep_any_order = [&](auto&& f) {
    tk_sweep_strided(col, qs4, (int64_t)sr, (int64_t)SUB, G, f);
  };
  uint32_t* myhist = hist + (sr % HCOPY) * 256 * NQS + 4 * qv;
  uint32_t mask = 0;
  auto radix_pass = [&](int pass) {
    const int shift = 24 - 8 * pass;
    for (int j = threadIdx.x; j < HCOPY * 256 * NQS; j += TK_THREADS) hist[j] = 0;
    __syncthreads();
    uint32_t prefix[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) prefix[c] = s_prefix[4 * qv + c];
    // block maxima cluster: consecutive values of a query mostly share the digit (in the first pass -- sign and the upper
    // exponent bits -- nearly all of them), so a thread counts runs in registers and adds a run at once; one LDS atomic per
    // value on a handful of addresses was what the sweep spent its time on
    uint32_t run_d[4] = {0, 0, 0, 0}, run_n[4] = {0, 0, 0, 0};
    sweep_any_order([&](int64_t, const float4& v4) {
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t key = tk_key(vv[c]);
        if ((key & mask) == prefix[c]) {
          const uint32_t dgt = (key >> shift) & 255;
          if (dgt != run_d[c]) {
            if (run_n[c]) atomicAdd(&myhist[run_d[c] * NQS + c], run_n[c]);
            run_d[c] = dgt;
            run_n[c] = 0;
          }
          ++run_n[c];
        }
      }
    });
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (run_n[c]) atomicAdd(&myhist[run_d[c] * NQS + c], run_n[c]);
    __syncthreads();
    if (threadIdx.x < NQS) {
      const int qi = threadIdx.x;
      const uint32_t rem = s_rem[qi];
      if (rem > 0) {   // the digit of the rem-th largest among the keys that share the prefix
        uint32_t cum = 0;
        int dg = 255;
        for (; dg > 0; --dg) {
          uint32_t cnt = 0;
#pragma unroll
          for (int hc = 0; hc < HCOPY; ++hc) cnt += hist[(hc * 256 + dg) * NQS + qi];
          if (cum + cnt >= rem) break;
          cum += cnt;
        }
        s_prefix[qi] |= (uint32_t)dg << shift;
        s_rem[qi] = rem - cum;
      }
    }
    mask |= 0xFFu << shift;
    __syncthreads();
  };
  // ---- the coarse route (see TK_GRP above)
  bool coarse_done = false;
  if (coarse) {
    uint4* my = coarse + ((int64_t)strip * NG) * TK_THREADS + threadIdx.x;      // my group jg: my[jg * TK_THREADS]
    const int64_t n_vis = (int64_t)sr < G ? (G - sr + SUB - 1) / SUB : 0;       // my visits: blocks sr, sr + SUB, ...
    const int64_t n_grp = (n_vis + TK_GRP - 1) / TK_GRP;
    {   // sweep 1: the largest key of every TK_GRP visits
      uint32_t gm[4] = {0u, 0u, 0u, 0u};
      int64_t gj = 0;
      tk_sweep_strided(col, qs4, (int64_t)sr, (int64_t)SUB, G, [&](int64_t g, const float4& v4) {
        const int64_t jg = ((g - sr) / SUB) / TK_GRP;
        if (jg != gj) { my[gj * TK_THREADS] = make_uint4(gm[0], gm[1], gm[2], gm[3]); gm[0] = gm[1] = gm[2] = gm[3] = 0u; gj = jg; }
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t key = tk_key(vv[c]); gm[c] = key > gm[c] ? key : gm[c]; }
      });
      if (n_vis > 0) my[gj * TK_THREADS] = make_uint4(gm[0], gm[1], gm[2], gm[3]);
    }
    // the m-th largest coarse key of every query: four radix digits over the keys this thread wrote itself
    uint32_t cmask = 0;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int j = threadIdx.x; j < HCOPY * 256 * NQS; j += TK_THREADS) hist[j] = 0;
      __syncthreads();
      uint32_t prefix[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) prefix[c] = s_prefix[4 * qv + c];
      tk_coarse_walk(my, n_grp, [&](int64_t, const uint4& k4) {
        const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if ((kk[c] & cmask) == prefix[c]) atomicAdd(&myhist[((kk[c] >> shift) & 255) * NQS + c], 1u);
      });
      __syncthreads();
      if (threadIdx.x < NQS) {
        const int qi = threadIdx.x;
        const uint32_t rem = s_rem[qi];
        if (rem > 0) {
          uint32_t cum = 0;
          int dg = 255;
          for (; dg > 0; --dg) {
            uint32_t cnt = 0;
#pragma unroll
            for (int hc = 0; hc < HCOPY; ++hc) cnt += hist[(hc * 256 + dg) * NQS + qi];
            if (cum + cnt >= rem) break;
            cum += cnt;
          }
          s_prefix[qi] |= (uint32_t)dg << shift;
          s_rem[qi] = rem - cum;
        }
      }
      cmask |= 0xFFu << shift;
      __syncthreads();
    }
    // the groups that reach tau_c: their maxima >= tau_c into the query's list
    {
      uint32_t tauc[4];
      bool live[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { tauc[c] = s_prefix[4 * qv + c]; live[c] = valid[c]; }
      tk_coarse_walk(my, n_grp, [&](int64_t jg, const uint4& k4) {
        const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
        bool q_[4], any = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) { q_[c] = live[c] && kk[c] >= tauc[c]; any |= q_[c]; }
        if (!any) return;
        for (int u0 = 0; u0 < TK_GRP; u0 += 8) {      // the group's maxima again, eight loads in flight
          float4 f8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int64_t g = sr + (jg * TK_GRP + u0 + u) * (int64_t)SUB;
            f8[u] = col[(g < G ? g : (int64_t)sr) * qs4];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int64_t g = sr + (jg * TK_GRP + u0 + u) * (int64_t)SUB;
            if (g >= G) continue;
            const float vv[4] = {f8[u].x, f8[u].y, f8[u].z, f8[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (!q_[c]) continue;
              const uint32_t key = tk_key(vv[c]);
              if (key >= tauc[c]) {
                const uint32_t i = atomicAdd(&s_bn[4 * qv + c], 1u);
                if (i < (uint32_t)TK_CL) { s_bkey[(4 * qv + c) * TK_CL + i] = key; s_bg[(4 * qv + c) * TK_CL + i] = (int32_t)g; }
              }
            }
          }
        }
      });
    }
    __syncthreads();
    bool over_c = false;
    if (threadIdx.x < NQS) over_c = s_bn[threadIdx.x] > (uint32_t)TK_CL;
    if (!__syncthreads_or(over_c ? 1 : 0)) {
      // rank the lists: entry e of query qi is candidate number `ahead` if ahead < m
      for (int t = threadIdx.x; t < NQS * TK_CL; t += TK_THREADS) {
        const int qi = t / TK_CL, e = t % TK_CL;
        const int64_t ql = qs0 + qi;
        const uint32_t nb = s_bn[qi];
        if (ql >= Bc || (uint32_t)e >= nb) continue;
        const int64_t qg = q0 + ql;
        const int64_t beg = indptr[qg] - indptr[0];
        const int64_t nslots = (int64_t)k + (indptr[qg + 1] - indptr[0] - beg);
        const uint32_t m = (uint32_t)(nslots < G ? nslots : G);
        const uint32_t key = s_bkey[qi * TK_CL + e];
        const int32_t g = s_bg[qi * TK_CL + e];
        uint32_t ahead = 0;
        for (uint32_t t2 = 0; t2 < nb; ++t2) {
          const uint32_t k2 = s_bkey[qi * TK_CL + t2];
          ahead += (k2 > key || (k2 == key && s_bg[qi * TK_CL + t2] < g)) ? 1u : 0u;
        }
        if (ahead >= m) continue;
        const int64_t o = (int64_t)k * qg + beg;
        cand_blk[o + ahead] = key > kinf ? g : -1;       // blocks whose maximum is -inf hold nothing: their slots stay unused
        if (ahead == m - 1) cand_tau[qg] = (nslots <= G && key > kinf) ? key : 0u;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!valid[c]) continue;
        const int qi = 4 * qv + c;
        const int64_t m = slots[c] < G ? slots[c] : G;
        for (int64_t j = sr; j < slots[c]; j += SUB) {
          cand_q[off[c] + j] = (int32_t)(q0 + qs0 + qi);
          if (j >= m) cand_blk[off[c] + j] = -1;
        }
      }
      coarse_done = true;
    } else {
      // a list overflowed: the strip takes the three-sweep route from the start
      if (threadIdx.x < NQS) {
        const int64_t ql = qs0 + threadIdx.x;
        int64_t m64 = 0;
        if (ql < Bc) {
          m64 = (int64_t)k + (indptr[q0 + ql + 1] - indptr[q0 + ql]);
          if (m64 > G) m64 = G;
        }
        s_prefix[threadIdx.x] = 0;
        s_rem[threadIdx.x] = (uint32_t)m64;
        s_slot[threadIdx.x] = 0;
        s_bn[threadIdx.x] = 0;
      }
      __syncthreads();
    }
  }
  if (!coarse_done) {
  radix_pass(0);
  radix_pass(1);

  // ---- sweep 3: above the 16-bit bin -> candidate; inside it -> LDS list
  {
    uint32_t p16[4];
    bool live[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      p16[c] = s_prefix[4 * qv + c] >> 16;
      live[c] = valid[c] && s_rem[4 * qv + c] > 0;
    }
    sweep_any_order([&](int64_t g, const float4& v4) {
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!live[c]) continue;
        const uint32_t key = tk_key(vv[c]);
        const uint32_t k16 = key >> 16;
        if (k16 > p16[c]) {
          cand_blk[off[c] + atomicAdd(&s_slot[4 * qv + c], 1u)] = (int32_t)g;
        } else if (k16 == p16[c]) {
          const uint32_t i = atomicAdd(&s_bn[4 * qv + c], 1u);
          if (i < (uint32_t)TK_BIN) { s_bkey[(4 * qv + c) * TK_BIN + i] = key; s_bg[(4 * qv + c) * TK_BIN + i] = (int32_t)g; }
        }
      }
    });
  }
  __syncthreads();
  bool over = false;
  if (threadIdx.x < NQS) over = s_bn[threadIdx.x] > (uint32_t)TK_BIN;
  if (!__syncthreads_or(over ? 1 : 0)) {
    // rank the bin lists: entry e of query qi is candidate number (gt + ahead) if ahead < rem
    for (int t = threadIdx.x; t < NQS * TK_BIN; t += TK_THREADS) {
      const int qi = t / TK_BIN, e = t % TK_BIN;
      const int64_t ql = qs0 + qi;
      const uint32_t nb = s_bn[qi], rem = s_rem[qi];
      if (ql >= Bc || (uint32_t)e >= nb || rem == 0) continue;
      const uint32_t key = s_bkey[qi * TK_BIN + e];
      const int32_t g = s_bg[qi * TK_BIN + e];
      uint32_t ahead = 0;
      for (uint32_t t2 = 0; t2 < nb; ++t2) {
        const uint32_t k2 = s_bkey[qi * TK_BIN + t2];
        ahead += (k2 > key || (k2 == key && s_bg[qi * TK_BIN + t2] < g)) ? 1u : 0u;
      }
      if (ahead >= rem) continue;
      const int64_t qg = q0 + ql;
      const int64_t beg = indptr[qg] - indptr[0];
      const int64_t o = (int64_t)k * qg + beg;
      const int64_t nslots = (int64_t)k + (indptr[qg + 1] - indptr[0] - beg);
      // blocks whose maximum is -inf hold nothing: their slots stay unused
      cand_blk[o + s_slot[qi] + ahead] = key > kinf ? g : -1;
      // what the selection may discard unseen: logits below tau, when tau is a real m-th largest (at least k unmasked
      // logits >= tau exist then); with every block a candidate (m clamped to G) or a -inf tau nothing is discarded
      if (ahead == rem - 1) cand_tau[qg] = (nslots <= G && key > kinf) ? key : 0u;
    }
    // query ids of all slots; slots past the m candidates (m clamped to the number of blocks) are unused
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!valid[c]) continue;
      const int qi = 4 * qv + c;
      const int64_t used = (int64_t)s_slot[qi] + s_rem[qi];
      for (int64_t j = sr; j < slots[c]; j += SUB) {
        cand_q[off[c] + j] = (int32_t)(q0 + qs0 + qi);
        if (j >= used) cand_blk[off[c] + j] = -1;
      }
    }
  } else {
#ifdef COPER_DBG_TK_OVER
    if (threadIdx.x == 0) atomicAdd(&g_tk_over, 1);
#endif
    // ---- general route: the remaining two digits, then count and emit in block order
    radix_pass(2);
    radix_pass(3);
    uint32_t tau[4], cgt[4] = {0, 0, 0, 0}, ceq[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) tau[c] = s_prefix[4 * qv + c];
    tk_sweep(col, qs4, g_lo, g_hi, [&](int64_t, const float4& v4) {
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t key = tk_key(vv[c]);
        cgt[c] += key > tau[c] ? 1u : 0u;
        ceq[c] += key == tau[c] ? 1u : 0u;
      }
    });
#pragma unroll
    for (int c = 0; c < 4; ++c) { s_cgt[sr * NQS + 4 * qv + c] = cgt[c]; s_ceq[sr * NQS + 4 * qv + c] = ceq[c]; }
    __syncthreads();
    uint32_t bgt[4], beq[4], c1[4], need[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int qi = 4 * qv + c;
      bgt[c] = beq[c] = c1[c] = 0;
      for (int s2 = 0; s2 < SUB; ++s2) {
        const uint32_t x = s_cgt[s2 * NQS + qi];
        if (s2 < sr) { bgt[c] += x; beq[c] += s_ceq[s2 * NQS + qi]; }
        c1[c] += x;
      }
      need[c] = tau[c] > kinf ? s_rem[qi] : 0u;
      if (valid[c]) {
        if (sr == 0) cand_tau[q0 + qs0 + qi] = (slots[c] <= G && tau[c] > kinf) ? tau[c] : 0u;
        for (int64_t j = sr; j < slots[c]; j += SUB) {
          cand_q[off[c] + j] = (int32_t)(q0 + qs0 + qi);
          if (j >= (int64_t)(c1[c] + need[c])) cand_blk[off[c] + j] = -1;
        }
      }
    }
    tk_sweep(col, qs4, g_lo, g_hi, [&](int64_t g, const float4& v4) {
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!valid[c]) continue;
        const uint32_t key = tk_key(vv[c]);
        if (key > tau[c]) {
          cand_blk[off[c] + bgt[c]++] = (int32_t)g;
        } else if (key == tau[c]) {
          if (beq[c] < need[c]) cand_blk[off[c] + c1[c] + beq[c]] = (int32_t)g;
          ++beq[c];
        }
      }
    });
  }
  }   // (!coarse_done)
  // ---- candidates are scored block by block: how many slots want block g (nseg counters per block thin out the
  // same-address atomics when there are few blocks)
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (!valid[c] || !blk_cnt) continue;      // (blk_cnt == NULL: 64-entity candidates, counted after their expansion: k_topk_expand64)
    for (int64_t j = sr; j < slots[c]; j += SUB) {
      const int32_t g = cand_blk[off[c] + j];
      if (g >= 0) atomicAdd(&blk_cnt[(int64_t)g * nseg + ((off[c] + j) & (nseg - 1))], 1);
    }
  }
}

// Large tables (topk_expand == 2): the threshold kernel worked on 64-entity block maxima; every candidate slot becomes two slots, the block's two
// 32-entity halves (2g, 2g + 1) -- what the grouping, the re-scoring and the selection below are written for.  A query's slots
// stay contiguous: [2 off, 2 (off + n)).
__global__ __launch_bounds__(256) void k_topk_expand64(const int32_t* __restrict__ blk64, const int32_t* __restrict__ q64, int64_t T64,
                                                       int32_t* __restrict__ cand_blk, int32_t* __restrict__ cand_q,
                                                       int32_t* __restrict__ blk_cnt, int nseg) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= T64) return;
  const int32_t g = blk64[j], q = q64[j];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t j2 = 2 * j + u;
    cand_blk[j2] = g >= 0 ? 2 * g + u : -1;
    cand_q[j2] = q;
    if (g >= 0) atomicAdd(&blk_cnt[(int64_t)(2 * g + u) * nseg + (j2 & (nseg - 1))], 1);
  }
}

template <int QV, int HCOPY>
constexpr size_t tk_emit_lds() {
  return sizeof(uint32_t) * (HCOPY * 256 * 4 * QV + 4 * 4 * QV + 2 * (TK_THREADS / QV) * 4 * QV + 2 * 4 * QV * TK_CL);
}

// ---- group the candidate slots by entity block: every block's slots padded to a multiple of 32 (one wave each)
// blk_off[g] = first position of block g in `sorted`, blk_off[G] = total (a multiple of 32)
__global__ __launch_bounds__(1024) void k_topk_blk_scan(const int32_t* __restrict__ blk_cnt, int64_t G, int32_t* __restrict__ blk_off) {
  __shared__ int32_t part[1024];
  const int64_t per = (G + 1023) / 1024;
  const int64_t lo_g = threadIdx.x * per, hi_g = lo_g + per < G ? lo_g + per : G;
  int32_t sum = 0;
  for (int64_t g = lo_g; g < hi_g; ++g) sum += (blk_cnt[g] + 31) & ~31;
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // inclusive scan
    const int32_t add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int32_t run = part[threadIdx.x] - sum;
  for (int64_t g = lo_g; g < hi_g; ++g) { blk_off[g] = run; run += (blk_cnt[g] + 31) & ~31; }
  if (threadIdx.x == 1023) blk_off[G] = part[1023];
}

// Long block axes (the 10M-entity table: 312,500 counters): the same exclusive scan in three coalesced launches -- chunk sums,
// a scan of the chunk sums, chunk-local scans.  The single workgroup above walks 305 counters per thread with a stride of
// 1.2 KB between lanes: 0.58 ms; these take ~0.02 ms together.
constexpr int TK_SCAN_CHUNK = 4096;   // counters per workgroup: 256 threads x 16
__global__ __launch_bounds__(256) void k_topk_blk_chunk_sums(const int32_t* __restrict__ blk_cnt, int64_t G, int32_t* __restrict__ chunk_sum) {
  __shared__ int32_t red[256];
  const int64_t base = (int64_t)blockIdx.x * TK_SCAN_CHUNK;
  int32_t sum = 0;
#pragma unroll
  for (int j = 0; j < TK_SCAN_CHUNK / 256; ++j) {
    const int64_t g = base + threadIdx.x + 256 * j;
    if (g < G) sum += (blk_cnt[g] + 31) & ~31;
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = red[0];
}

// exclusive scan of the chunk sums in place (one workgroup; any number of chunks), total -> blk_off[G]
__global__ __launch_bounds__(1024) void k_topk_blk_chunk_scan(int32_t* __restrict__ chunk_sum, int nchunk, int32_t* __restrict__ total_out) {
  __shared__ int32_t part[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nchunk; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int32_t v = i < nchunk ? chunk_sum[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int32_t add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nchunk) chunk_sum[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_topk_blk_chunk_apply(const int32_t* __restrict__ blk_cnt, int64_t G, const int32_t* __restrict__ chunk_off,
                                                              int32_t* __restrict__ blk_off) {
  __shared__ int32_t v[TK_SCAN_CHUNK];
  __shared__ int32_t tsum[256];
  const int64_t base = (int64_t)blockIdx.x * TK_SCAN_CHUNK;
#pragma unroll
  for (int j = 0; j < TK_SCAN_CHUNK / 256; ++j) {
    const int64_t g = base + threadIdx.x + 256 * j;
    v[threadIdx.x + 256 * j] = g < G ? ((blk_cnt[g] + 31) & ~31) : 0;
  }
  __syncthreads();
  // thread t owns counters [16 t, 16 t + 16) of the chunk
  int32_t loc[16], s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) { loc[j] = s; s += v[16 * threadIdx.x + j]; }
  tsum[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int32_t add = (int)threadIdx.x >= o ? tsum[threadIdx.x - o] : 0;
    __syncthreads();
    tsum[threadIdx.x] += add;
    __syncthreads();
  }
  const int32_t off = chunk_off[blockIdx.x] + tsum[threadIdx.x] - s;
#pragma unroll
  for (int j = 0; j < 16; ++j) v[16 * threadIdx.x + j] = off + loc[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TK_SCAN_CHUNK / 256; ++j) {
    const int64_t g = base + threadIdx.x + 256 * j;
    if (g < G) blk_off[g] = v[threadIdx.x + 256 * j];
  }
}

// blk_off[0..GV] from blk_cnt[0..GV): one workgroup for short axes, three coalesced launches for long ones (chunk sums in `tmp`)
static void tk_launch_blk_scan(const int32_t* blk_cnt, int64_t GV, int32_t* blk_off, int32_t* tmp, int64_t tmp_cap, hipStream_t s) {
  const int64_t nchunk = (GV + TK_SCAN_CHUNK - 1) / TK_SCAN_CHUNK;
  if (GV <= 4 * TK_SCAN_CHUNK || !tmp || nchunk > tmp_cap) {
    hipLaunchKernelGGL(k_topk_blk_scan, dim3(1), dim3(1024), 0, s, blk_cnt, GV, blk_off);
    return;
  }
  hipLaunchKernelGGL(k_topk_blk_chunk_sums, dim3((unsigned)nchunk), dim3(256), 0, s, blk_cnt, GV, tmp);
  hipLaunchKernelGGL(k_topk_blk_chunk_scan, dim3(1), dim3(1024), 0, s, tmp, (int)nchunk, blk_off + GV);
  hipLaunchKernelGGL(k_topk_blk_chunk_apply, dim3((unsigned)nchunk), dim3(256), 0, s, blk_cnt, GV, tmp, blk_off);
}

__global__ __launch_bounds__(256) void k_topk_blk_scatter(const int32_t* __restrict__ cand_blk, int64_t T, const int32_t* __restrict__ blk_off,
                                                          int32_t* __restrict__ blk_cur, int nseg, int32_t* __restrict__ sorted) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= T) return;
  const int32_t g = cand_blk[w];
  if (g < 0) return;
  const int64_t vb = (int64_t)g * nseg + (w & (nseg - 1));
  sorted[blk_off[vb] + atomicAdd(&blk_cur[vb], 1)] = (int32_t)w;
}

// One wave per 32 candidate slots of one entity block: A = the block's fragments (read as the count pass reads them),
// B column c = the query of slot c (gathered from the row-major planes, as the pair kernel does).  Lane (c, half)
// ends with 16 rows of its slot's 32 logits; known answers of the slot's query, except its target, become -inf.
__global__ __launch_bounds__(256) void k_topk_score_blocks(const uint4* __restrict__ Ehi, const uint4* __restrict__ Elo,
                                                           const float* __restrict__ bias_pad, const uint4* __restrict__ Hrm_hi,
                                                           const uint4* __restrict__ Hrm_lo, int KS, int64_t G,
                                                           const int64_t* __restrict__ e2, const int64_t* __restrict__ indptr,
                                                           const int64_t* __restrict__ idx, const int32_t* __restrict__ cand_blk,
                                                           const int32_t* __restrict__ cand_q, const int32_t* __restrict__ blk_off,
                                                           const int32_t* __restrict__ sorted, int64_t lo,
                                                           float* __restrict__ cand_val, const int32_t* __restrict__ x3s) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
  const int sexp = x3s[1];   // the accumulators' power of two (split16.h): candidate values stay in the units of the block maxima
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i * 32 >= blk_off[G]) return;   // G here: number of (block, segment) counters
  const int64_t g = cand_blk[sorted[i * 32]];   // the first slot of a wave's 32 is always in use
  const int32_t w = sorted[i * 32 + c];          // -1: padding
  const int64_t q = w >= 0 ? cand_q[w] : 0;
  f32x16 acc;
  {
    const float4* bp = (const float4*)(bias_pad + g * 32 + 4 * half);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b4 = bp[2 * j];
      acc[4 * j + 0] = x3_scale(b4.x, sexp); acc[4 * j + 1] = x3_scale(b4.y, sexp);
      acc[4 * j + 2] = x3_scale(b4.z, sexp); acc[4 * j + 3] = x3_scale(b4.w, sexp);
    }
  }
  const uint4* pa_h = Ehi + g * KS * 64 + lane;
  const uint4* pa_l = Elo + g * KS * 64 + lane;
  const uint4* pb_h = Hrm_hi + q * (2 * KS) + half;
  const uint4* pb_l = Hrm_lo + q * (2 * KS) + half;
  // pairs of k-steps, then the last one of an odd count (bf16x3_chain.h); batches of two pairs
  for (int ks = 0; ks < KS; ks += 4) {
    uint4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = ks + u < KS ? ks + u : KS - 1;
      ah[u] = pa_h[k * 64]; al[u] = pa_l[k * 64];
      bh[u] = pb_h[k * 2]; bl[u] = pb_l[k * 2];
    }
#pragma unroll
    for (int u = 0; u < 4; u += 2) {   // wave-uniform
      if (ks + u + 1 < KS) { BX3_PAIR(ah[u], al[u], bh[u], bl[u], ah[u + 1], al[u + 1], bh[u + 1], bl[u + 1], acc); }
      else if (ks + u < KS) { BX3_LAST(ah[u], al[u], bh[u], bl[u], acc); }
    }
  }
  if (w < 0) return;
  // known answers of this lane's query inside the block, except the target (metrics.py:45-46)
  const int64_t target = e2[q];
  uint32_t masked = 0;
  for (int64_t j = indptr[q]; j < indptr[q + 1]; ++j) {
    const int64_t f = idx[j];
    const int64_t r = f - lo - g * 32;
    if (f != target && r >= 0 && r < 32) masked |= 1u << (int)r;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 v = make_float4(acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    const int row = 8 * j + 4 * half;
    if ((masked >> (row + 0)) & 1u) v.x = -INFINITY;
    if ((masked >> (row + 1)) & 1u) v.y = -INFINITY;
    if ((masked >> (row + 2)) & 1u) v.z = -INFINITY;
    if ((masked >> (row + 3)) & 1u) v.w = -INFINITY;
    *(float4*)(cand_val + (int64_t)w * 32 + row) = v;
  }
}

// One wave per query.  Only candidates >= tau (cand_tau, from the threshold kernel) can be in the top-k: they are
// compacted into LDS (a few dozen out of (k + filter entries) * 32) and placed by counting, for each survivor, the
// survivors ahead of it in (score desc, entity id asc) order.  More survivors than the LDS list holds (tiny entity
// sets where tau is -inf, heavy ties): k rounds of arg-max through memory instead.
constexpr int TK_SURV = 256;   // survivors per query held in LDS
__global__ __launch_bounds__(256) void k_topk_select_cand(float* __restrict__ cand_val, const int32_t* __restrict__ cand_blk,
                                                          const uint32_t* __restrict__ cand_tau, const int64_t* __restrict__ indptr,
                                                          int64_t B, int k, int64_t lo, float* __restrict__ out_val,
                                                          int64_t* __restrict__ out_idx, const int32_t* __restrict__ x3s, int xf) {
  __shared__ float s_v[4][TK_SURV];
  const int dexp = x3s ? -x3s[1] : 0;   // x3 mode: candidate values carry 2^(e_E + e_h); what is written out does not
  __shared__ int s_id[4][TK_SURV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  if (q >= B) return;
  const int64_t beg = indptr[q] - indptr[0];
  const int64_t off = xf * ((int64_t)k * q + beg);           // (xf = 2: slots expanded from 64-entity candidates)
  const int n = xf * (int)((int64_t)k + (indptr[q + 1] - indptr[0] - beg)) * 32;
  float* val = cand_val + off * 32;
  const int32_t* blk = cand_blk + off;
  float* ov = out_val + q * k;
  int64_t* oi = out_idx + q * k;
  const uint32_t tau = cand_tau[q];
  int S = 0;   // survivors so far (wave-uniform)
  for (int j0 = 0; j0 < n; j0 += 64 * 4) {
    float v[4];
    int32_t b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 64 * u + lane;
      b[u] = j < n ? blk[j >> 5] : -1;
      v[u] = b[u] >= 0 ? val[j] : -INFINITY;   // unused slots (block -1) hold nothing
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool keep = v[u] > -INFINITY && tk_key(v[u]) >= tau;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = S + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < TK_SURV) { s_v[wave][pos] = v[u]; s_id[wave][pos] = b[u] * 32 + ((j0 + 64 * u + lane) & 31); }
      }
      S += __popcll(m);
    }
  }
  if (S <= TK_SURV) {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the list is wave-local
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += 64) {
      const float v = s_v[wave][i];
      const int id = s_id[wave][i];
      int ahead = 0;
      for (int t = 0; t < S; ++t) {
        const float v2 = s_v[wave][t];
        const int i2 = s_id[wave][t];
        ahead += (v2 > v || (v2 == v && i2 < id)) ? 1 : 0;
      }
      if (ahead < k) { ov[ahead] = x3_scale(v, dexp); oi[ahead] = lo + id; }
    }
    for (int r = S + lane; r < k; r += 64) { ov[r] = -INFINITY; oi[r] = -1; }
    return;
  }
  for (int round = 0; round < k; ++round) {
    float best = -INFINITY;
    int bid = 0x7fffffff, bpos = -1;
    for (int j = lane; j < n; j += 64) {
      const int32_t b = blk[j >> 5];
      if (b < 0) continue;
      const float v = val[j];
      if (!(v > -INFINITY)) continue;
      const int id = b * 32 + (j & 31);
      if (v > best || (v == best && id < bid)) { best = v; bid = id; bpos = j; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float v2 = __shfl_xor(best, o);
      const int i2 = __shfl_xor(bid, o);
      const int p2 = __shfl_xor(bpos, o);
      if (v2 > best || (v2 == best && i2 < bid)) { best = v2; bid = i2; bpos = p2; }
    }
    // every lane holds the winner now; the lane that read it retires it (its own later reads see its own store)
    if (lane == 0) {
      ov[round] = bpos >= 0 ? x3_scale(best, dexp) : -INFINITY;
      oi[round] = bpos >= 0 ? lo + bid : -1;
    }
    if (bpos >= 0 && (bpos & 63) == lane) val[bpos] = -INFINITY;
  }
}

static bool tk_pair_xcd() {
  static const bool on = getenv("COPER_TK_NO_XCD_PAIRS") == nullptr;   // A/B switch
  return on;
}

template <int QV, int HCOPY>
static void tk_launch_emit(coper_handle* h, int64_t G, int64_t qs, int64_t q0, int64_t bc, int k, const int64_t* indptr, hipStream_t s,
                           int32_t* out_blk, int32_t* out_q, int32_t* out_cnt) {
  const size_t lds = tk_emit_lds<QV, HCOPY>();
  static uint64_t attr_done = 0;   // per instantiation, one bit per device (function attributes are per device)
  const uint64_t bit = 1ull << (h->cfg.device & 63);
  if (!(attr_done & bit)) {
    (void)hipFuncSetAttribute((const void*)k_topk_threshold_emit<QV, HCOPY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= bit;
  }
  // the coarse route (TK_GRP): long block axes, when the scratch was reserved for this (G, qs)
  static const bool no_coarse = getenv("COPER_TK_NO_COARSE") != nullptr;     // A/B switch, read once
  constexpr int64_t SUB = TK_THREADS / QV;
  const int64_t NG = ((G + SUB - 1) / SUB + TK_GRP - 1) / TK_GRP, strips = qs / (4 * QV);
  const size_t cbytes = (size_t)strips * NG * TK_THREADS * sizeof(uint4);
  uint4* coarse = (!no_coarse && G >= TK_COARSE_MIN_BLOCKS && h->tk_coarse_ws && cbytes <= h->tk_coarse_cap) ? (uint4*)h->tk_coarse_ws : nullptr;
  hipLaunchKernelGGL((k_topk_threshold_emit<QV, HCOPY>), dim3((unsigned)(qs / (4 * QV))), dim3(TK_THREADS), lds, s, h->gmax_ws, G, qs, q0, bc,
                     k, indptr, out_blk, out_q, out_cnt, topk_nseg(G), h->cand_tau_ws, tk_pair_xcd() ? 1 : 0, coarse, NG);
}

// strip width / histogram copies of the threshold kernel by shape
static void tk_dispatch_emit(coper_handle* h, int64_t G, int64_t qs, int64_t q0, int64_t bc, int k, const int64_t* indptr, hipStream_t s,
                             int32_t* out_blk = nullptr, int32_t* out_q = nullptr, int32_t* out_cnt = nullptr) {
  if (!out_blk) { out_blk = h->cand_blk_ws; out_q = h->cand_q_ws; out_cnt = h->blk_cnt_ws; }
  static const char* force = getenv("COPER_TK_EMIT");   // experiments: "8_1", "8_2", "4_4"
  if (force && force[0] == '8' && force[2] == '1') return tk_launch_emit<8, 1>(h, G, qs, q0, bc, k, indptr, s, out_blk, out_q, out_cnt);
  if (force && force[0] == '8' && force[2] == '2') return tk_launch_emit<8, 2>(h, G, qs, q0, bc, k, indptr, s, out_blk, out_q, out_cnt);
  if (force && force[0] == '4') return tk_launch_emit<4, 4>(h, G, qs, q0, bc, k, indptr, s, out_blk, out_q, out_cnt);
  // long block axis: more histogram copies; half-width strips (twice the workgroups) when 32-query strips would
  // not cover the chip
  if (G < 4096) tk_launch_emit<8, 1>(h, G, qs, q0, bc, k, indptr, s, out_blk, out_q, out_cnt);
  else if (qs / 32 >= h->num_cus) tk_launch_emit<8, 2>(h, G, qs, q0, bc, k, indptr, s, out_blk, out_q, out_cnt);
  else tk_launch_emit<4, 4>(h, G, qs, q0, bc, k, indptr, s, out_blk, out_q, out_cnt);
}

int launch_topk_pruned_bf16x3(coper_handle* h, const float* hvec, const float* tgt_x, const int64_t* e2, const int64_t* indptr,
                              const int64_t* idx, int64_t nnz, int64_t B, int k, int32_t* ng, int32_t* ne, float* topk_val,
                              int64_t* topk_idx, hipStream_t s) {
  const Dims& dm = h->dm;
  const int XF = topk_expand(h);            // 2: the count kernel writes 64-entity maxima (large tables), candidates are expanded below
  const int64_t G = dm.n_eblk, Gm = G / XF;
  const int64_t qc = topk_chunk_queries(G, B, h->gmax_max_floats);
  const int64_t T64 = (int64_t)k * B + nnz;   // candidate blocks: k + (filter entries) per query
  const int64_t T = XF * T64;                 // ... as 32-entity slots
  if ((size_t)(Gm * qc) > h->gmax_cap || (size_t)T > h->cand_cap)
    return fail(h, COPER_ESTATE, "pruned top-k: workspace not reserved");
  int rc;
  score_count_begin_bf16x3(h, B, ng, ne, s);
  // slots no query owns (filt_nnz may be a capacity larger than the CSR) must read as unused
  COPER_HIP_TRY(h, hipMemsetAsync(h->cand_blk_ws, 0xFF, sizeof(int32_t) * (T + (XF > 1 ? T64 : 0)), s));
  const int nseg = topk_nseg(G);
  const int64_t GV = G * nseg;   // (block, segment) counters
  COPER_HIP_TRY(h, hipMemsetAsync(h->blk_cnt_ws, 0, sizeof(int32_t) * 2 * GV, s));          // counts | scatter cursors
  COPER_HIP_TRY(h, hipMemsetAsync(h->cand_sorted_ws, 0xFF, sizeof(int32_t) * topk_sorted_cap(GV, T), s));
  int32_t* blk64 = XF > 1 ? h->cand_blk_ws + T : nullptr;      // the 64-entity level's own lists, behind the expanded ones
  int32_t* q64 = XF > 1 ? h->cand_q_ws + T : nullptr;
  for (int64_t q0 = 0; q0 < B; q0 += qc) {
    const int64_t bc = B - q0 < qc ? B - q0 : qc;
    const int64_t qs = (bc + 127) / 128 * 128;
    if ((rc = score_count3_chunk_bf16x3(h, q0, bc, hvec, tgt_x, e2, indptr, idx, ng, ne, h->gmax_ws, qs, s))) return rc;
    if (XF > 1) tk_dispatch_emit(h, Gm, qs, q0, bc, k, indptr, s, blk64, q64, nullptr);
    else tk_dispatch_emit(h, G, qs, q0, bc, k, indptr, s);
  }
  if (XF > 1)
    hipLaunchKernelGGL(k_topk_expand64, dim3((unsigned)((T64 + 255) / 256)), dim3(256), 0, s, blk64, q64, T64, h->cand_blk_ws, h->cand_q_ws,
                       h->blk_cnt_ws, nseg);
  tk_launch_blk_scan(h->blk_cnt_ws, GV, h->blk_off_ws, h->blk_off_ws + GV + 1, GV / TK_SCAN_CHUNK + 2, s);   // (chunk sums behind blk_off: reserved with it)
  hipLaunchKernelGGL(k_topk_blk_scatter, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, h->cand_blk_ws, T, h->blk_off_ws,
                     h->blk_cnt_ws + GV, nseg, h->cand_sorted_ws);
  const int64_t waves = topk_sorted_cap(GV, T) / 32;
  hipLaunchKernelGGL(k_topk_score_blocks, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const uint4*)h->Ef16_hi,
                     (const uint4*)h->Ef16_lo, h->bias_pad, (const uint4*)h->hrm16_hi, (const uint4*)h->hrm16_lo, dm.KS16, GV, e2,
                     indptr, idx, h->cand_blk_ws, h->cand_q_ws, h->blk_off_ws, h->cand_sorted_ws, (int64_t)h->cfg.shard_lo,
                     h->cand_val_ws, h->x3s);
  hipLaunchKernelGGL(k_topk_select_cand, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, h->cand_val_ws, h->cand_blk_ws, h->cand_tau_ws, indptr, B,
                     k, (int64_t)h->cfg.shard_lo, topk_val, topk_idx, h->x3s, XF);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// fp32-exact mode: same threshold / selection kernels on the block maxima of k_score_count_f32; candidates are
// rescored by the VALU chain (kernels_score.hip), which needs no grouping by block
int launch_topk_pruned_f32(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2, const int64_t* indptr,
                           const int64_t* idx, int64_t nnz, int64_t B, int k, int32_t* ng, int32_t* ne, float* topk_val,
                           int64_t* topk_idx, hipStream_t s) {
  const Dims& dm = h->dm;
  const int64_t G = dm.n_eblk;
  const int64_t qc = topk_chunk_queries(G, B, h->gmax_max_floats);
  const int64_t T = (int64_t)k * B + nnz;
  if ((size_t)(G * qc) > h->gmax_cap || (size_t)T > h->cand_cap)
    return fail(h, COPER_ESTATE, "pruned top-k: workspace not reserved");
  int rc;
  score_count_begin_f32(h, hvec, B, ng, ne, s);
  COPER_HIP_TRY(h, hipMemsetAsync(h->cand_blk_ws, 0xFF, sizeof(int32_t) * T, s));
  COPER_HIP_TRY(h, hipMemsetAsync(h->blk_cnt_ws, 0, sizeof(int32_t) * 2 * G * topk_nseg(G), s));
  for (int64_t q0 = 0; q0 < B; q0 += qc) {
    const int64_t bc = B - q0 < qc ? B - q0 : qc;
    const int64_t qs = (bc + 127) / 128 * 128;
    if ((rc = score_count_chunk_f32(h, q0, bc, tgt, ng, ne, h->gmax_ws, qs, s))) return rc;
    tk_dispatch_emit(h, G, qs, q0, bc, k, indptr, s);
  }
  if ((rc = launch_topk_score_blocks_f32(h, hvec, T, e2, indptr, idx, s))) return rc;
  hipLaunchKernelGGL(k_topk_select_cand, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, h->cand_val_ws, h->cand_blk_ws, h->cand_tau_ws, indptr, B,
                     k, (int64_t)h->cfg.shard_lo, topk_val, topk_idx, (const int32_t*)nullptr, 1);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
