// COPER_SCORE_BF16X3 encoder, fused conv + generated dense for every tile (1..128 queries of one relation).
//
// The unfused pair (k_conv3x3_bn_relu_bf16 -> x planes in HBM -> k_dense_reg_bf16x3) is bound by the x
// traffic: 64-B pieces of 16 different rows per instruction make x 42 % of the dense kernel's memory
// requests for 26 % of its bytes, and the conv kernel spends its time writing those bytes.  Here a workgroup
// (one relation tile x one K slice) keeps the slice's rows of the e1 images in LDS and produces each k-step's
// x fragments itself -- 3x3 conv, BN, ReLU, hi/lo split -- straight into the B-operand layout in LDS, one
// k-step ahead of the MFMAs that consume them.  HBM sees the weight stream (read once, non-temporal when one tile reads a set, PF
// k-steps ahead in registers of the owning wave, as in k_dense_reg_bf16x3), the gathered e1 rows and z.
//
// Supported when the filter is 3x3 with C = 32 channels (one k-step of 32 features = one output pixel), no
// concat_rel tail, and the conv filters are either shared or generated together with the dense weights (a
// tile is then one relation).  Arithmetic per x value is conv_x8() below, shared with the stand-alone conv
// kernel that still serves the <= 32-query tiles: x, and with it h[b], stays a pure function of (e1, rel).
#include "bf16x3_chain.h"
#include "coper_internal.h"
#ifdef COPER_DBG_GROUP_CLK
#include <hip/hip_runtime.h>
__device__ unsigned long long g_group_clk[8];
extern "C" __attribute__((visibility("default"))) int coper_dbg_group_clock(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_group_clk), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : 1;
}
#define GROUP_CLK(i) do { __syncthreads(); if (threadIdx.x == 0) g_group_clk[i] = wall_clock64(); } while (0)
#endif
#include "group_body.h"
#include "conv_fold.h"

namespace coper {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16_BF16(a, b, c) S16_MFMA16(a, b, c)
#define MFMA16_X3(ahi, alo, bhi, blo, c) \
  { (c) = MFMA16_BF16(alo, bhi, c); (c) = MFMA16_BF16(ahi, blo, c); (c) = MFMA16_BF16(ahi, bhi, c); }


#ifndef COPER_FUSED_PMAX
#define COPER_FUSED_PMAX 3
#endif
#ifndef COPER_FUSED_PBUDGET
// Register budget that decides the weight prefetch depth P per tile size (fused_matrix_role).  Round 5: 216 -> 220, which gives
// the 81..96-query tiles (six query blocks: 147 of the 237 tiles of an FB15k-237 pass) THREE k-steps of weights in flight
// instead of two -- the launch is bound by what one CU keeps in flight (16 -> 24 KB per wave; its tiles all take ~200 us whatever
// their size): fused encoder 0.1768 -> 0.1647 ms, pass 0.5115 -> 0.498 ms on one box (profiles/r05b_experiments.txt).  232 (seven
// blocks too) needs the x fragments requested one block ahead instead of two to stay out of scratch, and measures no gain.
#define COPER_FUSED_PBUDGET 220
#endif


// (Round 4 measured the weights' second term streamed as BYTES -- 3 bytes per value instead of 4, rebuilt as fp16 in front of the
// k-step's MFMAs: 25 % fewer bytes bought nothing, 0.1764 ms against 0.173 - 0.182, and cost 0.2 % of the queries their
// float64-equal rank; the build switch is gone, the story is in DESIGN_LOG.md.)
#define FW_DECL u32x4 W[P][NW][2]
#define FW_HI(s_, j_) W[s_][j_][0]
#define FW_LO(s_, j_) W[s_][j_][1]
#define FW_LO_RAW(s_, j_) W[s_][j_][1]
#define FW_PRE(s_)
#define FW_WREG3 (3 * NW * 8)

struct FusedConvArgs {
  const float* e1_rows;
  const int32_t* sorted_row;
  const int32_t* sorted_rid;
  const float* ent;
  const float* rel_emb;
  const float* conv_w;
  const float* conv_b;
  const float* scale;
  const float* shift;
  int per_rel_conv, d, r, in_w, in_hw, Wo, img_stride, x_exp;
  int img_exp;      // e_I: the image planes in LDS hold (e1 row | rel row) 2^e_I as fp16 hi + lo (round 5: the conv on the matrix cores)
  int64_t* chk;     // the pass runs on a grouping prepared ahead (coper_group_next): its check words (group_body.h), else nullptr
  int w_div, w_rem; // coper_config.rel_mod_*: the weight planes hold relation r at slot r / w_div, for r % w_div == w_rem only
  int32_t* bad;     //   ... a tile of another relation is counted here (coper_check_ids) and runs on the slot's weights
};

// The dense finalize in this kernel's epilogue (round 4; one K slice only -- the workgroup's accumulators are then the whole
// sum): dense bias, folded FCBN, ReLU -> the fp32 h rows (the arithmetic of k_dense_finalize / k_finalize_h_publish, bit for
// bit), and the workgroup's largest value folded into a slot of x3m (bf16x3_chain.h: the batch's exponent; k_rel_scatter
// zeroed the slots at the start of the pass).  Saves the 17 MB round trip of the partial sums and a launch: 14 us of an
// FB15k-237-shaped pass.  h_out == NULL: partial sums to z_part as before.
// (The constant part lives in device memory, written at prepare: as kernel arguments its nine words pushed the main loop's
// scalar registers into spills -- the kernel ran 19 us longer to save a 14 us launch.)
#ifdef COPER_DBG_FUSED_CLOCK
// phase stamps of a workgroup (wave 0: matrix role, wave 4: conv role), read with coper_dbg_fused_phases
__device__ unsigned long long g_fused_ph[8 * 2048];
extern "C" __attribute__((visibility("default"))) int coper_dbg_fused_phases(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_ph), sizeof(unsigned long long) * 8 * (n < 2048 ? n : 2048)) == hipSuccess ? 0 : 1;
}
#define FUSED_PH(i_)                                                                                            \
  do {                                                                                                          \
    const int w_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                              \
    if ((threadIdx.x & 63) == 0 && w_ < 2048) g_fused_ph[8 * w_ + (i_)] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define FUSED_PH(i_) do { } while (0)
#endif
struct FusedFinConst {
  const int32_t* perm; const float* fc_b; const float* scale; const float* shift; const int32_t* w_exp; float* x3m;
  int per_rel_bias, x_exp, d, pad;
};
struct FusedFin {
  float* h_out; const FusedFinConst* c;
};
constexpr int FUSED_STAGE_WGS = 16;     // workgroups of the staging role (512 threads, one 16-byte PCIe read each per round)

// Roles.  A workgroup is 8 waves, two per SIMD: waves 0..3 are MATRIX waves (wave w owns feature blocks
// fb = w, w+4, ...: weight fragments prefetched into its registers, accumulators, all the MFMAs), waves 4..7
// are CONV waves (produce the x fragments of k-step k+1 while the matrix waves consume k-step k).  One wave
// cannot overlap its own MFMAs with its own VALU work on gfx950 (measured: a 54-MFMA / 216-FMA loop costs
// the sum of the two alone, whatever the interleaving), two waves on one SIMD can: the conv rides in the
// matrix pipe's shadow and each role hides the other's LDS latency.
//   barrier protocol (every wave, nk + 1 barriers): image rows in LDS -> [conv(0)] ->
//   for k: barrier (x(k) visible, stage (k+1)&1 free) ; matrix: MFMA(k) | conv: conv(k+1)

// ---- prologue shared by both roles: the slice's rows of the tile's images -> LDS.
// Two dependent latencies in all: the sorted row / relation ids of the wave's 16 queries (written by
// k_rel_scatter), then every row piece of those queries in flight at once.
// Round 5: the image is stored SPLIT -- per query a plane of fp16 hi and a plane of fp16 lo of (value 2^e_I), two values per dword
// (PD = (img_stride - 1) / 2 dwords per plane, hi plane first; img_stride = 2 PD + 1 is odd: 16 queries read 16 banks) -- because
// the conv runs on the matrix cores now (fused_conv_role): its B operand is eight fp16 values of a query's 3x3 window, taken
// from these planes with four LDS reads.  Same LDS bytes as the fp32 image of rounds 2 - 4.
__device__ __forceinline__ void fused_load_images(unsigned* __restrict__ img, const FusedConvArgs& A, int start, int n,
                                                  int t0, int t1) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int q0 = wave * 16;
  if (q0 < n) {
    int qi = q0 + (lane & 15);
    if (qi > n - 1) qi = n - 1;
    int my_row = A.sorted_row[start + qi], my_rid = A.sorted_rid[start + qi];
    // a grouping prepared ahead is CHECKED by the pass that consumes it (group_body.h): every query's live (e1, rel) against what was
    // sorted -- once per query (K slice 0, feature group 0), its loads issued with the ones above and below, the verdict at the end
#ifdef COPER_DBG_NO_GROUP_CHK      // (A/B of the guard's cost: tools/ab_bench.sh)
    const bool chk = false;
#else
    const bool chk = A.chk != nullptr && (blockIdx.y | blockIdx.z) == 0;
#endif
    int64_t live_rel = 0, live_e1 = 0;
    const int chk_row = my_row, chk_rid = my_rid;
    if (chk) group_chk_issue(A.chk, start + qi, live_rel, live_e1);
#ifdef COPER_DBG_FUSED_NO_IMG
    my_row = -1;
#endif
    const float* base = A.e1_rows ? A.e1_rows : A.ent;
    const int len = t1 - t0, PD = (A.img_stride - 1) >> 1, eI = A.img_exp;     // (len, t0, d even: dense_fused_supported)
    auto fetch2 = [&](const int row, const int rid, const int t) -> float2 {   // elements t, t + 1 of the query's (stacked) image
      if (t >= t1) return make_float2(0.f, 0.f);
      if (t < A.d) return row >= 0 ? *(const float2*)(base + (int64_t)row * A.d + t) : make_float2(0.f, 0.f);
      return *(const float2*)(A.rel_emb + (int64_t)rid * A.r + (t - A.d));
    };
    auto put2 = [&](const int q, const int pp, const float2 x) {
      unsigned short h0, l0, h1, l1;
      split1_s16(x3_scale(x.x, eI), h0, l0);
      split1_s16(x3_scale(x.y, eI), h1, l1);
      img[q * A.img_stride + pp] = (unsigned)h0 | ((unsigned)h1 << 16);
      img[q * A.img_stride + PD + pp] = (unsigned)l0 | ((unsigned)l1 << 16);
    };
    float2 v[16][2];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int row = __builtin_amdgcn_readlane(my_row, u), rid = __builtin_amdgcn_readlane(my_rid, u);
#pragma unroll
      for (int h = 0; h < 2; ++h) v[u][h] = fetch2(row, rid, t0 + 2 * (lane + 64 * h));
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (q0 + u < n) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (2 * (lane + 64 * h) < len) put2(q0 + u, lane + 64 * h, v[u][h]);
        if (lane == 0) img[(q0 + u) * A.img_stride + 2 * PD] = 0u;     // the pad dword (a window's unused third half may fall on it)
      }
    // rows longer than 256 values (not the shipped shapes): the rest, plainly
    if (len > 256) {
      for (int u = 0; u < 16 && q0 + u < n; ++u) {
        const int row = __builtin_amdgcn_readlane(my_row, u), rid = __builtin_amdgcn_readlane(my_rid, u);
        for (int pp = 128 + lane; 2 * pp < len; pp += 64) put2(q0 + u, pp, fetch2(row, rid, t0 + 2 * pp));
      }
    }
    if (chk) group_chk_verdict(A.chk, live_rel, live_e1, chk_row, chk_rid);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- matrix role.  Wave w (0..3) owns feature blocks w, w+4, ... of the NFB/4*4 "full" ones for every query
// block, and -- when NFB is not a multiple of 4 (d = 200: 13 blocks) -- query blocks w, w+4 of the one left
// over, so that every SIMD carries the same MFMA load to within one query block (dealing whole feature blocks
// would leave one wave with 4 of 13).  All four waves then stream that last block's weight fragments; the
// repeats hit L1/L2.
typedef unsigned fused_img_t;
template <int NFB, int NB, bool WNT>
__device__ __forceinline__ void fused_matrix_role(const uint4* __restrict__ xring, fused_img_t* __restrict__ img,
                                                  const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                  const FusedConvArgs& A, int64_t relw, int64_t relw_w, int start, int n, int fb0,
                                                  int nfb, int64_t ks32n, int64_t kb, int64_t ke, int t0, int t1,
                                                  float* __restrict__ zdst, int d_pad16, int wave, const FusedFin& Fn, int64_t ks32s) {
  constexpr int NFULL = NFB / 4;              // whole feature blocks per wave
  constexpr int REM = NFB % 4 ? 1 : 0;        // 1: one more block, shared by query block
  static_assert(NFB % 4 <= 1, "at most one left-over feature block");
  constexpr int NRQ = REM ? (NB + 3) / 4 : 0; // its query blocks per wave (w, w+4)
  constexpr int NW = NFULL + REM;             // weight fragment streams per wave
  // two waves per SIMD: 256 registers each, accumulators included
  constexpr int P = (COPER_FUSED_PMAX >= 3 && (NFULL * NB + NRQ) * 4 + FW_WREG3 + 44 <= COPER_FUSED_PBUDGET) ? 3 : 2;
  constexpr int XSTAGE = 2 * NB * 64;         // uint4 per stage: x hi [NB] | x lo [NB]
  const int lane = threadIdx.x & 63;
  const int nk = (int)(ke - kb);
  const uint4* wp[NW][2];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    int fb = fb0 + (j < NFULL ? wave + 4 * j : 4 * NFULL);
    if (fb > nfb - 1) fb = nfb - 1;
    int64_t o = ((relw_w * nfb + fb) * ks32s + kb) * 64;   // wave-uniform: scalar base + one shared lane offset (relw_w: the relation's slot)
    wp[j][0] = Whi + o;
    wp[j][1] = (const uint4*)(Wlo + o);
  }
  FW_DECL;
  // WNT: non-temporal loads when a weight set is read by one tile only (FB15k-237 CoPER shapes: -2.3 % on the whole
  // pass against cached loads, which push the entity table out of L2 ahead of the count pass); plain loads when every
  // tile of a relation re-reads it and L2 serves the repeats (3 tiles per relation at WN18RR shapes: -11 % on this
  // kernel; all 160 tiles for the static layer of plain ConvE: -3.5 %)
  // ... and per tile: a relation cut into several tiles (skewed batches: a few relations carry most queries) keeps cached
  // loads even under WNT, so that L2 / the Infinity Cache serve the repeats -- with every tile streaming past the caches a
  // Zipf-distributed batch (328 tiles for 237 relations) cost 261 us against 184 for the uniform one
  // (the workgroup-level dispatch in k_dense_fused_bf16x3 instantiates this role with WNT = false for such tiles: a
  // run-time choice between the two load forms inside the loop cost the uniform batch 3.7 % on this kernel)
#define FUSED_W_LOAD(p_) (WNT ? __builtin_nontemporal_load(p_) : *(p_))
#define W_ISSUE(s, kk)                                                                                 \
  {                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                                   \
      FW_HI(s, j) = FUSED_W_LOAD((const u32x4*)(wp[j][0] + (int64_t)(kk)*64) + lane);                 \
      FW_LO_RAW(s, j) = FUSED_W_LOAD((const u32x4*)((const uint4*)wp[j][1] + (int64_t)(kk)*64) + lane); \
    }                                                                                                  \
  }
#pragma unroll
  for (int t = 0; t < P; ++t)
    if (t < nk) W_ISSUE(t, t);      // the weight stream starts before anything else
  // (Round 5, late, -DCOPER_DBG_FUSED_CLOCK phases: the images stand in LDS 9 us (WN18RR) to 15 us (FB15k-237) after the workgroup's
  //  start -- three dependent round trips, tile -> ids -> rows, while every workgroup of the launch asks for its first 300 KB.
  //  Requesting the rows BEFORE the weights, or by the conv waves alone: the same 9 / 15 us, or worse; see DESIGN_LOG.md.)
  if (wave == 0) FUSED_PH(0);
  fused_load_images(img, A, start, n, t0, t1);
  if (wave == 0) FUSED_PH(1);
  f32x4 acc[NFULL][NB], accr[NRQ > 0 ? NRQ : 1];
#pragma unroll
  for (int j = 0; j < NFULL; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NRQ; ++t) accr[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // one k-step: barrier (x(k) visible), MFMAs on ring stage s2&1 with the weights of register stage s2 % P.
  // Term-major MFMA order: consecutive MFMAs write different accumulators; each accumulator still sees
  // lo*hi, hi*lo, hi*hi in that order.
#ifdef COPER_DBG_FUSED_NO_MFMA
#define M_STEP(s2)                                                                                  \
  {                                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                   \
    const uint4* xb = xring + ((s2)&1) * XSTAGE + lane;                                             \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                                \
      acc[0][0][0] += __uint_as_float(FW_HI((s2) % P, j)[0] ^ FW_LO_RAW((s2) % P, j)[1]);           \
      acc[0][0][1] += __uint_as_float(FW_HI((s2) % P, j)[2] ^ FW_LO_RAW((s2) % P, j)[0]);           \
    }                                                                                               \
    acc[0][0][2] += __uint_as_float(xb[0].x);                                                       \
  }
#else
  // Round 5: the x fragments of query block q + 1 are requested BEFORE the nine MFMAs of block q are issued (two register
  // pairs, alternating).  Rounds 2 - 4 left the order to the compiler, which used one pair: ds_read x 2, wait, 9 MFMAs, ds_read x 2,
  // wait ... -- every query block paid an LDS round trip with the matrix pipe idle (~110 of its ~250 cycles: the role ran at
  // 31 cycles per MFMA, half the pipe's rate, on every shape -- profiles/r05a_encoder_ablations.txt, the NOCONV / NOW builds).
  // __builtin_amdgcn_sched_group_barrier pins the issue order inside the step's basic block: 2 LDS reads (block q + 1), then
  // the 3 NFULL MFMAs of block q.
#ifdef COPER_DBG_FUSED_NO_XREAD      /* diagnostic: the B operands come from nowhere (no LDS read in the matrix role) */
#define M_LD(i_) make_uint4((unsigned)lane, (unsigned)(i_), 0x3c003c00u, 0x3c003c00u)
#else
#define M_LD(i_) xb[i_]
#endif
#ifndef COPER_FUSED_XPF
#define COPER_FUSED_XPF 2     // query blocks the x fragments are requested ahead of their MFMAs (1 or 2)
#endif
  // the blocks a wave consumes in one step, in order: 0 .. NB-1, then (NRQ > 0) its own block of the left-over feature block
#ifndef COPER_FUSED_XBUDGET
#define COPER_FUSED_XBUDGET 220    // tiles whose registers (at P = 3) exceed this request their x fragments ONE block ahead (8 registers less)
#endif
  constexpr int NV = NB + (NRQ > 0 ? 1 : 0);
  constexpr int XPF = (P == 3 && (NFULL * NB + NRQ) * 4 + FW_WREG3 + 44 > COPER_FUSED_XBUDGET) ? 1 : COPER_FUSED_XPF, XR = XPF + 1;
  const int qr = wave < NB ? wave : NB - 1;
#define M_LDV_H(v_) M_LD(((v_) < NB ? (v_) : qr) * 64)
#define M_LDV_L(v_) M_LD((NB + ((v_) < NB ? (v_) : qr)) * 64)
#define M_STEP(s2)                                                                                  \
  {                                                                                                 \
    FW_PRE((s2) % P)                                                                                \
    __builtin_amdgcn_s_barrier();                                                                   \
    const uint4* xb = xring + ((s2)&1) * XSTAGE + lane;                                             \
    uint4 bhq[XR], blq[XR];                                                                         \
    _Pragma("unroll") for (int v = 0; v < XPF && v < NV; ++v) { bhq[v] = M_LDV_H(v); blq[v] = M_LDV_L(v); } \
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (XPF < NV ? XPF : NV), 0);                      \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                \
      if (q + XPF < NV) {                                                                           \
        bhq[(q + XPF) % XR] = M_LDV_H(q + XPF); blq[(q + XPF) % XR] = M_LDV_L(q + XPF);             \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                          \
      }                                                                                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * NFULL, 0);                                    \
      const uint4 bh = bhq[q % XR], bl = blq[q % XR];                                               \
      _Pragma("unroll") for (int j = 0; j < NFULL; ++j)                                             \
          acc[j][q] = MFMA16_BF16(FW_LO((s2) % P, j), bh, acc[j][q]);                               \
      _Pragma("unroll") for (int j = 0; j < NFULL; ++j)                                             \
          acc[j][q] = MFMA16_BF16(FW_HI((s2) % P, j), bl, acc[j][q]);                               \
      _Pragma("unroll") for (int j = 0; j < NFULL; ++j)                                             \
          acc[j][q] = MFMA16_BF16(FW_HI((s2) % P, j), bh, acc[j][q]);                               \
    }                                                                                               \
    _Pragma("unroll") for (int t = 0; t < NRQ; ++t) {                                               \
      const int q = wave + 4 * t < NB ? wave + 4 * t : NB - 1; /* past the tile: recompute, not stored */ \
      uint4 bh, bl;                                                                                 \
      if (t == 0) { bh = bhq[NB % XR]; bl = blq[NB % XR]; }    /* (requested beside the last full blocks) */ \
      else { bh = M_LD(q * 64); bl = M_LD((NB + q) * 64); }                                         \
      accr[t] = MFMA16_BF16(FW_LO((s2) % P, NFULL), bh, accr[t]);                                   \
      accr[t] = MFMA16_BF16(FW_HI((s2) % P, NFULL), bl, accr[t]);                                   \
      accr[t] = MFMA16_BF16(FW_HI((s2) % P, NFULL), bh, accr[t]);                                   \
    }                                                                                               \
  }
#endif
  // main loop, 2P steps per trip (lcm of ring stages and prefetch depth: static register indices), with NO
  // conditional code: hipcc's s_waitcnt insertion counts the loads in flight per path, and a branch around a
  // step or around its prefetch makes the only safe count 0 -- the prefetch would be waited for the moment
  // it is issued.  A prefetch past the slice's end re-reads the slice's first fragments instead (cache hits,
  // never used).
  int k0 = 0;
  for (; k0 + 2 * P <= nk; k0 += 2 * P) {
#pragma unroll
    for (int s2 = 0; s2 < 2 * P; ++s2) {
      M_STEP(s2);
#ifndef COPER_DBG_FUSED_NO_W
      const int kn = k0 + s2 + P;
      W_ISSUE(s2 % P, kn < nk ? kn : 0);
#endif
    }
    if (k0 == 0 && wave == 0) FUSED_PH(2);      // (the first 2 P steps done)
  }
  // remaining nk % 2P steps
#pragma unroll
  for (int s2 = 0; s2 < 2 * P - 1; ++s2) {
    if (k0 + s2 < nk) {
      M_STEP(s2);
#ifndef COPER_DBG_FUSED_NO_W
      if (k0 + s2 + P < nk) W_ISSUE(s2 % P, k0 + s2 + P);
#endif
    }
  }
#undef M_STEP
#undef W_ISSUE
  if (wave == 0) FUSED_PH(3);
#ifdef COPER_DBG_FUSED_NO_STORE
  if (acc[0][0][0] != 1.2345f) return;
#endif
  if (Fn.h_out) {   // (kernel-uniform) finalize here: see FusedFin
    const FusedFinConst Fc = *Fn.c;
    const int d = Fc.d;
    const int zexp = -((Fc.w_exp ? Fc.w_exp[Fc.per_rel_bias ? relw : 0] : 0) + Fc.x_exp);   // the sums carry 2^(e_W + e_x) (split16.h)
    const float* bsrc = Fc.per_rel_bias ? Fc.fc_b + relw * d : Fc.fc_b;
    float m = 0.f;
    int qrow[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int qi = q * 16 + (lane & 15);
      qrow[q] = qi < n ? Fc.perm[start + qi] : -1;
    }
    auto fin = [&](const f32x4& a, const int fb, const int q, const float4& b4, const float4& s4, const float4& t4) {
      const int k0 = fb * 16 + 4 * (lane >> 4);
      if (fb >= nfb || qrow[q] < 0 || k0 >= d) return;
      float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;       // (the finalize kernels add the slice to a zero: same bits, also for -0)
      z0 += a[0]; z1 += a[1]; z2 += a[2]; z3 += a[3];
      float4 v;
      v.x = fmaxf(fmaf(x3_scale(z0, zexp) + b4.x, s4.x, t4.x), 0.f);
      v.y = fmaxf(fmaf(x3_scale(z1, zexp) + b4.y, s4.y, t4.y), 0.f);
      v.z = fmaxf(fmaf(x3_scale(z2, zexp) + b4.z, s4.z, t4.z), 0.f);
      v.w = fmaxf(fmaf(x3_scale(z3, zexp) + b4.w, s4.w, t4.w), 0.f);
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
      *(float4*)(Fn.h_out + (int64_t)qrow[q] * d + k0) = v;
    };
    auto quad = [&](const float* p, const int fb) -> float4 {
      const int k0 = fb * 16 + 4 * (lane >> 4);
      return (fb < nfb && k0 < d) ? *(const float4*)(p + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int j = 0; j < NFULL; ++j) {
      const int fb = fb0 + wave + 4 * j;
      const float4 b4 = quad(bsrc, fb), s4 = quad(Fc.scale, fb), t4 = quad(Fc.shift, fb);
#pragma unroll
      for (int q = 0; q < NB; ++q) fin(acc[j][q], fb, q, b4, s4, t4);
    }
    if (NRQ > 0) {
      const int fb = fb0 + 4 * NFULL;
      const float4 b4 = quad(bsrc, fb), s4 = quad(Fc.scale, fb), t4 = quad(Fc.shift, fb);
#pragma unroll
      for (int t = 0; t < NRQ; ++t) {
        const int q = wave + 4 * t;
        if (q < NB) {
          // (qrow is indexed with compile-time constants above; here the wave's own query block: read it again)
          const int qi = q * 16 + (lane & 15);
          const int row = qi < n ? Fc.perm[start + qi] : -1;
          const int k0 = fb * 16 + 4 * (lane >> 4);
          if (fb < nfb && row >= 0 && k0 < d) {
            float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
            z0 += accr[t][0]; z1 += accr[t][1]; z2 += accr[t][2]; z3 += accr[t][3];
            float4 v;
            v.x = fmaxf(fmaf(x3_scale(z0, zexp) + b4.x, s4.x, t4.x), 0.f);
            v.y = fmaxf(fmaf(x3_scale(z1, zexp) + b4.y, s4.y, t4.y), 0.f);
            v.z = fmaxf(fmaf(x3_scale(z2, zexp) + b4.z, s4.z, t4.z), 0.f);
            v.w = fmaxf(fmaf(x3_scale(z3, zexp) + b4.w, s4.w, t4.w), 0.f);
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            *(float4*)(Fn.h_out + (int64_t)row * d + k0) = v;
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && m > 0.f)
      atomicMax((unsigned*)&Fc.x3m[(blockIdx.x + gridDim.x * blockIdx.z) & (X3M_SLOTS - 1)], __float_as_uint(m));   // (<= ~1 block per slot: no contention)
    return;
  }
#pragma unroll
  for (int j = 0; j < NFULL; ++j) {
    const int fb = fb0 + wave + 4 * j;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int qi = q * 16 + (lane & 15);
      if (fb < nfb && qi < n) {
        float* dst = zdst + (int64_t)(start + qi) * d_pad16 + fb * 16 + 4 * (lane >> 4);
        *(float4*)dst = make_float4(acc[j][q][0], acc[j][q][1], acc[j][q][2], acc[j][q][3]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NRQ; ++t) {
    const int fb = fb0 + 4 * NFULL, q = wave + 4 * t;
    const int qi = q * 16 + (lane & 15);
    if (fb < nfb && q < NB && qi < n) {
      float* dst = zdst + (int64_t)(start + qi) * d_pad16 + fb * 16 + 4 * (lane >> 4);
      *(float4*)dst = make_float4(accr[t][0], accr[t][1], accr[t][2], accr[t][3]);
    }
  }
}

// (Round 5, measured and removed: the conv waves TOUCHING the weight stream ahead of the matrix waves -- one dword per 128-byte line
// of the k-step four, six or ten steps on, an asm load into a reserved register nobody waits for -- so that the lines would sit in
// the XCD's L2 when the matrix waves ask.  The launch got slower with every step of distance: fused encoder 0.165 -> 0.200 / 0.223 /
// 0.266 ms at FB15k-237 shapes, WN18RR 0.040 -> 0.047, plain ConvE 0.233 -> 0.241: the touched lines are fetched again by the
// non-temporal loads that follow (profiles/r05b_experiments.txt, F).)
// ---- conv role: conv wave cw (0..3) produces x fragments f = cw, cw+4
// Round 5: the 3x3 conv on the matrix cores.  Rounds 2 - 4 ran it as 72 fp32 FMAs per lane and output pixel (conv_x8) -- ~150
// vector instructions per 16-query fragment and k-step with the split, 1,700 cycles per k-step for the eight fragments of a
// 128-query tile (profiles/r05a_encoder_ablations.txt, NOMFMA build), against 1,250 for the dense layer's own MFMAs: the conv, not
// the weight stream, bounded the static-weight shapes, and it shared every SIMD's issue with the matrix waves on all of them.
// Here one output pixel of 16 queries x 32 channels is TWO v_mfma_f32_16x16x32_f16:
//     D[channel m][query n] = b'[m] + sum_k A[m][k] B[k][n],   K = 32 slots holding the 27 products of the x3 arithmetic
//         slots  0.. 7   tap_hi[t] img_hi[t]   t = 0..7      (lane group g = 0)
//         slots  8..15   tap_hi[t] img_lo[t]                 (g = 1)
//         slots 16..23   tap_lo[t] img_hi[t]                 (g = 2)
//         slots 24..26   tap_hi[8] img_hi[8], tap_hi[8] img_lo[8], tap_lo[8] img_hi[8];  27..31 zero weights   (g = 3)
// with tap' = conv_w scale (folded BN, conv_fold.h) 2^(e_x - e_I) and img' = image 2^e_I split into fp16 hi + lo (the planes
// fused_load_images leaves in LDS), so that D = x 2^e_x as before: the bias enters as the C operand, ReLU and the split of x
// follow on the 8 accumulator values of a lane.  The rows of the two MFMAs are channels 8 (m / 4) + 4 u + m % 4 (u = 0, 1): a lane's
// eight outputs are then the channels 8 g .. 8 g + 7 of its query -- the B-operand layout of the dense layer's MFMAs, written
// to the ring without a shuffle.  A lane's B operand: three ds_read2_b32 (the window's rows, its own plane) + one ds_read_b32
// (tap 8 of the other plane), two v_alignbit per row for the window's column parity, two v_perm.  ~50 vector instructions and two
// MFMAs per fragment and k-step instead of ~150; 60 fewer registers in this role (no taps).
// Arithmetic: every product carries 22 bits (hi hi + hi lo + lo hi; lo lo dropped: 2^-22), fp32 accumulation inside the MFMA --
// against the fp32 fma chain of rounds 2 - 4 this moves x by ~2^-22 of the pixel's magnitude, the size of the split x gets
// anyway; tests/test_gpu_scale.py holds h to 1e-5 of float64 relative to its magnitude at every scale as before.
// x (and with it h) remains a pure function of (e1, rel): a column of the MFMA sees only its own query's window.
template <int NB>
__device__ __forceinline__ void fused_conv_role(uint4* __restrict__ xring, unsigned* __restrict__ img,
                                                const FusedConvArgs& A, int64_t relw, int start, int n, int64_t kb,
                                                int64_t ke, int i_lo, int t0, int t1, int cw) {
  constexpr int NFR = (NB + 3) / 4;
  constexpr int XSTAGE = 2 * NB * 64;
  const int lane = threadIdx.x & 63;
  const int nk = (int)(ke - kb);
  const int Wo = A.Wo, in_w = A.in_w, W2 = in_w >> 1;
  const int g = lane >> 4, m = lane & 15;
  const int PD = (A.img_stride - 1) >> 1;
  // the A operands (the relation's folded taps, split) and the bias rows (loads in flight during the image prologue)
  u32x4 aop[2];
  f32x4 cop[2];
  {
    const float* wsrc = A.per_rel_conv ? A.conv_w + relw * (int64_t)(9 * 32) : A.conv_w;
    const float* bsrc = A.per_rel_conv ? A.conv_b + relw * (int64_t)32 : A.conv_b;
    const int te = A.x_exp - A.img_exp;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ch = 8 * (m >> 2) + 4 * u + (m & 3);
      const float sc = A.scale[ch];
      unsigned short th[9], tl[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) split1_s16(x3_scale(wsrc[k * 32 + ch] * sc, te), th[k], tl[k]);
      unsigned short sl[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) sl[k] = g == 2 ? tl[k] : th[k];
      if (g == 3) {
        sl[0] = th[8]; sl[1] = th[8]; sl[2] = tl[8];
#pragma unroll
        for (int k = 3; k < 8; ++k) sl[k] = 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) aop[u][k] = (unsigned)sl[2 * k] | ((unsigned)sl[2 * k + 1] << 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 8 * g + 4 * u + i;
        cop[u][i] = x3_scale(fmaf(bsrc[c], A.scale[c], A.shift[c]), A.x_exp);
      }
    }
  }
  fused_load_images(img, A, start, n, t0, t1);
  if (cw == 0) FUSED_PH(5);
  // dword bases of this lane's query in each of the wave's fragments (padding lanes repeat the last query): own plane (hi for
  // g = 0, 2, 3; lo for g = 1) and the other one (tap 8's second term, used by g = 3)
  int qown[NFR], qoth[NFR];
#pragma unroll
  for (int t = 0; t < NFR; ++t) {
    int qi = (cw + 4 * t) * 16 + m;
    if (qi > n - 1) qi = n - 1;
    qown[t] = qi * A.img_stride + (g == 1 ? PD : 0);
    qoth[t] = qi * A.img_stride + (g == 1 ? 0 : PD);
  }
  int poff = (int)(kb - (int64_t)i_lo * Wo);   // pixel offset ci*in_w + cj of the NEXT conv step (values; a dword holds two)
  int cj = poff;
  for (int kk = 0; kk <= nk; ++kk) {
    // kk = 0: x(0) before the first barrier; kk >= 1: barrier k = kk-1, then x(kk) while the matrix waves run k
    if (kk > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my x writes of step kk-1 have landed in LDS
      __builtin_amdgcn_s_barrier();
    }
#ifdef COPER_DBG_FUSED_NO_CONV
    if (kk < 1) {
#else
    if (kk < nk) {
#endif
      const int pd = poff >> 1;
      const unsigned sh = (unsigned)(poff & 1) << 4;      // wave-uniform: the window starts in the low or the high half of a dword
#pragma unroll
      for (int t = 0; t < NFR; ++t) {
        if (cw + 4 * t < NB) {   // wave-uniform
          const unsigned* r = img + qown[t] + pd;
          const unsigned d00 = r[0], d01 = r[1], d10 = r[W2], d11 = r[W2 + 1], d20 = r[2 * W2], d21 = r[2 * W2 + 1];
          const unsigned o21 = img[qoth[t] + pd + 2 * W2 + 1];
          const unsigned a01 = __builtin_amdgcn_alignbit(d01, d00, sh), a2 = d01 >> sh;
          const unsigned b01 = __builtin_amdgcn_alignbit(d11, d10, sh), b2 = d11 >> sh;
          const unsigned c01 = __builtin_amdgcn_alignbit(d21, d20, sh), c2 = d21 >> sh;
          u32x4 bop;
          bop[0] = a01;                                                  // taps 0, 1
          bop[1] = __builtin_amdgcn_perm(b01, a2, 0x05040100u);          // taps 2, 3
          bop[2] = __builtin_amdgcn_perm(b2, b01, 0x05040302u);          // taps 4, 5
          bop[3] = c01;                                                  // taps 6, 7
          if (g == 3) {                                                  // tap 8: [hi, lo], [hi, -]; the other slots meet zero weights
            bop[0] = __builtin_amdgcn_perm(o21 >> sh, c2, 0x05040100u);
            bop[1] = c2 & 0xffffu;
          }
          const f32x4 y0 = MFMA16_BF16(aop[0], bop, cop[0]);
          const f32x4 y1 = MFMA16_BF16(aop[1], bop, cop[1]);
          // ReLU: one v_med3_f32 per value (fmaxf compiles to two: it canonicalises its operand first).  NOT inline asm: the hardware
          // does not interlock a vector instruction that reads an MFMA's result, the compiler's hazard recognizer pads the wait
          // states for its own instructions only -- an asm v_max_f32 right behind the MFMA read stale registers (measured: h = 0).
          // (the upper bound is fp16's largest finite value: activations of e1_rows beyond the agreed x3_ent_absmax saturate, as in
          //  round 4, instead of leaving v_cvt_pk_f16_f32 as inf and h as NaN rows -- ADVICE r5; in-range values: the same bits)
          float y[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            y[i] = __builtin_amdgcn_fmed3f(y0[i], 0.f, 65504.f);
            y[4 + i] = __builtin_amdgcn_fmed3f(y1[i], 0.f, 65504.f);
          }
          uint4 h4, l4;
          split8_pos_s16(y, h4, l4);
          uint4* dst = xring + (kk & 1) * XSTAGE + (cw + 4 * t) * 64 + lane;
          dst[0] = h4;
          dst[NB * 64] = l4;
        }
      }
      if (++cj == Wo) { cj = 0; poff += in_w - Wo + 1; } else { ++poff; }
    }
  }
}

// (Round 2, measured and not kept: a persistent form -- one workgroup per CU walking the (tile, K slice) items round-robin
// instead of a grid sized by the worst case, most of it empty workgroups -- took 0.24 ms where this launch takes 0.19
// (FB15k-237 CoPER shapes; plain ConvE 0.30 vs 0.27): the hardware dispatcher hands a freed CU the next workgroup, a static
// walk does not.)
#ifdef COPER_DBG_FUSED_CLOCK
// diagnostic build: per workgroup (queries of its tile, start and end on the 100 MHz s_memrealtime clock), wave 0
__device__ unsigned long long g_fused_clk[3 * 2048];
extern "C" __attribute__((visibility("default"))) int coper_dbg_fused_clock(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_clk), sizeof(unsigned long long) * 3 * (n < 2048 ? n : 2048)) == hipSuccess ? 0 : 1;
}
#endif

template <int NFB, bool WNT>
__global__ __launch_bounds__(512) void k_dense_fused_bf16x3(const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                            FusedConvArgs A, const int32_t* __restrict__ tiles,
                                                            const int32_t* __restrict__ n_tiles, int64_t cap_small,
                                                            int nfb, int64_t ks32n, int nslices, int64_t Bcap,
                                                            int d_pad16, float* __restrict__ z_part, FusedFin Fn,
                                                            int n_special, const int32_t* __restrict__ stage_src, int64_t stage_n,
                                                            int64_t* __restrict__ stage_dst, int64_t ks32s, GroupJob J, int n_group,
                                                            const int32_t* __restrict__ post_src, int64_t post_n, int32_t* __restrict__ post_dst) {
  extern __shared__ uint4 fused_lds[];
  // Staging role (coper_stage_ids_next): workgroups in front of the tile lists bring the NEXT pass's int32 batch in from pinned
  // host memory (PCIe reads) and widen it to the int64 arrays the ABI takes, while the tiles stream their weights.  A pass has
  // fewer relation tiles than the chip has CUs at the BASELINE shapes (237 of 256), so these land on CUs that would idle.
  // Grouping role (coper_group_next): ONE workgroup in front of them sorts the NEXT pass's batch by relation into a set of grouping
  // arrays this pass does not read (group_body.h) -- the two grouping launches of that pass, done in this launch's shadow.
  // Posting role (coper_post_i32_next): when the pass that runs was grouped ahead it has no grouping launch to carry the LAST
  // pass's ranks out; the staging workgroups copy them.
  // The special workgroups come FIRST in the grid: the tile lists are sized for the worst case (B / 33 + R blocks, most of which
  // find no tile and leave), and on a launch of one workgroup per CU the dispatcher works through those one CU slot at a time --
  // behind them the staging workgroups started ~140 us into a 165 us launch (measured when the grouping role was added).
  if ((int)blockIdx.x < n_special) {
    if (blockIdx.y | blockIdx.z) return;
    const int n_copy = n_special - n_group;                            // staging workgroups first, the grouping workgroup behind them:
    if ((int)blockIdx.x >= n_copy) {                                   // when it runs they have all been dispatched, so it may wait for them
      GROUP_CLK(0);
      if (J.wait_for > 0) {
        if (threadIdx.x == 0) {
          // (bounded: ~1 s.  A ticket that never arrives -- it cannot, the staging workgroups stand in front of this one in the
          //  grid -- would leave this workgroup sorting ids that are not there yet; the pass that consumes the sorting checks
          //  it against the live ids and reports it stale: group_body.h)
          for (int spin = 0; spin < (1 << 21) && __hip_atomic_load(J.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < J.wait_for; ++spin)
            __builtin_amdgcn_s_sleep(16);
          // (round 6, ADVICE r5: the hand-over is a release / acquire pair at agent scope -- ONE per staging workgroup and one here,
          //  +1 us per pass measured; round 5 relied on relaxed atomics, write-through stores and in-order dispatch alone.  The
          //  release / acquire FENCES in every wave that round 5 had tried first stretched the launch from 165 to 240 us.)
          (void)__hip_atomic_load(J.ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(J.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
      }
      GROUP_CLK(1);
      group_role_body<512>(J, (int*)fused_lds);
      GROUP_CLK(5);
      return;
    }
    const int64_t nb = n_copy, b = (int64_t)blockIdx.x;
    if (post_n > 0) {
      const bool pvec = ((((uintptr_t)post_src) | ((uintptr_t)post_dst)) & 15) == 0;
      const int64_t p4 = pvec ? post_n / 4 : 0;
      for (int64_t i = b * 512 + threadIdx.x; i < p4; i += nb * 512) ((int4*)post_dst)[i] = ((const int4*)post_src)[i];
      for (int64_t i = 4 * p4 + b * 512 + threadIdx.x; i < post_n; i += nb * 512) post_dst[i] = post_src[i];
    }
    const bool vec = ((((uintptr_t)stage_src) | ((uintptr_t)stage_dst)) & 15) == 0;
    const int64_t n4 = vec ? stage_n / 4 : 0;
    auto copy4 = [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo + b * 512 + threadIdx.x; i < hi; i += nb * 512) {
        const int4 v = ((const int4*)stage_src)[i];
        longlong2* o = (longlong2*)(stage_dst + 4 * i);
        o[0] = make_longlong2(v.x, v.y);
        o[1] = make_longlong2(v.z, v.w);
      }
    };
    // The grouping workgroup reads the FRONT of what this job stages (the id arrays it sorts): that part first, with stores and (on
    // its side) loads that are coherent across the XCDs' L2s by themselves, then the signal.  (Release / acquire FENCES at agent
    // scope write back and invalidate whole L2s: 128 waves of them stretched this launch from 165 to 240 us.)
    const bool sig = n_group && J.wait_for > 0;
    int64_t f4 = sig ? (J.front + 3) / 4 : 0;
    if (f4 > n4 || f4 * 4 < J.front) f4 = 0;      // (an unaligned job: element by element below)
    for (int64_t i = b * 512 + threadIdx.x; i < f4; i += nb * 512) {
      const int4 v = ((const int4*)stage_src)[i];
      int64_t* o = stage_dst + 4 * i;
      __hip_atomic_store(o + 0, (int64_t)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 1, (int64_t)v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 2, (int64_t)v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 3, (int64_t)v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (sig && f4 == 0)
      for (int64_t i = b * 512 + threadIdx.x; i < J.front; i += nb * 512)
        __hip_atomic_store(stage_dst + i, (int64_t)stage_src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sig) {
      __syncthreads();                              // (every wave's stores have been acknowledged)
      if (threadIdx.x == 0) __hip_atomic_fetch_add(J.ticket, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    copy4(f4, n4);
    for (int64_t i = 4 * n4 + b * 512 + threadIdx.x; i < stage_n; i += nb * 512) stage_dst[i] = stage_src[i];
    return;
  }
  // One index over both tile lists, the real tiles FIRST and contiguous: the 33..128-query tiles (n_tiles[1] of them), then the
  // <= 32-query tiles (n_tiles[0]); the grid holds B / 128 + min(R, B) + 1 blocks per slice -- an upper bound of their sum (a
  // relation with c queries has at most c / 128 + 1 tiles) -- and the blocks beyond leave.  (Until round 5 each list had its own
  // worst-case range, B / 33 + 1 and R: at plain ConvE 461 empty blocks stood between the 160 tiles of one K slice and those of
  // the next, each needing a whole free CU for its LDS before the dispatcher could move on.)
  int tile = (int)blockIdx.x - n_special;
  const int32_t* tl;
  {
    const int nbig = n_tiles[1];
    if (tile < nbig) {
      tl = tiles + 4 * (cap_small + tile);
    } else {
      tile -= nbig;
      if (tile >= n_tiles[0]) return;
      tl = tiles + 4 * (int64_t)tile;
    }
  }
#ifdef COPER_DBG_FUSED_EXIT
  return;
#endif
  const int slice = blockIdx.y;
  const int fb0 = blockIdx.z * NFB;
  const int64_t relw = __builtin_amdgcn_readfirstlane(tl[0]);
  int64_t relw_w = relw;         // the relation's slot in the weight planes
  if (A.w_div > 1) {
    relw_w = (int)relw / A.w_div;
    if ((int)relw - (int)relw_w * A.w_div != A.w_rem && threadIdx.x == 0 && (blockIdx.y | blockIdx.z) == 0) atomicAdd(A.bad, (int)tl[2]);
  }
  const int start = __builtin_amdgcn_readfirstlane(tl[1]);
  const int n = __builtin_amdgcn_readfirstlane(tl[2]);
  const bool shared_w = __builtin_amdgcn_readfirstlane(tl[3]) != 0;
  const int64_t kb = ks32n * slice / nslices, ke = ks32n * (slice + 1) / nslices;
  float* zdst = z_part + (int64_t)slice * Bcap * d_pad16;
  const int nb = (n + 15) >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef COPER_DBG_FUSED_CLOCK
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  uint4* xring = fused_lds;                       // 2 stages x 16 slots x 1 KiB
  unsigned* img = (unsigned*)(fused_lds + 2 * 16 * 64);
  // pixel p = k-step index: the slice needs image rows i_lo .. i_hi + 2
  const int i_lo = (int)(kb / A.Wo);
  const int t0 = i_lo * A.in_w;
  int t1 = ((int)((ke - 1) / A.Wo) + 3) * A.in_w;
  if (t1 > A.in_hw) t1 = A.in_hw;
#define TILE_WNT(x_) (x_)
#define BODY(NB_)                                                                                                      \
  if (wave < 4) {                                                                                                      \
    if (WNT && !shared_w)                                                                                              \
      fused_matrix_role<NFB, NB_, TILE_WNT(true)>(xring, img, Whi, Wlo, A, relw, relw_w, start, n, fb0, nfb, ks32n, kb, ke, t0, t1, zdst, \
                                                  d_pad16, wave, Fn, ks32s);                                           \
    else                                                                                                               \
      fused_matrix_role<NFB, NB_, TILE_WNT(false)>(xring, img, Whi, Wlo, A, relw, relw_w, start, n, fb0, nfb, ks32n, kb, ke, t0, t1, zdst, \
                                                   d_pad16, wave, Fn, ks32s);                                          \
  } else                                                                                                               \
    fused_conv_role<NB_>(xring, img, A, relw, start, n, kb, ke, i_lo, t0, t1, wave - 4);
  switch (nb) {
    case 1: BODY(1); break;
    case 2: BODY(2); break;
    case 3: BODY(3); break;
    case 4: BODY(4); break;
    case 5: BODY(5); break;
    case 6: BODY(6); break;
    case 7: BODY(7); break;
    default: BODY(8); break;
  }
#undef BODY
#ifdef COPER_DBG_FUSED_CLOCK
  if (threadIdx.x == 0) {
    const int w = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (w < 2048) { g_fused_clk[3 * w] = (unsigned long long)n; g_fused_clk[3 * w + 1] = dbg_t0; g_fused_clk[3 * w + 2] = __builtin_amdgcn_s_memrealtime(); }
  }
#endif
}

// slice geometry: rows of the image a K slice needs (same formula as the kernel), maximum over the slices
static int fused_rows_max(const Dims& dm, int nslices) {
  int64_t ks32n = dm.F_pad / 32;
  int m = 0;
  for (int s = 0; s < nslices; ++s) {
    int64_t kb = ks32n * s / nslices, ke = ks32n * (s + 1) / nslices;
    if (ke <= kb) continue;
    int rows = (int)((ke - 1) / dm.Wo) - (int)(kb / dm.Wo) + 3;
    if (rows > m) m = rows;
  }
  return m;
}

bool dense_fused_supported(const coper_handle* h, int nslices) {
  const Dims& dm = h->dm;
  if (!(dm.fh == 3 && dm.fw == 3 && dm.C == 32) || dm.concat_rel) return false;
  if (dm.F != dm.F_pad || dm.F != (int64_t)dm.Ho * dm.Wo * 32) return false;
  if (dm.gen_conv && !dm.gen_fc) return false;       // per-relation filters need single-relation tiles
  if (!(dm.nfb == 13 || dm.nfb == 8 || dm.nfb == 16)) return false;
  if ((dm.in_w & 1) || (dm.d & 1) || (dm.stacked && (dm.r & 1))) return false;   // the image planes hold two values per dword
  int stride = fused_rows_max(dm, nslices) * dm.in_w;
  stride |= 1;
  size_t lds = (size_t)2 * 16 * 64 * sizeof(uint4) + (size_t)128 * stride * sizeof(float);
  return lds <= 160 * 1024;
}

template <int NFB, bool WNT>
static void dense_fused_launch(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                               int nslices, int zgroups, float* h_fin, hipStream_t s) {
  const Dims& dm = h->dm;
  FusedFin Fn;
  Fn.h_out = h_fin; Fn.c = (const FusedFinConst*)h->fused_fin_dev;
  int64_t cap_small = (dm.gen_fc ? dm.R : 1) + 1;
  FusedConvArgs A;
  A.e1_rows = e1_rows; A.sorted_row = h->sorted_row; A.sorted_rid = h->sorted_rid;
  A.ent = h->params["ent_emb"].ptr;
  A.rel_emb = dm.lookup ? nullptr : h->params["rel_emb"].ptr;
  A.conv_w = dm.gen_conv ? h->conv_w_rel : h->params["conv1_weights"].ptr;
  A.conv_b = dm.gen_conv ? h->conv_b_rel : h->params["conv1_bias"].ptr;
  A.scale = h->conv_scale; A.shift = h->conv_shift; A.x_exp = h->x_exp; A.img_exp = h->img_exp;
  A.per_rel_conv = dm.gen_conv ? 1 : 0;
  A.d = dm.d; A.r = dm.r; A.in_w = dm.in_w; A.in_hw = dm.in_h * dm.in_w; A.Wo = dm.Wo;
  A.img_stride = (fused_rows_max(dm, nslices) * dm.in_w) | 1;
  A.chk = h->pass_chk;
  A.w_div = h->w_div; A.w_rem = h->w_rem; A.bad = h->rel_count + dm.R + 1;
  size_t lds = (size_t)2 * 16 * 64 * sizeof(uint4) + (size_t)128 * A.img_stride * sizeof(float);
  static uint64_t attr_done = 0;   // per instantiation, one bit per device: always the hardware maximum
  const uint64_t bit = 1ull << (h->cfg.device & 63);
  if (!(attr_done & bit)) {
    (void)hipFuncSetAttribute((const void*)k_dense_fused_bf16x3<NFB, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done |= bit;
  }
  // a pending staging job (coper_stage_ids_next) rides in this launch: FUSED_STAGE_WGS more workgroups
  // (Round 6 measured the tiles of one relation dealt to ONE XCD -- an XCD-chunked tile index for launches with several K slices:
  //  the encoder's HBM traffic at WN18RR shapes fell from 156 to 94 - 100 MB and the pass got 9 - 12 us slower, profiles/r06_wn_xcd.txt:
  //  the launch is one round of workgroups bound by its prologue and the matrix pipe, and the deal below ends every K slice's row
  //  on its smallest tiles.  Removed.)
  const int n_tile_blocks = (int)(B / 128 + (cap_small - 1 < B ? cap_small - 1 : B) + 1);     // (an upper bound of the two tile lists together: see the kernel)
  // (a pass being captured into a hipGraph leaves a pending staging job to the next eager call, as coper_post_i32_next does: a
  // replay would repeat the PCIe read with the pointers recorded at capture time and overwrite whatever staging buffer they name)
  coper_handle::PassPipeline& pp = h->pipe;
  bool capturing = false;
  if (pp.stage.n > 0 || pp.post.n > 0 || pp.gnext.ride) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); capturing = true; }
    else capturing = cap != hipStreamCaptureStatusNone;
  }
  const bool stage_now = pp.stage.n > 0 && !capturing;
  // a pass that was grouped ahead has no grouping launch for a pending coper_post_i32_next to ride in: the staging workgroups take it
  bool post_now = pp.post.n > 0 && pp.post.here && !capturing;
  pp.post.here = false;
  // coper_group_next: one more workgroup sorts the next pass's batch into the set this pass does not use
  GroupJob J = {};
  int n_group = 0;
  if (pp.gnext.ride && !capturing) {
    const int64_t R = dm.gen_fc ? dm.R : 1;
    const int t = h->gcur == 1 ? 2 : 1;
    const coper_handle::GroupSet& g = h->gset[t];
    if (g.slab && pp.gnext.B <= h->ws_queries && group_role_lds_ints(R, 512) * sizeof(int) <= lds) {
      J.B = pp.gnext.B; J.R = R; J.R_all = dm.R; J.shard_lo = h->cfg.shard_lo; J.n_local = dm.n_local; J.cap_small = cap_small;
      J.count = g.rel_count; J.offset = g.rel_offset; J.tiles = g.tiles; J.n_tiles = g.n_tiles; J.perm = g.perm;
      J.sorted_row = g.sorted_row; J.sorted_rid = g.sorted_rid; J.inv_perm = g.inv_perm; J.x3m = g.x3m; J.x3m_slots = X3M_SLOTS;
      J.chk = g.chk;
      J.use_rel = dm.gen_fc ? 1 : 0; J.have_e1_rows = pp.gnext.rows;
      // ids inside the batch this launch stages: the grouping workgroup waits for the staging workgroups (a ticket), then reads
      // the device arrays like any other (one workgroup reading pinned host memory itself manages 0.65 GB/s: measured, 370 us)
      auto in_stage = [&](const int64_t* p) { return stage_now && p && p + J.B > pp.stage.dst && p < pp.stage.dst + pp.stage.n; };
      J.rel64 = pp.gnext.rel; J.e1_64 = pp.gnext.e1;
      J.ticket = h->group_done + 1;
      J.wait_for = (in_stage(pp.gnext.rel) || in_stage(pp.gnext.e1)) ? FUSED_STAGE_WGS : 0;
      J.front = 0;
      if (in_stage(pp.gnext.rel)) J.front = (pp.gnext.rel + J.B) - pp.stage.dst;
      if (in_stage(pp.gnext.e1) && (pp.gnext.e1 + J.B) - pp.stage.dst > J.front) J.front = (pp.gnext.e1 + J.B) - pp.stage.dst;
      if (J.front > pp.stage.n) J.front = pp.stage.n;
      n_group = 1;
      pp.gdone.e1 = pp.gnext.e1; pp.gdone.rel = pp.gnext.rel; pp.gdone.B = pp.gnext.B; pp.gdone.rows = pp.gnext.rows;
      pp.gdone.done = true; pp.gdone.set = t;
    }
  }
  pp.gnext.ride = false; pp.gnext.pending = false;
  const int n_stage = (stage_now || post_now) ? FUSED_STAGE_WGS : 0;
  hipLaunchKernelGGL((k_dense_fused_bf16x3<NFB, WNT>), dim3((unsigned)(n_tile_blocks + n_group + n_stage), (unsigned)nslices, (unsigned)zgroups), dim3(512),
                     lds, s, (const uint4*)h->Wf16_hi, (const uint4*)h->Wf16_lo, A, h->tiles, h->n_tiles, cap_small, dm.nfb,
                     dm.F_pad / 32, nslices, h->ws_queries, dm.d_pad16, h->z_part, Fn, n_group + n_stage, pp.stage.src,
                     stage_now ? pp.stage.n : 0, pp.stage.dst, w16_ks_stride(dm), J, n_group, pp.post.src, post_now ? pp.post.n : 0, pp.post.dst);
  if (stage_now) pp.take_stage();
  if (post_now) pp.take_post();
}

// the constant part of FusedFin, (re)written when the workspace or the parameters move (ensure_workspace / prepare)
int fused_fin_update(coper_handle* h, hipStream_t s) {
  const Dims& dm = h->dm;
  if (!h->enc_bf16 || !h->x3m || !h->perm) return COPER_OK;
  const int cur = h->gcur;
  group_use_set(h, 0);
  // one block of constants per set of grouping arrays (coper_group_next): perm and x3m are the set's
  for (int i = 0; i < 3; ++i) {
    coper_handle::GroupSet& g = h->gset[i];
    if (i > 0 && !g.slab) continue;
    FusedFinConst c;
    c.perm = i ? g.perm : h->perm; c.fc_b = dm.gen_fc ? h->fc_b_rel : h->params["fc_bias"].ptr; c.scale = h->fc_scale; c.shift = h->fc_shift;
    c.w_exp = h->w_exp; c.x3m = i ? g.x3m : h->x3m; c.per_rel_bias = dm.gen_fc ? 1 : 0; c.x_exp = h->x_exp; c.d = dm.d; c.pad = 0;
    void** dev = i ? &g.fused_fin_dev : &h->fused_fin_dev;
    if (!*dev && tracked_malloc(dev, sizeof c) != hipSuccess) return fail(h, COPER_ENOMEM, "hipMalloc failed (fused finalize constants)");
    COPER_HIP_TRY(h, hipMemcpyAsync(*dev, &c, sizeof c, hipMemcpyHostToDevice, s));
    COPER_HIP_TRY(h, hipStreamSynchronize(s));      // (c is on the stack)
    if (i) g.fused_fin_perm = g.perm; else h->fused_fin_perm = h->perm;
  }
  group_snapshot_home(h);
  group_use_set(h, cur);
  return COPER_OK;
}

// the finalize can ride in this kernel's epilogue: one K slice, rows of whole 16-byte quads
bool dense_fused_finalizes(const coper_handle* h, int nslices, const float* h_out) {
  static const bool off = getenv("COPER_FUSED_NO_FINALIZE") != nullptr;     // A/B switch, read once
  const Dims& dm = h->dm;
  const float* fcb = dm.gen_fc ? h->fc_b_rel : h->params.at("fc_bias").ptr;
  return !off && h_out && nslices == 1 && h->x3m && h->fused_fin_dev && h->fused_fin_perm == h->perm && (dm.d & 3) == 0 &&
         ((((uintptr_t)h_out) | ((uintptr_t)fcb) | ((uintptr_t)h->fc_scale) | ((uintptr_t)h->fc_shift)) & 15) == 0;
}

int launch_dense_fused_bf16(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                            int nslices, float* h_fin, hipStream_t s) {
  if (!dense_fused_finalizes(h, nslices, h_fin)) h_fin = nullptr;
#ifndef COPER_FUSED_SPLIT_B
#define COPER_FUSED_SPLIT_B 2048
#endif
  // Small batches leave most CUs idle (one workgroup per tile and slice): split the feature blocks over two
  // (B <= 2048) or four (B <= 1024)
  // workgroups (grid.z; every accumulation chain is unchanged, so h keeps its bits) -- the image / conv work is
  // repeated per group, which small tiles do not notice.
  // one tile per weight set (per-relation weights, on average at most 128 queries per forward relation): stream them
  // past the caches; otherwise let L2 serve the tiles that re-read a set
  const bool wnt = h->dm.gen_fc && B * 2 <= 128 * h->dm.R;
#define FUSED_GO(NFB_, Z_)                                                             \
  {                                                                                    \
    if (wnt) dense_fused_launch<NFB_, true>(h, e1, rel, e1_rows, B, nslices, Z_, h_fin, s);   \
    else dense_fused_launch<NFB_, false>(h, e1, rel, e1_rows, B, nslices, Z_, h_fin, s);      \
  }
  if (B <= COPER_FUSED_SPLIT_B / 2) FUSED_GO(4, (h->dm.nfb + 3) / 4)
  else if (h->dm.nfb == 13 && B > COPER_FUSED_SPLIT_B) FUSED_GO(13, 1)
  else FUSED_GO(8, (h->dm.nfb + 7) / 8)
#undef FUSED_GO
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
