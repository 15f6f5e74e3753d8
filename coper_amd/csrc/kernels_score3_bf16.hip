// k_score_count3_bf16x3 -- the fused score + count pass of the bf16x3 mode (round 3): v_mfma_f32_16x16x32_bf16, software-
// pipelined for ONE wave per SIMD, with the EXACT-BAND output that makes the mode's ranks the ranks of the fp32 chain.
//
// Arithmetic: bf16x3_chain.h (every logit of the mode is that one sequence of K = 16 accumulation steps; a 16x16x32
// instruction carries two of them).  Why this shape: at this kernel's MFMA density the chip holds its clock down, and the
// 16x16x32 loop delivers 1.12x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP (tools/microbench/mfma_shape.hip:
// 1.96 against 1.75 PFLOP/s with the operands re-read from LDS; MI355X_MICROARCH.md, DVFS item 7).
//
// Schedule (the structure of round 2's kernel, kernels_score2_bf16.hip, re-cut for the new shape): a workgroup is 4 waves,
// one per SIMD, 128 queries in LDS as f3 fragments (8 column blocks of 16); a wave owns two 32-entity blocks of a row of 8
// and alternates between them: while the MFMAs of block M run (per step and column block b: 6 instructions on two
// independent accumulator chains, or 4 in the tail step), the compare epilogue of block 1-M is issued between them, one or
// two values per region; accumulators live in AGPRs and are read out at the point of use; the first instruction of a chain
// takes pred_bias as its C operand; query fragments come from LDS two regions ahead (a ring of 4 register pairs), entity
// fragments straight from the f3 image into registers PD steps ahead (two sets of PD slots), across block and row
// boundaries.  All register state is indexed with template constants.
//
// The exact band.  The bf16x3 logit differs from the fp32-chain logit of the same (h, E) by a few
// 1e-5 relative to |h||E| -- enough to move a rank whenever a competitor's logit is that close to the target's.  So the
// kernel does not decide such comparisons: with  t_hi = t + tau_q,  t_lo = t - tau_q  (tau_q = 2 kappa (|h_q| max|E_e| +
// max|bias|), kernels below) it counts only  s > t_hi  (greater under either arithmetic) and writes ONE BIT per logit,
// "t_lo <= s <= t_hi", to a mask (the 128 bits a lane produces per row of a wave = one 16-byte store; 1/32 of the bytes of
// the logits it stands for).  k_band_exact then walks the mask and decides every marked pair that is not a known answer
// with the fp32 chain itself (exact_chain on the fp32 rows the caller registered) -- a few pairs per query.  Ranks and
// tie counts of the mode are therefore those of COPER_SCORE_F32 on the same h, bit for bit, as long as the bf16x3 error
// stays inside tau (tests measure the margin; coper_config.rank_band_kappa widens it up to the proven worst case).
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "bf16x3_chain.h"
#include "coper_internal.h"
#include "tail_tile.h"

namespace coper {

size_t score_count3_mask_words_bytes(const coper_handle* h, int64_t Bc);

#define SC3_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef COPER_SC3_LD
#define COPER_SC3_LD 2
#endif
constexpr int SC3_LD = COPER_SC3_LD;     // regions between an LDS read of query fragments and their use (ring of 4: 1..3)
static_assert(SC3_LD >= 1 && SC3_LD <= 3, "ring of four register pairs");

#ifndef COPER_SC3_MB
#define COPER_SC3_MB 4
#endif
// 16-row blocks per entity block of a wave.  A pair of query fragments read from LDS feeds 3 MB instructions: with MB = 2
// (32-entity blocks, the first form of this kernel: 0.30 ms) the wave was bound by the ISSUE of its ds_read_b128 -- an
// ablation that re-used every fragment pair for two regions ran in 0.23 ms -- so MB = 4: 64-entity blocks, both accumulator
// sets fill the 256 AGPRs.
constexpr int SC3_MB = COPER_SC3_MB;
static_assert(SC3_MB == 2 || SC3_MB == 4, "mask words are written as whole 16-byte pieces");
// Block maxima of the top-k launches.  GM = 1: one per (32 entities, query) -- the eight values of two consecutive 16-row
// blocks; GM = 2: one per (the wave's 64-entity block, query) -- sixteen values, half the cross-lane reductions and stores in
// this kernel and half the bytes for the threshold kernel's three sweeps (kernels_topk_bf16.hip expands a candidate block into
// its two 32-entity halves before the re-scoring, which therefore doubles).  Large tables take GM = 2 (10 M entities, 4,096
// queries, top-10: 48.9 -> 45.6 ms), small ones GM = 1 (FB15k-237, k = 10: 0.69 against 0.84 ms): topk_expand (coper_internal.h).
__host__ __device__ constexpr int sc3_gmask(int GM) { return GM == 2 ? 15 : 7; }        // a maximum is complete at value V with (V & mask) == mask
__host__ __device__ constexpr int sc3_gm_rows(int GM) { return GM == 2 ? 1 : SC3_MB / 2; }   // rows of gmax per entity block of 16 SC3_MB rows
template <int NP, int TAIL, int PD, int GM>
struct SC3 {
  static constexpr int MB = SC3_MB;
  static constexpr int NS = NP + TAIL;                    // steps per half-row
  static constexpr int NB = 8;                            // 16-query column blocks of the tile
  static constexpr int NR = NS * NB;                      // regions per half-row
  static constexpr int NV = 32 * MB;                      // accumulator values per lane per half-row: MB row blocks x 8 x 4
  static constexpr int CH = (NV + NS - 1) / NS;           // epilogue values handled per step
  static constexpr int G = (NS + PD - 1) / PD;            // prefetch groups per half-row
  f32x4 acc[2][MB][NB];                                   // [entity block M][16-row block m2][column block b]
  uint4 a0[2][PD][MB], a1[2][PD][MB];                     // entity fragments: two sets of PD steps, [m2]; reg 0 / reg 1 of the step
  uint4 q0[4], q1[4];                                     // query fragments of four consecutive regions (ring)
  f32x4 biasv[2][MB];                                     // [M][m2]
  float thi[NB], tlo[NB];
  int cg[NB];
  unsigned mk[2 * MB], mg[2 * MB];                        // (logit >= t_lo) / (logit > t_hi): word MB M + (V >> 5), value V at bit 31 - (V & 31)
  float mx, px;
  int sexp;                                               // e_E + e_h: the power of two the accumulators carry (split16.h)
  const uint4* mask_base;                                 // (to find a row's summary word from its mask pointer)
  unsigned long long* summ_base;
  int64_t gm_stride;
};

// a 16-byte LDS read through an explicit LDS pointer (HIP's uint4 struct has no assignment from another address space)
typedef unsigned sc3_u4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 sc3_lds_hi(const uint4 __attribute__((address_space(3)))* p, const int i) {
  const sc3_u4n v = ((const sc3_u4n __attribute__((address_space(3)))*)p)[i];
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Entity fragments: global loads from a 64-bit vector address per stream; constants beyond the 13-bit immediate cost a v_add_co /
// v_addc pair and the wait states of their carry (58 loads per half-row).  Round 4 tried BUFFER loads instead (the build switch is gone:
// a resource descriptor per row in four scalar registers, the lane's 16-byte slot as the one vector offset, the register's place
// inside the row as a scalar offset: `s_movk` + `buffer_load`, no vector instruction, wait states 29 -> 7 per half-row): 0.2645
// against 0.2551 ms on the same box, three alternating runs -- the address arithmetic is not what the loads cost.
struct sc3_rsrc_t { const char* p; };
__device__ __forceinline__ sc3_rsrc_t sc3_make_rsrc(const void* p) { return sc3_rsrc_t{(const char*)p}; }
__device__ __forceinline__ uint4 sc3_bload(const sc3_rsrc_t r, const int voff, const int soff) { return *(const uint4*)(r.p + soff + voff); }
struct SC3Ptrs {
  sc3_rsrc_t ra;               // this row's two entity blocks: block M, 16-row block m2, step t, register w at byte
  sc3_rsrc_t rn;               //   ((M BLK_REGS + (m2 NS + t) 2 + w) 64 + lane) 16;   rn: the next row's block 0
  int voff;                    // lane * 16
  const uint4* hl;       // the query tile in LDS, lane included
  const uint4 __attribute__((address_space(3)))* hl_hi;    // ... its part beyond 64 KiB (an LDS pointer the compiler cannot fold back)
};


// what follows the comparisons of value V: every 32 values the word pair gives the counts, every 8 (top-k launches) the block
// maximum is reduced across lanes and stored, the row's last value stores the band words that carry a bit
template <int NP, int TAIL, int PD, int GM, int M, int V>
__device__ __forceinline__ void sc3_value_tail(SC3<NP, TAIL, PD, GM>& S, const int lane, const bool store_ok, float* __restrict__ gm_row,
                                               const int64_t gm_col, uint4* __restrict__ mask_row) {
  constexpr int MB = SC3_MB, NV = 32 * MB;
  constexpr int b = V / (4 * MB), m2 = (V >> 2) % MB, w = MB * M + (V >> 5);
  if constexpr ((V & 31) == 31) {      // a word is complete: 32 / (4 MB) queries' worth
    constexpr int QW = 32 / (4 * MB), B0 = (V - 31) / (4 * MB);       // column blocks in the word, the first of them
    constexpr unsigned FM = QW == 1 ? 0xFFFFFFFFu : (1u << (4 * MB)) - 1u;
#pragma unroll
    for (int i = 0; i < QW; ++i)       // value V sits at bit 31 - (V & 31): column block B0 + i in bits [32 - 4 MB (i + 1), 32 - 4 MB i)
      S.cg[B0 + i] += __builtin_popcount(S.mg[w] & (FM << (32 - 4 * MB * (i + 1))));
    S.mk[w] &= ~S.mg[w];
  }
  if constexpr (GM) {
    if constexpr ((V & sc3_gmask(GM)) == sc3_gmask(GM)) {
      // the other rows of the 32-entity block sit in lanes + 16, + 32, + 48: two lane swaps inside the vector unit
      // (v_permlane32_swap / v_permlane16_swap; __shfl_xor goes through the LDS pipe and its wait falls on the query-
      // fragment reads in flight).  After "swap a, b" with a == b: a = {lower, lower}, b = {upper, upper} halves (rows).
      float a = S.mx, b2 = S.mx, mxx;
      // (a swap must not read a register a vector instruction wrote within the last two wait states: the s_nop 1)
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"
                   "v_permlane16_swap_b32 %0, %1\n\tv_max_f32 %2, %0, %1"
                   : "+v"(a), "+v"(b2), "=&v"(mxx));
      if (store_ok && lane < 16) gm_row[(sc3_gm_rows(GM) > 1 ? (m2 >> 1) : 0) * S.gm_stride + gm_col + b * 16 + lane] = mxx;
    }
  }
  if constexpr (M == 1 && V == NV - 1) {     // the row's 64 MB band bits of this lane are complete
#ifndef COPER_DBG_SC3_NO_BAND
    // only words that carry a bit are written (about one in a hundred), and one 64-bit summary per row of the wave says
    // which lanes wrote: k_band_exact reads 8 bytes per row of a wave instead of its 1 - 2 KiB
    unsigned any = 0u;
#pragma unroll
    for (int i = 0; i < 2 * MB; ++i) any |= S.mk[i];
    const bool nz = any != 0u;
    const unsigned long long which = __ballot(nz);
    if (store_ok) {
      if (nz) {
#pragma unroll
        for (int i = 0; i < MB / 2; ++i) mask_row[lane * (MB / 2) + i] = make_uint4(S.mk[4 * i], S.mk[4 * i + 1], S.mk[4 * i + 2], S.mk[4 * i + 3]);
      }
      if (lane == 0) S.summ_base[(mask_row - S.mask_base) / (64 * (MB / 2))] = which;
    }
#endif
  }
}

// value V = 4 MB b + 4 m2 + j of block M: entity row 16 m2 + 4 (lane >> 4) + j of the block, query 16 b + (lane & 15) of the tile
template <int NP, int TAIL, int PD, int GM, int M, int V>
__device__ __forceinline__ void sc3_value(SC3<NP, TAIL, PD, GM>& S, const int lane, const bool store_ok, float* __restrict__ gm_row,
                                          const int64_t gm_col, uint4* __restrict__ mask_row) {
  constexpr int MB = SC3_MB;
  [[maybe_unused]] constexpr int NV = 32 * MB;
  constexpr int b = V / (4 * MB), m2 = (V >> 2) % MB, j = V & 3, w = MB * M + (V >> 5);
#ifdef COPER_DBG_SC3_EPI_R0   /* ablation: one value per column block keeps the chains alive, the epilogue nearly free */
  if constexpr ((V & 7) != 0) return;
#endif
  float sc;
  // The accumulator is read out of its AGPR here, at the point of use, and compared with both edges of the band; each
  // result is shifted into a mask word (m = 2 m + bit: one add-with-carry).  No counting per value and no scalar
  // instruction: the first form (v_cmp into an SGPR pair, add-with-carry from it, s_andn2, add-with-carry) paid wait
  // states at every VALU -> SGPR -> VALU hand-over, in a kernel that is bound by its instruction issue.  Every 32 values
  // the word pair gives  band = ge & ~gt  and  count += popcount(gt) per query.
#ifdef COPER_DBG_SC3_NO_BAND
  asm volatile("v_accvgpr_read_b32 %1, %2\n\tv_cmp_gt_f32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
               : "+v"(S.mg[w]), "=&v"(sc)
               : "a"(S.acc[M][m2][b][j]), "v"(S.thi[b])
               : "vcc");
#else
  asm volatile(
      "v_accvgpr_read_b32 %2, %3\n\t"
      "v_cmp_gt_f32 vcc, %2, %4\n\t"
      "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f32 vcc, %2, %5\n\t"
      "v_addc_co_u32 %1, vcc, %1, %1, vcc"
      : "+v"(S.mg[w]), "+v"(S.mk[w]), "=&v"(sc)
      : "a"(S.acc[M][m2][b][j]), "v"(S.thi[b]), "v"(S.tlo[b])
      : "vcc");
#endif
  if constexpr (GM) {   // block maxima per (32 entities, query): the eight values of two consecutive 16-row blocks
    // two values per v_max3 (the kernel is bound by instruction issue: a v_max per value and the library fmaxf's quieting
    // moves were a fifth of the top-k launch), written as instructions because the values come out of an asm block
    if constexpr ((V & 1) == 0) {
      S.px = sc;
    } else if constexpr ((V & sc3_gmask(GM)) == 1) {
      asm volatile("v_max_f32 %0, %1, %2" : "=v"(S.mx) : "v"(S.px), "v"(sc));
    } else {
      asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(S.mx) : "v"(S.px), "v"(sc));
    }
  }
  sc3_value_tail<NP, TAIL, PD, GM, M, V>(S, lane, store_ok, gm_row, gm_col, mask_row);
}

template <int NP, int TAIL, int PD, int GM, int M, int V0, int... I>
__device__ __forceinline__ void sc3_values(SC3<NP, TAIL, PD, GM>& S, const int lane, const bool store_ok, float* __restrict__ gm_row,
                                           const int64_t gm_col, uint4* __restrict__ mask_row, std::integer_sequence<int, I...>) {
  (sc3_value<NP, TAIL, PD, GM, M, V0 + I>(S, lane, store_ok, gm_row, gm_col, mask_row), ...);
}

// ---- the epilogue of a value in two pieces (3 and 2 vector instructions), to be placed behind two consecutive MFMAs.
// A 16x16x32 MFMA occupies the matrix pipe for 16 cycles and keeps the vector-issue port for the first 8 of them: TWO vector
// instructions fit in its shadow, a third delays the next MFMA by 4 cycles.  The one-block form above (5 instructions behind
// one MFMA: sc3_value) cost 12 cycles of an idle matrix pipe per value -- 256 values a row, the whole gap between the
// measured 29,300 cycles per row and the 20,480 of its MFMAs (build/isa statistics: 991 of 1,280 MFMAs with nothing behind
// them, 154 with 5 vector instructions, 47 with 10).  The carry travels in an SGPR pair from piece to piece (VCC could be
// clobbered by the compiler's own address arithmetic between two asm statements); the MFMA between producer and consumer
// hides the VALU -> SGPR -> VALU hand-over that made round 3's first scalar form slow.
template <int NP, int TAIL, int PD, int GM, int M, int V>
__device__ __forceinline__ void sc3_piece1(SC3<NP, TAIL, PD, GM>& S, float& sc) {
  constexpr int MB = SC3_MB, b = V / (4 * MB), m2 = (V >> 2) % MB, j = V & 3, w = MB * M + (V >> 5);
  asm volatile("v_accvgpr_read_b32 %1, %2\n\tv_cmp_gt_f32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
               : "+v"(S.mg[w]), "=&v"(sc) : "a"(S.acc[M][m2][b][j]), "v"(S.thi[b]) : "vcc");
}
template <int NP, int TAIL, int PD, int GM, int M, int V>
__device__ __forceinline__ void sc3_piece2(SC3<NP, TAIL, PD, GM>& S, const float sc, const int lane, const bool store_ok,
                                           float* __restrict__ gm_row, const int64_t gm_col, uint4* __restrict__ mask_row) {
  constexpr int MB = SC3_MB, b = V / (4 * MB), w = MB * M + (V >> 5);
#ifndef COPER_DBG_SC3_NO_BAND
  asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(S.mk[w]) : "v"(sc), "v"(S.tlo[b]) : "vcc");
#endif
  if constexpr (GM) {
    if constexpr ((V & 1) == 0) S.px = sc;
    else if constexpr ((V & sc3_gmask(GM)) == 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(S.mx) : "v"(S.px), "v"(sc));
    else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(S.mx) : "v"(S.px), "v"(sc));
  }
  sc3_value_tail<NP, TAIL, PD, GM, M, V>(S, lane, store_ok, gm_row, gm_col, mask_row);
}

template <int NP, int TAIL, int PD, int GM, int M, int... m2>
__device__ __forceinline__ void sc3_load_bias_(SC3<NP, TAIL, PD, GM>& S, const float4* __restrict__ bp, std::integer_sequence<int, m2...>) {
  ((S.biasv[M][m2] = f32x4{x3_scale(bp[4 * m2].x, S.sexp), x3_scale(bp[4 * m2].y, S.sexp), x3_scale(bp[4 * m2].z, S.sexp),
                           x3_scale(bp[4 * m2].w, S.sexp)}), ...);
}
template <int NP, int TAIL, int PD, int GM, int M>
__device__ __forceinline__ void sc3_load_bias(SC3<NP, TAIL, PD, GM>& S, const float* __restrict__ bias_pad, const int64_t blk, const int lane) {
  sc3_load_bias_<NP, TAIL, PD, GM, M>(S, (const float4*)(bias_pad + blk * (16 * SC3_MB) + 4 * (lane >> 4)), std::make_integer_sequence<int, SC3_MB>{});
}

// the instructions of region (step s, column block b) on the MB accumulator chains of block M, chains interleaved
template <int NP, int TAIL, int PD, int GM, int M, int s, int b, int sa, int sl, int rs, bool tail, int... m2>
__device__ __forceinline__ void sc3_mfmas(SC3<NP, TAIL, PD, GM>& S, std::integer_sequence<int, m2...>) {
  // (e.reg0, q.reg1): T1 of both k-steps (tail: T1 then T2); step 0 starts the chains from pred_bias
  if constexpr (s == 0) ((S.acc[M][m2][b] = BX3_MFMA16(S.a0[sa][sl][m2], S.q1[rs], S.biasv[M][m2])), ...);
  else ((S.acc[M][m2][b] = BX3_MFMA16(S.a0[sa][sl][m2], S.q1[rs], S.acc[M][m2][b])), ...);
  // (e.reg1, q.reg0): T2 of both k-steps (tail: T3 and an all-zero half)
  ((S.acc[M][m2][b] = BX3_MFMA16(S.a1[sa][sl][m2], S.q0[rs], S.acc[M][m2][b])), ...);
  // (e.reg1, q.reg1): T3 of both k-steps
  if constexpr (!tail) ((S.acc[M][m2][b] = BX3_MFMA16(S.a1[sa][sl][m2], S.q1[rs], S.acc[M][m2][b])), ...);
}

template <int NP, int TAIL, int PD, int GM, int M, int v0, int cnt, int K0, int... U>
__device__ __forceinline__ void sc3_slot_pieces(SC3<NP, TAIL, PD, GM>& S, float* sc, const int lane,
                                                const bool store_ok, float* __restrict__ gm_row, const int64_t gm_col,
                                                uint4* __restrict__ mask_row, std::integer_sequence<int, U...>) {
  // (M here is the block whose values are finished: the OTHER block of the region that issues the MFMAs)
  ([&] {
    constexpr int k = K0 + U, vi = k / 2, pk = k % 2;
    if constexpr (vi < cnt) {
      if constexpr (pk == 0) sc3_piece1<NP, TAIL, PD, GM, M, v0 + vi>(S, sc[vi]);
      else sc3_piece2<NP, TAIL, PD, GM, M, v0 + vi>(S, sc[vi], lane, store_ok, gm_row, gm_col, mask_row);
    }
  }(), ...);
}

// A region as ONE asm block (sc3_region_asm.inc, generated): its MFMAs in the order of sc3_mfmas with the other block's
// epilogue instructions two behind each MFMA -- the order the microbenchmark asks for (mfma_valu_mix.hip: instructions spread
// behind the MFMAs cost the matrix pipe nothing, clumps of 5 and 10 cost 7 - 23 %) and the compiler would not keep.  Registers
// stay the compiler's (named operands).  Shapes: steps s > 0 (step 0 starts the chains from pred_bias; with the bias quads as
// four more operands the allocator spilled 174 registers), 1 - 3 values that share one mask word; top-k launches fold their
// block maxima in (sc3_region_asm_gm.inc).  Everything else takes the form above.
template <int NP, int TAIL, int PD, int GM, int M, int s, int b, int sa, int sl, int rs, bool tail, int v0, int cnt>
__device__ __forceinline__ void sc3_region_asm(SC3<NP, TAIL, PD, GM>& S, const SC3Ptrs& X, const int lane, const bool store_ok, float* __restrict__ gm_row,
                                               const int64_t gm_col, uint4* __restrict__ mask_row) {
  constexpr int MB = SC3_MB;
  static_assert(MB == 4 && cnt >= 1 && cnt <= 3, "sc3_region_asm: shape not covered");
  constexpr int w = MB * (1 - M) + (v0 >> 5);
  constexpr int V1 = cnt > 1 ? v0 + 1 : v0, V2 = cnt > 2 ? v0 + 2 : V1;
  constexpr int xb0 = v0 / (4 * MB), xm0 = (v0 >> 2) % MB, xj0 = v0 & 3;
  constexpr int xb1 = V1 / (4 * MB), xm1 = (V1 >> 2) % MB, xj1 = V1 & 3;
  constexpr int xb2 = V2 / (4 * MB), xm2 = (V2 >> 2) % MB, xj2 = V2 & 3;
  constexpr int bA = xb0, bB = xb2;
  constexpr int second = xb1 != xb0 ? 1 : (cnt > 2 && xb2 != xb0 ? 2 : cnt);
  float sc;
  typedef unsigned sc3_u4 __attribute__((ext_vector_type(4)));     // (a HIP uint4 is a struct: not a register operand)
#define SC3_Q(x) (*(const sc3_u4*)&(x))
  if constexpr (GM) {       // top-k launches: block maxima folded in (even values wait in S.px, odd ones fold the pair into S.mx)
    constexpr int vm = v0 & sc3_gmask(GM);
    static_assert(!(cnt == 3 && vm == sc3_gmask(GM)), "the maximum of a group is stored after the block: it must not hold the next group's first pair");
    if constexpr (GM == 2) {
#include "sc3_region_asm_gm64.inc"
    } else {
#include "sc3_region_asm_gm.inc"
    }
  } else {
#include "sc3_region_asm.inc"
  }
#undef SC3_Q
  sc3_value_tail<NP, TAIL, PD, GM, 1 - M, v0>(S, lane, store_ok, gm_row, gm_col, mask_row);
  if constexpr (cnt > 1) sc3_value_tail<NP, TAIL, PD, GM, 1 - M, v0 + 1>(S, lane, store_ok, gm_row, gm_col, mask_row);
  if constexpr (cnt > 2) sc3_value_tail<NP, TAIL, PD, GM, 1 - M, v0 + 2>(S, lane, store_ok, gm_row, gm_col, mask_row);
}

// One slot of a region: MFMA number I of the region (term I / MB on chain I % MB) and, behind it, PP pieces of the other
// block's epilogue (piece k of the region = piece k % 3 of value v0 + k / 3); a scheduling barrier pins the order.
template <int NP, int TAIL, int PD, int GM, int M, int s, int b, int sa, int sl, int rs, bool tail, int v0, int cnt, int PP, int I>
__device__ __forceinline__ void sc3_slot(SC3<NP, TAIL, PD, GM>& S, float* sc, const int lane, const bool store_ok,
                                         float* __restrict__ gm_row, const int64_t gm_col, uint4* __restrict__ mask_row) {
  constexpr int MB = SC3_MB, NM = tail ? 2 * MB : 3 * MB;
  if constexpr (I < NM) {
    constexpr int t = I / MB, m2 = I % MB;
    if constexpr (t == 0) {
      if constexpr (s == 0) S.acc[M][m2][b] = BX3_MFMA16(S.a0[sa][sl][m2], S.q1[rs], S.biasv[M][m2]);
      else S.acc[M][m2][b] = BX3_MFMA16(S.a0[sa][sl][m2], S.q1[rs], S.acc[M][m2][b]);
    } else if constexpr (t == 1) {
      S.acc[M][m2][b] = BX3_MFMA16(S.a1[sa][sl][m2], S.q0[rs], S.acc[M][m2][b]);
    } else {
      S.acc[M][m2][b] = BX3_MFMA16(S.a1[sa][sl][m2], S.q1[rs], S.acc[M][m2][b]);
    }
  }
#ifndef COPER_DBG_SC3_NO_EPI
  sc3_slot_pieces<NP, TAIL, PD, GM, 1 - M, v0, cnt, I * PP>(S, sc, lane, store_ok, gm_row, gm_col, mask_row, std::make_integer_sequence<int, PP>{});
#endif
  SC3_FENCE();
}

template <int NP, int TAIL, int PD, int GM, int M, int s, int b, int sa, int sl, int rs, bool tail, int v0, int cnt, int PP, int... I>
__device__ __forceinline__ void sc3_slots(SC3<NP, TAIL, PD, GM>& S, float* sc, const int lane, const bool store_ok,
                                          float* __restrict__ gm_row, const int64_t gm_col, uint4* __restrict__ mask_row,
                                          std::integer_sequence<int, I...>) {
  (sc3_slot<NP, TAIL, PD, GM, M, s, b, sa, sl, rs, tail, v0, cnt, PP, I>(S, sc, lane, store_ok, gm_row, gm_col, mask_row), ...);
}

// Region (step s, column block b) of block M: in front, one entity-fragment load PD steps ahead (regions b < 2 MB: the
// registers of a step) and the two LDS reads of the region SC3_LD ahead; then the instructions of the MB accumulator
// chains interleaved; behind them this region's share of the other block's epilogue.
template <int NP, int TAIL, int PD, int GM, int M, int s, int b>
__device__ __forceinline__ void sc3_region(SC3<NP, TAIL, PD, GM>& S, const SC3Ptrs& X, const int lane, const bool prev_valid,
                                           float* __restrict__ gm_row, const int64_t gm_col, uint4* __restrict__ mask_row) {
  typedef SC3<NP, TAIL, PD, GM> ST;
  constexpr int MB = SC3_MB, NS = ST::NS, NR = ST::NR, G = ST::G, CH = ST::CH, NV = ST::NV;
  constexpr bool tail = TAIL && s == NP;
  constexpr int PA = (G & 1) ? M : 0;        // entity-fragment set of this block's step 0
  constexpr int PA_NEXT = (PA + G) & 1;      // ... of the next block's step 0
  constexpr int sa = (PA + s / PD) & 1, sl = s % PD, tk = s + PD;
#ifndef COPER_DBG_SC3_SKIP_GL
  if constexpr (b < 2 * MB) {
    constexpr int m2 = b >> 1, wh = b & 1;
    constexpr int BLKB = MB * NS * 2 * 1024;     // bytes of one entity block's registers
    if constexpr (tk < NS) {
      constexpr int ta = (PA + tk / PD) & 1, tl = tk % PD;
      if constexpr (wh == 0) S.a0[ta][tl][m2] = sc3_bload(X.ra, X.voff, M * BLKB + ((m2 * NS + tk) * 2 + 0) * 1024);
      else S.a1[ta][tl][m2] = sc3_bload(X.ra, X.voff, M * BLKB + ((m2 * NS + tk) * 2 + 1) * 1024);
    } else if constexpr (M == 0) {           // step tk - NS of this row's block 1
      constexpr int u = tk - NS;
      if constexpr (wh == 0) S.a0[PA_NEXT][u][m2] = sc3_bload(X.ra, X.voff, BLKB + ((m2 * NS + u) * 2 + 0) * 1024);
      else S.a1[PA_NEXT][u][m2] = sc3_bload(X.ra, X.voff, BLKB + ((m2 * NS + u) * 2 + 1) * 1024);
    } else {                                 // ... of the next row's block 0
      constexpr int u = tk - NS;
      if constexpr (wh == 0) S.a0[PA_NEXT][u][m2] = sc3_bload(X.rn, X.voff, ((m2 * NS + u) * 2 + 0) * 1024);
      else S.a1[PA_NEXT][u][m2] = sc3_bload(X.rn, X.voff, ((m2 * NS + u) * 2 + 1) * 1024);
    }
  }
#endif
  constexpr int R = s * 8 + b;
#ifndef COPER_DBG_SC3_SKIP_LDS
#ifdef COPER_DBG_SC3_HALF_LDS   /* ablation (wrong results): every second region re-uses its predecessor's query fragments */
  if constexpr ((R & 1) == 0)
#endif
  {
    constexpr int R2 = (R + SC3_LD) % NR, s2 = R2 / 8, b2 = R2 % 8;
#ifdef COPER_DBG_SC3_DUMMY_LDS
    if constexpr (((R + SC3_LD) & 3) != 0)
#endif
    {
      // (a ds_read offset holds 16 bits: registers beyond 64 KiB go through the second base instead of an add per read)
      constexpr int i0 = ((b2 * NS + s2) * 2 + 0) * 64, i1 = i0 + 64;
      if constexpr (i0 <= 4095) S.q0[(R + SC3_LD) & 3] = X.hl[i0]; else S.q0[(R + SC3_LD) & 3] = sc3_lds_hi(X.hl_hi, i0 - 4096);
#ifdef COPER_DBG_SC3_ONE_LDS   /* ablation (wrong results): one of the two reads */
      if constexpr (R < 4)
#endif
      if constexpr (i1 <= 4095) S.q1[(R + SC3_LD) & 3] = X.hl[i1]; else S.q1[(R + SC3_LD) & 3] = sc3_lds_hi(X.hl_hi, i1 - 4096);
    }
  }
#endif
#ifdef COPER_DBG_SC3_HALF_LDS
  constexpr int rs = (R & ~1) & 3;
#elif defined(COPER_DBG_SC3_DUMMY_LDS)   /* ablation: the reads are issued and waited for, the instructions use fixed fragments */
  asm volatile("" :: "v"(S.q0[R & 3].x), "v"(S.q1[R & 3].x));
  constexpr int rs = 0;
#else
  constexpr int rs = R & 3;
#endif
#ifdef COPER_DBG_SC3_LOADS_FIRST
  SC3_FENCE();     // experiment: the region's loads are issued before its first instruction of the matrix pipe
#endif
  // epilogue of the other block: this step's chunk of CH values is dealt to the eight regions in order
  constexpr int c0 = b * CH / 8, c1 = (b + 1) * CH / 8, v0 = s * CH + c0;
  constexpr int cnt = v0 >= NV ? 0 : (v0 + (c1 - c0) > NV ? NV - v0 : c1 - c0);
#if !defined(COPER_DBG_SC3_NO_EPI) && !defined(COPER_DBG_SC3_NO_BAND) && !defined(COPER_DBG_SC3_EPI_R0)
  if constexpr (MB == 4 && s > 0 && cnt >= 1 && cnt <= 3 && (v0 >> 5) == ((v0 + cnt - 1) >> 5) && !(GM && cnt == 3 && (v0 & sc3_gmask(GM)) == sc3_gmask(GM))) {
    sc3_region_asm<NP, TAIL, PD, GM, M, s, b, sa, sl, rs, tail, v0, cnt>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
    SC3_FENCE();
    return;
  }
#endif
  sc3_mfmas<NP, TAIL, PD, GM, M, s, b, sa, sl, rs, tail>(S, std::make_integer_sequence<int, MB>{});
#ifndef COPER_DBG_SC3_NO_EPI
  // (Also tried: pinning value i between terms i and i + 1 of the region by a never-read accumulator operand, so that no two
  // values end up behind one MFMA -- 0.275 against 0.263 ms: the compiler's own placement is the better one.)
  if constexpr (cnt > 0) sc3_values<NP, TAIL, PD, GM, 1 - M, v0>(S, lane, prev_valid, gm_row, gm_col, mask_row, std::make_integer_sequence<int, cnt>{});
#endif
  SC3_FENCE();
}

template <int NP, int TAIL, int PD, int GM, int M, int s>
__device__ __forceinline__ void sc3_step(SC3<NP, TAIL, PD, GM>& S, const SC3Ptrs& X, const float* __restrict__ bias_pad,
                                         const int64_t bias_blk_next, const int lane, const bool prev_valid, float* __restrict__ gm_row,
                                         const int64_t gm_col, uint4* __restrict__ mask_row) {
  sc3_region<NP, TAIL, PD, GM, M, s, 0>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 1>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 2>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 3>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 4>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 5>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 6>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  sc3_region<NP, TAIL, PD, GM, M, s, 7>(S, X, lane, prev_valid, gm_row, gm_col, mask_row);
  if constexpr (s == 0) {
    // pred_bias of this block in the NEXT row: the chains have consumed biasv[M] (program order)
    sc3_load_bias<NP, TAIL, PD, GM, M>(S, bias_pad, bias_blk_next, lane);
    SC3_FENCE();
  }
}

template <int NP, int TAIL, int PD, int GM, int M, int... K>
__device__ __forceinline__ void sc3_half(SC3<NP, TAIL, PD, GM>& S, const SC3Ptrs& X, const float* __restrict__ bias_pad,
                                         const int64_t bias_blk_next, const int lane, const bool prev_valid, float* __restrict__ gm_row,
                                         const int64_t gm_col, uint4* __restrict__ mask_row, std::integer_sequence<int, K...>) {
  (sc3_step<NP, TAIL, PD, GM, M, K>(S, X, bias_pad, bias_blk_next, lane, prev_valid, gm_row, gm_col, mask_row), ...);
}

template <int NP, int TAIL, int PD, int GM, int J, int... m2>
__device__ __forceinline__ void sc3_prologue_a_(SC3<NP, TAIL, PD, GM>& S, const uint4* __restrict__ pa, std::integer_sequence<int, m2...>) {
  constexpr int NS = NP + TAIL, JJ = J < NS ? J : NS - 1;
  ((S.a0[0][J][m2] = pa[((m2 * NS + JJ) * 2 + 0) * 64], S.a1[0][J][m2] = pa[((m2 * NS + JJ) * 2 + 1) * 64]), ...);
}
template <int NP, int TAIL, int PD, int GM, int... J>
__device__ __forceinline__ void sc3_prologue_a(SC3<NP, TAIL, PD, GM>& S, const uint4* __restrict__ pa, std::integer_sequence<int, J...>) {
  (sc3_prologue_a_<NP, TAIL, PD, GM, J>(S, pa, std::make_integer_sequence<int, SC3_MB>{}), ...);
}
template <int NP, int TAIL, int PD, int GM, int... J>
__device__ __forceinline__ void sc3_prologue_a1(SC3<NP, TAIL, PD, GM>& S, std::integer_sequence<int, J...>) {   // ablation builds only
  for (int m2 = 0; m2 < SC3_MB; ++m2) ((S.a0[1][J][m2] = S.a0[0][J][m2], S.a1[1][J][m2] = S.a1[0][J][m2]), ...);
}

#ifdef COPER_DBG_CLOCK
// diagnostic build (tools/ab_build.py): shader clock held inside the kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz;
// the stamps go to a buffer of their own, no output depends on them
__device__ unsigned long long g_sc3_clk[2 * 1024];
#endif

// Ef3: the entities' f3 image; Hf3: the queries' (one 128-query tile = 8 column blocks = 16 NS KiB, copied to LDS as it lies);
// tband[q] = {t_lo, t_hi}; mask: [tile][row][wave][lane] MB / 2 x 16 bytes (band bits of the 32 MB entities x 128 queries of a
// wave's row), written only where a bit is set; summ: [tile][row][wave] 8 bytes: the lanes whose mask words were written
template <int NP, int TAIL, int PD, int GM>
__global__ __launch_bounds__(256, 1) void k_score_count3_bf16x3(const uint4* __restrict__ Ef3, const float* __restrict__ bias_pad,
                                                                 const uint4* __restrict__ Hf3, const float2* __restrict__ tband,
                                                                 int64_t B, int64_t rows_per_tile, int64_t total_rows,
                                                                 int32_t* __restrict__ ng, uint4* __restrict__ mask,
                                                                 unsigned long long* __restrict__ summ,
                                                                 float* __restrict__ gmax, int64_t gm_stride, int64_t rows_per_item,
                                                                 const int32_t* __restrict__ x3s) {
  typedef SC3<NP, TAIL, PD, GM> ST;
  constexpr int NS = ST::NS, NB = ST::NB, NV = ST::NV;
  static_assert(PD <= NS, "the prefetch reaches at most one half-row ahead");
  extern __shared__ uint4 hl3[];  // [NB][NS][2][64]
  constexpr int TILE_REGS = NB * NS * 2;          // KiB of a query tile
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // Two ways to deal the rows (a row: this workgroup's 8 entity blocks against one query tile):
  //  rows_per_item == 0: an equal contiguous share for each of num_cus workgroups -- entity tables that L2 / MALL hold whole;
  //  rows_per_item  > 0: one workgroup per item of that many rows of one tile, numbered so that the items of an XCD
  //    (workgroups reach the XCDs round-robin: blockIdx & 7) run through the tiles of one chunk of entity rows before the
  //    next chunk.  Workgroups start in blockIdx order, so those of an XCD stream the same entity rows at the same time
  //    however long the launch runs.  With the static split the 16 workgroups on one stretch of a 10M-row table drifted
  //    apart over their 20 ms and L2 served half of what they shared (PMC: 80 GB fetched per launch for a 10 GB table).
  // (static split: a tile start -- drain of the accumulators, 104 KB of LDS, prologue -- costs a workgroup about a third of a
  // row (4.5 us against 13.2 us per row at FB15k-237 shapes), so the share is cut on an axis where every tile is
  // rows_per_tile + 1/3 long: workgroups that cross into a new tile get fewer rows)
  int64_t r_begin, r_end;
  {
#ifndef COPER_SC3_TILE_COST
#define COPER_SC3_TILE_COST 16
#endif
    constexpr int64_t K = 48, C = COPER_SC3_TILE_COST;       // units per row, units per tile start
    const int64_t per_tile = rows_per_tile * K + C, n_t = total_rows / rows_per_tile, total_u = n_t * per_tile;
    auto row_at = [&](const int64_t u) -> int64_t {
      const int64_t t = u / per_tile, w = u - t * per_tile;
      int64_t r = w <= C ? 0 : (w - C + K - 1) / K;
      if (r > rows_per_tile) r = rows_per_tile;
      return t * rows_per_tile + r;
    };
    r_begin = row_at(total_u * blockIdx.x / gridDim.x);
    r_end = blockIdx.x + 1 == gridDim.x ? total_rows : row_at(total_u * (blockIdx.x + 1) / gridDim.x);
  }
  if (rows_per_item > 0) {
    const int64_t n_tiles = total_rows / rows_per_tile;
    const int64_t j = blockIdx.x >> 3;
    const int64_t c = (blockIdx.x & 7) + 8 * (j / n_tiles), t = j % n_tiles;
    r_begin = t * rows_per_tile + c * rows_per_item;
    r_end = (c + 1) * rows_per_item < rows_per_tile ? r_begin + rows_per_item : (t + 1) * rows_per_tile;
  }
  if (r_begin >= r_end) return;
#ifdef COPER_DBG_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  ST S;
  S.mask_base = mask;
  S.summ_base = summ;
  S.gm_stride = gm_stride;
  S.sexp = __builtin_amdgcn_readfirstlane(x3s[1]);   // (tband and the block maxima are in the same units)
  int64_t cur_tile = -1;
  int64_t eb_prev = 0;
  bool prev_valid = false;
  const std::make_integer_sequence<int, NS> SSEQ{};
  const std::make_integer_sequence<int, NV> VSEQ{};
  constexpr int MB = SC3_MB;
  constexpr int64_t BLK_REGS = MB * NS * 2;   // f3 registers of one entity block (MB 16-row blocks)
  constexpr int64_t MW = 64 * (MB / 2);       // 16-byte mask pieces of one row of a wave

  // (Round 4 tried carrying (tile, row) along instead of these four 64-bit divisions per row -- some 100 instructions of a wave
  // that issues one every four cycles: the two loop-carried values pushed the allocator into scratch (100 bytes, vmcnt(0) waits
  // inside the loop) and the launch from 0.2775 to 0.3097 ms.  The divisions stay.)
  for (int64_t r = r_begin; r < r_end; ++r) {
    const int64_t tile = r / rows_per_tile;
    const int64_t row = r % rows_per_tile;
    const int64_t eb = (row * 4 + wave) * 2;
    if (tile != cur_tile) {   // workgroup-uniform: (re)start of the pipeline
      __syncthreads();
      const uint4* sh = Hf3 + tile * (TILE_REGS * 64);
      {
        // the query tile into LDS: TILE_REGS / 4 pieces per thread, half of them in flight before the first LDS store
        constexpr int NPC = TILE_REGS * 64 / 256;
        constexpr int HB = (NPC + 1) / 2;
        uint4 th[HB];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
          for (int u = 0; u < HB; ++u) {
            const int jx = (part * HB + u) * 256 + threadIdx.x;
            if (part * HB + u < NPC) th[u] = sh[jx];
          }
#pragma unroll
          for (int u = 0; u < HB; ++u) {
            const int jx = (part * HB + u) * 256 + threadIdx.x;
            if (part * HB + u < NPC) hl3[jx] = th[u];
          }
        }
      }
      cur_tile = tile;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int64_t q = tile * 128 + b * 16 + (lane & 15);
        float2 tb = make_float2(INFINITY, INFINITY);
        if (q < B) tb = tband[q];
        S.tlo[b] = tb.x; S.thi[b] = tb.y;
        S.cg[b] = 0;
      }
#pragma unroll
      for (int i = 0; i < 2 * MB; ++i) { S.mk[i] = 0u; S.mg[i] = 0u; }
      // entity fragments of block 0's first PD steps, pred_bias of both blocks, "previous block" accumulators that count nothing
      sc3_prologue_a<NP, TAIL, PD, GM>(S, Ef3 + eb * BLK_REGS * 64 + lane, std::make_integer_sequence<int, PD>{});
      sc3_load_bias<NP, TAIL, PD, GM, 0>(S, bias_pad, eb, lane);
      sc3_load_bias<NP, TAIL, PD, GM, 1>(S, bias_pad, eb + 1, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int m2 = 0; m2 < MB; ++m2) S.acc[1][m2][b] = f32x4(-INFINITY);
      prev_valid = false;
      __syncthreads();
      // query fragments of the first SC3_LD regions (step 0, column blocks 0 ..)
      S.q0[0] = hl3[((0 * NS + 0) * 2 + 0) * 64 + lane]; S.q1[0] = hl3[((0 * NS + 0) * 2 + 1) * 64 + lane];
      if constexpr (SC3_LD >= 2) { S.q0[1] = hl3[((1 * NS + 0) * 2 + 0) * 64 + lane]; S.q1[1] = hl3[((1 * NS + 0) * 2 + 1) * 64 + lane]; }
      if constexpr (SC3_LD >= 3) { S.q0[2] = hl3[((2 * NS + 0) * 2 + 0) * 64 + lane]; S.q1[2] = hl3[((2 * NS + 0) * 2 + 1) * 64 + lane]; }
#if defined(COPER_DBG_SC3_SKIP_LDS) || defined(COPER_DBG_SC3_SKIP_GL)
      S.q0[1] = S.q0[0]; S.q1[1] = S.q1[0]; S.q0[2] = S.q0[0]; S.q1[2] = S.q1[0]; S.q0[3] = S.q0[0]; S.q1[3] = S.q1[0];
      sc3_prologue_a1<NP, TAIL, PD, GM>(S, std::make_integer_sequence<int, PD>{});
#endif
    }
    const bool has_next = r + 1 < r_end;
    const bool last_of_tile = !has_next || (r + 1) / rows_per_tile != tile;
    const int64_t eb_next = has_next ? (((r + 1) % rows_per_tile) * 4 + wave) * 2 : eb;     // past the end: re-read this row's blocks
    const int64_t gm_col = cur_tile * 128;
    SC3Ptrs X;
    X.ra = sc3_make_rsrc(Ef3 + eb * BLK_REGS * 64);
    X.rn = sc3_make_rsrc(Ef3 + eb_next * BLK_REGS * 64);
    X.voff = lane * 16;
    X.hl = hl3 + lane;
    {
      unsigned hi_off = (unsigned)(uintptr_t)((const uint4 __attribute__((address_space(3)))*)(hl3 + lane + 4096));
      asm volatile("" : "+v"(hi_off));     // opaque: otherwise the second base is re-derived from the first with an add per read
      X.hl_hi = (const uint4 __attribute__((address_space(3)))*)(uintptr_t)hi_off;
    }
    uint4* mask_cur = mask + ((cur_tile * rows_per_tile + row) * 4 + wave) * MW;
    // block 0 (epilogue of the previous row's block 1 beside it: its last value completes that row's mask), then block 1
    sc3_half<NP, TAIL, PD, GM, 0>(S, X, bias_pad, eb_next, lane, prev_valid, GM ? gmax + (eb_prev + 1) * sc3_gm_rows(GM) * gm_stride : nullptr, gm_col,
                                  mask_cur - 4 * MW, SSEQ);
    sc3_half<NP, TAIL, PD, GM, 1>(S, X, bias_pad, eb_next + 1, lane, true, GM ? gmax + eb * sc3_gm_rows(GM) * gm_stride : nullptr, gm_col, mask_cur, SSEQ);
    eb_prev = eb;
    prev_valid = true;
    if (last_of_tile) {
      // drain: block 1's accumulators have no next row of the same tile to hide behind
      sc3_values<NP, TAIL, PD, GM, 1, 0>(S, lane, true, GM ? gmax + (eb + 1) * sc3_gm_rows(GM) * gm_stride : nullptr, gm_col, mask_cur, VSEQ);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int m2 = 0; m2 < MB; ++m2) S.acc[1][m2][b] = f32x4(-INFINITY);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        int g = S.cg[b] + __shfl_xor(S.cg[b], 16);
        g += __shfl_xor(g, 32);
        const int64_t q = cur_tile * 128 + b * 16 + lane;
        if (lane < 16 && q < B && g) atomicAdd(&ng[q], g);
        S.cg[b] = 0;
      }
      prev_valid = false;
    }
  }
#ifdef COPER_DBG_CLOCK
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_sc3_clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_t0;
    g_sc3_clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
}


#ifdef COPER_DBG_CLOCK
extern "C" __attribute__((visibility("default"))) int coper_dbg_clock(int n_wg, double* ghz_median, double* us_median) {
  static unsigned long long hbuf[2 * 1024];
  if (hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_sc3_clk), sizeof hbuf) != hipSuccess) return 1;
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
// the exact band
// ------------------------------------------------------------------------------------------------
// max over the shard of |E_e|_2, |pred_bias[e]| and |E_e[k]| (prepare): consts[0..2] as float bit patterns (non-negative floats
// order like unsigned integers)
__global__ __launch_bounds__(256) void k_band_consts(const float* __restrict__ ent, const float* __restrict__ bias, int64_t n, int d,
                                                     unsigned* __restrict__ consts) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  float emax = 0.f, bmax = 0.f, xmax = 0.f;
  for (int64_t e = wv; e < n; e += nw) {
    float s2 = 0.f;
    for (int k = lane; k < d; k += 64) { const float v = ent[e * d + k]; s2 = fmaf(v, v, s2); xmax = fmaxf(xmax, fabsf(v)); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
    emax = fmaxf(emax, s2);
    if (lane == 0) bmax = fmaxf(bmax, fabsf(bias[e]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o));
  if (lane == 0) {
    const float en = sqrtf(emax) * 1.0000005f;     // rounding of the sum of squares and of the root
    if (en == en) atomicMax(&consts[0], __float_as_uint(en));
    if (bmax == bmax) atomicMax(&consts[1], __float_as_uint(bmax));
    if (xmax == xmax) atomicMax(&consts[2], __float_as_uint(xmax));
  }
}

// tau_q: x3_band_tau (bf16x3_chain.h)
__device__ __forceinline__ float2 band_of(float t, float tau) {
  // outward rounding of t -+ tau is immaterial (tau carries a safety factor); NaN / inf targets: every comparison false
  return make_float2(t - tau, t + tau);
}

// two-call path (coper_rank_counts): tband from the fp32 h rows and the mode's target logits, in the units of the packed batch's
// accumulators (x 2^(e_E + e_h): what the count kernel and the filter correction compare against); 16 lanes per query
__global__ __launch_bounds__(256) void k_band_setup(const float* __restrict__ hvec, const float* __restrict__ tgt, int64_t B, int d, float kappa,
                                                    const unsigned* __restrict__ consts, const int32_t* __restrict__ x3s,
                                                    float2* __restrict__ tband) {
  const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float s2 = 0.f;
  if (q < B)
    for (int k = sub; k < d; k += 16) { const float v = hvec[q * d + k]; s2 = fmaf(v, v, s2); }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
  if (q < B && sub == 0) {
    const int sexp = x3s[1];
    // centred on the EXACT target tgt[B + q] (what the band walk compares against): the target's own x3 error -- its row may
    // live on another shard, with a larger norm or bias than this shard's maxima -- then does not enter at all
    tband[q] = band_of(x3_scale(tgt[B + q], sexp), x3_scale(x3_band_tau(s2, kappa, consts, d, x3s), sexp));
  }
}

int launch_band_consts(coper_handle* h, const float* ent, const float* bias, hipStream_t s) {
  COPER_HIP_TRY(h, hipMemsetAsync(h->band_consts, 0, BAND_NCONST * sizeof(unsigned), s));
  int64_t blocks = (h->dm.n_local + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_band_consts, dim3((unsigned)blocks), dim3(256), 0, s, ent, bias, h->dm.n_local, h->dm.d, h->band_consts);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

float band_kappa(const coper_handle* h) {
  const float k = h->cfg.rank_band_kappa;
  return (k > 0.f ? k : COPER_BAND_KAPPA_DEFAULT) * h->band_kappa_mult;     // (the multiplier: coper_band_policy)
}

int launch_band_setup(coper_handle* h, const float* hvec, const float* tgt, int64_t B, hipStream_t s) {
  hipLaunchKernelGGL(k_band_setup, dim3((unsigned)((B * 16 + 255) / 256)), dim3(256), 0, s, hvec, tgt, B, h->dm.d, band_kappa(h),
                     h->band_consts, h->x3s, (float2*)h->tband_ws);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// exact-chain targets: out[b] = fp32-chain logit of (b, e2[b]) if e2[b] is on this shard else 0
__global__ void k_exact_targets(const float* __restrict__ ent, const float* __restrict__ bias, const float* __restrict__ hvec,
                                const int64_t* __restrict__ e2, int64_t B, int d, int64_t lo, int64_t n_local, float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t row = e2[b] - lo;
  float sx = 0.f, unused = 0.f;
  if (row >= 0 && row < n_local) exact_chain_pair(ent + row * d, nullptr, hvec + b * d, bias[row], 0.f, d, sx, unused);
  out[b] = sx;
}

// the same chain on rows the caller holds (coper_score_rows)
__global__ void k_exact_rows(const float* __restrict__ rows, const float* __restrict__ bias, const float* __restrict__ hvec, int64_t B, int d,
                             float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float sx = 0.f, unused = 0.f;
  exact_chain_pair(rows + b * d, nullptr, hvec + b * d, bias[b], 0.f, d, sx, unused);
  out[b] = sx;
}

int launch_exact_rows(coper_handle* h, const float* hvec, const float* rows, const float* bias, int64_t B, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_exact_rows, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, rows, bias, hvec, B, h->dm.d, out);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_exact_targets(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_exact_targets, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->params["ent_emb"].ptr, h->params["pred_bias"].ptr,
                     hvec, e2, B, h->dm.d, (int64_t)h->cfg.shard_lo, h->dm.n_local, out);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// The exact decision of the band: k_band_exact walks the mask of one count launch (one bit per logit; with the fp16 split a
// query has ~0.3 marked competitors besides its own target) and decides every marked (query, entity) pair with the fp32 chain:
//       known answers of the query and its target: nothing to add (the count kernel counted none of the band, the filter
//         correction subtracted only what lies above t_hi);  otherwise  n_greater += (s > t),  n_equal += (s == t)
//       with s, t from exact_chain_pair on the registered fp32 rows.
// A workgroup takes 4,096 consecutive mask words, all loads in flight at once; the marked pairs (a few dozen) are compacted
// into an LDS list and dealt one per lane: a pair's walk is a handful of dependent round trips to L2 (ids and CSR bounds,
// the row's known answers, the rows in batches of nine k-steps), so the launch takes the time of ONE walk as long as a
// workgroup's pairs fit its lanes.  (Measured on the way here: 64-bit divisions to decode every word -- set or not -- cost
// 20 of 34 us; the bf16 split needed a 16x wider band, 78,000 pairs per pass, 50 us of uncoalesced row reads.)  A list that
// overflows -- heavy ties: every logit of a row inside the band -- is worked off in rounds.  tgt_x == NULL: the exact target
// is computed with the pair from the query's e2 row (unsharded handles); sharded: the all-reduced exact targets of
// coper_target_scores.
struct BandArgs {
  const float* hvec; const float* ent; const float* bias; const int64_t* e2; const int64_t* indptr; const int64_t* idx;
  const float* tgt_x; int32_t* ng; int32_t* ne; int64_t Bc, n_local, shard_lo; int d; int dbg;
  // the audit (below): the mode's own operands of the pairs the walk decides
  const uint4* Ehi; const uint4* Elo; const float* bias_pad; const float2* tband; const int32_t* x3s; unsigned* consts; int KS16; int audit;
  int audit_wg_mask;     // workgroups with (index & mask) == 0 audit: at most ~512 per launch (a 10M-entity launch has 20,000)
};

// returns true when the pair was decided by the chain; sx_out / tx_out: the chain's logits of the competitor and of the target
template <int CB = COPER_CHAIN_CB>
__device__ __forceinline__ bool band_decide(const BandArgs& A, const int64_t q, const int64_t e, float& sx_out, float& tx_out) {
  if (q >= A.Bc || e >= A.n_local || A.dbg == 1) return false;
  const int64_t eg = e + A.shard_lo, tq = A.e2[q];
  const int64_t lo0 = A.indptr[q], hi0 = A.indptr[q + 1];
  if (eg == tq) return false;                        // the target itself (metrics.py:46)
  // known answer?  (ids sorted ascending inside a row)  Short rows -- nearly all -- in one round trip
  bool known = false;
  if (hi0 - lo0 <= 8) {
    int64_t f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) f[u] = lo0 + u < hi0 ? A.idx[lo0 + u] : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u) known |= f[u] == eg;
  } else {
    int64_t lo_i = lo0, hi_i = hi0;
    while (lo_i < hi_i) {
      const int64_t mid = (lo_i + hi_i) >> 1;
      const int64_t f = A.idx[mid];
      if (f == eg) { known = true; break; }
      if (f < eg) lo_i = mid + 1; else hi_i = mid;
    }
  }
  if (known || A.dbg == 2) return false;
  const float* hr = A.hvec + q * A.d;
  float sx, tx = 0.f;
  if (A.tgt_x) {
    tx = A.tgt_x[q];
    exact_chain_pair<CB>(A.ent + e * A.d, nullptr, hr, A.bias[e], 0.f, A.d, sx, tx);
  } else {
    const int64_t trow = tq - A.shard_lo;
    if (trow >= 0 && trow < A.n_local) exact_chain_pair<CB>(A.ent + e * A.d, A.ent + trow * A.d, hr, A.bias[e], A.bias[trow], A.d, sx, tx);
    else exact_chain_pair<CB>(A.ent + e * A.d, nullptr, hr, A.bias[e], 0.f, A.d, sx, tx);
  }
  if (sx > tx) atomicAdd(&A.ng[q], 1);
  else if (A.ne && sx == tx) atomicAdd(&A.ne[q], 1);
  sx_out = sx;
  tx_out = tx;
  return true;
}

// The audit of the band: kappa, the relative half-width, is empirical -- the proven bound of the split and
// of fp32 accumulation in any order is 75x wider -- so every count launch CHECKS it on the pairs its band walk decides anyway
// (the competitors closest to the target: a few per query, ~26,000 per FB15k-237-shaped pass): a wave takes 32 decided pairs,
// scores them once more with the mode's own sequence (their entity rows from the row-major planes, their query fragments
// rebuilt from the fp32 rows: the bits the count kernel compared) and folds
//        |s_x3 - s_chain| / (tau_q / 2)        (tau_q / 2 = the error the band allows ONE logit)
// and the same for the x3 target the band is centred on into band_consts[3] (a maximum of non-negative floats) and the number of
// audited pairs into band_consts[4].  coper_band_audit reads them: a ratio that approaches 1 means the mode's ranks may no
// longer be the chain's (tests assert <= 0.5 on every configuration and at every operand scale; the drop-in ranker warns).
__device__ __forceinline__ void band_audit_group(const BandArgs& A, const unsigned long long* __restrict__ s_p, const float* __restrict__ s_sx,
                                                 const float* __restrict__ s_tx, const int base, const int n, int64_t* __restrict__ s_e) {
  const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
  const int pi = base + i;
  const bool have = pi < n;
  const float sx = have ? s_sx[pi] : NAN, tx = have ? s_tx[pi] : 0.f;
  const int64_t q = have ? (int64_t)(s_p[pi] >> 32) : 0, e = have ? (int64_t)(s_p[pi] & 0xFFFFFFFFull) : 0;
  const bool live = have && sx == sx;
  const int eh = A.x3s[0], sexp = A.x3s[1];
  const int KS = A.KS16, d = A.d;
  __builtin_amdgcn_wave_barrier();
  if (half == 0) s_e[i] = live ? e : -1;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t er = s_e[(r & 3) + 8 * (r >> 2) + 4 * half];
    acc[r] = er >= 0 ? x3_scale(A.bias_pad[er], sexp) : 0.f;
  }
  const uint4* pa_h = A.Ehi + (live ? e : 0) * (2 * KS) + half;
  const uint4* pa_l = A.Elo + (live ? e : 0) * (2 * KS) + half;
  const float* hr = A.hvec + q * d;
  const bool vec_ok = (d & 7) == 0 && (((uintptr_t)A.hvec) & 15) == 0;
  // a wave walks its 32 pairs alone: the loads of CB k-steps are issued before their instructions (a round trip per k-step
  // otherwise: the audit took 40 us of a 15 us launch)
  constexpr int CB = 4;
  for (int ks = 0; ks < KS; ks += CB) {
    uint4 ah[CB], al[CB];
    float4 y0[CB], y1[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int k = ks + u < KS ? ks + u : KS - 1;
      ah[u] = pa_h[k * 2]; al[u] = pa_l[k * 2];
      const int kk = 16 * k + 8 * half;
      if (vec_ok && kk + 8 <= d) {
        y0[u] = live ? *(const float4*)(hr + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        y1[u] = live ? *(const float4*)(hr + kk + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) t[c] = (live && kk + c < d) ? hr[kk + c] : 0.f;
        y0[u] = make_float4(t[0], t[1], t[2], t[3]);
        y1[u] = make_float4(t[4], t[5], t[6], t[7]);
      }
    }
    uint4 bh[CB], bl[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const float y[8] = {x3_scale(y0[u].x, eh), x3_scale(y0[u].y, eh), x3_scale(y0[u].z, eh), x3_scale(y0[u].w, eh),
                          x3_scale(y1[u].x, eh), x3_scale(y1[u].y, eh), x3_scale(y1[u].z, eh), x3_scale(y1[u].w, eh)};
      split8_s16(y, bh[u], bl[u]);
    }
#pragma unroll
    for (int u = 0; u < CB; u += 2) {   // wave-uniform
      if (ks + u + 1 < KS) { BX3_PAIR(ah[u], al[u], bh[u], bl[u], ah[u + 1], al[u + 1], bh[u + 1], bl[u + 1], acc); }
      else if (ks + u < KS) { BX3_LAST(ah[u], al[u], bh[u], bl[u], acc); }
    }
  }
  const bool diag_lane = ((i >> 2) & 1) == half;
  const int reg = (i & 3) + 4 * (i >> 3);
  float sc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sc = (r == reg) ? acc[r] : sc;
  float ratio = 0.f;
  if (diag_lane && live) {
    const float2 tb = A.tband[q];
    const float allow = 0.25f * (tb.y - tb.x);           // tau / 2, in the accumulators' units
    if (allow > 0.f) {
      ratio = fabsf(sc - x3_scale(sx, sexp)) / allow;
      ratio = fmaxf(ratio, fabsf(0.5f * (tb.x + tb.y) - x3_scale(tx, sexp)) / allow);
      if (!(ratio == ratio)) ratio = 0.f;                // (inf - inf of a padded row: nothing to learn)
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, o));
  const int cnt = __builtin_popcountll(__ballot(diag_lane && live));
  if (lane == 0 && cnt) {
    atomicMax(&A.consts[3], __float_as_uint(ratio));
    atomicAdd(&A.consts[4], (unsigned)cnt);
  }
}

#ifndef COPER_BE_UPW
#define COPER_BE_UPW 64
#endif
// units (rows of waves) per workgroup: a pair's walk is latency, so the launch is as long as the longest chain of walks one
// thread makes; at FB15k-237 shapes 256 units gave a workgroup ~350 pairs for its 256 threads (two walks for many), 64
// units give ~90 (one walk) and 290 workgroups instead of 73
constexpr int BE_CAP = 4096, BE_ITEMS = 2048, BE_UPW = COPER_BE_UPW;
constexpr int BE_AUDIT = 128;     // pairs of a round the audit re-scores, 32 per wave (a workgroup walks ~90 at FB15k-237 shapes: all of them)
template <int CB = COPER_CHAIN_CB>
__device__ __forceinline__ void band_exact_body(const uint4* __restrict__ mask, const unsigned long long* __restrict__ summ, const int64_t n_units,
                                                const unsigned rows4 /* rows per tile x 4 waves */, const BandArgs& A, const int64_t wg) {
  constexpr int MB = SC3_MB, NW = 2 * MB;      // 32-bit mask words per lane and row of a wave
  __shared__ unsigned long long s_p[BE_CAP];   // marked pairs: (query << 32) | entity
  __shared__ float s_sx[BE_AUDIT], s_tx[BE_AUDIT];   // the chain's logits of the first pairs of a round (NaN: not decided): audited
  __shared__ int64_t s_ae[4][32];
  __shared__ unsigned s_it[BE_ITEMS];          // (unit in the workgroup << 6) | lane: the mask words to fetch
  __shared__ int s_n, s_ni;
  // three phases, each spread over all threads (a thread that walks its unit's lanes one after the other pays a round trip
  // per lane): a thread per unit (one wave's row of 32 MB entities x 128 queries) reads the summary and lists the lanes
  // that wrote mask words; a thread per listed lane fetches its words and lists the marked pairs; a thread per pair walks.
  const int64_t unit0 = wg * BE_UPW;
  if (threadIdx.x == 0) { s_n = 0; s_ni = 0; }
  __syncthreads();
  {
    const int64_t unit = unit0 + threadIdx.x;
    unsigned long long lanes = (threadIdx.x < BE_UPW && unit < n_units) ? summ[unit] : 0ull;
    while (lanes) {
      const int l = __builtin_ctzll(lanes);
      lanes &= lanes - 1;
      const int slot = atomicAdd(&s_ni, 1);
      if (slot < BE_ITEMS) s_it[slot] = ((unsigned)threadIdx.x << 6) | (unsigned)l;
    }
  }
  __syncthreads();
  bool audited = false;      // (workgroup-uniform)
  const int ni = s_ni;       // beyond BE_ITEMS (a workgroup's 256 x 64 lanes nearly all marked: heavy ties): the slow loop below
  for (int it0 = 0; it0 < (ni < BE_ITEMS ? ni : BE_ITEMS); it0 += 256) {
    const int it = it0 + threadIdx.x;
    unsigned wc[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) wc[c] = 0u;
    unsigned tile = 0, eb = 0;
    int l = 0;
    if (it < ni && it < BE_ITEMS) {
      const int64_t unit = unit0 + (s_it[it] >> 6);
      l = (int)(s_it[it] & 63u);
      tile = (unsigned)unit / rows4;
      eb = ((unsigned)unit - tile * rows4) * 2;       // (row * 4 + wave) * 2: the wave's first entity block (16 MB rows each)
#pragma unroll
      for (int i = 0; i < MB / 2; ++i) {
        const uint4 w = mask[(unit * 64 + l) * (MB / 2) + i];
        wc[4 * i] = w.x; wc[4 * i + 1] = w.y; wc[4 * i + 2] = w.z; wc[4 * i + 3] = w.w;
      }
    }
    while (true) {      // rounds: a pair list that fills up is worked off and the words are gone on with
      bool left = false;
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        while (wc[c] && !left) {
          const int p = 31 - __builtin_clz(wc[c]);     // highest set bit first = lowest value index first
          const int slot = atomicAdd(&s_n, 1);
          if (slot >= BE_CAP) { left = true; break; }
          wc[c] &= ~(1u << p);
          const int V = 32 * (c % MB) + (31 - p);
          const int b = V / (4 * MB), m2 = (V >> 2) % MB, j = V & 3;
          const unsigned long long e = (unsigned long long)(eb + c / MB) * (16 * MB) + 16 * m2 + 4 * (l >> 4) + j;
          const unsigned long long q = (unsigned long long)tile * 128 + 16 * b + (l & 15);
          s_p[slot] = (q << 32) | e;
        }
      }
      const int full = __syncthreads_or(left ? 1 : 0);
      const bool last = it0 + 256 >= (ni < BE_ITEMS ? ni : BE_ITEMS);
      if (full || last) {
        const int n = s_n < BE_CAP ? s_n : BE_CAP;
        const bool aud_round = A.audit && !audited && (wg & (int64_t)A.audit_wg_mask) == 0;
        for (int p = threadIdx.x; p < n; p += 256) {
          float sx = NAN, tx = 0.f;
          const int64_t pq = (int64_t)(s_p[p] >> 32), pe = (int64_t)(s_p[p] & 0xFFFFFFFFull);
          bool dec = band_decide<CB>(A, pq, pe, sx, tx);
          // Round 5: a query's own TARGET is always inside its band (it is the band's centre), so its pair is always listed --
          // and audited like a decided one: the chain's logit of the target against the mode's.  A band too narrow to hold any
          // competitor (nothing decided, nothing to audit: the audit was blind exactly when it mattered) still reports the error
          // of every target the sampled workgroups see.  Counts are not touched.
          if (!dec && aud_round && p < BE_AUDIT && pq < A.Bc && pe < A.n_local && A.dbg == 0 && pe + A.shard_lo == A.e2[pq]) {
            if (A.tgt_x) tx = A.tgt_x[pq];
            else { float unused = 0.f; exact_chain_pair(A.ent + pe * A.d, nullptr, A.hvec + pq * A.d, A.bias[pe], 0.f, A.d, tx, unused); }
            sx = tx;
            dec = true;
          }
          if (p < BE_AUDIT) { s_sx[p] = dec ? sx : NAN; s_tx[p] = tx; }
        }
        __syncthreads();
        if (aud_round && n > 0) {   // the first round of (a sample of) the workgroups
          audited = true;
          const int na = n < BE_AUDIT ? n : BE_AUDIT;
          for (int g = (int)(threadIdx.x >> 6); g * 32 < na; g += 4) band_audit_group(A, s_p, s_sx, s_tx, g * 32, na, s_ae[threadIdx.x >> 6]);
          __syncthreads();
        }
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
      }
      if (!full) break;
    }
  }
  if (ni > BE_ITEMS) {
    // the lanes that did not fit the item list: every thread walks its own unit again and skips the lanes that were listed
    // (the list took them in no particular order, so the listed ones are looked up) -- correctness path, not a fast one
    const int64_t unit = unit0 + threadIdx.x;
    unsigned long long lanes = (threadIdx.x < BE_UPW && unit < n_units) ? summ[unit] : 0ull;
    for (int i = 0; i < BE_ITEMS; ++i)
      if ((s_it[i] >> 6) == (unsigned)threadIdx.x) lanes &= ~(1ull << (s_it[i] & 63u));
    const unsigned tile = (unsigned)unit / rows4, eb = ((unsigned)unit - tile * rows4) * 2;
    while (lanes) {
      const int l = __builtin_ctzll(lanes);
      lanes &= lanes - 1;
      for (int i = 0; i < MB / 2; ++i) {
        const uint4 w = mask[(unit * 64 + l) * (MB / 2) + i];
        const unsigned ww[4] = {w.x, w.y, w.z, w.w};
        for (int cc = 0; cc < 4; ++cc) {
          unsigned bits = ww[cc];
          const int c = 4 * i + cc;
          while (bits) {
            const int p = 31 - __builtin_clz(bits);
            bits &= ~(1u << p);
            const int V = 32 * (c % MB) + (31 - p);
            const int b = V / (4 * MB), m2 = (V >> 2) % MB, j = V & 3;
            float sx_u, tx_u;
            band_decide<CB>(A, (int64_t)tile * 128 + 16 * b + (l & 15), (int64_t)(eb + c / MB) * (16 * MB) + 16 * m2 + 4 * (l >> 4) + j, sx_u, tx_u);
          }
        }
      }
    }
  }
}

template <int CB>
__global__ __launch_bounds__(256) void k_band_exact(const uint4* __restrict__ mask, const unsigned long long* __restrict__ summ, int64_t n_units,
                                                    unsigned rows4, BandArgs A) {
  band_exact_body<CB>(mask, summ, n_units, rows4, A, blockIdx.x);
}

// k_filter_excess_bf16x3 -- the CSR entries beyond the first TL_OWN_ENTRIES of a 32-query block (real KGs hold (e1, rel) pairs
// with thousands of known tails; inside the workgroups that own the block they would run ~7 us per 128 entries while the rest
// of the chip waits).  The listed blocks' remaining tiles are dealt over the waves of FX_GRID workgroups: a wave rebuilds the
// block's query fragments from the fp32 rows (same values, same split), scores a tile exactly as the owners do and takes the
// known answers above the band back from `ranks`.  No listed block (every pass of the benchmark shapes): the waves read one
// word and leave.  The last workgroup to finish empties the list for the next pass.
template <int KS>
__device__ __forceinline__ void filter_excess_body(const FilterArgs& F, const int wg, const int n_wg, int64_t (*s_e_all)[32]) {
  const int n = F.heavy[0];
  if (n == 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
  int64_t* s_e = s_e_all[wave];
  const int G = n_wg * FX_WAVES;
  const int64_t me = (int64_t)wg * FX_WAVES + wave;
  for (int hb = 0; hb < n; ++hb) {
    const int64_t blk = F.heavy[2 + hb], q0 = blk * 32, q = q0 + i;
    const bool live = q < F.B;
    const int64_t qe = q0 + 32 < F.B ? q0 + 32 : F.B;
    const int64_t p_end = F.indptr[qe], p0 = F.indptr[q0] + TL_OWN_ENTRIES;
    // tile t of listed block hb belongs to wave (t + 61 hb) mod G: consecutive blocks start on different waves
    const int64_t t0 = ((me - 61 * (int64_t)hb) % G + G) % G;
    if (p0 + 32 * t0 >= p_end) continue;
    const int64_t my_lo = live ? F.indptr[q] : p_end;
    const int64_t my_e2 = live ? F.e2[q] : -1;
    const float t_hi = live ? F.tband[q].y : 0.f;
    uint4 bh[KS], bl[KS];
    const int eh = F.x3s[0], sexp = F.x3s[1];
    tail_fragments_from_rows<KS>(F.hvec, q, live, F.d, half, eh, bh, bl);
    for (int64_t pb = p0 + 32 * t0; pb < p_end; pb += 32 * (int64_t)G) {
      float sc;
      int qi;
      const int64_t frow = tail_filter_tile<KS>(pb, p_end, my_lo, my_e2, F.idx, F.n_local, s_e, F.Ehi, F.Elo, F.bias_pad, bh, bl, i, half, sexp, sc, qi);
      const float tq = __shfl(t_hi, qi);
      tail_take_back(half == 0 && frow >= 0 && sc > tq, qi, i, half, q0, F.ranks);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&F.heavy[1], 1) == n_wg - 1) {   // every workgroup of the role has read the list
      F.heavy[0] = 0;
      F.heavy[1] = 0;
    }
  }
}


template <int KS>
__global__ __launch_bounds__(64 * FX_WAVES) void k_filter_excess_bf16x3(FilterArgs F) {
  __shared__ int64_t s_e_all[FX_WAVES][32];
  filter_excess_body<KS>(F, (int)blockIdx.x, (int)gridDim.x, s_e_all);
}

// the band walk of a count launch with the excess role in the same launch (coper_encode_rank: the tail kernel listed the blocks
// before the count launch; as its own launch the role cost 4 - 5 us per pass to find an empty list)
template <int KS>
__global__ __launch_bounds__(256) void k_band_excess_bf16x3(const uint4* __restrict__ mask, const unsigned long long* __restrict__ summ,
                                                            int64_t n_units, unsigned rows4, BandArgs A, int n_band, FilterArgs F) {
  static_assert(FX_WAVES == 4, "the roles share a launch of 256 threads");
  __shared__ int64_t s_e_all[FX_WAVES][32];
  if ((int)blockIdx.x < n_band) band_exact_body(mask, summ, n_units, rows4, A, blockIdx.x);
  else filter_excess_body<KS>(F, (int)blockIdx.x - n_band, (int)gridDim.x - n_band, s_e_all);
}

static FilterArgs filter_args(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* indptr, const int64_t* idx,
                              const float2* tband, int64_t B, int32_t* ranks) {
  FilterArgs F;
  F.hvec = hvec; F.Ehi = (const uint4*)h->Erm16_hi; F.Elo = (const uint4*)h->Erm16_lo; F.bias_pad = h->bias_pad; F.e2 = e2;
  F.indptr = indptr; F.idx = idx; F.tband = tband; F.ranks = ranks; F.heavy = h->heavy_ws; F.x3s = h->x3s; F.B = B; F.n_local = h->dm.n_local;
  F.d = h->dm.d;
  return F;
}

// after a launch whose workgroups listed blocks beyond their own share (the tail kernel's filter phase).  defer: the role
// joins the band launch of the count pass that follows on the same stream (score_count3_chunk_bf16x3) instead of its own
int launch_filter_excess_bf16x3(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* indptr, const int64_t* idx,
                                int64_t nnz, int64_t B, int32_t* ranks, bool defer, hipStream_t s) {
  h->excess_pending = false;
  if (nnz <= TL_OWN_ENTRIES) return COPER_OK;     // no block can exceed its own share
  if (h->dm.KS16 != 13 && h->dm.KS16 != 16) return fail(h, COPER_EUNSUPPORTED, "filter excess: ent_emb_size not served by the fused tail");
  const FilterArgs F = filter_args(h, hvec, e2, indptr, idx, (const float2*)h->tband_ws, B, ranks);
#ifndef COPER_DBG_SC3_NO_BAND
  if (defer) {
    static_assert(sizeof(FilterArgs) <= sizeof(h->excess_args), "coper_internal.h: excess_args too small");
    memcpy(h->excess_args, &F, sizeof F);
    h->excess_pending = true;
    return COPER_OK;
  }
#endif
  if (h->dm.KS16 == 13) hipLaunchKernelGGL(k_filter_excess_bf16x3<13>, dim3(FX_GRID), dim3(64 * FX_WAVES), 0, s, F);
  else hipLaunchKernelGGL(k_filter_excess_bf16x3<16>, dim3(FX_GRID), dim3(64 * FX_WAVES), 0, s, F);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int NP, int TAIL, int GM>
static int sc3_go(coper_handle* h, int64_t q0, int64_t Bc, int32_t* ng, float* gmax, int64_t gm_stride, hipStream_t s) {
  constexpr int NS = NP + TAIL;
#ifdef COPER_SC3_PD
  constexpr int PD = COPER_SC3_PD < NS ? COPER_SC3_PD : NS;
#else
  constexpr int PDW = SC3_MB == 4 ? 2 : 3;  // steps ahead: 2 x 12 x 8 (MB = 4) or 3 x 6 x 8 (MB = 2) instructions of 16 cycles
  constexpr int PD = PDW < NS ? PDW : NS;
#endif
  const Dims& dm = h->dm;
  const int64_t q_tiles = (Bc + 127) / 128;
  const int64_t rows_per_tile = dm.n_eblk * 2 / SC3_MB / 8;    // a row: 4 waves x 2 entity blocks of 16 MB rows
  const int64_t total_rows = q_tiles * rows_per_tile;
  int64_t grid = h->num_cus;
  if (grid > total_rows) grid = total_rows;
  // items (the kernel's header): when several query tiles stream an entity table that L2 + MALL cannot hold
  static const int64_t item_rows = getenv("COPER_SC3_ITEM_ROWS") ? atoll(getenv("COPER_SC3_ITEM_ROWS")) : 32;
  const bool big = (size_t)dm.n_local * dm.d * 4 > ((size_t)128 << 20);
  int64_t rows_per_item = (big && q_tiles > 1 && item_rows > 0 && rows_per_tile >= 16 * item_rows) ? item_rows : 0;
  if (rows_per_item && !getenv("COPER_SC3_ITEM_ROWS")) {
    // Round 5: the chunks of a tile are dealt to the eight XCDs round-robin, so the busiest XCD works through ceil(chunks / 8) of
    // them -- at an entity SHARD's size (1.25 M rows of the 10M table: 2,441 rows per tile, 77 chunks of 32) that is 10 against an
    // average of 9.6: 4 % of the launch (the shard-shaped count ran at 0.175 of the roof against 0.204 for the whole table,
    // VERDICT r4 weak 9).  The item length is chosen among 24 .. 40 rows so that the chunk count sits just under a multiple of
    // eight (a tile start costs about a third of a row: part of the price of shorter items).
    double best = 1e30;
    for (int64_t ri = 24; ri <= 40; ++ri) {
      const int64_t chunks = (rows_per_tile + ri - 1) / ri, per_xcd = (chunks + 7) / 8;
      const double cost = (double)per_xcd * ((double)ri + 0.35);
      if (cost < best) { best = cost; rows_per_item = ri; }
    }
  }
  if (rows_per_item) grid = ((rows_per_tile + rows_per_item - 1) / rows_per_item + 7) / 8 * 8 * q_tiles;
  const size_t lds = (size_t)8 * NS * 2 * 64 * sizeof(uint4);
  const uint4* hf3 = (const uint4*)h->hf3_ws + (q0 / 16) * NS * 2 * 64;
  static bool attr_done[16] = {};
  const int dev = h->cfg.device & 15;
  if (!attr_done[dev]) {
    COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_score_count3_bf16x3<NP, TAIL, PD, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((k_score_count3_bf16x3<NP, TAIL, PD, GM>), dim3((unsigned)grid), dim3(256), lds, s, (const uint4*)h->Ef3, h->bias_pad, hf3,
                     (const float2*)h->tband_ws + q0, Bc, rows_per_tile, total_rows, ng + q0, (uint4*)h->mask_ws,
                     (unsigned long long*)((char*)h->mask_ws + score_count3_mask_words_bytes(h, Bc)), gmax, gm_stride, rows_per_item,
                     h->x3s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

template <int NP, int TAIL>
static int sc3_gm(coper_handle* h, int64_t q0, int64_t Bc, int32_t* ng, float* gmax, int64_t gm_stride, hipStream_t s) {
  if (!gmax) return sc3_go<NP, TAIL, 0>(h, q0, Bc, ng, gmax, gm_stride, s);
  if constexpr (SC3_MB == 4 && (2 * NP + TAIL == 13 || 2 * NP + TAIL == 16)) {     // (64-entity maxima: topk_expand, coper_internal.h)
    if (topk_expand(h) == 2) return sc3_go<NP, TAIL, 2>(h, q0, Bc, ng, gmax, gm_stride, s);
  }
  return sc3_go<NP, TAIL, 1>(h, q0, Bc, ng, gmax, gm_stride, s);
}

// bytes of the band mask of a count launch over Bc queries: the words (16 bytes per lane and row of a wave), then the
// summaries (8 bytes per row of a wave)
static int64_t sc3_units(const coper_handle* h, int64_t Bc) { return ((Bc + 127) / 128) * (h->dm.n_eblk * 2 / SC3_MB / 8) * 4; }   // rows of waves
size_t score_count3_mask_words_bytes(const coper_handle* h, int64_t Bc) {
  return (size_t)sc3_units(h, Bc) * 64 * (SC3_MB / 2) * sizeof(uint4);     // one bit per logit
}
size_t score_count3_mask_bytes(const coper_handle* h, int64_t Bc) {
  return score_count3_mask_words_bytes(h, Bc) + (size_t)sc3_units(h, Bc) * sizeof(unsigned long long);
}

// Count launch over queries [q0, q0 + Bc) (q0 a multiple of 128) of the packed batch + the exact decision of its band.
// tband_ws / hf3_ws hold the whole batch; hvec (fp32 rows of the whole batch), tgt_x (exact targets of the whole batch or NULL).
int score_count3_chunk_bf16x3(coper_handle* h, int64_t q0, int64_t Bc, const float* hvec, const float* tgt_x, const int64_t* e2,
                              const int64_t* indptr, const int64_t* idx, int32_t* ng, int32_t* ne, float* gmax, int64_t gm_stride,
                              hipStream_t s) {
  if (score_count3_mask_bytes(h, Bc) > h->mask_cap) return fail(h, COPER_ESTATE, "score_count3: band mask workspace not reserved");
  int rc;
  {
    ScopedKernelTimer t(h, "score_count", s);
    switch (h->dm.KS16) {
#define SC3_CASE(KS_) case KS_: rc = sc3_gm<(KS_) / 2, (KS_) & 1>(h, q0, Bc, ng, gmax, gm_stride, s); break;
      SC3_CASE(1) SC3_CASE(2) SC3_CASE(3) SC3_CASE(4) SC3_CASE(5) SC3_CASE(6) SC3_CASE(7) SC3_CASE(8) SC3_CASE(9) SC3_CASE(10)
      SC3_CASE(11) SC3_CASE(12) SC3_CASE(13) SC3_CASE(14) SC3_CASE(15) SC3_CASE(16) SC3_CASE(17) SC3_CASE(18) SC3_CASE(19) SC3_CASE(20)
#undef SC3_CASE
      default: rc = fail(h, COPER_EUNSUPPORTED, "score_count3: ent_emb_size beyond 320");
    }
  }
  if (rc) return rc;
  COPER_DBG_SYNC(h, s, "score_count3");
#ifndef COPER_DBG_SC3_NO_BAND
  {
    ScopedKernelTimer t(h, "band_exact", s);
    const int64_t rows_per_tile = h->dm.n_eblk * 2 / SC3_MB / 8;
    BandArgs A;
    A.hvec = hvec + q0 * h->dm.d; A.ent = h->params["ent_emb"].ptr; A.bias = h->params["pred_bias"].ptr;
    A.e2 = e2 + q0; A.indptr = indptr + q0; A.idx = idx; A.tgt_x = tgt_x ? tgt_x + q0 : nullptr;
    A.ng = ng + q0; A.ne = ne ? ne + q0 : nullptr; A.Bc = Bc; A.n_local = h->dm.n_local; A.shard_lo = (int64_t)h->cfg.shard_lo; A.d = h->dm.d;
    { static const int dbg = getenv("COPER_DBG_BAND") ? atoi(getenv("COPER_DBG_BAND")) : 0; A.dbg = dbg; }
    A.Ehi = (const uint4*)h->Erm16_hi; A.Elo = (const uint4*)h->Erm16_lo; A.bias_pad = h->bias_pad; A.tband = (const float2*)h->tband_ws + q0;
    A.x3s = h->x3s; A.consts = h->band_consts; A.KS16 = h->dm.KS16;
    {
      // which launches are audited: a wave re-scores 32 pairs in ~25 us of gathers -- beside a 15 us walk that is too much to pay
      // on every 0.5 ms pass, nothing beside a launch of milliseconds.  Default: the first count launch after prepare and
      // every 8th from there (+3 us per pass), every launch of more than 2^31 logits; coper_config.band_audit_period
      // overrides (1: every launch, what the tests run; negative: never)
      static const int off = getenv("COPER_BAND_NO_AUDIT") != nullptr;      // (A/B timing switch)
      int period = h->cfg.band_audit_period;
      if (period == 0) period = (double)Bc * (double)h->dm.n_local >= 2147483648.0 ? 1 : 8;
      // a launch recorded into a hipGraph is replayed as recorded: it carries the audit only when every launch does
      // (period 1), never by the accident of where the counter stood at capture
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (period != 1 && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) period = -1;
      A.audit = (!off && period > 0 && (h->band_launches++ % (unsigned)period) == 0u) ? 1 : 0;
      int mask = 0;
      while ((sc3_units(h, Bc) + BE_UPW - 1) / BE_UPW / (mask + 1) > 512) mask = 2 * mask + 1;
      A.audit_wg_mask = mask;
    }
    const int64_t n_units = sc3_units(h, Bc);
    if (rows_per_tile * 4 > 0x7fffffffLL || n_units > 0x7fffffffLL) return fail(h, COPER_EUNSUPPORTED, "band mask beyond 2^31 units");
    const unsigned n_band = (unsigned)((n_units + BE_UPW - 1) / BE_UPW);
    const unsigned long long* summ = (const unsigned long long*)((const char*)h->mask_ws + score_count3_mask_words_bytes(h, Bc));
    if (h->excess_pending) {
      FilterArgs F;
      memcpy(&F, h->excess_args, sizeof F);
      h->excess_pending = false;
      if (h->dm.KS16 == 13)
        hipLaunchKernelGGL(k_band_excess_bf16x3<13>, dim3(n_band + FX_GRID), dim3(256), 0, s, (const uint4*)h->mask_ws, summ, n_units,
                           (unsigned)(rows_per_tile * 4), A, (int)n_band, F);
      else
        hipLaunchKernelGGL(k_band_excess_bf16x3<16>, dim3(n_band + FX_GRID), dim3(256), 0, s, (const uint4*)h->mask_ws, summ, n_units,
                           (unsigned)(rows_per_tile * 4), A, (int)n_band, F);
    } else {
#ifndef COPER_BAND_CB_LARGE
#define COPER_BAND_CB_LARGE 3
#endif
      // large tables / shards: hundreds of thousands of pairs, one per lane -- three k-steps per round trip keep more walks resident
      // (bf16x3_chain.h: exact_chain_pair); the same chain, the same decisions
      if (h->dm.n_local >= 500000)
        hipLaunchKernelGGL(k_band_exact<COPER_BAND_CB_LARGE>, dim3(n_band), dim3(256), 0, s, (const uint4*)h->mask_ws, summ, n_units, (unsigned)(rows_per_tile * 4), A);
      else
        hipLaunchKernelGGL(k_band_exact<COPER_CHAIN_CB>, dim3(n_band), dim3(256), 0, s, (const uint4*)h->mask_ws, summ, n_units, (unsigned)(rows_per_tile * 4), A);
    }
    COPER_HIP_TRY(h, hipGetLastError());
  }
  COPER_DBG_SYNC(h, s, "band_exact");
#endif
  return COPER_OK;
}

}  // namespace coper
