// The 16-bit split of the "x3" arithmetic (COPER_SCORE_BF16X3): every fp32 operand x of a matrix product becomes two
// 16-bit terms  x = hi + lo  (hi = rne16(x), lo = rne16(x - hi))  and a product is  lo*hi + hi*lo + hi*hi  on the 16-bit
// matrix cores with fp32 accumulation.  The bf16 and the fp16 MFMAs run at the same rate (same cycles per instruction:
// MI355X_MICROARCH.md, Matrix cores); what differs is where the 16 bits go:
//   bf16 (8-bit exponent, 8-bit significand):  hi + lo carries 16-17 significant bits of x (error <= 2^-16 |x|), any fp32
//        magnitude;
//   fp16 (5-bit exponent, 11-bit significand): hi + lo carries 22-23 significant bits (error <= 2^-22 |x|) while lo is a
//        NORMAL fp16, i.e. 2^-3 <= |x| < 65,504; below that lo is subnormal and the error is 2^-25 ABSOLUTE (gfx950's fp16
//        MFMA keeps subnormal inputs: measured, tools/microbench/mfma_shape.hip).
// Round 3 measured what the 8 missing bits cost (tools/rank_decomp.py): the split is fp16 (round 2's bf16 arithmetic was a
// build option until round 6: DESIGN_LOG.md).
//
// Round 4: the split is SCALE-INVARIANT.  fp16's 22 bits only exist in a window of magnitudes, and nothing about the model
// puts its operands there: the reference initialises ent_emb with xavier_initializer (models.py:205-208: +-0.02 for
// FB15k-237, +-7.7e-4 for a 10M-entity table), where EVERY element sat in the absolute-error regime.  So every operand class
// is multiplied by an exact power of two before the split, chosen from its largest magnitude so that  max |x| 2^e  lies in
// [2^14, 2^15)  (x3_exp_for below: one binade of headroom under fp16's 65,504; the clamp can no longer trigger on finite
// data), and the inverse power of two is folded into everything that leaves the arithmetic:
//   entity table      one exponent per handle (prepare; coper_config.x3_ent_absmax lets the shards of one table agree)
//   queries h         one exponent per packed batch, from the batch's largest |h| element (k_absmax_publish / the finalize)
//   dense weights W_r one exponent per relation (prepare);  conv activations x: one exponent per handle from a bound (prepare)
// A power-of-two factor commutes with every fp32 rounding (no overflow / underflow: exponents are clamped to +-X3_EXP_CLAMP),
// so the fp32 chain, the f32 mode and the oracle do not change, and an x3 accumulator is 2^(e_E + e_h) times the logit the
// unscaled arithmetic would have produced with all 22 bits.  Elements more than 2^17 below the largest of their class still
// fall into the absolute regime -- 2^-25 2^-e, i.e. 2^-39 of the class maximum: the exact band carries that term (band_tau).
#pragma once
#include <hip/hip_runtime.h>

namespace coper {

typedef _Float16 s16_t;
#define S16_NAME "fp16"
#define S16_MFMA32_BUILTIN __builtin_amdgcn_mfma_f32_32x32x16_f16
#define S16_MFMA16_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_f16
__device__ __forceinline__ float s16_clamp(float v) { return __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f); }   // (NaN stays NaN)
typedef s16_t s16x8 __attribute__((ext_vector_type(8)));

// exponent e with  maxabs * 2^e  in [2^14, 2^15)  (0 for zero / non-finite maxima); |e| <= X3_EXP_CLAMP
constexpr int X3_EXP_CLAMP = 60;
__host__ __device__ __forceinline__ int x3_exp_for_bits(unsigned bits) {   // bits of a non-negative float
  const int be = (int)((bits >> 23) & 255u);
  if ((bits & 0x7fffffffu) == 0u || be == 255) return 0;
  // normal: maxabs in [2^(be-127), 2^(be-126));  subnormal maxima are treated like the smallest normal
  const int e = 14 - ((be ? be : 1) - 127);
  return e > X3_EXP_CLAMP ? X3_EXP_CLAMP : (e < -X3_EXP_CLAMP ? -X3_EXP_CLAMP : e);
}
// v * 2^e, exact (v_ldexp_f32)
__device__ __forceinline__ float x3_scale(float v, int e) { return __builtin_ldexpf(v, e); }
typedef s16_t s16x2 __attribute__((ext_vector_type(2)));

#define S16_MFMA32(a, b, c) S16_MFMA32_BUILTIN(*(const coper::s16x8*)&(a), *(const coper::s16x8*)&(b), (c), 0, 0, 0)
#define S16_MFMA16(a, b, c) S16_MFMA16_BUILTIN(*(const coper::s16x8*)&(a), *(const coper::s16x8*)&(b), (c), 0, 0, 0)

// 8 consecutive values -> the 16 bytes of the hi plane and of the lo plane (plain casts: round-to-nearest-even)
__device__ __forceinline__ void split8_s16(const float* v, uint4& hi, uint4& lo) {
  unsigned hw[4], lw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = s16_clamp(v[2 * j]), b = s16_clamp(v[2 * j + 1]);
    s16x2 hp = {(s16_t)a, (s16_t)b};
    const float r0 = a - (float)hp[0], r1 = b - (float)hp[1];
    s16x2 lp = {(s16_t)r0, (s16_t)r1};
    hw[j] = __builtin_bit_cast(unsigned, hp);
    lw[j] = __builtin_bit_cast(unsigned, lp);
  }
  hi = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  lo = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}
// The same split for values KNOWN to lie in [0, 65504) (the fused encoder's conv activations: after the ReLU, bounded by e_x):
// no clamp; hi by v_cvt_pk_f16_f32 (two values per instruction, round-to-nearest-even like the cast), and the second term by ONE
// mixed-precision instruction per value -- v_fma_mixlo_f16 / v_fma_mixhi_f16: fp16(fma(hi, -1, a)) written to one half of the
// destination; a - hi is exact in fp32 (the difference of a float and its fp16 rounding is representable), so the bits are
// split8_s16's.  12 vector instructions per 8 values where the compiler's form of split8_s16 takes 40.
__device__ __forceinline__ void split8_pos_s16(const float* v, uint4& hi, uint4& lo) {
  unsigned hw[4], lw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hw[j]) : "v"(a), "v"(b));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lw[j]) : "v"(hw[j]), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw[j]) : "v"(hw[j]), "v"(b));
  }
  hi = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  lo = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

// one value -> its two 16-bit patterns
__device__ __forceinline__ void split1_s16(float v, unsigned short& hi, unsigned short& lo) {
  const float a = s16_clamp(v);
  const s16_t h = (s16_t)a;
  const s16_t l = (s16_t)(a - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

}  // namespace coper
