// The 16-bit split of the "x3" arithmetic (COPER_SCORE_BF16X3): every fp32 operand x of a matrix product becomes two
// 16-bit terms  x = hi + lo  (hi = rne16(x), lo = rne16(x - hi))  and a product is  lo*hi + hi*lo + hi*hi  on the 16-bit
// matrix cores with fp32 accumulation.  The bf16 and the fp16 MFMAs run at the same rate (same cycles per instruction:
// MI355X_MICROARCH.md, Matrix cores); what differs is where the 16 bits go:
//   bf16 (8-bit exponent, 8-bit significand):  hi + lo carries 16-17 significant bits of x (error <= 2^-16 |x|), any fp32
//        magnitude;
//   fp16 (5-bit exponent, 11-bit significand): hi + lo carries 22-23 significant bits (error <= 2^-22 |x| while
//        |x| >= 2^-3: lo normal; 2^-25 absolute below that -- gfx950's fp16 MFMA keeps subnormal inputs: measured,
//        tools/microbench/mfma_shape.hip), for |x| < 65,504.
// Round 3 measured what the 8 missing bits cost (tools/rank_decomp.py, FB15k-237-shaped pass): the bf16 split's logit
// error moves 2.1 % of the ranks, the bf16-split encoder's error in h another 2.1 %, against 0.4 % for fp32 arithmetic
// throughout -- and the exact band around the target (kernels_score3_bf16.hip) has to be 16x wider, with 16x the pairs to
// re-score.  So the split is fp16 (COPER_SPLIT_BF16 builds the round-2 arithmetic for A/B).  Range: operands are clamped to
// +-65,504 before the split (an embedding, weight or activation beyond that saturates instead of becoming inf; prepare
// refuses an entity table whose largest row norm says the clamp could matter).
#pragma once
#include <hip/hip_runtime.h>

namespace coper {

#ifdef COPER_SPLIT_BF16
typedef __bf16 s16_t;
#define S16_NAME "bf16"
#define S16_MFMA32_BUILTIN __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define S16_MFMA16_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_bf16
__device__ __forceinline__ float s16_clamp(float v) { return v; }
#else
typedef _Float16 s16_t;
#define S16_NAME "fp16"
#define S16_MFMA32_BUILTIN __builtin_amdgcn_mfma_f32_32x32x16_f16
#define S16_MFMA16_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_f16
__device__ __forceinline__ float s16_clamp(float v) { return __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f); }   // (NaN stays NaN)
#endif
typedef s16_t s16x8 __attribute__((ext_vector_type(8)));
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
// one value -> its two 16-bit patterns
__device__ __forceinline__ void split1_s16(float v, unsigned short& hi, unsigned short& lo) {
  const float a = s16_clamp(v);
  const s16_t h = (s16_t)a;
  const s16_t l = (s16_t)(a - (float)h);
  hi = __builtin_bit_cast(unsigned short, h);
  lo = __builtin_bit_cast(unsigned short, l);
}

}  // namespace coper
