// Conv + BN + ReLU arithmetic of the COPER_SCORE_BF16X3 encoder, shared by the stand-alone conv kernel
// (kernels_encode_bf16.hip) and the fused conv + dense kernel (kernels_dense_fused_bf16.hip) so that both
// produce the same x bits for the same (e1, rel):
//   inference BN folded into the filter:  tap'[k][c] = tap[k][c] * scale[c],  b'[c] = bias[c]*scale[c] + shift[c]
//   x[c] = max(fma chain over the 9 taps started from b'[c], 0)           (models.py:390-404 at inference)
//   x = hi + lo,  hi = fp16(x 2^e_x), lo = fp16(x 2^e_x - hi)      (split16.h)
// Round 4: the power of two of the activations (e_x, one per handle, from a bound on x computed at prepare: coper_abi.hip) is
// folded into the taps and the bias -- a positive power of two commutes with every fma of the chain and with the ReLU, so the
// chain below yields x 2^e_x exactly (no overflow / underflow) at no cost per value.
#pragma once
#include <hip/hip_runtime.h>

#include "split16.h"

namespace coper {

__device__ __forceinline__ void conv_fold_taps(const float* __restrict__ w, const float* __restrict__ b,
                                               const float* __restrict__ scale, const float* __restrict__ shift, int C,
                                               int c0, int x_exp, float (&tap)[9][8], float (&bs)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float sc = scale[c0 + c];
    bs[c] = x3_scale(fmaf(b[c0 + c], sc, shift[c0 + c]), x_exp);
#pragma unroll
    for (int k = 0; k < 9; ++k) tap[k][c] = x3_scale(w[k * C + c0 + c] * sc, x_exp);
  }
}

__device__ __forceinline__ void conv_x8(const float (&w)[9], const float (&tap)[9][8], const float (&bs)[8],
                                        float (&y)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float a = bs[c];
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(w[k], tap[k][c], a);
    y[c] = fmaxf(a, 0.f);
  }
}

__device__ __forceinline__ void split8_bf16(const float* v, uint4& hi, uint4& lo) { split8_s16(v, hi, lo); }   // (split16.h: fp16 since round 3)

}  // namespace coper
