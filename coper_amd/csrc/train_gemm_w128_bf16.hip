// Split-bf16 NT GEMM of the training step, large shapes: ONE WAVE owns a 128 x 128 output tile.
//
// The 4-wave kernel of train_gemm_bf16.hip (64 x 64 per wave) requests 683 B of fragments per MFMA: 85 B/clk per CU at
// full matrix rate against the 64 B/clk a CU's vector-memory path delivers -- it measured 130-165 us for the 30 GFLOP
// shapes of the FB15k-237 step, 40 % matrix-pipe use, whatever the grid mapping or K split.  Here a wave keeps all 16
// accumulator tiles of 32 x 32 (256 AGPRs: the whole accumulator file of a SIMD, so one wave per SIMD -- a workgroup is a
// single wave) and streams 4 + 4 fragments x 2 planes per k-step for 48 MFMAs: 341 B per MFMA = 42 B/clk per CU.
// Fragments are fetched two k-steps (96 MFMAs, ~1.5 us) ahead into three register buffers (192 VGPRs).
// Few-tile shapes get their parallelism from K slices (tg_split_k: tiles x slices ~ 1024 SIMDs).
// Measured (MI355X, FB15k-237 step, 30 GFLOP each): T = x P 82 us + 19 us slice sum (four-wave kernel: 130), dP 95 (124),
// dx 83 + 16 (139); main loop alone 68 us, without its loads 54 us (COPER_DBG_W128_NO_EPI / _NO_LOADS builds).
// This file is built WITHOUT -amdgpu-mfma-vgpr-form: the accumulators must live in AGPRs.
// Round 5: fp16 planes carrying one power of two per operand (train_gemm.h); the epilogue takes 2^(e_X + e_Y) out.
#include "coper_internal.h"
#include "split16.h"
#include "train_gemm.h"

namespace coper {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TGW_MFMA(a, b, c) S16_MFMA32(a, b, c)

__device__ __forceinline__ int64_t tgw_off(const TgIdx& a, int64_t i) {
  return a.seg > 0 ? (i / a.seg) * a.s_hi + (i % a.seg) * a.s_lo : i * a.s_lo;
}

template <int NBUF>
__global__ __launch_bounds__(256) void k_gemm_nt_w128_bf16x3(const uint4* __restrict__ Xhi, const uint4* __restrict__ Xlo,
                                                            const uint4* __restrict__ Yhi, const uint4* __restrict__ Ylo, int KS16_all, int KST,
                                                            float* __restrict__ C, TgIdx ci, TgIdx cj, int64_t M, int64_t N, int nsplit,
                                                            float* __restrict__ part, double* __restrict__ sumsq, int cs,
                                                            const int32_t* __restrict__ ex, const int32_t* __restrict__ ey) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int se = -(*ex + *ey);
  // a workgroup is four independent waves: the (up to) 4 i tiles of one (j tile, K slice), which share the Y fragments
  const int ti = (int)((M + 127) / 128), tj = (int)((N + 127) / 128), nic = (ti + 3) / 4;
  const int c = blockIdx.x;
  const int bx = (c % nic) * 4 + wave, by = (c / nic) % tj, bz = c / (nic * tj);
  if (bx >= ti) return;
  const int64_t ib0 = (int64_t)bx * 4, jb0 = (int64_t)by * 4;
  const int kb = (int)((int64_t)KS16_all * bz / nsplit), KS16 = (int)((int64_t)KS16_all * (bz + 1) / nsplit) - kb;
  Xhi += (int64_t)kb * 64 + lane; Xlo += (int64_t)kb * 64 + lane; Yhi += (int64_t)kb * 64 + lane; Ylo += (int64_t)kb * 64 + lane;
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  uint4 xh[NBUF][4], xl[NBUF][4], yh[NBUF][4], yl[NBUF][4];
#define TGW_LOAD(s_, ks_)                                                                    \
  {                                                                                          \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                          \
      const int64_t o = ((ib0 + a) * KST + (ks_)) * 64;                                       \
      xh[s_][a] = Xhi[o]; xl[s_][a] = Xlo[o];                                                \
    }                                                                                        \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                          \
      const int64_t o = ((jb0 + b) * KST + (ks_)) * 64;                                       \
      yh[s_][b] = Yhi[o]; yl[s_][b] = Ylo[o];                                                \
    }                                                                                        \
  }
  // term-major: consecutive MFMAs write different accumulators; every accumulator sees lo*hi, hi*lo, hi*hi in that order
#define TGW_STEP(s_)                                                                         \
  {                                                                                          \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b) \
      acc[a][b] = TGW_MFMA(xl[s_][a], yh[s_][b], acc[a][b]);                                 \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b) \
      acc[a][b] = TGW_MFMA(xh[s_][a], yl[s_][b], acc[a][b]);                                 \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) _Pragma("unroll") for (int b = 0; b < 4; ++b) \
      acc[a][b] = TGW_MFMA(xh[s_][a], yh[s_][b], acc[a][b]);                                 \
  }
#define TGW_KCL(k_) ((k_) < KS16 ? (k_) : KS16 - 1)
  static_assert(NBUF == 3, "three register buffers");
  TGW_LOAD(0, 0);
  TGW_LOAD(1, TGW_KCL(1));
  int ks = 0;
#ifdef COPER_DBG_W128_NO_LOADS
  TGW_LOAD(2, TGW_KCL(2));
#define TGW_LOADX(s_, k_)
#else
#define TGW_LOADX(s_, k_) TGW_LOAD(s_, k_)
#endif
  for (; ks + 3 <= KS16; ks += 3) {
    TGW_LOADX(2, TGW_KCL(ks + 2));
    __builtin_amdgcn_sched_barrier(0);
    TGW_STEP(0);
    TGW_LOADX(0, TGW_KCL(ks + 3));
    __builtin_amdgcn_sched_barrier(0);
    TGW_STEP(1);
    TGW_LOADX(1, TGW_KCL(ks + 4));
    __builtin_amdgcn_sched_barrier(0);
    TGW_STEP(2);
  }
  if (ks < KS16) TGW_STEP(0);
  if (ks + 1 < KS16) TGW_STEP(1);
#undef TGW_KCL
#undef TGW_LOAD
#undef TGW_STEP
#ifdef COPER_DBG_W128_NO_EPI
  {
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) t += acc[a][b][0];
    if (t == 1.2345f) C[0] = t;
    return;
  }
#endif
  // register r of lane l: i = (r & 3) + 8 (r >> 2) + 4 (l >> 5), j = l & 31.  ONE store path with static register
  // indices (partial sums and C differ in base and strides only; the host sends two-level row views to the other kernel):
  // with two paths the 2 x 256 stores were left as loops over the accumulator array, i.e. through scratch -- the epilogue
  // then cost 125-215 us on top of a 68 us main loop.
  float ss = 0.f;
  float* const base = nsplit > 1 ? part + (int64_t)bz * M * N : C;
  const int64_t si = nsplit > 1 ? N : ci.s_lo;
  float* pj[4];
  bool jok[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int64_t j = (jb0 + b) * 32 + (lane & 31);
    jok[b] = j < N;
    pj[b] = base + (nsplit > 1 ? j : tgw_off(cj, jok[b] ? j : 0));
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i0 = (ib0 + a) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t i = i0 + (r & 3) + 8 * (r >> 2);
      const int64_t oi = i * si;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (i < M && jok[b]) {
          const float v = x3_scale(acc[a][b][r], se);
          pj[b][oi] = v;
          ss = fmaf(v, v, ss);
        }
    }
  }
  if (sumsq && nsplit <= 1) {
    double w = (double)ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    if (lane == 0 && w != 0.0) atomicAdd(sumsq + (blockIdx.x * 4 + wave) % TG_SUMSQ_SLOTS, w);
  }
}

void tg_launch_w128(const TgPlanes& X, int64_t M, const TgPlanes& Y, int64_t N, int KS16, int KST, float* C, TgIdx ci, TgIdx cj, hipStream_t s,
                    int nsplit, float* part, double* sumsq) {
  const int ti = (int)((M + 127) / 128), tj = (int)((N + 127) / 128);
  const int64_t nclu = (int64_t)((ti + 3) / 4) * tj * nsplit;
  dim3 grid((unsigned)nclu);
  hipLaunchKernelGGL((k_gemm_nt_w128_bf16x3<3>), grid, dim3(256), 0, s, X.hi, X.lo, Y.hi, Y.lo, KS16, KST, C, ci, cj, M, N, nsplit, part, sumsq, 4, X.exp, Y.exp);
}

}  // namespace coper
