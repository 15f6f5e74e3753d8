// COPER_SCORE_BF16X3 encoder, fused conv + generated dense for tiles of 33..128 queries.
//
// The unfused pair (k_conv3x3_bn_relu_bf16 -> x planes in HBM -> k_dense_reg_bf16x3) is bound by the x
// traffic: 64-B pieces of 16 different rows per instruction make x 42 % of the dense kernel's memory
// requests for 26 % of its bytes, and the conv kernel spends its time writing those bytes.  Here a workgroup
// (one relation tile x one K slice) keeps the slice's rows of the e1 images in LDS and produces each k-step's
// x fragments itself -- 3x3 conv, BN, ReLU, hi/lo split -- straight into the B-operand layout in LDS, one
// k-step ahead of the MFMAs that consume them.  HBM sees the weight stream (read once, non-temporal, PF
// k-steps ahead in registers of the owning wave, as in k_dense_reg_bf16x3), the gathered e1 rows and z.
//
// Supported when the filter is 3x3 with C = 32 channels (one k-step of 32 features = one output pixel), no
// concat_rel tail, and the conv filters are either shared or generated together with the dense weights (a
// tile is then one relation).  Arithmetic per x value is conv_x8() below, shared with the stand-alone conv
// kernel that still serves the <= 32-query tiles: x, and with it h[b], stays a pure function of (e1, rel).
#include "coper_internal.h"
#include "conv_fold.h"

namespace coper {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&(a), *(const bf16x8*)&(b), (c), 0, 0, 0)
#define MFMA16_X3(ahi, alo, bhi, blo, c) \
  { (c) = MFMA16_BF16(alo, bhi, c); (c) = MFMA16_BF16(ahi, blo, c); (c) = MFMA16_BF16(ahi, bhi, c); }

#ifndef COPER_FUSED_SCHED
#define COPER_FUSED_SCHED 0
#endif
#ifndef COPER_FUSED_PF
#define COPER_FUSED_PF 3
#endif

struct FusedConvArgs {
  const int64_t* e1;
  const int64_t* rel;
  const float* e1_rows;
  const int32_t* perm;
  const float* ent;
  int64_t shard_lo, n_local, R;
  const float* rel_emb;
  const float* conv_w;
  const float* conv_b;
  const float* scale;
  const float* shift;
  int per_rel_conv, d, r, in_w, in_hw, Wo, img_stride;
};

// One wave's share of a tile x slice.  WAVE is a template argument so that everything a k-step does -- which
// feature blocks (fb = WAVE, WAVE+4, ...) and which x fragments (f = 3-WAVE, 7-WAVE) the wave owns -- is static
// and the whole step is ONE basic block: the scheduler can then put the conv's VALU work into the issue
// slots the MFMAs leave free (an MFMA 16x16x32 holds the vector issue port for 8 of its 16 cycles).
template <int NFB, int NB, int WAVE>
__device__ __forceinline__ void dense_fused_body(uint4* __restrict__ xring, float* __restrict__ img,
                                                 const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                 const FusedConvArgs& A, int64_t relw, int start, int n, int fb0,
                                                 int nfb, int64_t ks32n, int64_t kb, int64_t ke,
                                                 float* __restrict__ zdst, int d_pad16) {
  constexpr int P = COPER_FUSED_PF;
  constexpr int NOWN = (NFB - WAVE + 3) / 4;                       // feature blocks of this wave
  constexpr int NFR = 3 - WAVE < NB ? (NB - (3 - WAVE) + 3) / 4 : 0;  // x fragments this wave produces
  constexpr int XSTAGE = 2 * NB * 64;                              // uint4 per stage: x hi [NB] | x lo [NB]
  const int lane = threadIdx.x & 63;
  const int nk = (int)(ke - kb);
  // ---- weight stream: start it before anything else
  const uint4* wp[NOWN][2];
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    int fb = fb0 + WAVE + 4 * j;
    if (fb > nfb - 1) fb = nfb - 1;
    int64_t o = ((relw * nfb + fb) * ks32n + kb) * 64 + lane;
    wp[j][0] = Whi + o;
    wp[j][1] = Wlo + o;
  }
  u32x4 W[P][NOWN][2];
#define W_ISSUE(s, kk)                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NOWN; ++j) {                                          \
      W[s][j][0] = __builtin_nontemporal_load((const u32x4*)(wp[j][0] + (int64_t)(kk)*64));     \
      W[s][j][1] = __builtin_nontemporal_load((const u32x4*)(wp[j][1] + (int64_t)(kk)*64));     \
    }                                                                                           \
  }
#pragma unroll
  for (int t = 0; t < P; ++t)
    if (t < nk) W_ISSUE(t, t);
  // ---- the slice's rows of the tile's images -> LDS  (pixel p = k-step index; rows i_lo .. i_hi + 2)
  const int Wo = A.Wo, in_w = A.in_w;
  const int i_lo = (int)(kb / Wo);
  const int t0 = i_lo * in_w;
  int t1 = ((int)((ke - 1) / Wo) + 3) * in_w;
  if (t1 > A.in_hw) t1 = A.in_hw;
  for (int qi = WAVE; qi < n; qi += 4) {
    const int64_t q = __builtin_amdgcn_readfirstlane(A.perm[start + qi]);
    int64_t rid = A.rel[q];
    if (rid < 0 || rid >= A.R) rid = 0;
    const int64_t row = A.e1_rows ? q : A.e1[q] - A.shard_lo;
    const bool ok = A.e1_rows || (row >= 0 && row < A.n_local);
    const float* src = (A.e1_rows ? A.e1_rows : A.ent) + row * A.d;
    float* dst = img + qi * A.img_stride - t0;
    for (int t = t0 + lane; t < t1; t += 64) {
      float v;
      if (t < A.d) v = ok ? src[t] : 0.f;
      else v = A.rel_emb[rid * A.r + (t - A.d)];
      dst[t] = v;
    }
  }
  // ---- folded taps of this lane's channel octet
  const int g = lane >> 4;
  float tap[9][8], bs[8];
  {
    const float* wsrc = A.per_rel_conv ? A.conv_w + relw * (int64_t)(9 * 32) : A.conv_w;
    const float* bsrc = A.per_rel_conv ? A.conv_b + relw * (int64_t)32 : A.conv_b;
    conv_fold_taps(wsrc, bsrc, A.scale, A.shift, 32, 8 * g, tap, bs);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // image rows of this lane's query in each of the wave's fragments (padding lanes repeat the last query)
  const float* qimg[NFR > 0 ? NFR : 1];
#pragma unroll
  for (int t = 0; t < NFR; ++t) {
    int qi = (3 - WAVE + 4 * t) * 16 + (lane & 15);
    if (qi > n - 1) qi = n - 1;
    qimg[t] = img + qi * A.img_stride;
  }
  const unsigned xaddr = (unsigned)(uintptr_t)(xring + lane);   // LDS byte address of this lane's fragment piece
  int poff = (int)(kb - (int64_t)i_lo * Wo);   // pixel offset ci*in_w + cj of the NEXT conv step
  int cj = poff;
  // conv of the next pixel for the wave's fragments -> ring stage `stage`; advance = 0 freezes the pixel
  // (the step past the slice's end recomputes the last pixel instead of branching around the conv)
#define CONV_LOAD()                                                                               \
  float cw[NFR > 0 ? NFR : 1][9];                                                                 \
  _Pragma("unroll") for (int t = 0; t < NFR; ++t) {                                               \
    const float* r0 = qimg[t] + poff;                                                             \
    cw[t][0] = r0[0]; cw[t][1] = r0[1]; cw[t][2] = r0[2];                                         \
    cw[t][3] = r0[in_w]; cw[t][4] = r0[in_w + 1]; cw[t][5] = r0[in_w + 2];                        \
    cw[t][6] = r0[2 * in_w]; cw[t][7] = r0[2 * in_w + 1]; cw[t][8] = r0[2 * in_w + 2];            \
  }
#define CONV_COMPUTE(stage, advance)                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int t = 0; t < NFR; ++t) {                                             \
      float y[8];                                                                                 \
      conv_x8(cw[t], tap, bs, y);                                                                 \
      uint4 h4, l4;                                                                               \
      split8_bf16(y, h4, l4);                                                                     \
      uint4* dst = xring + (stage)*XSTAGE + (3 - WAVE + 4 * t) * 64 + lane;                       \
      dst[0] = h4;                                                                                \
      dst[NB * 64] = l4;                                                                          \
    }                                                                                             \
    const int wrap = (cj + 1 == Wo);                                                              \
    poff += (advance) ? (wrap ? in_w - Wo + 1 : 1) : 0;                                           \
    cj = (advance) ? (wrap ? 0 : cj + 1) : cj;                                                    \
  }
#define CONV_STEP(stage, advance) { CONV_LOAD(); CONV_COMPUTE(stage, advance); }

  f32x4 acc[NOWN][NB];
#pragma unroll
  for (int j = 0; j < NOWN; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  CONV_STEP(0, 1 < nk);
  for (int k0 = 0; k0 < nk; k0 += 2 * P) {
#pragma unroll
    for (int s2 = 0; s2 < 2 * P; ++s2) {   // unrolled over lcm(ring stages, prefetch depth): static indices
      const int k = k0 + s2;
      const int s = s2 % P;
      if (k < nk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my x writes of step k have landed in LDS
        __builtin_amdgcn_s_barrier();                         // everyone's have; stage (k+1)&1 is free again
        const uint4* xb = xring + (s2 & 1) * XSTAGE + lane;
        // every LDS read of the step is issued here and waited for once (one wave per SIMD: nobody else
        // hides LDS latency, and left alone hipcc sinks each read to just before its first use): the empty
        // asm "uses" all of them, so they cannot move below it and nothing that needs them moves above
        CONV_LOAD();
        u32x4 bh[NB], bl[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) { bh[q] = *(const u32x4*)(xb + q * 64); bl[q] = *(const u32x4*)(xb + (NB + q) * 64); }
#pragma unroll
        for (int q = 0; q < NB; ++q) asm volatile("" : "+v"(bh[q]), "+v"(bl[q]));
#pragma unroll
        for (int t = 0; t < NFR; ++t)
#pragma unroll
          for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(cw[t][i]));
#ifndef COPER_DBG_FUSED_NO_CONV
        CONV_COMPUTE((s2 + 1) & 1, k + 2 < nk);
#endif
#ifdef COPER_DBG_FUSED_NO_MFMA
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
          acc[j][0][0] += __uint_as_float(W[s][j][0][0] ^ W[s][j][1][1]);
          acc[j][0][1] += __uint_as_float(W[s][j][0][2] ^ W[s][j][1][3]);
        }
        acc[0][0][2] += __uint_as_float(bh[0][0]);
#else
        // term-major MFMA order: consecutive MFMAs write different accumulators; each accumulator still
        // sees lo*hi, hi*lo, hi*hi in that order
#pragma unroll
        for (int q = 0; q < NB; ++q) {
#pragma unroll
          for (int j = 0; j < NOWN; ++j) acc[j][q] = MFMA16_BF16(W[s][j][1], bh[q], acc[j][q]);
#pragma unroll
          for (int j = 0; j < NOWN; ++j) acc[j][q] = MFMA16_BF16(W[s][j][0], bl[q], acc[j][q]);
#pragma unroll
          for (int j = 0; j < NOWN; ++j) acc[j][q] = MFMA16_BF16(W[s][j][0], bh[q], acc[j][q]);
        }
#endif
#if !defined(COPER_FUSED_NO_SCHED) && !defined(COPER_DBG_FUSED_NO_MFMA) && !defined(COPER_DBG_FUSED_NO_CONV)
        {
          // issue order: the conv's patch reads and the first x fragments, then per query block the next
          // block's two fragment reads followed by its MFMAs, each with VPM conv VALU instructions behind it
          constexpr int NM = 3 * NOWN, NV = NFR * 128 + 16, VPM = (NV + NM * NB - 1) / (NM * NB);
#if COPER_FUSED_SCHED == 1
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
          for (int q = 0; q < NB; ++q) {
            if (q + 1 < NB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
          }
#elif COPER_FUSED_SCHED == 2
          // all DS reads up front
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * NB + 9 * NFR, 0);
#pragma unroll
          for (int i = 0; i < NM * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          }
#else
#pragma unroll
          for (int i = 0; i < NM * NB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          }
#endif
        }
#endif
#ifndef COPER_DBG_FUSED_NO_W
        if (k + P < nk) W_ISSUE(s, k + P);
#endif
      }
    }
  }
#undef W_ISSUE
#undef CONV_STEP
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    int fb = fb0 + WAVE + 4 * j;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      int qi = q * 16 + (lane & 15);
      if (fb < nfb && qi < n) {
        float* dst = zdst + (int64_t)(start + qi) * d_pad16 + fb * 16 + 4 * (lane >> 4);
        *(float4*)dst = make_float4(acc[j][q][0], acc[j][q][1], acc[j][q][2], acc[j][q][3]);
      }
    }
  }
}

template <int NFB>
__global__ __launch_bounds__(256) void k_dense_fused_bf16x3(const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                            FusedConvArgs A, const int32_t* __restrict__ tiles,
                                                            const int32_t* __restrict__ n_tiles, int64_t cap_small,
                                                            int nfb, int64_t ks32n, int nslices, int64_t Bcap,
                                                            int d_pad16, float* __restrict__ z_part) {
  extern __shared__ uint4 fused_lds[];
  int tile = blockIdx.x;
  if (tile >= n_tiles[1]) return;
  const int32_t* tl = tiles + 4 * (cap_small + tile);
  const int slice = blockIdx.y;
  const int fb0 = blockIdx.z * NFB;
  const int64_t relw = __builtin_amdgcn_readfirstlane(tl[0]);
  const int start = __builtin_amdgcn_readfirstlane(tl[1]);
  const int n = __builtin_amdgcn_readfirstlane(tl[2]);
  const int64_t kb = ks32n * slice / nslices, ke = ks32n * (slice + 1) / nslices;
  float* zdst = z_part + (int64_t)slice * Bcap * d_pad16;
  const int nb = (n + 15) >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint4* xring = fused_lds;                       // 2 stages x 16 slots x 1 KiB
  float* img = (float*)(fused_lds + 2 * 16 * 64);
#define BODYW(NB_, W_) dense_fused_body<NFB, NB_, W_>(xring, img, Whi, Wlo, A, relw, start, n, fb0, nfb, ks32n, kb, ke, zdst, d_pad16)
#define BODY(NB_)                                   \
  switch (wave) {                                   \
    case 0: BODYW(NB_, 0); break;                   \
    case 1: BODYW(NB_, 1); break;                   \
    case 2: BODYW(NB_, 2); break;                   \
    default: BODYW(NB_, 3); break;                  \
  }
  switch (nb) {
    case 3: BODY(3); break;
    case 4: BODY(4); break;
    case 5: BODY(5); break;
    case 6: BODY(6); break;
    case 7: BODY(7); break;
    default: BODY(8); break;
  }
#undef BODY
#undef BODYW
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
  int stride = fused_rows_max(dm, nslices) * dm.in_w;
  stride |= 1;
  size_t lds = (size_t)2 * 16 * 64 * sizeof(uint4) + (size_t)128 * stride * sizeof(float);
  return lds <= 160 * 1024;
}

template <int NFB>
static void dense_fused_launch(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                               int nslices, int zgroups, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t cap_small = (dm.gen_fc ? dm.R : 1) + 1;
  int64_t n_big_max = B / 33 + 1;
  FusedConvArgs A;
  A.e1 = e1; A.rel = rel; A.e1_rows = e1_rows; A.perm = h->perm;
  A.ent = h->params["ent_emb"].ptr;
  A.shard_lo = h->cfg.shard_lo; A.n_local = dm.n_local; A.R = dm.R;
  A.rel_emb = dm.lookup ? nullptr : h->params["rel_emb"].ptr;
  A.conv_w = dm.gen_conv ? h->conv_w_rel : h->params["conv1_weights"].ptr;
  A.conv_b = dm.gen_conv ? h->conv_b_rel : h->params["conv1_bias"].ptr;
  A.scale = h->conv_scale; A.shift = h->conv_shift;
  A.per_rel_conv = dm.gen_conv ? 1 : 0;
  A.d = dm.d; A.r = dm.r; A.in_w = dm.in_w; A.in_hw = dm.in_h * dm.in_w; A.Wo = dm.Wo;
  A.img_stride = (fused_rows_max(dm, nslices) * dm.in_w) | 1;
  size_t lds = (size_t)2 * 16 * 64 * sizeof(uint4) + (size_t)128 * A.img_stride * sizeof(float);
  if (!h->fused_attr_done) {   // process-wide attribute: always the hardware maximum, whatever this handle needs
    (void)hipFuncSetAttribute((const void*)k_dense_fused_bf16x3<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    h->fused_attr_done = true;
  }
  hipLaunchKernelGGL((k_dense_fused_bf16x3<NFB>), dim3((unsigned)n_big_max, (unsigned)nslices, (unsigned)zgroups), dim3(256),
                     lds, s, (const uint4*)h->Wf16_hi, (const uint4*)h->Wf16_lo, A, h->tiles, h->n_tiles, cap_small, dm.nfb,
                     dm.F_pad / 32, nslices, h->ws_queries, dm.d_pad16, h->z_part);
}

int launch_dense_fused_bf16(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                            int nslices, hipStream_t s) {
  if (B <= 32) return COPER_OK;   // no tile above 32 queries can exist
  if (h->dm.nfb == 13) dense_fused_launch<13>(h, e1, rel, e1_rows, B, nslices, 1, s);
  else dense_fused_launch<8>(h, e1, rel, e1_rows, B, nslices, (h->dm.nfb + 7) / 8, s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
