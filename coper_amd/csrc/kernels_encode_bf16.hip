// COPER_SCORE_BF16X3 encoder: the conv stage writes x as two bf16 planes (hi / lo split of the fp32
// activation, kernels_score_bf16.hip header), the generated-dense layer runs on v_mfma_f32_16x16x32_bf16
// with the split product  lo*hi + hi*lo + hi*hi  (3 MFMAs per 32 k at the bf16 rate instead of 8 exact-f32
// MFMAs: 5.3x the matrix throughput), fp32 accumulation, fp32 z / h.  With the matrix time out of the way
// the layer is bound by the one pass over the per-relation weight cache (HBM).
//
// Same decomposition as kernels_encode.hip: fixed K slices (a function of F only), one accumulator chain
// per (query, feature, slice) whichever kernel produces it, slices summed in order by k_dense_finalize:
// h[b] stays a pure function of (e1[b], rel[b]).
//
//   A operand (features on rows):  lane l holds 8 bf16 W[f = 32ks + 8(l>>4) + j][feat = 16fb + (l&15)]
//   B operand (queries on columns): lane l holds 8 bf16 x[q = l&15][f = 32ks + 8(l>>4) + j]
//   D: col = lane&15 (query), row = 4(lane>>4) + reg (feature).
#include <cstring>

#include "bf16x3_chain.h"
#include "coper_internal.h"
#include "conv_fold.h"

namespace coper {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8_e(const float* v, uint4& hi, uint4& lo) { split8_s16(v, hi, lo); }

#define MFMA16_BF16(a, b, c) S16_MFMA16(a, b, c)
#define MFMA16_X3(ahi, alo, bhi, blo, c) \
  { (c) = MFMA16_BF16(alo, bhi, c); (c) = MFMA16_BF16(ahi, blo, c); (c) = MFMA16_BF16(ahi, bhi, c); }

// ------------------------------------------------------------------------------------------------
// prepare: fp32 fragment image Wf (v_mfma_f32_16x16x4 layout, built by k_gen_dense_frag / k_dense_frag_copy)
// -> two bf16 planes in the 16x16x32 layout:  W16[(rel*nfb + fb)*ks32n + ks][lane] = 8 bf16.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wfrag_to_bf16(const float4* __restrict__ Wf, int64_t n_relfb, int64_t ks32n,
                                                       uint4* __restrict__ hi, uint4* __restrict__ lo, int nfb,
                                                       const int32_t* __restrict__ w_exp, int64_t ks32s) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (relfb*ks32n + ks)*64 + lane
  if (j >= n_relfb * ks32n * 64) return;
  int l = (int)(j & 63);
  int64_t rest = j >> 6;
  int64_t ks = rest % ks32n;
  int64_t relfb = rest / ks32n;
  // target f = 32ks + 8(l>>4) + jj  ->  source k-step 2ks + (l>>5), k-groups g = 2((l>>4)&1), g+1, feature lane l&15
  int64_t ks16 = 2 * ks + (l >> 5);
  int g = 2 * ((l >> 4) & 1);
  const float4* src = Wf + (relfb * (2 * ks32n) + ks16) * 64 + (l & 15);
  float4 a = src[16 * g], b = src[16 * (g + 1)];
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const int ew = w_exp[relfb / nfb];       // the relation's power of two (split16.h): the planes hold W_r 2^e_W
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = x3_scale(v[t], ew);
  uint4 h4, l4;
  const int64_t jo = (relfb * ks32s + ks) * 64 + l;     // (ks32s: k-steps between two feature blocks of the image, coper_internal.h)
  split8_e(v, h4, l4);
  hi[jo] = h4;
  lo[jo] = l4;
}

// largest |W| of every relation's fragment image -> its power of two (split16.h): w_exp[rel] first accumulates the maximum's
// float bits (non-negative floats order like unsigned integers), k_bits_to_exp turns them into exponents in place
__global__ __launch_bounds__(256) void k_w_absmax(const float4* __restrict__ Wf, int64_t per_rel4, int32_t* __restrict__ w_exp) {
  const int64_t rel = blockIdx.y;
  const float4* src = Wf + rel * per_rel4;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_rel4; i += (int64_t)gridDim.x * 256) {
    const float4 v = src[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax((unsigned*)&w_exp[rel], __float_as_uint(m));
}
__global__ void k_bits_to_exp(int32_t* __restrict__ w_exp, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w_exp[i] = x3_exp_for_bits((unsigned)w_exp[i]);
}

int launch_wfrag_to_bf16(coper_handle* h, const float* Wf, int64_t Rw, void* hi, void* lo, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t ks32n = dm.F_pad / 32;
  int64_t total = Rw * dm.nfb * ks32n * 64;
  const int64_t per_rel4 = (int64_t)dm.nfb * (dm.F_pad / 16) * 64;
  COPER_HIP_TRY(h, hipMemsetAsync(h->w_exp, 0, sizeof(int32_t) * Rw, s));
  hipLaunchKernelGGL(k_w_absmax, dim3(64, (unsigned)Rw), dim3(256), 0, s, (const float4*)Wf, per_rel4, h->w_exp);
  hipLaunchKernelGGL(k_bits_to_exp, dim3((unsigned)((Rw + 255) / 256)), dim3(256), 0, s, h->w_exp, Rw);
  hipLaunchKernelGGL(k_wfrag_to_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4*)Wf,
                     Rw * dm.nfb, ks32n, (uint4*)hi, (uint4*)lo, dm.nfb, h->w_exp, w16_ks_stride(dm));
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// A bound on the conv activations, for their power of two (conv_fold.h):  x[c] = relu(b'[c] + sum_k w_k tap'[k][c])
// <= |b'[c]| + in_max sum_k |tap'[k][c]|  with in_max the largest |element| of an input image (entity rows, and relation
// rows when they are stacked under them).  One thread per (relation, channel); *out = the maximum's float bits.  Loose by the
// gap between a 9-term L1 bound and a sum of mixed signs (a few binades at most: the window below the bound is 17 wide).
__global__ void k_x_bound(const float* __restrict__ conv_w, const float* __restrict__ conv_b, const float* __restrict__ scale,
                          const float* __restrict__ shift, int64_t Rc, int C, int taps, float in_max, unsigned* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Rc * C) return;
  const int64_t rel = i / C;
  const int c = (int)(i - rel * C);
  const float sc = scale[c];
  float l1 = 0.f;
  for (int k = 0; k < taps; ++k) l1 += fabsf(conv_w[rel * taps * C + k * C + c] * sc);
  const float b = fabsf(fmaf(conv_b[rel * C + c], sc, shift[c])) + in_max * l1;
  if (b > 0.f) atomicMax(out, __float_as_uint(b * 1.0001f));
}
__global__ void k_absmax_bits(const float* __restrict__ src, int64_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// e_x of the handle (synchronises: prepare).  scratch: two device words.
int compute_x_exp(coper_handle* h, unsigned* scratch, hipStream_t s) {
  const Dims& dm = h->dm;
  const float* cw = dm.gen_conv ? h->conv_w_rel : h->params["conv1_weights"].ptr;
  const float* cb = dm.gen_conv ? h->conv_b_rel : h->params["conv1_bias"].ptr;
  COPER_HIP_TRY(h, hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned), s));
  float rel_max = 0.f;
  if ((dm.stacked || dm.concat_rel) && !dm.lookup) {
    hipLaunchKernelGGL(k_absmax_bits, dim3(64), dim3(256), 0, s, h->params["rel_emb"].ptr, dm.R * (int64_t)dm.r, scratch + 1);
    COPER_HIP_TRY(h, hipMemcpyAsync(&rel_max, scratch + 1, sizeof(float), hipMemcpyDeviceToHost, s));
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
  }
  const float in_max = dm.stacked ? fmaxf(h->x3_ent_absmax, rel_max) : h->x3_ent_absmax;
  {   // e_I: the fused encoder's image planes hold (e1 row | rel row) 2^e_I with the largest input in [2^7, 2^8) -- the middle of
      // fp16's window, eleven binades of full precision below it and eight above (kernels_dense_fused_bf16.hip, round 5)
    unsigned ib;
    memcpy(&ib, &in_max, sizeof ib);
    h->img_exp = x3_exp_for_bits(ib) - 7;
  }
  const int64_t Rc = dm.gen_conv ? dm.R : 1;
  hipLaunchKernelGGL(k_x_bound, dim3((unsigned)((Rc * dm.C + 255) / 256)), dim3(256), 0, s, cw, cb, h->conv_scale, h->conv_shift, Rc, dm.C,
                     dm.fh * dm.fw, in_max, scratch);
  COPER_HIP_TRY(h, hipGetLastError());
  float bound = 0.f;
  COPER_HIP_TRY(h, hipMemcpyAsync(&bound, scratch, sizeof(float), hipMemcpyDeviceToHost, s));
  COPER_HIP_TRY(h, hipStreamSynchronize(s));
  if (dm.concat_rel) bound = fmaxf(bound, rel_max);
  unsigned bits;
  memcpy(&bits, &bound, sizeof bits);
  h->x_exp = x3_exp_for_bits(bits);
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// conv + BN + ReLU -> x planes.  3x3 filters, C a multiple of 8.  lane = (pixel of a 16-pixel group,
// channel octet): 72 taps (9 x 8 channels) in registers, 9 LDS reads of the image per pixel (shared by the
// octet lanes), 72 FMAs, and one 16-B store per plane; a wave-store is 16 pixels x 64 B = 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------
template <int QPB>
__global__ __launch_bounds__(256) void k_conv3x3_bn_relu_bf16(
    const int64_t* __restrict__ e1, const int64_t* __restrict__ rel, const float* __restrict__ e1_rows,
    const int32_t* __restrict__ perm, const float* __restrict__ ent, int64_t shard_lo, int64_t n_local,
    const float* __restrict__ rel_emb, const float* __restrict__ conv_w, const float* __restrict__ conv_b,
    int per_rel_conv, const float* __restrict__ scale, const float* __restrict__ shift, int d, int r, int in_h,
    int in_w, int stacked, int C, int Ho, int Wo, int concat_rel, int64_t F, int64_t F_pad, int64_t R, int64_t B,
    const int32_t* __restrict__ small_tiles, const int32_t* __restrict__ n_tiles, int by_small_tile,
    unsigned short* __restrict__ x_hi, unsigned short* __restrict__ x_lo, int x_exp) {
  extern __shared__ float lds[];  // img[QPB][in_h*in_w]
  const int img_sz = in_h * in_w;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // by_small_tile: the fused conv + dense kernel serves every tile above 32 queries; this launch covers only
  // the <= 32-query tiles, 32 / QPB workgroups per small-tile slot (unused slots exit on one scalar load)
  int64_t pos0 = (int64_t)blockIdx.x * QPB, pos_end = B;
  if (by_small_tile) {
    const int t = blockIdx.x / (32 / QPB);
    if (t >= n_tiles[0]) return;
    const int start = small_tiles[4 * t + 1], n = small_tiles[4 * t + 2];
    pos0 = start + (int64_t)(blockIdx.x % (32 / QPB)) * QPB;
    pos_end = start + n;
    if (pos0 >= pos_end) return;
  }
  int64_t rids[QPB];
  bool live[QPB];
#pragma unroll
  for (int qq = 0; qq < QPB; ++qq) {
    int64_t pos = pos0 + qq;
    rids[qq] = 0;
    live[qq] = pos < pos_end;
    if (!live[qq]) continue;
    int64_t q = perm[pos];
    int64_t rid = rel[q];
    if (rid < 0 || rid >= R) rid = 0;
    rids[qq] = rid;
    float* img = lds + qq * img_sz;
    for (int k = threadIdx.x; k < d; k += 256) {
      float v;
      if (e1_rows) {
        v = e1_rows[q * d + k];
      } else {
        int64_t row = e1[q] - shard_lo;
        v = (row >= 0 && row < n_local) ? ent[row * d + k] : 0.f;
      }
      img[k] = v;
    }
    if (stacked)
      for (int k = threadIdx.x; k < r; k += 256) img[d + k] = rel_emb[rid * r + k];
  }
  __syncthreads();
  const int noct = C >> 3;                 // channel octets per pixel
  const int ppg = 64 / noct;               // pixels per wave-group (16 for C = 32)
  const int pl = lane / noct, oc = lane % noct;
  const int npix = Ho * Wo;
  float tap[9][8], bs[8];
  int64_t tap_rid = -1;
#pragma unroll
  for (int qq = 0; qq < QPB; ++qq) {
    int64_t pos = pos0 + qq;
    if (pos >= B) break;
    if (!live[qq]) continue;
    const int64_t rid = per_rel_conv ? rids[qq] : 0;
    if (rid != tap_rid) {  // workgroup-uniform
      const float* wsrc = per_rel_conv ? conv_w + rid * (int64_t)(9 * C) : conv_w;
      const float* bsrc = per_rel_conv ? conv_b + rid * (int64_t)C : conv_b;
      conv_fold_taps(wsrc, bsrc, scale, shift, C, 8 * oc, x_exp, tap, bs);
      tap_rid = rid;
    }
    const float* img = lds + qq * img_sz;
    unsigned short* xh = x_hi + pos * F_pad;
    unsigned short* xl = x_lo + pos * F_pad;
    int p = wave * ppg + pl;
    int i = p / Wo, j = p - i * Wo;             // one division per query; then (i, j) advance by 4*ppg pixels
    const int di = (4 * ppg) / Wo, dj = (4 * ppg) - di * Wo;
    for (; p < npix; p += 4 * ppg, i += di, j += dj) {
      if (j >= Wo) { j -= Wo; ++i; }
      {
        const float* r0 = img + i * in_w + j;
        float w[9] = {r0[0], r0[1], r0[2], r0[in_w], r0[in_w + 1], r0[in_w + 2],
                      r0[2 * in_w], r0[2 * in_w + 1], r0[2 * in_w + 2]};
        float y[8];
        conv_x8(w, tap, bs, y);
        uint4 h4, l4;
        split8_bf16(y, h4, l4);
        *(uint4*)(xh + (int64_t)p * C + 8 * oc) = h4;
        *(uint4*)(xl + (int64_t)p * C + 8 * oc) = l4;
      }
    }
  }
  // tail of each row: concat_rel columns and the zero padding up to F_pad
#pragma unroll
  for (int qq = 0; qq < QPB; ++qq) {
    int64_t pos = pos0 + qq;
    if (pos >= B) break;
    if (!live[qq]) continue;
    int64_t Fc = (int64_t)npix * C;
    unsigned short* xh = x_hi + pos * F_pad;
    unsigned short* xl = x_lo + pos * F_pad;
    if (concat_rel)
      for (int k = threadIdx.x; k < r; k += 256) {
        float v = x3_scale(rel_emb[rids[qq] * r + k], x_exp);      // (these columns of x carry e_x like the conv's)
        unsigned short hb, lb;
        split1_s16(v, hb, lb);
        xh[Fc + k] = hb;
        xl[Fc + k] = lb;
      }
    for (int64_t k = F + threadIdx.x; k < F_pad; k += 256) { xh[k] = 0; xl[k] = 0; }
  }
}

bool conv_bf16_supported(const Dims& dm) { return dm.fh == 3 && dm.fw == 3 && dm.C % 8 == 0 && 64 % (dm.C / 8) == 0; }

int launch_conv_bf16(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                     bool skip_big, hipStream_t s) {
  const Dims& dm = h->dm;
  const float* rel_emb = dm.lookup ? nullptr : h->params["rel_emb"].ptr;
  const float* cw = dm.gen_conv ? h->conv_w_rel : h->params["conv1_weights"].ptr;
  const float* cb = dm.gen_conv ? h->conv_b_rel : h->params["conv1_bias"].ptr;
  constexpr int QPB = 4;
  size_t lds = sizeof(float) * (size_t)QPB * dm.in_h * dm.in_w;
  unsigned short* xh = (unsigned short*)h->x_sorted;
  unsigned short* xl = xh + (size_t)h->ws_queries * dm.F_pad;
  ScopedKernelTimer t(h, "conv", s);
  int64_t cap_small = (dm.gen_fc ? dm.R : 1) + 1;
  int64_t n_small_max = cap_small - 1 < B ? cap_small - 1 : B;
  int64_t grid = skip_big ? n_small_max * (32 / QPB) : (B + QPB - 1) / QPB;
  hipLaunchKernelGGL((k_conv3x3_bn_relu_bf16<QPB>), dim3((unsigned)grid), dim3(256), lds, s, e1, rel,
                     e1_rows, h->perm, h->params["ent_emb"].ptr, (int64_t)h->cfg.shard_lo, dm.n_local, rel_emb, cw, cb,
                     dm.gen_conv ? 1 : 0, h->conv_scale, h->conv_shift, dm.d, dm.r, dm.in_h, dm.in_w,
                     dm.stacked ? 1 : 0, dm.C, dm.Ho, dm.Wo, dm.concat_rel ? 1 : 0, dm.F, dm.F_pad, dm.R, B, h->tiles,
                     h->n_tiles, skip_big ? 1 : 0, xh, xl, h->x_exp);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// dense, tiles of <= 32 queries: one wave per K slice, weights streamed from the planes into VGPRs.
// ------------------------------------------------------------------------------------------------
template <int NFB>
__global__ __launch_bounds__(256) void k_dense_small_bf16x3(const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                            const unsigned short* __restrict__ x_hi,
                                                            const unsigned short* __restrict__ x_lo,
                                                            const int32_t* __restrict__ tiles,
                                                            const int32_t* __restrict__ n_tiles, int nfb, int64_t ks32n,
                                                            int64_t F_pad, int nslices, int64_t Bcap, int d_pad16,
                                                            float* __restrict__ z_part) {
  constexpr int NQ = 2;
  int tile = blockIdx.x;
  if (tile >= n_tiles[0]) return;
  const int fb0 = blockIdx.z * NFB;
  const int64_t relw = __builtin_amdgcn_readfirstlane(tiles[4 * tile + 0]);
  const int start = __builtin_amdgcn_readfirstlane(tiles[4 * tile + 1]);
  const int n = __builtin_amdgcn_readfirstlane(tiles[4 * tile + 2]);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int slice = blockIdx.y * 4 + wave;
  if (slice >= nslices) return;
  const int64_t kb = ks32n * slice / nslices, ke = ks32n * (slice + 1) / nslices;
  f32x4 acc[NFB][NQ];
#pragma unroll
  for (int a = 0; a < NFB; ++a)
#pragma unroll
    for (int b = 0; b < NQ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int64_t woff[NFB];
#pragma unroll
  for (int a = 0; a < NFB; ++a) {
    int fb = fb0 + a < nfb ? fb0 + a : nfb - 1;
    woff[a] = ((relw * nfb + fb) * w16_ks_stride_n(ks32n)) * 64 + lane;
  }
  int64_t xoff[NQ];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    int qi = b * 16 + (lane & 15);
    if (qi > n - 1) qi = n - 1;
    xoff[b] = (int64_t)(start + qi) * F_pad + 8 * (lane >> 4);
  }
  const bool second = n > 16;
  for (int64_t ks = kb; ks < ke; ++ks) {
    uint4 ah[NFB], al[NFB], bh[NQ], bl[NQ];
#pragma unroll
    for (int a = 0; a < NFB; ++a) { ah[a] = Whi[woff[a] + ks * 64]; al[a] = Wlo[woff[a] + ks * 64]; }
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      bh[b] = *(const uint4*)(x_hi + xoff[b] + 32 * ks);
      bl[b] = *(const uint4*)(x_lo + xoff[b] + 32 * ks);
    }
#pragma unroll
    for (int a = 0; a < NFB; ++a) {
      MFMA16_X3(ah[a], al[a], bh[0], bl[0], acc[a][0]);
      if (second) MFMA16_X3(ah[a], al[a], bh[1], bl[1], acc[a][1]);
    }
  }
#pragma unroll
  for (int a = 0; a < NFB; ++a) {
    if (fb0 + a >= nfb) continue;
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      int qi = b * 16 + (lane & 15);
      if (qi < n) {
        float* dst = z_part + ((int64_t)slice * Bcap + start + qi) * d_pad16 + (fb0 + a) * 16 + 4 * (lane >> 4);
        *(float4*)dst = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dense, tiles of 33..128 queries: 4 waves share the weight and x planes through an LDS ring filled by
// LDS-DMA (same item-balanced structure as dense_big_body in kernels_encode.hip).
// ------------------------------------------------------------------------------------------------
template <int NFB, int NB>
__device__ __forceinline__ void dense_big_body_bf16(uint4* __restrict__ ring, const uint4* __restrict__ Whi,
                                                    const uint4* __restrict__ Wlo,
                                                    const unsigned short* __restrict__ x_hi,
                                                    const unsigned short* __restrict__ x_lo, int64_t relw, int start,
                                                    int n, int fb0, int nfb, int64_t ks32n, int64_t F_pad, int64_t kb,
                                                    int64_t ke, float* __restrict__ zdst, int d_pad16) {
  constexpr int NSLOT = 2 * NFB + 2 * NB;  // W hi | W lo | x hi | x lo
  constexpr int L = (NSLOT + 3) / 4;
  constexpr int STAGE = 4 * L * 64;        // uint4 per ring stage
  constexpr int NSTAGE = 3;
  constexpr int T = NFB * NB;
  constexpr int MAXI = (T + 3) / 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i0 = T * wave / 4, cnt = T * (wave + 1) / 4 - i0;
  const char* src[L];
  int stride[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    int sl = wave + 4 * i;
    if (sl >= NSLOT) sl = wave;
    if (sl < 2 * NFB) {
      int plane = sl >= NFB;
      int fbi = sl - plane * NFB;
      int fb = fb0 + fbi < nfb ? fb0 + fbi : nfb - 1;
      const uint4* base = plane ? Wlo : Whi;
      src[i] = (const char*)(base + ((relw * nfb + fb) * w16_ks_stride_n(ks32n) + kb) * 64 + lane);
      stride[i] = 1024;
    } else {
      int xs = sl - 2 * NFB;
      int plane = xs >= NB;
      int qb = xs - plane * NB;
      int qi = qb * 16 + (lane & 15);
      if (qi > n - 1) qi = n - 1;
      const unsigned short* base = plane ? x_lo : x_hi;
      src[i] = (const char*)(base + (int64_t)(start + qi) * F_pad + 32 * kb + 8 * (lane >> 4));
      stride[i] = 64;
    }
  }
#ifndef COPER_W_AUX
#define COPER_W_AUX 2
#endif
  // weight fragments are read exactly once per pass: aux = 2 issues them non-temporal (A/B on MI355X: -4 %)
#define STAGE_ISSUE(buf, kk)                                                                                      \
  {                                                                                                               \
    uint4* dstb = ring + (buf)*STAGE;                                                                             \
    _Pragma("unroll") for (int i = 0; i < L; ++i) {                                                               \
      if (wave + 4 * i < 2 * NFB)                                                                                 \
        __builtin_amdgcn_global_load_lds(                                                                         \
            (const __attribute__((address_space(1))) void*)(src[i] + (int64_t)(kk)*stride[i]),                    \
            (__attribute__((address_space(3))) void*)(dstb + (wave + 4 * i) * 64), 16, 0, COPER_W_AUX);           \
      else                                                                                                        \
        __builtin_amdgcn_global_load_lds(                                                                         \
            (const __attribute__((address_space(1))) void*)(src[i] + (int64_t)(kk)*stride[i]),                    \
            (__attribute__((address_space(3))) void*)(dstb + (wave + 4 * i) * 64), 16, 0, 0);                     \
    }                                                                                                             \
  }
  int offA[MAXI], offB[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    int item = cnt > 0 ? i0 + (i < cnt ? i : cnt - 1) : 0;
    offA[i] = (item / NB) * 64 + lane;                 // W hi slot; W lo = + NFB*64
    offB[i] = (2 * NFB + item % NB) * 64 + lane;       // x hi slot; x lo = + NB*64
  }
  f32x4 acc[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = (int)(ke - kb);
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) STAGE_ISSUE(st, st < nk ? st : nk - 1);
  int cur = 0, nxt = NSTAGE - 1;
  for (int k = 0; k < nk; ++k) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * L) : "memory");  // stage k landed; younger stages fly
    __builtin_amdgcn_s_barrier();
    {
      int kk = k + NSTAGE - 1 < nk ? k + NSTAGE - 1 : nk - 1;
      STAGE_ISSUE(nxt, kk);
    }
    const uint4* ab = ring + cur * STAGE;
    constexpr int GI = 4;
#pragma unroll
    for (int g = 0; g < MAXI; g += GI) {
      uint4 ah[GI], al[GI], bh[GI], bl[GI];
#pragma unroll
      for (int j = 0; j < GI; ++j)
        if (g + j < MAXI) {
          ah[j] = ab[offA[g + j]];
          al[j] = ab[offA[g + j] + NFB * 64];
          bh[j] = ab[offB[g + j]];
          bl[j] = ab[offB[g + j] + NB * 64];
        }
#pragma unroll
      for (int j = 0; j < GI; ++j)
        if (g + j < MAXI) acc[g + j] = MFMA16_BF16(al[j], bh[j], acc[g + j]);
#pragma unroll
      for (int j = 0; j < GI; ++j)
        if (g + j < MAXI) acc[g + j] = MFMA16_BF16(ah[j], bl[j], acc[g + j]);
#pragma unroll
      for (int j = 0; j < GI; ++j)
        if (g + j < MAXI) acc[g + j] = MFMA16_BF16(ah[j], bh[j], acc[g + j]);
    }
    cur = cur == NSTAGE - 1 ? 0 : cur + 1;
    nxt = nxt == NSTAGE - 1 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef STAGE_ISSUE
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i >= cnt) continue;
    int item = i0 + i;
    int fb = fb0 + item / NB, qb = item % NB;
    int qi = qb * 16 + (lane & 15);
    if (fb < nfb && qi < n) {
      float* dst = zdst + (int64_t)(start + qi) * d_pad16 + fb * 16 + 4 * (lane >> 4);
      *(float4*)dst = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
  }
}

template <int NFB>
__global__ __launch_bounds__(256) void k_dense_big_bf16x3(const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                          const unsigned short* __restrict__ x_hi,
                                                          const unsigned short* __restrict__ x_lo,
                                                          const int32_t* __restrict__ tiles,
                                                          const int32_t* __restrict__ n_tiles, int64_t cap_small,
                                                          int nfb, int64_t ks32n, int64_t F_pad, int nslices,
                                                          int64_t Bcap, int d_pad16, float* __restrict__ z_part) {
  extern __shared__ uint4 ring16[];
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
#define BODY(NB_) dense_big_body_bf16<NFB, NB_>(ring16, Whi, Wlo, x_hi, x_lo, relw, start, n, fb0, nfb, ks32n, F_pad, kb, ke, zdst, d_pad16)
  switch (nb) {
    case 3: BODY(3); break;
    case 4: BODY(4); break;
    case 5: BODY(5); break;
    case 6: BODY(6); break;
    case 7: BODY(7); break;
    default: BODY(8); break;
  }
#undef BODY
}


// ------------------------------------------------------------------------------------------------
// dense, tiles of 33..128 queries, weights streamed to registers.  Each of the 4 waves OWNS feature
// blocks fb = wave, wave+4, ... and reads their W fragments straight from the fragment image into VGPRs
// (1 KiB coalesced per fragment, non-temporal: every weight byte is used once per pass), PF k-steps
// ahead; only the x planes, which all waves need, go through an LDS ring filled by LDS-DMA.  Against the
// all-through-LDS ring this keeps (PF-1) whole k-steps of weights in flight per CU instead of 2, and LDS
// reads drop from 4 fragments per MFMA triple to the x fragments once per wave.
// Accumulation order per (query, feature, slice) is unchanged: k ascending, lo*hi, hi*lo, hi*hi.
// ------------------------------------------------------------------------------------------------
#ifndef COPER_DENSE_PF
#define COPER_DENSE_PF 4
#endif
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// VMEM ops issued after x-stage k when r = nk-1-k steps remain (r < P-1): r weight groups, min(P-2, r) x groups
template <int P, int LX, int W2, int RR>
__device__ __forceinline__ void vm_wait_tail(int r) {
  if constexpr (RR >= 0) {
    if (r == RR) vm_wait<RR * W2 + (RR < P - 2 ? RR : P - 2) * LX>();
    else vm_wait_tail<P, LX, W2, RR - 1>(r);
  }
}

template <int NB, int NOWN>
__device__ __forceinline__ void dense_reg_body_bf16(uint4* __restrict__ ring, const uint4* __restrict__ Whi,
                                                    const uint4* __restrict__ Wlo,
                                                    const unsigned short* __restrict__ x_hi,
                                                    const unsigned short* __restrict__ x_lo, int64_t relw, int start,
                                                    int n, int fb0, int nfb, int64_t ks32n, int64_t F_pad, int64_t kb,
                                                    int64_t ke, float* __restrict__ zdst, int d_pad16, int wave) {
  constexpr int P = COPER_DENSE_PF;          // weight prefetch depth = x ring stages
  constexpr int LX = (2 * NB + 3) / 4;       // x DMA instructions per wave per k-step
  constexpr int XSTAGE = 4 * LX * 64;        // uint4 per ring stage: x hi [NB] | x lo [NB] (| padding)
  constexpr int W2 = 2 * NOWN;
  static_assert((P - 1) * W2 + (P - 2) * LX <= 63, "vmcnt is 6 bits");
  const int lane = threadIdx.x & 63;
  const uint4* wp[NOWN][2];
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    int fb = fb0 + wave + 4 * j;
    if (fb > nfb - 1) fb = nfb - 1;
    int64_t o = ((relw * nfb + fb) * w16_ks_stride_n(ks32n) + kb) * 64 + lane;
    wp[j][0] = Whi + o;
    wp[j][1] = Wlo + o;
  }
  const char* xsrc[LX];
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    int sl = wave + 4 * i;
    if (sl >= 2 * NB) sl = wave;
    int plane = sl >= NB;
    int qb = sl - plane * NB;
    int qi = qb * 16 + (lane & 15);
    if (qi > n - 1) qi = n - 1;
    xsrc[i] = (const char*)((plane ? x_lo : x_hi) + (int64_t)(start + qi) * F_pad + 32 * kb + 8 * (lane >> 4));
  }
#define X_ISSUE(buf, kk)                                                                                        \
  {                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < LX; ++i) __builtin_amdgcn_global_load_lds(                            \
        (const __attribute__((address_space(1))) void*)(xsrc[i] + (int64_t)(kk)*64),                            \
        (__attribute__((address_space(3))) void*)(ring + (buf)*XSTAGE + (wave + 4 * i) * 64), 16, 0, 0);        \
    asm volatile("" ::: "memory");                                                                              \
  }
#define W_ISSUE(s, kk)                                                                                          \
  {                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NOWN; ++j) {                                                          \
      W[s][j][0] = __builtin_nontemporal_load((const u32x4*)(wp[j][0] + (int64_t)(kk)*64));                     \
      W[s][j][1] = __builtin_nontemporal_load((const u32x4*)(wp[j][1] + (int64_t)(kk)*64));                     \
    }                                                                                                           \
    asm volatile("" ::: "memory");                                                                              \
  }
  u32x4 W[P][NOWN][2];
  f32x4 acc[NOWN][NB];
#pragma unroll
  for (int j = 0; j < NOWN; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = (int)(ke - kb);
  // issue order is the steady-state one:  W(0) | x(0) W(1) | x(1) W(2) | ... so the counted waits below hold
  // from the first step on; steps past the end are not issued (the tail waits count what is really there)
  W_ISSUE(0, 0);
#pragma unroll
  for (int t = 0; t < P - 1; ++t) {
#ifndef COPER_DBG_REG_NO_X
    if (t < nk) X_ISSUE(t, t);
#endif
    if (t + 1 < nk) W_ISSUE(t + 1, t + 1);
  }
  for (int k0 = 0; k0 < nk; k0 += P) {
#pragma unroll
    for (int s = 0; s < P; ++s) {
      const int k = k0 + s;
      if (k < nk) {
        const int r = nk - 1 - k;
#ifndef COPER_DBG_REG_NO_X
        if (r >= P - 1) vm_wait<(P - 1) * W2 + (P - 2) * LX>();   // x stage k landed, everything younger flies
        else vm_wait_tail<P, LX, W2, P - 2>(r);
#endif
#ifndef COPER_DBG_REG_NO_BAR
        __builtin_amdgcn_s_barrier();
#endif
#ifndef COPER_DBG_REG_NO_X
        if (k + P - 1 < nk) X_ISSUE((s + P - 1) % P, k + P - 1);
#endif
        const uint4* xb = ring + s * XSTAGE + lane;
#ifdef COPER_DBG_REG_NO_MFMA
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
          acc[j][0][0] += __uint_as_float(W[s][j][0][0] ^ W[s][j][1][1]);
          acc[j][0][1] += __uint_as_float(W[s][j][0][2] ^ W[s][j][1][3]);
        }
        acc[0][0][2] += __uint_as_float(xb[0].x);
#else
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          uint4 bh = xb[q * 64], bl = xb[(NB + q) * 64];
#pragma unroll
          for (int j = 0; j < NOWN; ++j) MFMA16_X3(W[s][j][0], W[s][j][1], bh, bl, acc[j][q]);
        }
#endif
        if (k + P < nk) W_ISSUE(s, k + P);
      }
    }
  }
#undef X_ISSUE
#undef W_ISSUE
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    int fb = fb0 + wave + 4 * j;
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
__global__ __launch_bounds__(256) void k_dense_reg_bf16x3(const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                                          const unsigned short* __restrict__ x_hi,
                                                          const unsigned short* __restrict__ x_lo,
                                                          const int32_t* __restrict__ tiles,
                                                          const int32_t* __restrict__ n_tiles, int64_t cap_small,
                                                          int nfb, int64_t ks32n, int64_t F_pad, int nslices,
                                                          int64_t Bcap, int d_pad16, float* __restrict__ z_part) {
  extern __shared__ uint4 ring16[];
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
  constexpr int OWN_HI = (NFB + 3) / 4, N_HI = NFB - 4 * (OWN_HI - 1);   // waves < N_HI own OWN_HI blocks
#define BODY(NB_)                                                                                                       \
  if (wave < N_HI)                                                                                                      \
    dense_reg_body_bf16<NB_, OWN_HI>(ring16, Whi, Wlo, x_hi, x_lo, relw, start, n, fb0, nfb, ks32n, F_pad, kb, ke, zdst, \
                                     d_pad16, wave);                                                                    \
  else                                                                                                                  \
    dense_reg_body_bf16<NB_, (OWN_HI > 1 ? OWN_HI - 1 : 1)>(ring16, Whi, Wlo, x_hi, x_lo, relw, start, n, fb0, nfb,     \
                                                            ks32n, F_pad, kb, ke, zdst, d_pad16, wave);
  switch (nb) {
    case 3: BODY(3); break;
    case 4: BODY(4); break;
    case 5: BODY(5); break;
    case 6: BODY(6); break;
    case 7: BODY(7); break;
    default: BODY(8); break;
  }
#undef BODY
}

template <int NFB>
static void dense_launch_bf16(coper_handle* h, int64_t B, int nslices, int zgroups, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t cap_small = (dm.gen_fc ? dm.R : 1) + 1;
  int64_t n_small_max = cap_small - 1 < B ? cap_small - 1 : B;
  int64_t n_big_max = B / 33 + 1;
  const unsigned short* xh = (const unsigned short*)h->x_sorted;
  const unsigned short* xl = xh + (size_t)h->ws_queries * dm.F_pad;
  if (n_small_max > 0)
    hipLaunchKernelGGL((k_dense_small_bf16x3<NFB>), dim3((unsigned)n_small_max, (unsigned)((nslices + 3) / 4), (unsigned)zgroups),
                       dim3(256), 0, s, (const uint4*)h->Wf16_hi, (const uint4*)h->Wf16_lo, xh, xl, h->tiles, h->n_tiles,
                       dm.nfb, dm.F_pad / 32, dm.F_pad, nslices, h->ws_queries, dm.d_pad16, h->z_part);
  if (h->dense_small_only) return;
#ifndef COPER_DENSE_RING
  if (B > 32 && NFB >= 8) {
    // x ring only: P stages of (x hi | x lo) for up to 8 query blocks
    size_t lds = (size_t)COPER_DENSE_PF * 16 * 64 * sizeof(uint4);
    hipLaunchKernelGGL((k_dense_reg_bf16x3<NFB>), dim3((unsigned)n_big_max, (unsigned)nslices, (unsigned)zgroups), dim3(256),
                       lds, s, (const uint4*)h->Wf16_hi, (const uint4*)h->Wf16_lo, xh, xl, h->tiles, h->n_tiles, cap_small,
                       dm.nfb, dm.F_pad / 32, dm.F_pad, nslices, h->ws_queries, dm.d_pad16, h->z_part);
    return;
  }
#endif
  if (B > 32) {
    size_t lds = (size_t)3 * (((2 * NFB + 16 + 3) / 4) * 4) * 64 * sizeof(uint4);
    if (!h->dense_attr_done) {
      (void)hipFuncSetAttribute((const void*)k_dense_big_bf16x3<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      h->dense_attr_done = true;
    }
    hipLaunchKernelGGL((k_dense_big_bf16x3<NFB>), dim3((unsigned)n_big_max, (unsigned)nslices, (unsigned)zgroups), dim3(256),
                       lds, s, (const uint4*)h->Wf16_hi, (const uint4*)h->Wf16_lo, xh, xl, h->tiles, h->n_tiles, cap_small,
                       dm.nfb, dm.F_pad / 32, dm.F_pad, nslices, h->ws_queries, dm.d_pad16, h->z_part);
  }
}

int launch_dense_bf16(coper_handle* h, int64_t B, int nslices, bool small_only, hipStream_t s) {
  const Dims& dm = h->dm;
  h->dense_small_only = small_only;
  int nfb = dm.nfb;
  if (nfb == 13) dense_launch_bf16<13>(h, B, nslices, 1, s);
  else if (nfb <= 2) dense_launch_bf16<2>(h, B, nslices, 1, s);
  else if (nfb <= 4) dense_launch_bf16<4>(h, B, nslices, 1, s);
  else dense_launch_bf16<8>(h, B, nslices, (nfb + 7) / 8, s);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// finalize + pack (coper_encode_rank without the fused tail): the h rows (k_finalize_h_publish, kernels_tail_bf16.hip: sum of the
// K slices + dense bias + folded FCBN + ReLU, and the batch's exponent), then the query planes from those rows.  (Rounds 2 - 3
// wrote the planes from the partial sums in one launch; the planes hold h 2^e_h now and e_h is known only when every row is.)
// The rank counters of the pass are preset by the packing launch.
// ------------------------------------------------------------------------------------------------
int launch_dense_finalize_pack(coper_handle* h, int64_t B, int ksplit, float* h_out, int32_t* cnt, int32_t cnt_base,
                               int32_t* cnt_eq, hipStream_t s) {
  int rc = launch_finalize_h_publish(h, B, ksplit, h_out, s);
  if (rc) return rc;
  const int32_t base_was = h->count_base;
  h->preset_cnt = cnt;
  h->preset_eq = cnt_eq;
  h->count_base = cnt_base;
  const int64_t n_blk = (B + 127) / 128 * 4;
  rc = launch_rows_to_frag_bf16(h, h_out, B, n_blk, (uint4*)h->hfrag16_hi, (uint4*)h->hfrag16_lo, (uint4*)h->hrm16_hi, (uint4*)h->hrm16_lo,
                                (uint4*)h->hf3_ws, true, s);
  h->count_base = base_was;
  return rc;
}

}  // namespace coper
