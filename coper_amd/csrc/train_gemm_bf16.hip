// The three GEMMs of the generated dense layer in the training step (coper_train.hip), on the bf16 matrix cores with
// split operands -- the arithmetic of the inference path's bf16x3 mode: every fp32 operand x = hi + lo (two bf16
// terms), every product lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative per
// product; the training parity tests hold gradients to 2e-4).  gfx950's exact-f32 MFMA runs at 1/16 of the bf16 rate,
// so three bf16 MFMAs per product are 5.3x the matrix throughput of an fp32 GEMM.
//
//   forward   T[rho][b][k]  = sum_f       x[b][f]      * P[rho][f][k]        (models.py:70,412 in factored form)
//   dP        dP[rho][f][k] = sum_b       x[b][f]      * dT[rho][b][k]       dT[rho][b][k] = ctx[b][rho] dz[b][k]
//   dx        dx[b][f]      = sum_(rho,k) dT[rho][b][k] * P[rho][f][k]       (the [B, r*F] intermediate dA = dz P2^T of the
//                                                                             library-GEMM version is never formed)
// Each is C(i, j) = sum_k X(i, k) Y(j, k): both operands are first packed into the fragment-major hi / lo planes of the
// inference path (k_pack_frag: any two-level strided view of an fp32 tensor -> [row block of 32][k-step][64 lanes] x 16 B,
// transposing through LDS so that the reads follow the contiguous dimension), then k_gemm_nt_bf16x3 streams fragments
// straight into the registers the MFMAs consume and stores C through a two-level strided view.
// Round 5: the split is fp16 with one exact power of two per packed operand (split16.h; round 2's was bf16):
// the planes hold X 2^e_X with max |X| 2^e_X in [2^14, 2^15), the epilogue multiplies by 2^-(e_X + e_Y).
#include "coper_internal.h"
#include "split16.h"
#include "train_gemm.h"

namespace coper {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TG_MFMA(a, b, c) S16_MFMA32(a, b, c)

__device__ __forceinline__ int64_t tg_off(const TgIdx& a, int64_t i) {
  return a.seg > 0 ? (i / a.seg) * a.s_hi + (i % a.seg) * a.s_lo : i * a.s_lo;
}

__device__ __forceinline__ void tg_split8(const float* v, int e, uint4& hi, uint4& lo) {
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = x3_scale(v[j], e);
  split8_s16(w, hi, lo);
}

// The largest |X| of an operand, as an exponent: ONE launch.  Every view tg_pack is given is a permutation of a dense tensor, so
// the maximum is taken over the `n` contiguous floats behind it (16-byte loads: 118 MB of projections in ~25 us); the last block
// to finish (ticket in scratch[1]) turns the bits into the power of two, stores it for the pack and the GEMM, and leaves the two
// scratch words zero for the next call.
__global__ __launch_bounds__(256) void k_tg_absmax_exp(const float* __restrict__ src, int64_t n, unsigned* __restrict__ scratch,
                                                       int32_t* __restrict__ exp_out) {
  unsigned m = 0u;
  const int64_t n4 = (((uintptr_t)src) & 15) == 0 ? n >> 2 : 0;
  // eight independent 16-byte loads in flight per thread (round 6: the loop carried ONE -- 118 MB of projections took 59 us,
  // 2 TB/s, and the 13 - 30 MB operands 18 - 32 us each: 137 us of a 1.25 ms step went into five of these launches)
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; e + 7 * stride < n4; e += 8 * stride) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ((const uint4*)src)[e + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned a = v[u].x & 0x7fffffffu, b = v[u].y & 0x7fffffffu, c = v[u].z & 0x7fffffffu, d = v[u].w & 0x7fffffffu;
      const unsigned ab = a > b ? a : b, cd = c > d ? c : d, x = ab > cd ? ab : cd;
      m = x > m ? x : m;
    }
  }
  for (; e < n4; e += stride) {
    const uint4 v = ((const uint4*)src)[e];
    const unsigned a = v.x & 0x7fffffffu, b = v.y & 0x7fffffffu, c = v.z & 0x7fffffffu, d = v.w & 0x7fffffffu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d, x = ab > cd ? ab : cd;
    m = x > m ? x : m;
  }
  for (int64_t e1 = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; e1 < n; e1 += stride) {
    const unsigned b = __float_as_uint(src[e1]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  __shared__ unsigned sm[4];
  __shared__ int s_last;
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned w = sm[0] > sm[1] ? sm[0] : sm[1], w2 = sm[2] > sm[3] ? sm[2] : sm[3];
    w = w > w2 ? w : w2;
    if (w) atomicMax(scratch, w);
    __threadfence();
    s_last = atomicAdd(scratch + 1, 1u) == gridDim.x - 1 ? 1 : 0;
    if (s_last) {
      const unsigned all = atomicExch(scratch, 0u);     // (memory-side: coherent whatever XCD added)
      *exp_out = x3_exp_for_bits(all);
      scratch[1] = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pack: X(row, k) = src[off(ri, row) + off(ki, k)], rows [0, R), k [0, K)  ->  planes[(blk * KS16 + ks) * 64 + l] =
// 8 bf16 { X[32 blk + (l & 31)][16 ks + 8 (l >> 5) + j] } (zero beyond R / K); grid (ceil(KS16 / 8), row blocks).
// MODE 0: consecutive threads read consecutive k (k contiguous in memory); 1: consecutive rows; +2: four elements per
// thread along that direction as one 16-byte load (alignment checked by the host).
// ------------------------------------------------------------------------------------------------
template <int MODE, int RT>
__global__ __launch_bounds__(256) void k_pack_frag(const float* __restrict__ src, TgIdx ri, TgIdx ki, int64_t R, int64_t K,
                                                   int KS16, int KST, uint4* __restrict__ hi, uint4* __restrict__ lo,
                                                   const int32_t* __restrict__ exp_dev, int32_t* __restrict__ exp_copy,
                                                   const unsigned* __restrict__ max_slots) {
  // tile of RT rows x KT k (4096 elements): 32 x 128 when k is the contiguous direction, 128 x 32 when rows are (a
  // wave then reads 512 contiguous bytes per k instead of 128: the [rho][f][k] -> rows (rho, k) view of the projection
  // went from 70 to 5x us)
  constexpr int KT = 4096 / RT;
  __shared__ float tile[RT][KT + 1];
  const int64_t row0 = (int64_t)blockIdx.y * RT, k0 = (int64_t)blockIdx.x * KT;
  const int t = threadIdx.x;
  if (MODE == 0) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i, row = e / KT, kk = e % KT;
      const bool ok = row0 + row < R && k0 + kk < K;
      tile[row][kk] = ok ? src[tg_off(ri, row0 + row) + tg_off(ki, k0 + kk)] : 0.f;
    }
  } else if (MODE == 1) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i, kk = e / RT, row = e % RT;
      const bool ok = row0 + row < R && k0 + kk < K;
      tile[row][kk] = ok ? src[tg_off(ri, row0 + row) + tg_off(ki, k0 + kk)] : 0.f;
    }
  } else if (MODE == 2) {   // four consecutive k per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t + 256 * i, row = e / (KT / 4), kk = (e % (KT / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < R && k0 + kk < K) v = *(const float4*)(src + tg_off(ri, row0 + row) + tg_off(ki, k0 + kk));   // K % 4 == 0
      tile[row][kk] = v.x; tile[row][kk + 1] = v.y; tile[row][kk + 2] = v.z; tile[row][kk + 3] = v.w;
    }
  } else {                  // four consecutive rows per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t + 256 * i, kk = e / (RT / 4), row = (e % (RT / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < R && k0 + kk < K) v = *(const float4*)(src + tg_off(ri, row0 + row) + tg_off(ki, k0 + kk));   // R % 4 == 0
      tile[row][kk] = v.x; tile[row + 1][kk] = v.y; tile[row + 2][kk] = v.z; tile[row + 3][kk] = v.w;
    }
  }
  // the operand's power of two: a word another pack left, or -- max_slots -- the maximum its producer left in TG_MAX_SLOTS slots
  __shared__ unsigned s_mx[4];
  if (max_slots) {
    unsigned mx = 0u;
    for (int i = t; i < TG_MAX_SLOTS; i += 256) { const unsigned v = max_slots[i]; mx = v > mx ? v : mx; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned u = __shfl_xor(mx, o, 64); mx = u > mx ? u : mx; }
    if ((t & 63) == 0) s_mx[t >> 6] = mx;
  }
  __syncthreads();
  int pe;
  if (max_slots) {
    const unsigned a = s_mx[0] > s_mx[1] ? s_mx[0] : s_mx[1], b = s_mx[2] > s_mx[3] ? s_mx[2] : s_mx[3];
    pe = x3_exp_for_bits(a > b ? a : b);
  } else {
    pe = *exp_dev;
  }
  if (exp_copy && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *exp_copy = pe;    // (a plane set packed with another set's exponent)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int f = t + 256 * p, blk = f >> 6, l = f & 63;          // 8 (row block, k-step) blocks of 64 lanes
    const int rb = blk / (KT / 16), ksl = blk % (KT / 16);
    const int64_t ks = (int64_t)blockIdx.x * (KT / 16) + ksl;
    if (ks >= KS16) continue;
    const int row = 32 * rb + (l & 31), kb = 16 * ksl + 8 * (l >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[row][kb + j];
    uint4 h4, l4;
    tg_split8(v, pe, h4, l4);
    const int64_t o = (((int64_t)blockIdx.y * (RT / 32) + rb) * KST + ks) * 64 + l;
    hi[o] = h4;
    lo[o] = l4;
  }
}

static bool tg_vec_ok(const float* src, const TgIdx& fast, const TgIdx& slow, int64_t n_fast) {
  // 16-byte loads along the fast index: unit stride, segments and extents in multiples of 4, every other stride too
  if (fast.s_lo != 1 || (n_fast & 3) || ((uintptr_t)src & 15)) return false;
  if (fast.seg > 0 && ((fast.seg & 3) || (fast.s_hi & 3))) return false;
  if (slow.s_lo & 3) return false;
  if (slow.seg > 0 && (slow.s_hi & 3)) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Both views of ONE tensor from one read (round 6: the generated dense layer's projection P is an operand of two of its three
// GEMMs, contracted over a different index in each -- 118 MB read by two pack launches, 64 + 60 us of a 1.2 ms step):
//   view A: rows i (ri), contraction k (ki)  -> planes A [ceil(R / 32)][KST_A][64]        (what k_pack_frag<2, 32> writes)
//   view B: rows k,      contraction i       -> planes B [ceil(K / 32) padded][KST_B][64]  (the transposed operand)
// A workgroup loads a 32 x 128 tile (32 rows i, 128 consecutive k, four per thread as one 16-byte load) into LDS and emits the
// eight (row block, k-step) fragment blocks of view A and the eight (4 row blocks of k) x (2 k-steps of i) blocks of view B.
// ------------------------------------------------------------------------------------------------
// Round 6: a workgroup takes PB_TPW row blocks one after the other (the next block's loads are requested before this one's fragments
// are formed; the reduction of the 1,024 maximum slots -- 4 KB per workgroup, a shuffle sum and a barrier -- once per workgroup).
constexpr int PB_TPW = 4;
__global__ __launch_bounds__(256) void k_pack_frag_both(const float* __restrict__ src, TgIdx ri, TgIdx ki, int64_t R, int64_t K,
                                                        int KS16_A, int KST_A, uint4* __restrict__ ahi, uint4* __restrict__ alo,
                                                        int KS16_B, int KST_B, uint4* __restrict__ bhi, uint4* __restrict__ blo,
                                                        const int32_t* __restrict__ exp_dev, int32_t* __restrict__ exp_a,
                                                        int32_t* __restrict__ exp_b, const unsigned* __restrict__ max_slots, int n_rb) {
  constexpr int RT = 32, KT = 128;
  __shared__ float tile[2][RT][KT + 1];
  const int64_t k0 = (int64_t)blockIdx.x * KT;
  const int t = threadIdx.x;
  const int rb0 = blockIdx.y * PB_TPW, rb1 = rb0 + PB_TPW < n_rb ? rb0 + PB_TPW : n_rb;
  float4 v[4];
  auto fetch = [&](int rb) {
    const int64_t row0 = (int64_t)rb * RT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t + 256 * i, row = e / (KT / 4), kk = (e % (KT / 4)) * 4;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < R && k0 + kk < K) v[i] = *(const float4*)(src + tg_off(ri, row0 + row) + tg_off(ki, k0 + kk));   // (K % 4 == 0: host)
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t + 256 * i, row = e / (KT / 4), kk = (e % (KT / 4)) * 4;
      tile[buf][row][kk] = v[i].x; tile[buf][row][kk + 1] = v[i].y; tile[buf][row][kk + 2] = v[i].z; tile[buf][row][kk + 3] = v[i].w;
    }
  };
  fetch(rb0);
  __shared__ unsigned s_mx[4];
  if (max_slots) {
    unsigned mx = 0u;
    for (int i = t; i < TG_MAX_SLOTS; i += 256) { const unsigned w = max_slots[i]; mx = w > mx ? w : mx; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned u = __shfl_xor(mx, o, 64); mx = u > mx ? u : mx; }
    if ((t & 63) == 0) s_mx[t >> 6] = mx;
  }
  stash(0);
  __syncthreads();
  int pe;
  if (max_slots) {
    const unsigned a = s_mx[0] > s_mx[1] ? s_mx[0] : s_mx[1], b = s_mx[2] > s_mx[3] ? s_mx[2] : s_mx[3];
    pe = x3_exp_for_bits(a > b ? a : b);
  } else {
    pe = *exp_dev;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) { *exp_a = pe; *exp_b = pe; }
  for (int rb = rb0; rb < rb1; ++rb) {
    const int buf = (rb - rb0) & 1;
    if (rb + 1 < rb1) fetch(rb + 1);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int f = t + 256 * p, blk = f >> 6, l = f & 63;
      // view A: row block rb, k-steps 8 blockIdx.x + blk
      {
        const int64_t ks = (int64_t)blockIdx.x * (KT / 16) + blk;
        if (ks < KS16_A) {
          const int row = l & 31, kb = 16 * blk + 8 * (l >> 5);
          float w[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) w[j] = tile[buf][row][kb + j];
          uint4 h4, l4;
          tg_split8(w, pe, h4, l4);
          const int64_t o = ((int64_t)rb * KST_A + ks) * 64 + l;
          ahi[o] = h4;
          alo[o] = l4;
        }
      }
      // view B: rows are the k of this tile (four blocks of 32), the contraction runs over the tile's 32 rows i (two k-steps)
      {
        const int rbb = blk >> 1, ksl = blk & 1;
        const int64_t ks = (int64_t)rb * (RT / 16) + ksl;
        if (ks < KS16_B) {
          const int col = 32 * rbb + (l & 31), ib = 16 * ksl + 8 * (l >> 5);
          float w[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) w[j] = tile[buf][ib + j][col];
          uint4 h4, l4;
          tg_split8(w, pe, h4, l4);
          const int64_t o = (((int64_t)blockIdx.x * (KT / 32) + rbb) * KST_B + ks) * 64 + l;
          bhi[o] = h4;
          blo[o] = l4;
        }
      }
    }
    if (rb + 1 < rb1) {
      stash(buf ^ 1);      // (the other buffer: its readers finished before the barrier at the end of the previous trip)
      __syncthreads();
    }
  }
}

// view A = (ri rows, ki contraction) into `a`, view B = the transposed operand into `b` (rows K padded to TG_ROW_PAD by the caller's
// allocation: tg_plane_elems).  The exponent: max_slots, or one reduction pass (scratch); both plane sets get it.
int tg_pack_both(coper_handle* h, const float* src, TgIdx ri, TgIdx ki, int64_t R, int64_t K, TgPlanes a, TgPlanes b, hipStream_t s,
                 unsigned* scratch, const unsigned* max_slots) {
  if (!a.exp || !b.exp || !scratch) return fail(h, COPER_ESTATE, "tg_pack_both: plane set without an exponent word");
  if (!tg_vec_ok(src, ki, ri, K)) return fail(h, COPER_ESTATE, "tg_pack_both: the contraction index of view A must be 16-byte loadable");
  if (!max_slots) {
    int64_t nb = (R * K + 256 * 32 - 1) / (256 * 32);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_tg_absmax_exp, dim3((unsigned)nb), dim3(256), 0, s, src, R * K, scratch, a.exp);
  }
  const int KS16_A = (int)((K + 15) / 16), KST_A = (int)tg_ks_stride(K), KS16_B = (int)((R + 15) / 16), KST_B = (int)tg_ks_stride(R);
  // rows of view B beyond K (up to its TG_ROW_PAD padding) and k-steps beyond KS16 are never read by the GEMM's stores, but its
  // loads touch whole row blocks: the tile grid covers them (bounds-checked loads write zeros)
  const int n_rb = (int)(tg_rows_pad(R) / 32);
  dim3 grid((unsigned)(tg_rows_pad(K) / 128), (unsigned)((n_rb + PB_TPW - 1) / PB_TPW));
  hipLaunchKernelGGL(k_pack_frag_both, grid, dim3(256), 0, s, src, ri, ki, R, K, KS16_A, KST_A, a.hi, a.lo, KS16_B, KST_B, b.hi, b.lo, a.exp,
                     a.exp, b.exp, max_slots, n_rb);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}


int tg_pack(coper_handle* h, const float* src, TgIdx ri, TgIdx ki, int64_t R, int64_t K, int64_t R_pad, bool rows_fast, TgPlanes out,
            hipStream_t s, unsigned* scratch, const int32_t* exp_from, const unsigned* max_slots) {
  const int KS16 = (int)((K + 15) / 16), KST = (int)tg_ks_stride(K);
  if (!out.exp || !scratch) return fail(h, COPER_ESTATE, "tg_pack: plane set without an exponent word");
  if (!exp_from && !max_slots) {       // (scratch is zero between calls: the reduction's last block resets it)
    // (at most 512 workgroups: every one ends on two atomics on one line -- a thousand of them took longer than the reduction)
    int64_t nb = (R * K + 256 * 32 - 1) / (256 * 32);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_tg_absmax_exp, dim3((unsigned)nb), dim3(256), 0, s, src, R * K, scratch, out.exp);
  }
  const int32_t* pexp = exp_from ? exp_from : out.exp;
  int32_t* pcopy = (exp_from || max_slots) ? out.exp : nullptr;
  if (exp_from) max_slots = nullptr;
  const bool vec = rows_fast ? tg_vec_ok(src, ri, ki, R) : tg_vec_ok(src, ki, ri, K);
  if (!rows_fast) {
    dim3 grid((unsigned)((KS16 + 7) / 8), (unsigned)(R_pad / 32));
    if (vec) hipLaunchKernelGGL((k_pack_frag<2, 32>), grid, dim3(256), 0, s, src, ri, ki, R, K, KS16, KST, out.hi, out.lo, pexp, pcopy, max_slots);
    else hipLaunchKernelGGL((k_pack_frag<0, 32>), grid, dim3(256), 0, s, src, ri, ki, R, K, KS16, KST, out.hi, out.lo, pexp, pcopy, max_slots);
  } else {
    dim3 grid((unsigned)((KS16 + 1) / 2), (unsigned)(R_pad / 128));   // R_pad is a multiple of TG_ROW_PAD = 128
    if (vec) hipLaunchKernelGGL((k_pack_frag<3, 128>), grid, dim3(256), 0, s, src, ri, ki, R, K, KS16, KST, out.hi, out.lo, pexp, pcopy, max_slots);
    else hipLaunchKernelGGL((k_pack_frag<1, 128>), grid, dim3(256), 0, s, src, ri, ki, R, K, KS16, KST, out.hi, out.lo, pexp, pcopy, max_slots);
  }
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// C(i, j) = sum_k X(i, k) Y(j, k).  A workgroup is 4 waves (2 x 2), a wave owns TI x TJ blocks of 32 x 32: X and Y
// fragments come straight from the planes into registers, two k-steps ahead (three register buffers); the i tiles of one
// j tile are neighbours in the grid, so the Y fragments they share are read from HBM once.
// Measured (MI355X, 30 GFLOP shapes of the FB15k-237 step): 140-145 us per GEMM -- bound by the CU's L2 -> L1 fill rate
// (~33 B/clk per CU for 683 B of fragments per MFMA); staging a k-step's 16 fragments in LDS by LDS-DMA (half the fill
// traffic, one barrier per 12-MFMA k-step) measured 245 us: the barrier per k-step costs more than the duplicates.
// Accumulator (i, j) layout of v_mfma_f32_32x32x16_bf16 with X as A and Y as B: register r of lane l is
// i = (r & 3) + 8 (r >> 2) + 4 (l >> 5), j = l & 31 -> one register of a wave is 32 consecutive j of two rows i.
// ------------------------------------------------------------------------------------------------
template <int TI, int TJ, int NBUF>
__global__ __launch_bounds__(256) void k_gemm_nt_bf16x3(const uint4* __restrict__ Xhi, const uint4* __restrict__ Xlo,
                                                        const uint4* __restrict__ Yhi, const uint4* __restrict__ Ylo, int KS16_all, int KST,
                                                        float* __restrict__ C, TgIdx ci, TgIdx cj, int64_t M, int64_t N, int nsplit,
                                                        float* __restrict__ part, double* __restrict__ sumsq, int cs,
                                                        const int32_t* __restrict__ ex, const int32_t* __restrict__ ey) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int se = -(*ex + *ey);      // the accumulators carry 2^(e_X + e_Y)
  // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8 as that XCD's (L / 8)-th workgroup.  The `cs` i tiles that share one
  // j tile's Y fragments are consecutive workgroups of ONE XCD, so its L2 fetches those fragments once (with i fastest in a
  // plain 3-D grid the four i tiles of the T / dx shapes sat on four XCDs: the 118 MB of projection planes crossed the
  // fabric four times per GEMM)
  const int ti = (int)((M + 127) / 128), tj = (int)((N + 127) / 128), nic = (ti + cs - 1) / cs;
  const int L = blockIdx.x, slot = L >> 3;
  const int c = (slot / cs) * 8 + (L & 7);
  if (c >= nic * tj * nsplit) return;
  const int bx = (c % nic) * cs + slot % cs, by = (c / nic) % tj, bz = c / (nic * tj);
  if (bx >= ti) return;
  const int64_t ib0 = ((int64_t)bx * 2 + (wave & 1)) * TI;
  const int64_t jb0 = ((int64_t)by * 2 + (wave >> 1)) * TJ;
  // split K (few output tiles, long K): slice bz of the k-steps, partial sums to `part` [nsplit][M][N], summed in
  // slice order by k_tg_reduce
  const int kb = (int)((int64_t)KS16_all * bz / nsplit), KS16 = (int)((int64_t)KS16_all * (bz + 1) / nsplit) - kb;
  Xhi += (int64_t)kb * 64; Xlo += (int64_t)kb * 64; Yhi += (int64_t)kb * 64; Ylo += (int64_t)kb * 64;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  uint4 xh[NBUF][TI], xl[NBUF][TI], yh[NBUF][TJ], yl[NBUF][TJ];
#define TG_LOAD(s_, ks_)                                                                     \
  {                                                                                          \
    _Pragma("unroll") for (int a = 0; a < TI; ++a) {                                         \
      const int64_t o = ((ib0 + a) * KST + (ks_)) * 64 + lane;                                \
      xh[s_][a] = Xhi[o]; xl[s_][a] = Xlo[o];                                                \
    }                                                                                        \
    _Pragma("unroll") for (int b = 0; b < TJ; ++b) {                                         \
      const int64_t o = ((jb0 + b) * KST + (ks_)) * 64 + lane;                                \
      yh[s_][b] = Yhi[o]; yl[s_][b] = Ylo[o];                                                \
    }                                                                                        \
  }
#define TG_STEP(s_)                                                                          \
  {                                                                                          \
    _Pragma("unroll") for (int a = 0; a < TI; ++a) _Pragma("unroll") for (int b = 0; b < TJ; ++b) { \
      acc[a][b] = TG_MFMA(xl[s_][a], yh[s_][b], acc[a][b]);                                  \
      acc[a][b] = TG_MFMA(xh[s_][a], yl[s_][b], acc[a][b]);                                  \
      acc[a][b] = TG_MFMA(xh[s_][a], yh[s_][b], acc[a][b]);                                  \
    }                                                                                        \
  }
  // three register buffers, fragments fetched two k-steps ahead (a k-step is 12 MFMAs per wave: shorter than an L2 round
  // trip); no conditional code around a step or its prefetch (a prefetch past the end re-reads the last k-step)
#define TG_KCL(k_) ((k_) < KS16 ? (k_) : KS16 - 1)
  if constexpr (NBUF == 3) {
    TG_LOAD(0, 0);
    TG_LOAD(1, TG_KCL(1));
    int ks = 0;
    for (; ks + 3 <= KS16; ks += 3) {
      TG_LOAD(2, TG_KCL(ks + 2));
      __builtin_amdgcn_sched_barrier(0);
      TG_STEP(0);
      TG_LOAD(0, TG_KCL(ks + 3));
      __builtin_amdgcn_sched_barrier(0);
      TG_STEP(1);
      TG_LOAD(1, TG_KCL(ks + 4));
      __builtin_amdgcn_sched_barrier(0);
      TG_STEP(2);
    }
    if (ks < KS16) TG_STEP(0);
    if (ks + 1 < KS16) TG_STEP(1);
  } else {   // two buffers (the 64 x 128 wave tile leaves no room for a third): one k-step ahead
    TG_LOAD(0, 0);
    int ks = 0;
    for (; ks + 2 <= KS16; ks += 2) {
      TG_LOAD(1, TG_KCL(ks + 1));
      __builtin_amdgcn_sched_barrier(0);
      TG_STEP(0);
      TG_LOAD(0, TG_KCL(ks + 2));
      __builtin_amdgcn_sched_barrier(0);
      TG_STEP(1);
    }
    if (ks < KS16) TG_STEP(0);
  }
#undef TG_KCL
#undef TG_LOAD
#undef TG_STEP
  float ss = 0.f;   // sum of the squares this lane stores (the global gradient norm takes it from here: no second pass over C)
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int64_t j = (jb0 + b) * 32 + (lane & 31);
      if (j >= N) continue;
      if (nsplit > 1) {
        float* pz = part + (int64_t)bz * M * N + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t i = (ib0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (i < M) pz[i * N] = x3_scale(acc[a][b][r], se);
        }
        continue;
      }
      const int64_t oj = tg_off(cj, j);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t i = (ib0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (i < M) {
          const float v = x3_scale(acc[a][b][r], se);
          C[tg_off(ci, i) + oj] = v;
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

__global__ __launch_bounds__(256) void k_tg_reduce(const float* __restrict__ part, int nsplit, int64_t M, int64_t N, float* __restrict__ C,
                                                   TgIdx ci, TgIdx cj, double* __restrict__ sumsq) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float a = 0.f;
  if (e < M * N) {
    for (int z = 0; z < nsplit; ++z) a += part[(int64_t)z * M * N + e];   // slice order: deterministic
    C[tg_off(ci, e / N) + tg_off(cj, e % N)] = a;
  }
  if (sumsq) {   // uniform: every lane takes part in the shuffles
    double w = (double)a * (double)a;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(sumsq + (blockIdx.x * 4 + (threadIdx.x >> 6)) % TG_SUMSQ_SLOTS, w);
  }
}

// one wave per 128 x 128 tile (train_gemm_w128_bf16.hip) when tiles x K slices can fill the 1024 SIMDs
static bool tg_use_w128(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128), ks16 = (K + 15) / 16;
  static const bool off = getenv("COPER_TG_NO_W128") != nullptr;   // A/B switch
  if (off) return false;
  if (tiles >= 768) return true;
  return tiles * (ks16 / 16) >= 768;      // slices of at least 16 k-steps
}

int tg_split_k(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128), ks16 = (K + 15) / 16;
  if (tg_use_w128(M, N, K)) {
    // tiles x slices just under a whole number of rounds of 1024 single-wave workgroups
    if (tiles >= 768) return 1;
    int64_t sp = 1024 / tiles;
    if (sp > ks16 / 16) sp = ks16 / 16;
    return sp < 2 ? 1 : (int)sp;
  }
  // four-wave kernel: two workgroups fit on a CU (512 places): as many slices as keep every workgroup resident at once, each
  // at least 8 k-steps
  if (tiles >= 160 || ks16 < 32) return 1;
  int64_t sp = 512 / tiles;
  if (sp > ks16 / 8) sp = ks16 / 8;
  if (sp > 64) sp = 64;
  return sp < 2 ? 1 : (int)sp;
}

void tg_launch_w128(const TgPlanes& X, int64_t M, const TgPlanes& Y, int64_t N, int KS16, int KST, float* C, TgIdx ci, TgIdx cj, hipStream_t s,
                    int nsplit, float* part, double* sumsq);     // (reads X.exp / Y.exp)

int tg_gemm_nt(coper_handle* h, TgPlanes X, int64_t M, TgPlanes Y, int64_t N, int64_t K, float* C, TgIdx ci, TgIdx cj, hipStream_t s,
               int nsplit, float* part, double* sumsq, bool leave_slices) {
  const int KS16 = (int)((K + 15) / 16), KST = (int)tg_ks_stride(K);
  if (nsplit < 1 || !part) nsplit = 1;
  // workgroup tiles of 128 x 128 (rows of both plane sets are padded to TG_ROW_PAD).  A 128 x 256 tile (64 x 128 per wave,
  // 512 B of fragments per MFMA instead of 683, two register buffers) measured 165 us against 148 on the dP shape
  // (4608 x 6400 x 512): one k-step of prefetch does not cover the fill latency.
  if (tg_use_w128(M, N, K) && ci.seg == 0) {   // (two-level row views of C: the four-wave kernel)
    tg_launch_w128(X, M, Y, N, KS16, KST, C, ci, cj, s, nsplit, part, sumsq);
    if (nsplit > 1 && !leave_slices)
      hipLaunchKernelGGL(k_tg_reduce, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, part, nsplit, M, N, C, ci, cj, sumsq);
    COPER_HIP_TRY(h, hipGetLastError());
    return COPER_OK;
  }
  const int ti = (int)((M + 127) / 128), tj = (int)((N + 127) / 128);
  const int cs = ti < 4 ? ti : 4;                       // i tiles of one j tile kept together on an XCD
  const int64_t nclu = (int64_t)((ti + cs - 1) / cs) * tj * nsplit;
  dim3 grid((unsigned)(8 * cs * ((nclu + 7) / 8)));
  hipLaunchKernelGGL((k_gemm_nt_bf16x3<2, 2, 3>), grid, dim3(256), 0, s, X.hi, X.lo, Y.hi, Y.lo, KS16, KST, C, ci, cj, M, N, nsplit, part, sumsq, cs, X.exp, Y.exp);
  if (nsplit > 1 && !leave_slices)
    hipLaunchKernelGGL(k_tg_reduce, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, s, part, nsplit, M, N, C, ci, cj, sumsq);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
