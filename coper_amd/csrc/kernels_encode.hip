// coper_encode kernels: ConvE._create_predictions (models.py:354-426) in inference mode.
//
//   group_by_relation  counting sort of the batch by relation id -> perm + per-relation tiles of <= TQ
//                      queries, so every tile shares ONE set of generated dense weights (the
//                      reference instead materialises a [B,F,d] tensor, models.py:350,412).
//   conv               gather e1 row (models.py:176) -> reshape [emb_h, emb_w] (:355) [-> stack rel
//                      rows (:360-362)] -> VALID cross-correlation with static or per-relation filters
//                      (:375-385) + bias -> Conv1BN folded affine (:386-388) -> ReLU (:389) ->
//                      NHWC flatten (i,j,c) (:404) [-> concat rel_emb (:406-407)] -> x_sorted[pos, F_pad].
//   dense              z = x . W_rel (:410 / :412) on v_mfma_f32_16x16x4_f32 (exact f32), K = F split
//                      over the 4 waves of a workgroup and optionally over grid.y; W streamed from
//                      the fragment-major cache straight into VGPRs (1 KiB per wave-instruction).
//   finalize           sum K-splits in fixed order + fc bias (:410/:412) -> FCBN folded (:416-418)
//                      -> ReLU (:419) -> h[perm[pos], d].
#include "coper_internal.h"

namespace coper {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// grouping
// ------------------------------------------------------------------------------------------------
__global__ void k_rel_hist(const int64_t* __restrict__ rel, int64_t B, int use_rel, int64_t R,
                           int32_t* __restrict__ count, int32_t* __restrict__ bad) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t key = use_rel ? rel[b] : 0;
  if (key < 0 || key >= R) { atomicAdd(bad, 1); key = 0; }  // clamped, and reported by coper_check_ids
  atomicAdd(&count[key], 1);
}

// single block: exclusive scan of the counts -> offsets, and the tile list
__global__ __launch_bounds__(1024) void k_rel_scan_tiles(const int32_t* __restrict__ count, int64_t R, int tq,
                                                         int32_t* __restrict__ offset,
                                                         int32_t* __restrict__ tiles,
                                                         int32_t* __restrict__ n_tiles) {
  __shared__ int s_cnt[1024];
  __shared__ int s_til[1024];
  __shared__ int carry_cnt, carry_til;
  if (threadIdx.x == 0) { carry_cnt = 0; carry_til = 0; }
  __syncthreads();
  for (int64_t base = 0; base < R; base += 1024) {
    int64_t rid = base + threadIdx.x;
    int c = rid < R ? count[rid] : 0;
    int nt = (c + tq - 1) / tq;
    s_cnt[threadIdx.x] = c;
    s_til[threadIdx.x] = nt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int a = 0, b2 = 0;
      if ((int)threadIdx.x >= off) { a = s_cnt[threadIdx.x - off]; b2 = s_til[threadIdx.x - off]; }
      __syncthreads();
      s_cnt[threadIdx.x] += a;
      s_til[threadIdx.x] += b2;
      __syncthreads();
    }
    int excl_c = carry_cnt + s_cnt[threadIdx.x] - c;
    int excl_t = carry_til + s_til[threadIdx.x] - nt;
    if (rid < R) {
      offset[rid] = excl_c;
      for (int j = 0; j < nt; ++j) {
        int32_t* t = tiles + 4 * (int64_t)(excl_t + j);
        t[0] = (int32_t)rid;
        t[1] = excl_c + j * tq;
        t[2] = (c - j * tq) < tq ? (c - j * tq) : tq;
        t[3] = 0;
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry_cnt += s_cnt[1023]; carry_til += s_til[1023]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { offset[R] = carry_cnt; *n_tiles = carry_til; }
}

__global__ void k_rel_scatter(const int64_t* __restrict__ rel, int64_t B, int use_rel, int64_t R,
                              const int32_t* __restrict__ offset, int32_t* __restrict__ cursor,
                              int32_t* __restrict__ perm) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t key = use_rel ? rel[b] : 0;
  if (key < 0 || key >= R) key = 0;
  int pos = atomicAdd(&cursor[key], 1);
  perm[offset[key] + pos] = (int32_t)b;
}

int launch_group_by_relation(coper_handle* h, const int64_t* rel, int64_t B, int tq, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t R = dm.gen_fc ? dm.R : 1;
  COPER_HIP_TRY(h, hipMemsetAsync(h->rel_count, 0, sizeof(int32_t) * (dm.R + 2), s));
  COPER_HIP_TRY(h, hipMemsetAsync(h->rel_cursor, 0, sizeof(int32_t) * (dm.R + 1), s));
  unsigned nb = (unsigned)((B + 255) / 256);
  // rel_count[R+1] doubles as the out-of-range counter (ids are validated on device, never trusted)
  hipLaunchKernelGGL(k_rel_hist, dim3(nb), dim3(256), 0, s, rel, B, dm.gen_fc ? 1 : 0, R, h->rel_count,
                     h->rel_count + dm.R + 1);
  hipLaunchKernelGGL(k_rel_scan_tiles, dim3(1), dim3(1024), 0, s, h->rel_count, R, tq, h->rel_offset, h->tiles,
                     h->n_tiles);
  hipLaunchKernelGGL(k_rel_scatter, dim3(nb), dim3(256), 0, s, rel, B, dm.gen_fc ? 1 : 0, R, h->rel_offset,
                     h->rel_cursor, h->perm);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// conv + BN + ReLU  (one workgroup per sorted position)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_bn_relu(
    const int64_t* __restrict__ e1, const int64_t* __restrict__ rel, const float* __restrict__ e1_rows,
    const int32_t* __restrict__ perm, const float* __restrict__ ent, int64_t shard_lo, int64_t n_local,
    const float* __restrict__ rel_emb, const float* __restrict__ conv_w, const float* __restrict__ conv_b,
    int per_rel_conv, const float* __restrict__ scale, const float* __restrict__ shift, int d, int r, int emb_w,
    int in_h, int in_w, int stacked, int fh, int fw, int C, int Ho, int Wo, int concat_rel, int64_t F,
    int64_t F_pad, int64_t R, float* __restrict__ x_sorted) {
  extern __shared__ float lds[];  // img[in_h*in_w] | taps[fh*fw*C] | kb[C] | sc[C] | sh[C]
  float* img = lds;
  float* taps = img + in_h * in_w;
  float* kb = taps + fh * fw * C;
  float* sc = kb + C;
  float* sh = sc + C;
  int64_t pos = blockIdx.x;
  int64_t q = perm[pos];
  int64_t rid = rel[q];
  if (rid < 0 || rid >= R) rid = 0;
  // image = e1 row (optionally stacked on the relation row)
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
  const float* wsrc = per_rel_conv ? conv_w + rid * (int64_t)(fh * fw * C) : conv_w;
  const float* bsrc = per_rel_conv ? conv_b + rid * (int64_t)C : conv_b;
  for (int k = threadIdx.x; k < fh * fw * C; k += 256) taps[k] = wsrc[k];
  for (int k = threadIdx.x; k < C; k += 256) { kb[k] = bsrc[k]; sc[k] = scale[k]; sh[k] = shift[k]; }
  __syncthreads();
  float* xo = x_sorted + pos * F_pad;
  int64_t Fc = (int64_t)Ho * Wo * C;
  for (int64_t idx = threadIdx.x; idx < Fc; idx += 256) {
    int c = (int)(idx % C);
    int p = (int)(idx / C);
    int i = p / Wo, j = p % Wo;
    float y = 0.f;
    for (int u = 0; u < fh; ++u)
      for (int v = 0; v < fw; ++v) y = fmaf(img[(i + u) * in_w + (j + v)], taps[(u * fw + v) * C + c], y);
    y += kb[c];
    y = fmaf(y, sc[c], sh[c]);
    xo[idx] = fmaxf(y, 0.f);
  }
  if (concat_rel)
    for (int k = threadIdx.x; k < r; k += 256) xo[Fc + k] = rel_emb[rid * r + k];
  for (int64_t k = F + threadIdx.x; k < F_pad; k += 256) xo[k] = 0.f;
}

int launch_conv(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                hipStream_t s) {
  const Dims& dm = h->dm;
  const float* rel_emb = dm.lookup ? nullptr : h->params["rel_emb"].ptr;
  const float* cw = dm.gen_conv ? h->conv_w_rel : h->params["conv1_weights"].ptr;
  const float* cb = dm.gen_conv ? h->conv_b_rel : h->params["conv1_bias"].ptr;
  size_t lds = sizeof(float) * ((size_t)dm.in_h * dm.in_w + (size_t)dm.fh * dm.fw * dm.C + 3 * (size_t)dm.C);
  ScopedKernelTimer t(h, "conv", s);
  hipLaunchKernelGGL(k_conv_bn_relu, dim3((unsigned)B), dim3(256), lds, s, e1, rel, e1_rows, h->perm,
                     h->params["ent_emb"].ptr, (int64_t)h->cfg.shard_lo, dm.n_local, rel_emb, cw, cb,
                     dm.gen_conv ? 1 : 0, h->conv_scale, h->conv_shift, dm.d, dm.r, dm.emb_w, dm.in_h, dm.in_w,
                     dm.stacked ? 1 : 0, dm.fh, dm.fw, dm.C, dm.Ho, dm.Wo, dm.concat_rel ? 1 : 0, dm.F, dm.F_pad,
                     dm.R, h->x_sorted);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// dense: Z^T[d, TQ] = W_rel^T[d, F] . X^T[F, TQ] per tile, v_mfma_f32_16x16x4_f32
//   A operand (features on rows):  lane l holds W[f = 16ks + 4(l>>4) + t][feat = 16fb + (l&15)]  (Wf float4, t = component)
//   B operand (queries on columns): lane l holds x[q = l&15][f = 16ks + 4(l>>4) + t]
//   MFMA t contracts k-group {4g + t : g = 0..3}; D: col = lane&15 (query), row = 4(lane>>4) + reg (feature).
// ------------------------------------------------------------------------------------------------
template <int NFB, int NQ>
__global__ __launch_bounds__(256) void k_dense_f32(const float4* __restrict__ Wf, const float* __restrict__ x_sorted,
                                                   const int32_t* __restrict__ tiles,
                                                   const int32_t* __restrict__ n_tiles, int nfb, int64_t ksteps,
                                                   int64_t F_pad, int ksplit, int64_t Bcap, int d_pad16,
                                                   float* __restrict__ z_part) {
  extern __shared__ float4 red[];  // [2][NFB*NQ][64]
  int tile = blockIdx.x;
  if (tile >= *n_tiles) return;
  const int slice = blockIdx.y;
  const int fb0 = blockIdx.z * NFB;
  const int64_t relw = tiles[4 * tile + 0];
  const int start = tiles[4 * tile + 1];
  const int n = tiles[4 * tile + 2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nparts = ksplit * 4, part = slice * 4 + wave;
  const int64_t kb = ksteps * part / nparts, ke = ksteps * (part + 1) / nparts;

  f32x4 acc[NFB][NQ];
#pragma unroll
  for (int a = 0; a < NFB; ++a)
#pragma unroll
    for (int b = 0; b < NQ; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float4* wbase[NFB];
#pragma unroll
  for (int a = 0; a < NFB; ++a) {
    int fb = fb0 + a < nfb ? fb0 + a : nfb - 1;  // clamp (results of clamped blocks are discarded)
    wbase[a] = Wf + ((relw * nfb + fb) * ksteps) * 64 + lane;
  }
  const float* xrow[NQ];
#pragma unroll
  for (int b = 0; b < NQ; ++b) {
    int qi = b * 16 + (lane & 15);
    if (qi > n - 1) qi = n - 1;
    xrow[b] = x_sorted + (int64_t)(start + qi) * F_pad + 4 * (lane >> 4);
  }
  const bool second = n > 16;  // wave-uniform: skip the second query block of a short tile

  for (int64_t ks = kb; ks < ke; ++ks) {
    float4 av[NFB];
    float4 bv[NQ];
#pragma unroll
    for (int a = 0; a < NFB; ++a) av[a] = wbase[a][ks * 64];
#pragma unroll
    for (int b = 0; b < NQ; ++b) bv[b] = *(const float4*)(xrow[b] + 16 * ks);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int a = 0; a < NFB; ++a) {
        float av_t = t == 0 ? av[a].x : t == 1 ? av[a].y : t == 2 ? av[a].z : av[a].w;
        float b0 = t == 0 ? bv[0].x : t == 1 ? bv[0].y : t == 2 ? bv[0].z : bv[0].w;
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_t, b0, acc[a][0], 0, 0, 0);
        if (NQ > 1 && second) {
          float b1 = t == 0 ? bv[NQ - 1].x : t == 1 ? bv[NQ - 1].y : t == 2 ? bv[NQ - 1].z : bv[NQ - 1].w;
          acc[a][NQ - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_t, b1, acc[a][NQ - 1], 0, 0, 0);
        }
      }
    }
  }

  // cross-wave reduction in fixed order: (w0 + w2) + (w1 + w3)
  float4* slot = red + (size_t)(wave & 1) * (NFB * NQ * 64);
  if (wave >= 2) {
#pragma unroll
    for (int a = 0; a < NFB; ++a)
#pragma unroll
      for (int b = 0; b < NQ; ++b)
        slot[(a * NQ + b) * 64 + lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int a = 0; a < NFB; ++a)
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        float4 o = slot[(a * NQ + b) * 64 + lane];
        acc[a][b][0] += o.x; acc[a][b][1] += o.y; acc[a][b][2] += o.z; acc[a][b][3] += o.w;
      }
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int a = 0; a < NFB; ++a)
#pragma unroll
      for (int b = 0; b < NQ; ++b)
        red[(a * NQ + b) * 64 + lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int a = 0; a < NFB; ++a) {
      if (fb0 + a >= nfb) continue;
#pragma unroll
      for (int b = 0; b < NQ; ++b) {
        float4 o = red[(a * NQ + b) * 64 + lane];
        int qi = b * 16 + (lane & 15);
        if (qi < n) {
          float4 v = make_float4(acc[a][b][0] + o.x, acc[a][b][1] + o.y, acc[a][b][2] + o.z, acc[a][b][3] + o.w);
          float* dst = z_part + ((int64_t)slice * Bcap + start + qi) * d_pad16 + (fb0 + a) * 16 + 4 * (lane >> 4);
          *(float4*)dst = v;
        }
      }
    }
  }
}

__global__ void k_dense_finalize(const float* __restrict__ z_part, int ksplit, int64_t Bcap, int64_t B, int d,
                                 int d_pad16, const int32_t* __restrict__ perm, const int64_t* __restrict__ rel,
                                 const float* __restrict__ fc_b, int per_rel_bias, int64_t R,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 float* __restrict__ h_out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * d) return;
  int64_t pos = idx / d;
  int k = (int)(idx % d);
  float z = 0.f;
  for (int s = 0; s < ksplit; ++s) z += z_part[((int64_t)s * Bcap + pos) * d_pad16 + k];
  int64_t q = perm[pos];
  int64_t rid = rel[q];
  if (rid < 0 || rid >= R) rid = 0;
  z += per_rel_bias ? fc_b[rid * d + k] : fc_b[k];
  z = fmaf(z, scale[k], shift[k]);
  h_out[q * d + k] = fmaxf(z, 0.f);
}

template <int NFB>
static void dense_launch(coper_handle* h, int64_t T_max, int ksplit, int zgroups, hipStream_t s) {
  const Dims& dm = h->dm;
  constexpr int NQ = 2;
  size_t lds = (size_t)2 * NFB * NQ * 64 * sizeof(float4);
  hipLaunchKernelGGL((k_dense_f32<NFB, NQ>), dim3((unsigned)T_max, (unsigned)ksplit, (unsigned)zgroups), dim3(256),
                     lds, s, (const float4*)h->Wf, h->x_sorted, h->tiles, h->n_tiles, dm.nfb, dm.F_pad / 16, dm.F_pad,
                     ksplit, h->ws_queries, dm.d_pad16, h->z_part);
}

int launch_dense(coper_handle* h, const int64_t* rel, int64_t B, int tq, int ksplit, float* h_out, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t Rk = dm.gen_fc ? dm.R : 1;
  int64_t T_max = (B + tq - 1) / tq + (Rk < B ? Rk : B);
  {
    ScopedKernelTimer t(h, "dense", s);
    int nfb = dm.nfb;
    if (nfb == 13) dense_launch<13>(h, T_max, ksplit, 1, s);
    else if (nfb == 16) dense_launch<16>(h, T_max, ksplit, 1, s);
    else if (nfb <= 2) dense_launch<2>(h, T_max, ksplit, 1, s);
    else if (nfb <= 4) dense_launch<4>(h, T_max, ksplit, 1, s);
    else dense_launch<8>(h, T_max, ksplit, (nfb + 7) / 8, s);
    COPER_HIP_TRY(h, hipGetLastError());
  }
  const float* fcb = dm.gen_fc ? h->fc_b_rel : h->params["fc_bias"].ptr;
  int64_t total = B * dm.d;
  hipLaunchKernelGGL(k_dense_finalize, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->z_part, ksplit,
                     h->ws_queries, B, dm.d, dm.d_pad16, h->perm, rel, fcb, dm.gen_fc ? 1 : 0, dm.R, h->fc_scale,
                     h->fc_shift, h_out);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void k_gather_entities(const float* __restrict__ ent, const int64_t* __restrict__ ids, int64_t B, int d,
                                  int64_t lo, int64_t n_local, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * d) return;
  int64_t b = idx / d;
  int k = (int)(idx % d);
  int64_t row = ids[b] - lo;
  out[idx] = (row >= 0 && row < n_local) ? ent[row * d + k] : 0.f;
}

int launch_gather_entities(coper_handle* h, const int64_t* ids, int64_t B, float* out, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t total = B * dm.d;
  hipLaunchKernelGGL(k_gather_entities, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     h->params["ent_emb"].ptr, ids, B, dm.d, (int64_t)h->cfg.shard_lo, dm.n_local, out);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
