// coper_prepare kernels: everything inference derives from the parameters, evaluated once
// per weight update instead of once per sample.
//
//  * fold_bn          tf.layers.batch_normalization in inference mode (models.py:63-65,386-388,
//                     416-418) folded to y = x*scale + shift, scale = gamma/sqrt(var+eps).
//  * gen_small        ContextualParameterGenerator.generate (models.py:56-76) for the small
//                     outputs: generator hidden layers, conv filters [R,9,C], conv bias [R,C],
//                     dense bias [R,d].
//  * gen_dense_frag   the same for the dense weights [R,F,d] (models.py:70,73,350), written
//                     directly in the MFMA-fragment-major layout the dense kernel streams:
//                       Wf[rel][fb][ks][lane] = float4{ W_rel[16ks + 4(lane>>4) + t][16fb + (lane&15)] , t=0..3 }
//                     i.e. lane l of a wave holds the A operand of v_mfma_f32_16x16x4_f32 for
//                     feature row (l&15) and k-group (l>>4); one wave-instruction = 1 KiB contiguous.
//  * entity_frag      ent_emb shard re-laid out for v_mfma_f32_32x32x2_f32:
//                       Ef[eb][ks][lane] = float4{ E[32eb + (lane&31)][8ks + 4(lane>>5) + t], t=0..3 }
//                     + pred_bias padded with -inf to the padded row count.
#include "coper_internal.h"

namespace coper {

__global__ void k_fold_bn(const float* __restrict__ gamma, const float* __restrict__ beta,
                          const float* __restrict__ mean, const float* __restrict__ var, int n, float eps,
                          float* __restrict__ scale, float* __restrict__ shift) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float inv = gamma[i] / sqrtf(var[i] + eps);
  scale[i] = inv;
  shift[i] = beta[i] - mean[i] * inv;
}

int launch_fold_bn(coper_handle* h, const float* gamma, const float* beta, const float* mean, const float* var,
                   int n, float eps, float* scale, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_bn, dim3((n + 255) / 256), dim3(256), 0, s, gamma, beta, mean, var, n, eps, scale,
                     shift);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// out[rel, n] = epi( sum_k ctx[rel,k] * P[k,n] )
__global__ void k_gen_small(const float* __restrict__ ctx, int64_t R, int K, const float* __restrict__ P, int64_t N,
                            const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int relu,
                            float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * N) return;
  int64_t rel = idx / N, n = idx % N;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(ctx[rel * K + k], P[(int64_t)k * N + n], acc);
  if (bn_scale) acc = fmaf(acc, bn_scale[n], bn_shift[n]);
  if (relu) acc = fmaxf(acc, 0.f);
  out[idx] = acc;
}

int launch_gen_small(coper_handle* h, const float* ctx, int64_t R, int K, const float* P, int64_t N,
                     const float* bn_scale, const float* bn_shift, bool relu, float* out, hipStream_t s) {
  int64_t total = R * N;
  hipLaunchKernelGGL(k_gen_small, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ctx, R, K, P, N, bn_scale,
                     bn_shift, relu ? 1 : 0, out);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// One thread owns one float4 of the fragment image and keeps its KC x 4 slice of P in registers
// while it walks the relations: P is read once, the [R, F, d] cache is written once, coalesced
// (HBM-write bound: R*F_pad*d_pad16*4 bytes).
template <int KC>
__global__ __launch_bounds__(256) void k_gen_dense_frag(const float* __restrict__ ctx, int64_t rel_begin,
                                                        int64_t rel_end, int Kfull, int k0, int kn,
                                                        const float* __restrict__ P, int64_t F, int d, int nfb,
                                                        int64_t ksteps, int accumulate, float4* __restrict__ Wf) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (fb, ks, lane)
  int64_t per_rel = (int64_t)nfb * ksteps * 64;
  if (gid >= per_rel) return;
  int lane = (int)(gid & 63);
  int64_t ks = (gid >> 6) % ksteps;
  int fb = (int)((gid >> 6) / ksteps);
  int i = fb * 16 + (lane & 15);
  int64_t fbase = 16 * ks + 4 * (lane >> 4);
  float p[KC][4];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int64_t f = fbase + t;
      p[k][t] = (k < kn && i < d && f < F) ? P[(int64_t)(k0 + k) * F * d + f * d + i] : 0.f;
    }
  }
  // relations of this block row
  int64_t rb = rel_begin + (int64_t)blockIdx.y * ((rel_end - rel_begin + gridDim.y - 1) / gridDim.y);
  int64_t re = rb + (rel_end - rel_begin + gridDim.y - 1) / gridDim.y;
  if (re > rel_end) re = rel_end;
  for (int64_t rel = rb; rel < re; ++rel) {
    const float* c = ctx + rel * Kfull + k0;  // wave-uniform -> scalar loads
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (k < kn) {
        float cv = c[k];
        a0 = fmaf(cv, p[k][0], a0);
        a1 = fmaf(cv, p[k][1], a1);
        a2 = fmaf(cv, p[k][2], a2);
        a3 = fmaf(cv, p[k][3], a3);
      }
    }
    float4* dst = Wf + rel * per_rel + gid;
    if (accumulate) {
      float4 o = *dst;
      a0 += o.x; a1 += o.y; a2 += o.z; a3 += o.w;
    }
    *dst = make_float4(a0, a1, a2, a3);
  }
}

// lookup table rows / static weights -> fragment image (pure re-layout, exact)
__global__ __launch_bounds__(256) void k_dense_frag_copy(const float* __restrict__ P, int64_t R, int64_t F, int d,
                                                         int nfb, int64_t ksteps, float4* __restrict__ Wf) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t per_rel = (int64_t)nfb * ksteps * 64;
  int64_t rel = blockIdx.y;
  if (gid >= per_rel) return;
  int lane = (int)(gid & 63);
  int64_t ks = (gid >> 6) % ksteps;
  int fb = (int)((gid >> 6) / ksteps);
  int i = fb * 16 + (lane & 15);
  int64_t fbase = 16 * ks + 4 * (lane >> 4);
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int64_t f = fbase + t;
    v[t] = (i < d && f < F) ? P[rel * F * d + f * d + i] : 0.f;
  }
  Wf[rel * per_rel + gid] = make_float4(v[0], v[1], v[2], v[3]);
}

int launch_gen_dense_frag(coper_handle* h, const float* ctx, int64_t R, int K, const float* P, int mode, float* Wf,
                          hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t ksteps = dm.F_pad / 16;
  int64_t per_rel = (int64_t)dm.nfb * ksteps * 64;
  unsigned bx = (unsigned)((per_rel + 255) / 256);
  if (mode == 1) {
    if (R > 65535) return fail(h, COPER_EUNSUPPORTED, "num_rel > 65535 with g_lookup dense weights");
    hipLaunchKernelGGL(k_dense_frag_copy, dim3(bx, (unsigned)R), dim3(256), 0, s, P, R, dm.F, dm.d, dm.nfb, ksteps,
                       (float4*)Wf);
    COPER_HIP_TRY(h, hipGetLastError());
    return COPER_OK;
  }
  // split relations over grid.y only when the fragment image alone cannot fill the chip
  unsigned by = 1;
  if (bx < 1024) {
    by = (unsigned)((1024 + bx - 1) / bx);
    if ((int64_t)by > R) by = (unsigned)R;
    if (by < 1) by = 1;
  }
  for (int k0 = 0; k0 < K; k0 += 32) {
    int kn = K - k0 < 32 ? K - k0 : 32;
    int acc = k0 > 0 ? 1 : 0;
    if (kn <= 8) {
      hipLaunchKernelGGL(k_gen_dense_frag<8>, dim3(bx, by), dim3(256), 0, s, ctx, (int64_t)0, R, K, k0, kn, P, dm.F,
                         dm.d, dm.nfb, ksteps, acc, (float4*)Wf);
    } else {
      hipLaunchKernelGGL(k_gen_dense_frag<32>, dim3(bx, by), dim3(256), 0, s, ctx, (int64_t)0, R, K, k0, kn, P,
                         dm.F, dm.d, dm.nfb, ksteps, acc, (float4*)Wf);
    }
    COPER_HIP_TRY(h, hipGetLastError());
  }
  return COPER_OK;
}

// One workgroup = one 32-row entity block: coalesced row-major read -> LDS -> fragment-major write.
__global__ __launch_bounds__(256) void k_entity_frag(const float* __restrict__ ent, const float* __restrict__ bias,
                                                     int64_t n_local, int d, int KS, float4* __restrict__ Ef,
                                                     float* __restrict__ bias_pad) {
  extern __shared__ float lds[];  // [32][8*KS + 4]  (+4: de-phase rows over LDS banks)
  const int ldw = 8 * KS + 4;
  int64_t eb = blockIdx.x;
  int64_t row0 = eb * 32;
  for (int idx = threadIdx.x; idx < 32 * 8 * KS; idx += 256) {
    int ri = idx / (8 * KS), k = idx % (8 * KS);
    int64_t row = row0 + ri;
    lds[ri * ldw + k] = (row < n_local && k < d) ? ent[row * d + k] : 0.f;
  }
  if (threadIdx.x < 32) {
    int64_t row = row0 + threadIdx.x;
    bias_pad[row] = row < n_local ? bias[row] : -INFINITY;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < KS * 64; j += 256) {
    int ks = j >> 6, l = j & 63;
    const float* src = lds + (l & 31) * ldw + 8 * ks + 4 * (l >> 5);
    Ef[(eb * KS + ks) * 64 + l] = make_float4(src[0], src[1], src[2], src[3]);
  }
}

int launch_entity_frag(coper_handle* h, const float* ent, const float* bias, hipStream_t s) {
  const Dims& dm = h->dm;
  size_t lds = (size_t)32 * (8 * dm.KS + 4) * sizeof(float);
  hipLaunchKernelGGL(k_entity_frag, dim3((unsigned)dm.n_eblk), dim3(256), lds, s, ent, bias, dm.n_local, dm.d,
                     dm.KS, (float4*)h->Ef, h->bias_pad);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

__global__ void k_bias_pad(const float* __restrict__ bias, int64_t n_local, int64_t n_pad, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) out[i] = i < n_local ? bias[i] : -INFINITY;
}

int launch_bias_pad(coper_handle* h, const float* bias, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t n_pad = dm.n_eblk * 32;
  hipLaunchKernelGGL(k_bias_pad, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, bias, dm.n_local, n_pad,
                     h->bias_pad);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

}  // namespace coper
