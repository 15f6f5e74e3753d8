// Training step (include/coper_hip.h "Training step").  fp32 throughout.
//
//   forward (train mode, models.py:354-426,438-443):
//     c = rel_emb[rel]; generator chains (g_MLP: projection, BN with batch statistics, ReLU, dropout per hidden layer)
//     conv filters: static, generated per sample (c or chain output times the last projection) or looked up
//     img = ent_emb[e1] (+ rel_emb[rel] stacked below it for plain ConvE)            k_tr_conv_fwd
//     y   = conv3x3(img) + bias                             [B, P = Ho*Wo, C]          k_tr_conv_fwd
//     Conv1BN (batch statistics when batch_norm_train_stats), ReLU, dropout -> x [B, F]   k_tr_bn1_fwd
//     concat_rel: x = [x | c]                                                          k_tr_concat / k_tr_split
//     dense: static z0 = x W (one GEMM); generated z[b] = sum_rho ctx[b,rho] (x[b] P[rho]) -- the FACTORED form: r
//            independent [B,F]x[F,d] products (one strided-batched GEMM), the [B,F,d] weight tensor of
//            models.py:70,412 is never formed; g_lookup: one pass over the looked-up [F,d] rows  (k_tr_lookup_*)
//     + dense bias, dropout, FCBN, ReLU -> h [B, d]                                   k_tr_fc_post(_slices), k_tr_fcbn_fwd
//     sampled scorer s[b,l] = h[b] . ent_emb[lookup[b,l]] + pred_bias[...] and the loss: k_tr_score_loss_dh (round 6: with ds and
//     dh = sum_l ds E[lookup] from the same pass over the gathered rows; k_tr_score_loss where d % 4 != 0) or 1-vs-all (GEMM)
//   backward: the transposes of the above (dense: dP[rho] = x^T (ctx[:,rho] . dz) batched, dA = dz P2^T one GEMM, then
//   the contraction with ctx / x); embedding-row gradients by float atomics or, when B*|E| is small, through a dense
//   d(loss)/d(logits) matrix and one GEMM.  Every GEMM is the split-fp16 MFMA kernel of train_gemm_bf16.hip (no library).
//   optimiser: tf.clip_by_global_norm + AMSGrad (amsgrad.py:130-159), one launch each over all tensors.
//   Schedule (round 6): the scorer's backward and the dP product run on two side streams of the training state, forked from and
//   joined to the caller's stream by events (train_step_impl: SideJoin); K slices of the few-tile products are added by the
//   kernels that consume them (k_tr_fc_post_slices, k_tr_bn1_bwd_sums<NS>), not by a launch of their own.
#include <cmath>
#include <cstring>

#include "coper_internal.h"
#include "train_common.h"
#include "train_gemm.h"

namespace coper {

namespace {

constexpr float BN_EPS = 1e-3f;

struct TrainParam {
  std::string name;
  float* p = nullptr;   // the caller's variable, updated in place
  int64_t n = 0;
  float* g = nullptr;   // gradient of the last step
  float* m = nullptr;   // AMSGrad slots
  float* v = nullptr;
  float* vh = nullptr;
};

}  // namespace

static const char* const kGenNames[4] = {"fc_weights", "fc_bias", "conv1_weights", "conv1_bias"};

struct TrainState {
  coper_train_config cfg;
  std::vector<TrainParam> tp;
  double b1p = 0, b2p = 0;
  uint32_t step = 0;
  int64_t capB = 0, capL = 0;
  // workspaces
  float *img = nullptr, *y = nullptr, *x = nullptr, *c = nullptr, *dA = nullptr;
  float *xc = nullptr, *dxc = nullptr;   // concat_rel: [B, F_conv + r] input of the dense layer and its gradient
  // g_MLP generator chains (models.py:56-70), one for fc_weights (0) and one for fc_bias (1):
  //   v[0] = c;  u[i] = v[i] P_i;  a[i] = relu(BN_i(u[i]));  v[i+1] = dropout(a[i]);  context = v[nh]
  struct Chain {
    int dims[COPER_MAX_CTX + 1];
    float* v[COPER_MAX_CTX + 1] = {};    // v[0] aliases TrainState::c
    float* u[COPER_MAX_CTX] = {};
    float* a[COPER_MAX_CTX] = {};
    float* dv[COPER_MAX_CTX + 1] = {};   // gradient w.r.t. v[i]
    float* du[COPER_MAX_CTX] = {};
    float* st[COPER_MAX_CTX] = {};       // [2][n]: mean | inv_std
  } chain[4];                 // 0 fc_weights, 1 fc_bias, 2 conv1_weights, 3 conv1_bias
  int nh = 0;                 // hidden layers of the dense-layer generators
  int nhc = 0;                // hidden layers of the conv generators
  float *Kt = nullptr, *Kbv = nullptr, *dKs = nullptr, *dkbs = nullptr;   // per-sample conv filters / biases and their gradients
  float* Sd = nullptr;       // [B, |E|] dense d(loss)/d(logits) when it fits (scorer backward by GEMM)
  int64_t capS = 0;
  float* A = nullptr;        // generated dense: T[r][B][d] (forward partials) | dT[r][B][d]
  // generated dense, split-bf16 GEMMs (train_gemm_bf16.hip): operand planes
  TgPlanes pX, pXt, pP1, pP3, pTn, pTb;   // x rows b | x rows f | P rows (rho,k) | P rows f | dT rows (rho,k) | dT rows b
  // the other products (static dense layer, 1-vs-all scorer, dE of the dense scorer backward): two operand plane sets and
  // the split-K partial sums, grown on demand
  TgPlanes mmX, mmY;
  int32_t* tg_exps = nullptr;    // [10 + TR_EXP_CACHE] the exponents of the ten plane sets (train_gemm.h), in the order pX pXt pP1 pP3 pTn pTb mmX mmY mmX2 mmY2,
                                 //   then the words tg_matmul hands out per operand tensor within a step (exp_cache)
  // the largest |W| of the dense weights (the last projection of the fc_weights generator / the static fc_weights), as the OPTIMIZER
  // left it: k_tr_amsgrad folds |p_new| of that leaf into TG_MAX_SLOTS slots while it writes it, the packs of the next step reduce
  // the slots -- the 118 MB pass that used to find the maximum (53 us of a 1.25 ms step) is gone.  Two sets: a step reads [wmax_cur],
  // its optimizer pass writes [wmax_cur ^ 1] (zeroed by the step's zero list).  Valid only from one train step to the next of this
  // handle with no coper_set_param in between (the tensors are the caller's: include/coper_hip.h, coper_train_step)
  unsigned* xmax = nullptr;      // TG_MAX_SLOTS: max x as k_tr_bn1_fwd wrote it;  dtmax: max |dT| as k_tr_scale_rows wrote it (zeroed per step)
  unsigned* dtmax = nullptr;
  unsigned* smax = nullptr;      // TG_MAX_SLOTS: max |S| as k_tr_build_S wrote it
  unsigned* wmax[2] = {nullptr, nullptr};
  int wmax_cur = 0;
  bool wmax_valid = false;
  // tg_matmul: the exponent of an operand tensor packed earlier in THIS step (x, the static W, dz and S are each packed for two
  // products): (tensor, its word).  Cleared at the start of a step and where a kernel rewrites a tensor in place.
  std::vector<std::pair<const float*, int32_t*>> exp_cache;
  unsigned* tg_scratch = nullptr;   // [2] the absmax reduction of tg_pack (zero between packs); [2..3]: the side stream's
  // round 6: a step is not one chain -- the projection's packs do not need the conv, the scorer's backward does not need the dense
  // layer's.  Those stretches run on a side stream of the state's own, forked from and joined to the caller's stream by events
  // (under capture they become branches of the graph).  side[0]: the packs in front, the scorer's backward; side[1]: the dP product.
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t ev_fork[3] = {nullptr, nullptr, nullptr}, ev_join[3] = {nullptr, nullptr, nullptr};
  size_t mmX_cap = 0, mmY_cap = 0, mmP_cap = 0;
  TgPlanes mmX2, mmY2;               // tg_matmul's operand planes on the side stream (TrainState::side[0])
  size_t mmX2_cap = 0, mmY2_cap = 0;
  float* mmP = nullptr;
  float *z0 = nullptr, *z1 = nullptr, *hv = nullptr, *dh = nullptr, *dz = nullptr, *ds = nullptr, *dx = nullptr, *dc = nullptr;
  double* red = nullptr;     // reduction scratch: [0] loss, [1] grad sumsq (total, written by the optimizer kernel), [2..] BN sums, then TG_SUMSQ_SLOTS partial sumsq
  float* bnst = nullptr;     // [4][max(C,d)]: mean1, inv1, mean2, inv2 ... see offsets below
  TrainParam* find(const char* name) {
    for (auto& t : tp)
      if (t.name == name) return &t;
    return nullptr;
  }
};

namespace {

template <typename T>
int talloc(coper_handle* h, T** p, size_t n) {
  if (*p) { (void)tracked_free(*p); *p = nullptr; }
  if (tracked_malloc((void**)p, n * sizeof(T)) != hipSuccess) return fail(h, COPER_ENOMEM, "hipMalloc failed (training workspace)");
  return COPER_OK;
}

// C(i, j) = sum_k X(i, k) Y(j, k) for two strided fp32 views, on the split-bf16 GEMM of train_gemm_bf16.hip: packs both
// operands into the state's plane sets (grown on demand), cuts K into slices when the output has few tiles.
struct MmView {
  const float* p;
  TgIdx ri, ki;
  bool rows_fast;   // consecutive rows contiguous in memory (else consecutive k)
};
static int tg_matmul(coper_handle* h, TrainState* T, hipStream_t s, const MmView& X, int64_t M, const MmView& Y, int64_t N, int64_t K,
                     float* C, TgIdx ci, TgIdx cj, double* sumsq = nullptr, const unsigned* x_slots = nullptr, const unsigned* y_slots = nullptr);

// ------------------------------------------------------------------------------------------------
// forward kernels
// ------------------------------------------------------------------------------------------------
// one workgroup per query: gather the image, 3x3 VALID cross-correlation + bias -> y[b, p, c]
__global__ __launch_bounds__(256) void k_tr_conv_fwd(const int64_t* __restrict__ e1, const int64_t* __restrict__ rel,
                                                     const float* __restrict__ ent, const float* __restrict__ rel_emb,
                                                     const float* __restrict__ K, const float* __restrict__ kb, int64_t E,
                                                     int64_t R, int d, int r, int in_h, int in_w, int stacked, int C, int Ho,
                                                     int Wo, float* __restrict__ img_out, float* __restrict__ c_out,
                                                     float* __restrict__ y, const float* __restrict__ K_ps,
                                                     const float* __restrict__ kb_ps, int fh, int fw) {
  extern __shared__ float lds[];  // img[in_h*in_w] | taps[fh*fw*C] | kb[C]
  const int nt = fh * fw;
  float* img = lds;
  float* taps = img + in_h * in_w;
  float* bias = taps + nt * C;
  const int64_t b = blockIdx.x;
  int64_t row = e1[b];
  if (row < 0 || row >= E) row = 0;
  int64_t rid = rel[b];
  if (rid < 0 || rid >= R) rid = 0;
  for (int t = threadIdx.x; t < d; t += 256) img[t] = ent[row * d + t];
  if (stacked)
    for (int t = threadIdx.x; t < r; t += 256) img[d + t] = rel_emb[rid * r + t];
  if (c_out)
    for (int t = threadIdx.x; t < r; t += 256) c_out[b * r + t] = rel_emb[rid * r + t];
  // per-sample filters (generated / looked up, models.py:374-380) or the shared static ones
  const float* Ksrc = K_ps ? K_ps + b * (int64_t)nt * C : K;
  const float* bsrc = kb_ps ? kb_ps + b * C : kb;
  for (int t = threadIdx.x; t < nt * C; t += 256) taps[t] = Ksrc[t];
  for (int t = threadIdx.x; t < C; t += 256) bias[t] = bsrc[t];
  __syncthreads();
  const int isz = in_h * in_w;
  for (int t = threadIdx.x; t < isz; t += 256) img_out[b * isz + t] = img[t];
  const int P = Ho * Wo;
  float* yb = y + b * (int64_t)P * C;
  for (int idx = threadIdx.x; idx < P * C; idx += 256) {
    const int cc = idx % C, p = idx / C;
    const int i = p / Wo, j = p - i * Wo;
    float a = 0.f;
    if (fh == 3 && fw == 3) {   // the shipped shape, unrolled; same summation order as the general loop
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) a = fmaf(img[(i + u) * in_w + j + v], taps[(u * 3 + v) * C + cc], a);
    } else {
      for (int u = 0; u < fh; ++u)
        for (int v = 0; v < fw; ++v) a = fmaf(img[(i + u) * in_w + j + v], taps[(u * fw + v) * C + cc], a);
    }
    yb[idx] = a + bias[cc];
  }
}

// per-column sums of a [rows, cols] matrix in double: out[0..cols) = sum, out[cols..2cols) = sum of squares
// (partial sums by row chunk, then atomics on doubles: order-dependent only in the last bits of a double)
// everything the step accumulates into with atomics, zeroed by ONE launch (was a dozen memsets of ~5 us each)
constexpr int TR_LK_NSL = 4;     // F slices of the looked-up dense layer's forward (partial sums in T->dx)
constexpr int TR_ZERO_MAX = 16;
constexpr int TR_COLSUM_SLICES = 20;   // column-sum scratch: one slice per use within a step (see colsum_slice)
struct ZeroList {
  void* p[TR_ZERO_MAX];
  size_t bytes[TR_ZERO_MAX];   // multiples of 4
  int n;
};
__global__ __launch_bounds__(256) void k_tr_zero_list(ZeroList zl) {
  const int e = blockIdx.y;
  if (e >= zl.n) return;
  char* base = (char*)zl.p[e];
  const size_t bytes = zl.bytes[e];
  const size_t head = ((16 - ((uintptr_t)base & 15)) & 15) < bytes ? ((16 - ((uintptr_t)base & 15)) & 15) : bytes;
  const size_t n16 = (bytes - head) / 16, tail0 = head + n16 * 16;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (size_t i = t; i < n16; i += stride) ((uint4*)(base + head))[i] = make_uint4(0, 0, 0, 0);
  for (size_t i = t * 4; i < head; i += stride * 4) *(uint32_t*)(base + i) = 0;
  for (size_t i = tail0 + t * 4; i < bytes; i += stride * 4) *(uint32_t*)(base + i) = 0;
}

constexpr int TR_CS_SLOTS = 16;   // copies of a column-sum slice that many workgroups add to (workgroup w: copy w % slots)
// ... folded into copy 0 by a launch of one workgroup (k_tr_fold_slots), so readers see one slice.  (Round 6 also tried the fold by the
// LAST workgroup of the adding launch -- a ticket behind a __threadfence: every workgroup then waits for its own stores to drain
// before the ticket, 140 us against 12 for k_tr_bn1_bwd_sums -- and the fold in every reading workgroup: + 10 us on 9,216 of them.)
__global__ __launch_bounds__(256) void k_tr_fold_slots(double* __restrict__ base, int n2, int nslots) {
  for (int j = threadIdx.x; j < n2; j += 256) {
    double v[TR_CS_SLOTS];
#pragma unroll
    for (int z = 0; z < TR_CS_SLOTS; ++z) v[z] = z < nslots ? base[(size_t)z * n2 + j] : 0.0;
    double a = 0;
#pragma unroll
    for (int z = 0; z < TR_CS_SLOTS; ++z) a += v[z];
    base[j] = a;
  }
}
__global__ __launch_bounds__(256) void k_tr_col_sums(const float* __restrict__ m, int64_t rows, int cols, double* __restrict__ out, int nslots) {
  // per-column sum and sum of squares in double; a block reduces its row lanes in LDS and issues ONE atomic pair per
  // column (many blocks adding to the same few addresses are contention-bound: 14x slower per add -- 1024 workgroups on the 64
  // addresses of Conv1BN's statistics took 33 us for a 3 us read, so those go to TR_CS_SLOTS copies of the slice)
  __shared__ double sh[2][256];
  out += (size_t)(blockIdx.x % nslots) * 2 * cols;
  if (cols <= 256) {
    const int cpt = 256 / cols;                         // row lanes per column
    const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
    double s = 0, q = 0;
    if (rl < cpt) {
      const int64_t st = (int64_t)gridDim.x * cpt;
      int64_t rr = (int64_t)blockIdx.x * cpt + rl;
      for (; rr + 3 * st < rows; rr += 4 * st) {          // four loads in flight, added in row order
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = m[(rr + u * st) * cols + col];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u]; q += (double)v[u] * v[u]; }
      }
      for (; rr < rows; rr += st) {
        const double v = m[rr * cols + col];
        s += v;
        q += v * v;
      }
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < cols) {
      for (int l = 1; l < cpt; ++l) { s += sh[0][threadIdx.x + l * cols]; q += sh[1][threadIdx.x + l * cols]; }
      atomicAdd(&out[col], s);
      atomicAdd(&out[cols + col], q);
    }
  } else {
    for (int cc = threadIdx.x; cc < cols; cc += 256) {
      double s = 0, q = 0;
      for (int64_t rr = blockIdx.x; rr < rows; rr += gridDim.x) {
        const double v = m[rr * cols + cc];
        s += v;
        q += v * v;
      }
      atomicAdd(&out[cc], s);
      atomicAdd(&out[cols + cc], q);
    }
  }
}

// BN statistics -> (mean, inv_std) used by forward and backward; moving statistics updated in place.
// unbiased_moving: [TF-semantics] the fused 4-D kernel feeds the unbiased variance into the moving average.
// unbiased_moving bit 1 (value 2): leave the moving statistics alone (coper_train_forward: a fetch without train_op runs none of
// the UPDATE_OPS, models.py:194-200).
__global__ void k_tr_bn_finish(const double* __restrict__ sums, int cols, double n, int use_batch, float momentum,
                               int unbiased_moving, float* __restrict__ mov_mean, float* __restrict__ mov_var,
                               float* __restrict__ mean_out, float* __restrict__ inv_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  if (use_batch) {
    const double mean = sums[c] / n;
    double var = sums[cols + c] / n - mean * mean;
    if (var < 0) var = 0;
    mean_out[c] = (float)mean;
    inv_out[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    if (unbiased_moving & 2) return;
    const double var_m = (unbiased_moving & 1) ? var * (n / (n - 1.0)) : var;
    mov_mean[c] = (float)((double)mov_mean[c] * momentum + mean * (1.0 - (double)momentum));
    mov_var[c] = (float)((double)mov_var[c] * momentum + var_m * (1.0 - (double)momentum));
  } else {
    mean_out[c] = mov_mean[c];
    inv_out[c] = 1.0f / sqrtf(mov_var[c] + BN_EPS);
  }
}

// x = keep * relu(bn(y)) / (1 - rate)     (elementwise over [B, P, C]; flat index = the dropout counter)
// the largest |value| a workgroup of 256 wrote -> one of TG_MAX_SLOTS slots (train_gemm.h: tg_pack's max_slots): the elementwise
// kernel that PRODUCES a GEMM operand leaves its maximum behind, so that the pack needs no pass of its own over the tensor (round 6)
__device__ __forceinline__ void tr_block_max_to_slot(float v_abs, unsigned* __restrict__ slots) {
  unsigned m = __float_as_uint(v_abs);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
  __shared__ unsigned s_bm[4];
  if ((threadIdx.x & 63) == 0) s_bm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = s_bm[0] > s_bm[1] ? s_bm[0] : s_bm[1], b = s_bm[2] > s_bm[3] ? s_bm[2] : s_bm[3], w = a > b ? a : b;
    if (w) atomicMax(slots + (blockIdx.x & (TG_MAX_SLOTS - 1)), w);
  }
}

__global__ __launch_bounds__(256) void k_tr_bn1_fwd(const float* __restrict__ y, const float* __restrict__ mean,
                                                    const float* __restrict__ inv, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int C, int64_t total, uint32_t seed,
                                                    uint32_t step, uint32_t thr, float keep_scale, float* __restrict__ x,
                                                    unsigned* __restrict__ max_slots) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float out = 0.f;
  if (i < total) {
    const int c = (int)(i % C);
    float v = (y[i] - mean[c]) * inv[c] * gamma[c] + beta[c];
    v = v > 0.f ? v : 0.f;
    out = dropout_keep_u32(seed, step, 1u, (uint32_t)i, thr) ? v * keep_scale : 0.f;
    x[i] = out;
  }
  if (max_slots) tr_block_max_to_slot(out, max_slots);       // (x >= 0)
}

// z1 = keep * (z0 + bias_b) / (1 - rate).  Static: z0 from the GEMM, bias_b = fc_bias[k].  Generated:
// z0[b,k] = sum_rho c[b,rho] T[rho][b,k] (T[rho] = x P[rho], the batched GEMM), bias_b = sum_rho c[b,rho] Pb[rho,k]
__global__ __launch_bounds__(256) void k_tr_fc_post(const float* __restrict__ z0, const float* __restrict__ fc_bias,
                                                    const float* __restrict__ cw, int rw, const float* __restrict__ cb,
                                                    const float* __restrict__ Pb, int rb, int d, int64_t total, uint32_t seed,
                                                    uint32_t step, uint32_t thr, float keep_scale, float* __restrict__ z1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % d);
  const int64_t b = i / d;
  float v;
  if (Pb) {
    v = 0.f;
    for (int rho = 0; rho < rw; ++rho) v = fmaf(cw[b * rw + rho], z0[(int64_t)rho * total + i], v);
    for (int rho = 0; rho < rb; ++rho) v = fmaf(cb[b * rb + rho], Pb[rho * d + k], v);
  } else {
    v = z0[i] + fc_bias[k];
  }
  z1[i] = dropout_keep_u32(seed, step, 2u, (uint32_t)i, thr) ? v * keep_scale : 0.f;
}

// the generated dense layer straight from the K slices of its product (tg_gemm_nt: leave_slices): T[rho][b, k] = the slices of
// part[z][b][rho d + k] added in slice order -- what k_tg_reduce stores, kept for the backward pass -- and z1 as k_tr_fc_post forms it.
// One kernel instead of two, the 13 MB of T written once and not read back (round 6: 39.8 us -> the slices' 65 MB at stream rate).
template <int NS>
__global__ __launch_bounds__(256) void k_tr_fc_post_slices(const float* __restrict__ part, const float* __restrict__ cw, int rw,
                                                           const float* __restrict__ cb, const float* __restrict__ Pb, int rb, int d,
                                                           int64_t total, uint32_t seed, uint32_t step, uint32_t thr, float keep_scale,
                                                           float* __restrict__ Tf, float* __restrict__ z1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % d);
  const int64_t b = i / d;
  const int64_t N = (int64_t)rw * d, MN = (total / d) * N;
  const float* p = part + b * N + k;
  float v = 0.f;
#pragma unroll 4
  for (int rho = 0; rho < rw; ++rho) {
    float t[NS];
#pragma unroll
    for (int z = 0; z < NS; ++z) t[z] = p[(int64_t)z * MN + (int64_t)rho * d];
    float a = 0.f;
#pragma unroll
    for (int z = 0; z < NS; ++z) a += t[z];
    Tf[(int64_t)rho * total + i] = a;
    v = fmaf(cw[b * rw + rho], a, v);
  }
  for (int rho = 0; rho < rb; ++rho) v = fmaf(cb[b * rb + rho], Pb[rho * d + k], v);
  z1[i] = dropout_keep_u32(seed, step, 2u, (uint32_t)i, thr) ? v * keep_scale : 0.f;
}

__global__ __launch_bounds__(256) void k_tr_fcbn_fwd(const float* __restrict__ z1, const float* __restrict__ mean,
                                                     const float* __restrict__ inv, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int d, int64_t total,
                                                     float* __restrict__ hv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % d);
  const float v = (z1[i] - mean[k]) * inv[k] * gamma[k] + beta[k];
  hv[i] = v > 0.f ? v : 0.f;
}

// sampled scorer + loss + d(loss)/ds.  One workgroup per query; h[b] in LDS; one lookup entry per thread.
__global__ __launch_bounds__(256) void k_tr_score_loss(const float* __restrict__ hv, const float* __restrict__ ent,
                                                       const float* __restrict__ pred_bias,
                                                       const int32_t* __restrict__ lookup, const float* __restrict__ labels,
                                                       int64_t E, int d, int64_t L, float ls_eps, float inv_E, float inv_BL,
                                                       float* __restrict__ ds, double* __restrict__ loss_acc) {
  extern __shared__ float hl[];
  __shared__ double part[256];
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < d; k += 256) hl[k] = hv[b * d + k];
  __syncthreads();
  double acc = 0.0;
  for (int64_t l = threadIdx.x; l < L; l += 256) {
    int64_t row = lookup[b * L + l];
    if (row < 0 || row >= E) row = 0;
    const float* er = ent + row * d;
    float s = 0.f;
    if ((d & 3) == 0) {   // 16-byte loads of the gathered row; the fma chain keeps its order
      const float4* er4 = (const float4*)er;
      for (int k4 = 0; k4 < d / 4; ++k4) {
        const float4 e = er4[k4];
        s = fmaf(hl[4 * k4 + 0], e.x, s); s = fmaf(hl[4 * k4 + 1], e.y, s);
        s = fmaf(hl[4 * k4 + 2], e.z, s); s = fmaf(hl[4 * k4 + 3], e.w, s);
      }
    } else {
      for (int k = 0; k < d; ++k) s = fmaf(hl[k], er[k], s);
    }
    s += pred_bias[row];
    const float t = (1.f - ls_eps) * labels[b * L + l] + inv_E;                  // models.py:450
    const float as = fabsf(s);
    acc += (double)(fmaxf(s, 0.f) - s * t + log1pf(expf(-as)));                 // sigmoid cross-entropy with logits
    const float sg = 1.f / (1.f + expf(-s));
    ds[b * L + l] = (sg - t) * inv_BL;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(loss_acc, part[0]);
}

// The sampled scorer of a training step in ONE pass over the gathered rows (round 6): scores, loss, ds AND dh = sum_l ds E[row].
// k_tr_score_loss + k_tr_dh_gather4 each read the B L rows of d floats (410 MB at 512 x 1000 x 200: 61 + 45 us); sigmoid
// cross-entropy is elementwise, so ds[b, l] is known as soon as row l's score is, while the row is still in registers.
// k_tr_dh_gather4's layout: a thread owns four features (one 16-byte load) of one of 256 / (d / 4) row slots, SF_U rows per
// slot and batch; the next batch's rows are requested before this batch is touched.  Per batch: every thread's four-term
// share of its rows' dot products -> LDS; one thread per row adds the d / 4 shares in feature order, forms loss and ds;
// every thread adds ds x its registers to its dh share.  (An earlier fused form -- a wave per row, butterfly sums -- was
// latency end to end: 242 us.)
#ifndef COPER_SF_U
#define COPER_SF_U 12
#endif
constexpr int SF_U = COPER_SF_U;      // rows per slot and batch
constexpr int SF_MAX_L = 8192;       // lookup entries of a query held in LDS (32 KB)
__global__ __launch_bounds__(256) void k_tr_score_loss_dh(const float* __restrict__ hv, const float* __restrict__ ent,
                                                          const float* __restrict__ pred_bias, const int32_t* __restrict__ lookup,
                                                          const float* __restrict__ labels, int64_t E, int d, int L, float ls_eps,
                                                          float inv_E, float inv_BL, float* __restrict__ ds, float* __restrict__ dh,
                                                          double* __restrict__ loss_acc) {
  extern __shared__ float4 sf_lds[];   // float4 [slots][d4] (the slots' dh shares at the end) | float part[RB][d4 + 1] | float g[RB] | int ids[L]
  __shared__ double red[256];
  const int64_t b = blockIdx.x;
  const int d4 = d >> 2, slots = 256 / d4 > 256 / SF_U ? 256 / SF_U : 256 / d4, RB = SF_U * slots, PS = d4 + 1;      // (RB <= 256: a thread per row of a batch)
  // threads that share a row's sum of partial products (a power of two, neighbours in a wave)
  const int tpr = 256 / RB >= 8 ? 8 : (256 / RB >= 4 ? 4 : (256 / RB >= 2 ? 2 : 1));
  float* part = (float*)(sf_lds + slots * d4);
  float* gsh = part + RB * PS;
  int* ids = (int*)(gsh + RB);
  const int slot = threadIdx.x / d4, q4 = threadIdx.x - slot * d4;
  const bool live = slot < slots;
  // the query's rows, range-checked once (a row id is read by the thread that loads the row, by the thread that scores it, ...)
  for (int l = threadIdx.x; l < L; l += 256) {
    const int32_t row = lookup[b * L + l];
    ids[l] = (row < 0 || row >= E) ? 0 : row;
  }
  const float4 h4 = live ? *(const float4*)(hv + b * d + 4 * q4) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  double lacc = 0.0;
  float4 va[SF_U], vb[SF_U];
  // (every load of the loop is unconditional, with clamped indices, and issued in ONE order -- this batch's label and bias, then the
  //  next batch's rows: the vector-memory counter retires in order, so a wait for a load issued after the prefetch would drain it,
  //  and a load under a branch makes the compiler wait for everything)
#define SF_FETCH(l0_, dst_)                                                                   \
  {                                                                                           \
    _Pragma("unroll") for (int u = 0; u < SF_U; ++u) {                                        \
      const int l = (l0_) + u * slots + slot;                                                 \
      const int64_t row = ids[l < L ? l : L - 1];                                             \
      dst_[u] = *(const float4*)(ent + row * d + 4 * qc);                                     \
    }                                                                                         \
  }
  const int qc = live ? q4 : 0;                                  // (threads past the last slot load, and drop, a valid address)
  const int sr = threadIdx.x / tpr, sj = threadIdx.x % tpr;      // scoring: row sr of the batch, share sj of its partial products
  const int q_lo = (int)((int64_t)d4 * sj / tpr), q_hi = (int)((int64_t)d4 * (sj + 1) / tpr);
#define SF_BATCH(l0_, cur_, nxt_)                                                             \
  {                                                                                           \
    const int lb = (l0_);                                                                     \
    const bool on = sr < RB && lb + sr < L;    /* (uniform over the tpr neighbours of a row) */ \
    const int lc = lb + sr < L ? lb + sr : L - 1;                                             \
    const float lab = labels[b * L + lc], pb = pred_bias[ids[lc]];                            \
    SF_FETCH(lb + RB, nxt_);                                                                  \
    if (live) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < SF_U; ++u) {                                      \
        float pz = h4.x * cur_[u].x;                                                          \
        pz = fmaf(h4.y, cur_[u].y, pz); pz = fmaf(h4.z, cur_[u].z, pz); pz = fmaf(h4.w, cur_[u].w, pz); \
        part[(u * slots + slot) * PS + q4] = pz;                                              \
      }                                                                                       \
    }                                                                                         \
    __syncthreads();                                                                          \
    float sc = 0.f;                                                                           \
    if (on) {                                                                                 \
      const float* pr = part + sr * PS;                                                       \
      for (int q = q_lo; q < q_hi; ++q) sc += pr[q];                                          \
    }                                                                                         \
    for (int o = 1; o < tpr; o <<= 1) sc += __shfl_xor(sc, o, 64);   /* (the same sum in every neighbour) */ \
    if (on && sj == 0) {                                                                      \
      sc += pb;                                                                               \
      const float t = (1.f - ls_eps) * lab + inv_E;                    /* models.py:450 */    \
      const float as = fabsf(sc);                                                             \
      lacc += (double)(fmaxf(sc, 0.f) - sc * t + log1pf(expf(-as)));   /* sigmoid cross-entropy with logits */ \
      const float sg = 1.f / (1.f + expf(-sc));                                               \
      const float g = (sg - t) * inv_BL;                                                      \
      ds[b * L + lb + sr] = g;                                                                \
      gsh[sr] = g;                                                                            \
    }                                                                                         \
    __syncthreads();                                                                          \
    if (live) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < SF_U; ++u)                                        \
        if (lb + u * slots + slot < L) {                                                      \
          const float g = gsh[u * slots + slot];                                              \
          acc.x = fmaf(g, cur_[u].x, acc.x); acc.y = fmaf(g, cur_[u].y, acc.y);               \
          acc.z = fmaf(g, cur_[u].z, acc.z); acc.w = fmaf(g, cur_[u].w, acc.w);               \
        }                                                                                     \
    }                                                                                         \
  }
  SF_FETCH(0, va);
  for (int l0 = 0; l0 < L; l0 += 2 * RB) {
    SF_BATCH(l0, va, vb);
    if (l0 + RB < L) SF_BATCH(l0 + RB, vb, va);      // (uniform)
  }
#undef SF_BATCH
#undef SF_FETCH
  if (live) sf_lds[slot * d4 + q4] = acc;
  red[threadIdx.x] = lacc;
  __syncthreads();
  if (live && slot == 0) {
    for (int s2 = 1; s2 < slots; ++s2) {   // fixed order
      const float4 o = sf_lds[s2 * d4 + q4];
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *(float4*)(dh + b * d + 4 * q4) = acc;
  }
  if (threadIdx.x == 0) {      // (the loss terms, in thread order)
    double a = 0.0;
    for (int t = 0; t < 256; ++t) a += red[t];
    atomicAdd(loss_acc, a);
  }
}

// ------------------------------------------------------------------------------------------------
// backward kernels
// ------------------------------------------------------------------------------------------------
// dh[b,k] = sum_l ds[b,l] E[lookup[b,l], k];  SCATTER: also dE[lookup, k] += ds h[b,k], dbias[lookup] += ds by
// float atomics (the route for entity tables too large for the dense S matrix below)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_tr_score_bwd(const float* __restrict__ hv, const float* __restrict__ ent,
                                                      const int32_t* __restrict__ lookup, const float* __restrict__ ds,
                                                      int64_t E, int d, int64_t L, float* __restrict__ dh,
                                                      float* __restrict__ dE, float* __restrict__ dbias) {
  const int64_t b = blockIdx.x;
  for (int k0 = 0; k0 < d; k0 += 256) {   // a thread per feature, 256 features at a time (d <= 256: one trip)
  const int k = k0 + threadIdx.x;
  const float hk = k < d ? hv[b * d + k] : 0.f;
  float acc = 0.f;
  int64_t l = 0;
  // eight gathered rows in flight per thread: the loop is a chain of dependent loads otherwise (ids -> row)
  for (; l + 8 <= L; l += 8) {
    int64_t row[8];
    float g[8], ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      row[u] = lookup[b * L + l + u];
      if (row[u] < 0 || row[u] >= E) row[u] = 0;
      g[u] = ds[b * L + l + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) ev[u] = k < d ? ent[row[u] * d + k] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // same summation order as the plain loop
      acc = fmaf(g[u], ev[u], acc);
      if (SCATTER) {
        if (k < d) atomicAdd(&dE[row[u] * d + k], g[u] * hk);
        if (k == 0) atomicAdd(&dbias[row[u]], g[u]);   // (k == 0 only in the first trip)
      }
    }
  }
  for (; l < L; ++l) {
    int64_t row = lookup[b * L + l];
    if (row < 0 || row >= E) row = 0;
    const float g = ds[b * L + l];
    if (k < d) {
      acc = fmaf(g, ent[row * d + k], acc);
      if (SCATTER) atomicAdd(&dE[row * d + k], g * hk);
    }
    if (SCATTER && k == 0) atomicAdd(&dbias[row], g);
  }
  if (k < d) dh[b * d + k] = acc;
  }
}

// dh[b,:] = sum_l ds[b,l] E[lookup[b,l], :] without the scatter (the dense-route backward adds dE by a GEMM): the gather is
// bandwidth work -- B L rows of d floats (410 MB at 512 x 1000 x 200) -- and needs tens of KB in flight per CU: a thread owns
// four features (16-byte loads) of one of 256 / (d / 4) row slots, eight rows ahead, so a workgroup keeps 8 * slots rows
// (32 KB at d = 200) in flight; the slots' partial sums meet in LDS.  (The per-feature form with eight 4-byte loads in
// flight per thread read at 3.1 TB/s.)
__global__ __launch_bounds__(256) void k_tr_dh_gather4(const float* __restrict__ ent, const int32_t* __restrict__ lookup,
                                                       const float* __restrict__ ds, int64_t E, int d, int64_t L,
                                                       float* __restrict__ dh) {
  extern __shared__ float4 sh4[];   // [slots][d / 4]
  const int64_t b = blockIdx.x;
  const int d4 = d >> 2, slots = 256 / d4;
  const int slot = threadIdx.x / d4, q4 = threadIdx.x - slot * d4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (slot < slots) {
    const int32_t* lk = lookup + b * L;
    const float* gs = ds + b * L;
    int64_t l = slot;
    for (; l + 7 * slots < L; l += 8 * slots) {
      int64_t row[8];
      float g[8];
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        row[u] = lk[l + u * slots];
        if (row[u] < 0 || row[u] >= E) row[u] = 0;
        g[u] = gs[l + u * slots];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(ent + row[u] * d + 4 * q4);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc.x = fmaf(g[u], v[u].x, acc.x); acc.y = fmaf(g[u], v[u].y, acc.y);
        acc.z = fmaf(g[u], v[u].z, acc.z); acc.w = fmaf(g[u], v[u].w, acc.w);
      }
    }
    for (; l < L; l += slots) {
      int64_t row = lk[l];
      if (row < 0 || row >= E) row = 0;
      const float g = gs[l];
      const float4 v = *(const float4*)(ent + row * d + 4 * q4);
      acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y); acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
    }
    sh4[slot * d4 + q4] = acc;
  }
  __syncthreads();
  if (slot == 0) {
    for (int s2 = 1; s2 < slots; ++s2) {   // fixed order
      const float4 o = sh4[s2 * d4 + q4];
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *(float4*)(dh + b * d + 4 * q4) = acc;
  }
}

// dense route of the scorer backward (small entity tables): S[b, lookup[b,l]] += ds[b,l], dbias[lookup] += ds;
// then dE = S^T h is a GEMM instead of B*L*d float atomics.
// Round 6: S is built ROW BY ROW in LDS -- a workgroup per query zeroes a stretch of its row in LDS (<= TR_S_CHUNK columns),
// adds its L sampled gradients with LDS atomics, writes the stretch out coalesced and keeps its largest magnitude for the pack of S
// (tg_pack's max_slots) -- instead of B L float atomics into a zero-filled 30 MB matrix, a pass over it for its maximum and a
// zeroing launch share: k_tr_scatter_ds 54 us + k_tg_absmax_exp 18 + the zero list's 30 MB at FB15k-237 shapes.  dbias = the column
// sums of S (k_tr_col_sums_add).  Duplicate ids of a row add in the order the LDS serves them, as the global atomics did.
constexpr int TR_S_CHUNK = 32768;      // columns per LDS stretch (128 KB)
__global__ __launch_bounds__(256) void k_tr_build_S(const int32_t* __restrict__ lookup, const float* __restrict__ ds, int64_t E, int64_t L,
                                                    float* __restrict__ S, unsigned* __restrict__ max_slots) {
  extern __shared__ float s_row[];
  const int64_t b = blockIdx.x;
  const int32_t* lk = lookup + b * L;
  const float* g = ds + b * L;
  float mx = 0.f;
  for (int64_t c0 = 0; c0 < E; c0 += TR_S_CHUNK) {
    const int n = (int)(E - c0 < TR_S_CHUNK ? E - c0 : TR_S_CHUNK);
    for (int i = threadIdx.x; i < n; i += 256) s_row[i] = 0.f;
    __syncthreads();
    for (int64_t l = threadIdx.x; l < L; l += 256) {
      int64_t row = lk[l];
      if (row < 0 || row >= E) row = 0;
      if (row >= c0 && row < c0 + n) atomicAdd(&s_row[row - c0], g[l]);
    }
    __syncthreads();
    float* out = S + b * E + c0;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float v = s_row[i];
      out[i] = v;
      mx = fmaxf(mx, fabsf(v));
    }
    __syncthreads();
  }
  tr_block_max_to_slot(mx, max_slots);
}

// out[c] += sum over rows of S[row, c]: row stretches of 64 per workgroup row, one float atomic per (stretch, column)
__global__ __launch_bounds__(256) void k_tr_col_sums_add(const float* __restrict__ S, int64_t rows, int64_t cols, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * 64, r1 = r0 + 64 < rows ? r0 + 64 : rows;
  float a = 0.f;
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {          // eight loads in flight (one at a time, a 64-row stretch was 64 dependent round trips: 17 - 20 us)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = S[(r + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; r < r1; ++r) a += S[r * cols + c];
  if (a != 0.f) atomicAdd(&out[c], a);
}

// out[i, j] += sum over rows b of w[b, i] v[b, j]   (a [ni x B] x [B x nj] product with a short ni: the generated dense bias'
// projection gradient dPb[rho, k] = sum_b c[b, rho] dz0[b, k]): a workgroup per (i, stretch of 64 rows), a thread per j
__global__ __launch_bounds__(256) void k_tr_wsum_rows_add(const float* __restrict__ w, const float* __restrict__ v, int64_t rows, int ni, int nj,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * 64, r1 = r0 + 64 < rows ? r0 + 64 : rows;
  for (int j = threadIdx.x; j < nj; j += 256) {
    float a = 0.f;
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
      float x[8], c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { x[u] = v[(r + u) * nj + j]; c[u] = w[(r + u) * ni + i]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) a = fmaf(c[u], x[u], a);
    }
    for (; r < r1; ++r) a = fmaf(w[r * ni + i], v[r * nj + j], a);
    atomicAdd(&out[(int64_t)i * nj + j], a);
  }
}

// 1-vs-all training (lookup == NULL, models.py:159-162,434-437): S holds the logits h E^T from a GEMM; add the bias,
// accumulate the loss, overwrite with d(loss)/d(logit)
__global__ __launch_bounds__(256) void k_tr_dense_loss(float* __restrict__ S, const float* __restrict__ pred_bias,
                                                       const float* __restrict__ labels, int64_t E, int64_t total, float ls_eps,
                                                       float inv_E, float inv_BL, double* __restrict__ loss_acc) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float s = S[i] + pred_bias[i % E];
    const float t = (1.f - ls_eps) * labels[i] + inv_E;
    acc += (double)(fmaxf(s, 0.f) - s * t + log1pf(expf(-fabsf(s))));
    S[i] = (1.f / (1.f + expf(-s)) - t) * inv_BL;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(loss_acc, part[0]);
}

// out[c] = sum_b S[b, c]   (pred_bias gradient of the 1-vs-all route)
__global__ __launch_bounds__(256) void k_tr_col_sum_f32(const float* __restrict__ S, int64_t rows, int64_t cols, float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float a = 0.f;
  for (int64_t b = 0; b < rows; ++b) a += S[b * cols + c];
  out[c] = a;
}

// ---- g_MLP generator chain pieces (small matrices: B <= a few thousand, widths <= a few hundred)
// out[b,j] = sum_i in[b,i] P[i,j]
__global__ __launch_bounds__(256) void k_tr_small_mm(const float* __restrict__ in, const float* __restrict__ P, int64_t B, int ni, int nj,
                                                     float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * nj) return;
  const int64_t b = idx / nj;
  const int j = (int)(idx % nj);
  float a = 0.f;
  for (int i = 0; i < ni; ++i) a = fmaf(in[b * ni + i], P[(int64_t)i * nj + j], a);
  out[idx] = a;
}
// dv[b,i] (+)= sum_j du[b,j] P[i,j]
__global__ __launch_bounds__(256) void k_tr_small_mm_nt(const float* __restrict__ du, const float* __restrict__ P, int64_t B, int ni, int nj,
                                                        int accumulate, float* __restrict__ dv) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * ni) return;
  const int64_t b = idx / ni;
  const int i = (int)(idx % ni);
  float a = accumulate ? dv[idx] : 0.f;
  for (int j = 0; j < nj; ++j) a = fmaf(du[b * nj + j], P[(int64_t)i * nj + j], a);
  dv[idx] = a;
}
// dP[i,j] = sum_b v[b,i] du[b,j]   (one thread per entry; B is small)
__global__ __launch_bounds__(256) void k_tr_small_mm_tn(const float* __restrict__ v, const float* __restrict__ du, int64_t B, int ni, int nj,
                                                        float* __restrict__ dP) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)ni * nj) return;
  const int i = (int)(idx / nj), j = (int)(idx % nj);
  float a = 0.f;
  for (int64_t b = 0; b < B; ++b) a = fmaf(v[b * ni + i], du[b * nj + j], a);
  dP[idx] = a;
}
// a = relu(BN(u)) (or relu(u) when the generator has no BN); v_next = dropout(a)
__global__ __launch_bounds__(256) void k_tr_chain_act(const float* __restrict__ u, const float* __restrict__ st, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int n, int64_t total, uint32_t seed, uint32_t step,
                                                      uint32_t stage, uint32_t thr, float keep_scale, float* __restrict__ a,
                                                      float* __restrict__ vnext) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % n);
  float v = u[i];
  if (gamma) v = (v - st[k]) * st[n + k] * gamma[k] + beta[k];
  v = v > 0.f ? v : 0.f;
  a[i] = v;
  vnext[i] = dropout_keep_u32(seed, step, stage, (uint32_t)i, thr) ? v * keep_scale : 0.f;
}
// g = keep * dv_next / (1 - rate)   (dropout backward; the ReLU / BN part is k_tr_fcbn_bwd)
__global__ __launch_bounds__(256) void k_tr_chain_drop_bwd(const float* __restrict__ dvn, int64_t total, uint32_t seed, uint32_t step,
                                                           uint32_t stage, uint32_t thr, float keep_scale, float* __restrict__ g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  g[i] = dropout_keep_u32(seed, step, stage, (uint32_t)i, thr) ? dvn[i] * keep_scale : 0.f;
}
__global__ __launch_bounds__(256) void k_tr_add(const float* __restrict__ a, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] += a[i];
}

// ---- g_lookup dense layer (ParameterLookup, models.py:79-94): W = table[rel[b]] of shape [F, d], per sample
// forward partials: part[sl][b][k] = sum_{f in slice sl} x[b,f] W[rel[b]][f,k]     (grid: B x NSL, thread = k)
__global__ __launch_bounds__(256) void k_tr_lookup_fwd(const float* __restrict__ x, const float* __restrict__ Wt,
                                                       const int64_t* __restrict__ rel, int64_t R, int64_t F, int d, int nsl,
                                                       int64_t B, float* __restrict__ part) {
  extern __shared__ float xs[];
  const int64_t b = blockIdx.x;
  const int sl = blockIdx.y;
  const int64_t f0 = F * sl / nsl, f1 = F * (sl + 1) / nsl;
  int64_t rid = rel[b];
  if (rid < 0 || rid >= R) rid = 0;
  for (int64_t f = f0 + threadIdx.x; f < f1; f += 256) xs[f - f0] = x[b * F + f];
  __syncthreads();
  const int k = threadIdx.x;
  if (k >= d) return;
  const float* W = Wt + (rid * F + f0) * d + k;
  float a = 0.f;
  for (int64_t f = 0; f < f1 - f0; ++f) a = fmaf(xs[f], W[f * d], a);
  part[((int64_t)sl * B + b) * d + k] = a;
}
// dW[r][f,k] = sum_{b: rel[b] = r} x[b,f] dz[b,k]   (grid: F-chunks x R; relations absent from the batch are skipped:
// their rows keep stale values and the optimiser kernels treat them as zero through the row mask)
__global__ __launch_bounds__(256) void k_tr_lookup_dW(const float* __restrict__ x, const float* __restrict__ dz,
                                                      const int32_t* __restrict__ perm, const int32_t* __restrict__ offset,
                                                      const int32_t* __restrict__ count, int64_t F, int d, int rows_per_wg,
                                                      float* __restrict__ dWt) {
  const int64_t r = blockIdx.y;
  const int n = count[r];
  if (n == 0) return;
  const int off = offset[r];
  const int k = threadIdx.x;
  if (k >= d) return;
  const int64_t f0 = (int64_t)blockIdx.x * rows_per_wg;
  for (int64_t f = f0; f < f0 + rows_per_wg && f < F; ++f) {
    float a = 0.f;
    for (int j = 0; j < n; ++j) {
      const int64_t b = perm[off + j];
      a = fmaf(x[b * F + f], dz[b * d + k], a);
    }
    dWt[(r * F + f) * d + k] = a;
  }
}
// dx[b,f] = sum_k dz[b,k] W[rel[b]][f,k]   (one wave per row f, lanes over k)
__global__ __launch_bounds__(256) void k_tr_lookup_dx(const float* __restrict__ dz, const float* __restrict__ Wt,
                                                      const int64_t* __restrict__ rel, int64_t R, int64_t F, int d,
                                                      float* __restrict__ dx) {
  extern __shared__ float dzs[];
  const int64_t b = blockIdx.y;
  int64_t rid = rel[b];
  if (rid < 0 || rid >= R) rid = 0;
  for (int k = threadIdx.x; k < d; k += 256) dzs[k] = dz[b * d + k];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t f = (int64_t)blockIdx.x * 64 + wave; f < F && f < (int64_t)(blockIdx.x + 1) * 64; f += 4) {
    const float* W = Wt + (rid * F + f) * d;
    float a = 0.f;
    for (int k = lane; k < d; k += 64) a = fmaf(dzs[k], W[k], a);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) dx[b * F + f] = a;
  }
}
// z1 = keep * (sum of the forward partials + bias_table[rel[b]]) / (1 - rate)
__global__ __launch_bounds__(256) void k_tr_lookup_post(const float* __restrict__ part, int nsl, const float* __restrict__ bias_t,
                                                        const int64_t* __restrict__ rel, int64_t R, int d, int64_t total,
                                                        uint32_t seed, uint32_t step, uint32_t thr, float keep_scale,
                                                        float* __restrict__ z1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int64_t rid = rel[i / d];
  if (rid < 0 || rid >= R) rid = 0;
  float v = bias_t[rid * d + i % d];
  for (int sl = 0; sl < nsl; ++sl) v += part[(int64_t)sl * total + i];
  z1[i] = dropout_keep_u32(seed, step, 2u, (uint32_t)i, thr) ? v * keep_scale : 0.f;
}
// dz0 = keep * dz1 / (1 - rate) (in place);  dbias_table[rel[b], k] += dz0[b,k]
__global__ __launch_bounds__(256) void k_tr_lookup_post_bwd(float* __restrict__ dz, const int64_t* __restrict__ rel, int64_t R, int d,
                                                            int64_t total, uint32_t seed, uint32_t step, uint32_t thr,
                                                            float keep_scale, float* __restrict__ dbias_t) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int64_t rid = rel[i / d];
  if (rid < 0 || rid >= R) rid = 0;
  const float v = dropout_keep_u32(seed, step, 2u, (uint32_t)i, thr) ? dz[i] * keep_scale : 0.f;
  dz[i] = v;
  atomicAdd(&dbias_t[rid * d + i % d], v);
}

// FCBN backward, one workgroup per feature k (a column of [B, d]): gamma/beta gradients and dz1
// (dh and dz1 may be the same buffer: every element is read, then written, by one thread)
__global__ __launch_bounds__(256) void k_tr_fcbn_bwd(const float* __restrict__ z1, const float* __restrict__ hv, const float* dh,
                                                     const float* __restrict__ mean, const float* __restrict__ inv,
                                                     const float* __restrict__ gamma, int64_t B, int d, int use_batch,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* dz1) {
  __shared__ double s1[256], s2[256];
  const int k = blockIdx.x;
  if (!mean) {   // generator layer without BN: ReLU only
    for (int64_t b = threadIdx.x; b < B; b += 256) dz1[b * d + k] = hv[b * d + k] > 0.f ? dh[b * d + k] : 0.f;
    return;
  }
  const float mu = mean[k], iv = inv[k], ga = gamma[k];
  double a1 = 0, a2 = 0;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float g = hv[b * d + k] > 0.f ? dh[b * d + k] : 0.f;   // through the ReLU
    const float zh = (z1[b * d + k] - mu) * iv;
    a1 += g;
    a2 += (double)g * zh;
  }
  s1[threadIdx.x] = a1; s2[threadIdx.x] = a2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  const double S1 = s1[0], S2 = s2[0];
  if (threadIdx.x == 0) { dbeta[k] = (float)S1; dgamma[k] = (float)S2; }
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float g = hv[b * d + k] > 0.f ? dh[b * d + k] : 0.f;
    const float zh = (z1[b * d + k] - mu) * iv;
    float dz;
    if (use_batch) dz = (float)((double)ga * iv * ((double)g - S1 / (double)B - (double)zh * S2 / (double)B));
    else dz = ga * iv * g;
    dz1[b * d + k] = dz;
  }
}

// dz0 = keep * dz1 / (1 - rate) (in place); dense-bias gradients: static dfc_bias[k] += dz0; generated
// dPb[rho,k] += c[b,rho] dz0[b,k], dc[b,rho] = sum_k dz0[b,k] Pb[rho,k]
__global__ __launch_bounds__(256) void k_tr_fc_post_bwd(float* __restrict__ dz, const float* __restrict__ c,
                                                        const float* __restrict__ Pb, int r, int d, uint32_t seed,
                                                        uint32_t step, uint32_t thr, float keep_scale,
                                                        float* __restrict__ dfc_bias, float* __restrict__ dPb,
                                                        float* __restrict__ dc) {
  extern __shared__ float row[];  // dz0[b, :]
  const int64_t b = blockIdx.x;
  for (int k = threadIdx.x; k < d; k += 256) {
    const int64_t i = b * d + k;
    const float v = dropout_keep_u32(seed, step, 2u, (uint32_t)i, thr) ? dz[i] * keep_scale : 0.f;
    dz[i] = v;
    row[k] = v;
    if (!Pb) atomicAdd(&dfc_bias[k], v);
  }
  __syncthreads();
  if (Pb) {
    // (dPb[rho, k] = sum_b c[b, rho] dz0[b, k] is a small product of its own behind this kernel since round 6: B workgroups adding
    //  r d values each to the same r d addresses took 32 us)
    for (int rho = threadIdx.x; rho < r; rho += 256) {
      float a = 0.f;
      for (int k = 0; k < d; ++k) a = fmaf(row[k], Pb[rho * d + k], a);
      dc[b * r + rho] = a;
    }
  }
}

// dT[rho][b,k] = c[b,rho] dz[b,k]   (operand of the batched weight-gradient GEMM dP[rho] = x^T dT[rho])
__global__ __launch_bounds__(256) void k_tr_scale_rows(const float* __restrict__ dz, const float* __restrict__ c, int r, int d,
                                                       int64_t total, float* __restrict__ dT, unsigned* __restrict__ max_slots) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float mx = 0.f;
  if (i < total) {
    const int64_t b = i / d;
    const float g = dz[i];
    for (int rho = 0; rho < r; ++rho) {
      const float v = c[b * r + rho] * g;
      dT[(int64_t)rho * total + i] = v;
      mx = fmaxf(mx, fabsf(v));
    }
  }
  if (max_slots) tr_block_max_to_slot(mx, max_slots);
}

// dx[b,f] = sum_rho c[b,rho] dA[b,rho*F+f];  dc[b,rho] += sum_f x[b,f] dA[b,rho*F+f]
// dx[b, f] = sum_rho c[b, rho] dA[b, rho, f]   (one pass over dA, eight loads in flight)

// dc[b, rho] = sum_k dz[b, k] T[rho][b][k]: z[b] = sum_rho c[b, rho] T[rho][b] with the forward partials T = x P[rho]
// still in place, so the context gradient needs no second pass over dA.  One wave per (b, rho).
__global__ __launch_bounds__(256) void k_tr_dc_from_partials(const float* __restrict__ dz, const float* __restrict__ Tf, int64_t B, int r,
                                                             int d, float* __restrict__ dc) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B * r) return;
  const int64_t b = w / r;
  const int rho = (int)(w - b * r);
  const float* t = Tf + ((int64_t)rho * B + b) * d;
  const float* g = dz + b * d;
  float a = 0.f;
  for (int k = lane; k < d; k += 64) a = fmaf(g[k], t[k], a);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) dc[b * r + rho] = a;
}

// Conv1BN backward, pass 1: g = dx * keep/(1-rate) through the ReLU; per-channel sums of g and g*yhat
// (dbeta, dgamma).  g overwrites dx.
// NS > 0 (round 6): dx arrives as the NS K slices of its product (tg_gemm_nt: leave_slices), added here in slice order -- the slice sum's
// launch, its store of dx and this kernel's read of it are gone.
template <int NS>
__global__ __launch_bounds__(256) void k_tr_bn1_bwd_sums(float* __restrict__ dx, const float* __restrict__ part, const float* __restrict__ y,
                                                         const float* __restrict__ mean, const float* __restrict__ inv,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                         int64_t total, uint32_t seed, uint32_t step, uint32_t thr,
                                                         float keep_scale, double* __restrict__ sums, int nslots) {
  __shared__ double s1[256], s2[256];
  sums += (size_t)(blockIdx.x % nslots) * 2 * C;      // (TR_CS_SLOTS copies: k_tr_col_sums)
  auto dx_in = [&](int64_t i) -> float {
    if constexpr (NS == 0) {
      return dx[i];
    } else {
      float t[NS];
#pragma unroll
      for (int z = 0; z < NS; ++z) t[z] = part[(int64_t)z * total + i];
      float a = 0.f;
#pragma unroll
      for (int z = 0; z < NS; ++z) a += t[z];      // slice order, as k_tg_reduce
      return a;
    }
  };
  if (256 % C != 0) {
    // channel counts that do not divide the workgroup: a thread meets every channel, so the channel sums are built in LDS
    // (C <= 256 doubles per array) with one LDS atomic pair per element, then added to the global sums
    for (int c = threadIdx.x; c < C; c += 256) { s1[c] = 0; s2[c] = 0; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int c = (int)(i % C);
      const float yh = (y[i] - mean[c]) * inv[c];
      const float act = yh * gamma[c] + beta[c];
      float g = dropout_keep_u32(seed, step, 1u, (uint32_t)i, thr) ? dx_in(i) * keep_scale : 0.f;
      if (!(act > 0.f)) g = 0.f;
      dx[i] = g;
      atomicAdd(&s1[c], (double)g);
      atomicAdd(&s2[c], (double)g * yh);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      atomicAdd(&sums[c], s1[c]);
      atomicAdd(&sums[C + c], s2[c]);
    }
    return;
  }
  double a1 = 0, a2 = 0;
  // grid-stride: 256 % C == 0, so a thread stays on one channel and the channel sums are built in registers
  const int64_t st = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i % C);
  const float mc = mean[c], ic = inv[c], gc = gamma[c], bc = beta[c];
  for (; i + 3 * st < total; i += 4 * st) {          // four load pairs in flight (one at a time: 20 dependent round trips, 40 us)
    float yv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { yv[u] = y[i + u * st]; dv[u] = dx_in(i + u * st); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float yh = (yv[u] - mc) * ic;
      const float act = yh * gc + bc;
      float g = dropout_keep_u32(seed, step, 1u, (uint32_t)(i + u * st), thr) ? dv[u] * keep_scale : 0.f;
      if (!(act > 0.f)) g = 0.f;
      dx[i + u * st] = g;
      a1 += g;
      a2 += (double)g * yh;
    }
  }
  for (; i < total; i += st) {
    const float yh = (y[i] - mc) * ic;
    const float act = yh * gc + bc;
    float g = dropout_keep_u32(seed, step, 1u, (uint32_t)i, thr) ? dx_in(i) * keep_scale : 0.f;
    if (!(act > 0.f)) g = 0.f;
    dx[i] = g;
    a1 += g;
    a2 += (double)g * yh;
  }
  // 256 % C == 0 here: threads with the same (threadIdx.x % C) share a channel
  s1[threadIdx.x] = a1; s2[threadIdx.x] = a2;
  __syncthreads();
  for (int o = 128; o >= C; o >>= 1) {
    if ((int)threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if ((int)threadIdx.x < C) {
    const int c = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % C);
    atomicAdd(&sums[c], s1[threadIdx.x]);
    atomicAdd(&sums[C + c], s2[threadIdx.x]);
  }
}

// pass 2: dy = gamma*inv*(g - S1/n - yhat*S2/n) (batch statistics) or gamma*inv*g; in place on dx
__global__ __launch_bounds__(256) void k_tr_bn1_bwd_apply(float* __restrict__ dx, const float* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ inv,
                                                          const float* __restrict__ gamma, const double* __restrict__ sums, int C,
                                                          int64_t total, double n, int use_batch, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < C) { dbeta[i] = (float)sums[i]; dgamma[i] = (float)sums[C + i]; }
  if (i >= total) return;
  const int c = (int)(i % C);
  const float g = dx[i];
  if (use_batch) {
    const float yh = (y[i] - mean[c]) * inv[c];
    dx[i] = (float)((double)gamma[c] * inv[c] * ((double)g - sums[c] / n - (double)yh * sums[C + c] / n));
  } else {
    dx[i] = gamma[c] * inv[c] * g;
  }
}

// conv backward, one workgroup per query: the query's filter / bias gradients dK_ps[b], dkb_ps[b] (reduced by the caller: through the
// generators / tables, or -- static filters -- by column sums), d(img) -> rows of dE / drel_emb
__global__ __launch_bounds__(256) void k_tr_conv_bwd(const float* __restrict__ dy, const float* __restrict__ img_all,
                                                     const float* __restrict__ K, const int64_t* __restrict__ e1,
                                                     const int64_t* __restrict__ rel, int64_t E, int64_t R, int d, int r,
                                                     int in_h, int in_w, int stacked, int C, int Ho, int Wo,
                                                     float* __restrict__ dE, float* __restrict__ drel,
                                                     const float* __restrict__ K_ps, float* __restrict__ dK_ps,
                                                     float* __restrict__ dkb_ps, int fh, int fw) {
  extern __shared__ float lds[];  // img[isz] | g[P][C + 1] | taps[fh*fw*C]
  // (round 6: a pixel's C gradients are C + 1 words apart -- the image-gradient loop below has every lane on another PIXEL and the
  //  same channel: with a stride of C = 32 words all 64 lanes sat on one LDS bank, 45 of this kernel's 60 us)
  const int isz = in_h * in_w, P = Ho * Wo, nt = fh * fw, CS = C + 1;
  float* img = lds;
  float* g = img + isz;
  float* taps = g + P * CS;
  const int64_t b = blockIdx.x;
  for (int t = threadIdx.x; t < isz; t += 256) img[t] = img_all[b * isz + t];
  for (int t = threadIdx.x; t < P * C; t += 256) g[(t / C) * CS + (t % C)] = dy[b * (int64_t)P * C + t];
  const float* Ksrc = K_ps ? K_ps + b * (int64_t)nt * C : K;
  for (int t = threadIdx.x; t < nt * C; t += 256) taps[t] = Ksrc[t];
  __syncthreads();
  // filter and bias gradients: entry (tap, c) = sum_p img[p + tap offset] * g[p, c]
  for (int idx = threadIdx.x; idx < (nt + 1) * C; idx += 256) {
    const int cc = idx % C, tap = idx / C;
    float a = 0.f;
    if (tap < nt) {
      const int u = tap / fw, v = tap % fw;
      // (the trip counts are run-time values: without the unrolls every iteration waits out its own two LDS reads -- with two
      //  waves per SIMD this kernel was LDS latency end to end, 31 us)
      for (int i = 0; i < Ho; ++i) {
        const float* ir = img + (i + u) * in_w + v;
        const float* gr = g + (i * Wo) * CS + cc;
#pragma unroll 6
        for (int j = 0; j < Wo; ++j) a = fmaf(ir[j], gr[j * CS], a);
      }
      dK_ps[b * (int64_t)nt * C + tap * C + cc] = a;
    } else {
#pragma unroll 8
      for (int p = 0; p < P; ++p) a += g[p * CS + cc];
      dkb_ps[b * C + cc] = a;
    }
  }
  // image gradient (full correlation), scattered to the embedding rows
  int64_t row = e1[b];
  if (row < 0 || row >= E) row = 0;
  int64_t rid = rel[b];
  if (rid < 0 || rid >= R) rid = 0;
  for (int t = threadIdx.x; t < isz; t += 256) {
    const int ii = t / in_w, jj = t - ii * in_w;
    float a = 0.f;
    for (int u = 0; u < fh; ++u) {
      const int i = ii - u;
      if (i < 0 || i >= Ho) continue;
      for (int v = 0; v < fw; ++v) {
        const int j = jj - v;
        if (j < 0 || j >= Wo) continue;
        const float* gp = g + (i * Wo + j) * CS;
        const float* tp = taps + (u * fw + v) * C;
#pragma unroll 8
        for (int cc = 0; cc < C; ++cc) a = fmaf(gp[cc], tp[cc], a);
      }
    }
    if (t < d) atomicAdd(&dE[row * d + t], a);
    else if (stacked) atomicAdd(&drel[rid * r + (t - d)], a);
  }
}

// out[b, :] = table[rel[b], :]   (relation rows; also the per-sample conv filters of g_lookup)
__global__ __launch_bounds__(256) void k_tr_gather_rows(const float* __restrict__ table, const int64_t* __restrict__ rel, int64_t R, int n,
                                                        int64_t total, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int64_t rid = rel[i / n];
  if (rid < 0 || rid >= R) rid = 0;
  out[i] = table[rid * n + i % n];
}

// concat_rel (models.py:406-407): xc[b] = [x[b] | c[b]], after the hidden dropout
__global__ __launch_bounds__(256) void k_tr_concat(const float* __restrict__ x, const float* __restrict__ c, int64_t Fc, int r, int64_t total,
                                                   float* __restrict__ xc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / (Fc + r), f = i - b * (Fc + r);
  xc[i] = f < Fc ? x[b * Fc + f] : c[b * r + (f - Fc)];
}

// ... and back: dx[b] = dxc[b, :Fc];  drel_emb[rel[b], :] += dxc[b, Fc:]
__global__ __launch_bounds__(256) void k_tr_split(const float* __restrict__ dxc, const int64_t* __restrict__ rel, int64_t R, int64_t Fc, int r,
                                                  int64_t total, float* __restrict__ dx, float* __restrict__ drel) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / (Fc + r), f = i - b * (Fc + r);
  if (f < Fc) { dx[b * Fc + f] = dxc[i]; return; }
  int64_t rid = rel[b];
  if (rid < 0 || rid >= R) rid = 0;
  atomicAdd(&drel[rid * r + (f - Fc)], dxc[i]);
}

// drel_emb[rel[b], :] += dc[b, :]
__global__ __launch_bounds__(256) void k_tr_scatter_rows(const float* __restrict__ dc, const int64_t* __restrict__ rel, int64_t R,
                                                         int r, int64_t total, float* __restrict__ drel) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int64_t rid = rel[i / r];
  if (rid < 0 || rid >= R) rid = 0;
  atomicAdd(&drel[rid * r + i % r], dc[i]);
}

// every trainable tensor in one launch (blockIdx.y = tensor): the small ones would otherwise cost a launch each
constexpr int TR_MAX_PARAMS = 40;
constexpr int TR_EXP_CACHE = 8;
struct TrainTensors {
  float* p[TR_MAX_PARAMS];
  float* g[TR_MAX_PARAMS];
  float* m[TR_MAX_PARAMS];
  float* v[TR_MAX_PARAMS];
  float* vh[TR_MAX_PARAMS];
  int64_t n[TR_MAX_PARAMS];
  // tensors whose gradient rows exist only for keys present in the batch (g_lookup tables): row length and the
  // per-key count; a row with count 0 has gradient 0 whatever the buffer holds
  int64_t rowlen[TR_MAX_PARAMS];
  const int32_t* rowcnt[TR_MAX_PARAMS];
  unsigned* wmax;     // TG_MAX_SLOTS slots for max |p_new| of tensor wmax_of (-1: none)
  int wmax_of;
};

__global__ __launch_bounds__(256) void k_tr_sumsq(TrainTensors tt, int skip, double* __restrict__ acc) {
  __shared__ double part[256];
  if ((int)blockIdx.y == skip) return;   // the GEMM that produced this gradient already added its squares
  const float* g = tt.g[blockIdx.y];
  const int64_t n = tt.n[blockIdx.y];
  double a = 0;
  const int32_t* rc = tt.rowcnt[blockIdx.y];
  const int64_t rl = tt.rowlen[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (!rc && ((uintptr_t)g & 15) == 0) {
    // 16-byte loads, four in flight per thread; the scalar loop below takes what is left
    const int64_t n4 = n >> 2;
    const float4* g4 = (const float4*)g;
    int64_t j = i;
    for (; j + 3 * stride < n4; j += 4 * stride) {
      const float4 v0 = g4[j], v1 = g4[j + stride], v2 = g4[j + 2 * stride], v3 = g4[j + 3 * stride];
      a += (double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w;
      a += (double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w;
      a += (double)v2.x * v2.x + (double)v2.y * v2.y + (double)v2.z * v2.z + (double)v2.w * v2.w;
      a += (double)v3.x * v3.x + (double)v3.y * v3.y + (double)v3.z * v3.z + (double)v3.w * v3.w;
    }
    for (; j < n4; j += stride) {
      const float4 v0 = g4[j];
      a += (double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w;
    }
    i += n4 * 4;   // the scalar loop: elements [4 n4, n)
  } else if (!rc) {
    for (; i + 3 * stride < n; i += 4 * stride) {   // four independent loads in flight
      const float g0 = g[i], g1 = g[i + stride], g2 = g[i + 2 * stride], g3 = g[i + 3 * stride];
      a += (double)g0 * g0 + (double)g1 * g1 + (double)g2 * g2 + (double)g3 * g3;
    }
  }
  for (; i < n; i += stride) {
    if (rc && rc[i / rl] == 0) continue;
    a += (double)g[i] * g[i];
  }
  part[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && part[0] != 0.0) atomicAdd(acc + (blockIdx.x + blockIdx.y * gridDim.x) % TG_SUMSQ_SLOTS, part[0]);
}

// tf.clip_by_global_norm + AMSGrad (amsgrad.py:130-159), all in one pass over the parameters
__global__ __launch_bounds__(256) void k_tr_amsgrad(TrainTensors tt, const double* __restrict__ ssq, double* __restrict__ total,
                                                    float clip, float lr_t, float b1, float b2, float eps) {
  float* p = tt.p[blockIdx.y];
  const float* g = tt.g[blockIdx.y];
  float* m = tt.m[blockIdx.y];
  float* v = tt.v[blockIdx.y];
  float* vh = tt.vh[blockIdx.y];
  const int64_t n = tt.n[blockIdx.y];
  double ss = 0;   // the slots in index order: every thread of every workgroup forms the same sum
  for (int i = 0; i < TG_SUMSQ_SLOTS; ++i) ss += ssq[i];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *total = ss;   // what coper_train_grad reports
  const double gn = sqrt(ss);
  const float scale = (float)((double)clip / (gn > (double)clip ? gn : (double)clip));
  const int32_t* rc = tt.rowcnt[blockIdx.y];
  const int64_t rl = tt.rowlen[blockIdx.y];
  // dense tensors (every gradient row exists): four elements per thread and step as 16-byte accesses -- nine streams of 4 bytes
  // per element, 1.17 GB per step at the FB15k-237 shapes; element by element (and an int64 division per element for the
  // row-count test that only the looked-up tables need) the pass ran at 5.4 TB/s
  if (!rc && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vh) & 15) == 0) {
    const int64_t n4 = n >> 2;
    unsigned pmx = 0u;       // the largest |p_new| this thread wrote (bit pattern): the next step's packs take their power of two from it
    auto upd4 = [&](const float4& g4, float4& m4, float4& v4, float4& h4, float4& p4) {
      float* gm = (float*)&m4; float* gv = (float*)&v4; float* gh = (float*)&h4; float* gp = (float*)&p4; const float* gg = (const float*)&g4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gi = gg[j] * scale;
        const float mi = b1 * gm[j] + (1.f - b1) * gi;
        const float vi = b2 * gv[j] + (1.f - b2) * gi * gi;
        const float vhi = fmaxf(gh[j], vi);
        gm[j] = mi; gv[j] = vi; gh[j] = vhi;
        gp[j] -= lr_t * mi / (sqrtf(vhi) + eps);
        const unsigned pb = __float_as_uint(gp[j]) & 0x7fffffffu;
        pmx = pb > pmx ? pb : pmx;
      }
    };
    const int64_t st = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // two elements-of-four per thread and trip: ten 16-byte loads in flight (round 6: five reached 4.9 TB/s on the 1.17 GB of a step)
    // the three slot streams (m, v, v_hat: read once and written once per step, 0.78 GB of the 1.17) with the non-temporal policy:
    // they pass the caches without displacing the gradient the products just wrote and the parameter the packs read next
    // (round 6 A/B, three workloads: - 25 ... - 35 us per step)
#ifndef COPER_DBG_AMSGRAD_NO_NT
    typedef float f4v __attribute__((ext_vector_type(4)));
#define AMS_LD(p_, i_) ([&] { const f4v t = __builtin_nontemporal_load((const f4v*)(p_) + (i_)); return make_float4(t.x, t.y, t.z, t.w); }())
#define AMS_ST(p_, i_, v_) __builtin_nontemporal_store(f4v{(v_).x, (v_).y, (v_).z, (v_).w}, (f4v*)(p_) + (i_))
#else
#define AMS_LD(p_, i_) (((const float4*)(p_))[i_])
#define AMS_ST(p_, i_, v_) (((float4*)(p_))[i_] = (v_))
#endif
    for (; i + st < n4; i += 2 * st) {
      const float4 ga = ((const float4*)g)[i], gb = ((const float4*)g)[i + st];
      float4 ma = AMS_LD(m, i), va = AMS_LD(v, i), ha = AMS_LD(vh, i), pa = ((const float4*)p)[i];
      float4 mb = AMS_LD(m, i + st), vb = AMS_LD(v, i + st), hb = AMS_LD(vh, i + st), pb4 = ((const float4*)p)[i + st];
      upd4(ga, ma, va, ha, pa);
      upd4(gb, mb, vb, hb, pb4);
      AMS_ST(m, i, ma); AMS_ST(v, i, va); AMS_ST(vh, i, ha); ((float4*)p)[i] = pa;
      AMS_ST(m, i + st, mb); AMS_ST(v, i + st, vb); AMS_ST(vh, i + st, hb); ((float4*)p)[i + st] = pb4;
    }
#undef AMS_LD
#undef AMS_ST
    for (; i < n4; i += st) {
      const float4 g4 = ((const float4*)g)[i];
      float4 m4 = ((const float4*)m)[i], v4 = ((const float4*)v)[i], h4 = ((const float4*)vh)[i], p4 = ((const float4*)p)[i];
      upd4(g4, m4, v4, h4, p4);
      ((float4*)m)[i] = m4; ((float4*)v)[i] = v4; ((float4*)vh)[i] = h4; ((float4*)p)[i] = p4;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const float gi = g[i] * scale;
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      const float vhi = fmaxf(vh[i], vi);
      m[i] = mi; v[i] = vi; vh[i] = vhi;
      const float pn = p[i] - lr_t * mi / (sqrtf(vhi) + eps);
      p[i] = pn;
      const unsigned pb = __float_as_uint(pn) & 0x7fffffffu;
      pmx = pb > pmx ? pb : pmx;
    }
    if (tt.wmax && tt.wmax_of == (int)blockIdx.y) {     // one atomic per wave, spread over the slots (8 per slot at 2,048 workgroups)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const unsigned u = __shfl_xor(pmx, o, 64); pmx = u > pmx ? u : pmx; }
      if ((threadIdx.x & 63) == 0 && pmx) atomicMax(tt.wmax + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (TG_MAX_SLOTS - 1)), pmx);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = (rc && rc[i / rl] == 0) ? 0.f : g[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float vhi = fmaxf(vh[i], vi);
    m[i] = mi; v[i] = vi; vh[i] = vhi;
    p[i] -= lr_t * mi / (sqrtf(vhi) + eps);
  }
}

__global__ void k_tr_store_loss(const double* __restrict__ acc, double inv_BL, float* __restrict__ out) { out[0] = (float)(acc[0] * inv_BL); }

// GEMM on planes packed by the caller, K cut into slices when the output has few tiles (the partial-sum pool grows on demand)
// slices_left: when not null and K was cut, the partial sums stay in T->mmP ([*slices_left][M][N]) for the caller's next kernel and C is
// not written (*slices_left = 1: C holds the product)
static int tg_gemm_split(coper_handle* h, TrainState* T, hipStream_t s, TgPlanes X, int64_t M, TgPlanes Y, int64_t N, int64_t K, float* C,
                         TgIdx ci, TgIdx cj, int* slices_left = nullptr) {
  int rc;
  const int nsplit = tg_split_k(M, N, K);
  if (slices_left) *slices_left = nsplit > 1 && nsplit <= 8 ? nsplit : 1;
  const size_t np = nsplit > 1 ? (size_t)nsplit * M * N : 0;
  if (np > T->mmP_cap) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    if ((rc = talloc(h, &T->mmP, np))) return rc;
    T->mmP_cap = np;
  }
  return tg_gemm_nt(h, X, M, Y, N, K, C, ci, cj, s, nsplit, T->mmP, nullptr, slices_left && *slices_left > 1);
}

static int tg_matmul(coper_handle* h, TrainState* T, hipStream_t s, const MmView& X, int64_t M, const MmView& Y, int64_t N, int64_t K,
                     float* C, TgIdx ci, TgIdx cj, double* sumsq, const unsigned* x_slots, const unsigned* y_slots) {
  int rc;
  const size_t nx = tg_plane_elems(M, K), ny = tg_plane_elems(N, K);
  const int nsplit = tg_split_k(M, N, K);
  const size_t np = nsplit > 1 ? (size_t)nsplit * M * N : 0;
  const bool on_side = T->side[0] && s == T->side[0];      // (the caller made sure that no K slices are needed there: one partial-sum pool)
  TgPlanes& MX = on_side ? T->mmX2 : T->mmX;
  TgPlanes& MY = on_side ? T->mmY2 : T->mmY;
  size_t& capX = on_side ? T->mmX2_cap : T->mmX_cap;
  size_t& capY = on_side ? T->mmY2_cap : T->mmY_cap;
  if (on_side && nsplit > 1) return fail(h, COPER_ESTATE, "tg_matmul: a product with K slices on the side stream");
  if (nx > capX || ny > capY || np > T->mmP_cap) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    if (nx > capX) {
      if ((rc = talloc(h, &MX.hi, nx)) || (rc = talloc(h, &MX.lo, nx))) return rc;
      capX = nx;
    }
    if (ny > capY) {
      if ((rc = talloc(h, &MY.hi, ny)) || (rc = talloc(h, &MY.lo, ny))) return rc;
      capY = ny;
    }
    if (np > T->mmP_cap) {
      if ((rc = talloc(h, &T->mmP, np))) return rc;
      T->mmP_cap = np;
    }
  }
  // an operand tensor packed earlier in this step keeps its power of two (its own word from the pool behind the plane sets' eight):
  // no second pass for the maximum.  A tensor with producer-side maxima (slots) needs no pass at all.
  TgPlanes px = MX, py = MY;
  unsigned* const scratch = on_side ? T->tg_scratch + 2 : T->tg_scratch;
  auto pack = [&](const MmView& V, int64_t rows, TgPlanes& pl, const unsigned* slots) -> int {
    const int32_t* from = nullptr;
    for (auto& e : T->exp_cache)
      if (e.first == V.p) from = e.second;
    if (from) {
      pl.exp = const_cast<int32_t*>(from);
      return tg_pack(h, V.p, V.ri, V.ki, rows, K, tg_rows_pad(rows), V.rows_fast, pl, s, scratch, from);
    }
    if ((int)T->exp_cache.size() < TR_EXP_CACHE) {
      pl.exp = T->tg_exps + 10 + T->exp_cache.size();
      T->exp_cache.emplace_back(V.p, pl.exp);
    }
    return tg_pack(h, V.p, V.ri, V.ki, rows, K, tg_rows_pad(rows), V.rows_fast, pl, s, scratch, nullptr, slots);
  };
  if ((rc = pack(X, M, px, x_slots)) || (rc = pack(Y, N, py, y_slots))) return rc;
  return tg_gemm_nt(h, px, M, py, N, K, C, ci, cj, s, nsplit, T->mmP, sumsq);
}


}  // namespace

// coper_set_param: whatever the optimizer's last pass knew about the parameters (TrainState::wmax) no longer describes them
void train_params_changed(coper_handle* h) {
  if (h->train) ((TrainState*)h->train)->wmax_valid = false;
}

void train_destroy(coper_handle* h) {
  TrainState* T = (TrainState*)h->train;
  if (!T) return;
  for (auto& t : T->tp) { (void)tracked_free(t.g); (void)tracked_free(t.m); (void)tracked_free(t.v); (void)tracked_free(t.vh); }
  for (TgPlanes* pl : {&T->pX, &T->pXt, &T->pP1, &T->pP3, &T->pTn, &T->pTb}) {
    if (pl->hi) (void)tracked_free(pl->hi);
    if (pl->lo) (void)tracked_free(pl->lo);
  }
  float* bufs[] = {T->Kt, T->Kbv, T->dKs, T->dkbs, T->Sd, T->img, T->y, T->x, T->c, T->A, T->dA, T->z0, T->z1, T->hv, T->dh, T->dz, T->ds, T->dx, T->dc, T->bnst, T->xc, T->dxc};
  for (float* b : bufs) (void)tracked_free(b);
  for (auto& ch : T->chain)
    for (int i = 0; i <= COPER_MAX_CTX; ++i) {
      if (i > 0) (void)tracked_free(ch.v[i]);
      (void)tracked_free(ch.dv[i]);
      if (i < COPER_MAX_CTX) { (void)tracked_free(ch.u[i]); (void)tracked_free(ch.a[i]); (void)tracked_free(ch.du[i]); (void)tracked_free(ch.st[i]); }
    }
  (void)tracked_free(T->red);
  if (T->tg_exps) (void)tracked_free(T->tg_exps);
  if (T->tg_scratch) (void)tracked_free(T->tg_scratch);
  for (hipStream_t& st : T->side)
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
  for (int i = 0; i < 3; ++i) {
    if (T->ev_fork[i]) (void)hipEventDestroy(T->ev_fork[i]);
    if (T->ev_join[i]) (void)hipEventDestroy(T->ev_join[i]);
  }
  for (unsigned* w : {T->wmax[0], T->wmax[1], T->xmax, T->dtmax, T->smax})
    if (w) (void)tracked_free(w);
  for (TgPlanes* pl : {&T->mmX, &T->mmY, &T->mmX2, &T->mmY2}) {
    if (pl->hi) (void)tracked_free(pl->hi);
    if (pl->lo) (void)tracked_free(pl->lo);
  }
  if (T->mmP) (void)tracked_free(T->mmP);
  delete T;
  h->train = nullptr;
}

}  // namespace coper

using namespace coper;

extern "C" {

COPER_API int coper_train_init(coper_handle* h, const coper_train_config* cfg) {
  if (!h || !cfg) return COPER_EINVAL;
  if (cfg->abi_version != COPER_ABI_VERSION) return fail(h, COPER_EINVAL, "coper_train_init: ABI version mismatch");
  if (h->cfg.role != COPER_ROLE_BOTH) return fail(h, COPER_ESTATE, "coper_train_init: training needs a COPER_ROLE_BOTH handle");
  const Dims& dm = h->dm;
  if (h->cfg.shard_lo != 0 || h->cfg.shard_hi != dm.E)
    return fail(h, COPER_EUNSUPPORTED, "coper_train_init: training needs the whole entity table on the handle");
  if (dm.C > 256)
    return fail(h, COPER_EUNSUPPORTED, "coper_train_init: at most 256 conv channels");
  if (!(cfg->learning_rate > 0) || cfg->hidden_dropout < 0 || cfg->hidden_dropout >= 1 || cfg->output_dropout < 0 ||
      cfg->output_dropout >= 1)
    return fail(h, COPER_EINVAL, "coper_train_init: bad hyper-parameter");
  for (auto& sp : h->specs)
    if (!h->params[sp.name].set) return fail(h, COPER_EINVAL, "coper_train_init: parameter not set: " + sp.name);
  COPER_HIP_TRY(h, hipSetDevice(h->cfg.device));
  train_destroy(h);
  TrainState* T = new TrainState();
  h->train = T;
  T->cfg = *cfg;
  T->b1p = cfg->beta1;   // the beta powers start at beta (amsgrad.py:108-113)
  T->b2p = cfg->beta2;
  std::vector<std::string> names = {"ent_emb", "pred_bias", "Conv1BN/gamma", "Conv1BN/beta", "FCBN/gamma", "FCBN/beta"};
  if (!dm.lookup) names.push_back("rel_emb");      // g_lookup has no relation embedding (models.py:210)
  T->nh = (dm.gen_fc && !dm.lookup) ? h->cfg.n_ctx_out : 0;
  T->nhc = (dm.gen_conv && !dm.lookup) ? h->cfg.n_ctx_conv : 0;
  if (dm.gen_conv && !dm.lookup) {
    for (const char* gname : {"conv1_weights", "conv1_bias"})
      for (int i = 0; i <= T->nhc; ++i) {
        std::string pn = std::string(gname) + "/CPG/Projection" + std::to_string(i);
        names.push_back(pn);
        if (i < T->nhc && dm.ctx_bn) { names.push_back(pn + "/BatchNorm/gamma"); names.push_back(pn + "/BatchNorm/beta"); }
      }
    for (int g = 2; g < 4; ++g) {
      T->chain[g].dims[0] = dm.r;
      for (int i = 0; i < T->nhc; ++i) T->chain[g].dims[i + 1] = h->cfg.ctx_conv[i];
    }
  } else {
    names.push_back("conv1_weights");   // static [3,3,1,C], or the [R, 9C] table of g_lookup
    names.push_back("conv1_bias");
  }
  if (dm.lookup) {
    names.push_back("fc_weights");   // [R, F*d] table
    names.push_back("fc_bias");      // [R, d] table
  } else if (dm.gen_fc) {
    for (const char* gname : {"fc_weights", "fc_bias"})
      for (int i = 0; i <= T->nh; ++i) {
        std::string pn = std::string(gname) + "/CPG/Projection" + std::to_string(i);
        names.push_back(pn);
        if (i < T->nh && dm.ctx_bn) { names.push_back(pn + "/BatchNorm/gamma"); names.push_back(pn + "/BatchNorm/beta"); }
      }
    for (int g = 0; g < 2; ++g) {
      T->chain[g].dims[0] = dm.r;
      for (int i = 0; i < T->nh; ++i) T->chain[g].dims[i + 1] = h->cfg.ctx_out[i];
    }
  } else {
    names.push_back("fc_weights"); names.push_back("fc_bias");
  }
  if ((int)names.size() > TR_MAX_PARAMS) return fail(h, COPER_EUNSUPPORTED, "coper_train_init: too many trainable tensors");
  for (auto& nm : names) {
    auto it = h->params.find(nm);
    if (it == h->params.end() || !it->second.set) return fail(h, COPER_EINVAL, "coper_train_init: missing parameter " + nm);
    TrainParam tp;
    tp.name = nm;
    tp.p = const_cast<float*>(it->second.ptr);
    tp.n = 1;
    for (int64_t s : it->second.shape) tp.n *= s;
    int rc;
    if ((rc = talloc(h, &tp.g, (size_t)tp.n)) || (rc = talloc(h, &tp.m, (size_t)tp.n)) || (rc = talloc(h, &tp.v, (size_t)tp.n)) ||
        (rc = talloc(h, &tp.vh, (size_t)tp.n)))
      return rc;
    COPER_HIP_TRY(h, hipMemset(tp.m, 0, sizeof(float) * tp.n));
    COPER_HIP_TRY(h, hipMemset(tp.v, 0, sizeof(float) * tp.n));
    COPER_HIP_TRY(h, hipMemset(tp.vh, 0, sizeof(float) * tp.n));
    COPER_HIP_TRY(h, hipMemset(tp.g, 0, sizeof(float) * tp.n));
    T->tp.push_back(tp);
  }
  int rc;
  int mx = dm.C > dm.d ? dm.C : dm.d;
  for (int i = 0; i < T->nh; ++i) mx = h->cfg.ctx_out[i] > mx ? h->cfg.ctx_out[i] : mx;
  for (int i = 0; i < T->nhc; ++i) mx = h->cfg.ctx_conv[i] > mx ? h->cfg.ctx_conv[i] : mx;
  if ((rc = talloc(h, &T->bnst, (size_t)4 * mx))) return rc;
  if ((rc = talloc(h, &T->red, (size_t)(2 + 2 * mx * TR_COLSUM_SLICES + TG_SUMSQ_SLOTS + 2 * TR_CS_SLOTS * 2 * mx)))) return rc;
  if ((rc = talloc(h, &T->tg_exps, (size_t)(10 + TR_EXP_CACHE))) || (rc = talloc(h, &T->tg_scratch, (size_t)4)) ||
      (rc = talloc(h, &T->wmax[0], (size_t)TG_MAX_SLOTS)) || (rc = talloc(h, &T->wmax[1], (size_t)TG_MAX_SLOTS)) ||
      (rc = talloc(h, &T->xmax, (size_t)TG_MAX_SLOTS)) || (rc = talloc(h, &T->dtmax, (size_t)TG_MAX_SLOTS)) ||
      (rc = talloc(h, &T->smax, (size_t)TG_MAX_SLOTS)))
    return rc;
  COPER_HIP_TRY(h, hipMemset(T->tg_exps, 0, (10 + TR_EXP_CACHE) * sizeof(int32_t)));
  COPER_HIP_TRY(h, hipMemset(T->wmax[0], 0, TG_MAX_SLOTS * sizeof(unsigned)));
  COPER_HIP_TRY(h, hipMemset(T->wmax[1], 0, TG_MAX_SLOTS * sizeof(unsigned)));
  COPER_HIP_TRY(h, hipMemset(T->tg_scratch, 0, 4 * sizeof(unsigned)));
  {
    static const bool one_stream = getenv("COPER_TRAIN_ONE_STREAM") != nullptr;   // A/B switch: the step as one chain
    if (!one_stream) {
      for (hipStream_t& st : T->side) COPER_HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      for (int i = 0; i < 3; ++i) {
        COPER_HIP_TRY(h, hipEventCreateWithFlags(&T->ev_fork[i], hipEventDisableTiming));
        COPER_HIP_TRY(h, hipEventCreateWithFlags(&T->ev_join[i], hipEventDisableTiming));
      }
    }
  }
  {
    TgPlanes* sets[10] = {&T->pX, &T->pXt, &T->pP1, &T->pP3, &T->pTn, &T->pTb, &T->mmX, &T->mmY, &T->mmX2, &T->mmY2};
    for (int i = 0; i < 10; ++i) sets[i]->exp = T->tg_exps + i;
  }
  return COPER_OK;
}

}  // extern "C"

// logits of the train-mode forward for coper_train_forward: the fma chain of k_tr_score_loss (sampled) / S + bias (1-vs-all)
__global__ __launch_bounds__(256) void k_tr_scores_out(const float* __restrict__ hv, const float* __restrict__ ent,
                                                       const float* __restrict__ pred_bias, const int32_t* __restrict__ lookup,
                                                       int64_t E, int d, int64_t L, float* __restrict__ out) {
  const int64_t b = blockIdx.x;
  for (int64_t l = threadIdx.x; l < L; l += 256) {
    int64_t row = lookup[b * L + l];
    if (row < 0 || row >= E) row = 0;
    const float* er = ent + row * d;
    float s = 0.f;
    for (int k = 0; k < d; ++k) s = fmaf(hv[b * d + k], er[k], s);
    out[b * L + l] = s + pred_bias[row];
  }
}
__global__ __launch_bounds__(256) void k_tr_add_bias_out(const float* __restrict__ S, const float* __restrict__ pred_bias, int64_t E,
                                                         int64_t total, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) out[i] = S[i] + pred_bias[i % E];
}

static int train_step_impl(coper_handle* h, const int64_t* e1, const int64_t* rel, const int32_t* lookup, const float* labels,
                           int64_t B, int64_t L, float* loss_out, void* stream, const int apply, float* pred_out, float* h_out) {
  if (!h) return COPER_EINVAL;
  TrainState* T = (TrainState*)h->train;
  if (!T) return fail(h, COPER_ESTATE, "coper_train_step: call coper_train_init first");
  const int nomov = apply ? 0 : 2;
  if (!e1 || !rel || !labels || B <= 0 || L <= 0 || B * L > 0x7fffffff)
    return fail(h, COPER_EINVAL, "coper_train_step: bad argument");
  const bool one_vs_all = lookup == nullptr;   // use_negative_sampling = False: labels are the dense e2_multi [B, |E|]
  if (one_vs_all && L != h->dm.E) return fail(h, COPER_EINVAL, "coper_train_step: lookup == NULL needs labels of shape [B, num_ent]");
  if (one_vs_all && (double)B * (double)h->dm.E * 4.0 > 512.0 * 1024 * 1024)
    return fail(h, COPER_EUNSUPPORTED, "coper_train_step: 1-vs-all training needs B*num_ent*4 <= 512 MiB in this version");
  const Dims& dm = h->dm;
  hipStream_t s = (hipStream_t)stream;
  COPER_HIP_TRY(h, hipSetDevice(h->cfg.device));
  // the side streams (TrainState::side): fork(i, k) lets side[k] start behind everything queued on s so far, join(i) lets s go on
  // behind what side[k] was given since.  Whatever leaves this function early joins what it forked (SideJoin).
  struct SideJoin {
    TrainState* T; hipStream_t s; int on[3] = {-1, -1, -1};
    hipError_t err = hipSuccess;      // the first failure of an event call (checked where the step ends)
    void note(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
    void fork(int i, int k) {
      note(hipEventRecord(T->ev_fork[i], s));
      note(hipStreamWaitEvent(T->side[k], T->ev_fork[i], 0));
      on[i] = k;
    }
    void join(int i) {
      if (on[i] < 0) return;
      note(hipEventRecord(T->ev_join[i], T->side[on[i]]));
      note(hipStreamWaitEvent(s, T->ev_join[i], 0));
      on[i] = -1;
    }
    ~SideJoin() { for (int i = 0; i < 3; ++i) join(i); }
  } sj{T, s};
  const bool two_streams = T->side[0] != nullptr;
  if (apply) h->prepared = false;   // the variables change: per-relation caches, fragment images and folded BN go stale
  const coper_train_config& tc = T->cfg;
  const int d = dm.d, r = dm.r, C = dm.C, P = dm.Ho * dm.Wo, isz = dm.in_h * dm.in_w, NT = dm.fh * dm.fw;   // NT: filter taps
  const int64_t F = dm.F, Fc = dm.F_conv;   // dense input width (F_conv + r under concat_rel), conv features
  const bool cat = dm.concat_rel;
  const bool lk = dm.lookup && dm.gen_fc;    // dense layer from g_lookup tables (otherwise static: models.py:217-228 with context_rel_out None)
  const bool gen = dm.gen_fc && !dm.lookup;
  const bool genc = dm.gen_conv && !dm.lookup;     // conv filters from projection generators
  const bool lkc = dm.gen_conv && dm.lookup;       // conv filters from g_lookup tables
  const int nh = T->nh, nhc = T->nhc;
  const int rc_cw = nhc ? T->chain[2].dims[nhc] : r;
  const int rc_cb = nhc ? T->chain[3].dims[nhc] : r;
  const int rc_w = nh ? T->chain[0].dims[nh] : r;   // width of the context that multiplies the last projection
  const int rc_b = nh ? T->chain[1].dims[nh] : r;
  if ((int64_t)B * F > 0xffffffffLL) return fail(h, COPER_EINVAL, "coper_train_step: batch too large for the dropout counter");
  int rc;
  if (B > T->capB || (!one_vs_all && L > T->capL)) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    int64_t cb = B > T->capB ? B : T->capB, cl = (!one_vs_all && L > T->capL) ? L : (T->capL > 0 ? T->capL : 1);
    if ((rc = talloc(h, &T->img, (size_t)cb * isz)) || (rc = talloc(h, &T->y, (size_t)cb * F)) ||
        (rc = talloc(h, &T->x, (size_t)cb * F)) ||
        // (dx doubles as the looked-up dense layer's scratch: TR_LK_NSL partial sums of [B, d] -- more than [B, F] where F < 4 d: a fuzz
        //  shape of round 6, d = 77 with three channels, wrote past the end of it)
        (rc = talloc(h, &T->dx, (size_t)cb * (size_t)(F > (int64_t)TR_LK_NSL * d ? F : (int64_t)TR_LK_NSL * d))) ||
        (rc = talloc(h, &T->c, (size_t)cb * r)) || (rc = talloc(h, &T->dc, (size_t)cb * r)) ||
        (rc = talloc(h, &T->z0, (size_t)cb * d)) || (rc = talloc(h, &T->z1, (size_t)cb * d)) ||
        (rc = talloc(h, &T->hv, (size_t)cb * d)) || (rc = talloc(h, &T->dh, (size_t)cb * d)) ||
        (rc = talloc(h, &T->dz, (size_t)cb * d)) || (rc = talloc(h, &T->ds, (size_t)cb * (one_vs_all ? 1 : cl))))
      return rc;
    if (cat && ((rc = talloc(h, &T->xc, (size_t)cb * F)) || (rc = talloc(h, &T->dxc, (size_t)cb * F)))) return rc;
    if (gen) {
      if ((rc = talloc(h, &T->A, (size_t)2 * rc_w * cb * d))) return rc;
      const int64_t nrk = (int64_t)rc_w * d;
      struct { TgPlanes* p; int64_t rows, K; } planes[] = {{&T->pX, cb, F}, {&T->pXt, F, cb}, {&T->pP1, nrk, F}, {&T->pP3, F, nrk},
                                                          {&T->pTn, nrk, cb}, {&T->pTb, cb, nrk}};
      for (auto& pl : planes)
        if ((rc = talloc(h, &pl.p->hi, tg_plane_elems(pl.rows, pl.K))) || (rc = talloc(h, &pl.p->lo, tg_plane_elems(pl.rows, pl.K)))) return rc;
    }
    for (int g = 0; g < 4; ++g) {
      if (g < 2 ? !gen : !genc) continue;
      const int nhx = g < 2 ? nh : nhc;
      TrainState::Chain& ch = T->chain[g];
      for (int i = 0; i <= nhx; ++i) {
        if (i > 0 && (rc = talloc(h, &ch.v[i], (size_t)cb * ch.dims[i]))) return rc;
        if ((rc = talloc(h, &ch.dv[i], (size_t)cb * ch.dims[i]))) return rc;
        if (i < nhx && ((rc = talloc(h, &ch.u[i], (size_t)cb * ch.dims[i + 1])) || (rc = talloc(h, &ch.a[i], (size_t)cb * ch.dims[i + 1])) ||
                        (rc = talloc(h, &ch.du[i], (size_t)cb * ch.dims[i + 1])) || (rc = talloc(h, &ch.st[i], (size_t)2 * ch.dims[i + 1]))))
          return rc;
      }
    }
    if (dm.gen_conv && ((rc = talloc(h, &T->Kt, (size_t)cb * NT * C)) || (rc = talloc(h, &T->Kbv, (size_t)cb * C)))) return rc;
    // per-query filter / bias gradients: what the generators and tables reduce (gen_conv), and -- round 6 -- what the STATIC filters'
    // gradients are summed from (512 workgroups adding to the same 320 addresses were 50 of k_tr_conv_bwd's 60 us)
    if ((rc = talloc(h, &T->dKs, (size_t)cb * NT * C)) || (rc = talloc(h, &T->dkbs, (size_t)cb * C))) return rc;
    T->capB = cb; T->capL = cl;
  }
  auto P_ = [&](const char* n) -> float* { return T->find(n)->p; };
  auto G_ = [&](const char* n) -> float* { return T->find(n)->g; };
  float* ent = P_("ent_emb");
  float* relp = dm.lookup ? nullptr : P_("rel_emb");
  const int use_batch = tc.batch_norm_train_stats ? 1 : 0;
  int mx = C > d ? C : d;
  for (int i = 0; i < nh; ++i) mx = h->cfg.ctx_out[i] > mx ? h->cfg.ctx_out[i] : mx;
  for (int i = 0; i < nhc; ++i) mx = h->cfg.ctx_conv[i] > mx ? h->cfg.ctx_conv[i] : mx;
  float *mean1 = T->bnst, *inv1 = T->bnst + mx, *mean2 = T->bnst + 2 * mx, *inv2 = T->bnst + 3 * mx;
  double* red = T->red;         // [0] loss, [1] sumsq, then TR_COLSUM_SLICES slices of 2 mx column sums, one per use
  auto colsum_slice = [&](int i) { return red + 2 + (size_t)i * 2 * mx; };
  double* ssq = red + 2 + (size_t)2 * mx * TR_COLSUM_SLICES;   // TG_SUMSQ_SLOTS partial sums of the squared gradient norm   // 0 Conv1BN, 1 FCBN, 2 Conv1BN backward, 3 + 4 g + i chains
  // ... then two slices in TR_CS_SLOTS copies each (Conv1BN's statistics and its backward sums: thousands of workgroups add to them)
  auto cs_slots = [&](int j) { return ssq + TG_SUMSQ_SLOTS + (size_t)j * TR_CS_SLOTS * 2 * mx; };
  const int cs_n = TR_CS_SLOTS;
  double* colsum;
  const uint32_t thr_h = dropout_threshold24(tc.hidden_dropout), thr_o = dropout_threshold24(tc.output_dropout);
  const float ks_h = 1.f / (1.f - tc.hidden_dropout), ks_o = 1.f / (1.f - tc.output_dropout);
  const uint32_t step = T->step;

  // ---- zero what is accumulated by atomics: one launch
  const bool dense_scorer_bwd = (double)B * (double)dm.E * 4.0 <= 512.0 * 1024 * 1024 && dm.E <= 0x7fffffff;
  // (Round 6 tried the sampled scorer's forward and dh in ONE pass over the gathered rows -- a wave per row, the score a butterfly
  //  sum over its lanes: 242 us against 61 + 45 for the two kernels.  Thirty-two sequential iterations per wave, each with two
  //  dependent round trips and eight six-step cross-lane sums, are latency end to end; removed.)
  if (!one_vs_all && dense_scorer_bwd && B * dm.E > T->capS) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    if ((rc = talloc(h, &T->Sd, (size_t)(B * dm.E)))) return rc;
    T->capS = B * dm.E;
  }
  const std::string wlast = "fc_weights/CPG/Projection" + std::to_string(nh), blast = "fc_bias/CPG/Projection" + std::to_string(nh);
  {
    ZeroList zl;
    zl.n = 0;
    auto add = [&](void* p, size_t bytes) { if (p && bytes && zl.n < TR_ZERO_MAX) { zl.p[zl.n] = p; zl.bytes[zl.n] = bytes; ++zl.n; } };
    add(red, sizeof(double) * (2 + (size_t)2 * mx * TR_COLSUM_SLICES + TG_SUMSQ_SLOTS + (size_t)2 * TR_CS_SLOTS * 2 * mx));
    for (const char* nm : {"ent_emb", "rel_emb", "conv1_weights", "conv1_bias", "pred_bias"})
      if (T->find(nm)) add(G_(nm), sizeof(float) * T->find(nm)->n);
    if (lk) add(G_("fc_bias"), sizeof(float) * T->find("fc_bias")->n);
    else if (gen) add(G_(blast.c_str()), sizeof(float) * rc_b * d);
    else add(G_("fc_bias"), sizeof(float) * d);
    // (the dense S of the sampled scorer's backward is written whole by k_tr_build_S: nothing to zero)
    if (apply) add(T->wmax[T->wmax_cur ^ 1], sizeof(unsigned) * TG_MAX_SLOTS);      // what this step's optimizer pass fills for the next step
    add(T->xmax, sizeof(unsigned) * TG_MAX_SLOTS);
    add(T->dtmax, sizeof(unsigned) * TG_MAX_SLOTS);
    add(T->smax, sizeof(unsigned) * TG_MAX_SLOTS);
    hipLaunchKernelGGL(k_tr_zero_list, dim3(256, (unsigned)zl.n), dim3(256), 0, s, zl);
  }
  T->exp_cache.clear();
  // the dense weights' maximum as the last step's optimizer pass left it (TrainState::wmax), when it is known to describe them
  const unsigned* w_slots = T->wmax_valid ? T->wmax[T->wmax_cur] : nullptr;
  // (a coper_group_next registration, or a grouping prepared ahead, was for an evaluation pass: a training step drops both)
  h->pipe.invalidate_grouping();
  if (lk) {
    // group the batch by relation (perm / rel_offset / rel_count of the inference path): the table gradient is
    // written per present relation, never zero-filled (1.75 GB at FB15k-237 shapes)
    if ((rc = coper_reserve(h, B, 0, stream))) return rc;
    group_use_set(h, 0);
    if ((rc = launch_group_by_relation(h, e1, rel, false, B, 32, s))) return rc;
  }

  // ---- forward
  const uint32_t thr_c = dropout_threshold24(tc.context_rel_dropout);
  const float ks_c = 1.f / (1.f - tc.context_rel_dropout);
  // g_MLP generator chain g: context rows c -> v[nhx] (models.py:56-68); g_linear: the context is c itself
  auto chain_forward = [&](int g, int nhx) -> int {
    TrainState::Chain& ch = T->chain[g];
    ch.v[0] = T->c;
    for (int i = 0; i < nhx; ++i) {
      const int ni = ch.dims[i], nj = ch.dims[i + 1];
      const std::string pn = std::string(kGenNames[g]) + "/CPG/Projection" + std::to_string(i);
      const int64_t tot = B * nj;
      hipLaunchKernelGGL(k_tr_small_mm, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ch.v[i], P_(pn.c_str()), B, ni, nj, ch.u[i]);
      const float *ga = nullptr, *be = nullptr;
      if (dm.ctx_bn) {
        double* cs = colsum_slice(3 + 4 * g + i);
        if (use_batch) hipLaunchKernelGGL(k_tr_col_sums, dim3(64), dim3(256), 0, s, ch.u[i], B, nj, cs, 1);
        hipLaunchKernelGGL(k_tr_bn_finish, dim3((nj + 63) / 64), dim3(64), 0, s, cs, nj, (double)B, use_batch, tc.batch_norm_momentum, 0 | nomov,
                           const_cast<float*>(h->params[pn + "/BatchNorm/moving_mean"].ptr),
                           const_cast<float*>(h->params[pn + "/BatchNorm/moving_variance"].ptr), ch.st[i], ch.st[i] + nj);
        ga = P_((pn + "/BatchNorm/gamma").c_str());
        be = P_((pn + "/BatchNorm/beta").c_str());
      }
      hipLaunchKernelGGL(k_tr_chain_act, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ch.u[i], ch.st[i], ga, be, nj, tot, tc.seed, step,
                         (uint32_t)(16 + 8 * g + i), thr_c, ks_c, ch.a[i], ch.v[i + 1]);
    }
    return COPER_OK;
  };
  // per-sample conv filters (models.py:231-250,374-380): generated from the relation rows, or looked up
  const std::string cwlast = "conv1_weights/CPG/Projection" + std::to_string(nhc), cblast = "conv1_bias/CPG/Projection" + std::to_string(nhc);
  const float *ccw = nullptr, *ccb = nullptr;   // contexts of the conv generators [B, rc_cw], [B, rc_cb]
  if (genc) {
    hipLaunchKernelGGL(k_tr_gather_rows, dim3((unsigned)((B * r + 255) / 256)), dim3(256), 0, s, relp, rel, dm.R, r, B * r, T->c);
    if ((rc = chain_forward(2, nhc)) || (rc = chain_forward(3, nhc))) return rc;
    ccw = nhc ? T->chain[2].v[nhc] : T->c;
    ccb = nhc ? T->chain[3].v[nhc] : T->c;
    hipLaunchKernelGGL(k_tr_small_mm, dim3((unsigned)((B * NT * C + 255) / 256)), dim3(256), 0, s, ccw, P_(cwlast.c_str()), B, rc_cw, NT * C, T->Kt);
    hipLaunchKernelGGL(k_tr_small_mm, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, ccb, P_(cblast.c_str()), B, rc_cb, C, T->Kbv);
  } else if (lkc) {
    hipLaunchKernelGGL(k_tr_gather_rows, dim3((unsigned)((B * NT * C + 255) / 256)), dim3(256), 0, s, P_("conv1_weights"), rel, dm.R, NT * C,
                       B * NT * C, T->Kt);
    hipLaunchKernelGGL(k_tr_gather_rows, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, P_("conv1_bias"), rel, dm.R, C, B * C, T->Kbv);
  }
  const float* K_ps = dm.gen_conv ? T->Kt : nullptr;
  const float* kb_ps = dm.gen_conv ? T->Kbv : nullptr;
  size_t lds_conv = sizeof(float) * (size_t)(isz + (NT + 1) * C);
  hipLaunchKernelGGL(k_tr_conv_fwd, dim3((unsigned)B), dim3(256), lds_conv, s, e1, rel, ent, relp, dm.gen_conv ? nullptr : P_("conv1_weights"),
                     dm.gen_conv ? nullptr : P_("conv1_bias"), dm.E, dm.R, d, r, dm.in_h, dm.in_w, dm.stacked ? 1 : 0, C, dm.Ho, dm.Wo, T->img,
                     ((gen || cat) && !genc) ? T->c : nullptr, T->y, K_ps, kb_ps, dm.fh, dm.fw);
  const int64_t nBF = B * Fc;
  if (use_batch) {
    hipLaunchKernelGGL(k_tr_col_sums, dim3(1024), dim3(256), 0, s, T->y, B * (int64_t)P, C, cs_slots(0), cs_n);
    hipLaunchKernelGGL(k_tr_fold_slots, dim3(1), dim3(256), 0, s, cs_slots(0), 2 * C, cs_n);
  }
  hipLaunchKernelGGL(k_tr_bn_finish, dim3((C + 63) / 64), dim3(64), 0, s, cs_slots(0), C, (double)B * P, use_batch,
                     tc.batch_norm_momentum, 1 | nomov, const_cast<float*>(h->params["Conv1BN/moving_mean"].ptr),
                     const_cast<float*>(h->params["Conv1BN/moving_variance"].ptr), mean1, inv1);
  hipLaunchKernelGGL(k_tr_bn1_fwd, dim3((unsigned)((nBF + 255) / 256)), dim3(256), 0, s, T->y, mean1, inv1, P_("Conv1BN/gamma"),
                     P_("Conv1BN/beta"), C, nBF, tc.seed, step, thr_h, ks_h, T->x, T->xmax);
  const unsigned* x_slots = cat ? nullptr : T->xmax;      // (concat_rel: the dense layer's input is another tensor)
  if (cat) hipLaunchKernelGGL(k_tr_concat, dim3((unsigned)((B * F + 255) / 256)), dim3(256), 0, s, T->x, T->c, Fc, r, B * F, T->xc);
  const float* xin = cat ? T->xc : T->x;   // [B, F]
  float* dxin = cat ? T->dxc : T->dx;
  if (nh > 0 && ((rc = chain_forward(0, nh)) || (rc = chain_forward(1, nh)))) return rc;
  const float* cw = nh ? T->chain[0].v[nh] : T->c;   // [B, rc_w]
  const float* cbv = nh ? T->chain[1].v[nh] : T->c;  // [B, rc_b]
  const float* Wmat = gen ? P_(wlast.c_str()) : P_("fc_weights");   // row-major [rc_w * F, d] (generated) or [F, d] (static)
  bool p3_packed = false;        // the projection's second view (rows f) was packed with its first
  const int64_t nBd = B * d;
  float* Tf = T->A;                 // T[rho][b][k]
  float* dTf = T->A + (size_t)rc_w * nBd;
  constexpr int LK_NSL = TR_LK_NSL;      // F slices of the looked-up dense layer (deterministic partial sums)
  int t_slices = 1;              // K slices of the generated dense layer's product left for k_tr_fc_post_slices
  int dx_slices = 1;             // K slices of the dx product left for k_tr_bn1_bwd_sums
  if (lk) {
    // z0[b] = x[b] W[rel[b]]: one pass over the looked-up rows (B * F * d * 4 bytes)
    hipLaunchKernelGGL(k_tr_lookup_fwd, dim3((unsigned)B, LK_NSL), dim3(256), sizeof(float) * (size_t)((F + LK_NSL - 1) / LK_NSL + 1), s, T->x,
                       P_("fc_weights"), rel, dm.R, F, d, LK_NSL, B, T->dx /* scratch: [NSL][B][d], allocated for it */);
    hipLaunchKernelGGL(k_tr_lookup_post, dim3((unsigned)((nBd + 255) / 256)), dim3(256), 0, s, T->dx, LK_NSL, P_("fc_bias"), rel, dm.R, d, nBd,
                       tc.seed, step, thr_o, ks_o, T->z1);
  } else {
    if (gen) {
      // T[rho][b][k] = sum_f x[b][f] P[rho][f][k] on the bf16 matrix cores with split operands (train_gemm_bf16.hip): x and P
      // are packed into fragment planes (P as rows (rho, k) with f contracted), one GEMM of [B] x [r*d] outputs
      const int64_t nrk = (int64_t)rc_w * d;
      if ((rc = tg_pack(h, xin, tg_idx(F), tg_idx(1), B, F, tg_rows_pad(B), false, T->pX, s, T->tg_scratch, nullptr, x_slots))) return rc;
      // the projection is an operand of two products, contracted over f here and over (rho, k) in dx: a training step packs BOTH
      // views from one read (tg_pack_both: 118 MB read once instead of twice, one launch instead of two)
      // (round 6 also ran this pack on a side stream beside the conv / Conv1BN launches in front of it: the stream takes the memory
      //  system, k_tr_bn1_fwd beside it 38 us for 9 -- 5 us gained, not kept)
      if (apply && (d & 3) == 0 && (((uintptr_t)Wmat) & 15) == 0) {
        if ((rc = tg_pack_both(h, Wmat, tg_idx(d), tg_idx2(d, F * (int64_t)d, 1), F, nrk, T->pP3, T->pP1, s, T->tg_scratch, w_slots))) return rc;
        p3_packed = true;
      } else if ((rc = tg_pack(h, Wmat, tg_idx2(d, F * (int64_t)d, 1), tg_idx(d), nrk, F, tg_rows_pad(nrk), true, T->pP1, s, T->tg_scratch, nullptr, w_slots)))
        return rc;
      if ((rc = tg_gemm_split(h, T, s, T->pX, B, T->pP1, nrk, F, Tf, tg_idx(d), tg_idx2(d, nBd, 1), &t_slices))) return rc;
    } else {
      // z0[B,d] = x[B,F] W[F,d]: 8 output tiles, K = F cut into slices
      if ((rc = tg_matmul(h, T, s, MmView{xin, tg_idx(F), tg_idx(1), false}, B, MmView{Wmat, tg_idx(1), tg_idx(d), true}, d, F, T->z0,
                          tg_idx(d), tg_idx(1), nullptr, x_slots, w_slots)))
        return rc;
    }
    if (t_slices > 1) {
#define COPER_FC_SLICES(NS)                                                                                                              \
  case NS:                                                                                                                               \
    hipLaunchKernelGGL(k_tr_fc_post_slices<NS>, dim3((unsigned)((nBd + 255) / 256)), dim3(256), 0, s, T->mmP, cw, rc_w, cbv,            \
                       P_(blast.c_str()), rc_b, d, nBd, tc.seed, step, thr_o, ks_o, Tf, T->z1);                                          \
    break;
      switch (t_slices) {
        COPER_FC_SLICES(2) COPER_FC_SLICES(3) COPER_FC_SLICES(4) COPER_FC_SLICES(5) COPER_FC_SLICES(6) COPER_FC_SLICES(7) COPER_FC_SLICES(8)
      }
#undef COPER_FC_SLICES
    } else
      hipLaunchKernelGGL(k_tr_fc_post, dim3((unsigned)((nBd + 255) / 256)), dim3(256), 0, s, gen ? Tf : T->z0, gen ? nullptr : P_("fc_bias"),
                         cw, rc_w, cbv, gen ? P_(blast.c_str()) : nullptr, rc_b, d, nBd, tc.seed, step, thr_o, ks_o, T->z1);
  }
  if (use_batch) hipLaunchKernelGGL(k_tr_col_sums, dim3(64), dim3(256), 0, s, T->z1, B, d, colsum_slice(1), 1);
  hipLaunchKernelGGL(k_tr_bn_finish, dim3((d + 63) / 64), dim3(64), 0, s, colsum_slice(1), d, (double)B, use_batch, tc.batch_norm_momentum, 0 | nomov,
                     const_cast<float*>(h->params["FCBN/moving_mean"].ptr), const_cast<float*>(h->params["FCBN/moving_variance"].ptr),
                     mean2, inv2);
  hipLaunchKernelGGL(k_tr_fcbn_fwd, dim3((unsigned)((nBd + 255) / 256)), dim3(256), 0, s, T->z1, mean2, inv2, P_("FCBN/gamma"),
                     P_("FCBN/beta"), d, nBd, T->hv);
  const float inv_BL = (float)(1.0 / ((double)B * (double)L));
  // a training step over a sampled lookup whose dE goes through the dense S matrix: scores, loss, ds and dh from one pass over the rows
  // (coper_train_forward takes the same kernel: its loss is the step's, bit for bit; the dh it leaves in the workspace is not used)
  const bool score_dh_fused = !one_vs_all && dense_scorer_bwd && (d & 3) == 0 && d >= 16 && d <= 1024 && L >= 1 && L <= SF_MAX_L &&
                              (((uintptr_t)ent | (uintptr_t)T->dh | (uintptr_t)T->hv) & 15) == 0;
  if (one_vs_all) {
    if (B * dm.E > T->capS) {
      COPER_HIP_TRY(h, hipStreamSynchronize(s));
      if ((rc = talloc(h, &T->Sd, (size_t)(B * dm.E)))) return rc;
      T->capS = B * dm.E;
    }
    // S[B,E] = h E^T
    if ((rc = tg_matmul(h, T, s, MmView{T->hv, tg_idx(d), tg_idx(1), false}, B, MmView{ent, tg_idx(d), tg_idx(1), false}, dm.E, d, T->Sd,
                        tg_idx(dm.E), tg_idx(1))))
      return rc;
    if (pred_out)
      hipLaunchKernelGGL(k_tr_add_bias_out, dim3((unsigned)((B * dm.E + 255) / 256)), dim3(256), 0, s, T->Sd, P_("pred_bias"), dm.E, B * dm.E, pred_out);
    for (size_t i = 0; i < T->exp_cache.size(); ++i)      // (the loss kernel rewrites S in place: its power of two as an OUTPUT operand is gone)
      if (T->exp_cache[i].first == T->Sd) T->exp_cache[i].first = nullptr;
    hipLaunchKernelGGL(k_tr_dense_loss, dim3(2048), dim3(256), 0, s, T->Sd, P_("pred_bias"), labels, dm.E, B * dm.E,
                       tc.label_smoothing_epsilon, (float)(1.0 / (double)dm.E), inv_BL, red);
  } else if (score_dh_fused) {
    const int d4 = d >> 2, slots = 256 / d4 > 256 / SF_U ? 256 / SF_U : 256 / d4, RB = SF_U * slots;
    hipLaunchKernelGGL(k_tr_score_loss_dh, dim3((unsigned)B), dim3(256),
                       sizeof(float4) * (size_t)slots * d4 + sizeof(float) * (size_t)RB * (d4 + 2) + sizeof(int) * (size_t)L, s, T->hv, ent,
                       P_("pred_bias"), lookup, labels, dm.E, d, (int)L, tc.label_smoothing_epsilon, (float)(1.0 / (double)dm.E), inv_BL, T->ds,
                       T->dh, red);
  } else {
    hipLaunchKernelGGL(k_tr_score_loss, dim3((unsigned)B), dim3(256), sizeof(float) * d, s, T->hv, ent, P_("pred_bias"), lookup, labels,
                       dm.E, d, L, tc.label_smoothing_epsilon, (float)(1.0 / (double)dm.E), inv_BL, T->ds, red);
  }
  if (loss_out) hipLaunchKernelGGL(k_tr_store_loss, dim3(1), dim3(1), 0, s, red, 1.0 / ((double)B * (double)L), loss_out);
  if (!one_vs_all && pred_out)
    hipLaunchKernelGGL(k_tr_scores_out, dim3((unsigned)B), dim3(256), 0, s, T->hv, ent, P_("pred_bias"), lookup, dm.E, d, L, pred_out);
  if (h_out) COPER_HIP_TRY(h, hipMemcpyAsync(h_out, T->hv, sizeof(float) * (size_t)nBd, hipMemcpyDeviceToDevice, s));
  if (!apply) {      // coper_train_forward: nothing is differentiated, nothing updated, the step counter (dropout masks) stays
    COPER_HIP_TRY(h, hipGetLastError());
    return COPER_OK;
  }

  // ---- backward
  std::string sumsq_done;   // the leaf whose squared gradient norm its own GEMM accumulates
  if (one_vs_all) {
    hipLaunchKernelGGL(k_tr_col_sum_f32, dim3((unsigned)((dm.E + 255) / 256)), dim3(256), 0, s, T->Sd, B, dm.E, G_("pred_bias"));
    // dE[E,d] = S^T h
    if ((rc = tg_matmul(h, T, s, MmView{T->Sd, tg_idx(1), tg_idx(dm.E), true}, dm.E, MmView{T->hv, tg_idx(1), tg_idx(d), true}, d, B,
                        G_("ent_emb"), tg_idx(d), tg_idx(1))))
      return rc;
    // dh[B,d] = S E: 8 output tiles, K = |E| cut into slices
    if ((rc = tg_matmul(h, T, s, MmView{T->Sd, tg_idx(dm.E), tg_idx(1), false}, B, MmView{ent, tg_idx(1), tg_idx(d), true}, d, dm.E,
                        T->dh, tg_idx(d), tg_idx(1))))
      return rc;
  } else if (dense_scorer_bwd) {
    // the scorer's backward (S, dbias, dE = S^T h: ~70 us of launches that need ds and h only) beside the dense layer's: on the side
    // stream (tg_matmul packs into a second pair of plane sets there) when dE needs no K slices (the partial-sum pool is the dx
    // product's); joined in front of the conv backward, which adds the e1 rows to dE
    const bool sb_side = two_streams && score_dh_fused && tg_split_k(dm.E, d, B) == 1;
    hipStream_t const s_main = s;
    if (sb_side) { sj.fork(1, 0); s = T->side[0]; }
    {
      const size_t lds_s = sizeof(float) * (size_t)(dm.E < TR_S_CHUNK ? dm.E : TR_S_CHUNK);
      // (the kernel also holds 16 bytes of static LDS: asking for the whole 160 KB as dynamic is refused, and so is the launch after it)
      if (lds_s > 64 * 1024)
        COPER_HIP_TRY(h, hipFuncSetAttribute((const void*)k_tr_build_S, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      hipLaunchKernelGGL(k_tr_build_S, dim3((unsigned)B), dim3(256), lds_s, s, lookup, T->ds, dm.E, L, T->Sd, T->smax);
      hipLaunchKernelGGL(k_tr_col_sums_add, dim3((unsigned)((dm.E + 255) / 256), (unsigned)((B + 63) / 64)), dim3(256), 0, s, T->Sd, B, dm.E,
                         G_("pred_bias"));
    }
    // dE[E,d] = S^T h  (overwrites the zeroed gradient; the e1-row contributions are added after it)
    if ((rc = tg_matmul(h, T, s, MmView{T->Sd, tg_idx(1), tg_idx(dm.E), true}, dm.E, MmView{T->hv, tg_idx(1), tg_idx(d), true}, d, B,
                        G_("ent_emb"), tg_idx(d), tg_idx(1), nullptr, T->smax)))
      return rc;
    s = s_main;
    // dh by the gather (a [d,B] = [d,|E|] x [|E|,B] GEMM has 8 output tiles and a long K: slower than the gather)
    if (score_dh_fused) {
      // (dh came with the scores: k_tr_score_loss_dh)
    } else if ((d & 3) == 0 && d >= 16 && d <= 1024 && (((uintptr_t)ent | (uintptr_t)T->dh) & 15) == 0)
      hipLaunchKernelGGL(k_tr_dh_gather4, dim3((unsigned)B), dim3(256), sizeof(float4) * (size_t)(256 / (d >> 2)) * (d >> 2), s, ent, lookup,
                         T->ds, dm.E, d, L, T->dh);
    else
      hipLaunchKernelGGL(k_tr_score_bwd<false>, dim3((unsigned)B), dim3(256), 0, s, T->hv, ent, lookup, T->ds, dm.E, d, L, T->dh, nullptr,
                         nullptr);
  } else {
    hipLaunchKernelGGL(k_tr_score_bwd<true>, dim3((unsigned)B), dim3(256), 0, s, T->hv, ent, lookup, T->ds, dm.E, d, L, T->dh, G_("ent_emb"),
                       G_("pred_bias"));
  }
  hipLaunchKernelGGL(k_tr_fcbn_bwd, dim3((unsigned)d), dim3(256), 0, s, T->z1, T->hv, T->dh, mean2, inv2, P_("FCBN/gamma"), B, d, use_batch,
                     G_("FCBN/gamma"), G_("FCBN/beta"), T->dz);
  if (lk) {
    hipLaunchKernelGGL(k_tr_lookup_post_bwd, dim3((unsigned)((nBd + 255) / 256)), dim3(256), 0, s, T->dz, rel, dm.R, d, nBd, tc.seed, step, thr_o,
                       ks_o, G_("fc_bias"));
    const int rows_per_wg = 32;
    hipLaunchKernelGGL(k_tr_lookup_dW, dim3((unsigned)((F + rows_per_wg - 1) / rows_per_wg), (unsigned)dm.R), dim3(256), 0, s, T->x, T->dz, h->perm,
                       h->rel_offset, h->rel_count, F, d, rows_per_wg, G_("fc_weights"));
    hipLaunchKernelGGL(k_tr_lookup_dx, dim3((unsigned)((F + 63) / 64), (unsigned)B), dim3(256), sizeof(float) * d, s, T->dz, P_("fc_weights"), rel,
                       dm.R, F, d, T->dx);
  } else {
  // gradients of the two contexts: dcw [B, rc_w] (k_tr_dc_from_partials) and dcb [B, rc_b] (k_tr_fc_post_bwd)
  float* dcw = gen ? T->chain[0].dv[nh] : nullptr;
  float* dcb = gen ? T->chain[1].dv[nh] : nullptr;
  hipLaunchKernelGGL(k_tr_fc_post_bwd, dim3((unsigned)B), dim3(256), sizeof(float) * d, s, T->dz, cbv,
                     gen ? P_(blast.c_str()) : nullptr, rc_b, d, tc.seed, step, thr_o, ks_o,
                     gen ? nullptr : G_("fc_bias"), gen ? G_(blast.c_str()) : nullptr, dcb);
  if (gen)      // (into the zeroed gradient: 8 atomics per address)
    hipLaunchKernelGGL(k_tr_wsum_rows_add, dim3((unsigned)rc_b, (unsigned)((B + 63) / 64)), dim3(256), 0, s, cbv, T->dz, B, rc_b, d, G_(blast.c_str()));
  float* dW = gen ? G_(wlast.c_str()) : G_("fc_weights");
  if (gen) {
    // dT[rho][b,:] = cw[b,rho] dz[b,:];  dP[rho][f][k] = sum_b x[b][f] dT[rho][b][k]  and
    // dx[b][f] = sum_(rho,k) dT[rho][b][k] P[rho][f][k]: two split-bf16 GEMMs, the [B, r*F] intermediate dz P2^T is never formed
    const int64_t nrk = (int64_t)rc_w * d;
    hipLaunchKernelGGL(k_tr_scale_rows, dim3((unsigned)((nBd + 255) / 256)), dim3(256), 0, s, T->dz, cw, rc_w, d, nBd, dTf, T->dtmax);
    // (x and the projection were packed for the forward pass: the same tensors, the same powers of two -- no second reduction)
    if ((rc = tg_pack(h, xin, tg_idx(1), tg_idx(F), F, B, tg_rows_pad(F), true, T->pXt, s, T->tg_scratch, T->pX.exp))) return rc;
    if ((rc = tg_pack(h, dTf, tg_idx2(d, nBd, 1), tg_idx(d), nrk, B, tg_rows_pad(nrk), true, T->pTn, s, T->tg_scratch, nullptr, T->dtmax))) return rc;
    if ((rc = tg_pack(h, dTf, tg_idx(d), tg_idx2(d, nBd, 1), B, nrk, tg_rows_pad(B), false, T->pTb, s, T->tg_scratch, T->pTn.exp))) return rc;
    if (!p3_packed && (rc = tg_pack(h, Wmat, tg_idx(d), tg_idx2(d, F * (int64_t)d, 1), F, nrk, tg_rows_pad(F), false, T->pP3, s, T->tg_scratch, T->pP1.exp))) return rc;
    // (the K slices of dx stay in the partial-sum pool for k_tr_bn1_bwd_sums when dx IS the conv features' gradient: no concat_rel)
    if ((rc = tg_gemm_split(h, T, s, T->pTb, B, T->pP3, F, nrk, dxin, tg_idx(F), tg_idx(1), cat ? nullptr : &dx_slices))) return rc;
    // the dP product's result is read by the optimizer only: BEHIND the dx product, on the second side stream, beside the dozen short
    // launches between here and the optimizer (slice sum, Conv1BN and conv backward, the generators' chains).  (Beside the dx product
    // itself the two took 211 us for 100 + 84: a SIMD holds one wave of either.)
    if (two_streams) sj.fork(2, 1);
    if ((rc = tg_gemm_nt(h, T->pXt, F, T->pTn, nrk, B, dW, tg_idx(d), tg_idx2(d, F * (int64_t)d, 1), two_streams ? T->side[1] : s, 1, nullptr, ssq))) return rc;
    sumsq_done = wlast;   // the GEMM added |dP|^2 to the global-norm accumulator as it stored
    hipLaunchKernelGGL(k_tr_dc_from_partials, dim3((unsigned)((B * rc_w + 3) / 4)), dim3(256), 0, s, T->dz, Tf, B, rc_w, d, dcw);
  } else {
    // static dense layer (plain ConvE): dW[F,d] = x^T dz and dx[B,F] = dz W^T
    if ((rc = tg_matmul(h, T, s, MmView{xin, tg_idx(1), tg_idx(F), true}, F, MmView{T->dz, tg_idx(1), tg_idx(d), true}, d, B, dW, tg_idx(d),
                        tg_idx(1), ssq)))
      return rc;
    sumsq_done = "fc_weights";
    if ((rc = tg_matmul(h, T, s, MmView{T->dz, tg_idx(d), tg_idx(1), false}, B, MmView{Wmat, tg_idx(d), tg_idx(1), false}, F, d, dxin,
                        tg_idx(F), tg_idx(1))))
      return rc;
  }
  if (cat) hipLaunchKernelGGL(k_tr_split, dim3((unsigned)((B * F + 255) / 256)), dim3(256), 0, s, T->dxc, rel, dm.R, Fc, r, B * F, T->dx, G_("rel_emb"));
  }
  // ---- back through a generator chain to the relation rows: dv[nhx] -> dv[0]
  auto chain_backward = [&](int g, int nhx) {
    TrainState::Chain& ch = T->chain[g];
    for (int i = nhx - 1; i >= 0; --i) {
      const int ni = ch.dims[i], nj = ch.dims[i + 1];
      const std::string pn = std::string(kGenNames[g]) + "/CPG/Projection" + std::to_string(i);
      const int64_t tot = B * nj;
      hipLaunchKernelGGL(k_tr_chain_drop_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ch.dv[i + 1], tot, tc.seed, step,
                         (uint32_t)(16 + 8 * g + i), thr_c, ks_c, ch.du[i]);
      // ReLU (+ BN) backward, in place on du; BN gamma / beta gradients
      if (dm.ctx_bn)
        hipLaunchKernelGGL(k_tr_fcbn_bwd, dim3((unsigned)nj), dim3(256), 0, s, ch.u[i], ch.a[i], ch.du[i], ch.st[i], ch.st[i] + nj,
                           P_((pn + "/BatchNorm/gamma").c_str()), B, nj, use_batch, G_((pn + "/BatchNorm/gamma").c_str()),
                           G_((pn + "/BatchNorm/beta").c_str()), ch.du[i]);
      else
        hipLaunchKernelGGL(k_tr_fcbn_bwd, dim3((unsigned)nj), dim3(256), 0, s, ch.u[i], ch.a[i], ch.du[i], (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, B, nj, 0, (float*)nullptr, (float*)nullptr, ch.du[i]);
      hipLaunchKernelGGL(k_tr_small_mm_tn, dim3((unsigned)(((int64_t)ni * nj + 255) / 256)), dim3(256), 0, s, ch.v[i], ch.du[i], B, ni, nj,
                         G_(pn.c_str()));
      hipLaunchKernelGGL(k_tr_small_mm_nt, dim3((unsigned)((B * ni + 255) / 256)), dim3(256), 0, s, ch.du[i], P_(pn.c_str()), B, ni, nj, 0,
                         ch.dv[i]);
    }
  };
  if (gen) { chain_backward(0, nh); chain_backward(1, nh); }
  colsum = cs_slots(1);
  {
    const dim3 g1((unsigned)((nBF + 255) / 256 < 2048 ? (nBF + 255) / 256 : 2048));
#define COPER_BN1_SUMS(NS)                                                                                                                \
  case NS:                                                                                                                                \
    hipLaunchKernelGGL(k_tr_bn1_bwd_sums<NS>, g1, dim3(256), 0, s, T->dx, NS ? T->mmP : nullptr, T->y, mean1, inv1, P_("Conv1BN/gamma"), \
                       P_("Conv1BN/beta"), C, nBF, tc.seed, step, thr_h, ks_h, colsum, cs_n);                                             \
    break;
    switch (dx_slices > 1 ? dx_slices : 0) {
      COPER_BN1_SUMS(0) COPER_BN1_SUMS(2) COPER_BN1_SUMS(3) COPER_BN1_SUMS(4) COPER_BN1_SUMS(5) COPER_BN1_SUMS(6) COPER_BN1_SUMS(7) COPER_BN1_SUMS(8)
    }
#undef COPER_BN1_SUMS
  }
  hipLaunchKernelGGL(k_tr_fold_slots, dim3(1), dim3(256), 0, s, colsum, 2 * C, cs_n);
  hipLaunchKernelGGL(k_tr_bn1_bwd_apply, dim3((unsigned)((nBF + 255) / 256)), dim3(256), 0, s, T->dx, T->y, mean1, inv1, P_("Conv1BN/gamma"),
                     colsum, C, nBF, (double)B * P, use_batch, G_("Conv1BN/gamma"), G_("Conv1BN/beta"));
  size_t lds_cb = sizeof(float) * (size_t)(isz + (size_t)P * (C + 1) + NT * C);
  if (lds_cb > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_tr_conv_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  sj.join(1);      // (dE = S^T h is stored: the conv backward adds the e1 rows to it)
  hipLaunchKernelGGL(k_tr_conv_bwd, dim3((unsigned)B), dim3(256), lds_cb, s, T->dx, T->img, dm.gen_conv ? nullptr : P_("conv1_weights"), e1, rel,
                     dm.E, dm.R, d, r, dm.in_h, dm.in_w, dm.stacked ? 1 : 0, C, dm.Ho, dm.Wo, G_("ent_emb"),
                     dm.lookup ? nullptr : G_("rel_emb"), K_ps, T->dKs, T->dkbs, dm.fh, dm.fw);
  if (!dm.gen_conv) {      // static filters: their gradients are the column sums of the per-query ones (added to the zeroed gradients)
    hipLaunchKernelGGL(k_tr_col_sums_add, dim3((unsigned)((NT * C + 255) / 256), (unsigned)((B + 63) / 64)), dim3(256), 0, s, T->dKs, B,
                       (int64_t)NT * C, G_("conv1_weights"));
    hipLaunchKernelGGL(k_tr_col_sums_add, dim3(1, (unsigned)((B + 63) / 64)), dim3(256), 0, s, T->dkbs, B, (int64_t)C, G_("conv1_bias"));
  }
  if (genc) {
    // per-sample filter gradients -> last projections and contexts, then back through the conv generators
    hipLaunchKernelGGL(k_tr_small_mm_tn, dim3((unsigned)(((int64_t)rc_cw * NT * C + 255) / 256)), dim3(256), 0, s, ccw, T->dKs, B, rc_cw, NT * C,
                       G_(cwlast.c_str()));
    hipLaunchKernelGGL(k_tr_small_mm_nt, dim3((unsigned)((B * rc_cw + 255) / 256)), dim3(256), 0, s, T->dKs, P_(cwlast.c_str()), B, rc_cw, NT * C,
                       0, T->chain[2].dv[nhc]);
    hipLaunchKernelGGL(k_tr_small_mm_tn, dim3((unsigned)(((int64_t)rc_cb * C + 255) / 256)), dim3(256), 0, s, ccb, T->dkbs, B, rc_cb, C,
                       G_(cblast.c_str()));
    hipLaunchKernelGGL(k_tr_small_mm_nt, dim3((unsigned)((B * rc_cb + 255) / 256)), dim3(256), 0, s, T->dkbs, P_(cblast.c_str()), B, rc_cb, C, 0,
                       T->chain[3].dv[nhc]);
    chain_backward(2, nhc);
    chain_backward(3, nhc);
  } else if (lkc) {
    // table rows: d(conv1_weights)[rel[b]] += dK[b] (the table gradients were zeroed above)
    hipLaunchKernelGGL(k_tr_scatter_rows, dim3((unsigned)((B * NT * C + 255) / 256)), dim3(256), 0, s, T->dKs, rel, dm.R, NT * C, B * NT * C,
                       G_("conv1_weights"));
    hipLaunchKernelGGL(k_tr_scatter_rows, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, T->dkbs, rel, dm.R, C, B * C, G_("conv1_bias"));
  }
  for (int g = 0; g < 4; ++g)
    if (g < 2 ? gen : genc)
      hipLaunchKernelGGL(k_tr_scatter_rows, dim3((unsigned)((B * r + 255) / 256)), dim3(256), 0, s, T->chain[g].dv[0], rel, dm.R, r, B * r,
                         G_("rel_emb"));

  // ---- clip + AMSGrad
  TrainTensors tt;
  int np = (int)T->tp.size();
  for (int i = 0; i < np; ++i) {
    tt.p[i] = T->tp[i].p; tt.g[i] = T->tp[i].g; tt.m[i] = T->tp[i].m; tt.v[i] = T->tp[i].v; tt.vh[i] = T->tp[i].vh;
    tt.n[i] = T->tp[i].n;
    const bool table = lk && T->tp[i].name == "fc_weights";
    tt.rowlen[i] = table ? F * d : 1;
    tt.rowcnt[i] = table ? h->rel_count : nullptr;
  }
  // the dense weights' largest |p_new| for the next step's packs (k_tr_amsgrad's 16-byte path only: it is the one that carries it)
  tt.wmax = nullptr; tt.wmax_of = -1;
  {
    const std::string wname = lk ? std::string() : (gen ? wlast : std::string("fc_weights"));
    for (int i = 0; i < np && !wname.empty(); ++i)
      if (T->tp[i].name == wname && !tt.rowcnt[i] &&
          ((((uintptr_t)tt.p[i]) | ((uintptr_t)tt.g[i]) | ((uintptr_t)tt.m[i]) | ((uintptr_t)tt.v[i]) | ((uintptr_t)tt.vh[i])) & 15) == 0) {
        tt.wmax = T->wmax[T->wmax_cur ^ 1];
        tt.wmax_of = i;
      }
  }
  int skip = -1;
  for (int i = 0; i < np; ++i)
    if (!sumsq_done.empty() && T->tp[i].name == sumsq_done) skip = i;
  sj.join(2);      // (dP and its squared norm)
  sj.join(1);
  if (sj.err != hipSuccess) return fail(h, COPER_EHIP, "coper_train_step: an event call of the side streams failed");
  hipLaunchKernelGGL(k_tr_sumsq, dim3(512, (unsigned)np), dim3(256), 0, s, tt, skip, ssq);
  const float lr_t = (float)((double)tc.learning_rate * std::sqrt(1.0 - T->b2p) / (1.0 - T->b1p));
  hipLaunchKernelGGL(k_tr_amsgrad, dim3(2048, (unsigned)np), dim3(256), 0, s, tt, ssq, red + 1, tc.clip_norm, lr_t, tc.beta1, tc.beta2,
                     tc.epsilon);
  COPER_HIP_TRY(h, hipGetLastError());
  T->b1p *= tc.beta1;
  T->b2p *= tc.beta2;
  T->step += 1;
  T->wmax_cur ^= 1;
  T->wmax_valid = tt.wmax != nullptr;
  return COPER_OK;
}

extern "C" {

COPER_API int coper_train_step(coper_handle* h, const int64_t* e1, const int64_t* rel, const int32_t* lookup, const float* labels,
                               int64_t B, int64_t L, float* loss_out, void* stream) {
  return train_step_impl(h, e1, rel, lookup, labels, B, L, loss_out, stream, 1, nullptr, nullptr);
}

namespace coper {
namespace {
// coper_train_grad on a looked-up table: rows of relations the last batch did not hold are never written by the step (the optimizer and
// the global norm skip them by their count) -- the copy handed out shows them as the zeros they are
__global__ __launch_bounds__(256) void k_tr_zero_absent_rows(float* __restrict__ out, const int32_t* __restrict__ rowcnt, int64_t rowlen, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (rowcnt[i / rowlen] == 0) out[i] = 0.f;
}
}  // namespace
}  // namespace coper

COPER_API int coper_train_forward(coper_handle* h, const int64_t* e1, const int64_t* rel, const int32_t* lookup, const float* labels,
                                  int64_t B, int64_t L, float* loss_out, float* pred_out, float* h_out, void* stream) {
  return train_step_impl(h, e1, rel, lookup, labels, B, L, loss_out, stream, 0, pred_out, h_out);
}

COPER_API int coper_train_grad(coper_handle* h, const char* leaf_name, float* out, int64_t cap, int64_t* n, double* global_norm,
                               void* stream) {
  if (!h || !leaf_name) return COPER_EINVAL;
  TrainState* T = (TrainState*)h->train;
  if (!T) return fail(h, COPER_ESTATE, "coper_train_grad: call coper_train_init first");
  TrainParam* t = T->find(leaf_name);
  if (!t) return fail(h, COPER_EINVAL, std::string("coper_train_grad: not a trainable leaf: ") + leaf_name);
  if (n) *n = t->n;
  if (out) {
    if (cap < t->n) return fail(h, COPER_EINVAL, "coper_train_grad: output buffer too small");
    COPER_HIP_TRY(h, hipMemcpyAsync(out, t->g, sizeof(float) * t->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (h->dm.lookup && h->dm.gen_fc && t->name == "fc_weights" && h->rel_count && T->step > 0) {
      const int64_t rowlen = h->dm.F * (int64_t)h->dm.d;
      hipLaunchKernelGGL(k_tr_zero_absent_rows, dim3((unsigned)((t->n + 255) / 256 < 4096 ? (t->n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, out,
                         h->rel_count, rowlen, t->n);
      COPER_HIP_TRY(h, hipGetLastError());
    }
  }
  if (global_norm) {
    COPER_HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    double ss = 0;
    COPER_HIP_TRY(h, hipMemcpy(&ss, T->red + 1, sizeof(double), hipMemcpyDeviceToHost));
    *global_norm = std::sqrt(ss);
  }
  return COPER_OK;
}

COPER_API int coper_train_slot(coper_handle* h, const char* leaf_name, int32_t which, float* buf, int64_t cap, int32_t set,
                               int64_t* n, void* stream) {
  if (!h || !leaf_name) return COPER_EINVAL;
  TrainState* T = (TrainState*)h->train;
  if (!T) return fail(h, COPER_ESTATE, "coper_train_slot: call coper_train_init first");
  TrainParam* t = T->find(leaf_name);
  if (!t) return fail(h, COPER_EINVAL, std::string("coper_train_slot: not a trainable leaf: ") + leaf_name);
  if (which < 0 || which > 2) return fail(h, COPER_EINVAL, "coper_train_slot: which = 0 (m), 1 (v), 2 (v_hat)");
  float* slot = which == 0 ? t->m : which == 1 ? t->v : t->vh;
  if (n) *n = t->n;
  if (!buf) return COPER_OK;
  if (cap < t->n) return fail(h, COPER_EINVAL, "coper_train_slot: buffer too small");
  COPER_HIP_TRY(h, hipMemcpyAsync(set ? slot : buf, set ? buf : slot, sizeof(float) * t->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return COPER_OK;
}

COPER_API int coper_train_powers(coper_handle* h, const double* set_beta1_power, const double* set_beta2_power,
                                 const int64_t* set_step, double* beta1_power, double* beta2_power, int64_t* step) {
  if (!h) return COPER_EINVAL;
  TrainState* T = (TrainState*)h->train;
  if (!T) return fail(h, COPER_ESTATE, "coper_train_powers: call coper_train_init first");
  if (set_beta1_power) T->b1p = *set_beta1_power;
  if (set_beta2_power) T->b2p = *set_beta2_power;
  if (set_step) T->step = (uint32_t)*set_step;
  if (beta1_power) *beta1_power = T->b1p;
  if (beta2_power) *beta2_power = T->b2p;
  if (step) *step = (int64_t)T->step;
  return COPER_OK;
}

}  // extern "C"
