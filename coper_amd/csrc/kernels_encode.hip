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
#include "group_body.h"

namespace coper {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// grouping
// ------------------------------------------------------------------------------------------------
// LDS-privatised when the relation table fits (R <= 8192): one global atomic per (block, relation present)
// instead of one per query on a few hundred hot counters.
constexpr int HIST_LDS_MAX = 8192;
constexpr int HIST_BLOCK = 1024;

__global__ __launch_bounds__(HIST_BLOCK) void k_rel_hist(const int64_t* __restrict__ rel, int64_t B, int use_rel,
                                                          int64_t R, int32_t* __restrict__ count,
                                                          int32_t* __restrict__ bad) {
  extern __shared__ int32_t sh[];
  const bool priv = R <= HIST_LDS_MAX;
  if (priv) {
    for (int k = threadIdx.x; k < R; k += HIST_BLOCK) sh[k] = 0;
    __syncthreads();
  }
  int64_t b = (int64_t)blockIdx.x * HIST_BLOCK + threadIdx.x;
  if (b < B) {
    int64_t key = use_rel ? rel[b] : 0;
    if (key < 0 || key >= R) { atomicAdd(bad, 1); key = 0; }  // clamped, and reported by coper_check_ids
    if (priv) atomicAdd(&sh[key], 1); else atomicAdd(&count[key], 1);
  }
  if (priv) {
    __syncthreads();
    for (int k = threadIdx.x; k < R; k += HIST_BLOCK)
      if (sh[k]) atomicAdd(&count[k], sh[k]);
  }
}

__global__ __launch_bounds__(1024) void k_rel_scan_tiles(const int32_t* __restrict__ count, int64_t R,
                                                         int64_t cap_small, int32_t* __restrict__ offset,
                                                         int32_t* __restrict__ tiles,
                                                         int32_t* __restrict__ n_tiles) {
  rel_scan_tiles_body(count, R, cap_small, offset, tiles, n_tiles);
}

// Histogram with the scan folded in: the last block to finish (ticket counter) takes the totals out of the accumulation
// buffer with memory-side atomic exchanges (coherent whatever XCD added them; the exchange leaves the buffer ZERO for the
// next call), publishes them in `count` (what later kernels and coper_check_ids read), runs the scan / tile lists, and
// zeroes the scatter cursors -- two launches per grouping (this one and k_rel_scatter).  Every pointer is fixed for the
// life of the workspace and the launch resets its own state: a hipGraph that captured it replays correctly any number
// of times, also interleaved with eager calls (round 2 alternated two count buffers from HOST state, which a captured
// graph froze: ADVICE r2).  `acc` must be zero on entry: ensure_workspace zeroes it once, every call leaves it zero.
__global__ __launch_bounds__(HIST_BLOCK) void k_rel_hist_scan(const int64_t* __restrict__ rel, int64_t B, int use_rel, int64_t R,
                                                               int64_t R_all, int32_t* __restrict__ acc, int32_t* __restrict__ count,
                                                               int32_t* __restrict__ done, int64_t cap_small,
                                                               int32_t* __restrict__ offset, int32_t* __restrict__ tiles,
                                                               int32_t* __restrict__ n_tiles, int32_t* __restrict__ cursor,
                                                               int n_hist_blocks, const int32_t* __restrict__ post_src, int64_t post_n,
                                                               int32_t* __restrict__ post_dst) {
  extern __shared__ int32_t sh[];   // [R]
  __shared__ int s_last;
  // Posting role (coper_post_i32_next): the blocks behind the histogram's copy the LAST pass's int32 results (its ranks) to
  // pinned host memory while this pass's histogram runs -- the copy that used to be a launch of its own behind every pass.
  if ((int)blockIdx.x >= n_hist_blocks) {
    const int64_t nb = (int64_t)gridDim.x - n_hist_blocks, b = (int64_t)blockIdx.x - n_hist_blocks;
    const bool vec = ((((uintptr_t)post_src) | ((uintptr_t)post_dst)) & 15) == 0;
    const int64_t n4 = vec ? post_n / 4 : 0;
    for (int64_t i = b * HIST_BLOCK + threadIdx.x; i < n4; i += nb * HIST_BLOCK) ((int4*)post_dst)[i] = ((const int4*)post_src)[i];
    for (int64_t i = 4 * n4 + b * HIST_BLOCK + threadIdx.x; i < post_n; i += nb * HIST_BLOCK) post_dst[i] = post_src[i];
    return;
  }
  int32_t* bad = acc + R_all + 1;
  for (int k = threadIdx.x; k < R; k += HIST_BLOCK) sh[k] = 0;
  __syncthreads();
  int64_t b = (int64_t)blockIdx.x * HIST_BLOCK + threadIdx.x;
  if (b < B) {
    int64_t key = use_rel ? rel[b] : 0;
    if (key < 0 || key >= R) { atomicAdd(bad, 1); key = 0; }  // clamped, and reported by coper_check_ids
    atomicAdd(&sh[key], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < R; k += HIST_BLOCK)
    if (sh[k]) atomicAdd(&acc[k], sh[k]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(done, 1) == n_hist_blocks - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  for (int k = threadIdx.x; k < R; k += HIST_BLOCK) {
    const int32_t c = atomicExch(&acc[k], 0);
    sh[k] = c;
    count[k] = c;
    cursor[k] = 0;
  }
  if (threadIdx.x == 0) {
    count[R_all + 1] = atomicExch(bad, 0);
    *done = 0;
  }
  __syncthreads();
  rel_scan_tiles_body(sh, R, cap_small, offset, tiles, n_tiles);
}

// Small batches (<= 4096 queries, <= 8192 relation keys): histogram, scan / tile lists and scatter in ONE
// single-workgroup launch (the four-launch path costs ~20 us of launch latency; measured: 110 -> 102 us per hipGraph
// batch of 512; at B = 20,480 one workgroup is latency-bound and loses, so large batches keep the four launches).
__global__ __launch_bounds__(1024) void k_rel_group_single(const int64_t* __restrict__ rel, const int64_t* __restrict__ e1,
                                                           int64_t B, int use_rel, int64_t R, int64_t R_all, int have_e1_rows,
                                                           int64_t shard_lo, int64_t n_local, int64_t cap_small,
                                                           int32_t* __restrict__ count, int32_t* __restrict__ bad,
                                                           int32_t* offset, int32_t* __restrict__ tiles,
                                                           int32_t* __restrict__ n_tiles, int32_t* __restrict__ perm,
                                                           int32_t* __restrict__ sorted_row, int32_t* __restrict__ sorted_rid,
                                                           int32_t* __restrict__ inv_perm, float* __restrict__ x3m) {
  extern __shared__ int32_t sh[];   // cnt[R] | cursor[R]
  if (x3m) x3m[threadIdx.x] = 0.f;  // (1024 threads, 1024 slots: see k_rel_scatter)
  int32_t* cnt = sh;
  int32_t* cur = sh + R;
  for (int k = threadIdx.x; k < 2 * R; k += 1024) sh[k] = 0;
  if (threadIdx.x == 0) *bad = 0;
  __syncthreads();
  int nbad = 0;
  for (int64_t b = threadIdx.x; b < B; b += 1024) {
    int64_t key = use_rel ? rel[b] : 0;
    if (key < 0 || key >= R) { ++nbad; key = 0; }
    atomicAdd(&cnt[key], 1);
  }
  if (nbad) atomicAdd(bad, nbad);
  __syncthreads();
  for (int k = threadIdx.x; k < R; k += 1024) count[k] = cnt[k];   // the global copy other kernels read
  rel_scan_tiles_body(cnt, R, cap_small, offset, tiles, n_tiles);
  __syncthreads();
  for (int64_t b = threadIdx.x; b < B; b += 1024) {
    int64_t rid = rel[b];
    if (rid < 0 || rid >= R_all) rid = 0;
    const int64_t key = use_rel ? rid : 0;
    int64_t row;
    if (have_e1_rows) {
      row = b;
    } else {
      row = e1[b] - shard_lo;
      if (row < 0 || row >= n_local) row = -1;
    }
    const int pos = offset[key] + atomicAdd(&cur[key], 1);
    perm[pos] = (int32_t)b;
    inv_perm[b] = pos;
    sorted_row[pos] = (int32_t)row;
    sorted_rid[pos] = (int32_t)rid;
  }
}

// Besides perm, the scatter leaves what the fused conv + dense kernel's image prologue needs in sorted order
// (one coalesced load instead of the dependent chain perm -> e1 / rel): the local entity row of e1 (-1 when it
// is not on this shard; the query index itself when the caller passes e1_rows) and the validated relation id.
__global__ __launch_bounds__(HIST_BLOCK) void k_rel_scatter(const int64_t* __restrict__ rel, int64_t B, int use_rel,
                                                             int64_t R, const int32_t* __restrict__ offset,
                                                             int32_t* __restrict__ cursor,
                                                             int32_t* __restrict__ perm,
                                                             const int64_t* __restrict__ e1, int have_e1_rows,
                                                             int64_t shard_lo, int64_t n_local, int64_t R_all,
                                                             int32_t* __restrict__ sorted_row,
                                                             int32_t* __restrict__ sorted_rid,
                                                             int32_t* __restrict__ inv_perm, float* __restrict__ x3m) {
  extern __shared__ int32_t sh[];  // [R] block counts, then block bases
  const bool priv = R <= HIST_LDS_MAX;
  // x3 mode: the slots the fused encoder folds its largest h values into start every pass at zero (bf16x3_chain.h)
  if (x3m && blockIdx.x == 0)
    for (int i = threadIdx.x; i < 1024; i += HIST_BLOCK) x3m[i] = 0.f;
  int64_t b = (int64_t)blockIdx.x * HIST_BLOCK + threadIdx.x;
  int64_t key = 0, rid = 0, row = -1;
  if (b < B) {
    rid = rel[b];
    if (rid < 0 || rid >= R_all) rid = 0;
    key = use_rel ? rid : 0;
    if (have_e1_rows) {
      row = b;
    } else {
      row = e1[b] - shard_lo;
      if (row < 0 || row >= n_local) row = -1;
    }
  }
  if (!priv) {
    if (b < B) {
      int pos = offset[key] + atomicAdd(&cursor[key], 1);
      perm[pos] = (int32_t)b;
      inv_perm[b] = pos;
      sorted_row[pos] = (int32_t)row;
      sorted_rid[pos] = (int32_t)rid;
    }
    return;
  }
  for (int k = threadIdx.x; k < R; k += HIST_BLOCK) sh[k] = 0;
  __syncthreads();
  int local = 0;
  if (b < B) local = atomicAdd(&sh[key], 1);           // rank inside the block
  __syncthreads();
  for (int k = threadIdx.x; k < R; k += HIST_BLOCK) {   // reserve the block's range per relation
    int c = sh[k];
    sh[k] = c ? atomicAdd(&cursor[k], c) : 0;
  }
  __syncthreads();
  if (b < B) {
    int pos = offset[key] + sh[key] + local;
    perm[pos] = (int32_t)b;
    inv_perm[b] = pos;
    sorted_row[pos] = (int32_t)row;
    sorted_rid[pos] = (int32_t)rid;
  }
}

// ONE relation key (static dense weights: plain ConvE): the counting sort is the identity -- perm[b] = b, one group of B queries cut
// into ceil(B / 128) balanced tiles -- so nothing is counted or scanned (round 5).  The two-launch path spent 22 us of a plain-ConvE
// pass having 20,480 threads add to ONE LDS counter and ONE global counter (k_rel_hist_scan: 22.1 us against 12.2 with 474 keys).
// Same outputs as k_rel_hist_scan + k_rel_scatter (tile order: the rem larger tiles first); the blocks behind the main ones
// carry a pending coper_post_i32_next job like k_rel_hist_scan's.
__global__ __launch_bounds__(HIST_BLOCK) void k_rel_group_identity(const int64_t* __restrict__ rel, const int64_t* __restrict__ e1, int64_t B,
                                                                    int64_t R_all, int have_e1_rows, int64_t shard_lo, int64_t n_local,
                                                                    int64_t cap_small, int32_t* __restrict__ count, int32_t* __restrict__ offset,
                                                                    int32_t* __restrict__ tiles, int32_t* __restrict__ n_tiles,
                                                                    int32_t* __restrict__ perm, int32_t* __restrict__ sorted_row,
                                                                    int32_t* __restrict__ sorted_rid, int32_t* __restrict__ inv_perm,
                                                                    float* __restrict__ x3m, int n_main, const int32_t* __restrict__ post_src,
                                                                    int64_t post_n, int32_t* __restrict__ post_dst) {
  if ((int)blockIdx.x >= n_main) {
    const int64_t nb = (int64_t)gridDim.x - n_main, b = (int64_t)blockIdx.x - n_main;
    const bool vec = ((((uintptr_t)post_src) | ((uintptr_t)post_dst)) & 15) == 0;
    const int64_t n4 = vec ? post_n / 4 : 0;
    for (int64_t i = b * HIST_BLOCK + threadIdx.x; i < n4; i += nb * HIST_BLOCK) ((int4*)post_dst)[i] = ((const int4*)post_src)[i];
    for (int64_t i = 4 * n4 + b * HIST_BLOCK + threadIdx.x; i < post_n; i += nb * HIST_BLOCK) post_dst[i] = post_src[i];
    return;
  }
  const int64_t b = (int64_t)blockIdx.x * HIST_BLOCK + threadIdx.x;
  if (b < B) {
    int64_t rid = rel[b];
    if (rid < 0 || rid >= R_all) rid = 0;
    int64_t row;
    if (have_e1_rows) row = b;
    else { row = e1[b] - shard_lo; if (row < 0 || row >= n_local) row = -1; }
    perm[b] = (int32_t)b;
    inv_perm[b] = (int32_t)b;
    sorted_row[b] = (int32_t)row;
    sorted_rid[b] = (int32_t)rid;
  }
  if (blockIdx.x == 0) {
    if (x3m) for (int i = threadIdx.x; i < 1024; i += HIST_BLOCK) x3m[i] = 0.f;
    const int c = (int)B;
    if (threadIdx.x == 0) {
      count[0] = c;
      count[R_all + 1] = 0;
      offset[0] = 0; offset[1] = c;
      n_tiles[0] = (c > 0 && c <= 32) ? 1 : 0;
      n_tiles[1] = c > 32 ? (c + 127) / 128 : 0;
      if (c > 0 && c <= 32) { tiles[0] = 0; tiles[1] = 0; tiles[2] = c; tiles[3] = 0; }
    }
    if (c > 32) {
      const int nb = (c + 127) / 128, bsz = c / nb, rem = c % nb;
      int32_t* tiles_big = tiles + 4 * cap_small;
      for (int j = threadIdx.x; j < nb; j += HIST_BLOCK) {
        int32_t* t = tiles_big + 4 * (int64_t)j;
        t[0] = 0;
        t[1] = j * bsz + (j < rem ? j : rem);
        t[2] = bsz + (j < rem ? 1 : 0);
        t[3] = nb > 1 ? 1 : 0;
      }
    }
  }
}

// capacity of the small-tile list (one tile per relation key at most); the big list follows it
static int64_t small_tile_cap(const coper_handle* h) { return (h->dm.gen_fc ? h->dm.R : 1) + 1; }

int launch_group_by_relation(coper_handle* h, const int64_t* e1, const int64_t* rel, bool have_e1_rows, int64_t B, int tq,
                             hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t R = dm.gen_fc ? dm.R : 1;
  bool post_now = h->pipe.post.n > 0;
  if (post_now) {      // (a pass being captured into a hipGraph leaves the job to the next eager call: a replay must not repeat it)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); post_now = false; }
    else if (cap != hipStreamCaptureStatusNone) post_now = false;
  }
  static const bool no_identity = getenv("COPER_GROUP_NO_IDENTITY") != nullptr;     // A/B switch, read once
  if (R == 1 && !no_identity) {
    const unsigned nbm = (unsigned)((B + HIST_BLOCK - 1) / HIST_BLOCK);
    unsigned npost = 0;
    const int32_t* psrc = h->pipe.post.src; const int64_t pn = h->pipe.post.n; int32_t* pdst = h->pipe.post.dst;
    if (pn > 0 && post_now) {
      npost = (unsigned)((pn + 4 * HIST_BLOCK - 1) / (4 * HIST_BLOCK));
      if (npost > 32) npost = 32;
      h->pipe.take_post();
    }
    hipLaunchKernelGGL(k_rel_group_identity, dim3(nbm + npost), dim3(HIST_BLOCK), 0, s, rel, e1, B, (int64_t)dm.R, have_e1_rows ? 1 : 0,
                       (int64_t)h->cfg.shard_lo, dm.n_local, small_tile_cap(h), h->rel_count, h->rel_offset, h->tiles, h->n_tiles, h->perm,
                       h->sorted_row, h->sorted_rid, h->inv_perm, h->x3m, (int)nbm, psrc, pn, pdst);
    (void)tq;
    COPER_HIP_TRY(h, hipGetLastError());
    return COPER_OK;
  }
  if (post_now && !(R <= HIST_LDS_MAX && B > 4096)) {     // (a pending coper_post_i32_next rides in the two-launch path only)
    const int64_t pn = h->pipe.take_post();
    int rc0 = launch_copy_i32(h, h->pipe.post.src, pn, h->pipe.post.dst, s);
    if (rc0) return rc0;
  }
  // rel_count (= rel_count_buf[0]) holds the counts of the last call; rel_count_buf[1] is the accumulation buffer of the
  // two-launch path, zero between calls (its last block takes the totals out with atomic exchanges).  No host-side state
  // changes per call: every path below is hipGraph-capturable and replayable, alone or mixed with eager calls.
  if (R <= HIST_LDS_MAX && B <= 4096) {   // a single workgroup is latency-bound beyond a few ids per thread
    // rel_count[R+1] doubles as the out-of-range counter (reset by the kernel)
    hipLaunchKernelGGL(k_rel_group_single, dim3(1), dim3(1024), sizeof(int32_t) * 2 * (size_t)R, s, rel, e1, B, dm.gen_fc ? 1 : 0, R,
                       dm.R, have_e1_rows ? 1 : 0, (int64_t)h->cfg.shard_lo, dm.n_local, small_tile_cap(h), h->rel_count,
                       h->rel_count + dm.R + 1, h->rel_offset, h->tiles, h->n_tiles, h->perm, h->sorted_row, h->sorted_rid, h->inv_perm,
                       h->x3m);
    (void)tq;
    COPER_HIP_TRY(h, hipGetLastError());
    return COPER_OK;
  }
  unsigned nb = (unsigned)((B + HIST_BLOCK - 1) / HIST_BLOCK);
  (void)tq;
  if (R <= HIST_LDS_MAX) {
    // two launches: histogram + scan (last block), scatter.  A pending coper_post_i32_next rides in the first one.
    unsigned npost = 0;
    const int32_t* psrc = h->pipe.post.src; const int64_t pn = h->pipe.post.n; int32_t* pdst = h->pipe.post.dst;
    if (pn > 0 && post_now) {
      npost = (unsigned)((pn + 4 * HIST_BLOCK - 1) / (4 * HIST_BLOCK));
      if (npost > 32) npost = 32;
      h->pipe.take_post();
    }
    hipLaunchKernelGGL(k_rel_hist_scan, dim3(nb + npost), dim3(HIST_BLOCK), sizeof(int32_t) * (size_t)R, s, rel, B, dm.gen_fc ? 1 : 0, R,
                       (int64_t)dm.R, h->rel_count_buf[1], h->rel_count, h->group_done, small_tile_cap(h), h->rel_offset, h->tiles,
                       h->n_tiles, h->rel_cursor, (int)nb, psrc, pn, pdst);
    hipLaunchKernelGGL(k_rel_scatter, dim3(nb), dim3(HIST_BLOCK), sizeof(int32_t) * (size_t)R, s, rel, B, dm.gen_fc ? 1 : 0, R,
                       h->rel_offset, h->rel_cursor, h->perm, e1, have_e1_rows ? 1 : 0, (int64_t)h->cfg.shard_lo, dm.n_local, dm.R,
                       h->sorted_row, h->sorted_rid, h->inv_perm, h->x3m);
    COPER_HIP_TRY(h, hipGetLastError());
    return COPER_OK;
  }
  // relation tables beyond the LDS histogram: memset + three launches
  COPER_HIP_TRY(h, hipMemsetAsync(h->rel_count, 0, sizeof(int32_t) * (dm.R + 2), s));
  COPER_HIP_TRY(h, hipMemsetAsync(h->rel_cursor, 0, sizeof(int32_t) * (dm.R + 2), s));
  size_t hl = 0;
  // rel_count[R+1] doubles as the out-of-range counter (ids are validated on device, never trusted)
  hipLaunchKernelGGL(k_rel_hist, dim3(nb), dim3(HIST_BLOCK), hl, s, rel, B, dm.gen_fc ? 1 : 0, R, h->rel_count,
                     h->rel_count + dm.R + 1);
  hipLaunchKernelGGL(k_rel_scan_tiles, dim3(1), dim3(1024), 0, s, h->rel_count, R, small_tile_cap(h), h->rel_offset,
                     h->tiles, h->n_tiles);
  hipLaunchKernelGGL(k_rel_scatter, dim3(nb), dim3(HIST_BLOCK), hl, s, rel, B, dm.gen_fc ? 1 : 0, R, h->rel_offset,
                     h->rel_cursor, h->perm, e1, have_e1_rows ? 1 : 0, (int64_t)h->cfg.shard_lo, dm.n_local, dm.R,
                     h->sorted_row, h->sorted_rid, h->inv_perm, h->x3m);
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

// 3x3 filters, C a multiple of 32 (every shipped config): lane = (channel c = lane&31, row half = lane>>5),
// the 4 waves of a workgroup take 8 output rows per sweep; each lane slides a 3x3 register window along its
// row (3 LDS reads -- broadcast across the 32 channels -- and 9 FMAs per output), taps/bias/BN in registers;
// a wave-store is two 128-B runs of x.  QPB queries per workgroup amortise the tap loads.
template <int QPB>
__global__ __launch_bounds__(256) void k_conv3x3_bn_relu(
    const int64_t* __restrict__ e1, const int64_t* __restrict__ rel, const float* __restrict__ e1_rows,
    const int32_t* __restrict__ perm, const float* __restrict__ ent, int64_t shard_lo, int64_t n_local,
    const float* __restrict__ rel_emb, const float* __restrict__ conv_w, const float* __restrict__ conv_b,
    int per_rel_conv, const float* __restrict__ scale, const float* __restrict__ shift, int d, int r, int in_h,
    int in_w, int stacked, int C, int Ho, int Wo, int concat_rel, int64_t F, int64_t F_pad, int64_t R, int64_t B,
    float* __restrict__ x_sorted) {
  extern __shared__ float lds[];  // img[QPB][in_h*in_w]
  const int img_sz = in_h * in_w;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cl = lane & 31, hf = lane >> 5;
  const int64_t pos0 = (int64_t)blockIdx.x * QPB;
  int64_t rids[QPB];
#pragma unroll
  for (int qq = 0; qq < QPB; ++qq) {
    int64_t pos = pos0 + qq;
    rids[qq] = 0;
    if (pos >= B) continue;
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
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int c = c0 + cl;
    float tap[9], kb = 0.f;
    const float sc = scale[c], sh = shift[c];
    int64_t tap_rid = -1;
#pragma unroll
    for (int qq = 0; qq < QPB; ++qq) {
      int64_t pos = pos0 + qq;
      if (pos >= B) break;
      const int64_t rid = per_rel_conv ? rids[qq] : 0;
      if (rid != tap_rid) {  // workgroup-uniform
        const float* wsrc = per_rel_conv ? conv_w + rid * (int64_t)(9 * C) : conv_w;
        const float* bsrc = per_rel_conv ? conv_b + rid * (int64_t)C : conv_b;
#pragma unroll
        for (int k = 0; k < 9; ++k) tap[k] = wsrc[k * C + c];
        kb = bsrc[c];
        tap_rid = rid;
      }
      const float* img = lds + qq * img_sz;
      float* xo = x_sorted + pos * F_pad;
      for (int i0 = 0; i0 < Ho; i0 += 8) {
        const int i = i0 + 2 * wave + hf;
        if (i < Ho) {
          const float* r0 = img + i * in_w;
          const float* r1 = r0 + in_w;
          const float* r2 = r1 + in_w;
          float w00 = r0[0], w01 = r0[1], w10 = r1[0], w11 = r1[1], w20 = r2[0], w21 = r2[1];
          float* dst = xo + (int64_t)i * Wo * C + c;
          for (int j = 0; j < Wo; ++j) {
            float w02 = r0[j + 2], w12 = r1[j + 2], w22 = r2[j + 2];
            float y = 0.f;
            y = fmaf(w00, tap[0], y); y = fmaf(w01, tap[1], y); y = fmaf(w02, tap[2], y);
            y = fmaf(w10, tap[3], y); y = fmaf(w11, tap[4], y); y = fmaf(w12, tap[5], y);
            y = fmaf(w20, tap[6], y); y = fmaf(w21, tap[7], y); y = fmaf(w22, tap[8], y);
            y += kb;
            y = fmaf(y, sc, sh);
            dst[(int64_t)j * C] = fmaxf(y, 0.f);
            w00 = w01; w01 = w02; w10 = w11; w11 = w12; w20 = w21; w21 = w22;
          }
        }
      }
    }
  }
  // tail of each row: concat_rel columns and the zero padding up to F_pad
#pragma unroll
  for (int qq = 0; qq < QPB; ++qq) {
    int64_t pos = pos0 + qq;
    if (pos >= B) break;
    float* xo = x_sorted + pos * F_pad;
    int64_t Fc = (int64_t)Ho * Wo * C;
    if (concat_rel)
      for (int k = threadIdx.x; k < r; k += 256) xo[Fc + k] = rel_emb[rids[qq] * r + k];
    for (int64_t k = F + threadIdx.x; k < F_pad; k += 256) xo[k] = 0.f;
  }
}

int launch_conv(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                hipStream_t s) {
  const Dims& dm = h->dm;
  const float* rel_emb = dm.lookup ? nullptr : h->params["rel_emb"].ptr;
  const float* cw = dm.gen_conv ? h->conv_w_rel : h->params["conv1_weights"].ptr;
  const float* cb = dm.gen_conv ? h->conv_b_rel : h->params["conv1_bias"].ptr;
  ScopedKernelTimer t(h, "conv", s);
  if (dm.fh == 3 && dm.fw == 3 && dm.C % 32 == 0) {
    constexpr int QPB = 4;
    size_t lds = sizeof(float) * (size_t)QPB * dm.in_h * dm.in_w;
    hipLaunchKernelGGL((k_conv3x3_bn_relu<QPB>), dim3((unsigned)((B + QPB - 1) / QPB)), dim3(256), lds, s, e1, rel,
                       e1_rows, h->perm, h->params["ent_emb"].ptr, (int64_t)h->cfg.shard_lo, dm.n_local, rel_emb, cw, cb,
                       dm.gen_conv ? 1 : 0, h->conv_scale, h->conv_shift, dm.d, dm.r, dm.in_h, dm.in_w,
                       dm.stacked ? 1 : 0, dm.C, dm.Ho, dm.Wo, dm.concat_rel ? 1 : 0, dm.F, dm.F_pad, dm.R, B,
                       h->x_sorted);
  } else {
    size_t lds = sizeof(float) * ((size_t)dm.in_h * dm.in_w + (size_t)dm.fh * dm.fw * dm.C + 3 * (size_t)dm.C);
    hipLaunchKernelGGL(k_conv_bn_relu, dim3((unsigned)B), dim3(256), lds, s, e1, rel, e1_rows, h->perm,
                       h->params["ent_emb"].ptr, (int64_t)h->cfg.shard_lo, dm.n_local, rel_emb, cw, cb,
                       dm.gen_conv ? 1 : 0, h->conv_scale, h->conv_shift, dm.d, dm.r, dm.emb_w, dm.in_h, dm.in_w,
                       dm.stacked ? 1 : 0, dm.fh, dm.fw, dm.C, dm.Ho, dm.Wo, dm.concat_rel ? 1 : 0, dm.F, dm.F_pad,
                       dm.R, h->x_sorted);
  }
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ------------------------------------------------------------------------------------------------
// dense: Z^T[d, n] = W_rel^T[d, F] . X^T[F, n] per tile, v_mfma_f32_16x16x4_f32 (exact f32)
//   A operand (features on rows):  lane l holds W[f = 16ks + 4(l>>4) + t][feat = 16fb + (l&15)]  (Wf float4, t = component)
//   B operand (queries on columns): lane l holds x[q = l&15][f = 16ks + 4(l>>4) + t]
//   MFMA t contracts k-group {4g + t : g = 0..3}; D: col = lane&15 (query), row = 4(lane>>4) + reg (feature).
// K = F is cut into DENSE_KSLICES fixed slices (a function of F only); every (query, feature, slice)
// partial is ONE accumulator chain over the slice's k-steps in order, whichever kernel produces it, and
// k_dense_finalize adds the slices in order: h[b] does not depend on the batch it was computed in.
//
//   k_dense_small_f32  tiles of <= 32 queries (HBM-bound: n/2 flop per weight byte): one wave per slice,
//                      weights stream from Wf straight into VGPRs (13 KiB per wave per k-step), no LDS.
//   k_dense_big_f32    tiles of 33..128 queries: 4 waves x 32 queries share the weight stream through an
//                      LDS double buffer filled by LDS-DMA (global_load_lds, 1 KiB per wave-instruction,
//                      the fragment image is lane-linear so no swizzle is needed); the x fragments take
//                      the same route (per-lane gather addresses).  One slice per workgroup (grid.y).
// ------------------------------------------------------------------------------------------------
// DENSE_KSLICES = 4 when F_pad/16 >= 64 else 1 (chosen in coper_encode from F only)

#define F4T(v, t) ((t) == 0 ? (v).x : (t) == 1 ? (v).y : (t) == 2 ? (v).z : (v).w)

template <int NFB>
__global__ __launch_bounds__(256) void k_dense_small_f32(const float4* __restrict__ Wf,
                                                         const float* __restrict__ x_sorted,
                                                         const int32_t* __restrict__ tiles,
                                                         const int32_t* __restrict__ n_tiles, int nfb, int64_t ksteps,
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
  const int64_t kb = ksteps * slice / nslices, ke = ksteps * (slice + 1) / nslices;

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
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(F4T(av[a], t), F4T(bv[0], t), acc[a][0], 0, 0, 0);
        if (second) acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(F4T(av[a], t), F4T(bv[1], t), acc[a][1], 0, 0, 0);
      }
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

#ifndef COPER_DENSE_NSTAGE
#define COPER_DENSE_NSTAGE 3
#endif
#ifndef COPER_DENSE_GI
#define COPER_DENSE_GI 4
#endif
// One tile of NB 16-query blocks: the NFB x NB (feature block, query block) accumulator tiles are dealt
// to the 4 waves as contiguous ranges of the fb-major item list, so every wave issues the same number
// of MFMAs (within one item) whatever the tile size.  Operands reach LDS by LDS-DMA through a 3-stage
// ring: two stages are in flight while one is consumed; every wave issues exactly L DMAs per stage (the
// slot list is padded with dummies) so one counted `s_waitcnt vmcnt(L)` + one raw s_barrier per k-step
// retires a stage without draining the younger ones.
template <int NFB, int NB>
__device__ __forceinline__ void dense_big_body(float4* __restrict__ ldsA, const float4* __restrict__ Wf,
                                               const float* __restrict__ x_sorted, int64_t relw, int start, int n,
                                               int fb0, int nfb, int64_t ksteps, int64_t F_pad, int64_t kb, int64_t ke,
                                               float* __restrict__ zdst /* z_part + slice*Bcap*d_pad16 */, int d_pad16) {
  constexpr int L = (NFB + NB + 3) / 4;   // DMA slots per wave per stage
  constexpr int STAGE = 4 * L * 64;       // float4 per ring stage (incl. dummy slots)
  constexpr int NSTAGE = COPER_DENSE_NSTAGE;
  constexpr int T = NFB * NB;
  constexpr int MAXI = (T + 3) / 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i0 = T * wave / 4, cnt = T * (wave + 1) / 4 - i0;

  // slot s = wave + 4i: s < NFB -> weight fragment fb = s (1 KiB per k-step);
  //                     s < NFB + NB -> x fragment of query block s - NFB (per-lane gather, 64 B per k-step);
  //                     else dummy (re-reads slot `wave` into its own ring slot).
  const char* src[L];
  int stride[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    int sl = wave + 4 * i;
    if (sl >= NFB + NB) sl = wave;
    if (sl < NFB) {
      int fb = fb0 + sl < nfb ? fb0 + sl : nfb - 1;
      src[i] = (const char*)(Wf + ((relw * nfb + fb) * ksteps + kb) * 64 + lane);
      stride[i] = 1024;
    } else {
      int qi = (sl - NFB) * 16 + (lane & 15);
      if (qi > n - 1) qi = n - 1;
      src[i] = (const char*)(x_sorted + (int64_t)(start + qi) * F_pad + 16 * kb + 4 * (lane >> 4));
      stride[i] = 64;
    }
  }
#ifdef COPER_DBG_DENSE_NO_DMA
#define DMA_N 1
#else
#define DMA_N L
#endif
#define STAGE_ISSUE(buf, kk)                                                                                      \
  {                                                                                                               \
    float4* dstb = ldsA + (buf)*STAGE;                                                                            \
    _Pragma("unroll") for (int i = 0; i < DMA_N; ++i) __builtin_amdgcn_global_load_lds(                           \
        (const __attribute__((address_space(1))) void*)(src[i] + (int64_t)(kk)*stride[i]),                        \
        (__attribute__((address_space(3))) void*)(dstb + (wave + 4 * i) * 64), 16, 0, 0);                         \
  }

  int offA[MAXI], offB[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    int item = cnt > 0 ? i0 + (i < cnt ? i : cnt - 1) : 0;  // a wave short of items recomputes one (discarded)
    offA[i] = (item / NB) * 64 + lane;
    offB[i] = (NFB + item % NB) * 64 + lane;
  }
  f32x4 acc[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (int)(ke - kb);
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) STAGE_ISSUE(st, st < nk ? st : nk - 1);
  int cur = 0, nxt = NSTAGE - 1;
  for (int k = 0; k < nk; ++k) {
    // stage k has landed once at most the (NSTAGE-2)*L youngest DMAs (stages k+1..) are outstanding
#ifdef COPER_DBG_DENSE_NO_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    {
      constexpr int N = (NSTAGE - 2) * L;
      static_assert(N <= 24, "vmcnt immediate");
      if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    __builtin_amdgcn_s_barrier();  // every wave's share of stage k is in LDS; everyone is done with stage k-1
    {
      int kk = k + NSTAGE - 1 < nk ? k + NSTAGE - 1 : nk - 1;  // past the end: harmless re-load into the ring slot nobody reads
      STAGE_ISSUE(nxt, kk);
    }
    const float4* ab = ldsA + cur * STAGE;
    // groups of GI items: GI independent accumulators between two MFMAs of one chain (the 16x16x4 f32
    // MFMA issues every 32 cycles but its dependent latency is 40), fragments of a group read together
    constexpr int GI = COPER_DENSE_GI;
#pragma unroll
    for (int g = 0; g < MAXI; g += GI) {
      float4 av[GI], bv[GI];
#pragma unroll
      for (int j = 0; j < GI; ++j)
        if (g + j < MAXI) { av[j] = ab[offA[g + j]]; bv[j] = ab[offB[g + j]]; }
#ifdef COPER_DBG_DENSE_NO_MFMA
#pragma unroll
      for (int j = 0; j < GI; ++j)
        if (g + j < MAXI) { acc[g + j][0] += av[j].x * bv[j].x; }
#else
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < GI; ++j)
          if (g + j < MAXI)
            acc[g + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(F4T(av[j], t), F4T(bv[j], t), acc[g + j], 0, 0, 0);
#endif
    }
    cur = cur == NSTAGE - 1 ? 0 : cur + 1;
    nxt = nxt == NSTAGE - 1 ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the workgroup's LDS
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
__global__ __launch_bounds__(256) void k_dense_big_f32(const float4* __restrict__ Wf,
                                                       const float* __restrict__ x_sorted,
                                                       const int32_t* __restrict__ tiles,
                                                       const int32_t* __restrict__ n_tiles, int64_t cap_small,
                                                       int nfb, int64_t ksteps, int64_t F_pad, int nslices,
                                                       int64_t Bcap, int d_pad16, float* __restrict__ z_part) {
  extern __shared__ float4 ldsA[];  // ring of [4L][64] float4 stages
  int tile = blockIdx.x;
  if (tile >= n_tiles[1]) return;
  const int32_t* tl = tiles + 4 * (cap_small + tile);
  const int slice = blockIdx.y;
  const int fb0 = blockIdx.z * NFB;
  const int64_t relw = __builtin_amdgcn_readfirstlane(tl[0]);
  const int start = __builtin_amdgcn_readfirstlane(tl[1]);
  const int n = __builtin_amdgcn_readfirstlane(tl[2]);
  const int64_t kb = ksteps * slice / nslices, ke = ksteps * (slice + 1) / nslices;
  float* zdst = z_part + (int64_t)slice * Bcap * d_pad16;
  const int nb = (n + 15) >> 4;  // 3..8 (tiles hold 33..128 queries)
#define BODY(NB_) dense_big_body<NFB, NB_>(ldsA, Wf, x_sorted, relw, start, n, fb0, nfb, ksteps, F_pad, kb, ke, zdst, d_pad16)
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

__global__ void k_dense_finalize(const float* __restrict__ z_part, int ksplit, int64_t Bcap, int64_t B, int d,
                                 int d_pad16, const int32_t* __restrict__ perm, const int64_t* __restrict__ rel,
                                 const float* __restrict__ fc_b, int per_rel_bias, int64_t R,
                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                 float* __restrict__ h_out, const int32_t* __restrict__ w_exp, int x_exp) {
  // one thread per 4 consecutive features of one query (d_pad16 is a multiple of 4: 16-B partial loads)
  const int nq4 = d_pad16 >> 2;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * nq4) return;
  int64_t pos = idx / nq4;
  int k0 = (int)(idx % nq4) * 4;
  if (k0 >= d) return;
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < ksplit; ++s) {
    float4 p = *(const float4*)(z_part + ((int64_t)s * Bcap + pos) * d_pad16 + k0);
    z.x += p.x; z.y += p.y; z.z += p.z; z.w += p.w;
  }
  int64_t q = perm[pos];
  int64_t rid = rel[q];
  if (rid < 0 || rid >= R) rid = 0;
  const float* bsrc = per_rel_bias ? fc_b + rid * d : fc_b;
  float zz[4] = {z.x, z.y, z.z, z.w};
  // x3 encoder: the partial sums carry 2^(e_W + e_x) (split16.h); exact to take out.  (fp32 encoder: w_exp == NULL, x_exp == 0)
  const int zexp = -((w_exp ? w_exp[per_rel_bias ? rid : 0] : 0) + x_exp);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int k = k0 + c;
    if (k < d) {
      float v = __builtin_ldexpf(zz[c], zexp) + bsrc[k];
      v = fmaf(v, scale[k], shift[k]);
      h_out[q * d + k] = fmaxf(v, 0.f);
    }
  }
}

template <int NFB>
static void dense_launch(coper_handle* h, int64_t B, int nslices, int zgroups, hipStream_t s) {
  const Dims& dm = h->dm;
  int64_t cap_small = small_tile_cap(h);
  int64_t n_small_max = cap_small - 1 < B ? cap_small - 1 : B;
  int64_t n_big_max = B / 33 + 1;
  if (n_small_max > 0)
    hipLaunchKernelGGL((k_dense_small_f32<NFB>), dim3((unsigned)n_small_max, (unsigned)((nslices + 3) / 4), (unsigned)zgroups),
                       dim3(256), 0, s, (const float4*)h->Wf, h->x_sorted, h->tiles, h->n_tiles, dm.nfb, dm.F_pad / 16,
                       dm.F_pad, nslices, h->ws_queries, dm.d_pad16, h->z_part);
  if (B > 32) {
    size_t lds = (size_t)COPER_DENSE_NSTAGE * (((NFB + 8 + 3) / 4) * 4) * 64 * sizeof(float4);  // ring, max NB = 8
    if (!h->dense_attr_done) {
      (void)hipFuncSetAttribute((const void*)k_dense_big_f32<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      h->dense_attr_done = true;
    }
    hipLaunchKernelGGL((k_dense_big_f32<NFB>), dim3((unsigned)n_big_max, (unsigned)nslices, (unsigned)zgroups), dim3(256),
                       lds, s, (const float4*)h->Wf, h->x_sorted, h->tiles, h->n_tiles, cap_small, dm.nfb, dm.F_pad / 16,
                       dm.F_pad, nslices, h->ws_queries, dm.d_pad16, h->z_part);
  }
}

int launch_dense(coper_handle* h, const int64_t* rel, int64_t B, int tq, int ksplit, float* h_out, hipStream_t s) {
  const Dims& dm = h->dm;
  (void)tq;
  {
    ScopedKernelTimer t(h, "dense", s);
    int nfb = dm.nfb;
    if (nfb == 13) dense_launch<13>(h, B, ksplit, 1, s);
    else if (nfb <= 2) dense_launch<2>(h, B, ksplit, 1, s);
    else if (nfb <= 4) dense_launch<4>(h, B, ksplit, 1, s);
    else dense_launch<8>(h, B, ksplit, (nfb + 7) / 8, s);
    COPER_HIP_TRY(h, hipGetLastError());
  }
  return launch_dense_finalize(h, rel, B, ksplit, h_out, s);
}

int launch_dense_finalize(coper_handle* h, const int64_t* rel, int64_t B, int ksplit, float* h_out, hipStream_t s) {
  const Dims& dm = h->dm;
  const float* fcb = dm.gen_fc ? h->fc_b_rel : h->params["fc_bias"].ptr;
  int64_t total = B * (dm.d_pad16 / 4);
  hipLaunchKernelGGL(k_dense_finalize, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->z_part, ksplit,
                     h->ws_queries, B, dm.d, dm.d_pad16, h->perm, rel, fcb, dm.gen_fc ? 1 : 0, dm.R, h->fc_scale,
                     h->fc_shift, h_out, h->enc_bf16 ? h->w_exp : nullptr, h->enc_bf16 ? h->x_exp : 0);
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

// int32 -> int64 (coper_widen_ids): 16 bytes in, 32 out per thread; src may live in pinned host memory
__global__ __launch_bounds__(256) void k_widen_ids(const int32_t* __restrict__ src, int64_t n, int64_t* __restrict__ dst) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 4 <= n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
    const int4 v = *(const int4*)(src + i4);
    longlong2* o = (longlong2*)(dst + i4);
    o[0] = make_longlong2(v.x, v.y);
    o[1] = make_longlong2(v.z, v.w);
  } else {
    for (int64_t i = i4; i < i4 + 4 && i < n; ++i) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(256) void k_copy_i32(const int32_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 4 <= n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
    *(int4*)(dst + i4) = *(const int4*)(src + i4);
  } else {
    for (int64_t i = i4; i < i4 + 4 && i < n; ++i) dst[i] = src[i];
  }
}

// the ranks of a pass AND the two words of its band audit (behind them in dst), the audit reset for the next pass: one launch for
// what coper_copy_out_i32 + coper_band_audit_post do in two launches and a memset (coper_post_ranks_audit)
__global__ __launch_bounds__(256) void k_copy_i32_audit(const int32_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst,
                                                        unsigned* __restrict__ audit, int reset) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 4 <= n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
    *(int4*)(dst + i4) = *(const int4*)(src + i4);
  } else {
    for (int64_t i = i4; i < i4 + 4 && i < n; ++i) dst[i] = src[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned a0 = 0u, a1 = 0u;
    if (audit) {
      a0 = audit[0]; a1 = audit[1];
      if (reset) { audit[0] = 0u; audit[1] = 0u; }
    }
    dst[n] = (int32_t)a0; dst[n + 1] = (int32_t)a1;
  }
}

int launch_copy_i32_audit(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst, unsigned* audit, int reset, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_i32_audit, dim3((unsigned)((n + 1023) / 1024 > 0 ? (n + 1023) / 1024 : 1)), dim3(256), 0, s, src, n, dst, audit, reset);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_copy_i32(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_i32, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, src, n, dst);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ---- step 1 of the entity-sharded exchange: the rows a shard OWNS among ent_emb[e1], ent_emb[e2] packed for the all-gather, and
// the gathered rows handed out to the queries -- one launch each for what were ~7 and ~5 torch launches per chunk ----
// buf[cap + 1][d + 1]: row 0 = the header { hdr0, hdr1, 0... }, row 1 + i = { ent_emb[loc[i]][0..d), pred_bias[loc[i]] }, zero beyond n
__global__ __launch_bounds__(256) void k_pack_owned_rows(const float* __restrict__ ent, const float* __restrict__ bias, const int64_t* __restrict__ loc,
                                                         int64_t n, int64_t cap, int d, int64_t n_local, float hdr0, float hdr1,
                                                         float* __restrict__ buf, int32_t* __restrict__ bad) {
  const int W = d + 1;
  const int64_t total = (cap + 1) * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / W;
    const int c = (int)(i - r * W);
    float v = 0.f;
    if (r == 0) v = c == 0 ? hdr0 : c == 1 ? hdr1 : 0.f;
    else if (r - 1 < n) {
      // a row number outside the shard: a ZERO row (what coper_gather_entities hands out for an id the shard does not hold) and one
      // more bad id for coper_check_ids -- round 5 clamped it to a real entity, silently
      const int64_t row = loc[r - 1];
      if (row < 0 || row >= n_local) {
        if (c == 0 && bad) atomicAdd(bad, 1);
      } else {
        v = c < d ? ent[row * d + c] : bias[row];
      }
    }
    buf[i] = v;
  }
}

// g1[b] = all[take1[b]][0..d), g2[b] = all[take2[b]][0..d), b2[b] = all[take2[b]][d]
__global__ __launch_bounds__(256) void k_unpack_rows(const float* __restrict__ all, const int64_t* __restrict__ take1, const int64_t* __restrict__ take2,
                                                     int64_t B, int d, float* __restrict__ g1, float* __restrict__ g2, float* __restrict__ b2) {
  const int W = d + 1;
  const int64_t total = B * (2 * (int64_t)d + 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / (2 * d + 1);
    const int c = (int)(i - b * (2 * d + 1));
    if (c < d) g1[b * d + c] = all[take1[b] * W + c];
    else if (c < 2 * d) g2[b * d + (c - d)] = all[take2[b] * W + (c - d)];
    else b2[b] = all[take2[b] * W + d];
  }
}

int launch_pack_owned_rows(coper_handle* h, const float* ent, const float* bias, const int64_t* loc, int64_t n, int64_t cap, float hdr0, float hdr1,
                           float* buf, hipStream_t s) {
  const int64_t total = (cap + 1) * (h->dm.d + 1);
  int64_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_pack_owned_rows, dim3((unsigned)nb), dim3(256), 0, s, ent, bias, loc, n, cap, h->dm.d, h->dm.n_local, hdr0, hdr1, buf,
                     h->rel_count ? h->rel_count + h->dm.R + 1 : nullptr);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_unpack_rows(coper_handle* h, const float* all, const int64_t* take1, const int64_t* take2, int64_t B, float* g1, float* g2, float* b2,
                       hipStream_t s) {
  const int64_t total = B * (2 * (int64_t)h->dm.d + 1);
  int64_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)nb), dim3(256), 0, s, all, take1, take2, B, h->dm.d, g1, g2, b2);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

// ---- the entity-sharded exchange's record (coper_amd/sharding.py, step 3) packed and merged on the device ----
// rec[B + 1][1 + 2 k] int64: row b = { n_greater << 32 | n_equal, the k top scores' float bits, the k ids }, row B = the
// shard's band-audit words { ratio bits << 32 | pairs } (read from the handle and reset: no host round trip).  One launch for
// what were ~10 torch launches and a synchronising audit read per chunk.
__global__ __launch_bounds__(256) void k_pack_shard_record(const int32_t* __restrict__ ng, const int32_t* __restrict__ ne,
                                                           const float* __restrict__ tv, const int64_t* __restrict__ ti, int64_t B, int k,
                                                           unsigned* __restrict__ audit, int reset, int64_t* __restrict__ rec) {
  const int W = 1 + 2 * k;
  const int64_t n = (B + 1) * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / W;
    const int c = (int)(i - b * W);
    int64_t v = 0;
    if (b < B) {
      if (c == 0) v = ((int64_t)ng[b] << 32) | (int64_t)(uint32_t)ne[b];
      else if (c <= k) v = (int64_t)(int32_t)__float_as_uint(tv[b * k + (c - 1)]);      // (sign-extended, as .view(int32).to(int64) gives)
      else v = ti[b * k + (c - 1 - k)];
    } else if (c == 0 && audit) {
      const unsigned a0 = audit[0], a1 = audit[1];
      v = ((int64_t)a0 << 32) | (int64_t)(a1 > 0x7fffffffu ? 0x7fffffffu : a1);
      if (reset) { audit[0] = 0u; audit[1] = 0u; }
    }
    rec[i] = v;
  }
}

// all[world][B + 1][1 + 2 k] -> ranks = 1 + sum of n_greater, n_equal summed, the candidates side by side: vals / ids [B][world k]
__global__ __launch_bounds__(256) void k_merge_shard_records(const int64_t* __restrict__ all, int world, int64_t B, int k,
                                                             int32_t* __restrict__ ranks, int32_t* __restrict__ ne,
                                                             float* __restrict__ vals, int64_t* __restrict__ ids) {
  const int W = 1 + 2 * k;
  const int64_t per = (int64_t)world * k + 1;          // work items per query: the counts, then every (shard, j)
  const int64_t n = B * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per;
    const int64_t r = i - b * per;
    if (r == 0) {
      int64_t g = 0, e = 0;
      for (int w = 0; w < world; ++w) {
        const int64_t v = all[((int64_t)w * (B + 1) + b) * W];
        g += v >> 32;
        e += v & 0xFFFFFFFFll;
      }
      ranks[b] = (int32_t)(1 + g);
      if (ne) ne[b] = (int32_t)e;
    } else {
      const int64_t wj = r - 1;
      const int w = (int)(wj / k), j = (int)(wj - (int64_t)w * k);
      const int64_t* row = all + ((int64_t)w * (B + 1) + b) * W;
      vals[b * ((int64_t)world * k) + wj] = __uint_as_float((uint32_t)row[1 + j]);
      ids[b * ((int64_t)world * k) + wj] = row[1 + k + j];
    }
  }
}

int launch_pack_shard_record(coper_handle* h, const int32_t* ng, const int32_t* ne, const float* tv, const int64_t* ti, int64_t B, int k,
                             unsigned* audit, int reset, int64_t* rec, hipStream_t s) {
  const int64_t n = (B + 1) * (1 + 2 * (int64_t)k);
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_pack_shard_record, dim3((unsigned)nb), dim3(256), 0, s, ng, ne, tv, ti, B, k, audit, reset, rec);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_merge_shard_records(coper_handle* h, const int64_t* all, int world, int64_t B, int k, int32_t* ranks, int32_t* ne, float* vals,
                               int64_t* ids, hipStream_t s) {
  const int64_t n = B * ((int64_t)world * k + 1);
  int64_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_merge_shard_records, dim3((unsigned)nb), dim3(256), 0, s, all, world, B, k, ranks, ne, vals, ids);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
}

int launch_widen_ids(coper_handle* h, const int32_t* src, int64_t n, int64_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_widen_ids, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, src, n, dst);
  COPER_HIP_TRY(h, hipGetLastError());
  return COPER_OK;
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
