// Grouping bodies shared by the grouping kernels (kernels_encode.hip) and the grouping role of the fused encoder
// (kernels_dense_fused_bf16.hip: coper_group_next).
#pragma once
#include "coper_internal.h"

namespace coper {

// single block: exclusive scan of the counts -> offsets, and the two tile lists:
//   small tiles: one per relation group with 1..32 queries          (k_dense_small_f32)
//   big tiles:   groups with > 32 queries cut into ceil(c/128) balanced tiles of <= 128 (k_dense_big_f32),
//                emitted largest-first (by 16-query block count) so the hardware dispatcher, which hands
//                out workgroups in index order, ends the launch on the cheapest tiles.
// tiles[] = small list at [0, 4*cap_small), big list after it; n_tiles[0] = #small, n_tiles[1] = #big.
// NT threads; `scratch`: REL_SCAN_SCRATCH(NT) ints of LDS (the callers with a fixed block size keep them static, the grouping
// role of the fused encoder carves them out of its launch's dynamic LDS).
#define REL_SCAN_SCRATCH(NT) (2 * (NT) + 2 + 27 + 3)
template <int NT>
__device__ __forceinline__ void rel_scan_tiles_body_t(const int32_t* count, int64_t R, int64_t cap_small,
                                                      int32_t* offset, int32_t* __restrict__ tiles,
                                                      int32_t* __restrict__ n_tiles, int* __restrict__ scratch) {
  int* s_cnt = scratch;
  int* s_sml = scratch + NT;
  int& carry_cnt = scratch[2 * NT];
  int& carry_sml = scratch[2 * NT + 1];
  int* cls_count = scratch + 2 * NT + 2;
  int* cls_base = cls_count + 9;
  int* cls_cursor = cls_base + 9;
  if (threadIdx.x == 0) { carry_cnt = 0; carry_sml = 0; }
  if (threadIdx.x < 9) { cls_count[threadIdx.x] = 0; cls_cursor[threadIdx.x] = 0; }
  __syncthreads();
  int32_t* tiles_big = tiles + 4 * cap_small;
  // pass 1: offsets, small tiles, and the number of big tiles per size class
  for (int64_t base = 0; base < R; base += NT) {
    int64_t rid = base + threadIdx.x;
    int c = rid < R ? count[rid] : 0;
    int ns = (c > 0 && c <= 32) ? 1 : 0;
    // inclusive scan of (c, ns) over the NT threads: shuffles inside a wave, the NT / 64 wave totals through LDS
    // (two barriers; a Hillis-Steele scan over LDS took twenty)
    {
      int ic = c, is = ns;
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(ic, off), b2 = __shfl_up(is, off);
        if (lane >= off) { ic += a; is += b2; }
      }
      if (lane == 63) { s_cnt[wv] = ic; s_sml[wv] = is; }
      __syncthreads();
      int pc = 0, ps = 0;
      for (int w2 = 0; w2 < wv; ++w2) { pc += s_cnt[w2]; ps += s_sml[w2]; }
      __syncthreads();
      s_cnt[threadIdx.x] = ic + pc;
      s_sml[threadIdx.x] = is + ps;
      __syncthreads();
    }
    int excl_c = carry_cnt + s_cnt[threadIdx.x] - c;
    int excl_s = carry_sml + s_sml[threadIdx.x] - ns;
    if (rid < R) {
      offset[rid] = excl_c;
      if (ns) {
        int32_t* t = tiles + 4 * (int64_t)excl_s;
        t[0] = (int32_t)rid; t[1] = excl_c; t[2] = c; t[3] = 0;
      }
      if (c > 32) {
        int nb = (c + 127) / 128, bsz = c / nb, rem = c % nb;
        if (rem) atomicAdd(&cls_count[(bsz + 1 + 15) >> 4], rem);
        atomicAdd(&cls_count[(bsz + 15) >> 4], nb - rem);
      }
    }
    __syncthreads();
    if (threadIdx.x == NT - 1) { carry_cnt += s_cnt[NT - 1]; carry_sml += s_sml[NT - 1]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 8; k >= 0; --k) { cls_base[k] = run; run += cls_count[k]; }
    offset[R] = carry_cnt;
    n_tiles[0] = carry_sml;
    n_tiles[1] = run;
  }
  __syncthreads();
  // pass 2: place the big tiles, class by class (order inside a class is irrelevant to the results)
  for (int64_t rid = threadIdx.x; rid < R; rid += NT) {
    int c = count[rid];
    if (c <= 32) continue;
    int nb = (c + 127) / 128, bsz = c / nb, rem = c % nb;
    int off = offset[rid];
    for (int j = 0; j < nb; ++j) {
      int sz = bsz + (j < rem ? 1 : 0);
      int k = (sz + 15) >> 4;
      int slot = cls_base[k] + atomicAdd(&cls_cursor[k], 1);
      int32_t* t = tiles_big + 4 * (int64_t)slot;
      t[0] = (int32_t)rid;
      t[1] = off + j * bsz + (j < rem ? j : rem);
      t[2] = sz;
      t[3] = nb > 1 ? 1 : 0;   // the relation has several tiles: its weights are worth caching (k_dense_fused_bf16x3)
    }
  }
}


__device__ __forceinline__ void rel_scan_tiles_body(const int32_t* count, int64_t R, int64_t cap_small,
                                                    int32_t* offset, int32_t* __restrict__ tiles,
                                                    int32_t* __restrict__ n_tiles) {
  __shared__ int s_scan[REL_SCAN_SCRATCH(1024)];
  rel_scan_tiles_body_t<1024>(count, R, cap_small, offset, tiles, n_tiles, s_scan);
}


// coper_group_next: the grouping of the NEXT pass's batch, done by ONE workgroup of NT threads inside a launch of the pass that
// runs (the fused encoder's: kernels_dense_fused_bf16.hip), into a set of grouping arrays the running pass does not read.  Same
// outputs as k_rel_group_single / k_rel_group_identity (the order inside a relation group is whatever the LDS cursors hand out:
// h[b] does not depend on it).  The ids are the int64 device arrays of that pass; when a staging job of the same launch is
// still bringing them in, the caller waits for its workgroups first (ticket / wait_for).
// The guard of a prepared grouping.  The grouping role leaves what it sorted in GROUP_CHK_WORDS 64-bit words
// of its set; the tiles of the pass that CONSUMES the set compare every query's live ids with its sorted_row / sorted_rid
// (group_chk_issue / group_chk_verdict: the id arrays at the registered addresses may have been rewritten between the two launches -- the natural
// mistake with two staging buffers) and raise GROUP_CHK_STALE; the kernel that presets the pass's rank counters then writes
// COPER_RANK_STALE into every rank instead of 1 (a stale grouping never reaches the ranks) and counts the pass (coper_stale_passes).
// (the word indices GROUP_CHK_*: coper_internal.h)

// one lane's query of a consuming tile: sorted position `pos`, what the set holds for it (row, rid).  Ids are read past the XCD's L2
// (the arrays may have been written by a staging job of the launch before: the same coherent loads the grouping role uses).
__device__ __forceinline__ void group_chk_issue(const int64_t* __restrict__ chk, int pos, int64_t& live_rel, int64_t& live_e1) {
  const int64_t* rel64 = (const int64_t*)chk[GROUP_CHK_REL];
  const int64_t* e1_64 = (const int64_t*)chk[GROUP_CHK_E1];
  const int32_t* perm = (const int32_t*)chk[GROUP_CHK_PERM];
  const int b = perm[pos];
  live_rel = __hip_atomic_load(rel64 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  live_e1 = e1_64 ? __hip_atomic_load(e1_64 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
}
__device__ __forceinline__ void group_chk_verdict(int64_t* __restrict__ chk, int64_t live_rel, int64_t live_e1, int row, int rid) {
  int64_t r = live_rel;
  if (r < 0 || r >= chk[GROUP_CHK_RALL]) r = 0;
  bool bad = (int)r != rid;
  if (chk[GROUP_CHK_E1]) {
    int64_t lr = live_e1 - chk[GROUP_CHK_LO];
    if (lr < 0 || lr >= chk[GROUP_CHK_NLOCAL]) lr = -1;
    bad |= (int)lr != row;
  }
  if (bad) atomicOr((int*)(chk + GROUP_CHK_STALE), 1);
}

struct GroupJob {
  const int64_t* rel64; const int64_t* e1_64;
  int32_t* ticket; int wait_for, pad2;     // the staging workgroups of the launch that bring the ids in (0: they are resident)
  int64_t front;                           //   ... and how many leading elements of their job the ids lie in
  int64_t B, R, R_all, shard_lo, n_local, cap_small;
  int32_t* count; int32_t* offset; int32_t* tiles; int32_t* n_tiles; int32_t* perm; int32_t* sorted_row; int32_t* sorted_rid; int32_t* inv_perm;
  float* x3m;
  int64_t* chk;                            // the set's check words (GROUP_CHK_*), written here, read by the consuming pass
  int use_rel, have_e1_rows, x3m_slots, pad;
};

// LDS ints the role needs: counters and cursors per relation key, and the scan's scratch
__host__ __device__ inline size_t group_role_lds_ints(int64_t R, int nt) { return (size_t)(R > 1 ? 2 * R : 0) + REL_SCAN_SCRATCH(nt); }

#ifndef GROUP_CLK
#define GROUP_CLK(i) do { } while (0)
#endif
template <int NT>
__device__ __forceinline__ void group_role_body(const GroupJob& J, int* __restrict__ lds) {
  const int64_t B = J.B;
  if (J.x3m) for (int i = threadIdx.x; i < J.x3m_slots; i += NT) J.x3m[i] = 0.f;
  if (J.chk && threadIdx.x == 0) {
    J.chk[GROUP_CHK_REL] = (int64_t)J.rel64; J.chk[GROUP_CHK_E1] = J.have_e1_rows ? 0 : (int64_t)J.e1_64; J.chk[GROUP_CHK_PERM] = (int64_t)J.perm;
    J.chk[GROUP_CHK_LO] = J.shard_lo; J.chk[GROUP_CHK_NLOCAL] = J.n_local; J.chk[GROUP_CHK_RALL] = J.R_all; J.chk[GROUP_CHK_STALE] = 0;
  }
  // (ids a staging job of this launch wrote: loads that do not trust a line the XCD's L2 may hold from the last pass)
  const bool coh = J.wait_for > 0;
  auto ld_id = [&](const int64_t* p) -> int64_t { return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
  auto ld_rel = [&](int64_t b) -> int64_t { return ld_id(J.rel64 + b); };
  auto ld_row = [&](int64_t b) -> int64_t {
    if (J.have_e1_rows) return b;
    int64_t row = ld_id(J.e1_64 + b) - J.shard_lo;
    return (row < 0 || row >= J.n_local) ? -1 : row;
  };
  if (J.R == 1) {        // one relation key: the identity (k_rel_group_identity)
    for (int64_t b = threadIdx.x; b < B; b += NT) {
      int64_t rid = ld_rel(b);
      if (rid < 0 || rid >= J.R_all) rid = 0;
      const int64_t row = ld_row(b);
      J.perm[b] = (int32_t)b; J.inv_perm[b] = (int32_t)b; J.sorted_row[b] = (int32_t)row; J.sorted_rid[b] = (int32_t)rid;
    }
    const int c = (int)B;
    if (threadIdx.x == 0) {
      J.count[0] = c; J.count[J.R_all + 1] = 0;
      J.offset[0] = 0; J.offset[1] = c;
      J.n_tiles[0] = (c > 0 && c <= 32) ? 1 : 0;
      J.n_tiles[1] = c > 32 ? (c + 127) / 128 : 0;
      if (c > 0 && c <= 32) { J.tiles[0] = 0; J.tiles[1] = 0; J.tiles[2] = c; J.tiles[3] = 0; }
    }
    if (c > 32) {
      const int nb = (c + 127) / 128, bsz = c / nb, rem = c % nb;
      int32_t* tiles_big = J.tiles + 4 * J.cap_small;
      for (int j = threadIdx.x; j < nb; j += NT) {
        int32_t* t = tiles_big + 4 * (int64_t)j;
        t[0] = 0; t[1] = j * bsz + (j < rem ? j : rem); t[2] = bsz + (j < rem ? 1 : 0); t[3] = nb > 1 ? 1 : 0;
      }
    }
    return;
  }
  const int64_t R = J.R;
  int* cnt = lds;
  int* cur = lds + R;
  int* scratch = lds + 2 * R;
  for (int64_t k = threadIdx.x; k < 2 * R; k += NT) lds[k] = 0;
  __syncthreads();
  int nbad = 0;
  // (eight ids in flight per thread: over PCIe a load is a ~2 us round trip)
  for (int64_t b0 = threadIdx.x; b0 < B; b0 += 8 * NT) {
    int64_t key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int64_t b = b0 + (int64_t)u * NT; key[u] = b < B ? (J.use_rel ? ld_rel(b) : 0) : 0; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (b0 + (int64_t)u * NT >= B) continue;
      int64_t k = key[u];
      if (k < 0 || k >= R) { ++nbad; k = 0; }
      atomicAdd(&cnt[k], 1);
    }
  }
  if (threadIdx.x == 0) J.count[J.R_all + 1] = 0;
  __syncthreads();
  GROUP_CLK(2);
  if (nbad) atomicAdd(&J.count[J.R_all + 1], nbad);
  for (int64_t k = threadIdx.x; k < R; k += NT) J.count[k] = cnt[k];
  rel_scan_tiles_body_t<NT>(cnt, R, J.cap_small, J.offset, J.tiles, J.n_tiles, scratch);
  GROUP_CLK(3);
  __syncthreads();
  for (int64_t k = threadIdx.x; k < R; k += NT) cur[k] = J.offset[k];     // the cursors start at the group offsets (own writes: visible)
  __syncthreads();
  for (int64_t b0 = threadIdx.x; b0 < B; b0 += 8 * NT) {
    int64_t rid[8], row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + (int64_t)u * NT;
      rid[u] = b < B ? ld_rel(b) : 0;
      row[u] = b < B ? ld_row(b) : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + (int64_t)u * NT;
      if (b >= B) continue;
      int64_t r = rid[u];
      if (r < 0 || r >= J.R_all) r = 0;
      const int64_t key = J.use_rel ? r : 0;
      const int pos = atomicAdd(&cur[key], 1);
      J.perm[pos] = (int32_t)b; J.inv_perm[b] = pos; J.sorted_row[pos] = (int32_t)row[u]; J.sorted_rid[pos] = (int32_t)r;
    }
  }
}

}  // namespace coper
