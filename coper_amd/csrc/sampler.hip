// The reference's two negative samplers (CoPER_ConvE/qa_cpg/data.py:228-311) as ONE launch per training batch (round 6).
//
// The reference draws, per (e1, rel) record, the head of a fresh uniform permutation of all entities (`np.random.permutation`) and lays the
// known tails in front of it; labels are the membership of the looked-up ids in the record's tail list.  Rounds 2 - 5 built the batch from
// a dozen torch launches with two read-backs in between (0.8 - 1.0 ms of host time per 512 x 1000 batch: more than the 0.9 ms training step
// it feeds).  Here a workgroup owns a row:
//   * the record's tails go into an LDS hash set (membership = the labels);
//   * "the first n entries of a uniform permutation of N items" = an ordered uniform sample without replacement, drawn as the reference's
//     result is distributed, not as it is computed:
//       N >= 4 n: uniform draws d_0, d_1, ... from a counter-based generator, a draw kept iff its value has not appeared at an EARLIER
//                 index (LDS hash table value -> smallest index, atomicMin: the result does not depend on which thread ran first), kept
//                 draws ranked by a workgroup scan, rounds of draws until n are kept;
//       N <  4 n: a partial Fisher-Yates shuffle of 0 .. N-1 in LDS by one thread (N < 4 L entries);
//   * one positive per row (data.py:278-311): lookup = [the row's tail | n = L - 1 sampled entities];
//     proportional (data.py:228-277): lookup = [the first `lead` of the record's tails in a fresh random order | L - lead sampled entities],
//     lead as data.py:243-262 computes it; e2 = lookup[0].
// Deterministic in (seed, batch, row).  One deviation from the reference, distributional like the rest (its RNG stream cannot be
// reproduced): every ROW draws its own permutation; the reference draws one per record and gives its rows different windows of it.
#include "coper_internal.h"

namespace coper {
namespace {

__device__ __forceinline__ uint32_t smp_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// 64 uniform bits from (key, counter): two rounds over the words of the key, then the counter folded in twice
__device__ __forceinline__ uint64_t smp_bits(uint32_t k0, uint32_t k1, uint32_t ctr) {
  const uint32_t a = smp_mix(k0 ^ smp_mix(ctr * 0x9E3779B1u + 0x85EBCA77u));
  const uint32_t b = smp_mix(k1 + smp_mix(ctr ^ 0xC2B2AE3Du) + a * 0x27D4EB2Fu);
  return ((uint64_t)smp_mix(a ^ b) << 32) | smp_mix(b + 0x165667B1u + (a >> 3));
}
// uniform in [0, n): the high word of a 64 x 64 product (bias n / 2^64)
__device__ __forceinline__ int64_t smp_below(uint64_t bits, int64_t n) { return (int64_t)__umul64hi(bits, (uint64_t)n); }

__device__ __forceinline__ uint32_t smp_slot(int64_t v, int mask) { return (uint32_t)(((uint64_t)v * 0x9E3779B97F4A7C15ull) >> 40) & (uint32_t)mask; }

struct SampleArgs {
  const int64_t* rec; const int64_t* pos_tail; const int64_t* rec_e1; const int64_t* rec_rel; const int64_t* ip; const int64_t* tails;
  int64_t L, E;
  int proportional, need, n_neg_big;
  int ts_mask;          // tail set: ts_mask + 1 slots (a power of two), 0: no set (every membership test scans the list)
  int h_mask;           // dedup table: h_mask + 1 slots
  int work_ints;        // ints of the work area (dedup table: 2 (h_mask + 1); Fisher-Yates: N)
  uint32_t seed0, seed1, batch;
  int64_t* e1; int64_t* rel; int64_t* e2; int32_t* lookup; float* labels;
};

// dst[0 .. n) (LDS, int64 in two int halves is avoided: values fit 63 bits but N <= 2^31 here, so int32 suffices for indices of tails;
// entity ids may exceed int32 only beyond 2^31 entities, which the lookup's own int32 dtype excludes)
__device__ void smp_distinct(int* __restrict__ dst, int n, int64_t N, int* __restrict__ work, const SampleArgs& A, uint32_t k0, uint32_t k1,
                             int* __restrict__ scan) {
  if (n <= 0) return;      // (uniform)
  const int tid = threadIdx.x;
  if (N < 4 * (int64_t)n) {
    // partial Fisher-Yates: N < 4 n <= 4 L ints of LDS
    for (int i = tid; i < (int)N; i += 256) work[i] = i;
    __syncthreads();
    if (tid == 0)
      for (int i = 0; i < n; ++i) {
        const int j = i + (int)smp_below(smp_bits(k0, k1, (uint32_t)i), N - i);
        const int a = work[i], b = work[j];
        work[i] = b; work[j] = a;
      }
    __syncthreads();
    for (int i = tid; i < n; i += 256) dst[i] = work[i];
    __syncthreads();
    return;
  }
  const int H = A.h_mask + 1;
  int* key = work;
  int* idx = work + H;
  for (int i = tid; i < H; i += 256) { key[i] = -1; idx[i] = 0x7fffffff; }
  __syncthreads();
  int got = 0, base = 0;
  // expected repeats among n draws: n^2 / 2N
  int M = n + 32 + (int)(2.0 * (double)n * (double)n / (double)N);
  for (int round = 0; round < 64 && got < n; ++round) {
    if (base + M > H / 2) M = H / 2 - base;      // (the table never fills beyond a half: N >= 4 n keeps the expected draws at 1.34 n)
    if (M <= 0) break;
    const int C = (M + 255) / 256, t0 = tid * C, t1 = t0 + C < M ? t0 + C : M;
    for (int t = t0; t < t1; ++t) {
      const int v = (int)smp_below(smp_bits(k0, k1, (uint32_t)(base + t)), N);
      uint32_t s = smp_slot(v, A.h_mask);
      for (;;) {
        const int was = atomicCAS(&key[s], -1, v);
        if (was == -1 || was == v) { atomicMin(&idx[s], base + t); break; }
        s = (s + 1) & (uint32_t)A.h_mask;
      }
    }
    __syncthreads();
    int kept = 0;
    for (int t = t0; t < t1; ++t) {
      const int v = (int)smp_below(smp_bits(k0, k1, (uint32_t)(base + t)), N);
      uint32_t s = smp_slot(v, A.h_mask);
      while (key[s] != v) s = (s + 1) & (uint32_t)A.h_mask;
      kept += idx[s] == base + t;
    }
    // exclusive scan of `kept` over the 256 threads (their stretches of t are in thread order)
    scan[tid] = kept;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int add = tid >= o ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    int rank = got + scan[tid] - kept;
    const int total = scan[255];
    for (int t = t0; t < t1; ++t) {
      const int v = (int)smp_below(smp_bits(k0, k1, (uint32_t)(base + t)), N);
      uint32_t s = smp_slot(v, A.h_mask);
      while (key[s] != v) s = (s + 1) & (uint32_t)A.h_mask;
      if (idx[s] == base + t) {
        if (rank < n) dst[rank] = v;
        ++rank;
      }
    }
    __syncthreads();
    got += total;
    base += M;
    M = 2 * (n - got) + 32;
  }
  // (never seen: the table's half filled before n distinct values came) -- finish in index order with values not taken yet
  if (got < n) {
    if (tid == 0) {
      int v = 0;
      for (int r = got; r < n; ++r) {
        for (;; ++v) {
          uint32_t s = smp_slot(v, A.h_mask);
          bool seen = false;
          for (int p = 0; p < H; ++p) {
            if (key[s] == -1) break;
            if (key[s] == v) { seen = true; break; }
            s = (s + 1) & (uint32_t)A.h_mask;
          }
          if (!seen) break;
        }
        dst[r] = v++;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_sample_rows(SampleArgs A) {
  extern __shared__ int smp_lds[];      // tail set [ts_mask + 1] | the row's lookup [L] | dst [L] | scan [256] | work [work_ints]
  const int TS = A.ts_mask ? A.ts_mask + 1 : 0;
  const int L = (int)A.L;
  int* tset = smp_lds;
  int* lkv = tset + TS;
  int* dst = lkv + L;
  int* scan = dst + L;
  int* work = scan + 256;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const int64_t r = A.rec[row];
  const int64_t lo = A.ip[r], hi = A.ip[r + 1];
  const int64_t cnt = hi - lo;
  const bool use_set = TS > 0 && 2 * cnt <= TS;      // (longer lists than the set holds are scanned)
  for (int i = tid; i < TS; i += 256) tset[i] = -1;
  __syncthreads();
  if (use_set)
    for (int64_t i = tid; i < cnt; i += 256) {
      const int v = (int)A.tails[lo + i];
      uint32_t s = smp_slot(v, A.ts_mask);
      for (;;) {
        const int was = atomicCAS(&tset[s], -1, v);
        if (was == -1 || was == v) break;
        s = (s + 1) & (uint32_t)A.ts_mask;
      }
    }
  __syncthreads();
  const uint32_t k0 = A.seed0 ^ smp_mix(A.batch * 0x9E3779B1u + 1u), k1 = A.seed1 + smp_mix((uint32_t)row * 0x85EBCA77u + (uint32_t)(row >> 32) + 7u);
  int lead;
  int64_t e2v;
  if (!A.proportional) {
    lead = 1;
    e2v = A.pos_tail[row];
    if (tid == 0) lkv[0] = (int)e2v;
  } else {
    // data.py:243-262: num_positives_needed = int(L / (1 + prop_negatives)); a record with more tails keeps L - min(|E|, L - needed) of them
    int64_t ld = cnt <= A.need ? cnt : (int64_t)L - A.n_neg_big;
    if (ld > L) ld = L;
    if (ld > cnt) ld = cnt;
    if (ld < 0) ld = 0;
    lead = (int)ld;
    // positions in the tail list, in a fresh random order (at least one: e2 = the first tail of that order, data.py:264)
    smp_distinct(dst, lead > 0 ? lead : (cnt > 0 ? 1 : 0), cnt, work, A, k0 ^ 0x5bd1e995u, k1, scan);
    e2v = cnt > 0 ? A.tails[lo + dst[0]] : -1;
    for (int i = tid; i < lead; i += 256) lkv[i] = (int)A.tails[lo + dst[i]];
    __syncthreads();
  }
  const int nn = L - lead;
  smp_distinct(dst, nn, A.E, work, A, k0, k1 ^ 0x2545F491u, scan);
  for (int i = tid; i < nn; i += 256) lkv[lead + i] = dst[i];
  __syncthreads();
  int32_t* lk = A.lookup + row * L;
  float* lab = A.labels + row * L;
  for (int j = tid; j < L; j += 256) {
    const int v = lkv[j];
    bool in = false;
    if (use_set) {
      uint32_t s = smp_slot(v, A.ts_mask);
      for (;;) {
        const int w = tset[s];
        if (w == -1) break;
        if (w == v) { in = true; break; }
        s = (s + 1) & (uint32_t)A.ts_mask;
      }
    } else {
      for (int64_t i = 0; i < cnt; ++i) in |= A.tails[lo + i] == v;
    }
    lk[j] = v;
    lab[j] = in ? 1.f : 0.f;
  }
  if (tid == 0) {
    A.e1[row] = A.rec_e1[r];
    A.rel[row] = A.rec_rel[r];
    A.e2[row] = e2v;
  }
}

int smp_pow2_at_least(int64_t n) {
  int p = 1;
  while (p < n && p < (1 << 30)) p <<= 1;
  return p;
}

}  // namespace
}  // namespace coper

using namespace coper;

extern "C" COPER_API int coper_sample_train_batch(int32_t device, const int64_t* rec, const int64_t* pos_tail, const int64_t* rec_e1,
                                                  const int64_t* rec_rel, const int64_t* tail_indptr, const int64_t* tail_idx, int64_t B, int64_t L,
                                                  int64_t num_ent, int32_t proportional, double prop_negatives, int64_t max_tails, uint64_t seed,
                                                  uint64_t batch, int64_t* e1, int64_t* rel, int64_t* e2, int32_t* lookup, float* labels,
                                                  void* stream) {
  if (!rec || !rec_e1 || !rec_rel || !tail_indptr || !tail_idx || !e1 || !rel || !e2 || !lookup || !labels) return COPER_EINVAL;
  if (!proportional && !pos_tail) return COPER_EINVAL;
  if (B <= 0 || L <= 0 || num_ent <= 0 || L > num_ent || max_tails < 0 || !(prop_negatives >= 0.0)) return COPER_EINVAL;
  if (L > 2048 || num_ent > 0x7fffffffLL) return COPER_EUNSUPPORTED;      // (the LDS plan below; entity ids as int32: the lookup's dtype)
  if (hipSetDevice(device) != hipSuccess) return COPER_EHIP;
  SampleArgs A;
  A.rec = rec; A.pos_tail = pos_tail; A.rec_e1 = rec_e1; A.rec_rel = rec_rel; A.ip = tail_indptr; A.tails = tail_idx;
  A.L = L; A.E = num_ent; A.proportional = proportional ? 1 : 0;
  A.need = (int)(1.0 / (1.0 + prop_negatives) * (double)L);      // data.py:243 (the same expression, float64)
  A.n_neg_big = (int)(num_ent < L - A.need ? num_ent : L - A.need);
  // tail set: twice the longest tail list of the batch, at most 16,384 slots (longer lists are scanned)
  int ts = max_tails > 0 ? smp_pow2_at_least(2 * max_tails) : 0;
  if (ts > 16384) ts = 16384;
  A.ts_mask = ts > 0 ? (ts < 64 ? 64 : ts) - 1 : 0;
  // dedup table: the largest sample is L of num_ent (or of a tail list): 2.5 x (L + its expected repeats + 32) slots
  const double rep = 2.0 * (double)L * (double)L / (double)(num_ent >= 4 * L ? num_ent : 4 * L);
  const int H = smp_pow2_at_least((int64_t)(2.5 * ((double)L + rep + 64.0)));
  A.h_mask = H - 1;
  A.work_ints = 2 * H > 4 * (int)L ? 2 * H : 4 * (int)L;
  A.seed0 = (uint32_t)seed; A.seed1 = (uint32_t)(seed >> 32) ^ 0x9E3779B9u; A.batch = (uint32_t)batch ^ (uint32_t)(batch >> 32) * 0x85EBCA77u;
  A.e1 = e1; A.rel = rel; A.e2 = e2; A.lookup = lookup; A.labels = labels;
  const size_t lds = sizeof(int) * ((size_t)(A.ts_mask ? A.ts_mask + 1 : 0) + 2 * (size_t)L + 256 + (size_t)A.work_ints);
  if (lds > 160 * 1024) return COPER_EUNSUPPORTED;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k_sample_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return COPER_EHIP;
  hipLaunchKernelGGL(k_sample_rows, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, A);
  return hipGetLastError() == hipSuccess ? COPER_OK : COPER_EHIP;
}
