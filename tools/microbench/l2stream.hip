// What rate does the score kernel's entity-fragment stream get out of L2 / Infinity Cache, and does it depend on
// which part of the table the workgroups of one XCD walk?
// 256 persistent workgroups x 8 waves; per unit a wave reads ME blocks x 2 planes x KS k-steps of 1 KiB, PD k-steps ahead
// (the access pattern of k_score_count_bf16x3), from a table of n_blk blocks.
//   mode 0: workgroup i starts at unit i * units_per_wg of a (tile-major) walk over the whole table (the kernel today)
//   mode 1: workgroup i walks only the 1/8 slice of the table owned by XCD (i % 8)
//   mode 2: every workgroup walks the same blocks (best case for L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ME, int PD>
__global__ __launch_bounds__(512) void k_l2stream(const u32x4* __restrict__ hi, const u32x4* __restrict__ lo, int n_blk, int KS,
                                                  int units_per_wg, int mode, unsigned* __restrict__ out, unsigned* __restrict__ xcc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int iters = n_blk / (8 * ME);            // units per sweep of the table
  const int slice_iters = iters / 8;             // units per sweep of one XCD's slice
  u32x4 acc = {0, 0, 0, 0};
  if (threadIdx.x == 0 && xcc) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
  }
  constexpr int NBF = PD + 1;
  u32x4 ah[NBF][ME], al[NBF][ME];
  for (int u = 0; u < units_per_wg; ++u) {
    long it;
    if (mode == 0) it = ((long)blockIdx.x * units_per_wg + u) % iters;
    else if (mode == 1) it = (long)(blockIdx.x & 7) * slice_iters + ((blockIdx.x >> 3) * 3 + u) % slice_iters;
    else it = u % iters;
    const long eb = (it * 8 + wave) * ME;
#pragma unroll
    for (int i = 0; i < PD; ++i)
#pragma unroll
      for (int m = 0; m < ME; ++m) {
        long o = ((eb + m) * KS + (i < KS ? i : KS - 1)) * 64 + lane;
        ah[i][m] = hi[o]; al[i][m] = lo[o];
      }
    int ks = 0;
    for (; ks + NBF <= KS; ks += NBF) {
#pragma unroll
      for (int j = 0; j < NBF; ++j) {
        const int kn = ks + j + PD < KS ? ks + j + PD : KS - 1;
#pragma unroll
        for (int m = 0; m < ME; ++m) {
          long o = ((eb + m) * KS + kn) * 64 + lane;
          ah[(j + PD) % NBF][m] = hi[o]; al[(j + PD) % NBF][m] = lo[o];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < ME; ++m) acc ^= ah[j][m] ^ al[j][m];
      }
    }
#pragma unroll
    for (int j = 0; j < NBF - 1; ++j)
      if (ks + j < KS) {
#pragma unroll
        for (int m = 0; m < ME; ++m) acc ^= ah[j][m] ^ al[j][m];
      }
  }
  unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u) out[0] = r;
}

template <int ME, int PD>
void run(const u32x4* hi, const u32x4* lo, int n_blk, int KS, int units_per_wg, int mode, unsigned* out, unsigned* xcc, const char* tag) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_l2stream<ME, PD>), dim3(256), dim3(512), 0, 0, hi, lo, n_blk, KS, units_per_wg, mode, out, xcc);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_l2stream<ME, PD>), dim3(256), dim3(512), 0, 0, hi, lo, n_blk, KS, units_per_wg, mode, out, xcc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  double bytes = 256.0 * units_per_wg * 8 * ME * 2 * KS * 1024;
  printf("%-34s ME=%d PD=%d n_blk=%6d (%.1f MB) mode=%d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", tag, ME, PD, n_blk,
         2.0 * n_blk * KS * 1024 / 1e6, mode, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3);
}

int main() {
  const int KS = 13;
  const int big_blk = 320000;   // 8.5 GB: HBM
  size_t bytes = (size_t)big_blk * KS * 1024;
  u32x4 *hi, *lo; unsigned *out, *xcc;
  hipMalloc(&hi, bytes); hipMalloc(&lo, bytes); hipMalloc(&out, 4); hipMalloc(&xcc, 256 * 4);
  hipMemset(hi, 1, bytes); hipMemset(lo, 2, bytes);
  unsigned hx[256];
  // FB15k-237-sized table: 456 blocks -> padded to 512 for the slicing arithmetic (13.6 MB both planes)
  for (int mode = 0; mode < 3; ++mode) {
    run<2, 3>(hi, lo, 512, KS, 18, mode, out, xcc, "fb15k-sized table");
    run<2, 6>(hi, lo, 512, KS, 18, mode, out, xcc, "fb15k-sized table, PD 6");
    run<1, 6>(hi, lo, 512, KS, 36, mode, out, xcc, "fb15k-sized table, ME 1 PD 6");
  }
  hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost);
  int agree = 0;
  for (int i = 0; i < 256; ++i) agree += (hx[i] == hx[i & 7]);
  printf("XCC_ID of block i equals XCC_ID of block i %% 8 for %d of 256 blocks; ids of blocks 0..7:", agree);
  for (int i = 0; i < 8; ++i) printf(" %u", hx[i]);
  printf("\n");
  // WN18RR-sized (1280 blocks = 34 MB), and a table in the Infinity Cache only (4096 blocks = 109 MB)
  for (int mode = 0; mode < 2; ++mode) {
    run<2, 3>(hi, lo, 1280, KS, 20, mode, out, xcc, "wn18rr-sized table");
    run<2, 3>(hi, lo, 4096, KS, 16, mode, out, xcc, "109 MB table");
  }
  // HBM: every block read once
  run<2, 3>(hi, lo, big_blk, KS, big_blk / 16 / 256, 0, out, nullptr, "HBM sweep");
  run<2, 6>(hi, lo, big_blk, KS, big_blk / 16 / 256, 0, out, nullptr, "HBM sweep PD 6");
  run<4, 3>(hi, lo, big_blk, KS, big_blk / 32 / 256, 0, out, nullptr, "HBM sweep ME 4");
  return 0;
}
