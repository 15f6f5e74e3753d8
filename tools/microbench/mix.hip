// MFMA + VALU mix microbenchmark: does co-issuing conv-like VALU with 16x16x32 bf16 MFMAs cost more than the issue model?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int VPM, bool DO_MFMA>
__global__ __launch_bounds__(256) void k_mix(const u32x4* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  extern __shared__ unsigned dummy[];
  const int lane = threadIdx.x & 63;
  u32x4 a[4], b[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = in[(blockIdx.x * 4 + i) * 64 + lane];
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = in[(blockIdx.x * 6 + i + 1000) * 64 + lane];
  f32x4 acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float v[8], t[9][8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { v[c] = __uint_as_float(a[c & 3][c >> 2]) * 1e-3f;
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k][c] = __uint_as_float(b[k % 6][c & 3]) * 1e-3f; }
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (DO_MFMA) {
#pragma unroll
      for (int m = 0; m < 54; ++m)
        acc[m % 18] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&a[m & 3], *(bf16x8*)&b[m % 6], acc[m % 18], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < (VPM * 54) / 8; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = fmaf(v[(c + 1) & 7], t[j % 9][c], v[c]);
    if (DO_MFMA && VPM > 0) {
#pragma unroll
      for (int m = 0; m < 54; ++m) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 18; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int VPM, bool DO_MFMA>
void run(const u32x4* in, float* out, unsigned long long* clk, const char* tag) {
  int iters = 2000, nwg = 256;
  hipFuncSetAttribute((const void*)k_mix<VPM, DO_MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_mix<VPM, DO_MFMA>), dim3(nwg), dim3(256), 100 * 1024, 0, in, out, clk, 100);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_mix<VPM, DO_MFMA>), dim3(nwg), dim3(256), 100 * 1024, 0, in, out, clk, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double ns_per_it = ms * 1e6 / iters;
  printf("%-14s VPM=%d mfma=%d: %.1f ns/iter  cyc/iter(readcyclecounter)=%.0f  wall ticks/iter=%.1f\n", tag, VPM, (int)DO_MFMA, ns_per_it, (double)h[0] / iters, (double)h[1] / iters);
}
int main() {
  u32x4* in; float* out; unsigned long long* clk;
  size_t n = 4096 * 64;
  hipMalloc(&in, n * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 16);
  unsigned* h = (unsigned*)malloc(n * 16);
  for (size_t i = 0; i < n * 4; ++i) { unsigned r = rand(); h[i] = ((r & 0xffff) | 0x3f800000u) ^ ((r >> 3) << 16 & 0x007f0000); h[i] = 0x3f803f80u ^ (r & 0x007f007fu); }
  hipMemcpy(in, h, n * 16, hipMemcpyHostToDevice);
  run<0, true>(in, out, clk, "mfma only");
  run<2, true>(in, out, clk, "mix");
  run<4, true>(in, out, clk, "mix");
  run<5, true>(in, out, clk, "mix");
  run<6, true>(in, out, clk, "mix");
  run<4, false>(in, out, clk, "valu only");
  run<5, false>(in, out, clk, "valu only");
  return 0;
}
