// sustained int8 MFMA rate (v_mfma_i32_32x32x32_i8) on random data, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_rate(const i32x4v* __restrict__ in, int* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  i32x4v a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = in[(blockIdx.x * 4 + i) * 64 + lane]; b[i] = in[(blockIdx.x * 4 + i + 2000) * 64 + lane]; }
  i32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m & 3], b[(m >> 2) & 3], acc[m & 3], 0, 0, 0);
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int THREADS>
void run(const i32x4v* in, int* out, const char* tag) {
  int iters = 4000, nwg = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k_rate<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_rate<THREADS>), dim3(nwg), dim3(THREADS), 100 * 1024, 0, in, out, 200);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_rate<THREADS>), dim3(nwg), dim3(THREADS), 100 * 1024, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)nwg * (THREADS / 64) * iters * 32.0 * (2.0 * 32 * 32 * 32);
  printf("%-10s i8 32x32x32 waves/SIMD=%d: %.3f ms  %.0f TOP/s\n", tag, THREADS / 256, ms, ops / ms / 1e9);
}
int main() {
  i32x4v* in; int* out;
  size_t n = 8192 * 64;
  hipMalloc(&in, n * 16); hipMalloc(&out, 256 * 512 * 4);
  unsigned* h = (unsigned*)malloc(n * 16);
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t i = 0; i < n * 4; ++i) h[i] = pass ? (unsigned)rand() * 2654435761u : 0u;
    hipMemcpy(in, h, n * 16, hipMemcpyHostToDevice);
    run<256>(in, out, pass ? "random" : "zeros"); run<512>(in, out, pass ? "random" : "zeros");
  }
  return 0;
}
