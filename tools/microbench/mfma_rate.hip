// sustained bf16 MFMA rate on random data: 16x16x32 vs 32x32x16, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rate(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = in[(blockIdx.x * 4 + i) * 64 + lane]; b[i] = in[(blockIdx.x * 4 + i + 2000) * 64 + lane]; }
  float s = 0;
  if (SHAPE == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 64; ++m)
        acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&a[m & 3], *(bf16x8*)&b[(m >> 2) & 3], acc[m & 15], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 32; ++m)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8*)&a[m & 3], *(bf16x8*)&b[(m >> 2) & 3], acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int SHAPE, int THREADS>
void run(const u32x4* in, float* out, const char* tag) {
  int iters = 4000, nwg = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k_rate<SHAPE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_rate<SHAPE, THREADS>), dim3(nwg), dim3(THREADS), 100 * 1024, 0, in, out, 200);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_rate<SHAPE, THREADS>), dim3(nwg), dim3(THREADS), 100 * 1024, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flop = (double)nwg * (THREADS / 64) * iters * 64.0 * 16384.0;   // both shapes: 64 x 16384 flop per iteration per wave
  printf("%-28s shape=%d waves/SIMD=%d: %.3f ms  %.0f TFLOP/s\n", tag, SHAPE, THREADS / 256, ms, flop / ms / 1e9);
}
// long mode (tools/power_probe.sh): `mfma_rate <seconds> [zeros]` keeps 32x32x16, one wave per SIMD, running for that long
static void run_long(const u32x4* in, float* out, double seconds, const char* tag) {
  const int iters = 4000, nwg = 256;
  hipFuncSetAttribute((const void*)k_rate<32, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  double total_ms = 0; long launches = 0;
  while (total_ms < seconds * 1e3) {
    hipEventRecord(a);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_rate<32, 256>), dim3(nwg), dim3(256), 100 * 1024, 0, in, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    total_ms += ms; launches += 100;
  }
  double flop = (double)nwg * 4 * iters * 64.0 * 16384.0 * launches;
  printf("%-8s shape=32 waves/SIMD=1, %.1f s: %.0f TFLOP/s\n", tag, total_ms / 1e3, flop / total_ms / 1e9);
}
int main(int argc, char** argv) {
  u32x4* in; float* out;
  size_t n = 8192 * 64;
  hipMalloc(&in, n * 16); hipMalloc(&out, 256 * 512 * 4);
  unsigned* h = (unsigned*)malloc(n * 16);
  if (argc > 1) {
    const bool zeros = argc > 2;
    for (size_t i = 0; i < n * 4; ++i) { unsigned r = rand(); h[i] = zeros ? 0u : (0x3f803f80u ^ (r & 0x007f007fu) ^ ((r >> 8) & 0x80008000u)); }
    hipMemcpy(in, h, n * 16, hipMemcpyHostToDevice);
    run_long(in, out, atof(argv[1]), zeros ? "zeros" : "random");
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t i = 0; i < n * 4; ++i) { unsigned r = rand(); h[i] = pass ? (0x3f803f80u ^ (r & 0x007f007fu) ^ ((r >> 8) & 0x80008000u)) : 0u; }
    hipMemcpy(in, h, n * 16, hipMemcpyHostToDevice);
    const char* tag = pass ? "random" : "zeros";
    run<16, 256>(in, out, tag); run<16, 512>(in, out, tag); run<32, 256>(in, out, tag); run<32, 512>(in, out, tag);
  }
  return 0;
}
