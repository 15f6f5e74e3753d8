// streaming microbenchmark: W waves per WG, each wave walks S streams of L 1-KiB chunks, PF chunks ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int S, int PF, bool NT>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ src, int L, unsigned* __restrict__ out, int lds_dummy) {
  extern __shared__ unsigned dummy[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const u32x4* p[S];
#pragma unroll
  for (int s = 0; s < S; ++s) p[s] = src + ((size_t)((size_t)blockIdx.x * nw + wave) * S + s) * L * 64 + lane;
  u32x4 buf[PF][S];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < PF; ++t)
#pragma unroll
    for (int s = 0; s < S; ++s) buf[t][s] = NT ? __builtin_nontemporal_load(p[s] + (size_t)(t < L ? t : L - 1) * 64) : p[s][(size_t)(t < L ? t : L - 1) * 64];
  for (int k0 = 0; k0 < L; k0 += PF) {
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      int k = k0 + t;
      if (k < L) {
#pragma unroll
        for (int s = 0; s < S; ++s) acc ^= buf[t][s];
        if (k + PF < L) {
#pragma unroll
          for (int s = 0; s < S; ++s) buf[t][s] = NT ? __builtin_nontemporal_load(p[s] + (size_t)(k + PF) * 64) : p[s][(size_t)(k + PF) * 64];
        }
      }
    }
  }
  if (lds_dummy < 0) dummy[threadIdx.x] = acc[0];
  unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u) out[0] = r;
}
template <int S, int PF, bool NT>
void run(const u32x4* d, size_t bytes, int L, int threads, int lds, unsigned* out, const char* tag) {
  int nw = threads / 64;
  size_t per_wg = (size_t)nw * S * L * 1024;
  int nwg = (int)(bytes / per_wg);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k_stream<S, PF, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_stream<S, PF, NT>), dim3(nwg), dim3(threads), lds, 0, d, L, out, 0);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<S, PF, NT>), dim3(nwg), dim3(threads), lds, 0, d, L, out, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("%-28s S=%d PF=%d NT=%d L=%3d thr=%d lds=%6d nwg=%5d: %.3f ms  %.2f TB/s\n", tag, S, PF, (int)NT, L, threads, lds, nwg, ms, (double)nwg * per_wg / ms / 1e9);
}
int main() {
  size_t bytes = (size_t)1400 << 20;
  u32x4* d; unsigned* out;
  hipMalloc(&d, bytes); hipMalloc(&out, 4);
  hipMemset(d, 1, bytes);
  // like the dense kernel: 4 waves, 8 streams each (~26 streams/WG), L = 36, one WG per CU (LDS 132 KB)
  run<8, 4, true>(d, bytes, 36, 256, 132 * 1024, out, "dense-like 1WG/CU nt");
  run<8, 4, false>(d, bytes, 36, 256, 132 * 1024, out, "dense-like 1WG/CU");
  run<8, 4, true>(d, bytes, 36, 256, 64 * 1024, out, "2WG/CU nt");
  run<8, 2, true>(d, bytes, 36, 256, 32 * 1024, out, "4WG/CU nt PF2");
  run<8, 4, true>(d, bytes, 36, 256, 32 * 1024, out, "4WG/CU nt PF4");
  run<8, 4, true>(d, bytes, 144, 256, 132 * 1024, out, "1WG/CU nt long");
  run<8, 6, true>(d, bytes, 36, 256, 132 * 1024, out, "1WG/CU nt PF6");
  run<4, 8, true>(d, bytes, 36, 256, 132 * 1024, out, "1WG/CU nt S4 PF8");
  run<2, 8, true>(d, bytes, 144, 256, 132 * 1024, out, "1WG/CU nt S2 PF8 long");
  run<1, 16, true>(d, bytes, 288, 256, 132 * 1024, out, "1WG/CU nt S1 PF16 long");
  run<1, 8, true>(d, bytes, 288, 256, 16 * 1024, out, "8WG/CU nt S1 PF8 long");
  run<8, 4, true>(d, bytes, 36, 512, 132 * 1024, out, "1WG/CU 8 waves nt");
  run<8, 4, true>(d, bytes, 36, 1024, 132 * 1024, out, "1WG/CU 16 waves nt");
  run<4, 4, true>(d, bytes, 36, 1024, 132 * 1024, out, "1WG/CU 16 waves S4 nt");
  return 0;
}
