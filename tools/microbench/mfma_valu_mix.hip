// How do vector instructions placed between v_mfma_f32_16x16x32_f16 share the issue with them?  (MI355X, one wave per SIMD --
// the count kernel's situation.)  Every pattern is ONE asm block the compiler cannot rearrange: 16 MFMAs on 16 independent
// accumulators (AGPRs) with vector instructions (v_accvgpr_read + compare + add-with-carry, the epilogue's kinds) placed
//   none | 1 behind every MFMA | 2 behind every MFMA | 5 behind every 5th | 10 behind every 5th | 3 behind every MFMA
// The kernel loops the block; s_memtime brackets the loop.  hipcc -O3 --offload-arch=gfx950 mfma_valu_mix.hip -o mix && ./mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define M(i) "v_mfma_f32_16x16x32_f16 a[" #i ":" #i "+3], v[8:11], v[12:15], a[" #i ":" #i "+3]\n\t"
#define V1 "v_accvgpr_read_b32 v20, a64\n\t"
#define V2 "v_accvgpr_read_b32 v20, a64\n\tv_cmp_gt_f32 vcc, v20, v21\n\t"
#define V3 "v_accvgpr_read_b32 v20, a64\n\tv_cmp_gt_f32 vcc, v20, v21\n\tv_addc_co_u32 v22, vcc, v22, v22, vcc\n\t"
#define V5 V3 "v_cmp_ge_f32 vcc, v20, v23\n\tv_addc_co_u32 v24, vcc, v24, v24, vcc\n\t"
#define V10 V5 V5
// plain fp32 vector ops instead of the epilogue's kinds
#define F1 "v_add_f32 v25, v25, v21\n\t"
#define F2 F1 "v_add_f32 v26, v26, v21\n\t"
#define S1 "s_nop 0\n\t"

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v20","v21","v22","v23","v24","v25","v26","vcc", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23", \
  "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
  "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64"

template <int P>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, int iters) {
  asm volatile("v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\t"
               "v_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v23, 0.5\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v24, 0\n\t"
               "v_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_accvgpr_write_b32 a64, 0" ::: CLOB);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (P == 0) asm volatile(M(0) M(4) M(8) M(12) M(16) M(20) M(24) M(28) M(32) M(36) M(40) M(44) M(48) M(52) M(56) M(60) ::: CLOB);
    if constexpr (P == 1) asm volatile(M(0) V1 M(4) V1 M(8) V1 M(12) V1 M(16) V1 M(20) V1 M(24) V1 M(28) V1 M(32) V1 M(36) V1 M(40) V1 M(44) V1 M(48) V1 M(52) V1 M(56) V1 M(60) V1 ::: CLOB);
    if constexpr (P == 2) asm volatile(M(0) V2 M(4) V2 M(8) V2 M(12) V2 M(16) V2 M(20) V2 M(24) V2 M(28) V2 M(32) V2 M(36) V2 M(40) V2 M(44) V2 M(48) V2 M(52) V2 M(56) V2 M(60) V2 ::: CLOB);
    if constexpr (P == 3) asm volatile(M(0) V3 M(4) V3 M(8) V3 M(12) V3 M(16) V3 M(20) V3 M(24) V3 M(28) V3 M(32) V3 M(36) V3 M(40) V3 M(44) V3 M(48) V3 M(52) V3 M(56) V3 M(60) V3 ::: CLOB);
    // 16 MFMAs, 15 vector instructions in three clumps of 5 (the shipped kernel's shape: about one per MFMA)
    if constexpr (P == 4) asm volatile(M(0) M(4) M(8) M(12) M(16) V5 M(20) M(24) M(28) M(32) M(36) V5 M(40) M(44) M(48) M(52) M(56) V5 M(60) ::: CLOB);
    // the same 15 spread: 1 behind 15 of the 16
    if constexpr (P == 5) asm volatile(M(0) V1 M(4) V1 M(8) V1 M(12) V1 M(16) V1 M(20) V1 M(24) V1 M(28) V1 M(32) V1 M(36) V1 M(40) V1 M(44) V1 M(48) V1 M(52) V1 M(56) V1 M(60) ::: CLOB);
    // 30 in three clumps of 10 / spread 2 behind 15
    if constexpr (P == 6) asm volatile(M(0) M(4) M(8) M(12) M(16) V10 M(20) M(24) M(28) M(32) M(36) V10 M(40) M(44) M(48) M(52) M(56) V10 M(60) ::: CLOB);
    if constexpr (P == 7) asm volatile(M(0) V2 M(4) V2 M(8) V2 M(12) V2 M(16) V2 M(20) V2 M(24) V2 M(28) V2 M(32) V2 M(36) V2 M(40) V2 M(44) V2 M(48) V2 M(52) V2 M(56) V2 M(60) ::: CLOB);
    // plain fp32 adds: 1 and 2 behind every MFMA
    if constexpr (P == 8) asm volatile(M(0) F1 M(4) F1 M(8) F1 M(12) F1 M(16) F1 M(20) F1 M(24) F1 M(28) F1 M(32) F1 M(36) F1 M(40) F1 M(44) F1 M(48) F1 M(52) F1 M(56) F1 M(60) F1 ::: CLOB);
    if constexpr (P == 9) asm volatile(M(0) F2 M(4) F2 M(8) F2 M(12) F2 M(16) F2 M(20) F2 M(24) F2 M(28) F2 M(32) F2 M(36) F2 M(40) F2 M(44) F2 M(48) F2 M(52) F2 M(56) F2 M(60) F2 ::: CLOB);
    // a scalar no-op behind every MFMA; two behind every MFMA
    if constexpr (P == 10) asm volatile(M(0) S1 M(4) S1 M(8) S1 M(12) S1 M(16) S1 M(20) S1 M(24) S1 M(28) S1 M(32) S1 M(36) S1 M(40) S1 M(44) S1 M(48) S1 M(52) S1 M(56) S1 M(60) S1 ::: CLOB);
    if constexpr (P == 11) asm volatile(M(0) S1 S1 M(4) S1 S1 M(8) S1 S1 M(12) S1 S1 M(16) S1 S1 M(20) S1 S1 M(24) S1 S1 M(28) S1 S1 M(32) S1 S1 M(36) S1 S1 M(40) S1 S1 M(44) S1 S1 M(48) S1 S1 M(52) S1 S1 M(56) S1 S1 M(60) S1 S1 ::: CLOB);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int P>
static double run(int iters, unsigned long long* d) {
  hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, d, 16);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 8);
  const int iters = 20000;
  const char* names[] = {"16 MFMAs, nothing between", "1 (accvgpr_read) behind every MFMA", "2 (read, cmp) behind every MFMA", "3 (read, cmp, addc) behind every MFMA",
                         "15 in 3 clumps of 5", "15 spread, 1 behind each", "30 in 3 clumps of 10", "30 spread, 2 behind each",
                         "1 v_add_f32 behind every MFMA", "2 v_add_f32 behind every MFMA", "1 s_nop behind every MFMA", "2 s_nop behind every MFMA"};
  double ms[12];
  ms[0] = run<0>(iters, d); ms[1] = run<1>(iters, d); ms[2] = run<2>(iters, d); ms[3] = run<3>(iters, d); ms[4] = run<4>(iters, d); ms[5] = run<5>(iters, d);
  ms[6] = run<6>(iters, d); ms[7] = run<7>(iters, d); ms[8] = run<8>(iters, d); ms[9] = run<9>(iters, d); ms[10] = run<10>(iters, d); ms[11] = run<11>(iters, d);
  for (int p = 0; p < 12; ++p)
    printf("%-42s %8.3f ms   %.2f x the bare MFMAs   (%.1f ns per MFMA)\n", names[p], ms[p], ms[p] / ms[0], ms[p] * 1e6 / (16.0 * iters));
  return 0;
}
