// Where should the count kernel's loads sit among its MFMAs?  (MI355X, one wave per SIMD.)  A "region" as in
// k_score_count3_bf16x3: 12 v_mfma_f32_16x16x32_f16 on 4 accumulator chains, 15 epilogue vector instructions two behind each
// MFMA, and per region ONE global_load_dwordx4 (cache resident) and TWO ds_read_b128 whose results nobody waits for beyond a
// bounded s_waitcnt.  One asm block = 8 regions; patterns:
//   0 no loads | 1 G r r + wait in front of the region (the shipped form) | 2 one load behind each of MFMAs 9, 10, 11
//   3 loads behind MFMAs 8, 9, 10 | 4 only G in front | 5 only r r in front | 6 as 1 without the epilogue | 7 as 0 without the epilogue
//   8 as 2 with the wait behind MFMA 12 instead of in front
// hipcc -O3 --offload-arch=gfx950 mfma_load_mix.hip -o mfma_load_mix && ./mfma_load_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(c) "v_mfma_f32_16x16x32_f16 a[" #c ":" #c "+3], v[8:11], v[12:15], a[" #c ":" #c "+3]\n\t"
#define E2a "v_accvgpr_read_b32 v20, a64\n\tv_cmp_gt_f32 vcc, v20, v21\n\t"
#define E2b "v_addc_co_u32 v22, vcc, v22, v22, vcc\n\tv_cmp_ge_f32 vcc, v20, v23\n\t"
#define E2c "v_addc_co_u32 v24, vcc, v24, v24, vcc\n\tv_accvgpr_read_b32 v20, a64\n\t"
#define E2d "v_cmp_gt_f32 vcc, v20, v21\n\tv_addc_co_u32 v22, vcc, v22, v22, vcc\n\t"
#define E2e "v_cmp_ge_f32 vcc, v20, v23\n\tv_addc_co_u32 v24, vcc, v24, v24, vcc\n\t"
#define E1  "v_accvgpr_read_b32 v20, a64\n\t"
#define G(o) "global_load_dwordx4 v[32:35], v[2:3], off offset:" #o "\n\t"
#define R0(o) "ds_read_b128 v[36:39], v4 offset:" #o "\n\t"
#define R1(o) "ds_read_b128 v[40:43], v4 offset:" #o "\n\t"
#define W "s_waitcnt vmcnt(8) lgkmcnt(4)\n\t"
#define NONE ""

// a region: FRONT, then 12 MFMAs; epilogue E* behind MFMAs 1..8 (15 instructions); L9 / L10 / L11 / L8 / L12 behind those MFMAs
#define REGION(FRONT, EA, EB, EC, ED, EE, EF, L8, L9, L10, L11, L12) \
  FRONT MF(0) EA MF(4) EB MF(8) EC MF(12) ED MF(0) EE MF(4) EA MF(8) EB MF(12) EF L8 MF(0) L9 MF(4) L10 MF(8) L11 MF(12) L12

#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v20","v21","v22","v23","v24","vcc","memory", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a64"

#define X8(R) R R R R R R R R

template <int P>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, const uint4* src, int iters) {
  __shared__ uint4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(i, 0, 0, 0);
  __syncthreads();
  const uint4* p = src + (blockIdx.x * 256 + threadIdx.x);
  const unsigned l = (unsigned)(threadIdx.x & 63) * 16u;
  asm volatile("v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\t"
               "v_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v23, 0.5\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v24, 0\n\t"
               "v_accvgpr_write_b32 a64, 0" ::: CLOB);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned plo = (unsigned)(unsigned long long)p, phi = (unsigned)((unsigned long long)p >> 32);
#define BODY(...) asm volatile("v_mov_b32 v2, %0\n\tv_mov_b32 v3, %1\n\tv_mov_b32 v4, %2\n\t" X8(REGION(__VA_ARGS__)) : : "v"(plo), "v"(phi), "v"(l) : "v2", "v3", "v4", CLOB)
    if constexpr (P == 0) BODY(NONE, E2a, E2b, E2c, E2d, E2e, E1, NONE, NONE, NONE, NONE, NONE);
    if constexpr (P == 1) BODY(G(0) R0(0) R1(1024) W, E2a, E2b, E2c, E2d, E2e, E1, NONE, NONE, NONE, NONE, NONE);
    if constexpr (P == 2) BODY(W, E2a, E2b, E2c, E2d, E2e, E1, NONE, G(0), R0(0), R1(1024), NONE);
    if constexpr (P == 3) BODY(W, E2a, E2b, E2c, E2d, E2e, E1, G(0), R0(0), R1(1024), NONE, NONE);
    if constexpr (P == 4) BODY(G(0) W, E2a, E2b, E2c, E2d, E2e, E1, NONE, NONE, NONE, NONE, NONE);
    if constexpr (P == 5) BODY(R0(0) R1(1024) W, E2a, E2b, E2c, E2d, E2e, E1, NONE, NONE, NONE, NONE, NONE);
    if constexpr (P == 6) BODY(G(0) R0(0) R1(1024) W, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE);
    if constexpr (P == 7) BODY(NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE);
    if constexpr (P == 8) BODY(NONE, E2a, E2b, E2c, E2d, E2e, E1, NONE, G(0), R0(0), R1(1024), W);
    if constexpr (P == 9) BODY(NONE, NONE, NONE, NONE, NONE, NONE, NONE, NONE, G(0), R0(0), R1(1024), W);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (lds[iters & 4095].x == 0xFFFFFFFFu);
}

template <int P>
static double run(int iters, unsigned long long* d, const uint4* src) {
  hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, d, src, 16);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, d, src, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 8);
  uint4* src; hipMalloc(&src, 256 * 256 * 16 + 8192); hipMemset(src, 0, 256 * 256 * 16 + 8192);
  const int iters = 4000;
  const char* names[] = {"no loads", "G r r wait in FRONT of the region (shipped)", "loads behind MFMAs 9, 10, 11; wait in front", "loads behind MFMAs 8, 9, 10; wait in front",
                         "only G in front", "only r r in front", "G r r in front, no epilogue", "bare: no loads, no epilogue", "loads behind 9, 10, 11; wait behind 12",
                         "loads behind 9, 10, 11, wait behind 12, no epilogue"};
  double ms[10];
  ms[0] = run<0>(iters, d, src); ms[1] = run<1>(iters, d, src); ms[2] = run<2>(iters, d, src); ms[3] = run<3>(iters, d, src); ms[4] = run<4>(iters, d, src);
  ms[5] = run<5>(iters, d, src); ms[6] = run<6>(iters, d, src); ms[7] = run<7>(iters, d, src); ms[8] = run<8>(iters, d, src); ms[9] = run<9>(iters, d, src);
  for (int p = 0; p < 10; ++p)
    printf("%-56s %8.3f ms   %.3f x bare   (%.2f ns per MFMA)\n", names[p], ms[p], ms[p] / ms[7], ms[p] * 1e6 / (96.0 * iters));
  return 0;
}
