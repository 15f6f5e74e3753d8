// Three questions behind round 3's count-kernel decisions (MI355X; hipcc -O3 --offload-arch=gfx950):
//  1. bits:   is ONE v_mfma_f32_16x16x32_bf16 the same function of (A, B, C) as TWO chained v_mfma_f32_32x32x16_bf16 over the
//             same 32 k?  (If not, every bf16x3 kernel has to change shape together to keep counts and pair logits consistent.)
//  2. f16:    does the f16 MFMA keep subnormal inputs (a two-term fp16 split has 22 bits when the low term may be subnormal)?
//  3. rate:   the count kernel's inner loop -- query fragments re-read from LDS by ds_read_b128, 3 MFMAs per product, 128
//             accumulator registers, one wave per SIMD, random data -- as 32x32x16 (the shipped tiling: 8 reads per 12
//             MFMAs), as 32x32x16 with both entity blocks per k-step (8 reads per 24), and as 16x16x32 (64 entities x 128
//             queries per wave: 16 reads per 96 MFMAs).  MI355X_MICROARCH.md reports 1.12-1.15x for 16x16x32 under the
//             power limit; tools/microbench/mfma_rate.hip (registers only) measured the opposite on this pool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static inline float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- 1. bits ----------------------------------------------------------------------------------------------------------
// A [32 rows][32 k], B [32 k][32 cols] bf16 row-major in global memory; C = bias[row].
__global__ void k_bits(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, const float* __restrict__ bias,
                       float* __restrict__ out32, float* __restrict__ out16) {
  const int l = threadIdx.x;
  // 32x32x16: lane l: A[row l&31][k 8(l>>5)..+7], B[k 8(l>>5)..+7][col l&31]; acc[r]: row 8(r>>2) + 4(l>>5) + (r&3), col l&31
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = bias[8 * (r >> 2) + 4 * (l >> 5) + (r & 3)];
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8 a, b;
    unsigned short ta[8], tb[8];
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * kk + 8 * (l >> 5) + j;
      ta[j] = A[(l & 31) * 32 + k];
      tb[j] = B[k * 32 + (l & 31)];
    }
    memcpy(&a, ta, 16); memcpy(&b, tb, 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) out32[(8 * (r >> 2) + 4 * (l >> 5) + (r & 3)) * 32 + (l & 31)] = acc[r];
  // 16x16x32: lane l: A[row l&15][k 8(l>>4)..+7], B[k 8(l>>4)..+7][col l&15]; acc[j]: row 4(l>>4) + j, col l&15 -- the top-left 16x16
  f32x4 c;
  for (int j = 0; j < 4; ++j) c[j] = bias[4 * (l >> 4) + j];
  {
    bf16x8 a, b;
    unsigned short ta[8], tb[8];
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * (l >> 4) + j;
      ta[j] = A[(l & 15) * 32 + k];
      tb[j] = B[k * 32 + (l & 15)];
    }
    memcpy(&a, ta, 16); memcpy(&b, tb, 16);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  for (int j = 0; j < 4; ++j) out16[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}

// ---- 2. f16 subnormals ------------------------------------------------------------------------------------------------
__global__ void k_f16(float* out, float aval, float bval) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)aval; b[j] = (_Float16)bval; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}

// ---- 3. rate ----------------------------------------------------------------------------------------------------------
constexpr int KS = 16;   // k-steps of 16 per half-row (d = 256)
#define M32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&(a), *(const bf16x8*)&(b), (c), 0, 0, 0)
#define M16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&(a), *(const bf16x8*)&(b), (c), 0, 0, 0)

// VARIANT 0: 32x32x16, a wave walks block 0 (KS k-steps) then block 1 (the shipped order): 8 LDS reads per 12 MFMAs
// VARIANT 1: 32x32x16, both blocks per k-step: 8 LDS reads per 24 MFMAs
// VARIANT 2: 16x16x32, 4 entity blocks of 16 x 8 query blocks of 16 per k-step of 32: 16 LDS reads per 96 MFMAs
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k_rate(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ uint4 lds[];   // [2 planes][8 blocks of 16 queries][KS/2 k32-steps][64] (the same bytes for every variant) + slack
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PLANE = 8 * (KS / 2) * 64 + 4 * 64;
  for (int i = threadIdx.x; i < 2 * PLANE; i += 256) lds[i] = in[(blockIdx.x * 2 * PLANE + i) % (1 << 16)];
  __syncthreads();
  const uint4* hl_hi = lds + lane;
  const uint4* hl_lo = lds + PLANE + lane;
  uint4 ah[8], al[8];
  for (int i = 0; i < 8; ++i) { ah[i] = in[((blockIdx.x * 4 + wave) * 16 + i) * 64 + lane]; al[i] = in[((blockIdx.x * 4 + wave) * 16 + 8 + i) * 64 + lane]; }
  float s = 0.f;
  if constexpr (VARIANT == 0) {
    f32x16 acc[2][4];
    for (int m = 0; m < 2; ++m) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[m][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int off = (it & 3) * 64;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          uint4 bh[4], bl[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) { bh[b] = hl_hi[(b * KS + ks) * 64 + off]; bl[b] = hl_lo[(b * KS + ks) * 64 + off]; }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            acc[m][b] = M32(al[(ks + m) & 7], bh[b], acc[m][b]);
            acc[m][b] = M32(ah[(ks + m) & 7], bl[b], acc[m][b]);
            acc[m][b] = M32(ah[(ks + m) & 7], bh[b], acc[m][b]);
          }
        }
    }
    for (int m = 0; m < 2; ++m) for (int b = 0; b < 4; ++b) s += acc[m][b][0] + acc[m][b][15];
  } else if constexpr (VARIANT == 1) {
    f32x16 acc[2][4];
    for (int m = 0; m < 2; ++m) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[m][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int off = (it & 3) * 64;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 bh[4], bl[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) { bh[b] = hl_hi[(b * KS + ks) * 64 + off]; bl[b] = hl_lo[(b * KS + ks) * 64 + off]; }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            acc[m][b] = M32(al[(ks + 4 * m) & 7], bh[b], acc[m][b]);
            acc[m][b] = M32(ah[(ks + 4 * m) & 7], bl[b], acc[m][b]);
            acc[m][b] = M32(ah[(ks + 4 * m) & 7], bh[b], acc[m][b]);
          }
      }
    }
    for (int m = 0; m < 2; ++m) for (int b = 0; b < 4; ++b) s += acc[m][b][0] + acc[m][b][15];
  } else {
    f32x4 acc[4][8];
    for (int m = 0; m < 4; ++m) for (int b = 0; b < 8; ++b) for (int r = 0; r < 4; ++r) acc[m][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int off = (it & 3) * 64;
#pragma unroll
      for (int ks = 0; ks < KS / 2; ++ks) {
        uint4 bh[8], bl[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) { bh[b] = hl_hi[(b * (KS / 2) + ks) * 64 + off]; bl[b] = hl_lo[(b * (KS / 2) + ks) * 64 + off]; }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            acc[m][b] = M16(al[(ks + 2 * m) & 7], bh[b], acc[m][b]);
            acc[m][b] = M16(ah[(ks + 2 * m) & 7], bl[b], acc[m][b]);
            acc[m][b] = M16(ah[(ks + 2 * m) & 7], bh[b], acc[m][b]);
          }
      }
    }
    for (int m = 0; m < 4; ++m) for (int b = 0; b < 8; ++b) s += acc[m][b][0] + acc[m][b][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT>
static void run_rate(const uint4* in, float* out, const char* tag) {
  const int nwg = 256;
  const size_t ldsb = (size_t)(2 * (8 * (KS / 2) * 64 + 4 * 64)) * 16;
  hipFuncSetAttribute((const void*)k_rate<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  // ~1.5 s of load first (the clock the chip holds under this load), then three timed launches
  for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((k_rate<VARIANT>), dim3(nwg), dim3(256), ldsb, 0, in, out, 4000);
  double best = 1e30, sum = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_rate<VARIANT>), dim3(nwg), dim3(256), ldsb, 0, in, out, 4000);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best; sum += ms;
  }
  // every variant: per iteration per wave 64 entities x 128 queries x 256 k x 3 MFMA-products
  const double flop = (double)nwg * 4 * 4000 * 64.0 * 128.0 * 256.0 * 2.0 * 3.0;
  printf("%-44s %.3f ms (best %.3f)  %.0f TFLOP/s of hardware MFMA\n", tag, sum / 3, best, flop / (sum / 3) / 1e9);
}

int main(int argc, char** argv) {
  // 1. bits
  {
    const int T = 4096;
    int mism = 0; double maxerr = 0, maxdiff = 0;
    unsigned short *dA, *dB; float *dbias, *d32, *d16;
    hipMalloc(&dA, 32 * 32 * 2); hipMalloc(&dB, 32 * 32 * 2); hipMalloc(&dbias, 32 * 4); hipMalloc(&d32, 32 * 32 * 4); hipMalloc(&d16, 16 * 16 * 4);
    srand(1);
    for (int t = 0; t < T; ++t) {
      unsigned short A[32 * 32], B[32 * 32]; float bias[32], o32[32 * 32], o16[16 * 16];
      const float sa = t % 3 == 0 ? 100.f : 1.f;     // some tiles with a wide dynamic range between the operands
      for (int i = 0; i < 32 * 32; ++i) { A[i] = f2bf(((rand() / (float)RAND_MAX) - 0.5f) * sa); B[i] = f2bf(((rand() / (float)RAND_MAX) - 0.5f) * (t % 5 == 0 ? 0.01f : 1.f)); }
      for (int i = 0; i < 32; ++i) bias[i] = ((rand() / (float)RAND_MAX) - 0.5f) * (t % 7 == 0 ? 50.f : 0.2f);
      hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice); hipMemcpy(dbias, bias, sizeof bias, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_bits, dim3(1), dim3(64), 0, 0, dA, dB, dbias, d32, d16);
      hipMemcpy(o32, d32, sizeof o32, hipMemcpyDeviceToHost); hipMemcpy(o16, d16, sizeof o16, hipMemcpyDeviceToHost);
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = bias[i];
          for (int k = 0; k < 32; ++k) ref += (double)bf2f(A[i * 32 + k]) * (double)bf2f(B[k * 32 + j]);
          const float x = o32[i * 32 + j], y = o16[i * 16 + j];
          if (memcmp(&x, &y, 4)) ++mism;
          maxerr = fmax(maxerr, fmax(fabs(x - ref), fabs(y - ref)) / (fabs(ref) + 1e-3));
          maxdiff = fmax(maxdiff, fabs((double)x - y));
        }
    }
    printf("bits: 2 x 32x32x16 against 1 x 16x16x32 over the same 32 k: %d of %d outputs differ (max |difference| %.3e); "
           "largest relative error of either against float64 %.3e (layout check)\n", mism, T * 256, maxdiff, maxerr);
  }
  // 2. f16 subnormals
  {
    float* d; hipMalloc(&d, 4); float v;
    const float sub = ldexpf(1.f, -20);   // subnormal in fp16 (smallest normal 2^-14)
    hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), 0, 0, d, sub, 1024.f);
    hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost);
    printf("f16 MFMA, A = 2^-20 (fp16 subnormal) x B = 1024 over 16 k: got %.6e, exact %.6e -> subnormal inputs are %s\n", v, 16.0 * ldexp(1.0, -10),
           v > 0 ? "KEPT" : "FLUSHED");
    hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), 0, 0, d, ldexpf(1.f, -24), 1.f);
    hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost);
    printf("f16 MFMA, A = 2^-24 (smallest fp16 subnormal) x B = 1: got %.6e, exact %.6e\n", v, 16.0 * ldexp(1.0, -24));
  }
  // 3. rate
  {
    const size_t n = (size_t)1 << 20;   // uint4s of random bf16 pairs (finite: exponent bits masked into a narrow range)
    std::vector<unsigned> hbuf(n * 4);
    srand(2);
    for (auto& w : hbuf) {
      const unsigned short a = f2bf((rand() / (float)RAND_MAX) - 0.5f), b = f2bf((rand() / (float)RAND_MAX) - 0.5f);
      w = (unsigned)a | ((unsigned)b << 16);
    }
    uint4* din; float* dout;
    hipMalloc(&din, n * 16); hipMalloc(&dout, 256 * 256 * 4);
    hipMemcpy(din, hbuf.data(), n * 16, hipMemcpyHostToDevice);
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    if (which < 0 || which == 0) run_rate<0>(din, dout, "32x32x16, block after block (8 reads / 12 MFMA)");
    if (which < 0 || which == 1) run_rate<1>(din, dout, "32x32x16, both blocks per k-step (8 / 24)");
    if (which < 0 || which == 2) run_rate<2>(din, dout, "16x16x32, 64 entities x 128 queries (16 / 96)");
    if (which < 0) {
      run_rate<0>(din, dout, "32x32x16, block after block (again)");
      run_rate<2>(din, dout, "16x16x32 (again)");
    }
  }
  return 0;
}
