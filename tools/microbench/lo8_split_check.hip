// Numeric check of split16.h: split8_q8 / lo8_decode (the -DCOPER_FUSED_LO8 experiment): the decoded bytes equal the fp16 plane bit for
// bit; |v - hi - lo| <= 2^-18 |v| (2^-19 but for the clamped +128).  hipcc -O3 --offload-arch=gfx950 -I../../coper_amd/csrc lo8_split_check.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include "split16.h"
using namespace coper;
__global__ void k(const float* v, uint4* hi, uint4* lo16, uint2* lo8, uint4* dec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 h, l; uint2 b;
  split8_q8(v + 8 * i, h, l, b);
  hi[i] = h; lo16[i] = l; lo8[i] = b;
  lo8_u32x4 d = lo8_decode((lo8_u32x2){b.x, b.y}, (lo8_u32x4){h.x, h.y, h.z, h.w});
  dec[i] = make_uint4(d[0], d[1], d[2], d[3]);
}
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
static float h2f(unsigned short h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
int main() {
  const int N = 1 << 16;
  std::vector<float> v(8 * N);
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (int i = 0; i < 8 * N; ++i) {
    float s = (i % 7 == 0) ? 32000.f : (i % 5 == 0 ? 1e-3f : (i % 3 == 0 ? 8.f : 6000.f));
    v[i] = nd(g) * s;
    if (i % 1001 == 0) v[i] = 0.f;
    if (i % 1003 == 0) v[i] = 16384.f + 8.f;     // exactly half an ulp above a power of two
    if (fabsf(v[i]) > 32767.f) v[i] = 32767.f;
  }
  float* dv; uint4 *dh, *dl, *dd; uint2* db;
  hipMalloc(&dv, 32 * N); hipMalloc(&dh, 16 * N); hipMalloc(&dl, 16 * N); hipMalloc(&dd, 16 * N); hipMalloc(&db, 8 * N);
  hipMemcpy(dv, v.data(), 32 * N, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, 0, dv, dh, dl, db, dd);
  std::vector<unsigned short> hh(8 * N), ll(8 * N), dd2(8 * N);
  hipMemcpy(hh.data(), dh, 16 * N, hipMemcpyDeviceToHost); hipMemcpy(ll.data(), dl, 16 * N, hipMemcpyDeviceToHost);
  hipMemcpy(dd2.data(), dd, 16 * N, hipMemcpyDeviceToHost);
  long bad = 0; double worst = 0, worst_small = 0; long nz = 0;
  for (int i = 0; i < 8 * N; ++i) {
    if ((ll[i] & 0x7fff) != (dd2[i] & 0x7fff) || (((ll[i] ^ dd2[i]) & 0x8000) && (ll[i] & 0x7fff))) ++bad;
    const double err = fabs((double)v[i] - (double)h2f(hh[i]) - (double)h2f(ll[i]));
    if (fabsf(v[i]) >= 16.f) { const double r = err / fabs((double)v[i]); if (r > worst) worst = r; ++nz; }
    else if (err > worst_small) worst_small = err;
  }
  printf("decode != lo16 plane: %ld of %d;  max rel err (|v| >= 16, %ld values): %.3g (2^-19 = %.3g);  max abs err below 16: %.3g\n", bad, 8 * N, nz, worst, ldexp(1.0, -19), worst_small);
  return bad != 0;
}
