// Bit-compare sp8_split4_mix (v_fma_mix_f32 remainder) with sp8_split4, and the two Mish forms, on 4M random inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mtg-vision_amd/csrc tools/micro/split_mix_check.hip -o /tmp/split_mix_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "act.h"
#include "sp8.h"
using namespace mtgv;

__global__ void k(const float* in, sp_h4* a, sp_h4* b, float* m, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const sp_f4 x = *reinterpret_cast<const sp_f4*>(in + i * 4);
  sp_h4 hi, lo, hi2, lo2;
  sp8_split4(x, hi, lo);
  sp8_split4_mix(x, hi2, lo2);
  a[2 * i] = hi, a[2 * i + 1] = lo, b[2 * i] = hi2, b[2 * i + 1] = lo2;
  for (int e = 0; e < 4; ++e) m[i * 4 + e] = act_mish(x[e]);
}

int main() {
  const long n = 1 << 22;
  std::vector<float> h(n);
  srand(1);
  for (long i = 0; i < n; ++i) {
    const float u = (float)rand() / RAND_MAX * 2 - 1;
    const int ex = rand() % 40 - 24;  // magnitudes 2^-24 .. 2^15
    h[i] = ldexpf(u, ex);
  }
  h[0] = 0.f, h[1] = -0.f, h[2] = 65504.f, h[3] = 1e-9f, h[4] = 100.f, h[5] = -100.f, h[6] = 30.f, h[7] = -30.f;
  float *d, *dm;
  sp_h4 *da, *db;
  hipMalloc(&d, n * 4), hipMalloc(&dm, n * 4), hipMalloc(&da, n * 4), hipMalloc(&db, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n / 4 + 255) / 256, 256>>>(d, da, db, dm, n / 4);
  std::vector<unsigned> a(n), b(n);
  std::vector<float> m(n);
  hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (long i = 0; i < n; ++i) bad += a[i] != b[i];
  double worst = 0, worst_abs = 0;
  for (long i = 0; i < n; ++i) {
    const double x = h[i], ref = x * tanh(log1p(exp(x)));
    const double err = fabs(m[i] - ref);
    if (err > worst_abs) worst_abs = err;
    const double rel = err / (fabs(x) + 1e-30);
    if (rel > worst) worst = rel;
  }
  printf("split: %ld of %ld words differ; mish: max abs err %.3g, max err / |x| %.3g\n", bad, n, worst_abs, worst);
  return bad != 0 || !(worst < 5e-7);
}
