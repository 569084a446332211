// Row-group dwconv7_ln (dwconv7_ln_rows_kernel<TW, TH>) against the single-row kernel: time and bit-equality.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mtg-vision_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops \
//       tools/micro/dwconv_rows_probe.hip -o tools/micro/build/dwconv_rows_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "dwconv7_ln_kernel.h"
namespace mtgv { void set_last_error(const std::string&) {} }
using namespace mtgv;

template <typename F>
static float time_us(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) f();
  hipEventRecord(e0);
  for (int it = 0; it < 20; ++it) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}

int main() {
  const int shapes[4][4] = {{256, 48, 32, 96}, {256, 24, 16, 192}, {256, 12, 8, 384}, {256, 6, 4, 768}};
  for (auto& sh : shapes) {
    const int N = sh[0], H = sh[1], W = sh[2], C = sh[3];
    const size_t n = (size_t)N * H * W * C;
    float *in, *out, *ref, *w49, *b, *lw, *lb;
    hipMalloc(&in, n * 4), hipMalloc(&out, n * 4), hipMalloc(&ref, n * 4), hipMalloc(&w49, 49 * C * 4), hipMalloc(&b, C * 4), hipMalloc(&lw, C * 4),
        hipMalloc(&lb, C * 4);
    std::vector<float> h(n);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (float* p : {w49, b, lw, lb}) {
      const size_t m = p == w49 ? 49 * C : C;
      for (size_t i = 0; i < m; ++i) h[i] = (rand() % 2001 - 1000) * 1e-3f;
      hipMemcpy(p, h.data(), m * 4, hipMemcpyHostToDevice);
    }
    auto base = [&] { dwconv7_ln_launch_t<0>(in, w49, b, lw, lb, ref, N, H, W, C, 1e-6f, nullptr, 1); };
    printf("%dx%dx%dx%d  single-row: %.1f us\n", N, H, W, C, time_us(base));
    std::vector<float> hr(n), ho(n);
    hipMemcpy(hr.data(), ref, n * 4, hipMemcpyDeviceToHost);
    auto run = [&](const char* name, auto f) {
      hipMemset(out, 0xff, n * 4);
      const float us = time_us(f);
      hipMemcpy(ho.data(), out, n * 4, hipMemcpyDeviceToHost);
      printf("   %-12s %.1f us  %s\n", name, us, memcmp(ho.data(), hr.data(), n * 4) ? "DIFFERS" : "bit-identical");
    };
#define GOW(TW_, TH_) if (C <= 192) run("TW" #TW_ " TH" #TH_ " WL", [&] { dwconv7_ln_rows_launch<TW_, TH_, true, true>(in, w49, b, lw, lb, out, N, H, W, C, 1e-6f, nullptr); })
#define GO(TW_, TH_) run("TW" #TW_ " TH" #TH_, [&] { dwconv7_ln_rows_launch<TW_, TH_, true>(in, w49, b, lw, lb, out, N, H, W, C, 1e-6f, nullptr); })
    if (W >= 8) {
      GO(8, 1); GO(8, 2); GO(8, 3); GO(8, 4);
      GOW(8, 1); GOW(8, 2); GOW(8, 3); GOW(8, 4);
    }
    GO(4, 3); GOW(4, 1); GOW(4, 2); GOW(4, 3); GOW(4, 6);
    hipFree(in), hipFree(out), hipFree(ref), hipFree(w49), hipFree(b), hipFree(lw), hipFree(lb);
  }
  return 0;
}
