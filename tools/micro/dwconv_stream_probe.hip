// Row-streaming dwconv7_ln (dwconv7_ln_stream_kernel: LDS-DMA row ring, one channel per thread, taps in registers) against
// the shipped row-group kernel: time and bit-equality, SP8 and f32 output.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mtg-vision_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops \
//       tools/micro/dwconv_stream_probe.hip -o tools/micro/build/dwconv_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "dwconv7_ln_kernel.h"
#include "dwconv7_ln_stream_kernel.h"
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

template <int C, int G, int TW>
static void shape(int N, int H) {
  constexpr int W = G * TW;
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
  std::vector<float> hr(n), ho(n);
  for (int fmt = 1; fmt >= 0; --fmt) {
    auto base = [&] { dwconv7_ln_launch_t<0>(in, w49, b, lw, lb, ref, N, H, W, C, 1e-6f, nullptr, fmt); };
    printf("%dx%dx%dx%d %s  shipped (row groups): %.1f us\n", N, H, W, C, fmt ? "SP8" : "f32", time_us(base));
    hipMemcpy(hr.data(), ref, n * 4, hipMemcpyDeviceToHost);
    auto run = [&](const char* name, auto f) {
      hipMemset(out, 0xff, n * 4);
      const float us = time_us(f);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(ho.data(), out, n * 4, hipMemcpyDeviceToHost);
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < n; ++i)
        if (memcmp(&ho[i], &hr[i], 4)) { if (!bad) first = i; ++bad; }
      printf("   %-22s %.1f us  %s", name, us, bad ? "DIFFERS" : "bit-identical");
      if (bad) printf(" (%zu of %zu words, first at %zu: %g vs %g)", bad, n, first, ho[first], hr[first]);
      if (e != hipSuccess) printf(" [%s]", hipGetErrorString(e));
      printf("\n");
    };
    for (int bands : {1, 2}) {
      char nm[64];
      if (fmt) {
        snprintf(nm, sizeof nm, "stream TH3 bands=%d", bands);
        run(nm, [&] { dwconv7_ln_stream_launch<C, G, TW, 3, true>(in, w49, b, lw, lb, out, N, H, bands, 1e-6f, nullptr); });
        snprintf(nm, sizeof nm, "stream TH2 bands=%d", bands);
        run(nm, [&] { dwconv7_ln_stream_launch<C, G, TW, 2, true>(in, w49, b, lw, lb, out, N, H, bands, 1e-6f, nullptr); });
      } else {
        snprintf(nm, sizeof nm, "stream TH3 bands=%d", bands);
        run(nm, [&] { dwconv7_ln_stream_launch<C, G, TW, 3, false>(in, w49, b, lw, lb, out, N, H, bands, 1e-6f, nullptr); });
      }
    }
  }
  hipFree(in), hipFree(out), hipFree(ref), hipFree(w49), hipFree(b), hipFree(lw), hipFree(lb);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 256;
  shape<96, 8, 4>(N, 48);
  shape<192, 4, 4>(N, 24);
  shape<384, 2, 4>(N, 12);
  shape<768, 1, 4>(N, 6);
  shape<96, 8, 4>(3, 47);  // ragged height, few images
  return 0;
}
