// Where does dwconv7_ln's time go?  One binary per DW_DBG value (0 full, 1 centre row only, 2 loads without the stencil FMAs):
//   for d in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDW_DBG=$d -I include -I mtg-vision_amd/csrc \
//       -Xclang -target-feature -Xclang -packed-fp32-ops tools/micro/dwconv_probe.hip -o tools/micro/build/dwconv_probe$d; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include "dwconv7_ln_kernel.h"
namespace mtgv { void set_last_error(const std::string&) {} }
using namespace mtgv;
int main() {
  const int shapes[3][4] = {{256, 48, 32, 96}, {256, 24, 16, 192}, {256, 12, 8, 384}};
  for (auto& sh : shapes) {
    const int N = sh[0], H = sh[1], W = sh[2], C = sh[3];
    const size_t n = (size_t)N * H * W * C;
    float *in, *out, *w49, *b, *lw, *lb;
    hipMalloc(&in, n * 4), hipMalloc(&out, n * 4), hipMalloc(&w49, 49 * C * 4), hipMalloc(&b, C * 4), hipMalloc(&lw, C * 4), hipMalloc(&lb, C * 4);
    hipMemset(in, 0x3c, n * 4), hipMemset(w49, 0x3c, 49 * C * 4), hipMemset(b, 0, C * 4), hipMemset(lw, 0x3c, C * 4), hipMemset(lb, 0, C * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) dwconv7_ln_launch_t<0>(in, w49, b, lw, lb, out, N, H, W, C, 1e-6f, nullptr, 1);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) dwconv7_ln_launch_t<0>(in, w49, b, lw, lb, out, N, H, W, C, 1e-6f, nullptr, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("DW_DBG=%d  %dx%dx%dx%d: %.1f us per launch (%.2f TB/s of in+out)\n", DW_DBG, N, H, W, C, ms / 20 * 1e3, 2.0 * n * 4 / (ms / 20 * 1e-3) / 1e12);
    hipFree(in), hipFree(out), hipFree(w49), hipFree(b), hipFree(lw), hipFree(lb);
  }
  return 0;
}
