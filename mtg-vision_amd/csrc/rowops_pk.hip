// dwconv7_ln with packed-FP32 VALU instructions allowed (build.py compiles *_pk.hip without "-packed-fp32-ops" off):
// v_pk_fma_f32 performs two of the 49-tap stencil's FMAs per instruction, the kernel's binding resource
// (profiles/r02_sq_counters_step.txt: VALU-bound).  Launched only while packed_fp32_allowed() (rowops.hip).
#include "dwconv7_ln_kernel.h"
#include "rowops.h"

namespace mtgv {

void dwconv7_ln_launch_pk(const float* in, const float* w49, const float* bias, const float* ln_w, const float* ln_b, float* out,
                          int N, int H, int W, int C, float eps, hipStream_t s, int out_fmt) {
  dwconv7_ln_launch_t<1>(in, w49, bias, ln_w, ln_b, out, N, H, W, C, eps, s, out_fmt);
}

}  // namespace mtgv
