// Split-precision instantiations of the implicit-GEMM kernel (gemm_kernel.h, PREC 1): every f32 operand is
// split on its way into LDS into fp16 hi + lo planes and each product costs three v_mfma_f32_32x32x16_f16
// (lo*hi + hi*lo + hi*hi, f32 accumulate).  A separate translation unit so that it compiles beside gemm_f32.hip.
#include "gemm_kernel.h"

namespace mtgv {

bool gemm_dispatch_f16x3(const GemmDev& g, const GemmPlan& pl, bool conv, bool apro, int grid, hipStream_t s) {
  return gemm_dispatch<1>(g, pl, conv, apro, grid, s);
}

}  // namespace mtgv
