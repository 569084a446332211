// One tile configuration of the LDS-DMA split GEMM per translation unit (they compile in parallel): a file defines
// SP_CFG_ID, SP_WM, SP_WN, SP_TM, SP_TN and includes this header.
#pragma once
#include "gemm_sp_kernel.h"

namespace mtgv {

namespace {
constexpr int SP_KS = 2;

template <int AMODE, int ACT>
void sp_launch_one(const SpDev& g, hipStream_t s) {
  constexpr int BM = 32 * SP_TM * SP_WM, BN = 32 * SP_TN * SP_WN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * 64 * SP_KS;
  static bool attr_done = false;
  auto kern = gemm_sp_kernel<SP_WM, SP_WN, SP_TM, SP_TN, SP_KS, AMODE, ACT>;
  if (!attr_done) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(64 * SP_WM * SP_WN), lds, s, g);
}
}  // namespace

#define SP_CAT2(a, b) a##b
#define SP_CAT(a, b) SP_CAT2(a, b)

// amode: 0 dense SP8 rows, 1 f32 rows through registers, 2 SP8 NHWC gather (conv)
void SP_CAT(gemm_sp_launch_cfg, SP_CFG_ID)(const SpDev& g, int amode, hipStream_t s) {
  if (amode == 0) {
    switch (g.act) {
      case ACT_NONE: sp_launch_one<0, ACT_NONE>(g, s); break;
      case ACT_MISH: sp_launch_one<0, ACT_MISH>(g, s); break;
      case ACT_GELU: sp_launch_one<0, ACT_GELU>(g, s); break;
      case ACT_SILU: sp_launch_one<0, ACT_SILU>(g, s); break;
      default: sp_launch_one<0, -1>(g, s); break;
    }
  } else if (amode == 1) {
    if (g.act == ACT_NONE) sp_launch_one<1, ACT_NONE>(g, s);
    else sp_launch_one<1, -1>(g, s);
  } else {
    switch (g.act) {
      case ACT_NONE: sp_launch_one<2, ACT_NONE>(g, s); break;
      case ACT_SILU: sp_launch_one<2, ACT_SILU>(g, s); break;
      default: sp_launch_one<2, -1>(g, s); break;
    }
  }
}

}  // namespace mtgv
