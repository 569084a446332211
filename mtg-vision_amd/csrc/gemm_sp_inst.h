// One tile configuration of the LDS-DMA split GEMM per translation unit (they compile in parallel): a file defines
// SP_CFG_ID, SP_WM, SP_WN, SP_TM, SP_TN and includes this header.
#pragma once
#include <stdlib.h>

#include "gemm_sp_kernel.h"

namespace mtgv {

namespace {
#ifndef SP_KS_VALUE
#define SP_KS_VALUE 2
#endif
constexpr int SP_KS = SP_KS_VALUE;  // k16 steps per stage: 2 (32-k stages); 1 in the 16-channel window-conv configuration
#ifndef SP_NST
#define SP_NST 2
#endif

template <int AMODE, int ACT, int EPI, int NST = SP_NST>
void sp_launch_nst(const SpDev& g, hipStream_t s) {
  constexpr int BM = 32 * SP_TM * SP_WM, BN = 32 * SP_TN * SP_WN;
  constexpr size_t ring = (size_t)NST * ((AMODE == 5 ? BN : BM + BN) * 64 * SP_KS + (AMODE == 3 ? 1024 : 0));
  size_t lds = ring;
  // chained 1x1 (EPI 32): after the main loop the block holds one accumulator column block and TN A2 stages per wave and W2
  constexpr size_t chain_lds = (size_t)SP_WM * SP_WN * 4096 * (1 + SP_TN) + (size_t)SP_TN * 96 * 128;
  if (AMODE == 5) {  // window of BM + 2 W + 2 pixels x (64 KS) B in front of the weight ring; the epilogue stages 32 rows per wave
    constexpr int RB = 64 * SP_KS, RPP = 1024 / RB;
    const size_t win = (size_t)((BM + 2 * g.Wd + 2 + RPP - 1) / RPP * RPP) * RB;
    const size_t stage = (size_t)SP_WM * SP_WN * 32 * 128 * SP_TN;
    lds = win + ring > stage ? win + ring : stage;
  }
  if (EPI == 32 && lds < chain_lds) lds = chain_lds;
  static bool attr_done_dev[MTGV_MAX_DEVICES] = {};  // hipFuncSetAttribute is per device
  bool& attr_done = attr_done_dev[current_device()];
  auto kern = gemm_sp_kernel<SP_WM, SP_WN, SP_TM, SP_TN, SP_KS, NST, AMODE, ACT, EPI>;
  if (!attr_done) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (AMODE == 5 || EPI == 32) ? 160 * 1024 : (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(64 * SP_WM * SP_WN), lds, s, g);
}

template <int AMODE, int ACT, int EPI>
void sp_launch_one(const SpDev& g, hipStream_t s) {
  if constexpr (AMODE == 5 && SP_NST == 2) {
    // window conv: a four-deep weight ring (three taps ahead) for launches of at most one round of tiles - there a tile's
    // latency is the launch's duration (12800-row layers -15..-25 %); with several rounds the blocks per CU matter more
    // (the deeper ring costs one: 204800 x 32 layers +12 %) and the two-deep ring stays
    constexpr int BM = 32 * SP_TM * SP_WM, BN = 32 * SP_TN * SP_WN, RB = 64 * SP_KS, RPP = 1024 / RB;
    const size_t win = (size_t)((BM + 2 * g.Wd + 2 + RPP - 1) / RPP * RPP) * RB;
    static const bool deep_on = [] { const char* e = getenv("MTGV_SP_WINRING"); return e == nullptr || atoi(e) != 0; }();
    if (deep_on && (long)g.tiles_m * g.tiles_n <= 512 && win + (size_t)4 * BN * RB <= 80 * 1024) {
      sp_launch_nst<AMODE, ACT, EPI, 4>(g, s);
      return;
    }
  }
  sp_launch_nst<AMODE, ACT, EPI, SP_NST>(g, s);
}

// the compile-time epilogue shape of a launch (gemm_sp_kernel.h, EPI), or -1 when only the generic one fits
int sp_epi_of(const SpDev& g) {
  if (g.W2 != nullptr) return 32;  // chained 1x1
  if (g.remap || g.N % 4 != 0) return -1;
  return (g.out_fmt == 1 ? 1 : 0) | (g.res != nullptr ? (g.res_fmt == 1 ? 4 : 2) : 0) | (g.grn_part != nullptr ? 8 : 0);
}

// launches the instance whose EPI equals `epi` if it is one of those listed, the generic one otherwise
template <int AMODE, int ACT>
void sp_pick(const SpDev& g, int epi, hipStream_t s) {
  MTGV_CHECK(epi != 32, ERR_RUNTIME, "gemm_sp: no chained-1x1 instance for A mode %d in configuration %d", AMODE, SP_CFG_ID);
  sp_launch_one<AMODE, ACT, -1>(g, s);
}
template <int AMODE, int ACT, int E0, int... ES>
void sp_pick(const SpDev& g, int epi, hipStream_t s) {
  if (epi == E0) sp_launch_one<AMODE, ACT, E0>(g, s);
  else sp_pick<AMODE, ACT, ES...>(g, epi, s);
}
}  // namespace

#define SP_CAT2(a, b) a##b
#define SP_CAT(a, b) SP_CAT2(a, b)

// amode: 0 dense SP8 rows, 1 f32 rows through registers, 2 SP8 NHWC gather (conv), 3 / 4 f32 rows by DMA with /
// without per-image multipliers, 5 SP8 3x3 stride-1 conv out of a staged input window
// Specialised epilogues: 0 f32 out; 1 SP8 out; 1|4 SP8 out + SP8 residual (detector); 2 f32 out + f32 residual
// (pwconv2); 8 f32 out + GRN sums (pwconv1).
void SP_CAT(gemm_sp_launch_cfg, SP_CFG_ID)(const SpDev& g, int amode, hipStream_t s) {
  if (g.topk > 0) {  // match path: f32 queries by DMA, fused top-k epilogue; only the 128 x 192 configuration carries it
#if SP_CFG_ID == 1
    MTGV_CHECK((amode == 4 || amode == 6) && g.act == ACT_NONE, ERR_INVALID, "gemm_sp: top-k needs aligned f32 queries");
    if (amode == 6) sp_launch_one<6, ACT_NONE, 16>(g, s);  // fp16 rows on both sides: the approximate first pass
    else sp_launch_one<4, ACT_NONE, 16>(g, s);
    return;
#else
    MTGV_CHECK(false, ERR_INVALID, "gemm_sp: no top-k instance in this configuration");
#endif
  }
  const int epi = sp_epi_of(g);
#if SP_KS_VALUE != 2
  // the 16-k configuration exists for the conv paths only (window conv, and the tap gather it falls back to)
  MTGV_CHECK(amode == 5 || amode == 2, ERR_RUNTIME, "gemm_sp: configuration %d runs convolutions only (A mode %d)", SP_CFG_ID, amode);
  if (amode == 5) {
    switch (g.act) {
      case ACT_SILU: sp_pick<5, ACT_SILU, 1, 5>(g, epi, s); break;
      default: sp_pick<5, -1>(g, epi, s); break;
    }
  } else {
    switch (g.act) {
      case ACT_SILU: sp_pick<2, ACT_SILU, 1, 5>(g, epi, s); break;
      default: sp_pick<2, -1>(g, epi, s); break;
    }
  }
  return;
#else
  if (amode == 0) {
    switch (g.act) {
      case ACT_NONE: sp_pick<0, ACT_NONE, 0, 1>(g, epi, s); break;
      case ACT_MISH: sp_pick<0, ACT_MISH, 8>(g, epi, s); break;
      case ACT_GELU: sp_pick<0, ACT_GELU, 8>(g, epi, s); break;
      case ACT_SILU: sp_pick<0, ACT_SILU, 1, 5>(g, epi, s); break;
      default: sp_pick<0, -1>(g, epi, s); break;
    }
  } else if (amode == 1) {
#if SP_NST == 2
    if (g.act == ACT_NONE) sp_pick<1, ACT_NONE>(g, epi, s);
    else sp_pick<1, -1>(g, epi, s);
#else
    MTGV_CHECK(false, ERR_RUNTIME, "gemm_sp: the register A path has no deep-ring instance");
#endif
  } else if (amode == 3) {
    if (g.act == ACT_NONE) sp_pick<3, ACT_NONE, 2>(g, epi, s);
    else sp_pick<3, -1>(g, epi, s);
  } else if (amode == 4) {
    if (g.act == ACT_NONE) sp_pick<4, ACT_NONE, 0, 2>(g, epi, s);
    else sp_pick<4, -1>(g, epi, s);
  } else if (amode == 5) {
#if SP_NST == 2
    switch (g.act) {
#if SP_WN == 1 && SP_TM == 1
      case ACT_SILU: sp_pick<5, ACT_SILU, 1, 5, 32>(g, epi, s); break;
#else
      case ACT_SILU: sp_pick<5, ACT_SILU, 1, 5>(g, epi, s); break;
#endif
      default: sp_pick<5, -1>(g, epi, s); break;
    }
#else
    MTGV_CHECK(false, ERR_RUNTIME, "gemm_sp: the window conv has no deep-ring instance");
#endif
  } else {
    switch (g.act) {
      case ACT_NONE: sp_pick<2, ACT_NONE, 0, 1>(g, epi, s); break;
#if SP_WN == 1 && SP_TM == 1
      case ACT_SILU: sp_pick<2, ACT_SILU, 1, 5, 32>(g, epi, s); break;
#else
      case ACT_SILU: sp_pick<2, ACT_SILU, 1, 5>(g, epi, s); break;
#endif
      default: sp_pick<2, -1>(g, epi, s); break;
    }
  }
#endif
}

}  // namespace mtgv
