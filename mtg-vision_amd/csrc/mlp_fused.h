// Fused ConvNeXt-V2 MLP (pwconv1 + activation + GRN + pwconv2 + residual) for the narrow stages: host side.
// Kernels and the reasoning: mlp_fused_kernel.h.
#pragma once
#include "common.h"

namespace mtgv {

// Can a Block with C channels and hw pixels per image run fused (f16x3 operand mode, MTGV_MLP_FUSED != 0)?
bool mlp_fused_supported(int C, int hw, int act);

// bytes of the permuted SP8 copy of W2 [C][4C] (followed by nothing: the row scales are a separate [C] float array)
inline size_t mlp_w2p_bytes(int C) { return (size_t)C * 4 * C * 4; }
// W2 (f32, [C][4C]) -> w2p + ws2[C]
void mlp_pack_w2p_launch(const float* W2, void* w2p, float* ws2, int C, hipStream_t s);

struct MlpArgs {
  const float* x_sp8 = nullptr;  // LayerNorm output, SP8 rows [M][C]
  const float* w1 = nullptr;     // f32 master [4C][C]: its registered SP8 copy is used
  const float* b1 = nullptr;
  const void* w2p = nullptr;
  const float* ws2 = nullptr;
  const float* b2 = nullptr;     // GRN beta folded in
  const float* gamma = nullptr;
  const float* res = nullptr;    // block input [M][C]
  float* out = nullptr;
  float* part = nullptr;         // >= (M / 32) * 4C floats
  float* scale = nullptr;        // >= n_img * 4C floats
  int n_img = 0, hw = 0, C = 0, act = 0;
  // optional: the LayerNorm over C that follows the stage (downsample) in the output pass's epilogue - out_ln receives SP8 rows
  // [M][C] and `out` is not written; out_ln may be the buffer `res` points into
  float* out_ln = nullptr;
  const float* ln_w = nullptr;
  const float* ln_b = nullptr;
  float ln_eps = 1e-6f;
};
// pass 1 -> grn_finalize -> pass 2 on stream s
void mlp_fused_launch(const MlpArgs& a, hipStream_t s);

}  // namespace mtgv
