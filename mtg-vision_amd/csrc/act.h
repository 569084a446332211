// Elementwise activations, f32, same formulas the reference's PyTorch CPU path evaluates.
#pragma once
#include "common.h"

namespace mtgv {

// nn.GELU() erf form - mtgvision/models/convnextv2.py:192-193
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// nn.Mish = x * tanh(softplus(x)), softplus threshold 20 - mtgvision/models/convnextv2ae.py:17-18
__device__ __forceinline__ float act_mish(float x) {
  const float sp = x > 20.0f ? x : log1pf(expf(x));
  return x * tanhf(sp);
}

// SiLU of the YOLO Conv block (ultralytics Conv.default_act; call site od_export.py:150)
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case ACT_GELU: return act_gelu(x);
    case ACT_MISH: return act_mish(x);
    case ACT_SILU: return act_silu(x);
    case ACT_SIGMOID: return act_sigmoid(x);
    default: return x;
  }
}

}  // namespace mtgv
