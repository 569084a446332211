// Elementwise activations, f32, same formulas the reference's PyTorch CPU path evaluates.
#pragma once
#include "common.h"

namespace mtgv {

// exp / reciprocal on the hardware transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp each).  The GEMM
// epilogues evaluate one activation per output element; with the library expf and an IEEE divide the
// K = 96 / 192 pointwise layers were VALU-bound, not MFMA-bound.  Relative error of the activations
// below stays under ~1e-6, well inside the 1e-4 embedding tolerance (tests/test_gpu_encoder.py).
//
// Every function below starts with "fp contract(off)": whether a*b+c fuses must not depend on the template
// instantiation a call is inlined into - a frame's detections are bit-identical alone or inside a batch, although
// the two cases pick different GEMM tiles (tests/test_gpu_fullsize.py).
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// v_rcp_f32 itself (1 ulp).  __frcp_rn is the correctly rounded reciprocal: hipcc expands it into the full division
// sequence (2 v_div_scale, v_rcp, 3 v_fma, v_div_fmas, v_div_fixup) - ten instructions per activation.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// nn.GELU() erf form - mtgvision/models/convnextv2.py:192-193
// erff is evaluated here, branch-free and inline, with the device library's own algorithm and coefficients (ROCm
// ocml erfF: an odd polynomial in x below 1, 1 - exp(-q(|x|)) above; same fma chain, same accurate expf, so the bits
// are erff's) - a call per element costs a stack frame in the GEMM epilogue (52k cycles per tile instead of 13k).
__device__ __forceinline__ float erf_inline(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = x * x;
  float p = __builtin_fmaf(t, -0x1.268bc2p-11f, 0x1.420828p-8f);
  p = __builtin_fmaf(t, p, -0x1.b5937p-6f);
  p = __builtin_fmaf(t, p, 0x1.ce077cp-4f);
  p = __builtin_fmaf(t, p, -0x1.81266p-2f);
  p = __builtin_fmaf(t, p, 0x1.06eba0p-3f);
  const float small = __builtin_fmaf(ax, p, ax);
  float q = __builtin_fmaf(ax, 0x1.1d3156p-16f, -0x1.8d129p-12f);
  q = __builtin_fmaf(ax, q, 0x1.f9a6d2p-9f);
  q = __builtin_fmaf(ax, q, -0x1.8c3164p-6f);
  q = __builtin_fmaf(ax, q, 0x1.b4e9c8p-4f);
  q = __builtin_fmaf(ax, q, 0x1.4515fap-1f);
  q = __builtin_fmaf(ax, q, 0x1.078e50p-3f);
  q = __builtin_fmaf(ax, q, ax);
  const float large = 1.0f - expf(-q);
  return __builtin_copysignf(ax < 1.0f ? small : large, x);
}

__device__ __forceinline__ float act_gelu(float x) {
#pragma clang fp contract(off)
  return 0.5f * x * (1.0f + erf_inline(x * 0.70710678118654752440f));
}

// nn.Mish = x * tanh(softplus(x)), softplus threshold 20 - mtgvision/models/convnextv2ae.py:17-18
// tanh(log(1 + e^x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = 1 - 2 / d with d = e^x (e^x + 2) + 2, so
//   mish(x) = x - 2 x / d:   one exp, one reciprocal, no log1p / tanh, and no clamp: for large x, e^x and d overflow to
// +inf, 1/d = 0 and the result is x exactly - what torch's softplus threshold gives (tanh(x) = 1 in f32 beyond 20); for
// very negative x, d -> 2 and the result -> 0.  Absolute error <= ~1.2e-7 |x| (the subtraction's rounding), i.e. at the
// f32 level of the GEMM that produced x.  scale: an optional factor applied to the result (GRN multiplier in the fused
// MLP kernel), folded into the two terms.
__device__ __forceinline__ float act_mish_scaled(float x, float scale) {
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
  const float d = __builtin_fmaf(e, e + 2.0f, 2.0f);
  const float xs = x * scale;
  return __builtin_fmaf(xs * fast_rcp(d), -2.0f, xs);
}
__device__ __forceinline__ float act_mish(float x) {
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
  const float d = __builtin_fmaf(e, e + 2.0f, 2.0f);
  return __builtin_fmaf(x * fast_rcp(d), -2.0f, x);
}

// SiLU of the YOLO Conv block (ultralytics Conv.default_act; call site od_export.py:150)
__device__ __forceinline__ float act_silu(float x) {
#pragma clang fp contract(off)
  return x * fast_rcp(1.0f + fast_exp(-x));
}

__device__ __forceinline__ float act_sigmoid(float x) {
#pragma clang fp contract(off)
  return 1.0f / (1.0f + expf(-x));
}

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
