// Bandwidth-bound NHWC kernels of the path: LayerNorm, depthwise 7x7, pooling, resampling, layout.
#pragma once
#include "common.h"

namespace mtgv {

// per-row LayerNorm over C contiguous floats (biased variance, eps inside the sqrt):
// convnextv2.py:150-160 (both data formats are "per pixel over C" once the tensor is NHWC).
// out_fmt 1: the output rows are written in SP8 (sp8.h) for an LDS-DMA GEMM (C, ldo, o_off multiples of 8)
void ln_rows_launch(const float* in, int ldi, int i_off, float* out, int ldo, int o_off, const float* w, const float* b,
                    long rows, int C, float eps, hipStream_t s, int out_fmt = 0);

// depthwise 7x7, pad 3, + bias.  w49 is the weight repacked to [49][C] (tap-major).  convnextv2.py:198-200, :214
void dwconv7_launch(const float* in, const float* w49, const float* bias, float* out, int N, int H, int W, int C,
                    hipStream_t s);

// depthwise 7x7 + bias followed by LayerNorm over C, fused (no un-normalised intermediate in HBM)
bool dwconv7_ln_supported(int W, int C);
void dwconv7_ln_launch(const float* in, const float* w49, const float* bias, const float* ln_w, const float* ln_b, float* out,
                       int N, int H, int W, int C, float eps, hipStream_t s, int out_fmt = 0);

// (N,C,H,W) f32 -> (N,H,W,Cp) f32, y = x*scale + shift, channels C..Cp-1 zero.  (x*2-1: convnextv2ae.py:257-258)
void nchw_to_nhwc_launch(const float* in, float* out, int N, int C, int H, int W, int Cp, float scale, float shift,
                         hipStream_t s);
// (N,H,W,C) u8 -> (N,H,W,Cp) f32, y = (u/255)*scale + shift   (img_float32: util/image.py:220-237)
void u8_to_f32_launch(const uint8_t* in, float* out, long pixels, int C, int Cp, float scale, float shift, int flip_rgb,
                      hipStream_t s);
// (N,H,W,C) f32 in [0,1] (clipped) -> (N,H,W,Cp) f32 scaled
void f32hwc_scale_launch(const float* in, float* out, long pixels, int C, int Cp, float scale, float shift, hipStream_t s);

// global average pool over HW: (N,HW,C) -> (N,C)   (convnextv2.py:292-296, convnextv2ae.py:38-41)
void gap_launch(const float* in, float* out, int N, int HW, int C, hipStream_t s);

// out = x / max(||x||_2, 1e-12) per row (the normalisation Distance.COSINE implies, qdrant.py:29-32)
void l2norm_rows_launch(const float* in, float* out, long rows, int D, hipStream_t s);

// YOLO plumbing on channel slices of NHWC buffers
void maxpool5_launch(const float* in, int ci_total, int ci_off, float* out, int co_total, int co_off, int N, int H, int W,
                     int C, hipStream_t s);
void upsample2x_launch(const float* in, int ci_total, int ci_off, float* out, int co_total, int co_off, int N, int H, int W,
                       int C, hipStream_t s);
void copy_channels_launch(const float* in, int ci_total, int ci_off, float* out, int co_total, int co_off, long pixels, int C,
                          hipStream_t s);

}  // namespace mtgv
