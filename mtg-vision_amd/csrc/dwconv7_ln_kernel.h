// Depthwise 7x7 + LayerNorm kernel bodies (instantiated in rowops.hip, built like the rest of the library without
// packed-FP32 VALU instructions; PK is always 0 - it kept a packed build of round 3 apart at link time).
#pragma once
#include "common.h"
#include "dwconv7_ln_stream_kernel.h"
#include "sp8.h"

namespace mtgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Depthwise 7x7 + LayerNorm over C in one pass (Block.dwconv + Block.norm, convnextv2.py:214-216): the conv
// result never goes to HBM un-normalised.  A block owns S whole strips (all C channels of TW pixels), so the
// per-pixel mean / variance over channels are block-local: partial sums go through LDS in a fixed order
// (two-pass variance like ln_rows_kernel, deterministic).
// ---------------------------------------------------------------------------
template <int TW, bool SP8, int PK, bool WL = false>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ in, const float* __restrict__ w49,
                                                        const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b, float* __restrict__ out, int H, int W,
                                                        int C, int nstrips, long total_strips, int S, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int c4n = C >> 2;
  float* part = sm;                      // [S][c4n][TW]: a pixel's partial sums sit TW floats apart, so the lanes that reduce
                                         // pixels 0..TW-1 read consecutive words (no bank conflicts) in the same c4 order
  float* stat = sm + S * TW * c4n;       // [S][TW][2] mean, rstd
  const float* wl = w49;                 // WL: the 49 x C tap table staged in LDS (a third of the kernel's L1 traffic otherwise)
  long blk;
  {
    const long nwg = gridDim.x, b = blockIdx.x;
    const long q = nwg >> 3, r = nwg & 7, x = b & 7;
    blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  const int tid = threadIdx.x;
  if (WL) {
    float* wdst = stat + ((S * TW * 2 + 3) & ~3);
    for (int i = tid; i < 49 * c4n; i += 256) reinterpret_cast<f32x4*>(wdst)[i] = reinterpret_cast<const f32x4*>(w49)[i];
    wl = wdst;
    __syncthreads();
  }
  const int sl = tid / c4n, c4 = tid % c4n;  // strip slot in block, channel quad
  const long strip = blk * S + sl;
  const bool live = sl < S && strip < total_strips;
  const int c = c4 * 4;
  int ws = 0, h = 0;
  long n = 0;
  if (live) {
    ws = (int)(strip % nstrips);
    const long t = strip / nstrips;
    h = (int)(t % H);
    n = t / H;
  }
  const int w0 = ws * TW;
  f32x4 acc[TW];
  if (live) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[j] = bv;
#ifndef DW_DBG
#define DW_DBG 0  // tools/micro/dwconv_probe.hip: 1 centre row only, 2 loads without the FMAs, 3 no LayerNorm
#endif
#pragma unroll 1
    for (int kh = 0; kh < 7; ++kh) {
      const int ih = h + kh - 3;
      if (ih < 0 || ih >= H) continue;
      if (DW_DBG == 1 && kh != 3) continue;
      const float* rowp = in + ((n * H + ih) * W) * C + c;
      f32x4 r[TW + 6];
#pragma unroll
      for (int j = 0; j < TW + 6; ++j) {
        const int iw = w0 + j - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iw >= 0 && iw < W) v = *reinterpret_cast<const f32x4*>(rowp + (long)iw * C);
        r[j] = v;
      }
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (kh * 7 + kw) * C + c);
#pragma unroll
        for (int j = 0; j < TW; ++j) {
          if (DW_DBG == 2) acc[j] += (kw == 0 ? r[j] + r[j + 6] : wv);  // keeps every load alive, 1/7 of the arithmetic
          else acc[j] += r[j + kw] * wv;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TW; ++j) part[(sl * c4n + c4) * TW + j] = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
  }
  __syncthreads();
  if (live && c4 < TW) {  // thread c4 of a strip reduces pixel j = c4
    const float* pp = part + sl * c4n * TW + c4;
    float sum = 0.f;
    for (int i = 0; i < c4n; ++i) sum += pp[i * TW];
    stat[(sl * TW + c4) * 2] = sum / (float)C;
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      const f32x4 d = acc[j] - stat[(sl * TW + j) * 2];
      part[(sl * c4n + c4) * TW + j] = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  __syncthreads();
  if (live && c4 < TW) {
    const float* pp = part + sl * c4n * TW + c4;
    float sq = 0.f;
    for (int i = 0; i < c4n; ++i) sq += pp[i * TW];
    stat[(sl * TW + c4) * 2 + 1] = 1.0f / sqrtf(sq / (float)C + eps);
  }
  __syncthreads();
  if (live) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(ln_w + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(ln_b + c);
    float* op = out + ((n * H + h) * W + w0) * C + c;
#pragma unroll
    for (int j = 0; j < TW; ++j)
      if (w0 + j < W) {
        const float mean = stat[(sl * TW + j) * 2], rstd = stat[(sl * TW + j) * 2 + 1];
        const f32x4 o = (acc[j] - mean) * rstd * wv + bv;
        if (SP8)  // quads c4, c4^1 of a strip are adjacent lanes with the same predicates (C % 8 == 0)
          *reinterpret_cast<sp_h8*>(reinterpret_cast<char*>(op + (long)j * C - c) + (c4 >> 1) * 32 + (c4 & 1) * 16) =
              sp8_piece_from_quad(o, c4);
        else
          *reinterpret_cast<f32x4*>(op + (long)j * C) = o;
      }
  }
}


// Row-group form of the same kernel: a thread keeps TH output rows of its TW-pixel strip in registers and walks the TH + 6
// input rows once, so an input row is fetched (TH + 6) / TH times instead of 7 (tools/micro/dwconv_probe.hip: the six
// extra row fetches, not the FMAs, are 45 % of the single-row kernel's time at stage 0).  Every output still accumulates
// bias, then kh ascending, kw ascending: bit-identical to dwconv7_ln_kernel.
template <int TW, int TH, bool SP8, bool WL = false>
__global__ __launch_bounds__(256) void dwconv7_ln_rows_kernel(const float* __restrict__ in, const float* __restrict__ w49,
                                                             const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                             const float* __restrict__ ln_b, float* __restrict__ out, int H, int W,
                                                             int C, int nstrips, int nhg, long total_strips, int S, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int P = TH * TW;             // pixels a thread owns
  const int c4n = C >> 2;
  float* part = sm;                      // [S][c4n][P]
  float* stat = sm + S * P * c4n;        // [S][P][2] mean, rstd
  const float* wl = w49;
  const int tid = threadIdx.x;
  if (WL) {
    float* wdst = stat + ((S * P * 2 + 3) & ~3);
    for (int i = tid; i < 49 * c4n; i += 256) reinterpret_cast<f32x4*>(wdst)[i] = reinterpret_cast<const f32x4*>(w49)[i];
    wl = wdst;
    __syncthreads();
  }
  const int sl = tid / c4n, c4 = tid % c4n;
  const int c = c4 * 4;
  long blk;
  {
    const long nwg = gridDim.x, b = blockIdx.x;
    const long q = nwg >> 3, r = nwg & 7, x = b & 7;
    blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  const long strip = blk * S + sl;
  const bool live = sl < S && strip < total_strips;
  int ws = 0, h0 = 0;
  long n = 0;
  if (live) {
    ws = (int)(strip % nstrips);
    const long t = strip / nstrips;
    h0 = (int)(t % nhg) * TH;
    n = t / nhg;
  }
  const int w0 = ws * TW;
  f32x4 acc[TH][TW];
  if (live) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[t][j] = bv;
#pragma unroll 1
    for (int ir = 0; ir < TH + 6; ++ir) {
      const int ih = h0 + ir - 3;
      if (ih < 0 || ih >= H) continue;
      const float* rowp = in + ((n * H + ih) * W) * C + c;
      f32x4 r[TW + 6];
#pragma unroll
      for (int j = 0; j < TW + 6; ++j) {
        const int iw = w0 + j - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iw >= 0 && iw < W) v = *reinterpret_cast<const f32x4*>(rowp + (long)iw * C);
        r[j] = v;
      }
#pragma unroll
      for (int t = 0; t < TH; ++t) {
        const int kh = ir - t;  // output row h0 + t sees this input row as its tap row kh
        if (kh < 0 || kh > 6) continue;
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (kh * 7 + kw) * C + c);
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[t][j] += r[j + kw] * wv;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < TW; ++j)
        part[(sl * c4n + c4) * P + t * TW + j] = (acc[t][j][0] + acc[t][j][1]) + (acc[t][j][2] + acc[t][j][3]);
  }
  __syncthreads();
  if (live)
    for (int p = c4; p < P; p += c4n) {  // thread c4 of a strip reduces pixels c4, c4 + c4n, ...
      const float* pp = part + sl * c4n * P + p;
      float sum = 0.f;
      for (int i = 0; i < c4n; ++i) sum += pp[i * P];
      stat[(sl * P + p) * 2] = sum / (float)C;
    }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const f32x4 d = acc[t][j] - stat[(sl * P + t * TW + j) * 2];
        part[(sl * c4n + c4) * P + t * TW + j] = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
  }
  __syncthreads();
  if (live)
    for (int p = c4; p < P; p += c4n) {
      const float* pp = part + sl * c4n * P + p;
      float sq = 0.f;
      for (int i = 0; i < c4n; ++i) sq += pp[i * P];
      stat[(sl * P + p) * 2 + 1] = 1.0f / sqrtf(sq / (float)C + eps);
    }
  __syncthreads();
  if (live) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(ln_w + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(ln_b + c);
#pragma unroll
    for (int t = 0; t < TH; ++t) {
      if (h0 + t >= H) continue;
      float* op = out + ((n * H + h0 + t) * W + w0) * C + c;
#pragma unroll
      for (int j = 0; j < TW; ++j)
        if (w0 + j < W) {
          const float mean = stat[(sl * P + t * TW + j) * 2], rstd = stat[(sl * P + t * TW + j) * 2 + 1];
          const f32x4 o = (acc[t][j] - mean) * rstd * wv + bv;
          if (SP8)
            *reinterpret_cast<sp_h8*>(reinterpret_cast<char*>(op + (long)j * C - c) + (c4 >> 1) * 32 + (c4 & 1) * 16) =
                sp8_piece_from_quad(o, c4);
          else
            *reinterpret_cast<f32x4*>(op + (long)j * C) = o;
        }
    }
  }
}

template <int TW, int TH, bool SP8, bool WL = false>
static void dwconv7_ln_rows_launch(const float* in, const float* w49, const float* bias, const float* ln_w, const float* ln_b,
                                   float* out, int N, int H, int W, int C, float eps, hipStream_t s) {
  const int nstrips = ceil_div(W, TW), nhg = ceil_div(H, TH);
  const int c4n = C / 4;
  const int S = 256 / c4n;
  const long total_strips = (long)N * nhg * nstrips;
  const unsigned grid = (unsigned)((total_strips + S - 1) / S);
  const size_t lds = (size_t)(S * TH * TW * (c4n + 2) + 4 + (WL ? 49 * C : 0)) * sizeof(float);
  if (lds > 65536) {  // (not reached by the library's own launches: C <= 96)
    static bool attr[MTGV_MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr[dev]) {
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv7_ln_rows_kernel<TW, TH, SP8, WL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr[dev] = true;
    }
  }
  hipLaunchKernelGGL((dwconv7_ln_rows_kernel<TW, TH, SP8, WL>), dim3(grid), dim3(256), lds, s, in, w49, bias, ln_w, ln_b, out, H, W, C,
                     nstrips, nhg, total_strips, S, eps);
  HIP_OK(hipGetLastError());
}

template <int PK>
static void dwconv7_ln_launch_t(const float* in, const float* w49, const float* bias, const float* ln_w, const float* ln_b,
                                float* out, int N, int H, int W, int C, float eps, hipStream_t s, int out_fmt) {
  // Row-group forms (tools/micro/dwconv_rows_probe.hip, profiles/r03_dwconv_rows_probe.txt; all bit-identical to the
  // single-row kernel): 4-pixel strips x 3 rows with the tap table in LDS while table + partial sums fit the default
  // 64 KB window (C <= 192: 121.5 vs 144-155 us at 256 x 48 x 32 x 96, 60.9 vs 66.7 at 24 x 16 x 192), else 3 rows of the
  // widest strip (34.7 vs 38.6 us at 12 x 8 x 384, 20.3 vs 24.7 at 6 x 4 x 768).  MTGV_DW_ROWS=0: single-row kernel.
  const char* const rows_env = getenv("MTGV_DW_ROWS");  // read per call: tests compare the forms inside one process
  const bool rows_on = !(rows_env && atoi(rows_env) == 0);
  // Row-streaming form (dwconv7_ln_stream_kernel.h: LDS-DMA row ring, one channel per thread, taps in registers;
  // bit-identical): one block per image, so it needs a batch that fills the CUs; measured against the row-group form
  // (tools/micro/dwconv_stream_probe.hip, profiles/r04_dwconv_stream_probe.txt) it wins where rows are long in pixels -
  // 48 x 32 x 96: 121 vs 129 us, 24 x 16 x 192: 61 vs 66 - and ties or loses at 12 x 8 x 384 / 6 x 4 x 768.
  // MTGV_DW_STREAM=0: off.
  const char* const stream_env = getenv("MTGV_DW_STREAM");
  if constexpr (PK == 0)
  if (rows_on && !(stream_env && atoi(stream_env) == 0) && N >= 128) {
#define DWSTREAM_GO(C_, G_)                                                                                                \
  (out_fmt == 1 ? dwconv7_ln_stream_launch<C_, G_, 4, 3, true>(in, w49, bias, ln_w, ln_b, out, N, H, 1, eps, s)            \
                : dwconv7_ln_stream_launch<C_, G_, 4, 3, false>(in, w49, bias, ln_w, ln_b, out, N, H, 1, eps, s))
    if (C == 96 && W == 32) { DWSTREAM_GO(96, 8); return; }
    if (C == 192 && W == 16) { DWSTREAM_GO(192, 4); return; }
#undef DWSTREAM_GO
  }
  if constexpr (PK == 0)  // the rows kernel is not keyed by PK: only the TU built without packed FP32 may instantiate it
  if (rows_on && W >= 4) {
    const int c4n_ = C / 4, S_ = 256 / c4n_;
    const bool table = (size_t)(S_ * 12 * (c4n_ + 2) + 4 + 49 * C) * sizeof(float) <= 65536;
#define DWROWS_GO(TW_, WL_)                                                                                          \
  (out_fmt == 1 ? dwconv7_ln_rows_launch<TW_, 3, true, WL_>(in, w49, bias, ln_w, ln_b, out, N, H, W, C, eps, s)        \
                : dwconv7_ln_rows_launch<TW_, 3, false, WL_>(in, w49, bias, ln_w, ln_b, out, N, H, W, C, eps, s))
    if (table) DWROWS_GO(4, true);
    else if (W >= 8) DWROWS_GO(8, false);
    else DWROWS_GO(4, false);
#undef DWROWS_GO
    return;
  }
  const int tw = W >= 8 ? 8 : (W >= 4 ? 4 : 2);
  const int nstrips = ceil_div(W, tw);
  const int c4n = C / 4;
  const int S = 256 / c4n;
  const long total_strips = (long)N * H * nstrips;
  const unsigned grid = (unsigned)((total_strips + S - 1) / S);
  const size_t lds = (size_t)(S * tw * c4n + S * tw * 2) * sizeof(float);
#define DWLN_GO(TW_, SP_) \
  hipLaunchKernelGGL((dwconv7_ln_kernel<TW_, SP_, PK>), dim3(grid), dim3(256), lds, s, in, w49, bias, ln_w, ln_b, out, H, W, C, nstrips, total_strips, S, eps)
  if (out_fmt == 1) {
    if (tw == 8) DWLN_GO(8, true);
    else if (tw == 4) DWLN_GO(4, true);
    else DWLN_GO(2, true);
  } else {
    if (tw == 8) DWLN_GO(8, false);
    else if (tw == 4) DWLN_GO(4, false);
    else DWLN_GO(2, false);
  }
#undef DWLN_GO
  HIP_OK(hipGetLastError());
}

}  // namespace mtgv
