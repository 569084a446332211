// Perspective de-warp of card quads: InstanceSeg.extract_dewarped, mtgvision/od_export.py:95-111
// (cv2.getPerspectiveTransform + cv2.warpPerspective, INTER_LINEAR, constant border 0).
//
//   warp_coeffs_kernel  one thread per quad: the 8x8 system of the homography that maps the
//                       output rectangle (expanded by expand_ratio) onto the source quad,
//                       Gaussian elimination with partial pivoting in float64
//   warp_kernel         one thread per (group of four) output pixel(s): projective map, 1/32-pixel quantised
//                       source position, 4-tap bilinear with 15-bit fixed-point weights
//
// The sub-pixel quantisation and fixed-point blend restate what OpenCV's INTER_LINEAR remap
// does (INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15); OpenCV is absent here, so this stage is
// checked against oracle/warp_ref.py only (parity unpinned).
// No FMA contraction anywhere in this file: the oracle evaluates the same single operations.
#include "common.h"
#include "mtgv.h"

#pragma clang fp contract(off)

namespace mtgv {

__global__ __launch_bounds__(64) void warp_coeffs_kernel(const float* __restrict__ quads, int nq, int out_h, int out_w,
                                                        double expand, double* __restrict__ coef) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= nq) return;
  const double w = (double)out_w, h = (double)out_h, e = expand;
  // dst_pts = (1 + e) * [[0,0],[w,0],[w,h],[0,h]] - 0.5 * e * [w,h]   (od_export.py:102-106), float32 like the reference
  const float dxs[4] = {0.f, (float)w, (float)w, 0.f}, dys[4] = {0.f, 0.f, (float)h, (float)h};
  double u[4], v[4], x[4], y[4];
  for (int i = 0; i < 4; ++i) {
    u[i] = (double)(float)((1.0 + e) * (double)dxs[i] - (0.5 * e) * w);
    v[i] = (double)(float)((1.0 + e) * (double)dys[i] - (0.5 * e) * h);
    x[i] = (double)quads[(q * 4 + i) * 2 + 0];
    y[i] = (double)quads[(q * 4 + i) * 2 + 1];
  }
  // unknowns c0..c7 of  X = (c0 u + c1 v + c2) / (c6 u + c7 v + 1),  Y = (c3 u + c4 v + c5) / (...)
  double A[8][9];
  for (int i = 0; i < 4; ++i) {
    double* r0 = A[i];
    double* r1 = A[i + 4];
    r0[0] = u[i], r0[1] = v[i], r0[2] = 1.0, r0[3] = 0.0, r0[4] = 0.0, r0[5] = 0.0, r0[6] = -u[i] * x[i], r0[7] = -v[i] * x[i], r0[8] = x[i];
    r1[0] = 0.0, r1[1] = 0.0, r1[2] = 0.0, r1[3] = u[i], r1[4] = v[i], r1[5] = 1.0, r1[6] = -u[i] * y[i], r1[7] = -v[i] * y[i], r1[8] = y[i];
  }
  bool singular = false;
  for (int c = 0; c < 8; ++c) {
    int piv = c;
    double best = fabs(A[c][c]);
    for (int r = c + 1; r < 8; ++r) {
      const double t = fabs(A[r][c]);
      if (t > best) best = t, piv = r;
    }
    if (best == 0.0) {
      singular = true;
      break;
    }
    if (piv != c)
      for (int k = 0; k < 9; ++k) {
        const double t = A[c][k];
        A[c][k] = A[piv][k];
        A[piv][k] = t;
      }
    for (int r = c + 1; r < 8; ++r) {
      const double f = A[r][c] / A[c][c];
      for (int k = c; k < 9; ++k) A[r][k] = A[r][k] - f * A[c][k];
    }
  }
  double sol[8];
  if (!singular) {
    for (int r = 7; r >= 0; --r) {
      double sacc = A[r][8];
      for (int k = r + 1; k < 8; ++k) sacc = sacc - A[r][k] * sol[k];
      sol[r] = sacc / A[r][r];
    }
  }
  double* o = coef + (long)q * 9;
  for (int k = 0; k < 8; ++k) o[k] = singular ? 0.0 : sol[k];
  o[8] = singular ? 0.0 : 1.0;
}

__device__ __forceinline__ int sat_short_rint(float v) {
  int r = (int)rintf(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

// PX output pixels of a row per thread (PX = 4 when out_w % 4 == 0 and the buffers are 4-byte aligned: the 12 output bytes
// leave as three dwords; PX = 1 otherwise).  Taps whose 2 x 2 window lies inside the frame are fetched as three aligned
// dwords per row (the six bytes of two neighbouring pixels start at any byte offset) instead of six byte loads; windows
// that touch the border take the byte path with constant border 0.  The arithmetic per pixel is the same in every path.
template <int PX>
__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ frames, int fh, int fw,
                                                  const double* __restrict__ coef, const int* __restrict__ frame_idx, int nq,
                                                  int out_h, int out_w, uint8_t* __restrict__ out) {
  const long gidx = (long)blockIdx.x * 256 + threadIdx.x;
  const long idx = gidx * PX;  // first output pixel of this thread
  const long total = (long)nq * out_h * out_w;
  if (idx >= total) return;
  const int x0 = (int)(idx % out_w);
  const long t = idx / out_w;
  const int y = (int)(t % out_h);
  const int q = (int)(t / out_h);
  const double* c = coef + (long)q * 9;
  const uint8_t* F = frames + (long)frame_idx[q] * fh * fw * 3;
  const long frame_px = (long)fh * fw;
  uint32_t pix[PX];  // b | g << 8 | r << 16 in memory order: three result bytes per pixel
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    const int x = x0 + j;
    const double X0 = c[0] * (double)x + c[1] * (double)y + c[2];
    const double Y0 = c[3] * (double)x + c[4] * (double)y + c[5];
    double W = c[6] * (double)x + c[7] * (double)y + c[8];
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * W));
    const int X = (int)rint(fX), Y = (int)rint(fY);
    const int sx = X >> 5, sy = Y >> 5;
    const float fx = (float)(X & 31) / 32.0f, fy = (float)(Y & 31) / 32.0f;
    // 15-bit weights, sum forced to 1 << 15 by adjusting the largest tap
    int wq[4];
    wq[0] = sat_short_rint((1.0f - fy) * (1.0f - fx) * 32768.0f);
    wq[1] = sat_short_rint((1.0f - fy) * fx * 32768.0f);
    wq[2] = sat_short_rint(fy * (1.0f - fx) * 32768.0f);
    wq[3] = sat_short_rint(fy * fx * 32768.0f);
    int big = 0;
    for (int i = 1; i < 4; ++i)
      if (wq[i] > wq[big]) big = i;
    wq[big] += 32768 - (wq[0] + wq[1] + wq[2] + wq[3]);

    int acc[3] = {0, 0, 0};
    const long p00 = (long)sy * fw + sx;  // pixel index of the top-left tap
    if (sx >= 0 && sy >= 0 && sx + 1 < fw && sy + 1 < fh && p00 + fw + 4 <= frame_px) {
      // both rows: bytes [3 p, 3 p + 6) out of the aligned 12-byte window that starts at (3 p) & ~3
#pragma unroll
      for (int row = 0; row < 2; ++row) {
        const long B = (p00 + (long)row * fw) * 3;
        const uint32_t* wp = reinterpret_cast<const uint32_t*>(F + (B & ~3L));
        const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
        const int sh = (int)(B & 3) * 8;
        const uint64_t lo = ((uint64_t)d1 << 32) | d0, hi = ((uint64_t)d2 << 32) | d1;
        const uint32_t b03 = (uint32_t)(lo >> sh), b47 = (uint32_t)(hi >> sh);  // bytes 0..3 and 4..7 from B
        const int wl = wq[row * 2], wr = wq[row * 2 + 1];
        acc[0] += wl * (int)(b03 & 0xff);
        acc[1] += wl * (int)((b03 >> 8) & 0xff);
        acc[2] += wl * (int)((b03 >> 16) & 0xff);
        acc[0] += wr * (int)(b03 >> 24);
        acc[1] += wr * (int)(b47 & 0xff);
        acc[2] += wr * (int)((b47 >> 8) & 0xff);
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int px = sx + (tap & 1), py = sy + (tap >> 1);
        if (px >= 0 && px < fw && py >= 0 && py < fh) {
          const uint8_t* p = F + ((long)py * fw + px) * 3;
          acc[0] += wq[tap] * (int)p[0];
          acc[1] += wq[tap] * (int)p[1];
          acc[2] += wq[tap] * (int)p[2];
        }
      }
    }
    uint32_t pk = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      int v = (acc[ch] + (1 << 14)) >> 15;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
#if defined(__HIP_DEVICE_COMPILE__)
      // Opaque to the optimiser on purpose: two of these next to each other are otherwise fused into v_ashr_pk_u8_i32,
      // and on gfx950 the upper half of that instruction's result is not the zero its selection pattern assumes (the
      // stale bits of the first operand showed up in every fourth pixel; tools/debug/warp_px_probe.py).
      asm volatile("" : "+v"(v));
#endif
      pk |= (uint32_t)v << (8 * ch);
    }
    pix[j] = pk;
  }
  uint8_t* o = out + idx * 3;
  if constexpr (PX == 4) {
    uint32_t* ow = reinterpret_cast<uint32_t*>(o);
    ow[0] = pix[0] | (pix[1] << 24);
    ow[1] = (pix[1] >> 8) | (pix[2] << 16);
    ow[2] = (pix[2] >> 16) | (pix[3] << 8);
  } else {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) o[ch] = (uint8_t)(pix[0] >> (8 * ch));
  }
}

// The K cards of every frame that go on to the crop stage (reference dataflow server.py:139-183):
// the K highest-confidence detections (NMS output is score-descending) or, where a frame has fewer, fixed pad boxes so
// that cards per step is constant on synthetic frames.  One thread per card: selected box, its corner quad in the order
// extract_dewarped matches to [[0,0],[w,0],[w,h],[0,h]] (od_export.py:97-103), and the card's frame index.
__global__ __launch_bounds__(256) void select_cards_kernel(const int* __restrict__ n_det, const float* __restrict__ boxes,
                                                          const float* __restrict__ pad, int F, int max_det, int K,
                                                          float* __restrict__ sel, float* __restrict__ quads,
                                                          int* __restrict__ frame_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F * K) return;
  const int f = i / K, k = i - f * K;
  const float* b = (k < n_det[f] && k < max_det) ? boxes + ((long)f * max_det + k) * 4 : pad + k * 4;
  const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
  sel[i * 4 + 0] = x1, sel[i * 4 + 1] = y1, sel[i * 4 + 2] = x2, sel[i * 4 + 3] = y2;
  if (quads != nullptr) {
    float* q = quads + (long)i * 8;
    q[0] = x1, q[1] = y1, q[2] = x2, q[3] = y1, q[4] = x2, q[5] = y2, q[6] = x1, q[7] = y2;
  }
  frame_idx[i] = f;
}

}  // namespace mtgv

using namespace mtgv;

extern "C" {
MTGV_API int mtgv_select_cards(const int32_t* n_det_dev, const float* boxes_dev, const float* pad_boxes_dev, int32_t frames,
                               int32_t max_det, int32_t k, float* sel_boxes_dev, float* quads_dev, int32_t* frame_idx_dev,
                               void* stream) {
  return guarded([&] {
    MTGV_CHECK(frames >= 0 && max_det > 0 && k > 0, ERR_INVALID, "select_cards: frames=%d max_det=%d k=%d", frames, max_det, k);
    if (frames == 0) return;
    MTGV_CHECK(n_det_dev && boxes_dev && pad_boxes_dev && sel_boxes_dev && frame_idx_dev, ERR_INVALID, "select_cards: null argument");
    hipLaunchKernelGGL(select_cards_kernel, dim3((unsigned)((frames * k + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int*)n_det_dev, boxes_dev, pad_boxes_dev, frames, max_det, k, sel_boxes_dev, quads_dev,
                       (int*)frame_idx_dev);
    HIP_OK(hipGetLastError());
  });
}

MTGV_API size_t mtgv_warp_workspace_bytes(int32_t nq) { return nq > 0 ? (size_t)nq * 9 * sizeof(double) : 0; }

MTGV_API int mtgv_warp_quads(const uint8_t* frames_dev, int32_t nf, int32_t fh, int32_t fw, const float* quads_dev,
                             const int32_t* frame_idx_dev, int32_t nq, int32_t out_h, int32_t out_w, double expand_ratio,
                             uint8_t* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  return guarded([&] {
    MTGV_CHECK(frames_dev && quads_dev && frame_idx_dev && out_dev, ERR_INVALID, "null argument");
    MTGV_CHECK(nf > 0 && fh > 0 && fw > 0 && out_h > 0 && out_w > 0 && nq >= 0, ERR_INVALID, "warp: bad geometry");
    if (nq == 0) return;
    MTGV_CHECK(workspace_dev != nullptr && workspace_bytes >= (size_t)nq * 9 * sizeof(double), ERR_INVALID,
               "warp: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double* coef = (double*)workspace_dev;
    hipLaunchKernelGGL(warp_coeffs_kernel, dim3((nq + 63) / 64), dim3(64), 0, s, quads_dev, nq, out_h, out_w, expand_ratio, coef);
    HIP_OK(hipGetLastError());
    const long total = (long)nq * out_h * out_w;
    // four pixels per thread need rows of whole groups and dword-aligned buffers (frame f starts at f * fh * fw * 3 bytes)
    static const bool px1 = getenv("MTGV_WARP_PX1") != nullptr;  // debugging aid: one pixel per thread
    if (!px1 && out_w % 4 == 0 && ((uintptr_t)out_dev & 3) == 0 && ((uintptr_t)frames_dev & 3) == 0 && ((long)fh * fw * 3) % 4 == 0)
      hipLaunchKernelGGL(warp_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, frames_dev, fh, fw, coef, frame_idx_dev,
                         nq, out_h, out_w, out_dev);
    else
      hipLaunchKernelGGL(warp_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames_dev, fh, fw, coef, frame_idx_dev,
                         nq, out_h, out_w, out_dev);
    HIP_OK(hipGetLastError());
  });
}
}
