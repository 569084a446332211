// Bank-side preprocessing: SyntheticBgFgMtgImages.make_cropped (mtgvision/encoder_datasets.py:733-753) =
// strip a border of ceil(max(0.02 H, 0.02 W)) pixels, then cv2.resize(..., INTER_AREA) to the encoder
// input size, clip to [0,1] (util/image.py:322-346).  A ragged batch (every image its own size) is one launch:
// blockIdx.y picks the image, a per-image table gives its byte offset and size.
//
// INTER_AREA is restated as the exact area integral: an output pixel is the coverage-weighted mean of the
// source pixels under its footprint [o*s, (o+1)*s) in x and y.  cv2 is absent here: parity unpinned,
// checked against oracle/resize_ref.py.
#include "common.h"
#include "mtgv.h"

namespace mtgv {

__global__ __launch_bounds__(256) void make_cropped_kernel(const uint8_t* __restrict__ images, const long* __restrict__ offsets,
                                                          const int* __restrict__ hw, int out_h, int out_w,
                                                          float* __restrict__ out) {
  const int img = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= out_h * out_w) return;
  const int oy = idx / out_w, ox = idx % out_w;
  const int H = hw[img * 2], W = hw[img * 2 + 1];
  // border_width = ceil(max(0.02*H, 0.02*W)) evaluated like the reference (float64 product, then ceil)
  const double bm = fmax(0.02 * (double)H, 0.02 * (double)W);
  const int bw = (int)ceil(bm);
  const int ch = H - 2 * bw, cw = W - 2 * bw;  // cropped size
  float* o = out + ((long)img * out_h * out_w + idx) * 3;
  if (ch <= 0 || cw <= 0) {
    o[0] = o[1] = o[2] = 0.f;
    return;
  }
  const uint8_t* src = images + offsets[img];
  const double sy = (double)ch / out_h, sx = (double)cw / out_w;
  const double y0 = oy * sy, y1 = (oy + 1) * sy, x0 = ox * sx, x1 = (ox + 1) * sx;
  const int iy0 = (int)floor(y0), ix0 = (int)floor(x0);
  int iy1 = (int)ceil(y1), ix1 = (int)ceil(x1);
  iy1 = iy1 > ch ? ch : iy1;
  ix1 = ix1 > cw ? cw : ix1;
  double acc[3] = {0.0, 0.0, 0.0};
  for (int y = iy0; y < iy1; ++y) {
    const double wy = fmin((double)(y + 1), y1) - fmax((double)y, y0);
    const uint8_t* row = src + ((long)(y + bw) * W + bw) * 3;
    for (int x = ix0; x < ix1; ++x) {
      const double w = wy * (fmin((double)(x + 1), x1) - fmax((double)x, x0));
      const uint8_t* p = row + (long)x * 3;
      acc[0] += w * (double)p[0];
      acc[1] += w * (double)p[1];
      acc[2] += w * (double)p[2];
    }
  }
  const double inv = 1.0 / (sx * sy * 255.0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (float)(acc[c] * inv);
    o[c] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
  }
}

// ultralytics LetterBox ahead of the detector (behind CardSegmenter.__call__, od_export.py:147-150): the frame scaled to
// fit size x size (bilinear; cv2.resize(INTER_LINEAR) upstream - absent here, parity unpinned), centred, the border
// filled with pad_value.  Thread = one output pixel.  The resample is the align_corners = False form PyTorch's
// interpolate uses, in float32 in this order: src = scale * (dst + 0.5) - 0.5 clamped at 0, h0 * (w0 * v00 + w1 * v01) +
// h1 * (w0 * v10 + w1 * v11), rounded to nearest even, clamped to [0, 255] (oracle/resize_ref.py: letterbox).  A frame
// that already has the target size is copied exactly (all weights 0 or 1).
__global__ __launch_bounds__(256) void letterbox_u8_kernel(const uint8_t* __restrict__ src, int h, int w, uint8_t* __restrict__ dst, int size,
                                                          int nh, int nw, int top, int left, int pad_value) {
#pragma clang fp contract(off)
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= size * size) return;
  const int y = idx / size, x = idx - y * size;
  uint8_t* o = dst + (long)idx * 3;
  const int ry = y - top, rx = x - left;
  if (ry < 0 || ry >= nh || rx < 0 || rx >= nw) {
    o[0] = o[1] = o[2] = (uint8_t)pad_value;
    return;
  }
  const float sch = (float)h / (float)nh, scw = (float)w / (float)nw;
  float sy = sch * ((float)ry + 0.5f) - 0.5f, sx = scw * ((float)rx + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  int y0 = (int)sy, x0 = (int)sx;
  y0 = y0 > h - 1 ? h - 1 : y0;
  x0 = x0 > w - 1 ? w - 1 : x0;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
  const uint8_t *p00 = src + ((long)y0 * w + x0) * 3, *p01 = src + ((long)y0 * w + x1) * 3, *p10 = src + ((long)y1 * w + x0) * 3,
                *p11 = src + ((long)y1 * w + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = ly0 * (lx0 * (float)p00[c] + lx1 * (float)p01[c]) + ly1 * (lx0 * (float)p10[c] + lx1 * (float)p11[c]);
    float r = rintf(v);
    r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
    o[c] = (uint8_t)r;
  }
}

}  // namespace mtgv

using namespace mtgv;
extern "C" {
MTGV_API int mtgv_letterbox_u8(const uint8_t* src_dev, int32_t h, int32_t w, uint8_t* dst_dev, int32_t size, int32_t nh, int32_t nw,
                               int32_t top, int32_t left, int32_t pad_value, void* stream) {
  return guarded([&] {
    MTGV_CHECK(src_dev && dst_dev, ERR_INVALID, "null argument");
    MTGV_CHECK(h > 0 && w > 0 && size > 0 && nh > 0 && nw > 0 && top >= 0 && left >= 0 && top + nh <= size && left + nw <= size &&
                   pad_value >= 0 && pad_value <= 255,
               ERR_INVALID, "letterbox: %dx%d -> %dx%d at (%d, %d) of %d", h, w, nh, nw, top, left, size);
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3((size * size + 255) / 256), dim3(256), 0, (hipStream_t)stream, src_dev, h, w, dst_dev, size,
                       nh, nw, top, left, pad_value);
    HIP_OK(hipGetLastError());
  });
}
MTGV_API int mtgv_make_cropped(const uint8_t* images_dev, const int64_t* offsets_dev, const int32_t* hw_dev, int32_t n,
                               int32_t out_h, int32_t out_w, float* out_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(images_dev && offsets_dev && hw_dev && out_dev, ERR_INVALID, "null argument");
    MTGV_CHECK(n >= 0 && out_h > 0 && out_w > 0, ERR_INVALID, "make_cropped: bad geometry");
    if (n == 0) return;
    hipLaunchKernelGGL(make_cropped_kernel, dim3((out_h * out_w + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, images_dev,
                       (const long*)offsets_dev, hw_dev, out_h, out_w, out_dev);
    HIP_OK(hipGetLastError());
  });
}
}
