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

}  // namespace mtgv

using namespace mtgv;
extern "C" {
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
