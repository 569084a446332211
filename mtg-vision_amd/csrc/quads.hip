// Detection mask -> oriented card quad on the GPU.
//
// Replaces the host geometry of InstanceSeg._orient (mtgvision/od_export.py:52-93: shapely buffer(+/-) to close the
// U-shaped mask, cv2.approxPolyN for four corners, centroid difference for "up").  Those libraries are third-party and
// absent; this is the build's own statement of the step (oracle/quad_ref.py restates it operation for operation):
//   1. per mask row: leftmost / rightmost foreground pixel, count, sum of x                  (all threads, integers)
//   2. convex hull of the row extremes, monotone chain in (y, x) order                         (one thread, integer cross products)
//   3. minimum-area rectangle over the hull edges                                              (all threads, float64, first minimum wins)
//   4. up = centroid(mask pixels) - area centroid(hull); the rectangle edge furthest along it becomes edge (0, 1),
//      corners clockwise (y down)                                                              (one thread, float64)
// One block per mask; every float64 expression is evaluated in the oracle's order with contraction off, so the quads
// are bit-identical to the oracle's.
#include "common.h"
#include "mtgv.h"

#include <math.h>

#pragma clang fp contract(off)

namespace mtgv {

struct P2 {
  int x, y;
};

__device__ __forceinline__ long long cross3(const P2 o, const P2 a, const P2 b) {
  return (long long)(a.x - o.x) * (long long)(b.y - o.y) - (long long)(a.y - o.y) * (long long)(b.x - o.x);
}

__global__ __launch_bounds__(256) void mask_quads_kernel(const uint8_t* __restrict__ masks, int H, int W,
                                                        const float* __restrict__ boxes, float* __restrict__ quads,
                                                        int* __restrict__ ok) {
  extern __shared__ __attribute__((aligned(16))) int sm_i[];
  int* xmin = sm_i;           // [H]
  int* xmax = xmin + H;       // [H]
  int* cnt = xmax + H;        // [H]
  int* sumx = cnt + H;        // [H]
  P2* pts = reinterpret_cast<P2*>(sumx + H);  // [2H]
  P2* lower = pts + 2 * H;                    // [2H]  (ends up holding the whole hull)
  P2* upper = lower + 2 * H;                  // [2H]
  __shared__ int s_h;
  __shared__ double s_area[256];
  __shared__ int s_edge[256];

  const int n = blockIdx.x, tid = threadIdx.x;
  const uint8_t* m = masks + (size_t)n * H * W;

  // ---- 1. row extents ----
  for (int y = tid; y < H; y += 256) {
    const uint8_t* row = m + (size_t)y * W;
    int lo = -1, hi = -1, c = 0, sx = 0;
    for (int x = 0; x < W; ++x) {
      if (row[x] != 0) {
        if (lo < 0) lo = x;
        hi = x;
        c += 1;
        sx += x;
      }
    }
    xmin[y] = lo, xmax[y] = hi, cnt[y] = c, sumx[y] = sx;
  }
  __syncthreads();

  // ---- 2. hull (thread 0) ----
  if (tid == 0) {
    int np = 0;
    for (int y = 0; y < H; ++y) {
      if (cnt[y] == 0) continue;
      pts[np++] = P2{xmin[y], y};
      if (xmax[y] != xmin[y]) pts[np++] = P2{xmax[y], y};
    }
    int h = 0;
    if (np <= 1) {
      for (int i = 0; i < np; ++i) lower[i] = pts[i];
      h = np;
    } else {
      int nl = 0, nu = 0;
      for (int i = 0; i < np; ++i) {
        while (nl >= 2 && cross3(lower[nl - 2], lower[nl - 1], pts[i]) <= 0) --nl;
        lower[nl++] = pts[i];
      }
      for (int i = np - 1; i >= 0; --i) {
        while (nu >= 2 && cross3(upper[nu - 2], upper[nu - 1], pts[i]) <= 0) --nu;
        upper[nu++] = pts[i];
      }
      h = nl - 1;
      for (int i = 0; i < nu - 1; ++i) lower[h++] = upper[i];
    }
    s_h = h;
  }
  __syncthreads();
  const int h = s_h;
  const P2* hull = lower;

  // ---- 3. minimum-area rectangle over the hull edges ----
  double my_area = INFINITY;
  int my_edge = -1;
  if (h >= 3) {
    for (int i = tid; i < h; i += 256) {
      const P2 p0 = hull[i], p1 = hull[(i + 1) % h];
      const double ex = (double)(p1.x - p0.x), ey = (double)(p1.y - p0.y);
      const double nn = sqrt(ex * ex + ey * ey);
      if (nn == 0.0) continue;
      const double ux = ex / nn, uy = ey / nn;
      double a0 = INFINITY, b0 = INFINITY, a1 = -INFINITY, b1 = -INFINITY;
      for (int k = 0; k < h; ++k) {
        const double px = (double)hull[k].x, py = (double)hull[k].y;
        const double a = px * ux + py * uy;
        const double b = py * ux - px * uy;
        a0 = a < a0 ? a : a0, a1 = a > a1 ? a : a1;
        b0 = b < b0 ? b : b0, b1 = b > b1 ? b : b1;
      }
      const double area = (a1 - a0) * (b1 - b0);
      if (area < my_area) my_area = area, my_edge = i;  // a thread walks its edges in ascending order
    }
  }
  s_area[tid] = my_area;
  s_edge[tid] = my_edge;
  __syncthreads();
  if (tid != 0) return;

  // ---- 4. corners, orientation (thread 0) ----
  long long ntot = 0, sx_tot = 0, sy_tot = 0;
  int ymin = -1, ymax = -1, xlo = 0x7fffffff, xhi = -1;
  for (int y = 0; y < H; ++y) {
    if (cnt[y] == 0) continue;
    ntot += cnt[y];
    sx_tot += sumx[y];
    sy_tot += (long long)y * cnt[y];
    if (ymin < 0) ymin = y;
    ymax = y;
    xlo = xmin[y] < xlo ? xmin[y] : xlo;
    xhi = xmax[y] > xhi ? xmax[y] : xhi;
  }
  float* q = quads + (size_t)n * 8;
  if (ntot == 0) {
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
    if (boxes != nullptr) x1 = boxes[n * 4 + 0], y1 = boxes[n * 4 + 1], x2 = boxes[n * 4 + 2], y2 = boxes[n * 4 + 3];
    q[0] = x1, q[1] = y1, q[2] = x2, q[3] = y1, q[4] = x2, q[5] = y2, q[6] = x1, q[7] = y2;
    ok[n] = 0;
    return;
  }
  double best_area = INFINITY;
  int best_edge = -1;
  for (int t = 0; t < 256; ++t)
    if (s_edge[t] >= 0 && (s_area[t] < best_area || (s_area[t] == best_area && s_edge[t] < best_edge)))
      best_area = s_area[t], best_edge = s_edge[t];
  double rect[4][2];
  if (best_edge >= 0) {
    const P2 p0 = hull[best_edge], p1 = hull[(best_edge + 1) % h];
    const double ex = (double)(p1.x - p0.x), ey = (double)(p1.y - p0.y);
    const double nn = sqrt(ex * ex + ey * ey);
    const double ux = ex / nn, uy = ey / nn;
    double a0 = INFINITY, b0 = INFINITY, a1 = -INFINITY, b1 = -INFINITY;
    for (int k = 0; k < h; ++k) {
      const double px = (double)hull[k].x, py = (double)hull[k].y;
      const double a = px * ux + py * uy;
      const double b = py * ux - px * uy;
      a0 = a < a0 ? a : a0, a1 = a > a1 ? a : a1;
      b0 = b < b0 ? b : b0, b1 = b > b1 ? b : b1;
    }
    const double vx = -uy, vy = ux;
    rect[0][0] = ux * a0 + vx * b0, rect[0][1] = uy * a0 + vy * b0;
    rect[1][0] = ux * a1 + vx * b0, rect[1][1] = uy * a1 + vy * b0;
    rect[2][0] = ux * a1 + vx * b1, rect[2][1] = uy * a1 + vy * b1;
    rect[3][0] = ux * a0 + vx * b1, rect[3][1] = uy * a0 + vy * b1;
  } else {  // a point or a straight run of pixels: its bounding box
    const double x1 = (double)xlo, x2 = (double)xhi, y1 = (double)ymin, y2 = (double)ymax;
    rect[0][0] = x1, rect[0][1] = y1, rect[1][0] = x2, rect[1][1] = y1;
    rect[2][0] = x2, rect[2][1] = y2, rect[3][0] = x1, rect[3][1] = y2;
  }
  const double mcx = (double)sx_tot / (double)ntot, mcy = (double)sy_tot / (double)ntot;
  double hcx = mcx, hcy = mcy;
  if (h >= 3) {
    double a2 = 0.0, cx = 0.0, cy = 0.0;
    for (int i = 0; i < h; ++i) {
      const double x0 = (double)hull[i].x, y0 = (double)hull[i].y;
      const double x1 = (double)hull[(i + 1) % h].x, y1 = (double)hull[(i + 1) % h].y;
      const double cr = x0 * y1 - x1 * y0;
      a2 += cr;
      cx += (x0 + x1) * cr;
      cy += (y0 + y1) * cr;
    }
    if (fabs(a2) < 1e-9) {
      double sx = 0.0, sy = 0.0;
      for (int i = 0; i < h; ++i) sx += (double)hull[i].x, sy += (double)hull[i].y;
      hcx = sx / (double)h, hcy = sy / (double)h;
    } else {
      hcx = cx / (3.0 * a2), hcy = cy / (3.0 * a2);
    }
  }
  double vx = mcx - hcx, vy = mcy - hcy;
  const double nv = sqrt(vx * vx + vy * vy);
  if (nv > 0.0) {
    vx = vx / nv, vy = vy / nv;
  } else {
    vx = 0.0, vy = -1.0;
  }
  const double ccx = (rect[0][0] + rect[1][0] + rect[2][0] + rect[3][0]) / 4.0;
  const double ccy = (rect[0][1] + rect[1][1] + rect[2][1] + rect[3][1]) / 4.0;
  int idx = 0;
  double best = -INFINITY;
  for (int i = 0; i < 4; ++i) {
    const double mx = (rect[i][0] + rect[(i + 1) % 4][0]) / 2.0 - ccx;
    const double my = (rect[i][1] + rect[(i + 1) % 4][1]) / 2.0 - ccy;
    const double d = mx * vx + my * vy;
    if (d > best) best = d, idx = i;
  }
  double o[4][2];
  for (int i = 0; i < 4; ++i) o[i][0] = rect[(idx + i) % 4][0], o[i][1] = rect[(idx + i) % 4][1];
  const double e0x = o[1][0] - o[0][0], e0y = o[1][1] - o[0][1];
  const double e1x = o[2][0] - o[1][0], e1y = o[2][1] - o[1][1];
  if (e0x * e1y - e0y * e1x < 0.0) {
    double t;
    t = o[0][0], o[0][0] = o[1][0], o[1][0] = t;
    t = o[0][1], o[0][1] = o[1][1], o[1][1] = t;
    t = o[2][0], o[2][0] = o[3][0], o[3][0] = t;
    t = o[2][1], o[2][1] = o[3][1], o[3][1] = t;
  }
  for (int i = 0; i < 4; ++i) q[2 * i] = (float)o[i][0], q[2 * i + 1] = (float)o[i][1];
  ok[n] = 1;
}

}  // namespace mtgv

using namespace mtgv;

extern "C" {

MTGV_API int mtgv_mask_quads(const uint8_t* masks_dev, int32_t n, int32_t h, int32_t w, const float* boxes_dev, float* quads_dev,
                             int32_t* ok_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(n >= 0 && h > 0 && w > 0, ERR_INVALID, "mask_quads: n=%d h=%d w=%d", n, h, w);
    if (n == 0) return;
    MTGV_CHECK(masks_dev != nullptr && quads_dev != nullptr && ok_dev != nullptr, ERR_INVALID, "mask_quads: null argument");
    const size_t lds = (size_t)h * (4 * sizeof(int) + 3 * 2 * sizeof(P2));
    MTGV_CHECK(lds <= 150 * 1024, ERR_INVALID, "mask_quads: mask height %d exceeds the LDS capacity", h);
    static bool attr_done = false;
    if (!attr_done) {
      HIP_OK(hipFuncSetAttribute((const void*)mask_quads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_done = true;
    }
    hipLaunchKernelGGL(mask_quads_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, masks_dev, h, w, boxes_dev, quads_dev,
                       (int*)ok_dev);
    HIP_OK(hipGetLastError());
  });
}
}
