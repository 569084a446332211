// Detection mask -> oriented card quad on the GPU.
//
// Replaces the host geometry of InstanceSeg._orient (mtgvision/od_export.py:52-93): close the U-shaped mask,
// cv2.approxPolyN(points, 4) for four corners of a GENERAL quadrilateral (a card seen at an angle is a trapezoid, which
// extract_dewarped :95-111 then rectifies), ray test along the centroid difference for "up", corners truncated to
// integers.  Those libraries are third-party and absent; oracle/quad_ref.py restates the step operation for operation:
//   1. per mask row: leftmost / rightmost foreground pixel, count, sum of x                  (all threads, integers)
//   2. convex hull of the row extremes, monotone chain in (y, x) order, clockwise on screen   (one thread, integers)
//   3. approxPolyN: while more than 4 vertices remain, contract the hull edge whose prolonged neighbours add the
//      smallest triangle (areas in float64; all threads search the minimum, one thread contracts, first minimum wins)
//   4. v = centroid(mask pixels) - area centroid(hull)                                         (one thread, float64)
//   5. the quad edge i in 1..3 first crossed by the ray centroid + t v becomes edge (0, 1); truncation toward zero
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

// LOGITS = false: `src` is the (n, H, W) uint8 mask.  LOGITS = true: `src` is the (n, H / scale, W / scale) float mask
// logits of process_mask and a pixel is foreground when their bilinear x`scale` interpolation (align_corners = False) is
// > 0 - exactly what mask_binarize_kernel (detector.hip) writes, so both paths give the same quads - but the
// full-resolution mask never exists: only output rows / columns near positive logits are evaluated.
constexpr int QT = 512;  // threads per block
template <bool LOGITS>
__global__ __launch_bounds__(QT) void mask_quads_kernel(const void* __restrict__ src, int H, int W, int scale,
                                                       const float* __restrict__ boxes, float* __restrict__ quads,
                                                       int* __restrict__ ok, int* __restrict__ extents) {
  extern __shared__ __attribute__((aligned(16))) int sm_i[];
  int* xmin = sm_i;           // [H]
  int* xmax = xmin + H;       // [H]
  int* cnt = xmax + H;        // [H]
  int* sumx = cnt + H;        // [H]
  P2* pts = reinterpret_cast<P2*>(sumx + H);  // [2H]
  P2* lower = pts + 2 * H;                    // [2H]  (ends up holding the whole hull)
  P2* upper = lower + 2 * H;                  // [2H]
  double* vx = reinterpret_cast<double*>(upper + 2 * H);  // [2H] polygon being contracted
  double* vy = vx + 2 * H;                                // [2H]
  double* ar = vy + 2 * H;                                // [2H] area added by contracting edge (slot, next)
  int* nxt = reinterpret_cast<int*>(ar + 2 * H);          // [2H]
  int* prv = nxt + 2 * H;                                 // [2H]
  int* alive = prv + 2 * H;                               // [2H]
  int* lr_lo = alive + 2 * H;                             // [H / scale] LOGITS: first / last positive column of a logit row
  int* lr_hi = lr_lo + H;                                 //            (sized H: scale >= 1)
  __shared__ int s_h, s_flip, s_pick, s_cnt;
  __shared__ double s_area[QT / 64];
  __shared__ int s_edge[QT / 64];

  const int n = blockIdx.x, tid = threadIdx.x;
  const uint8_t* m = LOGITS ? nullptr : reinterpret_cast<const uint8_t*>(src) + (size_t)n * H * W;
  const int mh = LOGITS ? H / scale : 0, mw = LOGITS ? W / scale : 0;
  const float* Lg = LOGITS ? reinterpret_cast<const float*>(src) + (size_t)n * mh * mw : nullptr;

  // ---- 1. row extents: one wave per row, 16 pixels per lane and pass, integer reductions over the wave ----
  {
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = QT / 64;
    if (LOGITS) {  // which logit rows / columns are positive at all
      for (int r = wave; r < mh; r += NWV) {
        int lo = 0x7fffffff, hi = -1;
        for (int c = lane; c < mw; c += 64)
          if (Lg[r * mw + c] > 0.f) lo = c < lo ? c : lo, hi = c > hi ? c : hi;
#pragma unroll
        for (int mask = 32; mask > 0; mask >>= 1) {
          const int olo = __shfl_xor(lo, mask), ohi = __shfl_xor(hi, mask);
          lo = olo < lo ? olo : lo;
          hi = ohi > hi ? ohi : hi;
        }
        if (lane == 0) lr_lo[r] = lo, lr_hi[r] = hi;
      }
      __syncthreads();
    }
    const bool vec = !LOGITS && (W % 16) == 0 && ((size_t)m % 16) == 0;
    const float inv = LOGITS ? 1.0f / (float)scale : 0.f;
    for (int y = wave; y < H; y += NWV) {
      int lo = 0x7fffffff, hi = -1, c = 0, sx = 0;
      if (LOGITS) {
        // same arithmetic as mask_binarize_kernel
        float sy = inv * ((float)y + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < mh - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0;
        const float ly0 = 1.0f - ly1;
        const int clo = lr_lo[y0] < lr_lo[y1] ? lr_lo[y0] : lr_lo[y1];
        const int chi = lr_hi[y0] > lr_hi[y1] ? lr_hi[y0] : lr_hi[y1];
        if (chi >= 0) {  // a pixel whose four taps are all <= 0 is background: only columns near positive logits count
          int xa = (clo - 1) * scale, xb = (chi + 2) * scale;
          xa = xa < 0 ? 0 : xa;
          xb = xb > W ? W : xb;
          for (int x = xa + lane; x < xb; x += 64) {
            float sxf = inv * ((float)x + 0.5f) - 0.5f;
            sxf = sxf < 0.f ? 0.f : sxf;
            const int x0 = (int)sxf;
            const int x1 = x0 + (x0 < mw - 1 ? 1 : 0);
            const float lx1 = sxf - (float)x0;
            const float lx0 = 1.0f - lx1;
            const float v = ly0 * (lx0 * Lg[y0 * mw + x0] + lx1 * Lg[y0 * mw + x1]) + ly1 * (lx0 * Lg[y1 * mw + x0] + lx1 * Lg[y1 * mw + x1]);
            if (v > 0.f) {
              lo = x < lo ? x : lo;
              hi = x > hi ? x : hi;
              c += 1;
              sx += x;
            }
          }
        }
      } else {
        const uint8_t* row = m + (size_t)y * W;
        for (int x0 = lane * 16; x0 < W; x0 += 64 * 16) {
          uint8_t px[16];
          if (vec) {
            *reinterpret_cast<uint4*>(px) = *reinterpret_cast<const uint4*>(row + x0);
          } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) px[j] = x0 + j < W ? row[x0 + j] : 0;
          }
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (px[j] != 0) {
              const int x = x0 + j;
              lo = x < lo ? x : lo;
              hi = x > hi ? x : hi;
              c += 1;
              sx += x;
            }
        }
      }
#pragma unroll
      for (int mask = 32; mask > 0; mask >>= 1) {
        const int olo = __shfl_xor(lo, mask), ohi = __shfl_xor(hi, mask);
        lo = olo < lo ? olo : lo;
        hi = ohi > hi ? ohi : hi;
        c += __shfl_xor(c, mask);
        sx += __shfl_xor(sx, mask);
      }
      if (lane == 0) {
        xmin[y] = c > 0 ? lo : -1, xmax[y] = hi, cnt[y] = c, sumx[y] = sx;
        // the mask's outline as row extents (leftmost, rightmost foreground pixel; -1, -1 for an empty row): what the
        // host needs of `masks.xy` (od_export.py:152-153) - 2 H integers instead of the H x W mask
        if (extents != nullptr) {
          extents[((size_t)n * H + y) * 2] = c > 0 ? lo : -1;
          extents[((size_t)n * H + y) * 2 + 1] = c > 0 ? hi : -1;
        }
      }
    }
  }
  __syncthreads();

  // ---- 2. hull ----
  // 2a. the row extents in row order (xmin, then xmax when it differs): block-wide prefix sum instead of a serial pass
  {
    const int R = (H + QT - 1) / QT;  // consecutive rows per thread
    const int y_begin = tid * R;
    int c_local = 0;
    for (int k = 0; k < R; ++k) {
      const int y = y_begin + k;
      if (y < H && cnt[y] > 0) c_local += xmax[y] != xmin[y] ? 2 : 1;
    }
    int incl = c_local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += v;
    }
    if ((tid & 63) == 63) s_edge[tid >> 6] = incl;  // s_edge is free until the contraction
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s_edge[w];
    int pos = base + incl - c_local;
    for (int k = 0; k < R; ++k) {
      const int y = y_begin + k;
      if (y < H && cnt[y] > 0) {
        pts[pos++] = P2{xmin[y], y};
        if (xmax[y] != xmin[y]) pts[pos++] = P2{xmax[y], y};
      }
    }
    if (tid == QT - 1) s_cnt = base + incl;
    __syncthreads();
  }
  // 2b. monotone chain: the two passes run side by side on two waves; the top two points of a chain stay in
  // registers (only a pop reads LDS) and the next point is fetched one step ahead
  {
    const int np = s_cnt;
    auto chain = [&](P2* out, bool fwd) -> int {
      int n = 0;
      P2 a{0, 0}, b{0, 0};
      int i = fwd ? 0 : np - 1;
      const int step = fwd ? 1 : -1;
      P2 nextp = pts[i];
      for (int k = 0; k < np; ++k) {
        const P2 p = nextp;
        i += step;
        if (k + 1 < np) nextp = pts[i];
        while (n >= 2 && cross3(a, b, p) <= 0) {
          --n;
          b = a;
          if (n >= 2) a = out[n - 2];
        }
        out[n++] = p;
        a = b, b = p;
      }
      return n;
    };
    if (np >= 2) {
      if (tid == 0) s_h = chain(lower, true);       // nl
      if (tid == 64) s_pick = chain(upper, false);  // nu
    }
    __syncthreads();
    if (np <= 1) {
      if (tid == 0) {
        for (int i = 0; i < np; ++i) lower[i] = pts[i];
        s_h = np;
      }
    } else {
      const int nl = s_h, nu = s_pick;
      __syncthreads();
      for (int i = tid; i < nu - 1; i += QT) lower[nl - 1 + i] = upper[i];
      if (tid == 0) s_h = nl - 1 + nu - 1;
    }
  }
  __syncthreads();
  const int h = s_h;
  const P2* hull = lower;

  // ---- 3. approxPolyN: greedy edge contraction down to 4 vertices ----
  // vertex slots = hull order made clockwise on the screen (positive shoelace sum in y-down coordinates)
  if (tid == 0) {
    long long a2 = 0;
    for (int i = 0; i < h; ++i) {
      const P2 p = hull[i], q = hull[(i + 1) % h];
      a2 += (long long)p.x * q.y - (long long)q.x * p.y;
    }
    s_flip = (h >= 3 && a2 < 0) ? 1 : 0;
  }
  __syncthreads();
  const int flip = s_flip;
  for (int i = tid; i < h; i += QT) {
    const P2 p = hull[(flip && i > 0) ? h - i : i];  // [h0, h(n-1), ..., h1] when flipped
    vx[i] = (double)p.x, vy[i] = (double)p.y;
    nxt[i] = (i + 1) % h, prv[i] = (i + h - 1) % h;
    alive[i] = 1;
  }
  __syncthreads();
  auto contract = [&](int i, double& px, double& py) -> double {
    const int a = prv[i], c = nxt[i], d = nxt[c];
    const double ax = vx[a], ay = vy[a], bx = vx[i], by = vy[i], cx = vx[c], cy = vy[c], dx = vx[d], dy = vy[d];
    const double rx = bx - ax, ry = by - ay, qx = cx - dx, qy = cy - dy, ex = cx - bx, ey = cy - by;
    const double den = rx * qy - ry * qx;
    px = 0.0, py = 0.0;
    if (den == 0.0) return INFINITY;
    const double t = (ex * qy - ey * qx) / den;
    const double u = (ex * ry - ey * rx) / den;
    if (!(t > 0.0 && u > 0.0)) return INFINITY;
    px = bx + t * rx, py = by + t * ry;
    const double ux = bx - px, uy = by - py, wx = cx - px, wy = cy - py;
    return 0.5 * fabs(ux * wy - uy * wx);
  };
  // the contraction point of every candidate edge is kept beside its area (pts / upper are free after the hull), so
  // a pick costs no recomputation, and the four candidates it invalidates are recomputed by four lanes at once
  double* const apx = reinterpret_cast<double*>(pts);
  double* const apy = reinterpret_cast<double*>(upper);
  if (h >= 4) {
    for (int i = tid; i < h; i += QT) {
      double px, py;
      ar[i] = contract(i, px, py);
      apx[i] = px, apy[i] = py;
    }
    __syncthreads();
    int cnt_v = h;
    while (cnt_v > 4) {  // cnt_v is block-uniform: every thread computes the same pick
      double best = INFINITY;
      int bi = 0x7fffffff;
      for (int i = tid; i < h; i += QT)
        if (alive[i] && ar[i] < best) best = ar[i], bi = i;  // ascending i: the first minimum of this thread
#pragma unroll
      for (int mask = 32; mask > 0; mask >>= 1) {
        const double ob = __shfl_xor(best, mask);
        const int oi = __shfl_xor(bi, mask);
        if (ob < best || (ob == best && oi < bi)) best = ob, bi = oi;
      }
      if ((tid & 63) == 0) s_area[tid >> 6] = best, s_edge[tid >> 6] = bi;
      __syncthreads();
      double gb = INFINITY;
      int gi = 0x7fffffff;
      for (int w = 0; w < QT / 64; ++w)
        if (s_area[w] < gb || (s_area[w] == gb && s_edge[w] < gi)) gb = s_area[w], gi = s_edge[w];
      if (!(gb < INFINITY)) break;  // nothing left to contract (block-uniform)
      if (tid < 64) {  // one wave: LDS operations of a wave execute in order
        if (tid == 0) {
          const int c = nxt[gi];
          vx[gi] = apx[gi], vy[gi] = apy[gi];
          alive[c] = 0;
          nxt[gi] = nxt[c];
          prv[nxt[c]] = gi;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tid < 4) {
          const int p1 = prv[gi];
          const int j = tid == 0 ? prv[p1] : tid == 1 ? p1 : tid == 2 ? gi : nxt[gi];
          double qx, qy;
          const double a = contract(j, qx, qy);
          // the four slots may coincide on tiny polygons; equal slots compute equal values
          ar[j] = a, apx[j] = qx, apy[j] = qy;
        }
      }
      __syncthreads();
      cnt_v -= 1;
    }
    if (tid == 0) s_cnt = cnt_v;
  } else if (tid == 0) {
    s_cnt = h;
  }
  // ---- 4. corners, orientation ----
  // integer sums / extrema over the rows by the whole block (order-free), the floating-point rest by thread 0
  __shared__ long long s_red[QT / 64][3];
  __shared__ int s_ext[QT / 64][4];
  {
    long long r_n = 0, r_sx = 0, r_sy = 0;
    int r_ymin = 0x7fffffff, r_ymax = -1, r_xlo = 0x7fffffff, r_xhi = -1;
    for (int y = tid; y < H; y += QT) {
      const int c = cnt[y];
      if (c == 0) continue;
      r_n += c;
      r_sx += sumx[y];
      r_sy += (long long)y * c;
      r_ymin = y < r_ymin ? y : r_ymin;
      r_ymax = y > r_ymax ? y : r_ymax;
      r_xlo = xmin[y] < r_xlo ? xmin[y] : r_xlo;
      r_xhi = xmax[y] > r_xhi ? xmax[y] : r_xhi;
    }
#pragma unroll
    for (int mask = 32; mask > 0; mask >>= 1) {
      r_n += __shfl_xor(r_n, mask);
      r_sx += __shfl_xor(r_sx, mask);
      r_sy += __shfl_xor(r_sy, mask);
      const int a0 = __shfl_xor(r_ymin, mask), a1 = __shfl_xor(r_ymax, mask), a2 = __shfl_xor(r_xlo, mask), a3 = __shfl_xor(r_xhi, mask);
      r_ymin = a0 < r_ymin ? a0 : r_ymin;
      r_ymax = a1 > r_ymax ? a1 : r_ymax;
      r_xlo = a2 < r_xlo ? a2 : r_xlo;
      r_xhi = a3 > r_xhi ? a3 : r_xhi;
    }
    if ((tid & 63) == 0) {
      s_red[tid >> 6][0] = r_n, s_red[tid >> 6][1] = r_sx, s_red[tid >> 6][2] = r_sy;
      s_ext[tid >> 6][0] = r_ymin, s_ext[tid >> 6][1] = r_ymax, s_ext[tid >> 6][2] = r_xlo, s_ext[tid >> 6][3] = r_xhi;
    }
  }
  __syncthreads();
  if (tid != 0) return;

  long long ntot = 0, sx_tot = 0, sy_tot = 0;
  int ymin = 0x7fffffff, ymax = -1, xlo = 0x7fffffff, xhi = -1;
  for (int w = 0; w < QT / 64; ++w) {
    ntot += s_red[w][0], sx_tot += s_red[w][1], sy_tot += s_red[w][2];
    ymin = s_ext[w][0] < ymin ? s_ext[w][0] : ymin;
    ymax = s_ext[w][1] > ymax ? s_ext[w][1] : ymax;
    xlo = s_ext[w][2] < xlo ? s_ext[w][2] : xlo;
    xhi = s_ext[w][3] > xhi ? s_ext[w][3] : xhi;
  }
  float* q = quads + (size_t)n * 8;
  if (ntot == 0) {
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
    if (boxes != nullptr) x1 = boxes[n * 4 + 0], y1 = boxes[n * 4 + 1], x2 = boxes[n * 4 + 2], y2 = boxes[n * 4 + 3];
    q[0] = x1, q[1] = y1, q[2] = x2, q[3] = y1, q[4] = x2, q[5] = y2, q[6] = x1, q[7] = y2;
    ok[n] = 0;
    return;
  }
  double quad[4][2];
  if (h >= 4 && s_cnt == 4) {
    int i = 0;
    while (!alive[i]) ++i;  // lowest surviving slot first
    for (int k = 0; k < 4; ++k) {
      quad[k][0] = vx[i], quad[k][1] = vy[i];
      i = nxt[i];
    }
  } else {  // fewer than 4 hull vertices (point, line, triangle) or nothing left to contract: the bounding box
    const double x1 = (double)xlo, x2 = (double)xhi, y1 = (double)ymin, y2 = (double)ymax;
    quad[0][0] = x1, quad[0][1] = y1, quad[1][0] = x2, quad[1][1] = y1;
    quad[2][0] = x2, quad[2][1] = y2, quad[3][0] = x1, quad[3][1] = y2;
  }
  const double mcx = (double)sx_tot / (double)ntot, mcy = (double)sy_tot / (double)ntot;
  // area centroid (shoelace, sequential) of a polygon given by a point getter; vertex mean when degenerate
  auto centroid = [&](int np, auto get, double& ox, double& oy) {
    double a2 = 0.0, cx = 0.0, cy = 0.0;
    for (int i = 0; i < np; ++i) {
      double x0, y0, x1, y1;
      get(i, x0, y0);
      get((i + 1) % np, x1, y1);
      const double cr = x0 * y1 - x1 * y0;
      a2 += cr;
      cx += (x0 + x1) * cr;
      cy += (y0 + y1) * cr;
    }
    if (fabs(a2) < 1e-9) {
      double sx = 0.0, sy = 0.0;
      for (int i = 0; i < np; ++i) {
        double x0, y0;
        get(i, x0, y0);
        sx += x0, sy += y0;
      }
      ox = sx / (double)np, oy = sy / (double)np;
    } else {
      ox = cx / (3.0 * a2), oy = cy / (3.0 * a2);
    }
  };
  double hcx = mcx, hcy = mcy;
  if (h >= 3) {
    // the oracle takes the centroid of the hull in clockwise order
    centroid(h, [&](int i, double& x, double& y) {
      const P2 p = hull[(flip && i > 0) ? h - i : i];
      x = (double)p.x, y = (double)p.y;
    }, hcx, hcy);
  }
  double dvx = mcx - hcx, dvy = mcy - hcy;
  const double nv = sqrt(dvx * dvx + dvy * dvy);
  if (nv > 0.0) {
    dvx = dvx / nv, dvy = dvy / nv;
  } else {
    dvx = 0.0, dvy = -1.0;
  }
  double qcx, qcy;
  centroid(4, [&](int i, double& x, double& y) { x = quad[i][0], y = quad[i][1]; }, qcx, qcy);
  const double rex = qcx + dvx * 10000000.0, rey = qcy + dvy * 10000000.0;
  int idx = 0;
  for (int i = 1; i < 4; ++i) {
    const double cx = quad[i][0], cy = quad[i][1], dx = quad[(i + 1) % 4][0], dy = quad[(i + 1) % 4][1];
    const double d1 = (rex - qcx) * (cy - qcy) - (rey - qcy) * (cx - qcx);
    const double d2 = (rex - qcx) * (dy - qcy) - (rey - qcy) * (dx - qcx);
    const double d3 = (dx - cx) * (qcy - cy) - (dy - cy) * (qcx - cx);
    const double d4 = (dx - cx) * (rey - cy) - (dy - cy) * (rex - cx);
    if (d1 * d2 <= 0.0 && d3 * d4 <= 0.0) {
      idx = i;
      break;
    }
  }
  for (int i = 0; i < 4; ++i) {
    q[2 * i] = (float)trunc(quad[(idx + i) % 4][0]);
    q[2 * i + 1] = (float)trunc(quad[(idx + i) % 4][1]);
  }
  ok[n] = 1;
}

}  // namespace mtgv

using namespace mtgv;

namespace {
template <bool LOGITS>
void launch_quads(const void* src, int n, int h, int w, int scale, const float* boxes, float* quads, int* ok, int* extents,
                  hipStream_t s) {
  const size_t lds = (size_t)h * (4 * sizeof(int) + 3 * 2 * sizeof(P2) + 3 * 2 * sizeof(double) + 3 * 2 * sizeof(int) + 2 * sizeof(int));
  MTGV_CHECK(lds <= 150 * 1024, ERR_INVALID, "mask_quads: mask height %d exceeds the LDS capacity", h);
  static bool attr_done_dev[MTGV_MAX_DEVICES] = {};  // hipFuncSetAttribute is per device
  bool& attr_done = attr_done_dev[current_device()];
  if (!attr_done) {
    HIP_OK(hipFuncSetAttribute((const void*)mask_quads_kernel<LOGITS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((mask_quads_kernel<LOGITS>), dim3(n), dim3(QT), lds, s, src, h, w, scale, boxes, quads, ok, extents);
  HIP_OK(hipGetLastError());
}
}  // namespace

extern "C" {

MTGV_API int mtgv_mask_quads(const uint8_t* masks_dev, int32_t n, int32_t h, int32_t w, const float* boxes_dev, float* quads_dev,
                             int32_t* ok_dev, int32_t* extents_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(n >= 0 && h > 0 && w > 0, ERR_INVALID, "mask_quads: n=%d h=%d w=%d", n, h, w);
    if (n == 0) return;
    MTGV_CHECK(masks_dev != nullptr && quads_dev != nullptr && ok_dev != nullptr, ERR_INVALID, "mask_quads: null argument");
    launch_quads<false>(masks_dev, n, h, w, 1, boxes_dev, quads_dev, (int*)ok_dev, (int*)extents_dev, (hipStream_t)stream);
  });
}

MTGV_API int mtgv_mask_quads_logits(const float* logits_dev, int32_t n, int32_t mh, int32_t mw, int32_t scale, const float* boxes_dev,
                                    float* quads_dev, int32_t* ok_dev, int32_t* extents_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(n >= 0 && mh > 0 && mw > 0 && scale > 0, ERR_INVALID, "mask_quads_logits: n=%d mh=%d mw=%d scale=%d", n, mh, mw, scale);
    if (n == 0) return;
    MTGV_CHECK(logits_dev != nullptr && quads_dev != nullptr && ok_dev != nullptr, ERR_INVALID, "mask_quads_logits: null argument");
    launch_quads<true>(logits_dev, n, mh * scale, mw * scale, scale, boxes_dev, quads_dev, (int*)ok_dev, (int*)extents_dev,
                       (hipStream_t)stream);
  });
}
}
