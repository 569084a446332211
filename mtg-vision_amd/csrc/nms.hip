// Per-image class-aware NMS on decoded predictions, one workgroup per image.
// (third-party `non_max_suppression` of ultralytics 8.3.x behind CardSegmenter,
//  mtgvision/od_export.py:147-150; defaults conf 0.25, iou 0.7, max_det 300, max_wh 7680.)
//
//   1. candidates: max class score > conf; key = score bits << 32 | ~anchor  (u64)
//   2. bitonic sort of the keys in LDS, descending -> score desc, anchor asc (deterministic)
//   3. sorted boxes -> xyxy + class offset, areas (workspace in HBM, L2-resident)
//   4. greedy sweep: for every surviving box, each wave tests 64 later boxes per step and
//      publishes the result with one __ballot into the suppression bitmask (no atomics)
//
// All box arithmetic uses explicitly rounded single operations (no FMA contraction), so the
// kept indices are bit-identical to the float32 CPU oracle (oracle/detector_ref.py nms_single).
#include "nms.h"

namespace mtgv {

static constexpr int NMS_THREADS = 1024;

__device__ __forceinline__ float box_iou_rn(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1,
                                            float bx2, float by2, float barea) {
  const float iw = fmaxf(0.f, __fsub_rn(fminf(ax2, bx2), fmaxf(ax1, bx1)));
  const float ih = fmaxf(0.f, __fsub_rn(fminf(ay2, by2), fmaxf(ay1, by1)));
  const float inter = __fmul_rn(iw, ih);
  return __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter));
}

// ws layout per image (floats): obox[cap][4], area[cap], then ints: sidx[cap], scls[cap]
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ pred, int nc, int nm, int na, int cap,
                                                         float conf_thres, float iou_thres, int max_det, float max_wh,
                                                         int* __restrict__ n_det, float* __restrict__ boxes,
                                                         float* __restrict__ conf_out, int* __restrict__ cls_out,
                                                         int* __restrict__ keep_idx, float* __restrict__ coef_out,
                                                         int* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // [cap] (cap = pow2 >= na)
  __shared__ int s_count;
  __shared__ int s_nkeep;
  __shared__ int s_keep[1024];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int no = 4 + nc + nm;
  const float* P = pred + (long)img * no * na;

  float* obox = reinterpret_cast<float*>(ws) + (long)img * cap * 7;
  float* area = obox + (long)cap * 4;
  int* sidx = reinterpret_cast<int*>(area + cap);
  int* scls = sidx + cap;

  if (tid == 0) s_count = 0, s_nkeep = 0;
  for (int i = tid; i < cap; i += NMS_THREADS) keys[i] = 0ull;
  __syncthreads();

  // 1. candidates
  for (int a = tid; a < na; a += NMS_THREADS) {
    float best = P[(long)4 * na + a];
    for (int c = 1; c < nc; ++c) {
      const float v = P[(long)(4 + c) * na + a];
      if (v > best) best = v;
    }
    if (best > conf_thres) {
      const int slot = atomicAdd(&s_count, 1);
      keys[slot] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~(unsigned)a);
    }
  }
  __syncthreads();
  const int count = s_count;
  int n2 = 1;
  while (n2 < count) n2 <<= 1;

  // 2. bitonic sort, descending
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += NMS_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) keys[i] = b, keys[l] = a;
        }
      }
      __syncthreads();
    }
  }

  // 3. sorted boxes
  for (int i = tid; i < count; i += NMS_THREADS) {
    const int a = (int)(~(unsigned)(keys[i] & 0xffffffffull));
    const float x = P[a], y = P[(long)na + a], w = P[(long)2 * na + a], h = P[(long)3 * na + a];
    float best = P[(long)4 * na + a];
    int cls = 0;
    for (int c = 1; c < nc; ++c) {
      const float v = P[(long)(4 + c) * na + a];
      if (v > best) best = v, cls = c;
    }
    const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);
    const float off = __fmul_rn((float)cls, max_wh);
    const float x1 = __fadd_rn(__fsub_rn(x, hw), off), y1 = __fadd_rn(__fsub_rn(y, hh), off);
    const float x2 = __fadd_rn(__fadd_rn(x, hw), off), y2 = __fadd_rn(__fadd_rn(y, hh), off);
    obox[(long)i * 4 + 0] = x1, obox[(long)i * 4 + 1] = y1, obox[(long)i * 4 + 2] = x2, obox[(long)i * 4 + 3] = y2;
    area[i] = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    sidx[i] = a;
    scls[i] = cls;
  }
  __syncthreads();  // global writes by this block are visible to it after the barrier

  // 4. greedy sweep; the key buffer is reused as the suppression bitmask (64 boxes per word)
  unsigned long long* supp = keys;
  const int nwords = (count + 63) >> 6;
  for (int i = tid; i < nwords; i += NMS_THREADS) supp[i] = 0ull;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nwaves = NMS_THREADS >> 6;
  int i = 0;
  while (true) {
    while (i < count && ((supp[i >> 6] >> (i & 63)) & 1ull)) ++i;  // same scan in every thread
    if (i >= count) break;
    if (tid == 0) s_keep[s_nkeep] = i;
    const int nk = s_nkeep + 1;  // read before the barrier below, written after it
    __syncthreads();
    if (tid == 0) s_nkeep = nk;
    if (nk >= max_det) {
      __syncthreads();
      break;
    }
    const float ax1 = obox[(long)i * 4], ay1 = obox[(long)i * 4 + 1], ax2 = obox[(long)i * 4 + 2], ay2 = obox[(long)i * 4 + 3];
    const float aarea = area[i];
    for (int wd = (i >> 6) + wave; wd < nwords; wd += nwaves) {
      const int j = (wd << 6) + lane;
      bool s = false;
      if (j > i && j < count) {
        const float iou = box_iou_rn(ax1, ay1, ax2, ay2, aarea, obox[(long)j * 4], obox[(long)j * 4 + 1], obox[(long)j * 4 + 2],
                                     obox[(long)j * 4 + 3], area[j]);
        s = iou > iou_thres;
      }
      const unsigned long long m = __ballot(s);
      if (lane == 0 && m) supp[wd] |= m;  // one writer per word per step
    }
    ++i;
    __syncthreads();
  }
  __syncthreads();

  // outputs, score-descending
  const int nkeep = s_nkeep;
  if (tid == 0) n_det[img] = nkeep;
  for (int t = tid; t < nkeep; t += NMS_THREADS) {
    const int k = s_keep[t];
    const int a = sidx[k], cls = scls[k];
    const float off = __fmul_rn((float)cls, max_wh);
    const long o = (long)img * max_det + t;
    // un-offset boxes are recomputed from the prediction so they carry no offset rounding
    const float x = P[a], y = P[(long)na + a], w = P[(long)2 * na + a], h = P[(long)3 * na + a];
    const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);
    boxes[o * 4 + 0] = __fsub_rn(x, hw);
    boxes[o * 4 + 1] = __fsub_rn(y, hh);
    boxes[o * 4 + 2] = __fadd_rn(x, hw);
    boxes[o * 4 + 3] = __fadd_rn(y, hh);
    (void)off;
    conf_out[o] = P[(long)(4 + cls) * na + a];
    cls_out[o] = cls;
    keep_idx[o] = a;
  }
  if (coef_out != nullptr) {
    // mask coefficients of the kept detections, (max_det, nm) per image, zero beyond n_det
    for (int t = tid; t < max_det * nm; t += NMS_THREADS) {
      const int d = t / nm, c = t - d * nm;
      float v = 0.f;
      if (d < nkeep) v = P[(long)(4 + nc + c) * na + sidx[s_keep[d]]];
      coef_out[((long)img * max_det + d) * nm + c] = v;
    }
  }
}

static int pow2_ge(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

size_t nms_workspace_bytes(int n, int na) { return (size_t)n * pow2_ge(na) * 7 * sizeof(float); }

void nms_launch(const float* pred, int n, int nc, int nm, int na, float conf, float iou, int max_det, float max_wh, int* n_det,
                float* boxes, float* conf_out, int* cls_out, int* keep_idx, float* coef_out, int* ws, size_t ws_bytes,
                hipStream_t s) {
  MTGV_CHECK(n > 0 && nc > 0 && nm >= 0 && na > 0, ERR_INVALID, "nms: n=%d nc=%d nm=%d na=%d", n, nc, nm, na);
  MTGV_CHECK(max_det > 0 && max_det <= 1024, ERR_INVALID, "nms: max_det=%d outside [1,1024]", max_det);
  const int cap = pow2_ge(na);
  const size_t lds = (size_t)cap * sizeof(unsigned long long);
  MTGV_CHECK(lds <= 150 * 1024, ERR_INVALID, "nms: %d anchors exceed the LDS sort capacity", na);
  MTGV_CHECK(ws != nullptr && ws_bytes >= nms_workspace_bytes(n, na), ERR_INVALID, "nms: workspace too small");
  static bool attr_done = false;
  if (!attr_done) {
    HIP_OK(hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(n), dim3(NMS_THREADS), lds, s, pred, nc, nm, na, cap, conf, iou, max_det, max_wh, n_det,
                     boxes, conf_out, cls_out, keep_idx, coef_out, ws);
  HIP_OK(hipGetLastError());
}

}  // namespace mtgv

extern "C" {
MTGV_API size_t mtgv_nms_workspace_bytes(int32_t n, int32_t na) {
  if (n <= 0 || na <= 0) return 0;
  return mtgv::nms_workspace_bytes(n, na);
}
MTGV_API int mtgv_nms(const float* pred_dev, int32_t n, int32_t nc, int32_t nm, int32_t na, float conf, float iou,
                      int32_t max_det, float max_wh, int32_t* n_det_dev, float* boxes_dev, float* conf_dev, int32_t* cls_dev,
                      int32_t* keep_idx_dev, int32_t* workspace_dev, size_t workspace_bytes, void* stream) {
  return mtgv::guarded([&] {
    MTGV_CHECK(pred_dev && n_det_dev && boxes_dev && conf_dev && cls_dev && keep_idx_dev, mtgv::ERR_INVALID, "null argument");
    mtgv::nms_launch(pred_dev, n, nc, nm, na, conf, iou, max_det, max_wh, n_det_dev, boxes_dev, conf_dev, cls_dev, keep_idx_dev,
                     nullptr, workspace_dev, workspace_bytes, (hipStream_t)stream);
  });
}
}
