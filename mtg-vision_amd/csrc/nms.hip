// Per-image class-aware NMS on decoded predictions, one workgroup per image.
// (third-party `non_max_suppression` of ultralytics 8.3.x behind CardSegmenter,
//  mtgvision/od_export.py:147-150; defaults conf 0.25, iou 0.7, max_det 300, max_wh 7680.)
//
//   1. candidates: max class score > conf; key = score bits << 32 | ~anchor  (u64)
//   2. bitonic sort of the keys in LDS, descending -> score desc, anchor asc (deterministic)
//   3. sorted boxes -> xyxy + class offset, areas (workspace in HBM, L2-resident)
//   4. greedy sweep, 64 sorted boxes at a time: one wave resolves the word in order (readlane broadcast + __ballot),
//      then all waves apply the word's kept boxes to the later words, one __ballot per word into the suppression
//      bitmask (no atomics)
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

  // 4. greedy sweep, one 64-box word of the sorted list at a time; the key buffer is reused as the suppression bitmask.
  //    (a) wave 0 resolves the word in order: the lowest surviving lane is kept (v_readlane broadcasts its box) and
  //        knocks out the later lanes it overlaps, published with one __ballot per keep - no block barrier;
  //    (b) every wave then applies the word's kept boxes (parked in LDS) to its share of the later words.
  //    Same keep list as the box-at-a-time sweep: a box is kept iff no earlier kept box overlaps it beyond the
  //    threshold; two block barriers per word instead of two per kept box.
  unsigned long long* supp = keys;
  __shared__ float kbox[64 * 5];  // x1 y1 x2 y2 area of the word being resolved
  __shared__ unsigned long long s_kmask;
  const int nwords = (count + 63) >> 6;
  for (int i = tid; i < nwords; i += NMS_THREADS) supp[i] = 0ull;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nwaves = NMS_THREADS >> 6;
  for (int w0 = 0; w0 < nwords; ++w0) {
    if (wave == 0) {
      const int j = (w0 << 6) + lane;
      const bool valid = j < count;
      const float bx1 = valid ? obox[(long)j * 4] : 0.f, by1 = valid ? obox[(long)j * 4 + 1] : 0.f;
      const float bx2 = valid ? obox[(long)j * 4 + 2] : 0.f, by2 = valid ? obox[(long)j * 4 + 3] : 0.f;
      const float barea = valid ? area[j] : 0.f;
      kbox[lane * 5 + 0] = bx1, kbox[lane * 5 + 1] = by1, kbox[lane * 5 + 2] = bx2, kbox[lane * 5 + 3] = by2, kbox[lane * 5 + 4] = barea;
      unsigned long long alive = __ballot(valid) & ~supp[w0];
      unsigned long long kept = 0ull;
      int nk = s_nkeep;
      while (alive != 0ull && nk < max_det) {
        const int k = __builtin_ctzll(alive);  // wave-uniform: the earliest surviving box of the word
        if (lane == 0) s_keep[nk] = (w0 << 6) + k;
        nk += 1;
        kept |= 1ull << k;
        alive &= ~(1ull << k);
        if (nk >= max_det) break;
        const float ax1 = __shfl(bx1, k), ay1 = __shfl(by1, k), ax2 = __shfl(bx2, k), ay2 = __shfl(by2, k), aarea = __shfl(barea, k);
        const bool s = lane > k && ((alive >> lane) & 1ull) && box_iou_rn(ax1, ay1, ax2, ay2, aarea, bx1, by1, bx2, by2, barea) > iou_thres;
        alive &= ~__ballot(s);
      }
      if (lane == 0) s_nkeep = nk, s_kmask = kept;
    }
    __syncthreads();
    const unsigned long long kept = s_kmask;
    if (s_nkeep >= max_det) break;  // block-uniform
    if (kept != 0ull) {
      for (int wd = w0 + 1 + wave; wd < nwords; wd += nwaves) {
        const int j = (wd << 6) + lane;
        bool s = false;
        if (j < count) {
          const float bx1 = obox[(long)j * 4], by1 = obox[(long)j * 4 + 1], bx2 = obox[(long)j * 4 + 2], by2 = obox[(long)j * 4 + 3];
          const float barea = area[j];
          unsigned long long m = kept;
          while (m != 0ull && !s) {  // any kept box of the word suffices; the order of the tests does not matter
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            s = box_iou_rn(kbox[k * 5], kbox[k * 5 + 1], kbox[k * 5 + 2], kbox[k * 5 + 3], kbox[k * 5 + 4], bx1, by1, bx2, by2, barea) > iou_thres;
          }
        }
        const unsigned long long mk = __ballot(s);
        if (lane == 0 && mk) supp[wd] |= mk;  // one writer per word per step
      }
    }
    __syncthreads();  // supp of the next word is complete; kbox and s_kmask may be rewritten
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
  // slots beyond the kept detections: zeros, so callers may hand in uninitialised output tensors
  for (int t = nkeep + tid; t < max_det; t += NMS_THREADS) {
    const long o = (long)img * max_det + t;
    boxes[o * 4 + 0] = 0.f, boxes[o * 4 + 1] = 0.f, boxes[o * 4 + 2] = 0.f, boxes[o * 4 + 3] = 0.f;
    conf_out[o] = 0.f;
    cls_out[o] = 0;
    keep_idx[o] = 0;
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
  static bool attr_done_dev[MTGV_MAX_DEVICES] = {};  // hipFuncSetAttribute is per device
  bool& attr_done = attr_done_dev[current_device()];
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
