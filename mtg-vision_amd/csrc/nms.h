// Class-aware NMS, one workgroup per image (see nms.hip).
#pragma once
#include "common.h"
#include "mtgv.h"

namespace mtgv {
size_t nms_workspace_bytes(int n, int na);
// coef_out (n, max_det, nm) may be null
void nms_launch(const float* pred, int n, int nc, int nm, int na, float conf, float iou, int max_det, float max_wh, int* n_det,
                float* boxes, float* conf_out, int* cls_out, int* keep_idx, float* coef_out, int* ws, size_t ws_bytes,
                hipStream_t s);
}  // namespace mtgv
