// TEMPORARY: entry points not implemented yet return a runtime error.
#include "mtgv.h"
#include "common.h"
using namespace mtgv;
static int ni(const char* n) { set_last_error(std::string(n) + ": not implemented yet"); return ERR_RUNTIME; }
extern "C" {
MTGV_API int mtgv_detector_create(const mtgv_detector_cfg*, mtgv_detector**) { return ni("mtgv_detector_create"); }
MTGV_API void mtgv_detector_destroy(mtgv_detector*) {}
MTGV_API int mtgv_detector_set_param(mtgv_detector*, const char*, const float*, int64_t) { return ni("mtgv_detector_set_param"); }
MTGV_API int mtgv_detector_missing_params(const mtgv_detector*) { return -1; }
MTGV_API int mtgv_detector_finalize(mtgv_detector*) { return ni("mtgv_detector_finalize"); }
MTGV_API int mtgv_detector_forward(mtgv_detector*, const uint8_t*, int32_t, int32_t, int32_t*, float*, float*, int32_t*, int32_t*, float*, void*) { return ni("mtgv_detector_forward"); }
MTGV_API int mtgv_detector_raw(mtgv_detector*, int32_t, float*, float*, void*) { return ni("mtgv_detector_raw"); }
MTGV_API int mtgv_detector_flops(const mtgv_detector*, double*) { return ni("mtgv_detector_flops"); }
MTGV_API int mtgv_nms(const float*, int32_t, int32_t, int32_t, int32_t, float, float, int32_t, float, int32_t*, float*, float*, int32_t*, int32_t*, int32_t*, size_t, void*) { return ni("mtgv_nms"); }
MTGV_API size_t mtgv_nms_workspace_bytes(int32_t, int32_t) { return 0; }
MTGV_API int mtgv_warp_quads(const uint8_t*, int32_t, int32_t, int32_t, const float*, const int32_t*, int32_t, int32_t, int32_t, float, uint8_t*, void*) { return ni("mtgv_warp_quads"); }
}
