// C ABI (include/mtgv.h): encoder, bank and single-op entry points.
// Detector / NMS / warp entry points live next to their kernels (detector.hip, nms.hip, warp.hip).
#include "mtgv.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <mutex>

#include "encoder.h"
#include "match.h"
#include "rowops.h"
#include "gemm_sp.h"

namespace mtgv {
const char* last_error_cstr();
}

using namespace mtgv;

namespace {
// The single-op entry points take raw weight pointers.  To run them on the same kernels the handles use, a constant
// operand is registered (SP8 copy + row scales) for the duration of the call; the call then synchronises the stream.
// Temporary registrations are visible to every thread through the registry: the single-op entry points therefore run
// one at a time (a process-wide lock held from before the lookup until the temporary copy has been released), so no
// other call can pick up a copy that is about to be freed.
std::recursive_mutex g_op_mu;
struct ScopedWeights {
  const float* w = nullptr;
  hipStream_t s;
  std::unique_lock<std::recursive_mutex> lk;
  ScopedWeights(const float* W, int n, int k, hipStream_t stream) : s(stream), lk(g_op_mu) {
    if (W == nullptr || k % 8 != 0 || gemm_precision() != GEMM_PREC_F16X3 || ((uintptr_t)W % 16) != 0) return;
    if (sp8_lookup(W, k, nullptr, nullptr)) return;  // the caller already owns a registration
    sp8_register(W, (size_t)n * k, k);
    sp8_refresh(W, 0, (size_t)n * k, s);
    w = W;
  }
  ~ScopedWeights() {
    if (w == nullptr) return;
    (void)hipStreamSynchronize(s);
    sp8_unregister(w);
  }
};
thread_local GrnLayout t_last_grn;

// max |x| of a device tensor (test / composition surface: one small kernel + a host read)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

// A power of two that brings an activation tensor into the fp16 range of the split GEMM (1 when it already is):
// the single-op entry points take arbitrary f32 data, the handles know their activations are LayerNorm outputs etc.
void range_guard(GemmArgs& g, const float* a_dev, long n, hipStream_t s) {
  if (gemm_precision() != GEMM_PREC_F16X3 || n <= 0) return;
  static unsigned* d_max_dev[MTGV_MAX_DEVICES] = {};  // one scratch word per device
  unsigned*& d_max = d_max_dev[current_device()];
  if (d_max == nullptr) HIP_OK(hipMalloc((void**)&d_max, sizeof(unsigned)));
  HIP_OK(hipMemsetAsync(d_max, 0, sizeof(unsigned), s));
  const unsigned grid = (unsigned)std::min<long>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, a_dev, n, d_max);
  unsigned bits = 0;
  HIP_OK(hipMemcpyAsync(&bits, d_max, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  float mx;
  memcpy(&mx, &bits, sizeof(float));
  if (!(mx > 16384.0f) || !(mx < INFINITY)) return;
  int ex;
  (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1): scaled maximum lands in [2^13, 2^14)
  g.a_mul = ldexpf(1.0f, 14 - ex);
  g.a_unmul = ldexpf(1.0f, ex - 14);
}
}  // namespace

struct mtgv_encoder {
  Encoder impl;
  explicit mtgv_encoder(const mtgv_encoder_cfg& c) : impl(c) {}
};
struct mtgv_bank {
  Bank impl;
  mtgv_bank(int d, int64_t c) : impl(d, c) {}
};

extern "C" {

MTGV_API const char* mtgv_last_error(void) { return last_error_cstr(); }
MTGV_API int mtgv_version(void) { return 100; }
MTGV_API int mtgv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- GEMM operand precision ----
MTGV_API int mtgv_set_gemm_precision(int32_t prec) {
  return guarded([&] { gemm_set_precision(prec); });
}
MTGV_API int mtgv_get_gemm_precision(int32_t* prec) {
  return guarded([&] {
    MTGV_CHECK(prec != nullptr, ERR_INVALID, "null output");
    *prec = gemm_precision();
  });
}

// ---- GEMM launch profiler ----
MTGV_API int mtgv_profile_gemm(int32_t enable) {
  return guarded([&] { gemm_profile_enable(enable != 0); });
}
MTGV_API int mtgv_profile_gemm_read(double* total_ms, double* total_flops, int64_t* launches) {
  return guarded([&] {
    long l = 0;
    gemm_profile_read(total_ms, total_flops, &l);
    if (launches) *launches = l;
  });
}

MTGV_API int mtgv_profile_gemm_bytes(double* total_bytes) {
  return guarded([&] {
    MTGV_CHECK(total_bytes != nullptr, ERR_INVALID, "null output");
    *total_bytes = gemm_profile_bytes();
  });
}

MTGV_API int mtgv_profile_gemm_dump(const char* csv_path) {
  return guarded([&] {
    MTGV_CHECK(csv_path != nullptr, ERR_INVALID, "null path");
    gemm_profile_dump(csv_path);
  });
}

// ---- encoder ----
MTGV_API int mtgv_encoder_create(const mtgv_encoder_cfg* cfg, mtgv_encoder** out) {
  return guarded([&] {
    MTGV_CHECK(cfg != nullptr && out != nullptr, ERR_INVALID, "null argument");
    *out = new mtgv_encoder(*cfg);
  });
}
MTGV_API void mtgv_encoder_destroy(mtgv_encoder* h) { delete h; }
MTGV_API int mtgv_encoder_set_param(mtgv_encoder* h, const char* key, const float* data_host, int64_t numel) {
  return guarded([&] {
    MTGV_CHECK(h && key && data_host, ERR_INVALID, "null argument");
    h->impl.set_param(key, data_host, numel);
  });
}
MTGV_API int mtgv_encoder_missing_params(const mtgv_encoder* h) { return h ? h->impl.missing() : -1; }
MTGV_API int mtgv_encoder_forward(mtgv_encoder* h, const void* x_dev, int32_t layout, int32_t n, float* z_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.forward(x_dev, layout, n, z_dev, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_encoder_set_graph(mtgv_encoder* h, int32_t mode, int32_t max_n) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.set_graph_mode(mode, max_n);
  });
}
MTGV_API int mtgv_encoder_set_capture(mtgv_encoder* h, int32_t on) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.set_capture(on != 0);
  });
}
MTGV_API int mtgv_encoder_stage_output(mtgv_encoder* h, int32_t stage, int32_t n, float* out_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && out_dev != nullptr, ERR_INVALID, "null argument");
    h->impl.stage_output(stage, n, out_dev, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_encoder_flops(const mtgv_encoder* h, double* gemm_flops, double* dw_flops) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.flops(gemm_flops, dw_flops);
  });
}

// ---- bank ----
MTGV_API int mtgv_bank_create(int32_t dim, int64_t capacity, mtgv_bank** out) {
  return guarded([&] {
    MTGV_CHECK(out != nullptr, ERR_INVALID, "null argument");
    *out = new mtgv_bank(dim, capacity);
  });
}
MTGV_API void mtgv_bank_destroy(mtgv_bank* h) { delete h; }
MTGV_API int64_t mtgv_bank_size(const mtgv_bank* h) { return h ? h->impl.size() : -1; }
MTGV_API int mtgv_bank_append(mtgv_bank* h, const float* vecs, int64_t n, int32_t is_device, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && (vecs != nullptr || n == 0), ERR_INVALID, "null argument");
    h->impl.append(vecs, n, is_device != 0, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_bank_set_row(mtgv_bank* h, int64_t row, const float* vec_host, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && vec_host != nullptr, ERR_INVALID, "null argument");
    h->impl.set_row(row, vec_host, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_bank_clear(mtgv_bank* h) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.clear();
  });
}
MTGV_API int mtgv_bank_get_rows(const mtgv_bank* h, int64_t row, int64_t n, float* out_host) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && (out_host != nullptr || n == 0), ERR_INVALID, "null argument");
    h->impl.get_rows(row, n, out_host);
  });
}
MTGV_API int mtgv_bank_topk(mtgv_bank* h, const float* q_dev, int32_t b, int32_t k, int64_t id_base, float score_threshold,
                            int64_t* ids_dev, float* scores_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    MTGV_CHECK(!(score_threshold != score_threshold), ERR_INVALID, "bank: score_threshold is NaN");
    h->impl.topk(q_dev, b, k, id_base, score_threshold, ids_dev, scores_dev, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_bank_topk_packed(mtgv_bank* h, const float* q_dev, int32_t b, int32_t k, int64_t id_base, int64_t* packed_dev,
                                   void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && packed_dev != nullptr, ERR_INVALID, "null argument");
    h->impl.topk(q_dev, b, k, id_base, -INFINITY, packed_dev, nullptr, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_topk_merge_gathered(const int64_t* gathered_dev, int32_t n_ranks, int32_t b_total, int32_t k, int32_t row0, int32_t b,
                                      float score_threshold, int64_t* ids_dev, float* scores_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(gathered_dev && ids_dev && scores_dev, ERR_INVALID, "null argument");
    MTGV_CHECK(!(score_threshold != score_threshold), ERR_INVALID, "topk_merge_gathered: score_threshold is NaN");
    topk_merge_gathered_launch(gathered_dev, n_ranks, b_total, k, row0, b, score_threshold, ids_dev, scores_dev, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_bank_prepass_fallbacks(const mtgv_bank* h, int64_t* count) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && count != nullptr, ERR_INVALID, "null argument");
    *count = h->impl.prepass_fallbacks();
  });
}
MTGV_API int mtgv_topk_merge(float* cand_scores_dev, const int64_t* cand_ids_dev, int32_t b, int32_t ncand, int32_t k,
                             float score_threshold, int64_t* ids_dev, float* scores_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(cand_scores_dev && cand_ids_dev && ids_dev && scores_dev, ERR_INVALID, "null argument");
    MTGV_CHECK(!(score_threshold != score_threshold), ERR_INVALID, "topk_merge: score_threshold is NaN");
    topk_merge_launch_i64(cand_scores_dev, cand_ids_dev, b, ncand, k, score_threshold, ids_dev, scores_dev, (hipStream_t)stream);
  });
}

// ---- single ops ----
MTGV_API int mtgv_op_linear(const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev, float* out_dev,
                            int32_t m, int32_t n, int32_t k, int32_t act, void* stream) {
  return guarded([&] {
    MTGV_CHECK(a_dev && w_dev && out_dev, ERR_INVALID, "null argument");
    ScopedWeights reg(w_dev, n, k, (hipStream_t)stream);
    GemmArgs g = linear_args(a_dev, k, w_dev, bias_dev, out_dev, n, m, n, k, act);
    g.res = res_dev;
    g.ldr = n;
    range_guard(g, a_dev, (long)m * k, (hipStream_t)stream);
    gemm_launch(g, gemm_plan(m, n, k, act != 0), (hipStream_t)stream);
  });
}
MTGV_API int64_t mtgv_op_linear_ex_part_floats(int32_t m, int32_t n, int32_t k, int32_t act, int32_t hw) {
  if (m <= 0 || n <= 0 || k <= 0 || hw <= 0) return 0;
  (void)act;
  return (int64_t)gemm_grn_part_floats_max(m, n, hw);
}
MTGV_API int mtgv_op_last_grn_layout(int32_t* unit_rows, int32_t* segmax) {
  return guarded([&] {
    MTGV_CHECK(unit_rows && segmax, ERR_INVALID, "null output");
    *unit_rows = t_last_grn.unit_rows;
    *segmax = t_last_grn.segmax;
  });
}
MTGV_API int mtgv_op_linear_ex(const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev, float* out_dev,
                               int32_t m, int32_t n, int32_t k, int32_t act, int32_t hw, const float* a_scale_dev,
                               const float* a_shift_dev, float* grn_part_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(a_dev && w_dev && out_dev, ERR_INVALID, "null argument");
    ScopedWeights reg(w_dev, n, k, (hipStream_t)stream);
    GemmArgs g = linear_args(a_dev, k, w_dev, bias_dev, out_dev, n, m, n, k, act);
    g.res = res_dev;
    g.ldr = n;
    g.hw = hw;
    g.a_scale = a_scale_dev;
    static DevBuf fold_tmp;  // test surface only: shift folded into a temporary bias
    if (a_shift_dev != nullptr) {
      fold_tmp.ensure((size_t)n);
      fold_shift_into_bias_launch(w_dev, a_shift_dev, bias_dev, fold_tmp.p, n, k, (hipStream_t)stream);
      g.bias = fold_tmp.p;
    }
    const GemmPlan pl = gemm_plan(m, n, k, act != 0, a_scale_dev != nullptr);
    if (grn_part_dev) {
      g.grn_part = grn_part_dev;  // before the layout: the kernel choice depends on it
      t_last_grn = gemm_grn_layout(g, pl);
      g.segmax = t_last_grn.segmax;
      g.grn_unit_rows = t_last_grn.unit_rows;
    }
    gemm_launch(g, pl, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_op_conv2d(const float* x_dev, const float* w_dev, const float* bias_dev, float* out_dev, int32_t n, int32_t h,
                            int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                            int32_t act, void* stream) {
  return guarded([&] {
    MTGV_CHECK(x_dev && w_dev && out_dev, ERR_INVALID, "null argument");
    MTGV_CHECK(stride > 0 && kh > 0 && kw > 0 && pad >= 0, ERR_INVALID, "bad conv geometry");
    const int oh = (h + 2 * pad - kh) / stride + 1, ow = (w + 2 * pad - kw) / stride + 1;
    MTGV_CHECK(oh > 0 && ow > 0, ERR_INVALID, "empty conv output");
    GemmArgs g;
    g.A = x_dev, g.W = w_dev, g.bias = bias_dev, g.Out = out_dev;
    g.M = n * oh * ow, g.N = cout, g.K = kh * kw * cin;
    g.H = h, g.Wd = w, g.c_total = cin, g.Cin = cin;
    g.KH = kh, g.KW = kw, g.stride = stride, g.pad = pad;
    g.OH = oh, g.OW = ow, g.OH2 = oh, g.OW2 = ow;
    g.ldo = cout;
    g.act = act;
    gemm_launch(g, gemm_plan(g.M, g.N, g.K, act != 0), (hipStream_t)stream);
  });
}
MTGV_API int mtgv_op_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, float* out_dev, int64_t rows,
                               int32_t c, float eps, void* stream) {
  return guarded([&] {
    MTGV_CHECK(x_dev && w_dev && b_dev && out_dev, ERR_INVALID, "null argument");
    ln_rows_launch(x_dev, c, 0, out_dev, c, 0, w_dev, b_dev, rows, c, eps, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_op_dwconv7(const float* x_dev, const float* w49_dev, const float* bias_dev, float* out_dev, int32_t n,
                             int32_t h, int32_t w, int32_t c, void* stream) {
  return guarded([&] {
    MTGV_CHECK(x_dev && w49_dev && bias_dev && out_dev, ERR_INVALID, "null argument");
    dwconv7_launch(x_dev, w49_dev, bias_dev, out_dev, n, h, w, c, (hipStream_t)stream);
  });
}
MTGV_API int64_t mtgv_op_block_workspace_floats(int32_t n, int32_t h, int32_t w, int32_t c) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return 0;
  return (int64_t)block_ws_size(n, h, w, c).total();
}
MTGV_API int mtgv_op_block(const float* x_dev, float* out_dev, int32_t n, int32_t h, int32_t w, int32_t c, int32_t act,
                           const float* dw_w49, const float* dw_b, const float* ln_w, const float* ln_b, const float* w1,
                           const float* b1, const float* gamma, const float* beta, const float* w2, const float* b2,
                           float* ws_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(x_dev && out_dev && ws_dev && x_dev != out_dev, ERR_INVALID, "null or aliased tensor");
    MTGV_CHECK(c % 4 == 0, ERR_INVALID, "block: C=%d must be a multiple of 4", c);
    BlockW bw;
    bw.dw_w49 = (float*)dw_w49, bw.dw_b = (float*)dw_b, bw.ln_w = (float*)ln_w, bw.ln_b = (float*)ln_b;
    bw.w1 = (float*)w1, bw.b1 = (float*)b1, bw.gamma = (float*)gamma, bw.beta = (float*)beta;
    bw.w2 = (float*)w2, bw.b2 = (float*)b2;
    ScopedWeights reg1(w1, 4 * c, c, (hipStream_t)stream), reg2(w2, c, 4 * c, (hipStream_t)stream);
    const BlockWsSize z = block_ws_size(n, h, w, c);
    BlockWs ws;
    ws.t1 = ws_dev;
    ws.t2 = ws.t1 + z.t;
    ws.hid = ws.t2 + z.t;
    ws.part = ws.hid + z.hid;
    ws.scale = ws.part + z.part;
    ws.bfold = ws.scale + z.scale;
    run_block(x_dev, out_dev, n, h, w, c, act, bw, ws, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_op_l2norm(const float* x_dev, float* out_dev, int64_t rows, int32_t d, void* stream) {
  return guarded([&] {
    MTGV_CHECK(x_dev && out_dev, ERR_INVALID, "null argument");
    l2norm_rows_launch(x_dev, out_dev, rows, d, (hipStream_t)stream);
  });
}

}  // extern "C"
