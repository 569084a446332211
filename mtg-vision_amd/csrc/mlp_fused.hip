// Fused ConvNeXt-V2 MLP: launch side (kernels in mlp_fused_kernel.h).
#include "mlp_fused.h"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gemm_sp.h"
#include "mlp_fused_kernel.h"

namespace mtgv {

namespace {
bool fused_on() {
  static const bool on = [] { const char* e = getenv("MTGV_MLP_FUSED"); return e == nullptr || atoi(e) != 0; }();
  return on;
}

template <int C16, int ACT, int PASS>
void launch_pass(const MlpDev& g, hipStream_t s) {
  constexpr int C = 16 * C16, KB = (C16 + 1) / 2;
  constexpr int NV = PASS == 2 ? 4 : 2;
  constexpr size_t lds = (size_t)((NV * 4 * C * 4 + (PASS == 2 ? 2 * C * 4 : 0) + 1023) / 1024) * 1024 + (size_t)(PASS == 2 ? 6 : 3) * KB * 4096;
  static_assert(lds <= 80 * 1024, "two blocks per CU");
  if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit: opt in once per device
    static bool attr_done_dev[MTGV_MAX_DEVICES] = {};
    bool& attr_done = attr_done_dev[current_device()];
    if (!attr_done) {
      HIP_OK(hipFuncSetAttribute((const void*)mlp_fused_kernel<C16, ACT, PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done = true;
    }
  }
  hipLaunchKernelGGL((mlp_fused_kernel<C16, ACT, PASS>), dim3((unsigned)ceil_div(g.M, 128)), dim3(256), lds, s, g);
  HIP_OK(hipGetLastError());
}

template <int PASS>
void launch_any(const MlpDev& g, int C, int act, hipStream_t s) {
  if (C == 96 && act == ACT_MISH) launch_pass<6, ACT_MISH, PASS>(g, s);
  else if (C == 96 && act == ACT_GELU) launch_pass<6, ACT_GELU, PASS>(g, s);
  else if (C == 80 && act == ACT_MISH) launch_pass<5, ACT_MISH, PASS>(g, s);
  else if (C == 80 && act == ACT_GELU) launch_pass<5, ACT_GELU, PASS>(g, s);
  else MTGV_CHECK(false, ERR_INVALID, "mlp_fused: no instance for C=%d act=%d", C, act);
}
}  // namespace

bool mlp_fused_supported(int C, int hw, int act) {
  return fused_on() && gemm_sp_active() && (C == 96 || C == 80) && (act == ACT_MISH || act == ACT_GELU) && hw >= 128 && hw % 32 == 0;
}

void mlp_pack_w2p_launch(const float* W2, void* w2p, float* ws2, int C, hipStream_t s) {
  MTGV_CHECK(C % 8 == 0 && W2 != nullptr && w2p != nullptr && ws2 != nullptr, ERR_INVALID, "mlp_pack_w2p: C=%d", C);
  hipLaunchKernelGGL(mlp_pack_w2p_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, W2, (sp_h8*)w2p, ws2, C, 4 * C);
  HIP_OK(hipGetLastError());
}

void mlp_fused_launch(const MlpArgs& a, hipStream_t s) {
  MTGV_CHECK(mlp_fused_supported(a.C, a.hw, a.act), ERR_INVALID, "mlp_fused: C=%d hw=%d act=%d not supported", a.C, a.hw, a.act);
  MTGV_CHECK(a.x_sp8 && a.w1 && a.b1 && a.w2p && a.ws2 && a.b2 && a.gamma && a.res && a.out && a.part && a.scale && a.n_img > 0,
             ERR_INVALID, "mlp_fused: null argument");
  const int M = a.n_img * a.hw, H4 = 4 * a.C;
  MlpDev g;
  g.X = reinterpret_cast<const char*>(a.x_sp8);
  const char* w8 = nullptr;
  const float* wsc = nullptr;
  MTGV_CHECK(sp8_lookup(a.w1, a.C, &w8, &wsc), ERR_RUNTIME, "mlp_fused: pwconv1 weights have no SP8 copy");
  g.W1 = w8, g.ws1 = wsc, g.b1 = a.b1;
  g.W2p = reinterpret_cast<const char*>(a.w2p), g.ws2 = a.ws2, g.b2 = a.b2;
  g.scale = a.scale, g.res = a.res, g.Out = a.out, g.part = a.part;
  if (a.out_ln != nullptr) {
    MTGV_CHECK(a.ln_w != nullptr && a.ln_b != nullptr, ERR_INVALID, "mlp_fused: LayerNorm epilogue without its weights");
    g.OutLn = reinterpret_cast<char*>(a.out_ln), g.ln_w = a.ln_w, g.ln_b = a.ln_b, g.ln_eps = a.ln_eps;
  }
  g.zero = sp_zero_page();
  g.M = M, g.hw = a.hw, g.n_img = a.n_img;
  g.d_hw = make_fastdiv((uint32_t)a.hw);

  // The two launches are recorded like the launches they replace: pass 1 as pwconv1 (GRN partials), pass 2 as pwconv2
  // (GRN-scaled A, residual) - algorithmic FLOPs of the layer each (the recomputation in pass 2 is not counted), and
  // as compulsory bytes what each pass has to move: x once per pass, residual + output in pass 2.
  GemmArgs r1;
  r1.M = M, r1.N = H4, r1.K = a.C, r1.act = a.act, r1.grn_part = a.part;
  const double wbytes = 2.0 * H4 * a.C * 4.0, tiles = (double)ceil_div(M, 128);
  gemm_profile_begin(r1, s, 2, tiles * (H4 * a.C * 4.0), 4.0 * ((double)M * a.C + (double)H4 * a.C + (double)(M / 32) * H4));
  launch_any<1>(g, a.C, a.act, s);
  gemm_profile_end(s);
  GrnLayout gl;
  gl.unit_rows = 32, gl.segmax = 1, gl.floats = (size_t)(M / 32) * H4;
  grn_finalize_launch(a.part, gl, a.n_img, a.hw, H4, a.gamma, a.scale, s);
  GemmArgs r2;
  r2.M = M, r2.N = a.C, r2.K = H4, r2.a_scale = a.scale, r2.res = a.res;
  gemm_profile_begin(r2, s, 2, tiles * wbytes, 4.0 * (3.0 * (double)M * a.C + 2.0 * H4 * a.C));
  // tuning aid: MTGV_MLP_STAMPS=<file> appends [M, C, tiles] + per-tile stamps of every pass-2 launch (synchronises)
  static const char* stamp_path = getenv("MTGV_MLP_STAMPS");
  long* sbuf = nullptr;
  const int ntiles = ceil_div(M, 128);
  if (stamp_path != nullptr && *stamp_path) {
    HIP_OK(hipMalloc(&sbuf, (size_t)ntiles * 8 * sizeof(long)));
    HIP_OK(hipMemsetAsync(sbuf, 0, (size_t)ntiles * 8 * sizeof(long), s));
    g.stamps = sbuf;
  }
  launch_any<2>(g, a.C, a.act, s);
  gemm_profile_end(s);
  if (sbuf != nullptr) {
    std::vector<long> host((size_t)ntiles * 8);
    HIP_OK(hipStreamSynchronize(s));
    HIP_OK(hipMemcpy(host.data(), sbuf, host.size() * sizeof(long), hipMemcpyDeviceToHost));
    HIP_OK(hipFree(sbuf));
    if (FILE* f = fopen(stamp_path, "ab")) {
      const long hdr[3] = {M, a.C, ntiles};
      fwrite(hdr, sizeof(long), 3, f);
      fwrite(host.data(), sizeof(long), host.size(), f);
      fclose(f);
    }
  }
}

}  // namespace mtgv
