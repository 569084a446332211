// ConvNeXt-V2 encoder forward, NHWC end to end (the reference permutes NCHW<->NHWC twice
// per block, convnextv2.py:215/:221; here the layout never changes).
//
// Per Block (convnextv2.py:212-224):
//   dwconv7 (+bias)            -> t1            rowops.hip  dwconv7_kernel
//   LayerNorm over C           -> t2            rowops.hip  ln_rows_kernel
//   pwconv1 + bias + act, GRN sum(x^2) partials -> hid   gemm_f32 (epilogue)
//   GRN finalize               -> scale[n][4C]  grn_finalize_kernel
//   (hid*scale + beta) @ W2^T + b2 + residual -> out      gemm_f32 (A prologue + epilogue)
#include "encoder.h"
#include <stdlib.h>
#include "rowops.h"
#include "gemm_sp.h"
#include "mlp_fused.h"

#include <string.h>

namespace mtgv {

void DevBuf::alloc(size_t floats) {
  release();
  if (floats == 0) return;
  HIP_OK(hipMalloc((void**)&p, floats * sizeof(float)));
  n = floats;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  n = 0;
}

// ---------------------------------------------------------------------------
// parameter store
// ---------------------------------------------------------------------------
ParamStore::~ParamStore() {
  for (auto& kv : slots_)
    if (kv.second.dev) {
      gemm_split_unregister(kv.second.dev);
      (void)hipFree(kv.second.dev);
    }
}

float* ParamStore::add(const std::string& key, std::vector<int> shape, Repack r, int perm_p, int perm_c, bool keep_host) {
  ParamSlot sl;
  sl.keep_host = keep_host;
  sl.shape = shape;
  sl.repack = r;
  sl.perm_p = perm_p;
  sl.perm_c = perm_c;
  sl.numel = 1;
  for (int d : shape) sl.numel *= d;
  HIP_OK(hipMalloc((void**)&sl.dev, (size_t)sl.numel * sizeof(float)));
  if (shape.size() >= 2 && r != R_DW49)  // conv / linear weights: B operands, rows of numel / shape[0] floats
    gemm_split_register(sl.dev, (size_t)sl.numel, (int)(sl.numel / shape[0]));
  MTGV_CHECK(slots_.find(key) == slots_.end(), ERR_INVALID, "duplicate parameter %s", key.c_str());
  slots_[key] = sl;
  return sl.dev;
}

void ParamStore::set(const std::string& key, const float* host, int64_t numel) {
  auto it = slots_.find(key);
  MTGV_CHECK(it != slots_.end(), ERR_KEY, "unknown parameter key '%s'", key.c_str());
  ParamSlot& sl = it->second;
  MTGV_CHECK(numel == sl.numel, ERR_INVALID, "parameter %s: got %lld elements, expected %lld", key.c_str(), (long long)numel,
             (long long)sl.numel);
  std::vector<float> tmp;
  const float* src = host;
  if (sl.repack == R_OIHW_OHWI) {
    const int O = sl.shape[0], I = sl.shape[1], KH = sl.shape[2], KW = sl.shape[3];
    tmp.resize((size_t)numel);
    for (int o = 0; o < O; ++o)
      for (int i = 0; i < I; ++i)
        for (int kh = 0; kh < KH; ++kh)
          for (int kw = 0; kw < KW; ++kw)
            tmp[(((size_t)o * KH + kh) * KW + kw) * I + i] = host[(((size_t)o * I + i) * KH + kh) * KW + kw];
    src = tmp.data();
  } else if (sl.repack == R_DW49) {
    const int C = sl.shape[0];
    tmp.resize((size_t)numel);
    for (int c = 0; c < C; ++c)
      for (int t = 0; t < 49; ++t) tmp[(size_t)t * C + c] = host[(size_t)c * 49 + t];
    src = tmp.data();
  } else if (sl.repack == R_HEADPERM) {
    // columns arrive in NCHW-flat order (c*P + p) (Reshape((-1, z)), convnextv2ae.py:230);
    // activations here are NHWC-flat (p*zc + c)
    const int Z = sl.shape[0], IN = sl.shape[1], P = sl.perm_p, ZC = sl.perm_c;
    MTGV_CHECK(P * ZC == IN, ERR_INVALID, "head permutation mismatch");
    tmp.resize((size_t)numel);
    for (int o = 0; o < Z; ++o)
      for (int c = 0; c < ZC; ++c)
        for (int p = 0; p < P; ++p) tmp[(size_t)o * IN + (size_t)p * ZC + c] = host[(size_t)o * IN + (size_t)c * P + p];
    src = tmp.data();
  }
  HIP_OK(hipMemcpy(sl.dev, src, (size_t)numel * sizeof(float), hipMemcpyHostToDevice));
  if (gemm_split_lookup(sl.dev) != nullptr) {
    gemm_split_refresh(sl.dev, 0, (size_t)numel, nullptr);
    HIP_OK(hipStreamSynchronize(nullptr));
  }
  if (sl.keep_host) sl.host.assign(host, host + numel);
  sl.set = true;
}

int ParamStore::missing() const {
  int m = 0;
  for (auto& kv : slots_) m += kv.second.set ? 0 : 1;
  return m;
}

// ---------------------------------------------------------------------------
// one block
// ---------------------------------------------------------------------------
GemmArgs linear_args(const float* A, int lda, const float* W, const float* bias, float* Out, int ldo, int M, int N, int K,
                     int act) {
  GemmArgs a;
  a.A = A;
  a.W = W;
  a.Out = Out;
  a.bias = bias;
  a.M = M, a.N = N, a.K = K;
  a.c_total = lda;
  a.Cin = K;
  a.ldo = ldo;
  a.act = act;
  return a;
}

BlockWsSize block_ws_size(int n, int h, int w, int c) {
  BlockWsSize z;
  const size_t M = (size_t)n * h * w;
  z.t = M * c;
  z.hid = M * 4 * c;
  z.part = gemm_grn_part_floats_max((int)M, 4 * c, h * w);
  z.scale = (size_t)n * 4 * c;
  z.bfold = (size_t)c;
  return z;
}

bool run_block(const float* x, float* out, int n, int h, int w, int c, int act, const BlockW& bw, const BlockWs& ws,
               hipStream_t s, const BlockLn* ln) {
  const int M = n * h * w, hw = h * w;
  // the normalised tensor has one consumer, pwconv1: written in SP8 when that launch runs on the LDS-DMA kernel
  const int fmt = gemm_sp_takes_sp8(bw.w1, M, 4 * c, c, c, 0) ? 1 : 0;
  if (dwconv7_ln_supported(w, c)) {
    dwconv7_ln_launch(x, bw.dw_w49, bw.dw_b, bw.ln_w, bw.ln_b, ws.t2, n, h, w, c, 1e-6f, s, fmt);
  } else {
    dwconv7_launch(x, bw.dw_w49, bw.dw_b, ws.t1, n, h, w, c, s);
    ln_rows_launch(ws.t1, c, 0, ws.t2, c, 0, bw.ln_w, bw.ln_b, M, c, 1e-6f, s, fmt);
  }

  if (fmt == 1 && mlp_fused_supported(c, hw, act)) {
    // narrow stage: pwconv1 + GRN + pwconv2 without the 4C hidden tensor in HBM (mlp_fused_kernel.h)
    MlpArgs m;
    m.x_sp8 = ws.t2, m.w1 = bw.w1, m.b1 = bw.b1, m.gamma = bw.gamma, m.res = x, m.out = out;
    m.part = ws.part, m.scale = ws.scale, m.n_img = n, m.hw = hw, m.C = c, m.act = act;
    if (bw.w2p != nullptr) {
      m.w2p = bw.w2p, m.ws2 = bw.w2p_scale;
    } else {  // single-op surface: the permuted copy of W2 is made per call, in the (unused) hidden-tensor workspace
      mlp_pack_w2p_launch(bw.w2, ws.hid, ws.hid + (size_t)c * 4 * c, c, s);
      m.w2p = ws.hid, m.ws2 = ws.hid + (size_t)c * 4 * c;
    }
    if (bw.b2_folded != nullptr) {
      m.b2 = bw.b2_folded;
    } else {
      fold_shift_into_bias_launch(bw.w2, bw.beta, bw.b2, ws.bfold, c, 4 * c, s);
      m.b2 = ws.bfold;
    }
    if (ln != nullptr && ln->out_sp8 != nullptr) m.out_ln = ln->out_sp8, m.ln_w = ln->w, m.ln_b = ln->b, m.ln_eps = ln->eps;
    mlp_fused_launch(m, s);
    return m.out_ln != nullptr;
  }

  GemmArgs g1 = linear_args(ws.t2, c, bw.w1, bw.b1, ws.hid, 4 * c, M, 4 * c, c, act);
  g1.a_fmt = fmt;
  const GemmPlan p1 = gemm_plan(M, 4 * c, c, true);
  g1.hw = hw;
  g1.grn_part = ws.part;  // set before the layout is computed: which kernel takes the launch depends on it
  const GrnLayout gl = gemm_grn_layout(g1, p1);
  g1.segmax = gl.segmax;
  g1.grn_unit_rows = gl.unit_rows;
  gemm_launch(g1, p1, s);

  grn_finalize_launch(ws.part, gl, n, hw, 4 * c, bw.gamma, ws.scale, s);

  GemmArgs g2 = linear_args(ws.hid, 4 * c, bw.w2, bw.b2, out, c, M, c, 4 * c, ACT_NONE);
  g2.res = x;
  g2.ldr = c;
  g2.hw = hw;
  g2.a_scale = ws.scale;
  if (bw.b2_folded != nullptr) {
    g2.bias = bw.b2_folded;  // beta already inside the bias (folded at weight load)
  } else {
    fold_shift_into_bias_launch(bw.w2, bw.beta, bw.b2, ws.bfold, c, 4 * c, s);
    g2.bias = ws.bfold;
  }
  gemm_launch(g2, gemm_plan(M, c, 4 * c, false, true), s);
  return false;
}

// ---------------------------------------------------------------------------
// Encoder
// ---------------------------------------------------------------------------
static std::string blk_key(int kind, int s, int j, const char* leaf) {
  char b[128];
  if (kind == MTGV_ENC_AE)
    snprintf(b, sizeof(b), "block%d.2.%d.%s", s, j, leaf);
  else
    snprintf(b, sizeof(b), "stages.%d.%d.%s", s, j, leaf);
  return b;
}

Encoder::Encoder(const mtgv_encoder_cfg& cfg) : cfg_(cfg) {
  MTGV_CHECK(cfg.kind == MTGV_ENC_AE || cfg.kind == MTGV_ENC_PLAIN, ERR_KEY, "encoder kind=%d not recognized", cfg.kind);
  MTGV_CHECK(cfg.in_chans == 3, ERR_INVALID, "in_chans=%d (only 3 supported)", cfg.in_chans);
  MTGV_CHECK(cfg.image_h > 0 && cfg.image_w > 0 && cfg.image_h % 32 == 0 && cfg.image_w % 32 == 0, ERR_INVALID,
             "image %dx%d must be a positive multiple of 32 (convnextv2ae.py:137-139)", cfg.image_h, cfg.image_w);
  MTGV_CHECK(cfg.max_batch > 0, ERR_INVALID, "max_batch=%d", cfg.max_batch);
  for (int i = 0; i < 4; ++i)
    MTGV_CHECK(cfg.depths[i] > 0 && cfg.dims[i] > 0 && cfg.dims[i] % 4 == 0, ERR_INVALID, "stage %d: depth=%d dim=%d", i,
               cfg.depths[i], cfg.dims[i]);
  const int P = (cfg.image_h / 32) * (cfg.image_w / 32);
  const int ht = cfg.head_type;
  if (cfg.kind == MTGV_ENC_AE) {
    MTGV_CHECK(ht >= MTGV_HEAD_CONV_LINEAR && ht <= MTGV_HEAD_POOL_MLP, ERR_KEY, "head_type=%d not recognized", ht);
    MTGV_CHECK(cfg.z_size % P == 0, ERR_INVALID, "z_size=%d %% internal_num=%d != 0 (convnextv2ae.py:126)", cfg.z_size, P);
  } else {
    MTGV_CHECK(ht == MTGV_HEAD_PLAIN, ERR_KEY, "plain encoder needs head_type plain");
  }
  MTGV_CHECK(cfg.z_size % 4 == 0, ERR_INVALID, "z_size=%d must be a multiple of 4", cfg.z_size);
  act_ = cfg.kind == MTGV_ENC_AE ? ACT_MISH : ACT_GELU;
  for (int s = 0; s < 4; ++s) {
    sh_[s] = cfg.image_h / (4 << s);
    sw_[s] = cfg.image_w / (4 << s);
  }
  const bool ae = cfg.kind == MTGV_ENC_AE;
  const int* d = cfg.dims;
  char k[128];

  // stem: Conv2d(3, C0, k4, s4) + LN  (convnextv2ae.py:193-196 / convnextv2.py:253-256)
  const char* stem_conv = ae ? "block0.0" : "downsample_layers.0.0";
  const char* stem_ln = ae ? "block0.1" : "downsample_layers.0.1";
  snprintf(k, sizeof(k), "%s.weight", stem_conv);
  stem_w_ = params_.add(k, {d[0], 3, 4, 4}, R_OIHW_OHWI);
  snprintf(k, sizeof(k), "%s.bias", stem_conv);
  stem_b_ = params_.add(k, {d[0]});
  snprintf(k, sizeof(k), "%s.weight", stem_ln);
  stem_ln_w_ = params_.add(k, {d[0]});
  snprintf(k, sizeof(k), "%s.bias", stem_ln);
  stem_ln_b_ = params_.add(k, {d[0]});
  for (int s = 1; s < 4; ++s) {
    char ln[64], cv[64];
    if (ae) {
      snprintf(ln, sizeof(ln), "block%d.0", s);
      snprintf(cv, sizeof(cv), "block%d.1", s);
    } else {
      snprintf(ln, sizeof(ln), "downsample_layers.%d.0", s);
      snprintf(cv, sizeof(cv), "downsample_layers.%d.1", s);
    }
    ds_ln_w_[s] = params_.add(std::string(ln) + ".weight", {d[s - 1]});
    ds_ln_b_[s] = params_.add(std::string(ln) + ".bias", {d[s - 1]});
    ds_w_[s] = params_.add(std::string(cv) + ".weight", {d[s], d[s - 1], 2, 2}, R_OIHW_OHWI);
    ds_b_[s] = params_.add(std::string(cv) + ".bias", {d[s]});
  }
  for (int s = 0; s < 4; ++s) {
    const int c = d[s];
    for (int j = 0; j < cfg.depths[s]; ++j) {
      BlockW b;
      b.dw_w49 = params_.add(blk_key(cfg.kind, s, j, "dwconv.weight"), {c, 1, 7, 7}, R_DW49);
      b.dw_b = params_.add(blk_key(cfg.kind, s, j, "dwconv.bias"), {c});
      b.ln_w = params_.add(blk_key(cfg.kind, s, j, "norm.weight"), {c});
      b.ln_b = params_.add(blk_key(cfg.kind, s, j, "norm.bias"), {c});
      b.w1 = params_.add(blk_key(cfg.kind, s, j, "pwconv1.weight"), {4 * c, c});
      b.b1 = params_.add(blk_key(cfg.kind, s, j, "pwconv1.bias"), {4 * c});
      b.gamma = params_.add(blk_key(cfg.kind, s, j, "grn.gamma"), {1, 1, 1, 4 * c});
      b.beta = params_.add(blk_key(cfg.kind, s, j, "grn.beta"), {1, 1, 1, 4 * c}, R_NONE, 0, 0, true);
      b.w2 = params_.add(blk_key(cfg.kind, s, j, "pwconv2.weight"), {c, 4 * c}, R_NONE, 0, 0, true);
      b.b2 = params_.add(blk_key(cfg.kind, s, j, "pwconv2.bias"), {c}, R_NONE, 0, 0, true);
      blocks_[s].push_back(b);
      blk_prefix_[s].push_back(blk_key(cfg.kind, s, j, ""));
    }
  }
  const int z = cfg.z_size, c3 = d[3];
  if (ht == MTGV_HEAD_PLAIN) {
    pool_ln_w_ = params_.add("norm.weight", {c3});
    pool_ln_b_ = params_.add("norm.bias", {c3});
    head_w_ = params_.add("head.weight", {z, c3});
    head_b_ = params_.add("head.bias", {z});
  } else {
    const bool conv_head = ht <= MTGV_HEAD_CONV_ACT_MLP;
    const bool mlp = ht == MTGV_HEAD_CONV_MLP || ht == MTGV_HEAD_CONV_ACT_MLP || ht == MTGV_HEAD_POOL_MLP;
    int head_in;
    if (conv_head) {
      const int zc = z / P;
      pool_w_ = params_.add("pool.0.weight", {zc, c3, 1, 1});
      pool_b_ = params_.add("pool.0.bias", {zc});
      pool_ln_w_ = params_.add("pool.2.weight", {zc});
      pool_ln_b_ = params_.add("pool.2.bias", {zc});
      head_in = z;
      if (mlp) {
        head_w_ = params_.add("head.layers.0.weight", {z, head_in}, R_HEADPERM, P, zc);
        head_b_ = params_.add("head.layers.0.bias", {z});
      } else {
        head_w_ = params_.add("head.weight", {z, head_in}, R_HEADPERM, P, zc);
        head_b_ = params_.add("head.bias", {z});
      }
    } else {
      pool_ln_w_ = params_.add("pool.1.weight", {c3});
      pool_ln_b_ = params_.add("pool.1.bias", {c3});
      head_in = c3;
      if (mlp) {
        head_w_ = params_.add("head.layers.0.weight", {z, head_in});
        head_b_ = params_.add("head.layers.0.bias", {z});
      } else {
        head_w_ = params_.add("head.weight", {z, head_in});
        head_b_ = params_.add("head.bias", {z});
      }
    }
    if (mlp) {
      head2_w_ = params_.add("head.layers.2.weight", {z, z});
      head2_b_ = params_.add("head.layers.2.bias", {z});
    }
  }

  // workspace for max_batch
  const int nb = cfg.max_batch;
  x0_.alloc((size_t)nb * cfg.image_h * cfg.image_w * 3);
  size_t act_max = 0, ws_max = 0;
  for (int s = 0; s < 4; ++s) {
    act_max = std::max(act_max, (size_t)nb * sh_[s] * sw_[s] * d[s]);
    ws_max = std::max(ws_max, block_ws_size(nb, sh_[s], sw_[s], d[s]).total());
  }
  xa_.alloc(act_max);
  xb_.alloc(act_max);
  ws_.alloc(ws_max);
  head_a_.alloc((size_t)nb * std::max(z, c3) + 16);
  head_b2_.alloc((size_t)nb * std::max(z, c3) + 16);
}

// GRN's "+ beta" (convnextv2.py:174) commutes with the following Linear: (h*s + beta) W2^T + b2 =
// (h*s) W2^T + (W2 beta + b2), so it is folded into the pwconv2 bias once per weight load (float64 dot).
void Encoder::prepare() {
  size_t total = 0;
  for (int s = 0; s < 4; ++s) total += blocks_[s].size() * (size_t)cfg_.dims[s];
  folded_bias_.ensure(total);
  std::vector<float> all(total);
  size_t off = 0;
  for (int s = 0; s < 4; ++s) {
    const int c = cfg_.dims[s];
    for (size_t j = 0; j < blocks_[s].size(); ++j) {
      const std::string& pre = blk_prefix_[s][j];
      const auto& w2 = params_.host(pre + "pwconv2.weight");
      const auto& b2 = params_.host(pre + "pwconv2.bias");
      const auto& beta = params_.host(pre + "grn.beta");
      for (int n = 0; n < c; ++n) {
        double acc = b2[n];
        for (int k = 0; k < 4 * c; ++k) acc += (double)w2[(size_t)n * 4 * c + k] * (double)beta[k];
        all[off + n] = (float)acc;
      }
      blocks_[s][j].b2_folded = folded_bias_.p + off;
      off += c;
    }
  }
  HIP_OK(hipMemcpy(folded_bias_.p, all.data(), total * sizeof(float), hipMemcpyHostToDevice));
  // permuted SP8 copies of pwconv2.weight for the stages the fused MLP kernel can take (any operand mode: the mode can
  // be switched after the weights are loaded)
  size_t w2p_floats = 0;
  for (int s = 0; s < 4; ++s)
    if (cfg_.dims[s] == 96 || cfg_.dims[s] == 80) w2p_floats += blocks_[s].size() * ((size_t)cfg_.dims[s] * 4 * cfg_.dims[s] + cfg_.dims[s]);
  if (w2p_floats > 0) {
    w2p_.ensure(w2p_floats);
    float* p = w2p_.p;
    for (int s = 0; s < 4; ++s) {
      const int c = cfg_.dims[s];
      if (!(c == 96 || c == 80)) continue;
      for (auto& b : blocks_[s]) {
        mlp_pack_w2p_launch(b.w2, p, p + (size_t)c * 4 * c, c, nullptr);
        b.w2p = p, b.w2p_scale = p + (size_t)c * 4 * c;
        p += (size_t)c * 4 * c + c;
      }
    }
    HIP_OK(hipStreamSynchronize(nullptr));
  }
  prepared_ = true;
}

Encoder::~Encoder() {
  for (auto& kv : graphs_) (void)hipGraphExecDestroy(kv.second);
  if (cap_stream_) (void)hipStreamDestroy(cap_stream_);
}

void Encoder::set_graph_mode(int mode, int max_n) {
  graph_mode_ = mode;
  if (max_n > 0) graph_max_n_ = max_n;
}

void Encoder::set_capture(bool on) {
  capture_ = on;
  if (on)
    for (int s = 0; s < 4; ++s) stage_[s].ensure((size_t)cfg_.max_batch * sh_[s] * sw_[s] * cfg_.dims[s]);
}

void Encoder::stage_output(int stage, int n, float* out, hipStream_t s) {
  MTGV_CHECK(capture_, ERR_RUNTIME, "stage capture is not enabled");
  MTGV_CHECK(stage >= 0 && stage < 4 && n > 0 && n <= last_n_, ERR_INVALID, "stage=%d n=%d", stage, n);
  HIP_OK(hipMemcpyAsync(out, stage_[stage].p, (size_t)n * sh_[stage] * sw_[stage] * cfg_.dims[stage] * sizeof(float),
                        hipMemcpyDeviceToDevice, s));
}

void Encoder::flops(double* gemm, double* dw) const {
  double g = 0, w = 0;
  const int* d = cfg_.dims;
  g += 2.0 * sh_[0] * sw_[0] * d[0] * 48;
  for (int s = 1; s < 4; ++s) g += 2.0 * sh_[s] * sw_[s] * d[s] * 4 * d[s - 1];
  for (int s = 0; s < 4; ++s) {
    const double hw = (double)sh_[s] * sw_[s];
    g += cfg_.depths[s] * 2.0 * (2.0 * hw * d[s] * 4 * d[s]);
    w += cfg_.depths[s] * 2.0 * 49 * hw * d[s];
  }
  const int z = cfg_.z_size, c3 = d[3], P = sh_[3] * sw_[3];
  const int ht = cfg_.head_type;
  const bool mlp = ht == MTGV_HEAD_CONV_MLP || ht == MTGV_HEAD_CONV_ACT_MLP || ht == MTGV_HEAD_POOL_MLP;
  if (ht <= MTGV_HEAD_CONV_ACT_MLP) {
    g += 2.0 * P * (z / P) * c3 + 2.0 * z * z;
  } else {
    g += 2.0 * z * c3;
  }
  if (mlp) g += 2.0 * z * z;
  if (gemm) *gemm = g;
  if (dw) *dw = w;
}

void Encoder::forward(const void* x, int layout, int n, float* z_out, hipStream_t s) {
  MTGV_CHECK(params_.missing() == 0, ERR_RUNTIME, "encoder has %d unset parameters", params_.missing());
  MTGV_CHECK(n > 0 && n <= cfg_.max_batch, ERR_INVALID, "batch %d outside [1, %d]", n, cfg_.max_batch);
  MTGV_CHECK(x != nullptr && z_out != nullptr, ERR_INVALID, "null tensor");
  if (!prepared_) prepare();
  const int H = cfg_.image_h, W = cfg_.image_w;
  const float sc = cfg_.scale_io ? 2.0f : 1.0f, sf = cfg_.scale_io ? -1.0f : 0.0f;
  if (layout == MTGV_IN_NCHW_F32)
    nchw_to_nhwc_launch((const float*)x, x0_.p, n, 3, H, W, 3, sc, sf, s);
  else if (layout == MTGV_IN_NHWC_F32)
    f32hwc_scale_launch((const float*)x, x0_.p, (long)n * H * W, 3, 3, sc, sf, s);
  else if (layout == MTGV_IN_NHWC_U8)
    u8_to_f32_launch((const uint8_t*)x, x0_.p, (long)n * H * W, 3, 3, sc, sf, 0, s);
  else
    MTGV_CHECK(false, ERR_INVALID, "unknown input layout %d", layout);

  // Small batches are launch-bound (~90 launches of a few microseconds each): replay the whole body as one
  // hipGraph.  The graph is captured per batch size after one eager pass (which performs every lazy
  // initialisation), writes into an internal buffer, and the result is copied to the caller's tensor.
  if (graph_mode_ != 0 && !capture_ && n <= graph_max_n_ && !gemm_profile_enabled()) {
    zbuf_.ensure((size_t)cfg_.max_batch * cfg_.z_size);
    const int gkey = n * 2 + gemm_precision();  // a captured graph holds the kernels of one operand precision
    auto it = graphs_.find(gkey);
    if (it == graphs_.end()) {
      body(n, zbuf_.p, s);  // eager warm-up, also a valid result
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      // capture on a private stream: the caller's stream may be the legacy default stream, which cannot capture
      if (cap_stream_ == nullptr) HIP_OK(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
      HIP_OK(hipStreamBeginCapture(cap_stream_, hipStreamCaptureModeThreadLocal));
      try {
        body(n, zbuf_.p, cap_stream_);
      } catch (...) {
        (void)hipStreamEndCapture(cap_stream_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
      }
      HIP_OK(hipStreamEndCapture(cap_stream_, &graph));
      HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      it = graphs_.emplace(gkey, exec).first;
    }
    HIP_OK(hipGraphLaunch(it->second, s));
    HIP_OK(hipMemcpyAsync(z_out, zbuf_.p, (size_t)n * cfg_.z_size * sizeof(float), hipMemcpyDeviceToDevice, s));
    last_n_ = n;
    return;
  }
  body(n, z_out, s);
  last_n_ = n;
}

void Encoder::body(int n, float* z_out, hipStream_t s) {
  const int H = cfg_.image_h, W = cfg_.image_w;
  const int* d = cfg_.dims;
  float* cur = xa_.p;
  float* alt = xb_.p;

  // stem: rows of 12 floats (4 pixels x RGB) are the "pixels" of a (4 x 1) conv, stride (4, 1)
  {
    GemmArgs g;
    g.A = x0_.p;
    g.W = stem_w_;
    g.bias = stem_b_;
    g.Out = alt;
    g.M = n * sh_[0] * sw_[0], g.N = d[0], g.K = 48;
    g.H = H, g.Wd = W / 4, g.c_total = 12, g.Cin = 12;
    g.KH = 4, g.KW = 1, g.stride = 4, g.stride_w = 1, g.pad = 0;
    g.OH = sh_[0], g.OW = sw_[0], g.OH2 = sh_[0], g.OW2 = sw_[0];
    g.ldo = d[0];
    const GemmPlan pl = gemm_plan(g.M, g.N, g.K);
    if (gemm_ln_fusable(g, pl)) {  // the stem's LayerNorm in the conv's epilogue: its output is written once
      g.Out = cur;
      g.ln_w = stem_ln_w_, g.ln_b = stem_ln_b_, g.ln_eps = 1e-6f;
      gemm_launch(g, pl, s);
    } else {
      gemm_launch(g, pl, s);
      ln_rows_launch(alt, d[0], 0, cur, d[0], 0, stem_ln_w_, stem_ln_b_, g.M, d[0], 1e-6f, s);
    }
  }

  const char* const ln_env = getenv("MTGV_LN_FUSE");  // read per call: a test compares both forms in one process
  const bool ln_fuse = ln_env == nullptr || atoi(ln_env) != 0;
  bool ln_done = false;
  for (int st = 0; st < 4; ++st) {
    const int h = sh_[st], w = sw_[st], c = d[st];
    if (st > 0) {
      // LayerNorm(channels_first) + Conv2d(k2, s2): convnextv2.py:258-263
      const int hp = sh_[st - 1], wp = sw_[st - 1], cp = d[st - 1];
      // the LayerNorm writes hi/lo-split rows when the conv runs on the LDS-DMA kernel (its 2x2 gather is then a DMA
      // source address, no conversion in the loader)
      const int fmt = gemm_sp_takes_sp8(ds_w_[st], n * h * w, c, 4 * cp, cp, 0) ? 1 : 0;
      if (ln_done) {
        std::swap(cur, alt);  // `alt` holds LayerNorm(x) in SP8 form already (the previous stage's last block wrote it in place)
        MTGV_CHECK(fmt == 1, ERR_RUNTIME, "encoder: fused LayerNorm output is SP8 but the downsample conv wants f32");
      } else {
        ln_rows_launch(cur, cp, 0, alt, cp, 0, ds_ln_w_[st], ds_ln_b_[st], (long)n * hp * wp, cp, 1e-6f, s, fmt);
      }
      ln_done = false;
      GemmArgs g;
      g.a_fmt = fmt;
      g.A = alt;
      g.W = ds_w_[st];
      g.bias = ds_b_[st];
      g.Out = cur;
      g.M = n * h * w, g.N = c, g.K = 4 * cp;
      g.H = hp, g.Wd = wp, g.c_total = cp, g.Cin = cp;
      g.KH = 2, g.KW = 2, g.stride = 2, g.pad = 0;
      g.OH = h, g.OW = w, g.OH2 = h, g.OW2 = w;
      g.ldo = c;
      gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
    }
    const BlockWsSize z = block_ws_size(n, h, w, c);
    BlockWs ws;
    ws.t1 = ws_.p;
    ws.t2 = ws.t1 + z.t;
    ws.hid = ws.t2 + z.t;
    ws.part = ws.hid + z.hid;
    ws.scale = ws.part + z.part;
    ws.bfold = ws.scale + z.scale;
    for (size_t j = 0; j < blocks_[st].size(); ++j) {
      // the stage's last block hands the downsample's LayerNorm to its fused output pass where it can (MTGV_LN_FUSE=0: off):
      // the normalised rows replace the block's input in place and the block's f32 output is never written
      BlockLn ln;
      if (ln_fuse && j + 1 == blocks_[st].size() && st < 3 && !capture_ &&
          gemm_sp_takes_sp8(ds_w_[st + 1], n * sh_[st + 1] * sw_[st + 1], d[st + 1], 4 * c, c, 0)) {
        ln.out_sp8 = cur, ln.w = ds_ln_w_[st + 1], ln.b = ds_ln_b_[st + 1], ln.eps = 1e-6f;
      }
      if (run_block(cur, alt, n, h, w, c, act_, blocks_[st][j], ws, s, &ln)) {
        ln_done = true;  // `cur` now holds LayerNorm(block output) as SP8 rows
      } else {
        std::swap(cur, alt);
      }
    }
    if (capture_)
      HIP_OK(hipMemcpyAsync(stage_[st].p, cur, (size_t)n * h * w * c * sizeof(float), hipMemcpyDeviceToDevice, s));
  }

  // head
  const int zs = cfg_.z_size, c3 = d[3], P = sh_[3] * sw_[3];
  const int ht = cfg_.head_type;
  const bool mlp = ht == MTGV_HEAD_CONV_MLP || ht == MTGV_HEAD_CONV_ACT_MLP || ht == MTGV_HEAD_POOL_MLP;
  float* feat = head_a_.p;  // input rows of the head linear
  int feat_dim;
  if (ht <= MTGV_HEAD_CONV_ACT_MLP) {
    // Conv1x1 C3 -> z/P [+Mish] -> LN over channels -> NHWC-flat (n, P*zc); convnextv2ae.py:219-231
    const int zc = zs / P;
    GemmArgs g = linear_args(cur, c3, pool_w_, pool_b_, head_b2_.p, zc, n * P, zc, c3,
                             ht == MTGV_HEAD_CONV_ACT_MLP ? ACT_MISH : ACT_NONE);
    gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
    ln_rows_launch(head_b2_.p, zc, 0, feat, zc, 0, pool_ln_w_, pool_ln_b_, (long)n * P, zc, 1e-6f, s);
    feat_dim = zs;
  } else {
    // GAP -> LN over C3: convnextv2ae.py:236-244 / convnextv2.py:292-296
    gap_launch(cur, head_b2_.p, n, P, c3, s);
    ln_rows_launch(head_b2_.p, c3, 0, feat, c3, 0, pool_ln_w_, pool_ln_b_, n, c3, 1e-6f, s);
    feat_dim = c3;
  }
  if (mlp) {
    GemmArgs g = linear_args(feat, feat_dim, head_w_, head_b_, head_b2_.p, zs, n, zs, feat_dim, ACT_MISH);
    gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
    GemmArgs g2 = linear_args(head_b2_.p, zs, head2_w_, head2_b_, z_out, zs, n, zs, zs, ACT_NONE);
    gemm_launch(g2, gemm_plan(g2.M, g2.N, g2.K), s);
  } else {
    GemmArgs g = linear_args(feat, feat_dim, head_w_, head_b_, z_out, zs, n, zs, feat_dim, ACT_NONE);
    gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
  }
}

}  // namespace mtgv
