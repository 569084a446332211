// ConvNeXt-V2 encoder executor: builds the layer plan from a config, owns weights and
// workspace, runs the whole forward on one stream with hand-written kernels only.
#pragma once
#include "common.h"
#include "gemm_f32.h"
#include "mtgv.h"

#include <map>
#include <string>
#include <vector>

namespace mtgv {

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void alloc(size_t floats);
  void ensure(size_t floats) {
    if (floats > n) alloc(floats);
  }
  void release();
};

// weights of one Block (convnextv2.py:198-207), device pointers
struct BlockW {
  float *dw_w49 = nullptr, *dw_b = nullptr, *ln_w = nullptr, *ln_b = nullptr;
  float *w1 = nullptr, *b1 = nullptr, *gamma = nullptr, *beta = nullptr, *w2 = nullptr, *b2 = nullptr;
  // b2 + W2 . beta: the GRN shift folded into the pwconv2 bias at load time (null -> shift applied in the A prologue)
  float* b2_folded = nullptr;
  // fused MLP path (mlp_fused.h): W2 in the k order of the fused kernel's operand + its row scales; null -> packed per call
  const void* w2p = nullptr;
  const float* w2p_scale = nullptr;
};
struct BlockWs {
  float *t1 = nullptr, *t2 = nullptr, *hid = nullptr, *part = nullptr, *scale = nullptr, *bfold = nullptr;
};
struct BlockWsSize {
  size_t t, hid, part, scale, bfold;
  size_t total() const { return 2 * t + hid + part + scale + bfold; }
};
BlockWsSize block_ws_size(int n, int h, int w, int c);
// Block.forward on NHWC x -> out (may not alias x)
// The LayerNorm over C that follows a stage's last block (the downsample's): when the block runs fused (mlp_fused_kernel.h) its
// output pass applies it in the epilogue and writes the normalised rows in SP8 form to `out_sp8` (which may be the block's own
// input buffer) instead of the block output.  run_block returns true when it did so.
struct BlockLn {
  float* out_sp8 = nullptr;
  const float* w = nullptr;
  const float* b = nullptr;
  float eps = 1e-6f;
};
bool run_block(const float* x, float* out, int n, int h, int w, int c, int act, const BlockW& bw, const BlockWs& ws,
               hipStream_t s, const BlockLn* ln = nullptr);

GemmArgs linear_args(const float* A, int lda, const float* W, const float* bias, float* Out, int ldo, int M, int N, int K,
                     int act);

enum Repack { R_NONE = 0, R_OIHW_OHWI = 1, R_DW49 = 2, R_HEADPERM = 3 };

struct ParamSlot {
  std::vector<int> shape;  // reference shape
  Repack repack = R_NONE;
  int perm_p = 0, perm_c = 0;  // R_HEADPERM: input index (c*P + p) -> (p*zc + c)
  float* dev = nullptr;
  int64_t numel = 0;
  bool set = false;
  bool keep_host = false;
  std::vector<float> host;  // reference-layout copy for load-time folding
};

class ParamStore {
 public:
  ~ParamStore();
  float* add(const std::string& key, std::vector<int> shape, Repack r = R_NONE, int perm_p = 0, int perm_c = 0,
             bool keep_host = false);
  const std::vector<float>& host(const std::string& key) const { return slots_.at(key).host; }
  void set(const std::string& key, const float* host, int64_t numel);
  int missing() const;
  const std::map<std::string, ParamSlot>& slots() const { return slots_; }

 private:
  std::map<std::string, ParamSlot> slots_;
};

class Encoder {
 public:
  explicit Encoder(const mtgv_encoder_cfg& cfg);
  ~Encoder();
  // mode 0 (default): every launch issued eagerly; 1: batches of <= max_n images replay a captured hipGraph
  void set_graph_mode(int mode, int max_n);
  void set_param(const char* key, const float* host, int64_t numel) {
    params_.set(key, host, numel);
    prepared_ = false;
  }
  int missing() const { return params_.missing(); }
  void forward(const void* x, int layout, int n, float* z, hipStream_t s);
  void set_capture(bool on);
  void stage_output(int stage, int n, float* out, hipStream_t s);
  void flops(double* gemm, double* dw) const;
  const mtgv_encoder_cfg& cfg() const { return cfg_; }

 private:
  mtgv_encoder_cfg cfg_;
  int act_;
  int sh_[4], sw_[4];  // stage spatial sizes
  ParamStore params_;
  // weights
  float *stem_w_, *stem_b_, *stem_ln_w_, *stem_ln_b_;
  float *ds_ln_w_[4], *ds_ln_b_[4], *ds_w_[4], *ds_b_[4];
  std::vector<BlockW> blocks_[4];
  float *pool_w_ = nullptr, *pool_b_ = nullptr, *pool_ln_w_ = nullptr, *pool_ln_b_ = nullptr;
  float *head_w_ = nullptr, *head_b_ = nullptr, *head2_w_ = nullptr, *head2_b_ = nullptr;
  // workspace
  DevBuf x0_, xa_, xb_, ws_, head_a_, head_b2_;
  DevBuf stage_[4];
  bool capture_ = false;
  int last_n_ = 0;
  bool prepared_ = false;
  std::vector<std::string> blk_prefix_[4];
  DevBuf folded_bias_, w2p_;
  void prepare();
  void body(int n, float* z_out, hipStream_t s);
  int graph_mode_ = 0, graph_max_n_ = 16;
  std::map<int, hipGraphExec_t> graphs_;
  hipStream_t cap_stream_ = nullptr;
  DevBuf zbuf_;
};

}  // namespace mtgv
