// YOLOv8n-seg executor: conv stack on the f32-MFMA implicit GEMM, decode, NMS, mask GEMM.
#pragma once
#include "common.h"
#include "encoder.h"
#include "mtgv.h"

#include <map>
#include <string>
#include <vector>

namespace mtgv {

struct ConvW {
  float* w = nullptr;  // [cout][k][k][cin] BN-folded
  float* b = nullptr;  // [cout]
  int cout = 0, cin = 0, k = 1;
};
struct View {
  float* p = nullptr;
  int H = 0, W = 0, ct = 0, co = 0, C = 0;
  int fmt = 0;  // 0: f32; 1: SP8 (sp8.h) - same bytes per element, channel offsets in multiples of 8
  View slice(int off, int c) const {
    View v = *this;
    v.co = co + off;
    v.C = c;
    return v;
  }
};

class Detector {
 public:
  explicit Detector(const mtgv_detector_cfg& cfg);
  ~Detector();
  void set_param(const char* key, const float* host, int64_t numel);
  int missing() const;
  void finalize();
  void forward(const uint8_t* frames, int n, int flip, int* n_det, float* boxes, float* conf, int* cls, int* keep_idx,
               float* mask_logits, int mask_rows, hipStream_t s);
  void raw(int n, float* pred, float* protos, hipStream_t s);
  double flops_per_frame() const { return flops_; }
  const mtgv_detector_cfg& cfg() const { return cfg_; }
  int na() const { return na_; }
  int no() const { return 4 + cfg_.nc + nm_; }

 private:
  struct Raw {
    std::vector<int> shape;
    std::vector<float> data;
    bool set = false;
  };
  void expect(const std::string& key, std::vector<int> shape);
  void expect_conv_bn(const std::string& prefix, int cout, int cin, int k);
  ConvW fold(const std::string& prefix, int cin_pad = 0);           // Conv+BN
  ConvW plain(const std::string& prefix);                           // Conv2d with bias
  ConvW concat_out(const std::vector<ConvW>& parts);                // stack along cout
  float* upload(const std::vector<float>& v, int row_k = 0);  // row_k > 0: a GEMM B operand with rows of row_k floats
  void conv(const ConvW& w, const View& in, const View& out, int stride, int act, const View* res, int n, hipStream_t s);
  void c2f(int idx, const View& in, const View& out, int n, hipStream_t s);
  View take(int n, int h, int w, int c);
  View view(const std::string& k) const;

  mtgv_detector_cfg cfg_;
  int nm_ = 32, npr_ = 64, reg_max_ = 16, na_ = 0;
  std::map<std::string, Raw> raw_;
  bool finalized_ = false;
  std::vector<float*> dev_allocs_;
  double flops_ = 0;
  bool count_flops_ = false;

  // weights
  std::map<std::string, ConvW> cw_;
  struct C2fInfo { int cout, n; bool shortcut; int cin; };
  std::map<int, C2fInfo> c2f_;
  ConvW head_first_[3], head_box2_[3], head_cls2_[3], head_coef2_[3], head_box3_[3], head_cls3_[3], head_coef3_[3];
  ConvW proto_up_[4];

  // activations (arena)
  DevBuf arena_;
  size_t arena_used_ = 0;
  std::map<std::string, View> v_;
  float *rawhead_[3] = {nullptr, nullptr, nullptr}, *pred_ = nullptr, *coef_ = nullptr;
  int* nms_ws_ = nullptr;
  size_t nms_ws_bytes_ = 0;
  int last_n_ = 0;
  int fmt_ = 0;  // activation format of the forward in progress (0 f32, 1 SP8)
};

}  // namespace mtgv
