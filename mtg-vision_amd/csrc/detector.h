// YOLOv8n-seg executor: conv stack on the f32-MFMA implicit GEMM, decode, NMS, mask GEMM.
#pragma once
#include "common.h"
#include "encoder.h"
#include "mtgv.h"

#include <map>
#include <string>
#include <vector>

namespace mtgv {

// raw head rows per anchor: [0,64) box logits (4 sides x 16 bins), [64,64+nc) class logits, [68,100) mask coefficients
static constexpr int RAW_CT = 100, RAW_CLS = 64, RAW_COEF = 68;

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
  // Conv(w1, SiLU) followed by the 1x1 conv w2 (act2) with w1's output consumed on chip (gemm_sp_kernel.h, EPI 32): `mid` is
  // where w1's output would go in two launches (used when the pair cannot be chained: f32 mode, flop counting, shapes)
  void conv_pair(const ConvW& w1, const View& in, const View& mid, int stride, const ConvW& w2, const View& out2, int act2, int n,
                 hipStream_t s);
  void c2f(int idx, const View& in, const View& out, int n, hipStream_t s, const ConvW* pre = nullptr, const View* pre_in = nullptr);
  // YOLO11 modules
  ConvW fold_dw(const std::string& prefix);                         // depthwise 3x3 Conv+BN -> weight [9][c], bias [c]
  void dwconv(const ConvW& w, const View& in, const View& out, int act, const float* add, int g_size, int g_stride, int n,
              hipStream_t s);
  void bottleneck(const std::string& prefix, const View& x, const View& tmp, const View& out, bool shortcut, int n, hipStream_t s);
  void c3k2(int idx, const View& in, const View& out, int n, hipStream_t s);
  void c2psa(int idx, const View& in, const View& out, int n, hipStream_t s);
  void build_v11();
  void arena_v11();
  void forward_v11(const uint8_t* frames, int n, int flip, hipStream_t s);
  void forward_v8(const uint8_t* frames, int n, int flip, hipStream_t s);
  void head_tail(int n, int* n_det, float* boxes, float* conf, int* cls, int* keep_idx, float* mask_logits, int mask_rows,
                 hipStream_t s);
  void conv0(const uint8_t* frames, int n, int flip, hipStream_t s);
  void sppf(const std::string& prefix, const View& in, const View& spp, const View& out, int n, hipStream_t s);
  void proto(const std::string& head, const View& p3, int n, hipStream_t s);
  void head_level_v8(int l, int n, hipStream_t s);
  void head_level_v11(int l, int n, hipStream_t s);
  // Fork-join inside one forward (library-owned streams and events, library kernels only - the concurrency contract
  // of include/mtgv.h): the prototype branch and the P3 / P4 head branches leave the caller's stream as soon as their
  // input exists and rejoin it before decode / the mask product.  fork_after(s, i): side stream i starts after
  // everything enqueued on s so far; join_into(s, i): s continues after everything enqueued on side stream i so far.
  bool fork_enabled() const;
 public:
  void set_fork(int mode) { fork_mode_ = mode; }  // -1: environment (MTGV_DET_FORK, default on), 0: off, 1: on
 private:
  int fork_mode_ = -1;
  hipStream_t fork_after(hipStream_t s, int i);
  void join_into(hipStream_t s, int i);
  bool v11() const { return cfg_.arch == 11; }
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
  struct C3k2Info { int cout, n, ch; bool c3k; };
  std::map<int, C3k2Info> c3k2_;
  ConvW head_bc_[3];            // v11: box + coefficient first convs merged
  ConvW cls_dw1_[3], cls_pw1_[3], cls_dw2_[3], cls_pw2_[3];
  std::string head_ = "model.22";
  ConvW head_first_[3], head_box2_[3], head_cls2_[3], head_coef2_[3], head_box3_[3], head_cls3_[3], head_coef3_[3];
  ConvW proto_up_[4];
  ConvW proto_up_all_;  // the four phase matrices stacked (kh, kw, cout): the ConvTranspose as one launch (GemmArgs::os_nq)

  // activations (arena)
  DevBuf arena_;
  size_t arena_used_ = 0;
  std::map<std::string, View> v_;
  float *rawhead_[3] = {nullptr, nullptr, nullptr}, *pred_ = nullptr, *coef_ = nullptr;
  int* nms_ws_ = nullptr;
  size_t nms_ws_bytes_ = 0;
  int last_n_ = 0;
  int fmt_ = 0;  // activation format of the forward in progress (0 f32, 1 SP8)
  static constexpr int NSIDE = 3;  // 0: prototype branch, 1: P3 head, 2: P4 head
  hipStream_t side_[NSIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork_[NSIDE] = {nullptr, nullptr, nullptr}, ev_join_[NSIDE] = {nullptr, nullptr, nullptr};
  bool side_busy_[NSIDE] = {false, false, false};  // forked in the forward in progress and not joined yet
};

}  // namespace mtgv
