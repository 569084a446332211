// YOLO11n-seg on the GPU: the modules YOLOv8 does not have (ultralytics 8.3.x nn/modules: C3k2, C3k, C2PSA / PSABlock /
// Attention, DWConv, Detect(legacy=False) class branch - third-party to the reference, which trains this family by
// default: mtgvision/od_train.py:20, :55-56, :138-151).  Conv / C2f-skeleton / SPPF / Proto / decode / NMS / masks are
// shared with detector.hip; every 1x1 and 3x3 convolution runs on the split-precision implicit GEMM.
//   dwconv3_kernel   depthwise 3x3 + folded BN (+ SiLU) (+ f32 addend), f32 or SP8 in / out, channel-group gather
//   attn_kernel      softmax(q^T k / sqrt(kd)) v per (image, head): f32 VALU, online softmax, K / V staged in LDS
#include "act.h"
#include "detector.h"
#include "gemm_sp.h"
#include "rowops.h"
#include "sp8.h"

#include <math.h>

namespace mtgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static int make_div8(double v) { return (int)(ceil(v / 8.0) * 8.0); }
static int chn(int c) { return make_div8(std::min(c, 1024) * 0.25); }

// ---------------------------------------------------------------------------
// depthwise 3x3, stride 1, pad 1.  One thread = one pixel x 8 channels.
// input channel of output channel c: (c / g_size) * g_stride + c % g_size (+ the view's offset) - the positional
// encoding of Attention reads the v rows out of the qkv tensor that way.
// ---------------------------------------------------------------------------
template <bool IN_SP8, bool OUT_SP8>
__global__ __launch_bounds__(256) void dwconv3_kernel(const float* __restrict__ in, int ci_total, int ci_off, int g_size, int g_stride,
                                                     const float* __restrict__ w9, const float* __restrict__ bias,
                                                     const float* __restrict__ add, int add_ld, float* __restrict__ out,
                                                     int co_total, int co_off, int H, int W, int C, int act, long total) {
#pragma clang fp contract(off)
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over N*H*W*(C/8)
  if (idx >= total) return;
  const int c8n = C >> 3;
  const int c = (int)(idx % c8n) * 8;
  long t = idx / c8n;
  const int x = (int)(t % W);
  t /= W;
  const int y = (int)(t % H);
  const long n = t / H;
  const int cin = (c / g_size) * g_stride + (c % g_size) + ci_off;
  f32x4 a0 = *reinterpret_cast<const f32x4*>(bias + c), a1 = *reinterpret_cast<const f32x4*>(bias + c + 4);
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int iy = y + kh - 1;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ix = x + kw - 1;
      if (ix < 0 || ix >= W) continue;
      const float* p = in + ((n * H + iy) * W + ix) * (long)ci_total + cin;
      f32x4 v0, v1;
      if (IN_SP8) {
        const sp_h8 hi = reinterpret_cast<const sp_h8*>(p)[0], lo = reinterpret_cast<const sp_h8*>(p)[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) v0[e] = (float)hi[e] + (float)lo[e], v1[e] = (float)hi[4 + e] + (float)lo[4 + e];
      } else {
        v0 = reinterpret_cast<const f32x4*>(p)[0], v1 = reinterpret_cast<const f32x4*>(p)[1];
      }
      const float* wp = w9 + (kh * 3 + kw) * C + c;
      const f32x4 w0 = reinterpret_cast<const f32x4*>(wp)[0], w1 = reinterpret_cast<const f32x4*>(wp)[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) a0[e] = __builtin_fmaf(v0[e], w0[e], a0[e]), a1[e] = __builtin_fmaf(v1[e], w1[e], a1[e]);
    }
  }
  if (act == ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a0[e] = act_silu(a0[e]), a1[e] = act_silu(a1[e]);
  }
  const long pix = (n * H + y) * W + x;
  if (add != nullptr) {
    const float* q = add + pix * add_ld + c;
    a0 = a0 + reinterpret_cast<const f32x4*>(q)[0], a1 = a1 + reinterpret_cast<const f32x4*>(q)[1];
  }
  float* o = out + pix * co_total + co_off + c;
  if (OUT_SP8) {
    sp_h8 hi, lo;
    sp8_split8(a0, a1, hi, lo);
    reinterpret_cast<sp_h8*>(o)[0] = hi, reinterpret_cast<sp_h8*>(o)[1] = lo;
  } else {
    reinterpret_cast<f32x4*>(o)[0] = a0, reinterpret_cast<f32x4*>(o)[1] = a1;
  }
}

// ---------------------------------------------------------------------------
// Attention core.  qkv (B, N, heads * (2 kd + hd)) f32, per head [q kd | k kd | v hd]; out (B, N, heads * hd) f32.
// Block = (query tile of 256, head, image); a thread owns one query: q in registers, keys / values stream through LDS
// in chunks of KC, softmax is the online form (running maximum and sum), one rescale per chunk.
// ---------------------------------------------------------------------------
template <int KD, int HD, int KC>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int heads,
                                                  float scale) {
  __shared__ __attribute__((aligned(16))) float ks[KC][KD];
  __shared__ __attribute__((aligned(16))) float vs[KC][HD];
  const int tid = threadIdx.x;
  const int qi = blockIdx.x * 256 + tid;
  const int head = blockIdx.y;
  const long img = blockIdx.z;
  const int ld = heads * (2 * KD + HD);
  const float* base = qkv + img * N * ld + head * (2 * KD + HD);
  float q[KD], acc[HD];
  const bool live = qi < N;
#pragma unroll
  for (int d = 0; d < KD; ++d) q[d] = live ? base[(long)qi * ld + d] * scale : 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < N; k0 += KC) {
    __syncthreads();
    for (int i = tid; i < KC * (KD / 4); i += 256) {
      const int r = i / (KD / 4), c4 = i % (KD / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k0 + r < N) v = *reinterpret_cast<const f32x4*>(base + (long)(k0 + r) * ld + KD + c4 * 4);
      *reinterpret_cast<f32x4*>(&ks[r][c4 * 4]) = v;
    }
    for (int i = tid; i < KC * (HD / 4); i += 256) {
      const int r = i / (HD / 4), c4 = i % (HD / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k0 + r < N) v = *reinterpret_cast<const f32x4*>(base + (long)(k0 + r) * ld + 2 * KD + c4 * 4);
      *reinterpret_cast<f32x4*>(&vs[r][c4 * 4]) = v;
    }
    __syncthreads();
    const int kn = (N - k0) < KC ? (N - k0) : KC;
    float sc[KC];
    float cm = -INFINITY;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      float dsum = 0.f;
#pragma unroll
      for (int d = 0; d < KD; ++d) dsum = __builtin_fmaf(q[d], ks[j][d], dsum);
      sc[j] = j < kn ? dsum : -INFINITY;
      cm = fmaxf(cm, sc[j]);
    }
    const float mn = fmaxf(m, cm);
    const float f = __expf(m - mn);  // 0 on the first chunk (m = -inf)
    l *= f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= f;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const float pj = __expf(sc[j] - mn);  // exp(-inf) = 0 for the padding keys
      l += pj;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = __builtin_fmaf(pj, vs[j][d], acc[d]);
    }
    m = mn;
  }
  if (!live) return;
  const float inv = 1.0f / l;
  float* o = out + (img * N + qi) * (long)(heads * HD) + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const f32x4 v = {acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
    *reinterpret_cast<f32x4*>(o + d) = v;
  }
}

// ---------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------
// depthwise Conv2d(c, c, 3, groups=c, bias=False) + BatchNorm2d(eps 1e-3) -> weight [9][c] (tap-major), bias [c]
ConvW Detector::fold_dw(const std::string& p) {
  const Raw& w = raw_.at(p + ".conv.weight");
  const int c = w.shape[0];
  const auto& g = raw_.at(p + ".bn.weight").data;
  const auto& b = raw_.at(p + ".bn.bias").data;
  const auto& mu = raw_.at(p + ".bn.running_mean").data;
  const auto& var = raw_.at(p + ".bn.running_var").data;
  std::vector<float> wf((size_t)9 * c), bf(c);
  for (int o = 0; o < c; ++o) {
    const double sc = (double)g[o] / sqrt((double)var[o] + 1e-3);
    bf[o] = (float)((double)b[o] - (double)mu[o] * sc);
    for (int t = 0; t < 9; ++t) wf[(size_t)t * c + o] = (float)((double)w.data[(size_t)o * 9 + t] * sc);
  }
  ConvW cw;
  cw.w = upload(wf), cw.b = upload(bf), cw.cout = c, cw.cin = c, cw.k = 3;
  return cw;
}

void Detector::dwconv(const ConvW& w, const View& in, const View& out, int act, const float* add, int g_size, int g_stride, int n,
                      hipStream_t s) {
  MTGV_CHECK(out.C == w.cout && w.cout % 8 == 0, ERR_RUNTIME, "detector: depthwise conv channel mismatch");
  if (count_flops_) {
    flops_ += 2.0 * 9.0 * n * out.H * out.W * w.cout;
    return;
  }
  const int gs = g_size > 0 ? g_size : w.cout;
  const long total = (long)n * out.H * out.W * (w.cout / 8);
  const unsigned grid = (unsigned)((total + 255) / 256);
#define DW_GO(I_, O_)                                                                                                             \
  hipLaunchKernelGGL((dwconv3_kernel<I_, O_>), dim3(grid), dim3(256), 0, s, in.p, in.ct, in.co, gs, g_stride, w.w, w.b, add, w.cout, \
                     out.p, out.ct, out.co, out.H, out.W, w.cout, act, total)
  if (in.fmt == 1 && out.fmt == 1) DW_GO(true, true);
  else if (in.fmt == 1) DW_GO(true, false);
  else if (out.fmt == 1) DW_GO(false, true);
  else DW_GO(false, false);
#undef DW_GO
  HIP_OK(hipGetLastError());
}

// ---------------------------------------------------------------------------
// construction: expected ultralytics keys of yolo11n-seg
// ---------------------------------------------------------------------------
void Detector::build_v11() {
  auto P = [](int i) { return "model." + std::to_string(i); };
  const int c16 = chn(64), c32 = chn(128), c64 = chn(256), c128 = chn(512), c256 = chn(1024);
  auto bott = [&](const std::string& p, int c1, int c_, int c2) {
    expect_conv_bn(p + ".cv1", c_, c1, 3);
    expect_conv_bn(p + ".cv2", c2, c_, 3);
  };
  auto c3k2 = [&](int idx, int cin, int cout, bool c3k, double e) {
    const int ch = (int)(cout * e);
    const std::string p = P(idx);
    expect_conv_bn(p + ".cv1", 2 * ch, cin, 1);
    expect_conv_bn(p + ".cv2", cout, 3 * ch, 1);
    if (c3k) {
      const int c_ = ch / 2;
      expect_conv_bn(p + ".m.0.cv1", c_, ch, 1);
      expect_conv_bn(p + ".m.0.cv2", c_, ch, 1);
      expect_conv_bn(p + ".m.0.cv3", ch, 2 * c_, 1);
      for (int j = 0; j < 2; ++j) bott(p + ".m.0.m." + std::to_string(j), c_, c_, c_);
    } else {
      bott(p + ".m.0", ch, ch / 2, ch);
    }
    c3k2_[idx] = {cout, 1, ch, c3k};
  };
  expect_conv_bn(P(0), c16, 3, 3);
  expect_conv_bn(P(1), c32, c16, 3);
  c3k2(2, c32, c64, false, 0.25);
  expect_conv_bn(P(3), c64, c64, 3);
  c3k2(4, c64, c128, false, 0.25);
  expect_conv_bn(P(5), c128, c128, 3);
  c3k2(6, c128, c128, true, 0.5);
  expect_conv_bn(P(7), c256, c128, 3);
  c3k2(8, c256, c256, true, 0.5);
  expect_conv_bn(P(9) + ".cv1", c256 / 2, c256, 1);
  expect_conv_bn(P(9) + ".cv2", c256, c256 / 2 * 4, 1);
  {  // C2PSA(c256, n = 1): c = 128, heads = c / 64, key_dim = head_dim / 2
    const int c = c256 / 2, nh = std::max(c / 64, 1), kd = (c / nh) / 2;
    const std::string p = P(10);
    expect_conv_bn(p + ".cv1", 2 * c, c256, 1);
    expect_conv_bn(p + ".cv2", c256, 2 * c, 1);
    expect_conv_bn(p + ".m.0.attn.qkv", c + 2 * nh * kd, c, 1);
    expect_conv_bn(p + ".m.0.attn.proj", c, c, 1);
    expect_conv_bn(p + ".m.0.attn.pe", c, 1, 3);
    expect_conv_bn(p + ".m.0.ffn.0", 2 * c, c, 1);
    expect_conv_bn(p + ".m.0.ffn.1", c, 2 * c, 1);
    MTGV_CHECK(nh == 2 && kd == 32 && c / nh == 64, ERR_INVALID, "detector: unexpected C2PSA geometry");
  }
  c3k2(13, c256 + c128, c128, false, 0.5);
  c3k2(16, c128 + c128, c64, false, 0.5);
  expect_conv_bn(P(17), c64, c64, 3);
  c3k2(19, c64 + c128, c128, false, 0.5);
  expect_conv_bn(P(20), c128, c128, 3);
  c3k2(22, c128 + c256, c256, true, 0.5);

  const int chs[3] = {c64, c128, c256};
  const int c2 = std::max(std::max(16, chs[0] / 4), reg_max_ * 4);
  const int c3 = std::max(chs[0], std::min(cfg_.nc, 100));
  const int c4 = std::max(chs[0] / 4, nm_);
  MTGV_CHECK(c2 == 64 && c3 == 64 && c4 == 32, ERR_INVALID, "detector: unexpected head widths");
  const std::string H = head_;
  for (int l = 0; l < 3; ++l) {
    const std::string ls = std::to_string(l);
    expect_conv_bn(H + ".cv2." + ls + ".0", c2, chs[l], 3);
    expect_conv_bn(H + ".cv2." + ls + ".1", c2, c2, 3);
    expect(H + ".cv2." + ls + ".2.weight", {4 * reg_max_, c2, 1, 1});
    expect(H + ".cv2." + ls + ".2.bias", {4 * reg_max_});
    // Detect(legacy=False): Sequential(DWConv(x, x, 3), Conv(x, c3, 1)), Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)), Conv2d
    expect_conv_bn(H + ".cv3." + ls + ".0.0", chs[l], 1, 3);
    expect_conv_bn(H + ".cv3." + ls + ".0.1", c3, chs[l], 1);
    expect_conv_bn(H + ".cv3." + ls + ".1.0", c3, 1, 3);
    expect_conv_bn(H + ".cv3." + ls + ".1.1", c3, c3, 1);
    expect(H + ".cv3." + ls + ".2.weight", {cfg_.nc, c3, 1, 1});
    expect(H + ".cv3." + ls + ".2.bias", {cfg_.nc});
    expect_conv_bn(H + ".cv4." + ls + ".0", c4, chs[l], 3);
    expect_conv_bn(H + ".cv4." + ls + ".1", c4, c4, 3);
    expect(H + ".cv4." + ls + ".2.weight", {nm_, c4, 1, 1});
    expect(H + ".cv4." + ls + ".2.bias", {nm_});
  }
  expect(H + ".dfl.conv.weight", {1, reg_max_, 1, 1});
  expect_conv_bn(H + ".proto.cv1", npr_, chs[0], 3);
  expect(H + ".proto.upsample.weight", {npr_, npr_, 2, 2});
  expect(H + ".proto.upsample.bias", {npr_});
  expect_conv_bn(H + ".proto.cv2", npr_, npr_, 3);
  expect_conv_bn(H + ".proto.cv3", nm_, npr_, 1);
}

// ---------------------------------------------------------------------------
// activation arena
// ---------------------------------------------------------------------------
void Detector::arena_v11() {
  const int nb = cfg_.max_batch, S = cfg_.imgsz;
  const int s2 = S / 2, s4 = S / 4, s8 = S / 8, s16 = S / 16, s32 = S / 32;
  const int c16 = chn(64), c32 = chn(128), c64 = chn(256), c128 = chn(512), c256 = chn(1024);
  struct B { const char* name; int h, w, c; };
  std::vector<B> bufs = {
      {"x0", S, S, 4}, {"l0", s2, s2, c16}, {"l1", s4, s4, c32},
      {"cat2", s4, s4, 48}, {"tmp2", s4, s4, 8}, {"l2", s4, s4, c64},
      {"l3", s8, s8, c64}, {"cat4", s8, s8, 96}, {"tmp4", s8, s8, 16},
      {"cat15", s8, s8, c128 + c128},                                     // concat 15 = [up(13), 4]
      {"l5", s16, s16, c128}, {"cat6", s16, s16, 192}, {"kcat6", s16, s16, 64}, {"tmp6", s16, s16, 32},
      {"cat12", s16, s16, c256 + c128},                                   // concat 12 = [up(10), 6]
      {"l7", s32, s32, c256}, {"cat8", s32, s32, 384}, {"kcat8", s32, s32, 128}, {"tmp8", s32, s32, 64}, {"l8", s32, s32, c256},
      {"sppcat", s32, s32, 2 * c256}, {"l9", s32, s32, c256},
      {"psacat", s32, s32, c256}, {"qkv", s32, s32, c256}, {"att", s32, s32, c128}, {"atty", s32, s32, c128}, {"ffn", s32, s32, c256},
      {"cat21", s32, s32, c128 + c256},                                   // concat 21 = [20, 10]
      {"cat13", s16, s16, 192}, {"tmp13", s16, s16, 32},
      {"cat18", s16, s16, c64 + c128},                                    // concat 18 = [17, 13]
      {"cat16", s8, s8, 96}, {"tmp16", s8, s8, 16}, {"p3", s8, s8, c64},
      {"cat19", s16, s16, 192}, {"tmp19", s16, s16, 32}, {"p4", s16, s16, c128},
      {"cat22", s32, s32, 384}, {"kcat22", s32, s32, 128}, {"tmp22", s32, s32, 64}, {"p5", s32, s32, c256},
      // head temporaries per level (the levels' branches run concurrently): box + coefficient branches, class branch
      {"t1_0", s8, s8, 96}, {"t2_0", s8, s8, 96}, {"dwa_0", s8, s8, c64}, {"dwb_0", s8, s8, 64}, {"dwc_0", s8, s8, 64}, {"dwd_0", s8, s8, 64},
      {"t1_1", s16, s16, 96}, {"t2_1", s16, s16, 96}, {"dwa_1", s16, s16, c128}, {"dwb_1", s16, s16, 64}, {"dwc_1", s16, s16, 64}, {"dwd_1", s16, s16, 64},
      {"t1_2", s32, s32, 96}, {"t2_2", s32, s32, 96}, {"dwa_2", s32, s32, c256}, {"dwb_2", s32, s32, 64}, {"dwc_2", s32, s32, 64}, {"dwd_2", s32, s32, 64},
      {"pr1", s8, s8, npr_}, {"pr2", s4, s4, npr_}, {"pr3", s4, s4, npr_}, {"protos", s4, s4, nm_},
  };
  size_t total = 0;
  auto sz = [&](size_t n, int h, int w, int c) { total += (n * h * w * c + 63) / 64 * 64; };
  for (const B& b : bufs) sz(nb, b.h, b.w, b.c);
  sz(nb, s8, s8, RAW_CT), sz(nb, s16, s16, RAW_CT), sz(nb, s32, s32, RAW_CT);
  sz(nb, 1, na_, 4 + cfg_.nc + nm_), sz(nb, 1, cfg_.max_det, nm_);
  arena_.alloc(total + 1024);
  arena_used_ = 0;
  v_.clear();
  for (const B& b : bufs) v_[b.name] = take(nb, b.h, b.w, b.c);
  rawhead_[0] = take(nb, s8, s8, RAW_CT).p, rawhead_[1] = take(nb, s16, s16, RAW_CT).p, rawhead_[2] = take(nb, s32, s32, RAW_CT).p;
  pred_ = take(nb, 1, na_, 4 + cfg_.nc + nm_).p;
  coef_ = take(nb, 1, cfg_.max_det, nm_).p;
}

// ---------------------------------------------------------------------------
// modules
// ---------------------------------------------------------------------------
// Bottleneck(c1, c2, shortcut, k = (3, 3)): out = cv2(cv1(x)) (+ x).  `out` may be x itself (in place: every output
// element is read as the residual by the wave that later stores it).
void Detector::bottleneck(const std::string& p, const View& x, const View& tmp, const View& out, bool shortcut, int n, hipStream_t s) {
  conv(cw_.at(p + ".cv1"), x, tmp, 1, ACT_SILU, nullptr, n, s);
  conv(cw_.at(p + ".cv2"), tmp, out, 1, ACT_SILU, shortcut ? &x : nullptr, n, s);
}

// C3k2: the C2f skeleton (cv1 -> two chunks, one inner module appended, cv2 over the concat); the inner module is
// Bottleneck(c, c, e = 0.5) or C3k(c, c, 2) = cv3(cat(m(cv1(x)), cv2(x))) with two Bottlenecks(c/2, c/2, e = 1)
void Detector::c3k2(int idx, const View& in, const View& out, int n, hipStream_t s) {
  const C3k2Info& ci = c3k2_.at(idx);
  const int ch = ci.ch;
  const std::string P = "model." + std::to_string(idx);
  const View cat = view("cat" + std::to_string(idx));
  const View tmp = view("tmp" + std::to_string(idx));
  conv(cw_.at(P + ".cv1"), in, cat.slice(0, 2 * ch), 1, ACT_SILU, nullptr, n, s);
  const View src = cat.slice(ch, ch), dst = cat.slice(2 * ch, ch);
  if (ci.c3k) {
    const View kcat = view("kcat" + std::to_string(idx));
    const int c_ = ch / 2;
    const std::string M = P + ".m.0";
    const View a = kcat.slice(0, c_);
    conv(cw_.at(M + ".cv1"), src, a, 1, ACT_SILU, nullptr, n, s);
    for (int j = 0; j < 2; ++j) bottleneck(M + ".m." + std::to_string(j), a, tmp, a, true, n, s);
    conv(cw_.at(M + ".cv2"), src, kcat.slice(c_, c_), 1, ACT_SILU, nullptr, n, s);
    conv(cw_.at(M + ".cv3"), kcat, dst, 1, ACT_SILU, nullptr, n, s);
  } else {
    bottleneck(P + ".m.0", src, tmp, dst, true, n, s);
  }
  conv(cw_.at(P + ".cv2"), cat.slice(0, 3 * ch), out, 1, ACT_SILU, nullptr, n, s);
}

// C2PSA(c1, c1, n = 1, e = 0.5): a, b = cv1(x).split; b = b + attn(b); b = b + ffn(b); cv2(cat(a, b))
void Detector::c2psa(int idx, const View& in, const View& out, int n, hipStream_t s) {
  const std::string P = "model." + std::to_string(idx);
  const View cat = view("psacat");
  const int c = cat.ct / 2;
  constexpr int NH = 2, KD = 32, HD = 64;
  conv(cw_.at(P + ".cv1"), in, cat, 1, ACT_SILU, nullptr, n, s);
  const View b = cat.slice(c, c);
  const std::string A = P + ".m.0.attn";
  View qkv = view("qkv"), att = view("att"), y = view("atty");
  qkv.fmt = 0, att.fmt = 0;  // the attention core and the positional encoding read f32
  conv(cw_.at(A + ".qkv"), b, qkv, 1, ACT_NONE, nullptr, n, s);
  const int N = cat.H * cat.W;
  if (count_flops_) {
    flops_ += 2.0 * n * NH * (double)N * N * (KD + HD);
  } else {
    hipLaunchKernelGGL((attn_kernel<KD, HD, 16>), dim3((N + 255) / 256, NH, n), dim3(256), 0, s, qkv.p, att.p, N, NH,
                       1.0f / sqrtf((float)KD));
    HIP_OK(hipGetLastError());
  }
  // y = attn_out + pe(v): depthwise 3x3 over the v rows of qkv (head h: channels h*(2 KD + HD) + 2 KD ...)
  View vin = qkv;
  vin.co = 2 * KD;
  dwconv(cw_.at(A + ".pe"), vin, y, ACT_NONE, att.p, HD, 2 * KD + HD, n, s);
  conv(cw_.at(A + ".proj"), y, b, 1, ACT_NONE, &b, n, s);                       // b += proj(y)
  const View f = view("ffn");
  conv(cw_.at(P + ".m.0.ffn.0"), b, f, 1, ACT_SILU, nullptr, n, s);
  conv(cw_.at(P + ".m.0.ffn.1"), f, b, 1, ACT_NONE, &b, n, s);                  // b += ffn(b)
  conv(cw_.at(P + ".cv2"), cat, out, 1, ACT_SILU, nullptr, n, s);
}

void Detector::forward_v11(const uint8_t* frames, int n, int flip, hipStream_t s) {
  const int c64 = chn(256), c128 = chn(512), c256 = chn(1024);
  auto V = [&](const char* k) -> View { return view(k); };
  conv0(frames, n, flip, s);
  conv(cw_.at("model.1"), V("l0"), V("l1"), 2, ACT_SILU, nullptr, n, s);
  c3k2(2, V("l1"), V("l2"), n, s);
  conv(cw_.at("model.3"), V("l2"), V("l3"), 2, ACT_SILU, nullptr, n, s);
  const View cat15 = V("cat15"), cat12 = V("cat12"), cat18 = V("cat18"), cat21 = V("cat21");
  const View n4 = cat15.slice(c128, c128);          // node 4 lives in concat 15 = [up(13), 4]
  c3k2(4, V("l3"), n4, n, s);
  conv(cw_.at("model.5"), n4, V("l5"), 2, ACT_SILU, nullptr, n, s);
  const View n6 = cat12.slice(c256, c128);          // concat 12 = [up(10), 6]
  c3k2(6, V("l5"), n6, n, s);
  conv(cw_.at("model.7"), n6, V("l7"), 2, ACT_SILU, nullptr, n, s);
  c3k2(8, V("l7"), V("l8"), n, s);
  sppf("model.9", V("l8"), V("sppcat"), V("l9"), n, s);
  const View n10 = cat21.slice(c128, c256);         // concat 21 = [20, 10]
  c2psa(10, V("l9"), n10, n, s);
  // top-down
  if (!count_flops_) upsample2x_launch(n10.p, n10.ct, n10.co, cat12.p, cat12.ct, 0, n, n10.H, n10.W, c256, s);
  const View n13 = cat18.slice(c64, c128);          // concat 18 = [17, 13]
  c3k2(13, cat12, n13, n, s);
  if (!count_flops_) upsample2x_launch(n13.p, n13.ct, n13.co, cat15.p, cat15.ct, 0, n, n13.H, n13.W, c128, s);
  c3k2(16, cat15, V("p3"), n, s);
  // the same fork-join as forward_v8: prototype branch and P3 / P4 heads beside the rest of the neck
  proto(head_, V("p3"), n, fork_after(s, 0));
  head_level_v11(0, n, fork_after(s, 1));
  conv(cw_.at("model.17"), V("p3"), cat18.slice(0, c64), 2, ACT_SILU, nullptr, n, s);
  c3k2(19, cat18, V("p4"), n, s);
  head_level_v11(1, n, fork_after(s, 2));
  conv(cw_.at("model.20"), V("p4"), cat21.slice(0, c128), 2, ACT_SILU, nullptr, n, s);
  c3k2(22, cat21, V("p5"), n, s);
  head_level_v11(2, n, s);
  join_into(s, 1), join_into(s, 2);
}

// Segment head of level l: box + coefficient branches (first 3x3 convs merged), class branch of depthwise + pointwise pairs
void Detector::head_level_v11(int l, int n, hipStream_t s) {
  const char* feats[3] = {"p3", "p4", "p5"};
  const std::string ls = std::to_string(l);
  const View f = view(feats[l]), t1 = view("t1_" + ls), t2 = view("t2_" + ls);
  View rh;
  rh.p = rawhead_[l], rh.H = f.H, rh.W = f.W, rh.ct = RAW_CT, rh.co = 0, rh.C = RAW_CT;
  conv(head_bc_[l], f, t1, 1, ACT_SILU, nullptr, n, s);
  conv_pair(head_box2_[l], t1.slice(0, 64), t2.slice(0, 64), 1, head_box3_[l], rh.slice(0, 64), ACT_NONE, n, s);
  conv_pair(head_coef2_[l], t1.slice(64, 32), t2.slice(64, 32), 1, head_coef3_[l], rh.slice(RAW_COEF, nm_), ACT_NONE, n, s);
  // class branch: (depthwise 3x3, 1x1) twice, then the plain 1x1
  const View da = view("dwa_" + ls), db = view("dwb_" + ls), dc = view("dwc_" + ls), dd = view("dwd_" + ls);
  dwconv(cls_dw1_[l], f, da, ACT_SILU, nullptr, 0, 0, n, s);
  conv(cls_pw1_[l], da, db, 1, ACT_SILU, nullptr, n, s);
  dwconv(cls_dw2_[l], db, dc, ACT_SILU, nullptr, 0, 0, n, s);
  conv(cls_pw2_[l], dc, dd, 1, ACT_SILU, nullptr, n, s);
  conv(head_cls3_[l], dd, rh.slice(RAW_CLS, cfg_.nc), 1, ACT_NONE, nullptr, n, s);
}

}  // namespace mtgv
