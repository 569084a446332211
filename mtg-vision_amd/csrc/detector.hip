// YOLOv8n-seg forward on the GPU, NHWC fp32.  Module graph: ultralytics yolov8-seg.yaml at
// scale "n" (third-party to the reference; call site mtgvision/od_export.py:141-160, model
// family od_train.py:46-70).  BatchNorm (eps 1e-3) is folded into the conv weights at
// finalize(); Concat is free (producers write channel slices of the consumer's buffer);
// every Conv+SiLU is one launch of the f32-MFMA implicit GEMM (gemm_f32.hip).
#include "detector.h"
#include "nms.h"
#include "rowops.h"
#include "gemm_sp.h"
#include "sp8.h"
#include "act.h"

#include <math.h>

namespace mtgv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static int make_div8(double v) { return (int)(ceil(v / 8.0) * 8.0); }
static int chn(int c) { return make_div8(std::min(c, 1024) * 0.25); }
static int rep(int n) { return n > 1 ? std::max((int)lround(n * 0.33), 1) : n; }

// ---------------------------------------------------------------------------
// decode: DFL expectation -> ltrb -> xywh * stride; class sigmoid; coefficient copy
// rawhead rows: [0,64) box logits (4 sides x 16 bins), [64,64+nc) class logits, [68,100) coeffs
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ r0, const float* __restrict__ r1,
                                                    const float* __restrict__ r2, float* __restrict__ pred, int n, int nc, int nm,
                                                    int imgsz, int na) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)n * na) return;
  const int img = (int)(idx / na), a = (int)(idx % na);
  const int w0 = imgsz / 8, w1 = imgsz / 16, w2 = imgsz / 32;
  const int n0 = w0 * w0, n1 = w1 * w1;
  const float* row;
  int gw, pix;
  float stride;
  if (a < n0) {
    pix = a, gw = w0, stride = 8.f;
    row = r0 + ((long)img * n0 + pix) * RAW_CT;
  } else if (a < n0 + n1) {
    pix = a - n0, gw = w1, stride = 16.f;
    row = r1 + ((long)img * n1 + pix) * RAW_CT;
  } else {
    pix = a - n0 - n1, gw = w2, stride = 32.f;
    row = r2 + ((long)img * w2 * w2 + pix) * RAW_CT;
  }
  float d[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // rows are RAW_CT = 100 floats: 16-byte loads (a lane's row shares no line with its neighbours')
      const f32x4 t = *reinterpret_cast<const f32x4*>(row + s * 16 + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[q * 4 + e] = t[e];
        mx = fmaxf(mx, t[e]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = expf(v[i] - mx);
      sum += v[i];
    }
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) e += (v[i] / sum) * (float)i;
    d[s] = e;
  }
  const float ax = (float)(pix % gw) + 0.5f, ay = (float)(pix / gw) + 0.5f;
  const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
  float* P = pred + (long)img * (4 + nc + nm) * na + a;
  P[0] = (x1 + x2) / 2.f * stride;
  P[(long)na] = (y1 + y2) / 2.f * stride;
  P[(long)2 * na] = (x2 - x1) * stride;
  P[(long)3 * na] = (y2 - y1) * stride;
  for (int c = 0; c < nc; ++c) P[(long)(4 + c) * na] = 1.0f / (1.0f + expf(-row[RAW_CLS + c]));
  for (int c = 0; c < nm; c += 4) {
    if (c + 4 <= nm) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(row + RAW_COEF + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) P[(long)(4 + nc + c + e) * na] = t[e];
    } else {
      for (int e = 0; c + e < nm; ++e) P[(long)(4 + nc + c + e) * na] = row[RAW_COEF + c + e];
    }
  }
}

// NHWC -> NCHW (raw protos for parity tests)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C, long HW,
                                                          long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over N*C*HW (output order)
  if (idx >= total) return;
  const long p = idx % HW;
  const long t = idx / HW;
  const int c = (int)(t % C);
  const long n = t / C;
  out[idx] = in[(n * HW + p) * C + c];
}

// process_mask(..., upsample=True) tail: F.interpolate(bilinear, align_corners=False) x scale, then > 0.
// PX consecutive output pixels of a row per thread (one 16-byte store instead of sixteen 1-byte stores).
template <int PX>
__global__ __launch_bounds__(256) void mask_binarize_kernel(const float* __restrict__ logits, uint8_t* __restrict__ out, int mh,
                                                           int mw, int scale, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over n * (mh*scale) * (mw*scale / PX)
  if (idx >= total) return;
  const int ow = mw * scale, oh = mh * scale;
  const int owp = ow / PX;
  const int xg = (int)(idx % owp);
  const long t = idx / owp;
  const int y = (int)(t % oh);
  const long n = t / oh;
  const float inv = 1.0f / (float)scale;
  float sy = inv * ((float)y + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  const int y0 = (int)sy;
  const int y1 = y0 + (y0 < mh - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0;
  const float ly0 = 1.0f - ly1;
  const float* L = logits + n * mh * mw;
  uint8_t px[PX];
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    const int x = xg * PX + j;
    float sx = inv * ((float)x + 0.5f) - 0.5f;
    sx = sx < 0.f ? 0.f : sx;
    const int x0 = (int)sx;
    const int x1 = x0 + (x0 < mw - 1 ? 1 : 0);
    const float lx1 = sx - (float)x0;
    const float lx0 = 1.0f - lx1;
    const float v = ly0 * (lx0 * L[y0 * mw + x0] + lx1 * L[y0 * mw + x1]) + ly1 * (lx0 * L[y1 * mw + x0] + lx1 * L[y1 * mw + x1]);
    px[j] = v > 0.f ? 1 : 0;
  }
  uint8_t* const o = out + (n * oh + y) * (long)ow + (long)xg * PX;
  if (PX == 16) {
    *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(px);
  } else {
#pragma unroll
    for (int j = 0; j < PX; ++j) o[j] = px[j];
  }
}

// ---------------------------------------------------------------------------
// model.0 straight from the uint8 frame: Conv(3 -> 16, k3, s2, p1) + folded BN + SiLU, output SP8 or f32.
// K = 27 is too short for the matrix cores and the layer is bound by its 16-channel output; a thread computes four
// neighbouring output pixels x 16 channels with f32 FMAs, weights broadcast from LDS.  Fuses the u8 -> float
// conversion (img / 255, ultralytics preprocess) that used to be a separate pass over a 4-channel float copy.
// ---------------------------------------------------------------------------
template <bool SP8>
__global__ __launch_bounds__(256) void conv0_u8_kernel(const uint8_t* __restrict__ frames, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int S, int flip,
                                                      long total) {
  __shared__ __attribute__((aligned(16))) float ws[16 * 9 * 4 + 16];  // [o][tap][4] (cin padded to 4) + bias
  for (int i = threadIdx.x; i < 16 * 9 * 4; i += 256) ws[i] = w[i];
  if (threadIdx.x < 16) ws[16 * 9 * 4 + threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int OS = S >> 1, OQ = OS >> 2;  // output size, groups of 4 output columns per row
  // Output staging: a thread's four pixels are 256 contiguous bytes and thread i + 1 continues where thread i ends, so
  // a store issued by every lane for its own piece would touch 64 different lines.  Each wave passes its pieces through
  // LDS (two pixels = 8 pieces of 16 B per thread at a time, rows padded to 144 B) and stores them back transposed:
  // eight lanes write one thread's 128 bytes, a store instruction writes eight whole lines.
  __shared__ __attribute__((aligned(16))) f32x4 stage[4][64][9];
  const long idx_raw = (long)blockIdx.x * 256 + threadIdx.x;  // over n * OS * OQ
  const long idx = idx_raw < total ? idx_raw : total - 1;     // (threads past the end compute a duplicate and store nothing)
  const int q = (int)(idx % OQ);
  const long t = idx / OQ;
  const int oh = (int)(t % OS);
  const long n = t / OS;
  const int ow0 = q * 4;
  float acc[4][16];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[p][o] = ws[16 * 9 * 4 + o];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = 2 * oh - 1 + kh;
    if (ih < 0 || ih >= S) continue;
    const uint8_t* const rowp = frames + ((n * S + ih) * (long)S) * 3;
    float x[9][3];  // input columns 2*ow0-1 .. 2*ow0+7
    // The nine pixels are bytes 24 q - 3 .. 24 q + 23 of the row: one dword for the pixel left of the strip (zero padding
    // at q == 0 - the only column that can fall outside, S = 8 OQ) and three aligned 8-byte loads for the other eight.
    uint32_t d[7];
    d[0] = q > 0 ? *reinterpret_cast<const uint32_t*>(rowp + 24 * q - 4) : 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint2 v = *reinterpret_cast<const uint2*>(rowp + 24 * q + 8 * k);
      d[1 + 2 * k] = v.x, d[2 + 2 * k] = v.y;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      float b[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int k = 1 + 3 * j + ch;  // byte index in d[]
        const float u = (float)((d[k >> 2] >> (8 * (k & 3))) & 0xffu);  // v_cvt_f32_ubyteN
        // u / 255 correctly rounded without the division sequence: one Newton step on u * fl(1/255) gives the IEEE
        // quotient for every byte value (tests/test_oracle_detector_cpu.py checks all 256 against exact rational arithmetic)
        const float r255 = 1.0f / 255.0f;
        const float q0 = u * r255;
        b[ch] = __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, u), r255, q0);
      }
      x[j][0] = flip ? b[2] : b[0], x[j][1] = b[1], x[j][2] = flip ? b[0] : b[2];
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&ws[(o * 9 + kh * 3 + kw) * 4]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          float a = acc[p][o];
          a = __builtin_fmaf(x[2 * p + kw][0], wv[0], a);
          a = __builtin_fmaf(x[2 * p + kw][1], wv[1], a);
          a = __builtin_fmaf(x[2 * p + kw][2], wv[2], a);
          acc[p][o] = a;
        }
      }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long wave_idx0 = (long)blockIdx.x * 256 + wave * 64;  // output is contiguous in idx order: 64 floats per thread
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int p = half * 2 + pp;
      f32x4 v[4];
#pragma unroll
      for (int o = 0; o < 16; ++o) v[o >> 2][o & 3] = act_silu(acc[p][o]);
      if (SP8) {
        sp_h8 hi, lo;
        sp8_split8(v[0], v[1], hi, lo);
        stage[wave][lane][pp * 4 + 0] = __builtin_bit_cast(f32x4, hi), stage[wave][lane][pp * 4 + 1] = __builtin_bit_cast(f32x4, lo);
        sp8_split8(v[2], v[3], hi, lo);
        stage[wave][lane][pp * 4 + 2] = __builtin_bit_cast(f32x4, hi), stage[wave][lane][pp * 4 + 3] = __builtin_bit_cast(f32x4, lo);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) stage[wave][lane][pp * 4 + k] = v[k];
      }
    }
    // (one wave reads only what it wrote itself: LDS operations of a wave complete in order)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int T = i * 8 + (lane >> 3), k = lane & 7;
      const f32x4 piece = stage[wave][T][k];
      if (wave_idx0 + T < total) *reinterpret_cast<f32x4*>(out + (wave_idx0 + T) * 64 + half * 32 + k * 4) = piece;
    }
  }
}

// 5x5 max pool (stride 1, pad 2) on SP8 channel slices: a thread owns one 8-channel chunk, compares hi + lo and keeps
// the winning pair as it is (no re-rounding)
__global__ __launch_bounds__(256) void maxpool5_sp8_kernel(const float* __restrict__ in, int ci_total, int ci_off,
                                                          float* __restrict__ out, int co_total, int co_off, int H, int W, int C,
                                                          long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over N*H*W*(C/8)
  if (idx >= total) return;
  const int c8n = C >> 3;
  const int c = (int)(idx % c8n) * 8;
  long t = idx / c8n;
  const int w = (int)(t % W);
  t /= W;
  const int h = (int)(t % H);
  const long n = t / H;
  float best[8];
  sp_h8 bh, bl;
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = -INFINITY, bh[e] = (_Float16)0.f, bl[e] = (_Float16)0.f;
  for (int dh = -2; dh <= 2; ++dh) {
    const int ih = h + dh;
    if (ih < 0 || ih >= H) continue;
    for (int dw = -2; dw <= 2; ++dw) {
      const int iw = w + dw;
      if (iw < 0 || iw >= W) continue;
      const sp_h8* const p = reinterpret_cast<const sp_h8*>(in + ((n * H + ih) * W + iw) * ci_total + ci_off + c);
      const sp_h8 vh = p[0], vl = p[1];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (float)vh[e] + (float)vl[e];
        if (v > best[e]) best[e] = v, bh[e] = vh[e], bl[e] = vl[e];
      }
    }
  }
  sp_h8* const o = reinterpret_cast<sp_h8*>(out + ((n * H + h) * W + w) * co_total + co_off + c);
  o[0] = bh, o[1] = bl;
}

// SPPF's three chained 5x5 max pools (y1 = m(x), y2 = m(y1), y3 = m(y2)) in ONE launch: a block owns one 8-channel chunk
// of one image, keeps the (hi, lo) pairs of all H x W pixels in LDS and runs the three rounds out of it - the three
// separate launches were 25 us each for 6.5 MB of data (latency-bound: 25 dependent loads per thread).  Same comparison
// (hi + lo), same scan order, same strict >: the pairs written are those of maxpool5_sp8_kernel, bit for bit.
__global__ __launch_bounds__(256) void sppf_pools_sp8_kernel(float* __restrict__ buf, int c_total, int ch, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) char sp_sm[];
  const int HW = H * W;
  sp_h8* const a = reinterpret_cast<sp_h8*>(sp_sm);  // [HW][2]: hi piece, lo piece
  sp_h8* const b = a + (size_t)HW * 2;
  const int c8n = ch >> 3;
  const long n = blockIdx.x / c8n;
  const int c = (int)(blockIdx.x % c8n) * 8;
  char* const img = reinterpret_cast<char*>(buf + n * (long)HW * c_total);
  for (int p = threadIdx.x; p < HW; p += 256) {
    const sp_h8* const src = reinterpret_cast<const sp_h8*>(img + ((long)p * c_total + c) * 4);
    a[2 * p] = src[0], a[2 * p + 1] = src[1];
  }
  __syncthreads();
  sp_h8 *in = a, *out = b;
  for (int round = 1; round <= 3; ++round) {
    for (int p = threadIdx.x; p < HW; p += 256) {
      const int h = p / W, w = p - h * W;
      float best[8];
      sp_h8 bh, bl;
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = -INFINITY, bh[e] = (_Float16)0.f, bl[e] = (_Float16)0.f;
      for (int dh = -2; dh <= 2; ++dh) {
        const int ih = h + dh;
        if (ih < 0 || ih >= H) continue;
        for (int dw = -2; dw <= 2; ++dw) {
          const int iw = w + dw;
          if (iw < 0 || iw >= W) continue;
          const sp_h8 vh = in[2 * (ih * W + iw)], vl = in[2 * (ih * W + iw) + 1];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = (float)vh[e] + (float)vl[e];
            if (v > best[e]) best[e] = v, bh[e] = vh[e], bl[e] = vl[e];
          }
        }
      }
      out[2 * p] = bh, out[2 * p + 1] = bl;
      sp_h8* const dst = reinterpret_cast<sp_h8*>(img + ((long)p * c_total + round * ch + c) * 4);
      dst[0] = bh, dst[1] = bl;
    }
    __syncthreads();
    sp_h8* const t = in;
    in = out, out = t;
  }
}

// ---------------------------------------------------------------------------
// construction: expected ultralytics keys
// ---------------------------------------------------------------------------
void Detector::expect(const std::string& key, std::vector<int> shape) {
  Raw r;
  r.shape = shape;
  raw_[key] = r;
}
void Detector::expect_conv_bn(const std::string& p, int cout, int cin, int k) {
  expect(p + ".conv.weight", {cout, cin, k, k});
  expect(p + ".bn.weight", {cout});
  expect(p + ".bn.bias", {cout});
  expect(p + ".bn.running_mean", {cout});
  expect(p + ".bn.running_var", {cout});
}

Detector::Detector(const mtgv_detector_cfg& cfg) : cfg_(cfg) {
  MTGV_CHECK(cfg.nc >= 1 && cfg.nc <= 4, ERR_INVALID, "detector: nc=%d (1..4 supported)", cfg.nc);
  MTGV_CHECK(cfg.imgsz > 0 && cfg.imgsz % 32 == 0, ERR_INVALID, "detector: imgsz=%d must be a multiple of 32", cfg.imgsz);
  MTGV_CHECK(cfg.max_batch > 0, ERR_INVALID, "detector: max_batch=%d", cfg.max_batch);
  MTGV_CHECK(cfg.max_det > 0 && cfg.max_det <= 1024, ERR_INVALID, "detector: max_det=%d", cfg.max_det);
  MTGV_CHECK(cfg.arch == 0 || cfg.arch == 8 || cfg.arch == 11, ERR_KEY, "detector: arch=%d (8: YOLOv8n-seg, 11: YOLO11n-seg)", cfg.arch);
  const int S = cfg.imgsz;
  na_ = (S / 8) * (S / 8) + (S / 16) * (S / 16) + (S / 32) * (S / 32);
  // The branch streams of the forward's fork-join are created with the handle, not at the first forward: the HIP runtime maps
  // streams to its hardware queues in creation order, and branch streams created after an application's high-priority
  // stream (mtgv.Pipeline's embed stream) ended up sharing a queue with the caller's stream - the forward on ONE stream then
  // took 5.1 instead of 2.5 ms (tools/debug/one_stream_after_overlap.py).
  for (int i = 0; i < NSIDE; ++i) {
    HIP_OK(hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&ev_fork_[i], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ev_join_[i], hipEventDisableTiming));
  }
  if (v11()) {
    head_ = "model.23";
    build_v11();
    return;
  }

  auto P = [](int i) { return "model." + std::to_string(i); };
  // backbone + neck
  struct L { int idx; char kind; int cout, n; bool sc; };  // kind: c conv, f c2f, s sppf
  const L layers[] = {{0, 'c', chn(64), 0, false},   {1, 'c', chn(128), 0, false},  {2, 'f', chn(128), rep(3), true},
                      {3, 'c', chn(256), 0, false},  {4, 'f', chn(256), rep(6), true}, {5, 'c', chn(512), 0, false},
                      {6, 'f', chn(512), rep(6), true}, {7, 'c', chn(1024), 0, false}, {8, 'f', chn(1024), rep(3), true},
                      {9, 's', chn(1024), 0, false}, {12, 'f', chn(512), rep(3), false}, {15, 'f', chn(256), rep(3), false},
                      {16, 'c', chn(256), 0, false}, {18, 'f', chn(512), rep(3), false}, {19, 'c', chn(512), 0, false},
                      {21, 'f', chn(1024), rep(3), false}};
  std::map<int, int> in_ch = {{0, 3},   {1, chn(64)},   {2, chn(128)},  {3, chn(128)},
                              {4, chn(256)}, {5, chn(256)},  {6, chn(512)},  {7, chn(512)},
                              {8, chn(1024)}, {9, chn(1024)}, {12, chn(1024) + chn(512)}, {15, chn(512) + chn(256)},
                              {16, chn(256)}, {18, chn(256) + chn(512)}, {19, chn(512)}, {21, chn(512) + chn(1024)}};
  for (const L& l : layers) {
    const int cin = in_ch[l.idx];
    if (l.kind == 'c') {
      expect_conv_bn(P(l.idx), l.cout, cin, 3);
    } else if (l.kind == 'f') {
      const int ch = l.cout / 2;
      expect_conv_bn(P(l.idx) + ".cv1", 2 * ch, cin, 1);
      expect_conv_bn(P(l.idx) + ".cv2", l.cout, (2 + l.n) * ch, 1);
      for (int j = 0; j < l.n; ++j) {
        expect_conv_bn(P(l.idx) + ".m." + std::to_string(j) + ".cv1", ch, ch, 3);
        expect_conv_bn(P(l.idx) + ".m." + std::to_string(j) + ".cv2", ch, ch, 3);
      }
      c2f_[l.idx] = {l.cout, l.n, l.sc, cin};
    } else {
      expect_conv_bn(P(l.idx) + ".cv1", cin / 2, cin, 1);
      expect_conv_bn(P(l.idx) + ".cv2", l.cout, cin / 2 * 4, 1);
    }
  }
  const int chs[3] = {chn(256), chn(512), chn(1024)};
  const int c2 = std::max(std::max(16, chs[0] / 4), reg_max_ * 4);
  const int c3 = std::max(chs[0], std::min(cfg.nc, 100));
  const int c4 = std::max(chs[0] / 4, nm_);
  MTGV_CHECK(c2 == 64 && c3 == 64 && c4 == 32, ERR_INVALID, "detector: unexpected head widths");
  const std::string H = head_;
  for (int l = 0; l < 3; ++l) {
    const std::string ls = std::to_string(l);
    expect_conv_bn(H + ".cv2." + ls + ".0", c2, chs[l], 3);
    expect_conv_bn(H + ".cv2." + ls + ".1", c2, c2, 3);
    expect(H + ".cv2." + ls + ".2.weight", {4 * reg_max_, c2, 1, 1});
    expect(H + ".cv2." + ls + ".2.bias", {4 * reg_max_});
    expect_conv_bn(H + ".cv3." + ls + ".0", c3, chs[l], 3);
    expect_conv_bn(H + ".cv3." + ls + ".1", c3, c3, 3);
    expect(H + ".cv3." + ls + ".2.weight", {cfg.nc, c3, 1, 1});
    expect(H + ".cv3." + ls + ".2.bias", {cfg.nc});
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

Detector::~Detector() {
  for (float* p : dev_allocs_) {
    gemm_split_unregister(p);
    (void)hipFree(p);
  }
  if (nms_ws_) (void)hipFree(nms_ws_);
  for (int i = 0; i < NSIDE; ++i) {
    if (side_[i]) (void)hipStreamDestroy(side_[i]);
    if (ev_fork_[i]) (void)hipEventDestroy(ev_fork_[i]);
    if (ev_join_[i]) (void)hipEventDestroy(ev_join_[i]);
  }
}

bool Detector::fork_enabled() const {
  if (count_flops_) return false;
  if (fork_mode_ >= 0) return fork_mode_ != 0;  // mtgv_detector_set_fork
  const char* e = getenv("MTGV_DET_FORK");  // read per call: tests and tools compare both schedules in one process
  return e == nullptr || atoi(e) != 0;
}

hipStream_t Detector::fork_after(hipStream_t s, int i) {
  if (!fork_enabled()) return s;
  HIP_OK(hipEventRecord(ev_fork_[i], s));
  HIP_OK(hipStreamWaitEvent(side_[i], ev_fork_[i], 0));
  side_busy_[i] = true;
  return side_[i];
}

void Detector::join_into(hipStream_t s, int i) {
  if (!side_busy_[i]) return;
  HIP_OK(hipEventRecord(ev_join_[i], side_[i]));
  HIP_OK(hipStreamWaitEvent(s, ev_join_[i], 0));
  side_busy_[i] = false;
}

void Detector::set_param(const char* key, const float* host, int64_t numel) {
  auto it = raw_.find(key);
  MTGV_CHECK(it != raw_.end(), ERR_KEY, "unknown detector parameter key '%s'", key);
  int64_t want = 1;
  for (int d : it->second.shape) want *= d;
  MTGV_CHECK(numel == want, ERR_INVALID, "parameter %s: got %lld elements, expected %lld", key, (long long)numel, (long long)want);
  it->second.data.assign(host, host + numel);
  it->second.set = true;
  finalized_ = false;
}

int Detector::missing() const {
  int m = 0;
  for (auto& kv : raw_) m += kv.second.set ? 0 : 1;
  return m;
}

float* Detector::upload(const std::vector<float>& v, int row_k) {
  float* d = nullptr;
  HIP_OK(hipMalloc((void**)&d, std::max<size_t>(v.size(), 4) * sizeof(float)));
  HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  dev_allocs_.push_back(d);
  gemm_split_register(d, v.size(), row_k);  // weights become pre-split B operands for the f16x3 GEMMs (small vectors are skipped)
  gemm_split_refresh(d, 0, v.size() % 4 == 0 ? v.size() : 0, nullptr);
  HIP_OK(hipStreamSynchronize(nullptr));
  return d;
}

// Conv2d(bias=False) + BatchNorm2d(eps=1e-3) -> weight [cout][k][k][cin_pad], bias [cout]
ConvW Detector::fold(const std::string& p, int cin_pad) {
  const Raw& w = raw_.at(p + ".conv.weight");
  const int cout = w.shape[0], cin = w.shape[1], k = w.shape[2];
  const int cp = cin_pad > 0 ? cin_pad : cin;
  const auto& g = raw_.at(p + ".bn.weight").data;
  const auto& b = raw_.at(p + ".bn.bias").data;
  const auto& mu = raw_.at(p + ".bn.running_mean").data;
  const auto& var = raw_.at(p + ".bn.running_var").data;
  std::vector<float> wf((size_t)cout * k * k * cp, 0.f), bf(cout);
  for (int o = 0; o < cout; ++o) {
    const double sc = (double)g[o] / sqrt((double)var[o] + 1e-3);
    bf[o] = (float)((double)b[o] - (double)mu[o] * sc);
    for (int i = 0; i < cin; ++i)
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw)
          wf[(((size_t)o * k + kh) * k + kw) * cp + i] = (float)((double)w.data[(((size_t)o * cin + i) * k + kh) * k + kw] * sc);
  }
  ConvW c;
  c.w = upload(wf, k * k * cp), c.b = upload(bf), c.cout = cout, c.cin = cp, c.k = k;
  return c;
}

ConvW Detector::plain(const std::string& p) {
  const Raw& w = raw_.at(p + ".weight");
  const int cout = w.shape[0], cin = w.shape[1], k = w.shape[2];
  std::vector<float> wf((size_t)cout * k * k * cin);
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i)
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw)
          wf[(((size_t)o * k + kh) * k + kw) * cin + i] = w.data[(((size_t)o * cin + i) * k + kh) * k + kw];
  ConvW c;
  c.w = upload(wf, k * k * cin), c.b = upload(raw_.at(p + ".bias").data), c.cout = cout, c.cin = cin, c.k = k;
  return c;
}

View Detector::take(int n, int h, int w, int c) {
  const size_t fl = ((size_t)n * h * w * c + 63) / 64 * 64;
  MTGV_CHECK(arena_used_ + fl <= arena_.n, ERR_RUNTIME, "detector arena exhausted");
  View v;
  v.p = arena_.p + arena_used_;
  v.H = h, v.W = w, v.ct = c, v.co = 0, v.C = c;
  arena_used_ += fl;
  return v;
}

// activation view by name, in the format of the current forward (fmt_)
View Detector::view(const std::string& k) const {
  View v = v_.at(k);
  v.fmt = (k == "x0" || k == "protos") ? 0 : fmt_;
  return v;
}

void Detector::finalize() {
  MTGV_CHECK(missing() == 0, ERR_RUNTIME, "detector has %d unset parameters", missing());
  if (finalized_) return;
  for (float* p : dev_allocs_) {
    gemm_split_unregister(p);
    (void)hipFree(p);
  }
  dev_allocs_.clear();
  cw_.clear();
  // every Conv+BN key prefix
  for (auto& kv : raw_) {
    const std::string& k = kv.first;
    const std::string suf = ".conv.weight";
    if (k.size() > suf.size() && k.compare(k.size() - suf.size(), suf.size(), suf) == 0) {
      const std::string pre = k.substr(0, k.size() - suf.size());
      if (raw_.find(pre + ".bn.weight") == raw_.end()) continue;  // dfl.conv has no BatchNorm
      const Raw& wr = kv.second;
      if (wr.shape[1] == 1 && wr.shape[0] > 1 && wr.shape[2] == 3) {
        cw_[pre] = fold_dw(pre);  // depthwise 3x3 (YOLO11 class branch, attention positional encoding)
        continue;
      }
      cw_[pre] = fold(pre, pre == "model.0" ? 4 : 0);
    }
  }
  const std::string H = head_;
  for (int l = 0; l < 3 && !v11(); ++l) {
    const std::string ls = std::to_string(l);
    // the three branches' first 3x3 convs read the same input: one conv with 64+64+32 outputs
    const ConvW &a = cw_.at(H + ".cv2." + ls + ".0"), &b = cw_.at(H + ".cv3." + ls + ".0"), &c = cw_.at(H + ".cv4." + ls + ".0");
    const size_t per = (size_t)9 * a.cin;
    std::vector<float> w((size_t)(a.cout + b.cout + c.cout) * per), bias(a.cout + b.cout + c.cout);
    size_t wo = 0, bo = 0;
    for (const ConvW* q : {&a, &b, &c}) {
      HIP_OK(hipMemcpy(w.data() + wo, q->w, (size_t)q->cout * per * sizeof(float), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(bias.data() + bo, q->b, (size_t)q->cout * sizeof(float), hipMemcpyDeviceToHost));
      wo += (size_t)q->cout * per, bo += q->cout;
    }
    head_first_[l].w = upload(w, (int)per), head_first_[l].b = upload(bias);
    head_first_[l].cout = a.cout + b.cout + c.cout, head_first_[l].cin = a.cin, head_first_[l].k = 3;
    head_box2_[l] = cw_.at(H + ".cv2." + ls + ".1");
    head_cls2_[l] = cw_.at(H + ".cv3." + ls + ".1");
    head_coef2_[l] = cw_.at(H + ".cv4." + ls + ".1");
    head_box3_[l] = plain(H + ".cv2." + ls + ".2");
    head_cls3_[l] = plain(H + ".cv3." + ls + ".2");
    head_coef3_[l] = plain(H + ".cv4." + ls + ".2");
  }
  for (int l = 0; l < 3 && v11(); ++l) {
    const std::string ls = std::to_string(l);
    // box and coefficient branches start with a 3x3 conv on the same input: one conv with 64+32 outputs
    const ConvW &a = cw_.at(H + ".cv2." + ls + ".0"), &c = cw_.at(H + ".cv4." + ls + ".0");
    const size_t per = (size_t)9 * a.cin;
    std::vector<float> w((size_t)(a.cout + c.cout) * per), bias(a.cout + c.cout);
    size_t wo = 0, bo = 0;
    for (const ConvW* q : {&a, &c}) {
      HIP_OK(hipMemcpy(w.data() + wo, q->w, (size_t)q->cout * per * sizeof(float), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(bias.data() + bo, q->b, (size_t)q->cout * sizeof(float), hipMemcpyDeviceToHost));
      wo += (size_t)q->cout * per, bo += q->cout;
    }
    head_bc_[l].w = upload(w, (int)per), head_bc_[l].b = upload(bias);
    head_bc_[l].cout = a.cout + c.cout, head_bc_[l].cin = a.cin, head_bc_[l].k = 3;
    head_box2_[l] = cw_.at(H + ".cv2." + ls + ".1");
    head_coef2_[l] = cw_.at(H + ".cv4." + ls + ".1");
    cls_dw1_[l] = cw_.at(H + ".cv3." + ls + ".0.0"), cls_pw1_[l] = cw_.at(H + ".cv3." + ls + ".0.1");
    cls_dw2_[l] = cw_.at(H + ".cv3." + ls + ".1.0"), cls_pw2_[l] = cw_.at(H + ".cv3." + ls + ".1.1");
    head_box3_[l] = plain(H + ".cv2." + ls + ".2");
    head_cls3_[l] = plain(H + ".cv3." + ls + ".2");
    head_coef3_[l] = plain(H + ".cv4." + ls + ".2");
  }
  // DFL weights must be arange(16) (they are a fixed buffer upstream); the decode kernel hard-codes them
  {
    const auto& d = raw_.at(H + ".dfl.conv.weight").data;
    for (int i = 0; i < reg_max_; ++i) MTGV_CHECK(d[i] == (float)i, ERR_INVALID, "dfl.conv.weight is not arange(16)");
  }
  // ConvTranspose2d(k2,s2): weight (in, out, kh, kw) -> four [out][in] matrices
  {
    const Raw& w = raw_.at(H + ".proto.upsample.weight");
    const int ci = w.shape[0], co = w.shape[1];
    float* bias = upload(raw_.at(H + ".proto.upsample.bias").data);
    for (int kh = 0; kh < 2; ++kh)
      for (int kw = 0; kw < 2; ++kw) {
        std::vector<float> m((size_t)co * ci);
        for (int o = 0; o < co; ++o)
          for (int i = 0; i < ci; ++i) m[(size_t)o * ci + i] = w.data[(((size_t)i * co + o) * 2 + kh) * 2 + kw];
        ConvW c;
        c.w = upload(m, ci), c.b = bias, c.cout = co, c.cin = ci, c.k = 1;
        proto_up_[kh * 2 + kw] = c;
      }
    // all four phases as one [4 co][ci] operand, rows (kh, kw, o): one launch reads the input once (proto())
    std::vector<float> m4((size_t)4 * co * ci), b4((size_t)4 * co);
    const auto& bsrc = raw_.at(H + ".proto.upsample.bias").data;
    for (int q = 0; q < 4; ++q)
      for (int o = 0; o < co; ++o) {
        b4[(size_t)q * co + o] = bsrc[o];
        for (int i = 0; i < ci; ++i) m4[((size_t)q * co + o) * ci + i] = w.data[(((size_t)i * co + o) * 2 + (q >> 1)) * 2 + (q & 1)];
      }
    proto_up_all_.w = upload(m4, ci), proto_up_all_.b = upload(b4), proto_up_all_.cout = 4 * co, proto_up_all_.cin = ci, proto_up_all_.k = 1;
  }

  if (v11()) {
    arena_v11();
  } else {
  // activation arena for max_batch
  const int nb = cfg_.max_batch, S = cfg_.imgsz;
  const int s2 = S / 2, s4 = S / 4, s8 = S / 8, s16 = S / 16, s32 = S / 32;
  const int c16 = chn(64), c32 = chn(128), c64 = chn(256), c128 = chn(512), c256 = chn(1024);
  size_t total = 0;
  auto sz = [&](int h, int w, int c) { total += ((size_t)nb * h * w * c + 63) / 64 * 64; };
  sz(S, S, 4), sz(s2, s2, c16), sz(s4, s4, c32);
  sz(s4, s4, 3 * c32 / 2), sz(s4, s4, c32 / 2), sz(s4, s4, c32);       // node 2
  sz(s8, s8, c64);                                                      // 3
  sz(s8, s8, 4 * c64 / 2), sz(s8, s8, c64 / 2);                         // 4
  sz(s8, s8, c128 + c64);                                               // cat14
  sz(s16, s16, c128);                                                   // 5
  sz(s16, s16, 4 * c128 / 2), sz(s16, s16, c128 / 2);                   // 6
  sz(s16, s16, c256 + c128);                                            // cat11
  sz(s32, s32, c256);                                                   // 7
  sz(s32, s32, 3 * c256 / 2), sz(s32, s32, c256 / 2), sz(s32, s32, c256);  // 8
  sz(s32, s32, 2 * c256);                                               // sppcat
  sz(s32, s32, c128 + c256);                                            // cat20
  sz(s16, s16, 3 * c128 / 2), sz(s16, s16, c128 / 2);                   // 12
  sz(s16, s16, c64 + c128);                                             // cat17
  sz(s8, s8, 3 * c64 / 2), sz(s8, s8, c64 / 2), sz(s8, s8, c64);        // 15, p3
  sz(s16, s16, 3 * c128 / 2), sz(s16, s16, c128 / 2), sz(s16, s16, c128);  // 18, p4
  sz(s32, s32, 3 * c256 / 2), sz(s32, s32, c256 / 2), sz(s32, s32, c256);  // 21, p5
  sz(s8, s8, 160), sz(s8, s8, 160), sz(s16, s16, 160), sz(s16, s16, 160), sz(s32, s32, 160), sz(s32, s32, 160);  // head t1/t2 per level (the levels' branches run concurrently)
  sz(s8, s8, RAW_CT), sz(s16, s16, RAW_CT), sz(s32, s32, RAW_CT);       // rawhead
  sz(s8, s8, npr_), sz(s4, s4, npr_), sz(s4, s4, npr_), sz(s4, s4, nm_);  // proto
  sz(1, na_, 4 + cfg_.nc + nm_);                                        // pred
  sz(1, cfg_.max_det, nm_);                                             // coef
  arena_.alloc(total + 1024);
  arena_used_ = 0;
  v_.clear();
  v_["x0"] = take(nb, S, S, 4);
  v_["l0"] = take(nb, s2, s2, c16);
  v_["l1"] = take(nb, s4, s4, c32);
  v_["cat2"] = take(nb, s4, s4, 3 * c32 / 2), v_["tmp2"] = take(nb, s4, s4, c32 / 2), v_["l2"] = take(nb, s4, s4, c32);
  v_["l3"] = take(nb, s8, s8, c64);
  v_["cat4"] = take(nb, s8, s8, 4 * c64 / 2), v_["tmp4"] = take(nb, s8, s8, c64 / 2);
  v_["cat14"] = take(nb, s8, s8, c128 + c64);
  v_["l5"] = take(nb, s16, s16, c128);
  v_["cat6"] = take(nb, s16, s16, 4 * c128 / 2), v_["tmp6"] = take(nb, s16, s16, c128 / 2);
  v_["cat11"] = take(nb, s16, s16, c256 + c128);
  v_["l7"] = take(nb, s32, s32, c256);
  v_["cat8"] = take(nb, s32, s32, 3 * c256 / 2), v_["tmp8"] = take(nb, s32, s32, c256 / 2), v_["l8"] = take(nb, s32, s32, c256);
  v_["sppcat"] = take(nb, s32, s32, 2 * c256);
  v_["cat20"] = take(nb, s32, s32, c128 + c256);
  v_["cat12"] = take(nb, s16, s16, 3 * c128 / 2), v_["tmp12"] = take(nb, s16, s16, c128 / 2);
  v_["cat17"] = take(nb, s16, s16, c64 + c128);
  v_["cat15"] = take(nb, s8, s8, 3 * c64 / 2), v_["tmp15"] = take(nb, s8, s8, c64 / 2), v_["p3"] = take(nb, s8, s8, c64);
  v_["cat18"] = take(nb, s16, s16, 3 * c128 / 2), v_["tmp18"] = take(nb, s16, s16, c128 / 2), v_["p4"] = take(nb, s16, s16, c128);
  v_["cat21"] = take(nb, s32, s32, 3 * c256 / 2), v_["tmp21"] = take(nb, s32, s32, c256 / 2), v_["p5"] = take(nb, s32, s32, c256);
  v_["t1_0"] = take(nb, s8, s8, 160), v_["t2_0"] = take(nb, s8, s8, 160);
  v_["t1_1"] = take(nb, s16, s16, 160), v_["t2_1"] = take(nb, s16, s16, 160);
  v_["t1_2"] = take(nb, s32, s32, 160), v_["t2_2"] = take(nb, s32, s32, 160);
  rawhead_[0] = take(nb, s8, s8, RAW_CT).p, rawhead_[1] = take(nb, s16, s16, RAW_CT).p, rawhead_[2] = take(nb, s32, s32, RAW_CT).p;
  v_["pr1"] = take(nb, s8, s8, npr_), v_["pr2"] = take(nb, s4, s4, npr_), v_["pr3"] = take(nb, s4, s4, npr_);
  v_["protos"] = take(nb, s4, s4, nm_);
  pred_ = take(nb, 1, na_, 4 + cfg_.nc + nm_).p;
  coef_ = take(nb, 1, cfg_.max_det, nm_).p;
  }
  // rawhead class padding column (index 67) is never written by a conv; keep it defined
  HIP_OK(hipMemset(arena_.p, 0, arena_.n * sizeof(float)));
  if (nms_ws_) (void)hipFree(nms_ws_);
  nms_ws_bytes_ = nms_workspace_bytes(cfg_.max_batch, na_);
  HIP_OK(hipMalloc((void**)&nms_ws_, nms_ws_bytes_));
  finalized_ = true;

  // algorithmic FLOPs of one frame: run the plan once in counting mode
  count_flops_ = true;
  flops_ = 0;
  forward(nullptr, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
  count_flops_ = false;
}

void Detector::conv(const ConvW& w, const View& in, const View& out, int stride, int act, const View* res, int n, hipStream_t s) {
  MTGV_CHECK(in.C == w.cin && out.C == w.cout, ERR_RUNTIME, "detector: conv channel mismatch (%d->%d vs %d->%d)", in.C, out.C, w.cin,
             w.cout);
  GemmArgs g;
  g.A = in.p, g.W = w.w, g.bias = w.b, g.Out = out.p;
  g.M = n * out.H * out.W, g.N = w.cout, g.K = w.k * w.k * w.cin;
  g.H = in.H, g.Wd = in.W, g.c_total = in.ct, g.c_off = in.co, g.Cin = in.C;
  g.KH = w.k, g.KW = w.k, g.stride = stride, g.pad = w.k / 2;
  g.OH = out.H, g.OW = out.W, g.OH2 = out.H, g.OW2 = out.W;
  g.ldo = out.ct, g.o_off = out.co;
  g.act = act;
  g.a_fmt = in.fmt, g.out_fmt = out.fmt;
  if (res) g.res = res->p + res->co, g.ldr = res->ct, g.res_fmt = res->fmt;
  if (count_flops_) {
    // model.0 is stored with a zero 4th input channel; count the real 3
    const double kk = (&w == &cw_.at("model.0")) ? 27.0 : (double)g.K;
    flops_ += 2.0 * g.M * g.N * kk;
    return;
  }
  gemm_launch(g, gemm_plan(g.M, g.N, g.K, act != ACT_NONE), s);
}

void Detector::conv_pair(const ConvW& w1, const View& in, const View& mid, int stride, const ConvW& w2, const View& out2, int act2, int n,
                         hipStream_t s) {
  MTGV_CHECK(in.C == w1.cin && mid.C == w1.cout && w2.cin == w1.cout && w2.k == 1 && out2.C == w2.cout, ERR_RUNTIME,
             "detector: conv pair channel mismatch (%d->%d, %d->%d)", w1.cin, w1.cout, w2.cin, w2.cout);
  const char* const ce = getenv("MTGV_DET_CHAIN");  // read per call (A/B in one process); 0: two launches
  if (!count_flops_ && fmt_ == 1 && !(ce != nullptr && atoi(ce) == 0)) {
    GemmArgs g;
    g.A = in.p, g.W = w1.w, g.bias = w1.b, g.Out = nullptr;
    g.M = n * mid.H * mid.W, g.N = w1.cout, g.K = w1.k * w1.k * w1.cin;
    g.H = in.H, g.Wd = in.W, g.c_total = in.ct, g.c_off = in.co, g.Cin = in.C;
    g.KH = w1.k, g.KW = w1.k, g.stride = stride, g.pad = w1.k / 2;
    g.OH = mid.H, g.OW = mid.W, g.OH2 = mid.H, g.OW2 = mid.W;
    g.ldo = mid.ct, g.o_off = mid.co;
    g.act = ACT_SILU;
    g.a_fmt = in.fmt, g.out_fmt = mid.fmt;
    g.W2 = w2.w, g.bias2 = w2.b, g.Out2 = out2.p, g.N2 = w2.cout, g.ldo2 = out2.ct, g.o_off2 = out2.co, g.out_fmt2 = out2.fmt, g.act2 = act2;
    if (gemm_sp_chain_ok(g)) {
      gemm_launch(g, gemm_plan(g.M, g.N, g.K, true), s);
      return;
    }
  }
  conv(w1, in, mid, stride, ACT_SILU, nullptr, n, s);
  conv(w2, mid, out2, 1, act2, nullptr, n, s);
}

// C2f: cv1 -> 2 chunks; n bottlenecks (3x3,3x3, +shortcut) each appended; cv2 over the concat
void Detector::c2f(int idx, const View& in, const View& out, int n, hipStream_t s, const ConvW* pre, const View* pre_in) {
  const C2fInfo& ci = c2f_.at(idx);
  const int ch = ci.cout / 2;
  const std::string P = "model." + std::to_string(idx);
  const View cat = view("cat" + std::to_string(idx));
  const View tmp = view("tmp" + std::to_string(idx));
  // pre: the stride-2 Conv in front of this block, whose only consumer is cv1 - the pair runs as one launch where it can
  if (pre != nullptr) conv_pair(*pre, *pre_in, in, 2, cw_.at(P + ".cv1"), cat.slice(0, 2 * ch), ACT_SILU, n, s);
  else conv(cw_.at(P + ".cv1"), in, cat.slice(0, 2 * ch), 1, ACT_SILU, nullptr, n, s);
  for (int j = 0; j < ci.n; ++j) {
    const View src = cat.slice((1 + j) * ch, ch);
    const View dst = cat.slice((2 + j) * ch, ch);
    const std::string M = P + ".m." + std::to_string(j);
    conv(cw_.at(M + ".cv1"), src, tmp, 1, ACT_SILU, nullptr, n, s);
    conv(cw_.at(M + ".cv2"), tmp, dst, 1, ACT_SILU, ci.shortcut ? &src : nullptr, n, s);
  }
  conv(cw_.at(P + ".cv2"), cat.slice(0, (2 + ci.n) * ch), out, 1, ACT_SILU, nullptr, n, s);
}

// model.0 (Conv 3 -> 16, k3 s2): on its own kernel straight from the uint8 frame, or (counting mode, f32 debugging
// switch) through the float copy and the implicit GEMM
void Detector::conv0(const uint8_t* frames, int n, int flip, hipStream_t s) {
  const int S = cfg_.imgsz;
  if (fmt_ == 1 || (!count_flops_ && getenv("MTGV_CONV0_GEMM") == nullptr)) {
    const ConvW& w0 = cw_.at("model.0");
    const View l0 = view("l0");
    const long total = (long)n * (S / 2) * (S / 8);
    MTGV_CHECK((S / 2) % 4 == 0 && w0.cout == 16 && w0.cin == 4 && w0.k == 3, ERR_RUNTIME, "detector: unexpected model.0 geometry");
    MTGV_CHECK(((uintptr_t)frames & 7) == 0, ERR_INVALID, "detector: the frame buffer must be 8-byte aligned");
    if (fmt_ == 1)
      hipLaunchKernelGGL((conv0_u8_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, w0.w, w0.b, l0.p, S, flip, total);
    else
      hipLaunchKernelGGL((conv0_u8_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, w0.w, w0.b, l0.p, S, flip, total);
    HIP_OK(hipGetLastError());
  } else {
    if (!count_flops_) u8_to_f32_launch(frames, view("x0").p, (long)n * S * S, 3, 4, 1.0f, 0.0f, flip, s);
    conv(cw_.at("model.0"), view("x0"), view("l0"), 2, ACT_SILU, nullptr, n, s);
  }
}

// Proto: Conv3 -> ConvTranspose2d(k2,s2) as four scattered 1x1 GEMMs -> Conv3 -> Conv1
void Detector::proto(const std::string& H, const View& p3, int n, hipStream_t s) {
  conv(cw_.at(H + ".proto.cv1"), p3, view("pr1"), 1, ACT_SILU, nullptr, n, s);
  {
    const View in = view("pr1"), out = view("pr2");
    // SP8 activations (LDS-DMA kernel): one launch with N = 4 * 64 columns whose epilogue scatters column group q to
    // output phase (q / 2, q % 2) - the input is read once instead of four times (round 3: 4 x 27 us at 3.9 TB/s, bound
    // by that re-read).  Same products in the same order per output element: bit-identical to the four launches.
    const char* const up1 = getenv("MTGV_PROTO_UP1");  // read per call (A/B in one process); 0: the four-launch form
    const bool one_launch = up1 == nullptr || atoi(up1) != 0;
    const bool single = one_launch && fmt_ == 1 && !count_flops_ && npr_ % 8 == 0;
    if (single) {
      const ConvW& w = proto_up_all_;
      GemmArgs g;
      g.A = in.p, g.W = w.w, g.bias = w.b, g.Out = out.p;
      g.M = n * in.H * in.W, g.N = w.cout, g.K = w.cin;
      g.H = in.H, g.Wd = in.W, g.c_total = in.ct, g.c_off = 0, g.Cin = w.cin;
      g.OH = in.H, g.OW = in.W;
      g.os = 2, g.oy = 0, g.ox = 0, g.os_nq = npr_, g.OH2 = out.H, g.OW2 = out.W;
      g.ldo = out.ct;
      g.a_fmt = in.fmt, g.out_fmt = out.fmt;
      gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
    }
    for (int q = 0; q < 4 && !single; ++q) {
      const ConvW& w = proto_up_[q];
      GemmArgs g;
      g.A = in.p, g.W = w.w, g.bias = w.b, g.Out = out.p;
      g.M = n * in.H * in.W, g.N = w.cout, g.K = w.cin;
      g.H = in.H, g.Wd = in.W, g.c_total = in.ct, g.c_off = 0, g.Cin = w.cin;
      g.OH = in.H, g.OW = in.W;
      g.os = 2, g.oy = q >> 1, g.ox = q & 1, g.OH2 = out.H, g.OW2 = out.W;
      g.ldo = out.ct;
      g.a_fmt = in.fmt, g.out_fmt = out.fmt;
      if (count_flops_)
        flops_ += 2.0 * g.M * g.N * g.K;
      else
        gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
    }
  }
  conv_pair(cw_.at(H + ".proto.cv2"), view("pr2"), view("pr3"), 1, cw_.at(H + ".proto.cv3"), view("protos"), ACT_SILU, n, s);
}

// Mask logits of a few detections per frame (process_mask + crop_mask behind od_export.py:152): out[z][m][px] =
// <coef[z][m], protos[z][px]> inside box m, 0 outside - f32 FMA chain in k order.  One thread per prototype pixel reads
// its 32 channels once (128 contiguous bytes) and serves all the frame's kept rows; coefficients and scaled boxes sit in
// LDS.  Rows beyond n_det[z] are written as zeros (empty masks).
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ coef, const float* __restrict__ protos,
                                                         const int* __restrict__ n_det, const float* __restrict__ boxes,
                                                         float* __restrict__ out, int npx, int pw, int mask_rows, int max_det,
                                                         float crop_scale) {
  __shared__ __attribute__((aligned(16))) float sc[16 * 32];
  __shared__ float sb[16 * 4];
  const int z = blockIdx.y;
  const int mc = n_det[z] < mask_rows ? n_det[z] : mask_rows;
  for (int i = threadIdx.x; i < mc * 32; i += 256) sc[i] = coef[(long)z * max_det * 32 + i];
  if (threadIdx.x < mc * 4) sb[threadIdx.x] = __fmul_rn(boxes[(long)z * max_det * 4 + threadIdx.x], crop_scale);
  __syncthreads();
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= npx) return;
  f32x4 p[8];
  const f32x4* src = reinterpret_cast<const f32x4*>(protos + ((long)z * npx + px) * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) p[q] = src[q];
  const int py = px / pw;
  const float fx = (float)(px - py * pw), fy = (float)py;
  for (int m = 0; m < mc; ++m) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(&sc[m * 32 + q * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(c[e], p[q][e], acc);
    }
    const bool inside = fx >= sb[m * 4] && fx < sb[m * 4 + 2] && fy >= sb[m * 4 + 1] && fy < sb[m * 4 + 3];
    out[((long)z * mask_rows + m) * npx + px] = inside ? acc : 0.f;
  }
  for (int m = mc; m < mask_rows; ++m) out[((long)z * mask_rows + m) * npx + px] = 0.f;
}

// decode -> NMS -> mask logits of the kept detections
void Detector::head_tail(int n, int* n_det, float* boxes, float* conf, int* cls, int* keep_idx, float* mask_logits, int mask_rows,
                         hipStream_t s) {
  const int S = cfg_.imgsz;
  const long tot = (long)n * na_;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rawhead_[0], rawhead_[1], rawhead_[2],
                     pred_, n, cfg_.nc, nm_, S, na_);
  HIP_OK(hipGetLastError());
  nms_launch(pred_, n, cfg_.nc, nm_, na_, cfg_.conf, cfg_.iou, cfg_.max_det, 7680.0f, n_det, boxes, conf, cls, keep_idx, coef_,
             nms_ws_, nms_ws_bytes_, s);
  join_into(s, 0);  // the prototype branch ran beside the heads, decode and NMS (one workgroup per frame: 32 of 256 CUs)
  if (mask_logits != nullptr) {
    // masks = coeffs @ protos^T per image, cropped to the box (process_mask / crop_mask)
    const View pr = view("protos");
    const int npx = pr.H * pr.W;
    if (nm_ == 32 && mask_rows <= 16 && !count_flops_) {  // a handful of masks per frame: one pass over the prototypes
      hipLaunchKernelGGL(mask_logits_kernel, dim3((unsigned)((npx + 255) / 256), (unsigned)n), dim3(256), 0, s, coef_, pr.p, n_det, boxes,
                         mask_logits, npx, pr.W, mask_rows, cfg_.max_det, (float)pr.W / (float)S);
      HIP_OK(hipGetLastError());
      return;
    }
    // the batched GEMM writes only the rows of kept detections: the rest is cleared first
    HIP_OK(hipMemsetAsync(mask_logits, 0, (size_t)n * mask_rows * npx * sizeof(float), s));
    GemmArgs g = linear_args(coef_, nm_, pr.p, nullptr, mask_logits, npx, mask_rows, npx, nm_, ACT_NONE);
    g.batch = n;
    g.strideA = (long)cfg_.max_det * nm_;
    g.strideW = (long)npx * nm_;
    g.strideO = (long)mask_rows * npx;
    g.m_count = n_det;
    g.crop_boxes = boxes;
    g.crop_rows = cfg_.max_det;
    g.crop_scale = (float)pr.W / (float)S;
    g.crop_w = pr.W;
    gemm_launch(g, gemm_plan(g.M, g.N, g.K), s);
  }
}

void Detector::forward(const uint8_t* frames, int n, int flip, int* n_det, float* boxes, float* conf, int* cls, int* keep_idx,
                       float* mask_logits, int mask_rows, hipStream_t s) {
  MTGV_CHECK(finalized_, ERR_RUNTIME, "detector: finalize() has not been called");
  if (!count_flops_) {
    MTGV_CHECK(n > 0 && n <= cfg_.max_batch, ERR_INVALID, "batch %d outside [1, %d]", n, cfg_.max_batch);
    MTGV_CHECK(frames && n_det && boxes && conf && cls && keep_idx, ERR_INVALID, "null tensor");
    MTGV_CHECK(mask_logits == nullptr || (mask_rows > 0 && mask_rows <= cfg_.max_det), ERR_INVALID, "mask_rows=%d", mask_rows);
  }
  // f16x3 on the LDS-DMA kernel: every intermediate activation is kept in SP8; the frame, the raw head rows and the
  // prototypes (decode / mask inputs) stay f32
  fmt_ = (!count_flops_ && gemm_sp_active()) ? 1 : 0;
  if (v11()) {
    forward_v11(frames, n, flip, s);
  } else {
    forward_v8(frames, n, flip, s);
  }
  if (count_flops_) return;
  head_tail(n, n_det, boxes, conf, cls, keep_idx, mask_logits, mask_rows, s);
  last_n_ = n;
}

// SPPF: cv1, three chained 5x5 max pools, cv2 over the concat
void Detector::sppf(const std::string& P, const View& in, const View& spp, const View& out, int n, hipStream_t s) {
  const int ch = spp.ct / 4;
  conv(cw_.at(P + ".cv1"), in, spp.slice(0, ch), 1, ACT_SILU, nullptr, n, s);
  const size_t pools_lds = (size_t)spp.H * spp.W * 64;  // two images of (hi, lo) pieces
  const char* const pools_env = getenv("MTGV_SPPF_POOLS1");  // read per call (A/B in one process); 0: three launches
  const bool pools1 = pools_env == nullptr || atoi(pools_env) != 0;
  if (!count_flops_ && fmt_ == 1 && pools1 && pools_lds <= 64 * 1024 && ch % 8 == 0) {
    hipLaunchKernelGGL(sppf_pools_sp8_kernel, dim3((unsigned)(n * (ch / 8))), dim3(256), pools_lds, s, spp.p, spp.ct, ch, spp.H, spp.W);
    HIP_OK(hipGetLastError());
  } else if (!count_flops_)
    for (int i = 0; i < 3; ++i) {
      if (fmt_ == 1) {
        const long total = (long)n * spp.H * spp.W * (ch / 8);
        hipLaunchKernelGGL(maxpool5_sp8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, spp.p, spp.ct, i * ch, spp.p,
                           spp.ct, (i + 1) * ch, spp.H, spp.W, ch, total);
        HIP_OK(hipGetLastError());
      } else {
        maxpool5_launch(spp.p, spp.ct, i * ch, spp.p, spp.ct, (i + 1) * ch, n, spp.H, spp.W, ch, s);
      }
    }
  conv(cw_.at(P + ".cv2"), spp, out, 1, ACT_SILU, nullptr, n, s);
}

void Detector::forward_v8(const uint8_t* frames, int n, int flip, hipStream_t s) {
  const int c64 = chn(256), c128 = chn(512), c256 = chn(1024);
  auto V = [&](const char* k) -> View { return view(k); };
  conv0(frames, n, flip, s);
  const View l0 = V("l0");
  c2f(2, V("l1"), V("l2"), n, s, &cw_.at("model.1"), &l0);  // model.1 (3x3 s2) + cv1 in one launch: l1 is never stored
  // (model.3 + node 4's cv1 measured 6 us SLOWER chained - 94 vs 61 + 27 - and stay two launches)
  conv(cw_.at("model.3"), V("l2"), V("l3"), 2, ACT_SILU, nullptr, n, s);
  const View n4 = V("cat14").slice(c128, c64);      // node 4 output lives in concat 14 = [up(12), 4]
  c2f(4, V("l3"), n4, n, s);
  conv(cw_.at("model.5"), n4, V("l5"), 2, ACT_SILU, nullptr, n, s);
  const View n6 = V("cat11").slice(c256, c128);     // concat 11 = [up(9), 6]
  c2f(6, V("l5"), n6, n, s);
  conv(cw_.at("model.7"), n6, V("l7"), 2, ACT_SILU, nullptr, n, s);
  c2f(8, V("l7"), V("l8"), n, s);
  const View n9 = V("cat20").slice(c128, c256);     // concat 20 = [19, 9]
  sppf("model.9", V("l8"), V("sppcat"), n9, n, s);
  // top-down
  const View cat11 = V("cat11"), cat14 = V("cat14"), cat17 = V("cat17"), cat20 = V("cat20");
  if (!count_flops_) upsample2x_launch(n9.p, n9.ct, n9.co, cat11.p, cat11.ct, 0, n, n9.H, n9.W, c256, s);
  const View n12 = cat17.slice(c64, c128);          // concat 17 = [16, 12]
  c2f(12, cat11, n12, n, s);
  if (!count_flops_) upsample2x_launch(n12.p, n12.ct, n12.co, cat14.p, cat14.ct, 0, n, n12.H, n12.W, c128, s);
  c2f(15, cat14, V("p3"), n, s);
  // P3 exists: the prototype branch (0.5 ms of chip-filling launches) and the P3 head leave the caller's stream; the
  // rest of the neck - 100..400-tile launches that cannot fill 256 CUs on their own - runs beside them
  proto(head_, V("p3"), n, fork_after(s, 0));
  head_level_v8(0, n, fork_after(s, 1));
  conv(cw_.at("model.16"), V("p3"), cat17.slice(0, c64), 2, ACT_SILU, nullptr, n, s);
  c2f(18, cat17, V("p4"), n, s);
  head_level_v8(1, n, fork_after(s, 2));
  conv(cw_.at("model.19"), V("p4"), cat20.slice(0, c128), 2, ACT_SILU, nullptr, n, s);
  c2f(21, cat20, V("p5"), n, s);
  head_level_v8(2, n, s);
  join_into(s, 1), join_into(s, 2);  // (the prototype branch is joined in head_tail, after decode + NMS)
}

// Segment head of level l (P3 / P4 / P5): the three branches' first 3x3 convs as one launch, then per branch 3x3 -> 1x1
void Detector::head_level_v8(int l, int n, hipStream_t s) {
  const char* feats[3] = {"p3", "p4", "p5"};
  const std::string ls = std::to_string(l);
  const View f = view(feats[l]), t1 = view("t1_" + ls), t2 = view("t2_" + ls);
  conv(head_first_[l], f, t1, 1, ACT_SILU, nullptr, n, s);
  View rh;
  rh.p = rawhead_[l], rh.H = f.H, rh.W = f.W, rh.ct = RAW_CT, rh.co = 0, rh.C = RAW_CT;
  // box and coefficient branches: the 3x3 and the final 1x1 as one launch each (the class branch's 3 outputs are no column quad)
  conv_pair(head_box2_[l], t1.slice(0, 64), t2.slice(0, 64), 1, head_box3_[l], rh.slice(0, 64), ACT_NONE, n, s);
  conv(head_cls2_[l], t1.slice(64, 64), t2.slice(64, 64), 1, ACT_SILU, nullptr, n, s);
  conv_pair(head_coef2_[l], t1.slice(128, 32), t2.slice(128, 32), 1, head_coef3_[l], rh.slice(RAW_COEF, nm_), ACT_NONE, n, s);
  conv(head_cls3_[l], t2.slice(64, 64), rh.slice(RAW_CLS, cfg_.nc), 1, ACT_NONE, nullptr, n, s);
}

void Detector::raw(int n, float* pred, float* protos, hipStream_t s) {
  MTGV_CHECK(n > 0 && n <= last_n_, ERR_INVALID, "raw: n=%d but the last forward had %d frames", n, last_n_);
  if (pred) HIP_OK(hipMemcpyAsync(pred, pred_, (size_t)n * no() * na_ * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (protos) {
    const View pr = v_.at("protos");
    const long hw = (long)pr.H * pr.W, total = (long)n * nm_ * hw;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pr.p, protos, nm_, hw, total);
    HIP_OK(hipGetLastError());
  }
}

}  // namespace mtgv

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace mtgv;
struct mtgv_detector {
  Detector impl;
  explicit mtgv_detector(const mtgv_detector_cfg& c) : impl(c) {}
};

extern "C" {
MTGV_API int mtgv_detector_create(const mtgv_detector_cfg* cfg, mtgv_detector** out) {
  return guarded([&] {
    MTGV_CHECK(cfg != nullptr && out != nullptr, ERR_INVALID, "null argument");
    *out = new mtgv_detector(*cfg);
  });
}
MTGV_API void mtgv_detector_destroy(mtgv_detector* h) { delete h; }
MTGV_API int mtgv_detector_set_param(mtgv_detector* h, const char* key, const float* data_host, int64_t numel) {
  return guarded([&] {
    MTGV_CHECK(h && key && data_host, ERR_INVALID, "null argument");
    h->impl.set_param(key, data_host, numel);
  });
}
MTGV_API int mtgv_detector_missing_params(const mtgv_detector* h) { return h ? h->impl.missing() : -1; }
MTGV_API int mtgv_detector_finalize(mtgv_detector* h) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.finalize();
  });
}
MTGV_API int mtgv_detector_forward(mtgv_detector* h, const uint8_t* frames_dev, int32_t n, int32_t flip_rgb, int32_t* n_det_dev,
                                   float* boxes_dev, float* conf_dev, int32_t* cls_dev, int32_t* keep_idx_dev,
                                   float* mask_logits_dev, int32_t mask_rows, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.forward(frames_dev, n, flip_rgb, n_det_dev, boxes_dev, conf_dev, cls_dev, keep_idx_dev, mask_logits_dev, mask_rows,
                    (hipStream_t)stream);
  });
}
MTGV_API int mtgv_detector_raw(mtgv_detector* h, int32_t n, float* pred_dev, float* protos_dev, void* stream) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr, ERR_INVALID, "null handle");
    h->impl.raw(n, pred_dev, protos_dev, (hipStream_t)stream);
  });
}
MTGV_API int mtgv_mask_binarize(const float* logits_dev, int32_t n, int32_t mh, int32_t mw, int32_t scale, uint8_t* out_dev,
                                void* stream) {
  return guarded([&] {
    MTGV_CHECK(logits_dev && out_dev && n >= 0 && mh > 0 && mw > 0 && scale > 0, ERR_INVALID, "mask_binarize: bad argument");
    const long npx = (long)n * mh * scale * mw * scale;
    if (npx == 0) return;
    if ((mw * scale) % 16 == 0 && ((uintptr_t)out_dev % 16) == 0) {
      const long total = npx / 16;
      hipLaunchKernelGGL((mask_binarize_kernel<16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits_dev,
                         out_dev, mh, mw, scale, total);
    } else {
      hipLaunchKernelGGL((mask_binarize_kernel<1>), dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits_dev,
                         out_dev, mh, mw, scale, npx);
    }
    HIP_OK(hipGetLastError());
  });
}
MTGV_API int mtgv_detector_set_fork(mtgv_detector* h, int32_t mode) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && mode >= -1 && mode <= 1, ERR_INVALID, "mtgv_detector_set_fork: handle %p, mode %d (-1, 0, 1)", (void*)h, mode);
    h->impl.set_fork(mode);
  });
}
MTGV_API int mtgv_detector_flops(const mtgv_detector* h, double* flops_per_frame) {
  return guarded([&] {
    MTGV_CHECK(h != nullptr && flops_per_frame != nullptr, ERR_INVALID, "null argument");
    *flops_per_frame = h->impl.flops_per_frame();
  });
}
}
