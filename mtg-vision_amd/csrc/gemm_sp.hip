// LDS-DMA split-precision GEMM: instantiations, SP8 operand registry, tile selection, launch.
#include "gemm_sp.h"

#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>

#include "gemm_sp_kernel.h"

namespace mtgv {

// ---------------------------------------------------------------------------
// SP8 packing
// ---------------------------------------------------------------------------
// One wave per row: row maximum -> power-of-two scale (maximum lands in [2^13, 2^14)) -> split.  wscale = 2^-e.
__global__ __launch_bounds__(256) void sp8_pack_rows_kernel(const float* __restrict__ in, sp_h8* __restrict__ out,
                                                           float* __restrict__ wscale, long rows, int K) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = in + row * K;
  float mx = 0.f;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(x[k]));
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
  int e = 0;
  if (mx > 0.f && mx < INFINITY) {
    int ex;
    (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
    e = 14 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  const float sc = ldexpf(1.0f, e);
  if (lane == 0) wscale[row] = ldexpf(1.0f, -e);
  sp_h8* o = out + row * (K / 4);  // two 16-byte pieces per chunk of 8
  for (int c = lane; c < K / 8; c += 64) {
    const sp_f4 a = *reinterpret_cast<const sp_f4*>(x + c * 8) * sc, b = *reinterpret_cast<const sp_f4*>(x + c * 8 + 4) * sc;
    sp_h8 hi, lo;
    sp8_split8(a, b, hi, lo);
    o[2 * c] = hi;
    o[2 * c + 1] = lo;
  }
}

__global__ __launch_bounds__(256) void sp8_pack_plain_kernel(const float* __restrict__ in, sp_h8* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  sp_h8 hi, lo;
  sp8_split8(*reinterpret_cast<const sp_f4*>(in + i * 8), *reinterpret_cast<const sp_f4*>(in + i * 8 + 4), hi, lo);
  out[2 * i] = hi;
  out[2 * i + 1] = lo;
}

void sp8_pack_plain_launch(const float* in, void* out, long rows, int K, hipStream_t s) {
  MTGV_CHECK(K % 8 == 0, ERR_INVALID, "sp8: K=%d must be a multiple of 8", K);
  const long n8 = rows * (K / 8);
  if (n8 <= 0) return;
  hipLaunchKernelGGL(sp8_pack_plain_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, in, (sp_h8*)out, n8);
  HIP_OK(hipGetLastError());
}

// ---------------------------------------------------------------------------
// registry
// ---------------------------------------------------------------------------
namespace {
struct Sp8Entry {
  char* buf = nullptr;     // SP8 rows
  float* wscale = nullptr; // [rows]
  size_t n = 0;            // floats of the master
  int row_k = 0;
};
std::map<const float*, Sp8Entry> g_sp8;
std::mutex g_sp8_mu;
char* g_zero[MTGV_MAX_DEVICES] = {};  // one zero page per device: a handle may live on any GPU of the process (mtgv.h)
}  // namespace

const char* sp_zero_page() {
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(g_sp8_mu);
  if (g_zero[dev] == nullptr) {
    HIP_OK(hipMalloc((void**)&g_zero[dev], 256));
    HIP_OK(hipMemset(g_zero[dev], 0, 256));
  }
  return g_zero[dev];
}

void sp8_register(const float* W, size_t n_floats, int row_k) {
  if (W == nullptr || row_k <= 0 || row_k % 8 != 0 || n_floats == 0 || n_floats % (size_t)row_k != 0 || ((uintptr_t)W % 16) != 0) return;
  std::lock_guard<std::mutex> lk(g_sp8_mu);
  Sp8Entry& e = g_sp8[W];
  if (e.buf != nullptr && e.n == n_floats && e.row_k == row_k) return;
  if (e.buf != nullptr) (void)hipFree(e.buf);
  if (e.wscale != nullptr) (void)hipFree(e.wscale);
  e.buf = nullptr, e.wscale = nullptr;
  e.n = n_floats;
  e.row_k = row_k;
  // a failed allocation must not leave an entry that lookups would report as a valid copy
  if (hipMalloc((void**)&e.buf, n_floats * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&e.wscale, (n_floats / row_k) * sizeof(float)) != hipSuccess) {
    if (e.buf != nullptr) (void)hipFree(e.buf);
    g_sp8.erase(W);
    MTGV_CHECK(false, ERR_RUNTIME, "sp8_register: out of device memory for %zu floats", n_floats);
  }
}

void sp8_refresh(const float* W, size_t offset_floats, size_t n_floats, hipStream_t s) {
  Sp8Entry e;
  {
    std::lock_guard<std::mutex> lk(g_sp8_mu);
    auto it = g_sp8.find(W);
    if (it == g_sp8.end()) return;
    e = it->second;
  }
  if (n_floats == 0) return;
  MTGV_CHECK(offset_floats % e.row_k == 0 && n_floats % e.row_k == 0 && offset_floats + n_floats <= e.n, ERR_INVALID,
             "sp8 refresh must cover whole rows inside the registered buffer");
  const long row0 = (long)(offset_floats / e.row_k), rows = (long)(n_floats / e.row_k);
  hipLaunchKernelGGL(sp8_pack_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, W + offset_floats,
                     reinterpret_cast<sp_h8*>(e.buf + offset_floats * sizeof(float)), e.wscale + row0, rows, e.row_k);
  HIP_OK(hipGetLastError());
}

void sp8_unregister(const float* W) {
  std::lock_guard<std::mutex> lk(g_sp8_mu);
  auto it = g_sp8.find(W);
  if (it == g_sp8.end()) return;
  if (it->second.buf != nullptr) (void)hipFree(it->second.buf);
  if (it->second.wscale != nullptr) (void)hipFree(it->second.wscale);
  g_sp8.erase(it);
}

bool sp8_lookup(const float* W, int K, const char** sp8, const float** wscale) {
  std::lock_guard<std::mutex> lk(g_sp8_mu);
  if (g_sp8.empty()) return false;
  auto it = g_sp8.upper_bound(W);  // first base > W
  if (it == g_sp8.begin()) return false;
  --it;
  const Sp8Entry& e = it->second;
  const size_t off = (size_t)(W - it->first);
  if (e.buf == nullptr || off >= e.n || e.row_k != K || off % (size_t)K != 0) return false;
  if (sp8) *sp8 = e.buf + off * sizeof(float);
  if (wscale) *wscale = e.wscale + off / K;
  return true;
}

// ---------------------------------------------------------------------------
// tile configurations (one translation unit each: gemm_sp_c<id>.hip)
// ---------------------------------------------------------------------------
void gemm_sp_launch_cfg0(const SpDev& g, int amode, hipStream_t s);
void gemm_sp_launch_cfg1(const SpDev& g, int amode, hipStream_t s);
void gemm_sp_launch_cfg2(const SpDev& g, int amode, hipStream_t s);
void gemm_sp_launch_cfg3(const SpDev& g, int amode, hipStream_t s);
void gemm_sp_launch_cfg4(const SpDev& g, int amode, hipStream_t s);
void gemm_sp_launch_cfg5(const SpDev& g, int amode, hipStream_t s);
void gemm_sp_launch_cfg6(const SpDev& g, int amode, hipStream_t s);

namespace {
struct SpCfg {
  int wm, wn, tm, tn;
  double eff;  // relative efficiency of the tile's main loop (fitted to tools/gemm_sp_sweep.py)
  int ks = 2;  // k16 steps per stage
  int bm() const { return 32 * tm * wm; }
  int bn() const { return 32 * tn * wn; }
  int rb() const { return 64 * ks; }          // bytes per staged row
  int kps() const { return 16 * ks; }         // k per stage
};
const SpCfg kCfg[] = {
    {2, 2, 2, 2, 0.93},  // 128 x 128
    {2, 2, 2, 3, 1.00},  // 128 x 192
    {4, 1, 1, 3, 0.88},  // 128 x  96
    {4, 1, 1, 2, 0.80},  // 128 x  64
    {4, 1, 1, 1, 0.62},  // 128 x  32
    {4, 2, 1, 3, 0.00},  // 128 x 192 on eight waves (swapped in for configuration 1 below; not part of the search)
    {4, 1, 1, 1, 0.00, 1},  // 128 x 32 in 16-k stages: window convs with 16-channel slices only (chosen below, not searched)
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

bool sp_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MTGV_GEMM_SP");
    on = (e != nullptr && !strcmp(e, "0")) ? 0 : 1;
  }
  return on != 0;
}

bool is_conv(const GemmArgs& a) { return !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.stride_w <= 1 && a.pad == 0); }
}  // namespace

bool window_conv_on() {
  static const bool on = [] { const char* e = getenv("MTGV_SP_WINDOW"); return e == nullptr || atoi(e) != 0; }();
  return on;
}

// 3x3 / stride 1 / pad 1 convs whose channels come in slices of 32: stage the tile's input window once per slice
// instead of gathering every tap from L2 (1.65 - 2.2x fewer LDS fill bytes), while two blocks still fit a CU
static size_t window_bytes(const SpCfg& k, int Wd) {
  const int rpp = 1024 / k.rb();
  return (size_t)((k.bm() + 2 * Wd + 2 + rpp - 1) / rpp * rpp) * k.rb();
}

bool window_conv_fits(const GemmArgs& a, const SpPlan& pl) {
  const SpCfg& k = kCfg[pl.cfg];
  if (!(window_conv_on() && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.stride_w <= 0 && a.Cin % k.kps() == 0 &&
        a.OH == a.H && a.OW == a.Wd))
    return false;
  return window_bytes(k, a.Wd) + (size_t)2 * k.bn() * k.rb() <= 80 * 1024;
}

bool topk_sp_on() {
  static const bool on = [] { const char* e = getenv("MTGV_SP_TOPK"); return e == nullptr || atoi(e) != 0; }();
  return on;
}

bool gemm_sp_active() { return sp_enabled() && gemm_precision() == GEMM_PREC_F16X3; }

bool gemm_sp_takes_sp8(const float* W, int M, int N, int K, int lda, int c_off) {
  if (!gemm_sp_active()) return false;
  if (K % 8 != 0 || N % 4 != 0 || lda % 8 != 0 || c_off % 8 != 0 || M <= 0) return false;
  return sp8_lookup(W, K, nullptr, nullptr);
}

// chained 1x1: SP8 conv / dense input, SiLU between the layers, the whole output row in one tile of a one-wave-column
// configuration (N = 32 / 64 / 96 -> configurations 4 / 3 / 2), plain epilogue otherwise
static int chain_cfg(const GemmArgs& a) {
  static const bool on = [] { const char* e = getenv("MTGV_SP_CHAIN"); return e == nullptr || atoi(e) != 0; }();
  if (!on || a.W2 == nullptr || a.Out2 == nullptr) return -1;
  if (!(gemm_sp_active() && a.a_fmt == 1 && a.act == ACT_SILU && a.res == nullptr && a.grn_part == nullptr && a.topk == 0 && a.batch == 1 &&
        a.os == 1 && a.os_nq == 0 && a.OH2 == a.OH && a.OW2 == a.OW && a.a_scale == nullptr && a.ln_w == nullptr))
    return -1;
  if (!(a.N == 32 || a.N == 64 || a.N == 96) || a.N2 <= 0 || a.N2 % 32 != 0 || a.N2 > a.N) return -1;
  if (a.K % 8 != 0 || a.c_total % 8 != 0 || a.c_off % 8 != 0 || a.ldo2 % 4 != 0 || a.o_off2 % 4 != 0 || ((uintptr_t)a.Out2 & 15) != 0) return -1;
  if (a.out_fmt2 == 1 && (a.ldo2 % 8 != 0 || a.o_off2 % 8 != 0)) return -1;
  if (is_conv(a) && (a.stride_w > 0 || a.Cin % 8 != 0)) return -1;
  if (!sp8_lookup(a.W, a.K, nullptr, nullptr) || !sp8_lookup(a.W2, a.N, nullptr, nullptr)) return -1;
  return a.N == 32 ? 4 : (a.N == 64 ? 3 : 2);
}
bool gemm_sp_chain_ok(const GemmArgs& a) { return chain_cfg(a) >= 0; }

SpPlan gemm_sp_plan(const GemmArgs& a) {
  SpPlan pl;
  if (a.W2 != nullptr) {
    const int c = chain_cfg(a);
    MTGV_CHECK(c >= 0, ERR_INVALID, "gemm: this launch cannot chain its second layer (M=%d N=%d K=%d N2=%d): ask gemm_sp_chain_ok first", a.M,
               a.N, a.K, a.N2);
    const SpCfg& k = kCfg[c];
    pl.cfg = c, pl.bm = k.bm(), pl.bn = k.bn(), pl.unit_rows = 32 * k.tm;
    pl.tiles_m = ceil_div(a.M, k.bm()), pl.tiles_n = 1;
    return pl;
  }
  const bool sp8_in = a.a_fmt == 1;
  auto none = [&]() -> SpPlan {
    MTGV_CHECK(!sp8_in && a.out_fmt == 0 && a.res_fmt == 0, ERR_INVALID,
               "gemm: SP8 tensors handed to a launch the SP kernel cannot run (M=%d N=%d K=%d)", a.M, a.N, a.K);
    return pl;
  };
  if (!sp_enabled() || gemm_precision() != GEMM_PREC_F16X3) return none();
  const bool conv = is_conv(a);
  const bool remap = !(a.os == 1 && a.oy == 0 && a.ox == 0 && a.OH2 == a.OH && a.OW2 == a.OW);
  if (a.batch != 1 || a.crop_boxes != nullptr || a.m_count != nullptr || a.ln_w != nullptr) return none();
  if (a.topk > 0) {  // match path: 128 x 192 tiles, f32 queries by DMA (A mode 4), fused top-k (gemm_sp_kernel.h, EPI 16)
    if (sp8_in || conv || a.K % 8 != 0 || a.c_total % 8 != 0 || a.c_off % 8 != 0 || ((uintptr_t)a.A & 15) != 0 || a.a_scale != nullptr ||
        a.a_mul != 1.0f || a.res != nullptr || a.act != ACT_NONE || a.M < 128 || a.N < 192 || !sp8_lookup(a.W, a.K, nullptr, nullptr) ||
        !topk_sp_on())
      return none();
    const SpCfg& k1 = kCfg[1];
    pl.cfg = 1;
    pl.bm = k1.bm(), pl.bn = k1.bn();
    pl.unit_rows = 32 * k1.tm;
    pl.tiles_m = ceil_div(a.M, k1.bm());
    pl.tiles_n = ceil_div(a.N, k1.bn());
    return pl;
  }
  if (conv && (!sp8_in || a.stride_w > 0 || a.Cin % 8 != 0)) return none();  // the gather is a DMA-path feature
  if (remap && !sp8_in) return none();
  if (a.K % 8 != 0 || a.c_total % 8 != 0 || a.c_off % 8 != 0 || a.ldo % 4 != 0 || a.o_off % 4 != 0 ||
      (a.res != nullptr && a.ldr % 4 != 0))
    return none();
  // a ragged last column quad is stored element by element: plain f32 outputs only
  if (a.N % 4 != 0 && (a.out_fmt != 0 || a.res != nullptr || a.grn_part != nullptr || remap)) return none();
  if (a.out_fmt == 1 && (a.N % 8 != 0 || a.ldo % 8 != 0 || a.o_off % 8 != 0 || a.grn_part != nullptr)) return none();
  if (a.res != nullptr && a.res_fmt == 1 && (a.ldr % 8 != 0 || a.N % 8 != 0)) return none();
  if (((uintptr_t)a.Out % 16) != 0 || (a.res != nullptr && ((uintptr_t)a.res % 16) != 0)) return none();
  if (sp8_in && a.a_scale != nullptr) return none();
  if (!sp8_lookup(a.W, a.K, nullptr, nullptr)) return none();
  if (!sp8_in && (a.N < 64 || a.M < 128)) return none();  // tiny problems: the convert-on-load kernel's narrow tiles fit better

  int best = -1;
  double best_cost = 0;
  if (const char* e = getenv("MTGV_SP_CFG")) {
    const int c = atoi(e);
    if (c >= 0 && c < kNumCfg && kCfg[c].ks == 2) best = c;
  }
  if (best < 0) {
    for (int c = 0; c < kNumCfg; ++c) {
      const SpCfg& k = kCfg[c];
      if (k.eff <= 0.0) continue;  // not part of the search
      // the 1 KB-per-stage multiplier image of the f32-by-DMA A path does not fit beside the 128 x 192 ring twice per CU
      if (c == 1 && !sp8_in && a.a_scale != nullptr) continue;
      const long tiles = (long)ceil_div(a.M, k.bm()) * ceil_div(a.N, k.bn());
      // two blocks per CU: a "round" is up to 512 tiles, each CU working on two at half speed
      const double rounds = (double)((tiles + 511) / 512);
      const double per_cu = rounds * 2.0 * k.bm() * k.bn();
      const double cost = per_cu / k.eff;
      if (best < 0 || cost < best_cost) best = c, best_cost = cost;
    }
  }
  // the whole-ConvTranspose launch (os_nq column groups of 64): one group per 128 x 64 tile measured 9 % faster than 128 x 128
  if (a.os_nq == 64 && getenv("MTGV_SP_CFG") == nullptr) best = 3;
  {  // 3x3 / stride-1 convs with 16-channel slices (Cin % 32 != 0): the window conv in 16-k stages instead of nine tap gathers
    static const bool on6 = [] { const char* e = getenv("MTGV_SP_WIN16"); return e == nullptr || atoi(e) != 0; }();
    SpPlan p6;
    p6.cfg = 6;
    if (on6 && getenv("MTGV_SP_CFG") == nullptr && conv && sp8_in && a.Cin % 32 != 0 && a.N <= 32 && window_conv_fits(a, p6)) best = 6;
  }
  {  // eight-wave twin of the 128 x 192 tile (four waves per SIMD) for pwconv1-shaped launches: SP8 rows in, activation
     // + GRN sums out; measured -3..-4 % on the stage 2-3 layers, nothing on the others (MTGV_SP_CFG8=0: off)
    static const bool on8 = [] { const char* e = getenv("MTGV_SP_CFG8"); return e == nullptr || atoi(e) != 0; }();
    // (not below 12288 rows: 6144 x 3072 x 768, the stage-3 pwconv1, is 8 % faster on the four-wave tile -
    // tools/sp_cfg_sweep.py, profiles/r04_sp_cfg_sweep.txt)
    if (on8 && best == 1 && !conv && sp8_in && a.grn_part != nullptr && a.topk == 0 && a.K >= 256 && a.M >= 12288) best = 5;
  }
  const SpCfg& k = kCfg[best];
  pl.cfg = best;
  pl.bm = k.bm(), pl.bn = k.bn();
  pl.unit_rows = 32 * k.tm;
  pl.tiles_m = ceil_div(a.M, k.bm());
  pl.tiles_n = ceil_div(a.N, k.bn());
  return pl;
}

// Tuning aid: with MTGV_SP_STAMPS=1 every launch made while the launch profiler is on leaves per-tile clock stamps
// (gemm_sp_kernel.h); gemm_sp_stamps_dump writes them next to the per-launch table (tools/sp_stamps.py reads them).
namespace {
struct StampRec { int M, N, K, cfg, amode, act, tiles; long* buf; };
std::vector<StampRec> g_stamps;
bool stamps_on() {
  static const bool on = [] { const char* e = getenv("MTGV_SP_STAMPS"); return e != nullptr && atoi(e) != 0; }();
  return on;
}
}  // namespace

// top-k candidate layout of a launch the SP kernel would take: groups of `cols` columns, `slots` groups per row
bool gemm_sp_topk_layout(const GemmArgs& a, int* slots, int* cols) {
  const SpPlan pl = gemm_sp_plan(a);
  if (pl.cfg < 0) return false;
  const SpCfg& k = kCfg[pl.cfg];
  *slots = pl.tiles_n * k.wn;
  *cols = 32 * k.tn;
  return true;
}

double gemm_sp_fill_bytes(const GemmArgs& a, const SpPlan& pl) {
  const SpCfg& k = kCfg[pl.cfg];
  const double tiles = (double)pl.tiles_m * pl.tiles_n;
  const double b_tile = (double)k.bn() * a.K * 4.0;
  double a_tile = (double)k.bm() * a.K * 4.0;  // dense rows, or one gather per tap
  if (a.a_fmt == 1 && is_conv(a) && window_conv_fits(a, pl)) a_tile = (double)window_bytes(k, a.Wd) * (a.Cin / k.kps());
  if (a.a_scale != nullptr) a_tile += (double)(a.K / 32) * 1024.0;
  return tiles * (a_tile + b_tile);
}

void gemm_sp_stamps_dump(const char* path) {
  if (g_stamps.empty()) return;
  HIP_OK(hipDeviceSynchronize());
  FILE* f = fopen(path, "wb");
  MTGV_CHECK(f != nullptr, ERR_RUNTIME, "cannot open %s", path);
  std::vector<long> host;
  for (const StampRec& r : g_stamps) {
    const int hdr[8] = {r.M, r.N, r.K, r.cfg, r.amode, r.act, r.tiles, 0};
    fwrite(hdr, sizeof(int), 8, f);
    host.resize((size_t)r.tiles * 8);
    HIP_OK(hipMemcpy(host.data(), r.buf, host.size() * sizeof(long), hipMemcpyDeviceToHost));
    fwrite(host.data(), sizeof(long), host.size(), f);
    HIP_OK(hipFree(r.buf));
  }
  fclose(f);
  g_stamps.clear();
}

int gemm_sp_topk_hi16_range_cols() { return 32 * kCfg[1].tn; }
int gemm_sp_topk_hi16_slots(int N) { return ceil_div(N, kCfg[1].bn()) * kCfg[1].wn; }

void gemm_sp_topk_hi16_launch(const void* q_hi, const void* bank_hi, const float* wscale, int b, int N, int K, int kp, float* cand_s,
                              int* cand_i, int* slots, hipStream_t s) {
  const SpCfg& k1 = kCfg[1];
  MTGV_CHECK(b >= 128 && N >= k1.bn() && K % 64 == 0 && kp >= 1 && kp <= 32 * k1.tn, ERR_INVALID, "topk_hi16: b=%d N=%d K=%d kp=%d", b, N, K, kp);
  MTGV_CHECK(((uintptr_t)q_hi & 15) == 0 && ((uintptr_t)bank_hi & 15) == 0, ERR_INVALID, "topk_hi16: operands must be 16-byte aligned");
  SpDev g;
  g.A = reinterpret_cast<const char*>(q_hi);
  g.a_rowb = (long)K * 2;
  g.W = reinterpret_cast<const char*>(bank_hi);
  g.wscale = wscale;
  g.M = b, g.N = N, g.K = K;
  g.zero = sp_zero_page();
  g.off32 = ((double)N * K * 2.0 < 4294967296.0 - 65536.0) ? 1 : 0;
  g.tiles_m = ceil_div(b, k1.bm()), g.tiles_n = ceil_div(N, k1.bn());
  g.cand_s = cand_s, g.cand_i = cand_i, g.topk = kp;
  g.d_hw = make_fastdiv(1), g.d_ohw = make_fastdiv(1), g.d_ow = make_fastdiv(1), g.d_cin = make_fastdiv(1), g.d_kw = make_fastdiv(1);
  if (slots) *slots = g.tiles_n * k1.wn;
  gemm_sp_launch_cfg1(g, 6, s);
  HIP_OK(hipGetLastError());
}

void gemm_sp_launch(const GemmArgs& a, const SpPlan& pl, hipStream_t s) {
  MTGV_CHECK(pl.cfg >= 0 && pl.cfg < kNumCfg, ERR_INVALID, "gemm_sp: no plan");
  SpDev g;
  g.A = reinterpret_cast<const char*>(a.A);
  g.a_rowb = (long)a.c_total * 4;
  g.a_offb = (long)a.c_off * 4;
  const char* w8 = nullptr;
  const float* wsc = nullptr;
  MTGV_CHECK(sp8_lookup(a.W, a.K, &w8, &wsc), ERR_RUNTIME, "gemm_sp: weights lost their SP8 copy");
  g.W = w8;
  g.wscale = wsc;
  g.bias = a.bias;
  g.res = a.res;
  g.ldr = a.ldr;
  g.res_fmt = a.res_fmt;
  g.Out = a.Out;
  g.ldo = a.ldo;
  g.o_off = a.o_off;
  g.out_fmt = a.out_fmt;
  g.M = a.M, g.N = a.N, g.K = a.K;
  g.grn_part = a.grn_part;
  g.hw = a.hw > 0 ? a.hw : 1;
  g.segmax = a.segmax;
  g.d_hw = make_fastdiv((uint32_t)g.hw);
  g.a_scale = a.a_scale;
  g.a_mul = a.a_fmt == 1 ? 1.0f : a.a_mul;
  g.a_unmul = a.a_fmt == 1 ? 1.0f : a.a_unmul;
  g.zero = sp_zero_page();
  {  // byte extents of the operands as the DMA addresses them
    const double lim = 4294967296.0 - 65536.0;
    g.off32 = ((double)a.M * g.a_rowb + (double)g.a_offb < lim && (double)a.N * a.K * 4.0 < lim) ? 1 : 0;
  }
  g.tiles_m = pl.tiles_m, g.tiles_n = pl.tiles_n;
  g.cand_s = a.cand_s, g.cand_i = a.cand_i, g.topk = a.topk;
  g.act = a.act;
  g.H = a.H, g.Wd = a.Wd, g.Cin = a.Cin, g.KW = a.KW, g.stride = a.stride, g.pad = a.pad, g.OH = a.OH, g.OW = a.OW;
  g.d_ohw = make_fastdiv((uint32_t)(a.OH * a.OW));
  g.d_ow = make_fastdiv((uint32_t)a.OW);
  g.d_cin = make_fastdiv((uint32_t)(a.Cin > 0 ? a.Cin : 1));
  g.d_kw = make_fastdiv((uint32_t)a.KW);
  g.remap = !(a.os == 1 && a.oy == 0 && a.ox == 0 && a.OH2 == a.OH && a.OW2 == a.OW);
  g.os = a.os, g.oy = a.oy, g.ox = a.ox, g.OH2 = a.OH2, g.OW2 = a.OW2;
  if (a.W2 != nullptr) {
    const char* w28 = nullptr;
    const float* w2s = nullptr;
    MTGV_CHECK(sp8_lookup(a.W2, a.N, &w28, &w2s), ERR_RUNTIME, "gemm_sp: second-layer weights lost their SP8 copy");
    g.W2 = w28, g.wscale2 = w2s, g.bias2 = a.bias2, g.Out2 = a.Out2, g.ldo2 = a.ldo2, g.o_off2 = a.o_off2, g.out_fmt2 = a.out_fmt2;
    g.act2 = a.act2, g.N2 = a.N2;
  }
  g.nq = a.os_nq;
  g.d_nq = make_fastdiv((uint32_t)(a.os_nq > 0 ? a.os_nq : 1)), g.d_os = make_fastdiv((uint32_t)(a.os > 0 ? a.os : 1));
  if (a.os_nq > 0)
    MTGV_CHECK(g.remap && a.os_nq % 8 == 0 && a.N == a.os * a.os * a.os_nq && a.oy == 0 && a.ox == 0 && a.res == nullptr &&
                   a.grn_part == nullptr,
               ERR_INVALID, "gemm_sp: os_nq=%d does not describe a %dx%d scatter of N=%d columns", a.os_nq, a.os, a.os, a.N);
  if (a.grn_part != nullptr) {
    MTGV_CHECK(a.segmax >= (pl.unit_rows - 1) / g.hw + 2, ERR_INVALID, "gemm_sp: segmax %d too small", a.segmax);
    // the caller sized and will reduce the partial sums for the unit it planned with (gemm_grn_layout)
    MTGV_CHECK(a.grn_unit_rows == 0 || a.grn_unit_rows == pl.unit_rows, ERR_RUNTIME,
               "gemm_sp: GRN partials planned for %d-row units, this launch writes %d-row units", a.grn_unit_rows, pl.unit_rows);
  }
  // f32 A: by DMA and split at the fragment read when the rows are 16-byte aligned, need no range multiplier and a
  // tile's rows span at most 8 images of the per-image multipliers; through registers otherwise
  int amode = a.a_fmt == 1 ? (is_conv(a) ? 2 : 0) : 1;
  if (amode == 2 && window_conv_fits(a, pl)) amode = 5;
  if (amode == 1 && g.a_mul == 1.0f && ((uintptr_t)g.A & 15) == 0) {
    if (g.a_scale == nullptr) amode = 4;
    else if (((uintptr_t)g.a_scale & 15) == 0 && (kCfg[pl.cfg].bm() - 1) / g.hw + 2 <= 8) amode = 3;
  }
  if (stamps_on() && gemm_profile_enabled()) {
    const int tiles = pl.tiles_m * pl.tiles_n;
    long* buf = nullptr;
    HIP_OK(hipMalloc(&buf, (size_t)tiles * 8 * sizeof(long)));
    HIP_OK(hipMemsetAsync(buf, 0, (size_t)tiles * 8 * sizeof(long), s));
    g.stamps = buf;
    g_stamps.push_back({a.M, a.N, a.K, pl.cfg, amode, a.act, tiles, buf});
  }
  switch (pl.cfg) {
    case 0: gemm_sp_launch_cfg0(g, amode, s); break;
    case 1: gemm_sp_launch_cfg1(g, amode, s); break;
    case 2: gemm_sp_launch_cfg2(g, amode, s); break;
    case 3: gemm_sp_launch_cfg3(g, amode, s); break;
    case 4: gemm_sp_launch_cfg4(g, amode, s); break;
    case 5: gemm_sp_launch_cfg5(g, amode, s); break;
    case 6: gemm_sp_launch_cfg6(g, amode, s); break;
    default: MTGV_CHECK(false, ERR_INVALID, "gemm_sp: bad cfg %d", pl.cfg);
  }
  HIP_OK(hipGetLastError());
}

}  // namespace mtgv
