// Host side of the implicit-GEMM conv / linear kernel (gemm_kernel.h) and its f32-operand instantiations.
//
// Why f32-grade products: the path's contract is fp32 embeddings within 1e-4 of the
// PyTorch CPU reference and identical top-1 ids; plain fp16 operands miss that
// by 25x.  gfx950 has no TF32.  Two operand precisions meet the contract:
//   f32    exact f32-in/f32-acc matrix instruction v_mfma_f32_32x32x2_f32 (157 TFLOP/s dense ceiling)
//   f16x3  each f32 operand split into fp16 hi + lo, three fp16 MFMAs per product (gemm_f16x3.hip);
//          measured error vs fp64 is at the f32 level (tools/micro/split_gemm.hip, tests/test_gpu_precision.py)
#include "gemm_kernel.h"
#include "gemm_sp.h"

#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace mtgv {

bool gemm_dispatch_f32(const GemmDev& g, const GemmPlan& pl, bool conv, bool apro, int grid, hipStream_t s) {
  return gemm_dispatch<0>(g, pl, conv, apro, grid, s);
}
bool gemm_dispatch_f16x3(const GemmDev& g, const GemmPlan& pl, bool conv, bool apro, int grid, hipStream_t s);  // gemm_f16x3.hip

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Optional launch profiler (bench.py roofline leg): HIP events bracket every GEMM launch on the
// stream it is launched on; algorithmic FLOPs = 2*M*N*K of the real (unpadded) problem.
namespace {
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  double flops = 0;
  double bytes = 0;  // compulsory HBM bytes: every operand element read once, every output written once
  long launches = 0;
  struct Rec { int M, N, K, KH, stride, batch, act, apro, grn, topk, sp; double bytes, fill, xflops; };
  std::vector<Rec> recs;
} g_prof;
}  // namespace

void gemm_profile_enable(bool on) {
  g_prof.on = on;
  g_prof.used = 0;
  g_prof.flops = 0;
  g_prof.bytes = 0;
  g_prof.launches = 0;
  g_prof.recs.clear();
}

// per-launch table (shape, ms, TFLOP/s) of everything recorded since enable; tuning aid
void gemm_profile_dump(const char* path) {
  FILE* f = fopen(path, "w");
  MTGV_CHECK(f != nullptr, ERR_RUNTIME, "cannot open %s", path);
  fprintf(f, "idx,M,N,K,KH,stride,batch,act,apro,grn,topk,tm,tn,bk,ms,tflops,bytes,sp,fill,xflops\n");
  for (size_t i = 0; i + 1 < g_prof.used && i / 2 < g_prof.recs.size(); i += 2) {
    HIP_OK(hipEventSynchronize(g_prof.ev[i + 1]));
    float t = 0.f;
    HIP_OK(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
    const auto& r = g_prof.recs[i / 2];
    const GemmPlan pl = r.topk ? GemmPlan{1, 2, 16, 0, 0} : gemm_plan(r.M, r.N, r.K, r.act != 0, r.apro != 0);
    const double fl = 2.0 * r.M * r.N * r.K * r.batch + r.xflops;  // xflops: a second layer chained into the launch
    fprintf(f, "%zu,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.2f,%.0f,%d,%.0f,%.0f\n", i / 2, r.M, r.N, r.K, r.KH, r.stride, r.batch, r.act,
            r.apro, r.grn, r.topk, pl.tm, pl.tn, pl.bk, t, fl / (t * 1e-3) / 1e12, r.bytes, r.sp, r.fill, r.xflops);
  }
  fclose(f);
  gemm_sp_stamps_dump((std::string(path) + ".stamps").c_str());
}

bool gemm_profile_enabled() { return g_prof.on; }
double gemm_profile_bytes() { return g_prof.bytes; }

void gemm_profile_read(double* ms, double* flops, long* launches) {
  double total = 0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    HIP_OK(hipEventSynchronize(g_prof.ev[i + 1]));
    float t = 0.f;
    HIP_OK(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
    total += t;
  }
  if (ms) *ms = total;
  if (flops) *flops = g_prof.flops;
  if (launches) *launches = g_prof.launches;
}

// fill: bytes the launch's tiles pull into LDS (every tile its A and B panels; 4 bytes per element in either format)
static void prof_begin(const GemmArgs& a, hipStream_t s, int sp, double fill, double bytes_override = -1.0) {
  if (!g_prof.on) return;
  while (g_prof.ev.size() < g_prof.used + 2) {
    hipEvent_t e;
    HIP_OK(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  HIP_OK(hipEventRecord(g_prof.ev[g_prof.used], s));
  const double xfl = a.W2 != nullptr ? 2.0 * (double)a.M * a.N2 * a.N : 0.0;
  g_prof.flops += 2.0 * (double)a.M * a.N * a.K * a.batch + xfl;
  {
    const bool conv = !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.stride_w <= 1 && a.pad == 0);
    const double a_el = conv ? (double)(a.M / (a.OH * a.OW)) * a.H * a.Wd * a.Cin : (double)a.M * a.K;
    const double w_el = (double)a.N * a.K;
    // (a chained launch stores only its second layer's output and reads the second weight matrix besides)
    const double o_el = a.topk > 0 ? (double)a.M * ceil_div(a.N, 64) * a.topk * 2
                                   : (a.W2 != nullptr ? (double)a.M * a.N2 + (double)a.N2 * a.N : (double)a.M * a.N);
    const double r_el = a.res != nullptr ? (double)a.M * a.N : 0.0;
    double by = 4.0 * ((a.strideA != 0 || a.batch == 1 ? a.batch : 1) * a_el + (a.strideW != 0 || a.batch == 1 ? a.batch : 1) * w_el +
                       a.batch * (o_el + r_el));
    if (bytes_override >= 0.0) by = bytes_override;
    g_prof.bytes += by;
    g_prof.recs.push_back({a.M, a.N, a.K, a.KH, a.stride, a.batch, a.act, a.a_scale != nullptr, a.grn_part != nullptr, a.topk, sp, by, fill, xfl});
  }
  g_prof.launches += 1;
}
static void prof_end(hipStream_t s) {
  if (!g_prof.on) return;
  HIP_OK(hipEventRecord(g_prof.ev[g_prof.used + 1], s));
  g_prof.used += 2;
}
void gemm_profile_begin(const GemmArgs& a, hipStream_t s, int sp, double fill, double bytes) { prof_begin(a, s, sp, fill, bytes); }
void gemm_profile_end(hipStream_t s) { prof_end(s); }

// ---------------------------------------------------------------------------
// pre-split weight registry (f16x3)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ in, float* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f16x4 hi, lo;
  split_f16(reinterpret_cast<const f32x4*>(in)[i], hi, lo);
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = hi[j], o[4 + j] = lo[j];
  reinterpret_cast<f16x8*>(out)[i] = o;
}

// rows of row_k floats, stored scaled by 1 / wscale[row] (the power of two the SP8 copy of the same buffer uses)
__global__ __launch_bounds__(256) void split_pack_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ wscale, long n4, int rk4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float sc = 1.0f / wscale[i / rk4];  // exact: wscale is a power of two
  f16x4 hi, lo;
  split_f16(reinterpret_cast<const f32x4*>(in)[i] * sc, hi, lo);
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = hi[j], o[4 + j] = lo[j];
  reinterpret_cast<f16x8*>(out)[i] = o;
}

namespace {
struct SplitEntry {
  float* buf = nullptr;
  size_t n = 0;
  int row_k = 0;  // > 0: rows of row_k floats stored with the per-row scale of the SP8 registry
};
std::map<const float*, SplitEntry> g_split;
std::mutex g_split_mu;
}  // namespace

void gemm_split_register(const float* W, size_t n_floats, int row_k) {
  if (W == nullptr || n_floats < 64 || n_floats % 4 != 0 || ((uintptr_t)W % 16) != 0) return;
  if (row_k > 0) sp8_register(W, n_floats, row_k);
  std::lock_guard<std::mutex> lk(g_split_mu);
  SplitEntry& e = g_split[W];
  const int rk = (row_k > 0 && row_k % 8 == 0 && n_floats % (size_t)row_k == 0) ? row_k : 0;  // what sp8_register accepts
  if (e.buf != nullptr && e.n == n_floats) {
    e.row_k = rk;
    return;
  }
  if (e.buf != nullptr) (void)hipFree(e.buf);
  e.n = n_floats;
  e.row_k = rk;
  HIP_OK(hipMalloc((void**)&e.buf, n_floats * sizeof(float)));
}

void gemm_split_refresh(const float* W, size_t offset_floats, size_t n_floats, hipStream_t s) {
  sp8_refresh(W, offset_floats, n_floats, s);  // also (re)computes the rows' scales
  float* out = nullptr;
  int row_k = 0;
  {
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split.find(W);
    if (it == g_split.end()) return;
    MTGV_CHECK(offset_floats % 4 == 0 && n_floats % 4 == 0 && offset_floats + n_floats <= it->second.n, ERR_INVALID,
               "split refresh outside the registered buffer");
    out = it->second.buf;
    row_k = it->second.row_k;
  }
  if (n_floats == 0) return;
  const long n4 = (long)(n_floats / 4);
  const float* wsc = nullptr;
  if (row_k > 0) {
    MTGV_CHECK(offset_floats % row_k == 0 && n_floats % row_k == 0, ERR_INVALID, "split refresh must cover whole rows");
    MTGV_CHECK(sp8_lookup(W + offset_floats, row_k, nullptr, &wsc), ERR_RUNTIME, "split refresh: row scales missing");
    hipLaunchKernelGGL(split_pack_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, W + offset_floats,
                       out + offset_floats, wsc, n4, row_k / 4);
  } else {
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, W + offset_floats, out + offset_floats, n4);
  }
  HIP_OK(hipGetLastError());
}

void gemm_split_unregister(const float* W) {
  sp8_unregister(W);
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split.find(W);
  if (it == g_split.end()) return;
  if (it->second.buf != nullptr) (void)hipFree(it->second.buf);
  g_split.erase(it);
}

const float* gemm_split_lookup(const float* W) {
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split.find(W);
  return it == g_split.end() ? nullptr : it->second.buf;
}

// pre-split copy usable by a launch with rows of K floats: unscaled copies always, scaled ones only with their scales
static const float* split_for_launch(const float* W, int K, const float** wscale) {
  *wscale = nullptr;
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split.find(W);
  if (it == g_split.end()) return nullptr;
  if (it->second.row_k == 0) return it->second.buf;
  if (it->second.row_k != K || !sp8_lookup(W, K, nullptr, wscale)) return nullptr;
  return it->second.buf;
}

static int g_prec = -1;  // -1: not read from the environment yet

int gemm_precision() {
  if (g_prec < 0) {
    const char* e = getenv("MTGV_GEMM_PREC");
    if (e == nullptr || !*e || !strcmp(e, "f16x3")) g_prec = GEMM_PREC_F16X3;  // default: both meet the contract, f16x3 is faster
    else if (!strcmp(e, "f32")) g_prec = GEMM_PREC_F32;
    else MTGV_CHECK(false, ERR_INVALID, "MTGV_GEMM_PREC=%s: expected f32 or f16x3", e);
  }
  return g_prec;
}

void gemm_set_precision(int prec) {
  MTGV_CHECK(prec == GEMM_PREC_F32 || prec == GEMM_PREC_F16X3, ERR_INVALID, "gemm precision %d: 0 (f32) or 1 (f16x3)", prec);
  g_prec = prec;
}

GemmPlan gemm_plan(int M, int N, int K, bool heavy_epilogue, bool scaled_a) {
  GemmPlan pl;
  if (const char* e = getenv("MTGV_GEMM_TILE")) {
    int tm = 0, tn = 0, bk = 0;
    if (sscanf(e, "%d,%d,%d", &tm, &tn, &bk) == 3 && (tm == 1 || tm == 2) && tn >= 1 && tn <= 5 &&
        (bk == 16 || bk == 32)) {
      pl.tm = tm, pl.tn = tn, pl.bk = bk;
      pl.tiles_m = ceil_div(M, pl.bm());
      pl.tiles_n = ceil_div(N, pl.bn());
      return pl;
    }
  }
  // Cost model fitted to tile sweeps on MI355X (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt):
  // narrow tiles with BK = 16 win - more resident blocks per CU overlap one block's tile load and
  // epilogue with another's MFMAs - and whole rounds over the 256 CUs matter more than tile width.
  //   cost = rounds(tiles / 256 CUs) * BM * BN * (K + per-tile overhead in K-equivalents) / efficiency(tn)
  // f16x3 re-splits the A panel once per column tile and reads four fragments per accumulator column, so its
  // narrow tiles fall off faster (profiles/r01_gemm_sweep_f16x3.txt).
  const bool split = gemm_precision() == GEMM_PREC_F16X3;
  const double ov = 36.0 + (heavy_epilogue ? 48.0 : 0.0);
  static const double eff_tab[2][3][6] = {
      {{0, 0.88, 0.93, 1.00, 0.80, 0.62},    // f32: plain
       {0, 0.85, 0.93, 0.98, 1.00, 0.85},    //      activation epilogue
       {0, 0.80, 0.90, 1.00, 0.85, 0.80}},   //      GRN multiplier on A (per-fragment multiply amortises over TN)
      {{0, 0.65, 0.88, 1.00, 0.90, 0.72},    // f16x3: plain
       {0, 0.62, 0.90, 0.97, 1.00, 0.90},    //      activation epilogue
       {0, 0.65, 0.88, 1.00, 0.88, 0.70}}};  //      GRN multiplier on A
  const double* eff = eff_tab[split ? 1 : 0][scaled_a ? 2 : heavy_epilogue ? 1 : 0];
  int best_tn = 1;
  double best = -1;
  const long tiles_m = ceil_div(M, 128);
  for (int tn = 1; tn <= 5; ++tn) {
    const long tiles = tiles_m * ceil_div(N, 32 * tn);
    const double rounds = (double)((tiles + 255) / 256);
    const double cost = rounds * 128.0 * 32.0 * tn * ((double)K + ov) / eff[tn];
    if (best < 0 || cost < best) best = cost, best_tn = tn;
  }
  pl.tm = 1;
  pl.tn = best_tn;
  pl.bk = (best_tn == 1 && K >= 256) ? 32 : 16;  // sweep: BK 32 pays only for the narrowest tile on long K
  pl.tiles_m = ceil_div(M, pl.bm());
  pl.tiles_n = ceil_div(N, pl.bn());
  return pl;
}

int gemm_grn_segmax(const GemmPlan& p, int hw) { return (p.bm() - 1) / hw + 2; }

size_t gemm_grn_part_floats(const GemmPlan& p, int N, int hw) {
  return (size_t)p.tiles_m * gemm_grn_segmax(p, hw) * N;
}

GrnLayout gemm_grn_layout(const GemmArgs& a, const GemmPlan& p) {
  GrnLayout l;
  const SpPlan sp = gemm_sp_plan(a);
  const int hw = a.hw > 0 ? a.hw : 1;
  if (sp.cfg >= 0) {
    l.unit_rows = sp.unit_rows;
    l.segmax = (sp.unit_rows - 1) / hw + 2;
  } else {
    l.unit_rows = p.bm();
    l.segmax = gemm_grn_segmax(p, hw);
  }
  l.floats = (size_t)ceil_div(a.M, l.unit_rows) * l.segmax * a.N;
  return l;
}

size_t gemm_grn_part_floats_max(int M, int N, int hw) {
  size_t mx = 0;
  for (int unit : {32, 64, 128, 256}) {
    const size_t f = (size_t)ceil_div(M, unit) * ((unit - 1) / hw + 2) * N;
    mx = f > mx ? f : mx;
  }
  return mx;
}

bool gemm_ln_fusable(const GemmArgs& a, const GemmPlan& pl) {
  static const bool on = [] { const char* e = getenv("MTGV_GEMM_LN"); return e == nullptr || atoi(e) != 0; }();
  if (!on) return false;
  const bool conv = !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.stride_w <= 1 && a.pad == 0);
  const bool remap = !(a.os == 1 && a.oy == 0 && a.ox == 0 && a.OH2 == a.OH && a.OW2 == a.OW);
  // the instance that carries the epilogue: conv gather, 128 x 96 x 16 tile; one tile per row, whole tiles, plain f32 rows out
  if (!(pl.tm == 1 && pl.tn == 3 && pl.bk == 16 && conv && a.N == pl.bn() && a.M % pl.bm() == 0)) return false;
  if (a.batch != 1 || a.act != ACT_NONE || a.res != nullptr || a.a_scale != nullptr || a.grn_part != nullptr || a.topk > 0 ||
      a.crop_boxes != nullptr || a.m_count != nullptr || remap || a.out_fmt != 0 || a.a_fmt != 0)
    return false;
  GemmArgs t = a;
  t.ln_w = nullptr;
  return gemm_sp_plan(t).cfg < 0;  // (the LDS-DMA kernel has no such epilogue)
}

void gemm_launch(const GemmArgs& a, const GemmPlan& pl, hipStream_t s) {
  if (a.ln_w != nullptr)
    MTGV_CHECK(a.ln_b != nullptr && gemm_ln_fusable(a, pl), ERR_INVALID, "gemm: this launch cannot normalise its rows in the epilogue");
  MTGV_CHECK(a.batch >= 1 && a.batch <= 65535, ERR_INVALID, "gemm: batch=%d", a.batch);
  MTGV_CHECK(a.M > 0 && a.N > 0 && a.K > 0, ERR_INVALID, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  MTGV_CHECK(a.K % 4 == 0 && a.Cin % 4 == 0 && a.c_total % 4 == 0 && a.c_off % 4 == 0, ERR_INVALID,
             "gemm: K=%d Cin=%d c_total=%d c_off=%d must be multiples of 4", a.K, a.Cin, a.c_total, a.c_off);
  MTGV_CHECK(a.K == a.KH * a.KW * a.Cin, ERR_INVALID, "gemm: K=%d != %d*%d*%d", a.K, a.KH, a.KW, a.Cin);
  MTGV_CHECK(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0, ERR_INVALID, "gemm: operands must be 16-byte aligned");
  MTGV_CHECK((long)a.M * (long)(a.OH * a.OW > a.hw ? a.OH * a.OW : a.hw) < (1l << 40), ERR_INVALID, "gemm: M too large for fastdiv");
  const bool conv = !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.stride_w <= 1 && a.pad == 0);
  const bool apro = a.a_scale != nullptr;
  MTGV_CHECK(!(apro && conv), ERR_INVALID, "gemm: GRN prologue only on 1x1");
  MTGV_CHECK(!(apro && a.act != ACT_NONE), ERR_INVALID, "gemm: GRN prologue is only combined with a linear epilogue");
  MTGV_CHECK(a.a_shift == nullptr, ERR_INVALID, "gemm: fold the GRN shift into the bias (fold_shift_into_bias_launch)");
  if (!conv) MTGV_CHECK(a.OH == a.H && a.OW == a.Wd, ERR_INVALID, "gemm: 1x1 geometry mismatch");
  if (apro || a.grn_part) MTGV_CHECK(a.hw > 0 && a.M % a.hw == 0, ERR_INVALID, "gemm: hw=%d must divide M=%d", a.hw, a.M);
  {
    const SpPlan sp = gemm_sp_plan(a);  // the LDS-DMA split kernel takes every launch it can run
    if (sp.cfg >= 0) {
      if (a.grn_part) MTGV_CHECK(a.hw > 0 && a.M % a.hw == 0, ERR_INVALID, "gemm: hw=%d must divide M=%d", a.hw, a.M);
      prof_begin(a, s, 1, g_prof.on ? gemm_sp_fill_bytes(a, sp) : 0.0);
      gemm_sp_launch(a, sp, s);
      prof_end(s);
      return;
    }
  }
  if (a.grn_part) {
    MTGV_CHECK(a.segmax >= gemm_grn_segmax(pl, a.hw), ERR_INVALID, "gemm: segmax too small");
    MTGV_CHECK(a.grn_unit_rows == 0 || a.grn_unit_rows == pl.bm(), ERR_RUNTIME,
               "gemm: GRN partials planned for %d-row units, this launch writes %d-row units", a.grn_unit_rows, pl.bm());
  }

  MTGV_CHECK(a.os_nq == 0, ERR_INVALID, "gemm: the grouped scatter epilogue (os_nq) exists on the LDS-DMA kernel only");
  MTGV_CHECK(a.W2 == nullptr, ERR_INVALID, "gemm: the chained 1x1 exists on the LDS-DMA kernel only");
  GemmDev g;
  g.a = a;
  if (gemm_precision() == GEMM_PREC_F16X3 && g.a.W_split == nullptr && a.strideW == 0)
    g.a.W_split = split_for_launch(a.W, a.K, &g.a.wscale);
  if (gemm_precision() != GEMM_PREC_F16X3) g.a.W_split = nullptr, g.a.wscale = nullptr;
  g.d_ohw = make_fastdiv((uint32_t)(a.OH * a.OW));
  g.d_ow = make_fastdiv((uint32_t)a.OW);
  g.d_cin = make_fastdiv((uint32_t)a.Cin);
  g.d_kwcin = make_fastdiv((uint32_t)(a.KW * a.Cin));
  g.d_hw = make_fastdiv((uint32_t)(a.hw > 0 ? a.hw : 1));
  g.d_cw = make_fastdiv((uint32_t)(a.crop_w > 0 ? a.crop_w : 1));
  g.tiles_m = pl.tiles_m;
  g.tiles_n = pl.tiles_n;
  g.nseg_max = a.hw > 0 ? std::min(pl.bm(), (pl.bm() - 1) / a.hw + 2) : 1;
  {
    const double lim = 4294967296.0 - 65536.0;
    const double a_bytes = (double)a.M * a.c_total * 4.0 + (double)a.c_off * 4.0, w_bytes = (double)a.N * a.K * 4.0;
    const double s_bytes = a.a_scale != nullptr && a.hw > 0 ? (double)(a.M / a.hw) * a.K * 4.0 : 0.0;
    g.off32_ok = a_bytes < lim && w_bytes < lim && s_bytes < lim;
  }
  g.remap = !(a.os == 1 && a.oy == 0 && a.ox == 0 && a.OH2 == a.OH && a.OW2 == a.OW);
  const int grid = pl.tiles_m * pl.tiles_n;

  if (a.topk > 0) {
    MTGV_CHECK(pl.tm == 1 && pl.tn == 2 && pl.bk == 16 && !conv && !apro, ERR_INVALID, "gemm: top-k epilogue needs the 128x64x16 tile");
    MTGV_CHECK(a.cand_s != nullptr && a.cand_i != nullptr && a.topk <= 128, ERR_INVALID, "gemm: bad top-k arguments");
  }
  prof_begin(a, s, 0, (double)grid * a.batch * (pl.bm() + pl.bn()) * a.K * 4.0);
  const bool found = gemm_precision() == GEMM_PREC_F16X3 ? gemm_dispatch_f16x3(g, pl, conv, apro, grid, s)
                                                         : gemm_dispatch_f32(g, pl, conv, apro, grid, s);
  if (found) {
    HIP_OK(hipGetLastError());
    prof_end(s);
    return;
  }
  MTGV_CHECK(false, ERR_INVALID, "gemm: no kernel for tile tm=%d tn=%d bk=%d", pl.tm, pl.tn, pl.bk);
}

// ---------------------------------------------------------------------------
// GRN finalize: partials -> per-(image, channel) multiplier.  One block per image.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grn_finalize_kernel(const float* __restrict__ part, int bm, int segmax, int hw, int N,
                                                          FastDiv d_hw, FastDiv d_bm, const float* __restrict__ gamma,
                                                          float* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* gx = sm;        // [N]
  float* red = sm + N;   // [256]
  const int img = blockIdx.x, tid = threadIdx.x;
  const int t_first = (int)fdiv((uint32_t)(img * hw), d_bm);
  const int t_last = (int)fdiv((uint32_t)((img + 1) * hw - 1), d_bm);
  float local = 0.f;
  for (int n = tid; n < N; n += 256) {
    float sum = 0.f;
    for (int t0 = t_first; t0 <= t_last; t0 += 8) {  // 8 loads in flight, added in unit order (same sum as one by one)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u;
        const int seg = img - (int)fdiv((uint32_t)(t * bm), d_hw);
        v[u] = t <= t_last ? part[((long)t * segmax + seg) * N + n] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
    const float gval = sqrtf(sum);
    gx[n] = gval;
    local += gval;
  }
  red[tid] = local;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const float denom = red[0] / (float)N + 1e-6f;
  for (int n = tid; n < N; n += 256) scale[(long)img * N + n] = gamma[n] * (gx[n] / denom) + 1.0f;
}

// out[n] = bias[n] + sum_k W[n][k] * shift[k]: folds GRN's "+ beta" into the bias of the Linear that follows
__global__ __launch_bounds__(256) void fold_shift_kernel(const float* __restrict__ W, const float* __restrict__ shift,
                                                        const float* __restrict__ bias, float* __restrict__ out, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += W[(long)n * K + k] * shift[k];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
  if (lane == 0) out[n] = acc + (bias != nullptr ? bias[n] : 0.f);
}

void fold_shift_into_bias_launch(const float* W, const float* shift, const float* bias, float* out, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(fold_shift_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, shift, bias, out, N, K);
  HIP_OK(hipGetLastError());
}

void grn_finalize_launch(const float* part, const GrnLayout& l, int n_img, int hw, int N, const float* gamma, float* scale,
                         hipStream_t s) {
  const size_t lds = (size_t)(N + 256) * sizeof(float);
  MTGV_CHECK(lds <= 160 * 1024, ERR_INVALID, "grn_finalize: N=%d too large", N);
  hipLaunchKernelGGL(grn_finalize_kernel, dim3(n_img), dim3(256), lds, s, part, l.unit_rows, l.segmax, hw, N,
                     make_fastdiv((uint32_t)hw), make_fastdiv((uint32_t)l.unit_rows), gamma, scale);
  HIP_OK(hipGetLastError());
}

void grn_finalize_launch(const float* part, const GemmPlan& p, int n_img, int hw, int N, const float* gamma, float* scale,
                         hipStream_t s) {
  GrnLayout l;
  l.unit_rows = p.bm();
  l.segmax = gemm_grn_segmax(p, hw);
  grn_finalize_launch(part, l, n_img, hw, N, gamma, scale, s);
}

}  // namespace mtgv
