// Implicit-GEMM convolution / linear layer on the gfx950 f32 matrix cores.
//
//   Out[orow(m)][o_off + n] = act( sum_k A(m,k) * W[n][k] + bias[n] ) (+ res[m][n])
//
// A(m,k) is gathered on the fly from an NHWC activation tensor (no im2col):
//   m -> (img, oh, ow), k -> (kh, kw, c), A = in[img][oh*s+kh-p][ow*s+kw-p][c_off+c].
// W is [N][K] row-major with K ordered (kh, kw, c) - the same order nn.Linear
// uses for 1x1 (convnextv2.py:202-207) and what conv weights are repacked to
// at load time.
#pragma once
#include "common.h"

namespace mtgv {

struct GemmArgs {
  const float* A = nullptr;     // NHWC activation base
  int a_fmt = 0;                // 0: f32; 1: SP8 (sp8.h) - only launches gemm_sp_plan accepts (gemm_sp_takes_sp8)
  const float* W = nullptr;     // [N][K]
  // f16x3 only, optional: W with every aligned group of 4 floats replaced by their 4 fp16 hi + 4 fp16 lo halves
  // (same byte layout, so the same offsets address it).  gemm_launch fills it in for registered weights.
  const float* W_split = nullptr;
  // per-column power-of-two scales of W_split (registered rows are stored scaled, gemm_sp.h); filled in by gemm_launch
  const float* wscale = nullptr;
  // f16x3 only: A is multiplied by a_mul (a power of two) before it is split and the accumulator by a_unmul = 1/a_mul.
  // For launches whose activations may exceed the fp16 range (|a| > 65504); 1 everywhere on the recognition path.
  float a_mul = 1.0f, a_unmul = 1.0f;
  float* Out = nullptr;
  int out_fmt = 0;              // 0: f32; 1: SP8 (LDS-DMA kernel only)
  const float* bias = nullptr;  // [N] or null
  const float* res = nullptr;   // residual [M][ldr] or null
  int res_fmt = 0;              // format of the residual rows (0 f32, 1 SP8)
  int M = 0, N = 0, K = 0;

  // A gather geometry (1x1/linear: KH=KW=1, stride=1, pad=0, H*W = rows per image)
  int H = 1, Wd = 1;            // input spatial size
  int c_total = 0;              // input channel stride (floats per pixel)
  int c_off = 0;                // first input channel used
  int Cin = 0;                  // channels consumed per tap (K = KH*KW*Cin)
  int KH = 1, KW = 1, stride = 1, pad = 0;
  int stride_w = 0;             // 0 = same as stride (the 4x4 stem views rows of 12 floats as pixels: stride 4 x 1)
  int OH = 1, OW = 1;           // output grid that m enumerates

  // output mapping: orow = (img*OH2 + oh*os + oy)*OW2 + ow*os + ox
  int ldo = 0, o_off = 0;
  int os = 1, oy = 0, ox = 0, OH2 = 1, OW2 = 1;
  // os_nq > 0 (LDS-DMA kernel only): the N columns are os * os groups of os_nq channels and group q scatters to
  // (oy, ox) = (q / os, q % os), channel n % os_nq - a whole ConvTranspose2d(k = s = os) as ONE launch that reads its
  // input once (W rows ordered (kh, kw, cout); bias repeated per group).  os_nq % 8 == 0.
  int os_nq = 0;
  // Chained 1x1 (LDS-DMA kernel only, gemm_sp_chain_ok): a second layer W2 [N2][N] (+ bias2, act2) applied to this launch's
  // activated output rows inside the epilogue; only Out2 is stored (Out may be null).  N in {32, 64, 96}, N2 % 32 == 0, N2 <= N.
  const float* W2 = nullptr;
  const float* bias2 = nullptr;
  float* Out2 = nullptr;
  int N2 = 0, ldo2 = 0, o_off2 = 0, out_fmt2 = 0, act2 = ACT_NONE;
  int ldr = 0;
  int act = ACT_NONE;

  // GRN (convnextv2.py:171-174): per-(tile, image-segment, n) partial sums of out^2
  float* grn_part = nullptr;    // [tiles_m][segmax][N]
  int hw = 0;                   // rows per image for GRN / prologue segmentation
  int segmax = 0;
  int grn_unit_rows = 0;        // rows per partial unit the caller planned for (GrnLayout::unit_rows); 0: not checked
  // GRN apply fused into the A load of pwconv2: a' = a * a_scale[img][k] + a_shift[k]
  const float* a_scale = nullptr;
  const float* a_shift = nullptr;

  // batched launch (grid.y = batch index z): operands advance by these element strides per z
  int batch = 1;
  long strideA = 0, strideW = 0, strideO = 0;
  const int* m_count = nullptr;       // [batch] valid rows of batch z (rows beyond are neither read nor written)
  // crop_mask of process_mask: rows are detections, columns mask pixels n = py*crop_w + px;
  // values outside the row's box (xyxy image pixels * crop_scale, x in [x1,x2), y in [y1,y2)) become 0
  const float* crop_boxes = nullptr;  // [batch][crop_rows][4]
  int crop_rows = 0;                  // boxes per batch (>= M)
  float crop_scale = 0.f;
  int crop_w = 1;

  // match path: instead of storing Out, keep the top-`topk` (score desc, id asc)
  // columns of every row per column tile: cand[m][tile_n][topk]
  float* cand_s = nullptr;
  int* cand_i = nullptr;
  int topk = 0;
  // LayerNorm over the N outputs of every row fused into the epilogue (the encoder's stem: conv k4 s4 + LN,
  // convnextv2.py:253-256).  Only where gemm_ln_fusable() says so: one tile covers the whole row, whole tiles only.
  const float* ln_w = nullptr;
  const float* ln_b = nullptr;
  float ln_eps = 0.f;
};

struct GemmPlan {
  int tm, tn, bk;               // tile: BM = 128*tm, BN = 32*tn
  int tiles_m, tiles_n;
  int bm() const { return 128 * tm; }
  int bn() const { return 32 * tn; }
};

// heavy_epilogue: the launch applies an activation (transcendentals per output element)
// scaled_a: the launch applies the GRN multiplier to A (a_scale)
GemmPlan gemm_plan(int M, int N, int K, bool heavy_epilogue = false, bool scaled_a = false);
int gemm_grn_segmax(const GemmPlan& p, int hw);
size_t gemm_grn_part_floats(const GemmPlan& p, int N, int hw);
void gemm_launch(const GemmArgs& a, const GemmPlan& p, hipStream_t s);
// can this launch (no activation, no residual, not routed to the LDS-DMA kernel) normalise its rows in the epilogue?
bool gemm_ln_fusable(const GemmArgs& a, const GemmPlan& p);

// Operand precision of every GEMM launch of the process (see gemm_f32.hip): GEMM_PREC_F32 = f32 MFMA,
// GEMM_PREC_F16X3 = fp16 hi+lo split, three fp16 MFMAs per product.  Initial value from MTGV_GEMM_PREC=f32|f16x3.
enum { GEMM_PREC_F32 = 0, GEMM_PREC_F16X3 = 1 };
int gemm_precision();
void gemm_set_precision(int prec);
// out[n] = bias[n] + W[n][:] . shift  (GRN beta folded into the next Linear's bias; a_shift is not applied by the GEMM)
void fold_shift_into_bias_launch(const float* W, const float* shift, const float* bias, float* out, int N, int K, hipStream_t s);

// Pre-split copies of constant B operands (weights, the bank) for the f16x3 mode: the loader then moves 16 bytes of
// ready fp16 halves instead of converting the same weights in every block that uses them.  An owner registers the
// base pointer of a buffer it allocated, refreshes a range after writing it, and unregisters before freeing.
// gemm_launch looks W up by exact base pointer; unregistered operands are split on the fly as before.
// row_k > 0 (a multiple of 8): the buffer holds rows of row_k floats; an SP8 copy with per-row scales is kept as well
// and launches with K == row_k go to the LDS-DMA kernel (gemm_sp.h).
void gemm_split_register(const float* W, size_t n_floats, int row_k = 0);
void gemm_split_refresh(const float* W, size_t offset_floats, size_t n_floats, hipStream_t s);
void gemm_split_unregister(const float* W);
const float* gemm_split_lookup(const float* W);

// launch profiler for the roofline measurement (off by default; adds two event records per launch)
void gemm_profile_enable(bool on);
bool gemm_profile_enabled();
void gemm_profile_read(double* ms, double* flops, long* launches);
void gemm_profile_dump(const char* path);
double gemm_profile_bytes();  // compulsory operand + result bytes of the launches recorded since enable
// For launches that do the work of a GEMM layer outside gemm_launch (mlp_fused.hip): record `a`'s shape (M, N, K, act,
// a_scale / grn_part / res as flags) with the given LDS fill and compulsory bytes, bracketed by events on s.
void gemm_profile_begin(const GemmArgs& a, hipStream_t s, int sp, double fill, double bytes);
void gemm_profile_end(hipStream_t s);

// Layout of the GRN partial sums a launch with these arguments writes (grn_part itself need not be set yet):
// [ceil(M / unit_rows)][segmax][N] floats.  Depends on which kernel gemm_launch will pick for the arguments.
struct GrnLayout {
  int unit_rows = 0, segmax = 0;
  size_t floats = 0;
};
GrnLayout gemm_grn_layout(const GemmArgs& a, const GemmPlan& p);
// upper bound of GrnLayout::floats over every kernel / tile that could be chosen
size_t gemm_grn_part_floats_max(int M, int N, int hw);

// sum the partials of one GEMM into the GRN apply table
//   scale[img][n] = gamma[n] * Gx / (mean_n Gx + 1e-6) + 1,  Gx = sqrt(sum x^2)
void grn_finalize_launch(const float* part, const GemmPlan& p, int n_img, int hw, int N, const float* gamma,
                         float* scale, hipStream_t s);
void grn_finalize_launch(const float* part, const GrnLayout& l, int n_img, int hw, int N, const float* gamma, float* scale,
                         hipStream_t s);

}  // namespace mtgv
