// Host side of the LDS-DMA split-precision GEMM (gemm_sp_kernel.h): SP8 copies of constant B operands, tile
// selection, launch.  gemm_launch (gemm_f32.hip) routes every eligible f16x3 launch here.
#pragma once
#include "gemm_f32.h"

namespace mtgv {

// SP8 copy (+ per-row power-of-two scale) of a constant [rows][row_k] f32 operand, keyed by its base pointer.
// Owners call these through gemm_split_register / _refresh / _unregister (gemm_f32.h).
void sp8_register(const float* W, size_t n_floats, int row_k);
void sp8_refresh(const float* W, size_t offset_floats, size_t n_floats, hipStream_t s);
void sp8_unregister(const float* W);
// SP8 rows and scales for the operand at `W` (base or a row-aligned interior pointer of a registered buffer) when its
// rows are K long; false if there is none
bool sp8_lookup(const float* W, int K, const char** sp8, const float** wscale);

struct SpPlan {
  int cfg = -1;          // index into the instantiated tile configurations; -1: not eligible
  int bm = 0, bn = 0;
  int unit_rows = 0;     // rows of one GRN partial unit (a wave's rows)
  int tiles_m = 0, tiles_n = 0;
};

// f16x3 operand mode with the LDS-DMA kernel enabled (MTGV_GEMM_SP != 0): executors then keep activations in SP8
bool gemm_sp_active();
bool topk_sp_on();  // MTGV_SP_TOPK != 0: bank matches of >= 128 queries on the LDS-DMA kernel
// Can (and should) this launch run on the SP kernel?  a.a_fmt says how A is stored.
SpPlan gemm_sp_plan(const GemmArgs& a);
// true when a dense [M][K] x [N][K]^T launch with these sizes would take SP8 activations (producer kernels ask before
// choosing their output format)
bool gemm_sp_takes_sp8(const float* W, int M, int N, int K, int lda, int c_off);
void gemm_sp_launch(const GemmArgs& a, const SpPlan& pl, hipStream_t s);
// can the launch described by `a` (W2 / bias2 / Out2 / N2 ... set) run with its second layer chained into the epilogue?
bool gemm_sp_chain_ok(const GemmArgs& a);
bool gemm_sp_topk_layout(const GemmArgs& a, int* slots, int* cols);  // candidate groups of a top-k launch the SP kernel takes
double gemm_sp_fill_bytes(const GemmArgs& a, const SpPlan& pl);  // LDS fill bytes of the launch (profiling aid)
void gemm_sp_stamps_dump(const char* path);  // tuning aid, see gemm_sp.hip

// Approximate scores for the bank match's first pass: q_hi [b][K] and bank_hi [N][K] are fp16 rows (the hi halves of the
// exact operands, the bank's scaled per row by 1 / wscale), one fp16 MFMA per product; per (query, 96-column wave
// range) the kp best (score, column) pairs go to cand_s / cand_i like the exact top-k launch's.  128 x 192 tiles;
// b >= 128, N >= 192, K % 64 == 0.  *slots = candidate groups per query.
void gemm_sp_topk_hi16_launch(const void* q_hi, const void* bank_hi, const float* wscale, int b, int N, int K, int kp, float* cand_s,
                              int* cand_i, int* slots, hipStream_t s);
int gemm_sp_topk_hi16_slots(int N);
int gemm_sp_topk_hi16_range_cols();  // columns per candidate group: group w covers columns [w * cols, (w + 1) * cols)

// >= 256 zero bytes on the current device (K tails and padding taps of the DMA paths read them)
const char* sp_zero_page();

// f32 [rows][K] -> SP8 rows, unscaled (activations; test surface)
void sp8_pack_plain_launch(const float* in, void* out, long rows, int K, hipStream_t s);

}  // namespace mtgv
