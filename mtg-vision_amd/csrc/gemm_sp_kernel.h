// Split-precision GEMM, second generation: operands arrive in LDS already split (SP8, sp8.h) and the K loop is
// pure LDS-DMA + ds_read_b128 + MFMA.
//
//   Out[orow(m)][o_off + n] = act( sum_k A(m,k) * W[n][k] * wscale[n] + bias[n] ) (+ res[m][n])
//
// Block: WM x WN waves; wave tile (32*TM) rows x (32*TN) columns; K in stages of 16*KS (RB = 64*KS bytes per staged
// row), a ring of NST LDS buffers (NST - 1 stages in flight), one raw s_barrier per stage.
//   B (weights)          always by LDS-DMA (global_load_lds_dwordx4) from the registered SP8 copy.
//   A, AMODE 0 (DMA)     SP8 activation rows written by the producer kernel: LDS-DMA.  Dense rows [M][lda].
//   A, AMODE 2 (CONV)    SP8 NHWC activations gathered by the DMA's per-lane source address (implicit GEMM, no
//                        im2col): m -> (img, oh, ow), k -> (kh, kw, c); taps outside the image and the K tail read a
//                        zero page.  Cin % 8 == 0 keeps every 8-channel chunk inside one tap.
//   A, AMODE 5 (WINDOW)  3x3 / stride 1 / pad 1 convs with Cin % 32 == 0: per 32-channel slice the tile's input window
//                        (BM + 2 W + 2 pixels x 128 B) is filled once and the nine taps read shifted fragments out of it,
//                        zeroed by predicate where the tap leaves the image; only the taps' weight stages ring.
//                        Accumulation order: channel slice outer, tap inner.
//   A, AMODE 3 / 4       f32 rows by LDS-DMA; the wave that feeds a fragment to the MFMAs multiplies it by the per-image
//                        multipliers (AMODE 3: GRN apply, convnextv2.py:171-174; one extra 1 KB DMA piece per stage holds
//                        [8 images][32 k]) and splits it into hi / lo on the spot.
//   EPI 32 (chain)       a following 1x1 conv / Linear whose input is this launch's whole output row (N == BN <= 96, one wave
//                        holds all N columns of its 32 rows: WN == 1) runs inside the epilogue: the activated values are
//                        split into SP8 pieces and written to LDS as the wave's own A stages, W2 arrives by DMA while
//                        that happens, a second MFMA pass multiplies them and a second read-back stores Out2 - the first
//                        layer's output never goes to HBM (detector: C2f cv1 behind a stride-2 conv, Proto cv3 behind
//                        cv2, the heads' final 1x1 behind their 3x3).  Same SP8 values, same product order as two
//                        launches: bit-identical.
//   A, AMODE 6 (HI16)    A and B are plain fp16 rows (the hi halves only, 2 bytes per element): a stage is 64 k, one MFMA
//                        per k16 step.  The approximate first pass of the bank match (match.hip); K % 64 == 0.
//   A, AMODE 1 (REG)     f32 rows: global -> VGPR (issued before the stage's MFMAs) -> optional per-image multiplier
//                        -> split -> ds_write_b128 (after the MFMAs).  Fallback of 3 / 4: unaligned rows, range-guarded
//                        inputs (a_mul != 1), tiles spanning more than 8 images.
// LDS image of a stage: [row][SPR slots of 16 B], slot' = slot ^ sw(row), sw = (row>>1)&7 for 128-byte rows,
// (row>>2)&3 for 64-byte rows: every ds_read_b128 lane group covers all 64 banks once.  The DMA destination is
// lane-linear, so the swizzle is applied to the per-lane source address (and to the ds_write address in REG mode).
//
// MFMA orientation: weights are the first operand, so the accumulator has n on registers and m on lanes - a lane owns
// one output row and 4 consecutive columns per register group.  Epilogue: the accumulators of a 32-row slab are staged
// raw through LDS and read back 8 lanes per 128-byte row segment; a lane keeps the same 4 columns of every column block,
// so wscale, bias and the GRN sum(x^2) partials are lane-local registers, and scale / bias / activation / residual / SP8
// packing (v_cvt_pk_f16_f32 + a DPP exchange between the two lanes of an 8-column chunk) all happen on the read-back
// side, followed by whole-line stores.  EPI >= 0 fixes the epilogue's shape at compile time.
//
// Products: lo*hi + hi*lo + hi*hi per k16 step into the same accumulator, k ascending (WINDOW: slice-major): results
// do not depend on the tile configuration.
#pragma once
#include <type_traits>

#include "act.h"
#include "gemm_f32.h"
#include "sp8.h"

namespace mtgv {

// Experiment switch (build-time, tools only): wave priority around the phases of a main-loop stage.
//   MTGV_SP_PRIO == 1: s_setprio 1 while a wave issues its MFMAs; == 2: s_setprio 1 while it issues DMA and LDS reads.
#ifndef MTGV_SP_PRIO
#define MTGV_SP_PRIO 0
#endif
#define MTGV_SP_PRIO_MFMA(v) do { if (MTGV_SP_PRIO == 1) __builtin_amdgcn_s_setprio(v); } while (0)
#define MTGV_SP_PRIO_LOAD(v) do { if (MTGV_SP_PRIO == 2) __builtin_amdgcn_s_setprio(v); } while (0)
// Experiment switch: half-stage stagger of waves 4..7 of the eight-wave SP8 tile (MI355X_MICROARCH.md, "Two waves per SIMD",
// item 9): those waves run their second k16 step's MFMAs right after the NEXT barrier, while waves 0..3 issue DMA and read.
#ifndef MTGV_SP_STAGGER
#define MTGV_SP_STAGGER 0
#endif
// TIMING EXPERIMENTS ONLY (results are wrong): what a dense main-loop stage pays for its operand traffic.
//   MTGV_SP_EXP == 1: no DMA after the prologue (the loop computes on stale LDS): MFMA + LDS reads + conversion alone;
//   MTGV_SP_EXP == 2: DMA issued as usual but never waited for (vmcnt left alone): issue cost without the latency;
//   MTGV_SP_EXP == 3: DMA of every second stage only (half the issue cost and bytes), waits as usual;
//   MTGV_SP_EXP == 4: (f32 A) the next stage's DMA issued behind this stage's LDS reads instead of ahead of them (results right);
//   MTGV_SP_EXP == 5: every dense DMA piece with a quarter of its lanes (sp8.h): the instruction count without the bytes;
//   MTGV_SP_EXP == 6: (f32 A) no LDS reads for the GRN multipliers; 7: (f32 A) a third of the weight-fragment LDS reads;
//   MTGV_SP_EXP == 9: (f32 A) no scaling and no hi / lo split: the 48 vector instructions of a stage gone (operands are garbage).
#ifndef MTGV_SP_EXP
#define MTGV_SP_EXP 0
#endif

typedef float spf16 __attribute__((ext_vector_type(16)));

// One MFMA of the main loop.  MTGV_SP_MFMA16 (build-time, TIMING EXPERIMENT ONLY - the results are not the product): the
// same operand registers and the same FLOPs issued as two v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH.md, DVFS item 7: the
// clock the chip holds under matrix load depends on the MFMA shape), each k16 step on its own half of the accumulator.
#ifndef MTGV_SP_MFMA16
#define MTGV_SP_MFMA16 0
#endif
__device__ __forceinline__ spf16 sp_mfma(sp_h8 b, sp_h8 a, spf16 c, int ks) {
#if MTGV_SP_MFMA16
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int o = (ks & 1) * 8;
  f4 c0 = {c[o + 0], c[o + 1], c[o + 2], c[o + 3]}, c1 = {c[o + 4], c[o + 5], c[o + 6], c[o + 7]};
  c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) c[o + e] = c0[e], c[o + 4 + e] = c1[e];
  return c;
#else
  (void)ks;
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
#endif
}

struct SpDev {
  float* cand_s = nullptr;      // EPI 16 (top-k): per (row, column tile, wave column) the k best (score, column) pairs
  int* cand_i = nullptr;
  int topk = 0;
  long* stamps = nullptr;       // tuning aid (MTGV_SP_STAMPS): [tile][8] s_memtime at entry / first stage in / loop end / exit, HW_ID, XCC_ID, s_memrealtime at entry / exit
  const char* A = nullptr;      // AMODE 0/2: SP8 bytes; AMODE 1: f32
  long a_rowb = 0;              // bytes per A row (pixel)
  long a_offb = 0;              // byte offset of the first channel used
  const char* W = nullptr;      // SP8 [N][K]
  const float* wscale = nullptr;  // [N]
  const float* bias = nullptr;
  const void* res = nullptr;    // residual rows, f32 or SP8 (res_fmt); channel offset already applied
  long ldr = 0;                 // elements per residual row
  int res_fmt = 0;
  float* Out = nullptr;
  long ldo = 0;
  int o_off = 0;
  int out_fmt = 0;              // 0: f32, 1: SP8
  int M = 0, N = 0, K = 0;
  float* grn_part = nullptr;    // [units][segmax][N], unit = one wave's rows (32*TM)
  int hw = 1, segmax = 0;
  FastDiv d_hw;
  const float* a_scale = nullptr;  // AMODE 1: [M/hw][K]
  float a_mul = 1.0f, a_unmul = 1.0f;  // AMODE 1: power-of-two pre-scale of A (range guard) and its inverse
  const char* zero = nullptr;   // >= 16 zero bytes (K tail / padding taps of the DMA paths)
  int off32 = 0;                // A and W extents < 4 GB: dense pieces are addressed as uniform base + 32-bit lane offset
  int tiles_m = 0, tiles_n = 0;
  int act = 0;
  // AMODE 2 geometry
  int H = 1, Wd = 1, Cin = 0, KW = 1, stride = 1, pad = 0, OH = 1, OW = 1;
  FastDiv d_ohw, d_ow, d_cin, d_kw;
  // output row remap (ConvTranspose scatter): orow = (img*OH2 + oh*os + oy)*OW2 + ow*os + ox
  int remap = 0, os = 1, oy = 0, ox = 0, OH2 = 1, OW2 = 1;
  int nq = 0;      // > 0: column group q = n / nq scatters to (oy, ox) = (q / os, q % os), channel n % nq (GemmArgs::os_nq)
  FastDiv d_nq, d_os;
  // EPI 32 (chained 1x1): Out2[m][o_off2 + n2] = act2( sum_n act(this launch's output)[m][n] * W2[n2][n] * wscale2[n2] + bias2[n2] );
  // the first layer's output is not stored.  N == BN (one column tile), N2 % 32 == 0, N2 <= N.
  const char* W2 = nullptr;        // SP8 [N2][N]
  const float* wscale2 = nullptr;  // [N2]
  const float* bias2 = nullptr;    // [N2] or null
  float* Out2 = nullptr;
  long ldo2 = 0;
  int o_off2 = 0, out_fmt2 = 0, act2 = 0, N2 = 0;
};

typedef const __attribute__((address_space(1))) void* sp_gptr;
typedef __attribute__((address_space(3))) void* sp_lptr;

template <int WM, int WN, int TM, int TN, int KS, int NST, int AMODE, int ACT, int EPI>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 ? 4 : 2)) void gemm_sp_kernel(const SpDev g) {
#pragma clang fp contract(off)
  constexpr int NW = WM * WN, NT = 64 * NW, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr bool GEN = EPI < 0;  // epilogue shape read from the arguments (see the epilogue)
  static_assert(TM <= 2, "the epilogue names its slabs");
  static_assert(NST >= 2 && NST <= 4 && (AMODE != 1 || NST == 2), "ring depth; the register A path is two-deep");
  constexpr int RB = 64 * KS, RPP = 1024 / RB, SPR = RB / 16;
  constexpr bool AWIN = AMODE == 5;  // 3x3 / stride 1 / pad 1 conv: the tile's input window is staged once per 32 channels
  constexpr bool ADMA = AMODE != 1 && !AWIN;
  constexpr bool AF32 = AMODE == 3 || AMODE == 4;  // f32 rows by DMA, split into hi / lo when a fragment is read
  constexpr bool HI16 = AMODE == 6;                // fp16 rows on both sides: 2 bytes per element, 64 k per 128-byte stage row
  constexpr int EB = HI16 ? 2 : 4;                 // bytes per operand element in memory
  static_assert(!HI16 || (KS == 2 && EPI == 16), "fp16 rows: 128-byte stage rows, top-k epilogue");
  constexpr int SA = AWIN ? 0 : BM * RB, SB = BN * RB, SSC = AMODE == 3 ? 1024 : 0, STG = SA + SB + SSC;
  // (window conv: 32-channel stages with KS == 2, 16-channel stages - the detector's 16-channel bottlenecks - with KS == 1)
  static_assert(AMODE != 3 || KS == 2, "the scale image is one DMA piece: 8 images x 32 k");
  constexpr int PA = ADMA ? BM / RPP : 0, PB = BN / RPP, NP = PA + PB;
  constexpr int PPW = (NP + NW - 1) / NW;
  constexpr int CPS = BM * 2 * KS;                  // REG: 8-float chunks of A per stage
  constexpr int CPT = AMODE == 1 ? CPS / NT : 1;    // per thread
  static_assert(AMODE != 1 || CPS % NT == 0, "A chunks must divide over the threads");
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const long st0 = g.stamps != nullptr ? (long)__builtin_amdgcn_s_memtime() : 0;
  const long rt0 = g.stamps != nullptr ? (long)__builtin_amdgcn_s_memrealtime() : 0;  // 100 MHz, clock-independent
  long st1 = 0;
  int L;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, x = b & 7;
    L = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (b >> 3);
  }
  const int tile_n = L % g.tiles_n, tile_m = L / g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long wrowb = (long)g.K * EB;
  const int kchunks = g.K >> 3;                 // valid 8-float chunks per row
  constexpr int KPS = 16 * KS * (4 / EB);       // k per stage
  const int nk = (g.K + KPS - 1) / KPS;
  const bool ktail = (g.K % KPS) != 0;          // (never with fp16 rows: the host requires K % 64 == 0)

  // ---- DMA pieces: piece p = wave + NW*u of the stage image (A pieces first, then B) ----
  const char* src[PPW];   // dense A / B: address of the lane's slot in stage 0.  CONV A: pixel (img, 0, 0) of the row
  bool tailz[PPW];
  int cslot[PPW], cih0[PPW], ciw0[PPW];  // CONV A pieces: logical slot, first input row / column of the window
  const char* csrc[PPW];             // CONV A pieces, fast form: address of the window's first pixel at this lane's chunk
  const bool conv_fast = AMODE == 2 && g.Cin % (16 * KS) == 0;
  // Dense pieces (B always, A unless it is gathered) of a launch without a K tail: the DMA's address is a uniform
  // base (SGPR pair, advanced per stage by scalar adds) plus a 32-bit lane offset fixed for the whole tile - the
  // saddr form of global_load_lds: no vector arithmetic per piece and half the address bytes per instruction.
  uint32_t off32[PPW];
  const bool sfast = g.off32 != 0 && !ktail;
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    const int p = wave + NW * u;
    const bool isA = p < PA;
    const int pp = isA ? p : p - PA;
    const int row = pp * RPP + lane / SPR;
    const int sw = KS == 2 ? (row >> 1) & 7 : (row >> 2) & 3;
    const int slot = (lane % SPR) ^ sw;
    cslot[u] = slot, cih0[u] = 0, ciw0[u] = 0, csrc[u] = nullptr;
    {
      const int mm = m0 + row < g.M ? m0 + row : g.M - 1, nn = n0 + row < g.N ? n0 + row : g.N - 1;
      off32[u] = isA ? (uint32_t)mm * (uint32_t)g.a_rowb + (uint32_t)(slot * 16) : (uint32_t)nn * (uint32_t)wrowb + (uint32_t)(slot * 16);
    }
    if (isA) {
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;
      if constexpr (AMODE == 2) {
        const uint32_t img = fdiv((uint32_t)m, g.d_ohw);
        const uint32_t rem = (uint32_t)m - img * (uint32_t)(g.OH * g.OW);
        const uint32_t oh = fdiv(rem, g.d_ow);
        const uint32_t ow = rem - oh * (uint32_t)g.OW;
        cih0[u] = (int)oh * g.stride - g.pad;
        ciw0[u] = (int)ow * g.stride - g.pad;
        csrc[u] = g.A + (((long)img * g.H + cih0[u]) * g.Wd + ciw0[u]) * g.a_rowb + g.a_offb + slot * 16;
        src[u] = g.A + (long)img * g.H * g.Wd * g.a_rowb + g.a_offb + (slot & 1) * 16;
      } else {
        src[u] = g.A + (long)m * g.a_rowb + g.a_offb + slot * 16;
      }
    } else {
      int n = n0 + row;
      n = n < g.N ? n : g.N - 1;
      src[u] = g.W + (long)n * wrowb + slot * 16;
    }
    tailz[u] = ktail && ((nk - 1) * 2 * KS + (slot >> 1)) >= kchunks;
  }
  // AMODE 3: lane l of the scale piece fetches 16 bytes (4 k) of image img0 + l / 8
  const int img0 = AMODE == 3 ? (int)fdiv((uint32_t)m0, g.d_hw) : 0;
  const char* sc_src = nullptr;
  bool sc_tailz = false;
  if constexpr (AMODE == 3) {
    const int img_last = (int)fdiv((uint32_t)(g.M - 1), g.d_hw);
    const int im = img0 + (lane >> 3) < img_last ? img0 + (lane >> 3) : img_last;
    sc_src = reinterpret_cast<const char*>(g.a_scale + (long)im * g.K) + (lane & 7) * 16;
    sc_tailz = ktail && ((nk - 1) * 2 * KS + ((lane & 7) >> 1)) >= kchunks;
  }
  // AWIN: [window: win_px pixels x 128 B][weight ring]; otherwise the ring starts at the base
  const int win_px = AWIN ? ((BM + 2 * g.Wd + 2 + RPP - 1) / RPP * RPP) : 0;  // whole 1 KB pieces
  char* const ring = smem + (AWIN ? win_px * RB : 0);
  // one piece the general way: per-lane 64-bit source (conv gather, K tail -> zero page)
  auto issue_piece = [&](int u, int t, int buf) {
    const int p = wave + NW * u;
    const char* s;
    if (AMODE == 2 && p < PA && conv_fast) {
      // Cin is a multiple of the stage depth: the whole stage lies inside one tap, so tap, kh, kw and the first
      // channel are wave-uniform (scalar registers) and a lane only tests its pixel against the padding
      const uint32_t k0 = (uint32_t)t * (16u * KS);
      const uint32_t tap = fdiv(k0, g.d_cin);
      const uint32_t kh = fdiv(tap, g.d_kw);
      const uint32_t kw = tap - kh * (uint32_t)g.KW;
      const long tap_off = ((long)kh * g.Wd + kw) * g.a_rowb + (long)(k0 - tap * (uint32_t)g.Cin) * 4;
      const bool ok = (unsigned)(cih0[u] + (int)kh) < (unsigned)g.H && (unsigned)(ciw0[u] + (int)kw) < (unsigned)g.Wd;
      s = ok ? csrc[u] + tap_off : g.zero;
    } else if (AMODE == 2 && p < PA) {
      // chunk -> (tap, channel); the tap's pixel may fall into the zero padding
      const int kc = t * 2 * KS + (cslot[u] >> 1);
      const uint32_t k = (uint32_t)(kc < kchunks ? kc : 0) * 8u;
      const uint32_t tap = fdiv(k, g.d_cin);
      const uint32_t c = k - tap * (uint32_t)g.Cin;
      const uint32_t kh = fdiv(tap, g.d_kw);
      const uint32_t kw = tap - kh * (uint32_t)g.KW;
      const int ih = cih0[u] + (int)kh, iw = ciw0[u] + (int)kw;
      const bool ok = kc < kchunks && ih >= 0 && ih < g.H && iw >= 0 && iw < g.Wd;
      s = ok ? src[u] + ((long)ih * g.Wd + iw) * g.a_rowb + c * 4 : g.zero;
    } else {
      s = src[u] + (long)t * RB;
      if (t == nk - 1 && tailz[u]) s = g.zero;
    }
    // A pieces fill [0, SA), B pieces [SA, STG) (in REG mode the A region is written by ds_write instead)
    __builtin_amdgcn_global_load_lds((sp_gptr)s, (sp_lptr)(ring + buf * STG + (SA - PA * 1024) + p * 1024), 16, 0, 0);
  };
  auto issue = [&](int t, int buf) {
    if (sfast) {  // no K tail, 32-bit extents: dense pieces in the saddr form, straight-line
      const char* const abase = g.A + g.a_offb + (long)t * RB;  // uniform
      const char* const wbase = g.W + (long)t * RB;
#pragma unroll
      for (int u = 0; u < PPW; ++u) {
        const int p = wave + NW * u;
        if (NP % NW == 0 || p < NP) {
          if (AMODE == 2 && p < PA) issue_piece(u, t, buf);
          else sp_dma16_saddr(p < PA ? abase : wbase, off32[u], ring + buf * STG + (SA - PA * 1024) + p * 1024);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < PPW; ++u)
        if (NP % NW == 0 || wave + NW * u < NP) issue_piece(u, t, buf);
    }
    if constexpr (AMODE == 3) {  // the stage's slice of the per-image A multipliers: [8 images from img0][32 k]
      if (wave == NP % NW) {
        const char* sp = sc_src + (long)t * RB;
        if (t == nk - 1 && sc_tailz) sp = g.zero;
        __builtin_amdgcn_global_load_lds((sp_gptr)sp, (sp_lptr)(smem + buf * STG + SA + SB), 16, 0, 0);
      }
    }
  };

  // ---- REG A loader: chunk c = tid + NT*v -> row c / (2 KS), chunk-in-stage c % (2 KS) ----
  const char* a_ptr[CPT];
  const float* s_ptr[CPT];
  unsigned a_lds[CPT];
  int a_ch[CPT];
  sp_f4 ra[CPT][2], rs[CPT][2];
  if constexpr (AMODE == 1) {
#pragma unroll
    for (int v = 0; v < CPT; ++v) {
      const int c = tid + NT * v;
      const int row = c / (2 * KS), ch = c % (2 * KS);
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;
      a_ptr[v] = g.A + (long)m * g.a_rowb + g.a_offb + ch * 32;
      s_ptr[v] = g.a_scale != nullptr ? g.a_scale + (long)fdiv((uint32_t)m, g.d_hw) * g.K + ch * 8 : nullptr;
      const int sw = KS == 2 ? (row >> 1) & 7 : (row >> 2) & 3;
      a_lds[v] = (unsigned)(row * RB + (((2 * ch) ^ sw) << 4));  // hi piece; the lo piece sits at ^16
      a_ch[v] = ch;
    }
  }
  auto loadA = [&](int t) {
    if constexpr (AMODE == 1) {
#pragma unroll
      for (int v = 0; v < CPT; ++v) {
        const bool ok = !ktail || (t * 2 * KS + a_ch[v]) < kchunks;
        const sp_f4 z = {0.f, 0.f, 0.f, 0.f};
        ra[v][0] = ra[v][1] = z;
        rs[v][0] = rs[v][1] = z;
        if (ok) {
          const sp_f4* p = reinterpret_cast<const sp_f4*>(a_ptr[v] + (long)t * RB);
          ra[v][0] = p[0], ra[v][1] = p[1];
          if (g.a_scale != nullptr) {
            const sp_f4* q = reinterpret_cast<const sp_f4*>(s_ptr[v] + t * 16 * KS);
            rs[v][0] = q[0], rs[v][1] = q[1];
          }
        }
      }
    }
  };
  auto storeA = [&](int buf) {
    if constexpr (AMODE == 1) {
#pragma unroll
      for (int v = 0; v < CPT; ++v) {
        sp_h8 hi, lo;
        if (g.a_scale != nullptr)
          sp8_split8(ra[v][0] * rs[v][0] * g.a_mul, ra[v][1] * rs[v][1] * g.a_mul, hi, lo);
        else
          sp8_split8(ra[v][0] * g.a_mul, ra[v][1] * g.a_mul, hi, lo);
        char* const d = ring + buf * STG;
        *reinterpret_cast<sp_h8*>(d + a_lds[v]) = hi;
        *reinterpret_cast<sp_h8*>(d + (a_lds[v] ^ 16u)) = lo;
      }
    }
  };

  // AWIN: window pixel w holds input pixel m0 - Wd - 1 + w (output and input share the linear (img, y, x) index:
  // stride 1, pad 1); pieces of RPP pixels x RB bytes (8 x 128 B, or 16 x 64 B for 16-channel slices), 16-byte slots
  // swizzled by the pixel like the rows of a dense A stage
  auto issue_window = [&](int cc) {
    if constexpr (AWIN) {
      for (int pw = wave; pw < win_px / RPP; pw += NW) {
        const int w = pw * RPP + lane / SPR;
        const long gp = (long)m0 - g.Wd - 1 + w;
        const int slot = (lane % SPR) ^ (KS == 2 ? (w >> 1) & 7 : (w >> 2) & 3);
        const char* sp = (gp >= 0 && gp < (long)g.M) ? g.A + gp * g.a_rowb + g.a_offb + cc * RB + slot * 16 : g.zero;
        __builtin_amdgcn_global_load_lds((sp_gptr)sp, (sp_lptr)(smem + pw * 1024), 16, 0, 0);
      }
    }
  };
  // The first stages are requested now, before the rest of the set-up (accumulators, fragment offsets, epilogue
  // constants): their latency runs under it.
  if constexpr (AWIN) {
    // weights of the first NST - 1 (slice, tap) steps: step tau = 9 cc + tap reads K stage tap * (Cin / KPS) + cc
    const int ncc0 = g.Cin / (16 * KS);
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < 9 * ncc0) issue((s0 % 9) * ncc0 + s0 / 9, s0);
    issue_window(0);
  } else {
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < nk) issue(s0, s0);
    loadA(0);
  }
  int wy[TM], wx[TM];  // AWIN: (y, x) of this lane's output pixels
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    wy[i] = wx[i] = 0;
    if constexpr (AWIN) {
      int m = m0 + wm * TM * 32 + i * 32 + r;
      m = m < g.M ? m : g.M - 1;
      const uint32_t img = fdiv((uint32_t)m, g.d_ohw);
      const uint32_t rem = (uint32_t)m - img * (uint32_t)(g.OH * g.OW);
      wy[i] = (int)fdiv(rem, g.d_ow);
      wx[i] = (int)(rem - (uint32_t)wy[i] * (uint32_t)g.OW);
    }
  }

  spf16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int swr = KS == 2 ? (r >> 1) & 7 : (r >> 2) & 3;
  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (unsigned)((wm * TM * 32 + i * 32 + r) * RB);
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = (unsigned)(SA + (wn * TN * 32 + j * 32 + r) * RB);
  unsigned sc_off[TM];  // AMODE 3: this lane's row of the scale image, at its half of the k step
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    sc_off[i] = 0;
    if constexpr (AMODE == 3) {
      int m = m0 + wm * TM * 32 + i * 32 + r;
      m = m < g.M ? m : g.M - 1;
      sc_off[i] = (unsigned)(SA + SB + ((int)fdiv((uint32_t)m, g.d_hw) - img0) * 128 + h * 32);
    }
  }

  // Epilogue geometry (see the epilogue): on the read-back side a lane owns columns nw0 + 32 j + 4 slot + 0..3 of
  // every row it touches; their scales and biases are fetched now, under the main loop.
  const int nw0 = n0 + wn * TN * 32;  // first column of this wave
  const int mw0 = m0 + wm * TM * 32;  // first row of this wave
  const int slot = lane & 7, lrow = lane >> 3;
  sp_f4 wsc[TN], bsv[TN];
  bool col_ok[TN];
  // (stagger experiment: waves 4..7 hold a k16 step's fragments across the barrier, so the column vectors are fetched
  // after the main loop instead of under it - 8 TN registers)
  constexpr bool COLVEC_LATE = MTGV_SP_STAGGER != 0 && WM * WN == 8 && AMODE == 0 && KS == 2;
  auto load_colvecs = [&]() {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw0 + j * 32 + slot * 4;
    col_ok[j] = n < g.N;
    sp_f4 w1 = {1.f, 1.f, 1.f, 1.f}, b0 = {0.f, 0.f, 0.f, 0.f};
    if (n + 4 <= g.N) {
      if (g.wscale != nullptr) w1 = *reinterpret_cast<const sp_f4*>(g.wscale + n);
      if (g.bias != nullptr) b0 = *reinterpret_cast<const sp_f4*>(g.bias + n);
    } else if (GEN && n < g.N) {  // ragged last quad (N % 4 != 0)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < g.N) {
          if (g.wscale != nullptr) w1[e] = g.wscale[n + e];
          if (g.bias != nullptr) b0[e] = g.bias[n + e];
        }
    }
    wsc[j] = w1 * g.a_unmul;
    bsv[j] = b0;
  }
  };
  if constexpr (!COLVEC_LATE) load_colvecs();


  // a wave whose rows all lie beyond M (ragged last tile, tiny-M problems) skips its MFMAs
  const bool wave_active = m0 + wm * TM * 32 < g.M;

  // DMA pieces this wave issues per stage: the waits below count them (vmcnt retires in order)
  int my_pieces = 0;
#pragma unroll
  for (int u = 0; u < PPW; ++u) my_pieces += (NP % NW == 0 || wave + NW * u < NP) ? 1 : 0;
  if (AMODE == 3 && wave == NP % NW) my_pieces += 1;
  constexpr int AHEAD = NST - 2;  // stages that may still be in flight when stage t is consumed
  auto wait_stage = [&](bool tail) {
    // steady state: everything but the AHEAD youngest stages has landed; near the end fewer stages are outstanding
    // than that, so the tail waits for all of them
    if (MTGV_SP_EXP == 2 && !tail) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (AHEAD == 0 || tail) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else if (my_pieces == PPW) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW * AHEAD) : "memory");
    } else if (my_pieces == PPW + 1) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PPW + 1) * AHEAD) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PPW > 1 ? PPW - 1 : 0) * AHEAD) : "memory");
    }
  };

  if constexpr (AWIN) {
    // per 32-channel slice: stage the window once, then 9 taps x 2 k-steps out of it while the taps' weight stages
    // ring through NST buffers (NST - 1 taps ahead: a tap's 6 TN MFMAs are far shorter than a DMA round trip, so the
    // two-deep ring stalled on every tap).  Accumulation order: channel slice outer, tap inner.
    const int ncc = g.Cin / (16 * KS);
    const int ntau = 9 * ncc;
    const char* const win = smem;
    int buf = 0, nbuf = NST - 1;
    int tau = 0;
    for (int cc = 0; cc < ncc; ++cc) {
      if (cc > 0) {
        __builtin_amdgcn_s_barrier();  // every wave is done with the previous slice's window
        issue_window(cc);
      }
      for (int tap = 0; tap < 9; ++tap, ++tau) {
        // this step's weights have landed; the window too at tap 0 (it is the youngest request: full drain)
        if (tap == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else wait_stage(tau + NST - 1 > ntau);
        __builtin_amdgcn_s_barrier();
        if (g.stamps != nullptr && cc == 0 && tap == 0) st1 = (long)__builtin_amdgcn_s_memtime();
        {
          const int ta = tau + NST - 1;  // step whose weights go into the buffer freed by the previous step
          if (ta < ntau) {
            const int ca = ta / 9;
            issue((ta - ca * 9) * ncc + ca, nbuf);
          }
        }
        if (wave_active) {
          const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
          const int shift = g.Wd + 1 + dy * g.Wd + dx;
          const char* const sb = ring + buf * STG;
          unsigned w_off[TM], w_sw[TM];
          bool w_ok[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int w = wm * TM * 32 + i * 32 + r + shift;
            w_off[i] = (unsigned)w * (unsigned)RB;
            w_sw[i] = KS == 2 ? (unsigned)(w >> 1) & 7u : (unsigned)(w >> 2) & 3u;
            w_ok[i] = (unsigned)(wy[i] + dy) < (unsigned)g.H && (unsigned)(wx[i] + dx) < (unsigned)g.Wd;
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ swr) << 4);
            const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ swr) << 4);
            sp_h8 ah[TM], al[TM], bh[TN], bl[TN];
            const sp_h8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const sp_h8 vh = *reinterpret_cast<const sp_h8*>(win + w_off[i] + (((ks * 4 + h * 2 + 0) ^ w_sw[i]) << 4));
              const sp_h8 vl = *reinterpret_cast<const sp_h8*>(win + w_off[i] + (((ks * 4 + h * 2 + 1) ^ w_sw[i]) << 4));
              ah[i] = w_ok[i] ? vh : z8;  // zero padding: the tap's pixel lies outside the image
              al[i] = w_ok[i] ? vl : z8;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              bh[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + shi);
              bl[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + slo);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int i = 0; i < TM; ++i) {
                acc[i][j] = sp_mfma(bl[j], ah[i], acc[i][j], ks);
                acc[i][j] = sp_mfma(bh[j], al[i], acc[i][j], ks);
                acc[i][j] = sp_mfma(bh[j], ah[i], acc[i][j], ks);
              }
          }
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
      }
    }
  } else {
  {
    storeA(0);
    int buf = 0;                      // ring slot of stage t
    int nbuf = NST - 1;               // ring slot of stage t + NST - 1
    constexpr bool STGR = MTGV_SP_STAGGER != 0 && NW == 8 && AMODE == 0 && KS == 2;
    const bool late = STGR && wave >= 4;
    sp_h8 hah[TM], hal[TM], hbh[TN], hbl[TN];  // (stagger) the second k16 step's fragments, held across the barrier
    auto frag_read = [&](const char* sb, int ks, sp_h8* fah, sp_h8* fal, sp_h8* fbh, sp_h8* fbl) {
      const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ swr) << 4);
      const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ swr) << 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fah[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + shi);
        fal[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + slo);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        fbh[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + shi);
        fbl[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + slo);
      }
    };
    auto frag_mfma = [&](const sp_h8* fah, const sp_h8* fal, const sp_h8* fbh, const sp_h8* fbl) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbl[j], fah[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[j], fal[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fbh[j], fah[i], acc[i][j], 0, 0, 0);
    };
    for (int t = 0; t < nk; ++t) {
      // stage t landed: this wave's DMA pieces (vmcnt) and REG-mode ds_writes (lgkmcnt), then everyone's (barrier).
      // The barrier also says every wave has finished reading the slot of stage t - 1, which is refilled next.
      wait_stage(t + NST - 1 > nk);
      __builtin_amdgcn_s_barrier();
      if (g.stamps != nullptr && t == 0) st1 = (long)__builtin_amdgcn_s_memtime();
      MTGV_SP_PRIO_LOAD(1);
      if constexpr (STGR) {
        if (late && wave_active && t > 0) {  // the previous stage's second k16 step, out of registers
          frag_mfma(hah, hal, hbh, hbl);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (t + NST - 1 < nk) {
        constexpr bool LATE_DMA = MTGV_SP_EXP == 4 && AF32 && KS == 2 && !HI16;  // (issued behind the stage's LDS reads, below)
        if (MTGV_SP_EXP != 1 && (MTGV_SP_EXP != 3 || (t & 1)) && !(LATE_DMA && wave_active)) issue(t + NST - 1, nbuf);
        loadA(t + 1);
      }
      if (STGR && late) {
        if (wave_active) {
          const char* const sb = ring + buf * STG;
          sp_h8 ah[TM], al[TM], bh[TN], bl[TN];
          frag_read(sb, 0, ah, al, bh, bl);
          frag_mfma(ah, al, bh, bl);
          frag_read(sb, 1, hah, hal, hbh, hbl);
        }
      }
      if (wave_active && HI16) {  // four k16 steps per stage, one product each
        const char* const sb = ring + buf * STG;
  #pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const unsigned so = (unsigned)(((ks * 2 + h) ^ swr) << 4);
          sp_h8 ah[TM], bh[TN];
  #pragma unroll
          for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + so);
  #pragma unroll
          for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + so);
  #pragma unroll
          for (int j = 0; j < TN; ++j)
  #pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bh[j], ah[i], acc[i][j], ks);
        }
      }
      if constexpr (AF32 && KS == 2 && !HI16) {
        // f32 A rows: both k16 steps' fragments are requested first, and the second step's scale + split (24 vector
        // instructions per fragment) is issued in the shadow of the first step's MFMAs - an in-order wave otherwise
        // converts, then multiplies, and the matrix pipe idles through every conversion.  Same products, same order.
        __builtin_amdgcn_sched_barrier(0);  // the next stage's DMA above stays ahead of this stage's arithmetic
        if (wave_active) {
          const char* const sb = ring + buf * STG;
          sp_f4 xa[2][TM][2], xs[2][TM][2];
          sp_h8 bh[2][TN], bl[2][TN];
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ swr) << 4);
            const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ swr) << 4);
  #pragma unroll
            for (int i = 0; i < TM; ++i) {
              xa[ks][i][0] = *reinterpret_cast<const sp_f4*>(sb + a_off[i] + shi);
              xa[ks][i][1] = *reinterpret_cast<const sp_f4*>(sb + a_off[i] + slo);
              if constexpr (AMODE == 3) {
                if (MTGV_SP_EXP == 6) {  // experiment: no LDS reads for the multipliers
                  xs[ks][i][0] = xs[ks][i][1] = sp_f4{1.f, 1.f, 1.f, 1.f};
                } else {
                  xs[ks][i][0] = *reinterpret_cast<const sp_f4*>(sb + sc_off[i] + ks * 64);
                  xs[ks][i][1] = *reinterpret_cast<const sp_f4*>(sb + sc_off[i] + ks * 64 + 16);
                }
              }
            }
  #pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (MTGV_SP_EXP == 7 && j > 0) {  // experiment: one column block's weight fragments for all (a third of the B reads)
                bh[ks][j] = bh[ks][0], bl[ks][j] = bl[ks][0];
              } else {
                bh[ks][j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + shi);
                bl[ks][j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + slo);
              }
            }
          }
          if constexpr (MTGV_SP_EXP == 4) {  // experiment: the stage's LDS latency runs under the next stage's DMA issue
            __builtin_amdgcn_sched_barrier(0);
            if (t + NST - 1 < nk) issue(t + NST - 1, nbuf);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (AMODE == 3 && MTGV_SP_EXP != 9) {
  #pragma unroll
            for (int ks = 0; ks < 2; ++ks)
  #pragma unroll
              for (int i = 0; i < TM; ++i) xa[ks][i][0] = xa[ks][i][0] * xs[ks][i][0], xa[ks][i][1] = xa[ks][i][1] * xs[ks][i][1];
          }
          MTGV_SP_PRIO_LOAD(0);
          MTGV_SP_PRIO_MFMA(1);
          sp_h8 ah[2][TM], al[2][TM];
  #pragma unroll
          for (int i = 0; i < TM; ++i) {
            if (MTGV_SP_EXP == 9) ah[0][i] = __builtin_bit_cast(sp_h8, xa[0][i][0]), al[0][i] = __builtin_bit_cast(sp_h8, xa[0][i][1]);  // experiment: no conversion
            else sp8_split8_mix(xa[0][i][0], xa[0][i][1], ah[0][i], al[0][i]);
          }
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) {
  #pragma unroll
              for (int i = 0; i < TM; ++i) {
                if (MTGV_SP_EXP == 9) ah[1][i] = __builtin_bit_cast(sp_h8, xa[1][i][0]), al[1][i] = __builtin_bit_cast(sp_h8, xa[1][i][1]);
                else sp8_split8_mix(xa[1][i][0], xa[1][i][1], ah[1][i], al[1][i]);
              }
            }
  #pragma unroll
            for (int j = 0; j < TN; ++j)
  #pragma unroll
              for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bl[ks][j], ah[ks][i], acc[i][j], ks);
  #pragma unroll
            for (int j = 0; j < TN; ++j)
  #pragma unroll
              for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bh[ks][j], al[ks][i], acc[i][j], ks);
  #pragma unroll
            for (int j = 0; j < TN; ++j)
  #pragma unroll
              for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bh[ks][j], ah[ks][i], acc[i][j], ks);
          }
          // issue order: every LDS read, the first step's conversion, then the first step's MFMAs one by one, each
          // followed by a share of the second step's conversion
          constexpr int CV = (AMODE == 3 ? 8 : 0) + 16;  // vector instructions per fragment: scale, split
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM * (AMODE == 3 ? 4 : 2) + 2 * TN), 0);  // DS read
          __builtin_amdgcn_sched_group_barrier(0x002, CV * TM, 0);                                     // VALU
  #pragma unroll
          for (int q = 0; q < 3 * TM * TN; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1 + MTGV_SP_MFMA16, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, (CV * TM + 3 * TM * TN - 1) / (3 * TM * TN), 0);  // VALU
          }
          MTGV_SP_PRIO_MFMA(0);
        }
      } else if (wave_active && !HI16 && !(STGR && late)) {
        const char* const sb = ring + buf * STG;
  #pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ swr) << 4);
          const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ swr) << 4);
          sp_h8 ah[TM], al[TM], bh[TN], bl[TN];
  #pragma unroll
          for (int i = 0; i < TM; ++i) {
            if constexpr (AF32) {  // 8 consecutive k of row r as f32: (scale,) split, and the fragments are ready
              sp_f4 x0 = *reinterpret_cast<const sp_f4*>(sb + a_off[i] + shi);
              sp_f4 x1 = *reinterpret_cast<const sp_f4*>(sb + a_off[i] + slo);
              if constexpr (AMODE == 3) {
                x0 = x0 * *reinterpret_cast<const sp_f4*>(sb + sc_off[i] + ks * 64);
                x1 = x1 * *reinterpret_cast<const sp_f4*>(sb + sc_off[i] + ks * 64 + 16);
              }
              sp8_split8_mix(x0, x1, ah[i], al[i]);  // (same values as sp8_split8, 8 vector instructions fewer per 8 elements)
            } else {
              ah[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + shi);
              al[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + slo);
            }
          }
  #pragma unroll
          for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + shi);
            bl[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + slo);
          }
          // small cross terms first, the hi*hi product last (per accumulator); the three products of an accumulator are
          // issued TM * TN instructions apart, so no MFMA waits on the one just before it
          if (ks == 0) { MTGV_SP_PRIO_LOAD(0); MTGV_SP_PRIO_MFMA(1); }
  #pragma unroll
          for (int j = 0; j < TN; ++j)
  #pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bl[j], ah[i], acc[i][j], ks);
  #pragma unroll
          for (int j = 0; j < TN; ++j)
  #pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bh[j], al[i], acc[i][j], ks);
  #pragma unroll
          for (int j = 0; j < TN; ++j)
  #pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = sp_mfma(bh[j], ah[i], acc[i][j], ks);
        }
        MTGV_SP_PRIO_MFMA(0);
      }
      MTGV_SP_PRIO_LOAD(0);
      if (AMODE == 1 && t + 1 < nk) storeA(buf ^ 1);
      buf = buf + 1 == NST ? 0 : buf + 1;
      nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    if constexpr (STGR) {
      if (late && wave_active && nk > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        frag_mfma(hah, hal, hbh, hbl);
      }
    }
  }
  }

  // ---- epilogue ----
  auto activate = [&](float x) -> float {
    if constexpr (ACT == ACT_NONE) return x;
    else if constexpr (ACT == ACT_GELU) return act_gelu(x);
    else if constexpr (ACT == ACT_MISH) return act_mish(x);
    else if constexpr (ACT == ACT_SILU) return act_silu(x);
    else return apply_act(x, g.act);
  };
  if constexpr (COLVEC_LATE) load_colvecs();
  __builtin_amdgcn_s_barrier();  // every wave is done with the ring: it becomes the store staging area
  const long st2 = g.stamps != nullptr ? (long)__builtin_amdgcn_s_memtime() : 0;

  if constexpr (EPI == 32) {
    static_assert(EPI != 32 || (WN == 1 && TM == 1 && KS == 2 && !HI16), "chained 1x1: one wave holds whole rows");
    // LDS after the main loop: [NW x 4 KB: one column block of each wave's accumulators][NW x TN x 4 KB: each wave's A2
    // stages, row-major 128-byte rows with the main loop's swizzle][TN stages x N2 rows x 128 B: W2]
    constexpr int CB = 32 * 128;
    char* const stg1 = smem + wave * CB;
    char* const a2 = smem + NW * CB + wave * (TN * CB);
    char* const w2 = smem + NW * CB * (1 + TN);
    const int n2g = g.N2 >> 3;  // 8-row DMA pieces per stage
    for (int p = wave; p < TN * n2g; p += NW) {
      const int s = p / n2g, rg = p - s * n2g;
      const int row = rg * 8 + (lane >> 3);
      const int slot_s = (lane & 7) ^ ((row >> 1) & 7);
      const char* const sp = g.W2 + (long)row * ((long)g.N * 4) + s * 128 + slot_s * 16;
      __builtin_amdgcn_global_load_lds((sp_gptr)sp, (sp_lptr)(w2 + s * (g.N2 * 128) + rg * 1024), 16, 0, 0);
    }
    const int slot = lane & 7, lrow = lane >> 3;
    const int mw0 = m0 + wm * 32;
    if (wave_active) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + slot * 4;
        const sp_f4 w1 = *reinterpret_cast<const sp_f4*>(g.wscale + n);
        const sp_f4 b1 = g.bias != nullptr ? *reinterpret_cast<const sp_f4*>(g.bias + n) : sp_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<sp_f4*>(stg1 + r * 128 + (((gq * 2 + h) ^ (r & 7)) << 4)) =
              sp_f4{acc[0][j][4 * gq], acc[0][j][4 * gq + 1], acc[0][j][4 * gq + 2], acc[0][j][4 * gq + 3]};
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + lrow;
          const sp_f4 raw = *reinterpret_cast<const sp_f4*>(stg1 + row * 128 + ((slot ^ lrow) << 4));
          sp_f4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = activate(__builtin_fmaf(raw[e], w1[e] * g.a_unmul, b1[e]));
          // this lane's 16-byte piece of the row's SP8 form (even quads: the chunk's hi halves, odd quads: its lo halves)
          *reinterpret_cast<sp_f4*>(a2 + j * CB + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) =
              __builtin_bit_cast(sp_f4, sp8_piece_from_quad(v, slot));
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // W2 has landed for every wave (each waited for its own pieces)
    if (!wave_active) return;
    const int tn2 = g.N2 >> 5;
    spf16 acc2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[j][q] = 0.f;
#pragma unroll
    for (int s = 0; s < TN; ++s) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ swr) << 4);
        const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ swr) << 4);
        const sp_h8 ah = *reinterpret_cast<const sp_h8*>(a2 + s * CB + r * 128 + shi);
        const sp_h8 al = *reinterpret_cast<const sp_h8*>(a2 + s * CB + r * 128 + slo);
        sp_h8 bh[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int rowb = j < tn2 ? j * 32 + r : r;  // (column blocks beyond N2 are computed on block 0's rows and dropped)
          bh[j] = *reinterpret_cast<const sp_h8*>(w2 + s * (g.N2 * 128) + rowb * 128 + shi);
          bl[j] = *reinterpret_cast<const sp_h8*>(w2 + s * (g.N2 * 128) + rowb * 128 + slo);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah, acc2[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc2[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah, acc2[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (j >= tn2) break;
      const int n = j * 32 + slot * 4;
      const sp_f4 w1 = *reinterpret_cast<const sp_f4*>(g.wscale2 + n);
      const sp_f4 b1 = g.bias2 != nullptr ? *reinterpret_cast<const sp_f4*>(g.bias2 + n) : sp_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<sp_f4*>(stg1 + r * 128 + (((gq * 2 + h) ^ (r & 7)) << 4)) =
            sp_f4{acc2[j][4 * gq], acc2[j][4 * gq + 1], acc2[j][4 * gq + 2], acc2[j][4 * gq + 3]};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = mw0 + it * 8 + lrow;
        const sp_f4 raw = *reinterpret_cast<const sp_f4*>(stg1 + (it * 8 + lrow) * 128 + ((slot ^ lrow) << 4));
        sp_f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = __builtin_fmaf(raw[e], w1[e], b1[e]);
          v[e] = g.act2 == ACT_SILU ? act_silu(t) : (g.act2 == ACT_NONE ? t : apply_act(t, g.act2));
        }
        sp_f4 piece = v;
        if (g.out_fmt2 == 1) piece = __builtin_bit_cast(sp_f4, sp8_piece_from_quad(v, slot));  // every lane takes part
        if (m < g.M) *reinterpret_cast<sp_f4*>(g.Out2 + (long)m * g.ldo2 + g.o_off2 + n) = piece;
      }
    }
    return;
  }
  if (!wave_active) return;

  if constexpr (EPI == 16) {
    // ---- fused top-k (match path): the scores never leave registers.  A lane holds 16 TN columns of its row, its
    // partner lane ^ 32 the other 16 TN; each round picks the (score desc, column asc) maximum of the wave's 32 TN
    // columns and retires it.  Candidates: cand[m][(tile_n * WN + wn) * topk + kk]; topk_merge_kernel finishes.
    const int ncol0 = n0 + wn * TN * 32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = ncol0 + j * 32 + gq * 8 + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool nok = n + e < g.N;
          const float ws = (g.wscale != nullptr && nok) ? g.wscale[n + e] * g.a_unmul : g.a_unmul;
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j][4 * gq + e] = nok ? acc[i][j][4 * gq + e] * ws : -INFINITY;
        }
      }
    const long slots = (long)g.tiles_n * WN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * TM * 32 + i * 32 + r;
      for (int kk = 0; kk < g.topk; ++kk) {
        float bs = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) {  // ascending column order inside the lane: the first maximum has the lowest column
            const float v = acc[i][j][q];
            const int n = ncol0 + j * 32 + (q >> 2) * 8 + 4 * h + (q & 3);
            if (v > bs || (v == bs && n < bi)) bs = v, bi = n;
          }
        const float os = __shfl_xor(bs, 32);
        const int oi = __shfl_xor(bi, 32);
        if (os > bs || (os == bs && oi < bi)) bs = os, bi = oi;
        if (h == 0 && m < g.M) {
          const long o = ((long)m * slots + (long)tile_n * WN + wn) * g.topk + kk;
          g.cand_s[o] = bs;
          g.cand_i[o] = (bs == -INFINITY) ? -1 : bi;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (ncol0 + j * 32 + (q >> 2) * 8 + 4 * h + (q & 3) == bi) acc[i][j][q] = -INFINITY;
      }
    }
    return;
  }

  // The accumulators of one 32-row slab go to LDS untouched (register group gq of column block j holds columns
  // 32 j + 8 gq + 4 h + 0..3 of row lane & 31) and are read back row-wise, 8 lanes per 128-byte row segment: a lane
  // keeps the same 4 columns of every column block for the whole tile, so its column scale, bias and GRN sums of
  // squares live in registers, every store instruction writes whole lines, and scale / bias / activation / residual /
  // SP8 packing all happen on the read-back side.  Residual rows are loaded one column block ahead.
  constexpr int SROW = 128 * TN;   // bytes per staged row (32*TN floats)
  constexpr int WREG = 32 * SROW;  // per wave
  // Eight-wave blocks (two blocks of them per CU = four waves per SIMD) stage their slabs in two rounds, waves 0..3 first:
  // the ring holds four slabs, not eight.
  constexpr int ER = (!AWIN && NW * WREG > NST * STG) ? 2 : 1;
  static_assert(AWIN || (NW / ER) * WREG <= NST * STG, "the store staging area must fit into the ring");  // AWIN: the host sizes LDS for it
  char* const stg = smem + (ER == 1 ? wave : wave % (NW / ER)) * WREG;
  if constexpr (ER == 2) {
    if (wave >= NW / 2) __builtin_amdgcn_s_barrier();  // released when the first round's waves have read their slabs back
  }
  constexpr int NIT = 4;              // 8 rows per read-back step

  // EPI >= 0 fixes the epilogue's shape at compile time (bit 0 SP8 output, 1 f32 residual, 2 SP8 residual, 3 GRN
  // partial sums; no output remap, N % 4 == 0) so that the read-back loop is one straight line of code the compiler can
  // interleave across steps; EPI < 0 reads all of it from the arguments.
  const bool out_sp8 = GEN ? g.out_fmt == 1 : (EPI & 1) != 0;
  const bool res_f32 = GEN ? (g.res != nullptr && g.res_fmt == 0) : (EPI & 2) != 0;
  const bool res_sp8 = GEN ? (g.res != nullptr && g.res_fmt == 1) : (EPI & 4) != 0;
  const bool has_res = res_f32 || res_sp8;
  const bool remap = GEN ? g.remap : false;
  const bool grn = GEN ? g.grn_part != nullptr : (EPI & 8) != 0;
  const int img_first = grn ? (int)fdiv((uint32_t)mw0, g.d_hw) : 0;
  const long unit = (long)tile_m * WM + wm;
  // per-lane element offsets of row mw0 + lrow, column block 0; every element this lane touches is a wave-uniform
  // number of rows and columns further on
  const int ncol0 = nw0 + slot * 4;
  const long o_lane = (long)(mw0 + lrow) * g.ldo + g.o_off + ncol0;
  const long r_lane = (long)(mw0 + lrow) * g.ldr + ncol0;
  const char* const stg_rd = stg + lrow * SROW + ((slot ^ lrow) << 4);  // + j * 128 + it * 8 * SROW

  // residual pieces of column block (i, j): 16 bytes per row step (f32 quad, or the SP8 hi and lo quads)
  auto load_res = [&](int i, int j, sp_f4 (&rr)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      rr[it] = sp_f4{0.f, 0.f, 0.f, 0.f};
      const long e = r_lane + (long)(i * 32 + it * 8) * g.ldr + j * 32;
      if (col_ok[j] && mw0 + i * 32 + it * 8 + lrow < g.M) {
        if (res_f32) {
          rr[it] = *reinterpret_cast<const sp_f4*>(reinterpret_cast<const float*>(g.res) + e);
        } else if (res_sp8) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          const char* const c = reinterpret_cast<const char*>(g.res) + (e - 4 * (slot & 1)) * 4 + 8 * (slot & 1);
          const f2 hi = *reinterpret_cast<const f2*>(c), lo = *reinterpret_cast<const f2*>(c + 16);
          rr[it] = sp_f4{hi[0], hi[1], lo[0], lo[1]};
        }
      }
    }
  };

  sp_f4 run[TN];  // sums of squares of this lane's columns over the rows of segment run_seg
#pragma unroll
  for (int j = 0; j < TN; ++j) run[j] = sp_f4{0.f, 0.f, 0.f, 0.f};
  int run_seg = 0;
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      sp_f4 t = run[j];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mask = 8; mask < 64; mask <<= 1) t[e] += __shfl_xor(t[e], mask);
      if (lane < 8 && col_ok[j])
        *reinterpret_cast<sp_f4*>(g.grn_part + (unit * g.segmax + run_seg) * g.N + ncol0 + j * 32) = t;
      run[j] = sp_f4{0.f, 0.f, 0.f, 0.f};
    }
  };

  sp_f4 rcur[NIT], rnext[NIT];
  if (has_res) load_res(0, 0, rcur);

  // One slab: FAST = every row and column of it exists and (GRN) all its rows belong to one image - no per-lane
  // predicates, one basic block the compiler interleaves across steps; otherwise the masked form.
  auto slab = [&](auto FAST_T, auto I_T) {
    constexpr bool FAST = decltype(FAST_T)::value;
    constexpr int i = decltype(I_T)::value;
    const int ms0 = mw0 + i * 32;
    int seg_lo = 0, seg_hi = 0;
    if (grn && ms0 < g.M) {
      const int m_last = ms0 + 31 < g.M ? ms0 + 31 : g.M - 1;
      seg_lo = (int)fdiv((uint32_t)ms0, g.d_hw) - img_first;
      seg_hi = (int)fdiv((uint32_t)m_last, g.d_hw) - img_first;
    }
    const bool single = FAST || seg_lo == seg_hi;
    if (grn && single && seg_lo != run_seg) {
      flush();
      run_seg = seg_lo;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      constexpr int NB = TM * TN;
      const int bnext = i * TN + j + 1;
      if (has_res && bnext < NB) load_res(bnext / TN, bnext % TN, rnext);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = ms0 + it * 8 + lrow;
        const bool ok = FAST || (col_ok[j] && m < g.M);
        const sp_f4 raw = *reinterpret_cast<const sp_f4*>(stg_rd + j * 128 + it * 8 * SROW);
        sp_f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)  // wsc is a power of two: the fused form rounds exactly like multiply-then-add
          v[e] = activate(__builtin_fmaf(raw[e], wsc[j][e], bsv[j][e]));
        if (grn) {
          if (single) {
            if (ok) {
#pragma unroll
              for (int e = 0; e < 4; ++e) run[j][e] = __builtin_fmaf(v[e], v[e], run[j][e]);
            }
          } else {  // the slab straddles images: leave the activated values in LDS for the per-image passes below
            *reinterpret_cast<sp_f4*>(stg + (it * 8 + lrow) * SROW + (((j * 8 + slot) ^ lrow) << 4)) = v;
          }
        }
        if (res_f32) {
          v = v + rcur[it];
        } else if (res_sp8) {
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          typedef float f2 __attribute__((ext_vector_type(2)));
          const h4 rh = __builtin_bit_cast(h4, f2{rcur[it][0], rcur[it][1]}), rl = __builtin_bit_cast(h4, f2{rcur[it][2], rcur[it][3]});
          v = v + (__builtin_convertvector(rh, sp_f4) + __builtin_convertvector(rl, sp_f4));
        }
        sp_f4 piece = v;
        if (out_sp8) piece = __builtin_bit_cast(sp_f4, sp8_piece_from_quad(v, slot));  // every lane takes part
        if (ok) {
          const long drow = i * 32 + it * 8;  // compile-time constant: drow * ld is scalar arithmetic
          if (remap) {
            const uint32_t img = fdiv((uint32_t)m, g.d_ohw);
            const uint32_t rem = (uint32_t)m - img * (uint32_t)(g.OH * g.OW);
            const uint32_t oh = fdiv(rem, g.d_ow);
            const uint32_t ow = rem - oh * (uint32_t)g.OW;
            uint32_t oy = (uint32_t)g.oy, ox = (uint32_t)g.ox, cn = (uint32_t)(ncol0 + j * 32);
            if (g.nq > 0) {  // whole ConvTranspose in one launch: this lane's column group names the output phase
              const uint32_t q = fdiv(cn, g.d_nq);
              cn -= q * (uint32_t)g.nq;
              oy = fdiv(q, g.d_os);
              ox = q - oy * (uint32_t)g.os;
            }
            const long orow = ((long)img * g.OH2 + oh * g.os + oy) * g.OW2 + ow * g.os + ox;
            *reinterpret_cast<sp_f4*>(g.Out + orow * g.ldo + g.o_off + cn) = piece;
          } else if (!GEN || ncol0 + j * 32 + 4 <= g.N) {
            *reinterpret_cast<sp_f4*>(g.Out + o_lane + drow * g.ldo + j * 32) = piece;
          } else {  // ragged last quad (N % 4 != 0; f32 output without residual only)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ncol0 + j * 32 + e < g.N) g.Out[o_lane + drow * g.ldo + j * 32 + e] = piece[e];
          }
        }
      }
      if (has_res && bnext < NB) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) rcur[it] = rnext[it];
      }
    }
    if (!FAST && grn && !single) {  // one masked pass over the staged (activated) slab per image
      for (int sgm = seg_lo; sgm <= seg_hi; ++sgm) {
        if (sgm != run_seg) {
          flush();
          run_seg = sgm;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int m = ms0 + it * 8 + lrow;
            if (col_ok[j] && m < g.M && (int)fdiv((uint32_t)m, g.d_hw) - img_first == sgm) {
              const sp_f4 v = *reinterpret_cast<const sp_f4*>(stg_rd + j * 128 + it * 8 * SROW);
#pragma unroll
              for (int e = 0; e < 4; ++e) run[j][e] = __builtin_fmaf(v[e], v[e], run[j][e]);
            }
          }
      }
    }
  };

  const bool cols_full = nw0 + 32 * TN <= g.N && !remap;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int sl = j * 8 + gq * 2 + h;
        *reinterpret_cast<sp_f4*>(stg + r * SROW + ((sl ^ (r & 7)) << 4)) =
            sp_f4{acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
      }
    // the slab is complete in LDS (same wave wrote it; LDS operations of one wave execute in order)
    const int ms0 = mw0 + i * 32;
    bool fast = !GEN && cols_full && ms0 + 32 <= g.M;  // the generic form keeps to the masked body (code size)
    if (fast && grn) fast = fdiv((uint32_t)ms0, g.d_hw) == fdiv((uint32_t)(ms0 + 31), g.d_hw);
    if constexpr (GEN) {
      if (i == 0) slab(std::false_type{}, std::integral_constant<int, 0>{});
      else slab(std::false_type{}, std::integral_constant<int, TM - 1>{});
    } else if (i == 0) {
      if (fast) slab(std::true_type{}, std::integral_constant<int, 0>{});
      else slab(std::false_type{}, std::integral_constant<int, 0>{});
    } else {
      if (fast) slab(std::true_type{}, std::integral_constant<int, TM - 1>{});
      else slab(std::false_type{}, std::integral_constant<int, TM - 1>{});
    }
  }
  if (grn) flush();
  if constexpr (ER == 2) {
    if (wave < NW / 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (g.stamps != nullptr && wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile's stores have been accepted
    const long st3 = (long)__builtin_amdgcn_s_memtime();
    if (lane == 0) {
      long* d = g.stamps + (long)blockIdx.x * 8;
      d[0] = st0, d[1] = st1, d[2] = st2, d[3] = st3;
      d[4] = (long)__builtin_amdgcn_s_getreg((15 << 11) | 4);   // HW_ID[15:0]: wave slot, SIMD, pipe, CU, SH, SE
      d[5] = (long)__builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID[3:0]
      d[6] = rt0, d[7] = (long)__builtin_amdgcn_s_memrealtime();  // (st3 - st0) / (d[7] - d[6]) x 100 MHz = the shader clock held
    }
  }
}

}  // namespace mtgv
