// Split-precision GEMM, second generation: operands arrive in LDS already split (SP8, sp8.h) and the K loop is
// pure LDS-DMA + ds_read_b128 + MFMA.
//
//   Out[orow(m)][o_off + n] = act( sum_k A(m,k) * W[n][k] * wscale[n] + bias[n] ) (+ res[m][n])
//
// Block: WM x WN waves; wave tile (32*TM) rows x (32*TN) columns; K in stages of 16*KS (RB = 64*KS bytes per staged
// row), two LDS buffers, one raw s_barrier per stage.
//   B (weights)          always by LDS-DMA (global_load_lds_dwordx4) from the registered SP8 copy.
//   A, AMODE 0 (DMA)     SP8 activation rows written by the producer kernel: LDS-DMA.  Dense rows [M][lda].
//   A, AMODE 2 (CONV)    SP8 NHWC activations gathered by the DMA's per-lane source address (implicit GEMM, no
//                        im2col): m -> (img, oh, ow), k -> (kh, kw, c); taps outside the image and the K tail read a
//                        zero page.  Cin % 8 == 0 keeps every 8-channel chunk inside one tap.
//   A, AMODE 1 (REG)     f32 rows: global -> VGPR (issued before the stage's MFMAs) -> optional per-image multiplier
//                        (GRN apply, convnextv2.py:171-174) -> split -> ds_write_b128 (after the MFMAs).
// LDS image of a stage: [row][SPR slots of 16 B], slot' = slot ^ sw(row), sw = (row>>1)&7 for 128-byte rows,
// (row>>2)&3 for 64-byte rows: every ds_read_b128 lane group covers all 64 banks once.  The DMA destination is
// lane-linear, so the swizzle is applied to the per-lane source address (and to the ds_write address in REG mode).
//
// MFMA orientation: weights are the first operand, so the accumulator has n on registers and m on lanes - a lane owns
// one output row and 4 consecutive columns per register group.  The epilogue applies wscale, bias and the activation
// in registers and stages each 32-row slab through LDS, so that global stores and residual loads are whole 128-byte
// lines; in the read-back a lane keeps the same 4 columns, which makes the GRN sum(x^2) partials lane-local.
// SP8 output (out_fmt 1): the two lanes that own the halves of an 8-column chunk trade halves (v_permlane32_swap), so
// one holds the chunk's hi piece and the other its lo piece - the same 16-byte slots the f32 path stages.
//
// Products: lo*hi + hi*lo + hi*hi per k16 step into the same accumulator, k ascending: results do not depend on the
// tile configuration.
#pragma once
#include "act.h"
#include "gemm_f32.h"
#include "sp8.h"

namespace mtgv {

typedef float spf16 __attribute__((ext_vector_type(16)));

struct SpDev {
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
  int tiles_m = 0, tiles_n = 0;
  int act = 0;
  // AMODE 2 geometry
  int H = 1, Wd = 1, Cin = 0, KW = 1, stride = 1, pad = 0, OH = 1, OW = 1;
  FastDiv d_ohw, d_ow, d_cin, d_kw;
  // output row remap (ConvTranspose scatter): orow = (img*OH2 + oh*os + oy)*OW2 + ow*os + ox
  int remap = 0, os = 1, oy = 0, ox = 0, OH2 = 1, OW2 = 1;
};

typedef const __attribute__((address_space(1))) void* sp_gptr;
typedef __attribute__((address_space(3))) void* sp_lptr;

template <int WM, int WN, int TM, int TN, int KS, int AMODE, int ACT>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_sp_kernel(const SpDev g) {
#pragma clang fp contract(off)
  constexpr int NW = WM * WN, NT = 64 * NW, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int RB = 64 * KS, RPP = 1024 / RB, SPR = RB / 16;
  constexpr int SA = BM * RB, SB = BN * RB, STG = SA + SB;
  constexpr bool ADMA = AMODE != 1;
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

  int L;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, x = b & 7;
    L = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (b >> 3);
  }
  const int tile_n = L % g.tiles_n, tile_m = L / g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long wrowb = (long)g.K * 4;
  const int kchunks = g.K >> 3;                 // valid 8-float chunks per row
  const int nk = (g.K + 16 * KS - 1) / (16 * KS);
  const bool ktail = (g.K % (16 * KS)) != 0;

  // ---- DMA pieces: piece p = wave + NW*u of the stage image (A pieces first, then B) ----
  const char* src[PPW];   // dense A / B: address of the lane's slot in stage 0.  CONV A: pixel (img, 0, 0) of the row
  bool tailz[PPW];
  int cslot[PPW], cih0[PPW], ciw0[PPW];  // CONV A pieces: logical slot, first input row / column of the window
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    const int p = wave + NW * u;
    const bool isA = p < PA;
    const int pp = isA ? p : p - PA;
    const int row = pp * RPP + lane / SPR;
    const int sw = KS == 2 ? (row >> 1) & 7 : (row >> 2) & 3;
    const int slot = (lane % SPR) ^ sw;
    cslot[u] = slot, cih0[u] = 0, ciw0[u] = 0;
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
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int p = wave + NW * u;
      if (NP % NW == 0 || p < NP) {
        const char* s;
        if (AMODE == 2 && p < PA) {
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
        __builtin_amdgcn_global_load_lds((sp_gptr)s, (sp_lptr)(smem + buf * STG + (SA - PA * 1024) + p * 1024), 16, 0, 0);
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
        char* const d = smem + buf * STG;
        *reinterpret_cast<sp_h8*>(d + a_lds[v]) = hi;
        *reinterpret_cast<sp_h8*>(d + (a_lds[v] ^ 16u)) = lo;
      }
    }
  };

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

  // a wave whose rows all lie beyond M (ragged last tile, tiny-M problems) skips its MFMAs
  const bool wave_active = m0 + wm * TM * 32 < g.M;

  issue(0, 0);
  loadA(0);
  storeA(0);
  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    // stage t landed: this wave's DMA pieces (vmcnt) and REG-mode ds_writes (lgkmcnt), then everyone's (barrier).
    // The barrier also says every wave has finished reading the other buffer, which is refilled next.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nk) {
      issue(t + 1, buf ^ 1);
      loadA(t + 1);
    }
    if (wave_active) {
      const char* const sb = smem + buf * STG;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ swr) << 4);
        const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ swr) << 4);
        sp_h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + shi);
          al[i] = *reinterpret_cast<const sp_h8*>(sb + a_off[i] + slo);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + shi);
          bl[j] = *reinterpret_cast<const sp_h8*>(sb + b_off[j] + slo);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i) {  // small cross terms first, the hi*hi product last
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          }
      }
    }
    if (t + 1 < nk) storeA(buf ^ 1);
    buf ^= 1;
  }

  // ---- epilogue ----
  auto activate = [&](float x) -> float {
    if constexpr (ACT == ACT_NONE) return x;
    else if constexpr (ACT == ACT_GELU) return act_gelu(x);
    else if constexpr (ACT == ACT_MISH) return act_mish(x);
    else if constexpr (ACT == ACT_SILU) return act_silu(x);
    else return apply_act(x, g.act);
  };
  __builtin_amdgcn_s_barrier();  // every wave is done with the ring: it becomes the store staging area
  if (!wave_active) return;

  constexpr int SROW = 128 * TN;   // bytes per staged row (32*TN floats)
  constexpr int PPR = 8 * TN;      // 16-byte pieces per staged row
  constexpr int WREG = 34 * SROW;  // per wave: 32 staged rows + one row of column scales + one row of biases
  static_assert(NW * WREG <= 2 * STG, "the store staging area must fit into the ring");
  char* const stg = smem + wave * WREG;
  const int nw0 = n0 + wn * TN * 32;  // first column of this wave

  // per-column constants through LDS (register group gq of column block j holds n = nw0 + 32 j + 8 gq + 4 h + 0..3;
  // keeping all of them in registers beside the accumulators spills on the wide tiles)
  float* const cst = reinterpret_cast<float*>(stg + 32 * SROW);
  if (lane < PPR) {
    const int n = nw0 + lane * 4;
    sp_f4 w1 = {1.f, 1.f, 1.f, 1.f}, b0 = {0.f, 0.f, 0.f, 0.f};
    if (n + 4 <= g.N) {
      if (g.wscale != nullptr) w1 = *reinterpret_cast<const sp_f4*>(g.wscale + n);
      if (g.bias != nullptr) b0 = *reinterpret_cast<const sp_f4*>(g.bias + n);
    } else {  // ragged last quad (N % 4 != 0)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < g.N) {
          if (g.wscale != nullptr) w1[e] = g.wscale[n + e];
          if (g.bias != nullptr) b0[e] = g.bias[n + e];
        }
    }
    *reinterpret_cast<sp_f4*>(cst + lane * 4) = w1 * g.a_unmul;
    *reinterpret_cast<sp_f4*>(cst + 32 * TN + lane * 4) = b0;
  }

  const int mw0 = m0 + wm * TM * 32;  // first row of this wave

  // read-back geometry: PPS lanes walk one staged row (PPS = PPR rounded up to a power of two), so a lane keeps the
  // same 4 columns for the whole tile - its column sums of squares (GRN) need no cross-lane work until the flush
  constexpr int PPS = PPR <= 8 ? 8 : (PPR <= 16 ? 16 : 32);
  constexpr int RPI = 64 / PPS, NIT = 32 / RPI;
  const int slot = lane % PPS, lrow = lane / PPS;
  const int ncol = nw0 + slot * 4;
  const bool col_ok = slot < PPR && ncol < g.N;

  const bool grn = g.grn_part != nullptr;
  const int img_first = grn ? (int)fdiv((uint32_t)mw0, g.d_hw) : 0;
  const long unit = (long)tile_m * WM + wm;
  // per-lane element offsets of row mw0 + lrow; every row this lane stores is a wave-uniform number of rows further on
  const long o_lane = (long)(mw0 + lrow) * g.ldo + g.o_off + ncol;
  const long r_lane = (long)(mw0 + lrow) * g.ldr + ncol;
  const bool res_regs = g.res != nullptr && g.res_fmt == 1;  // SP8 residual: added in registers, before the split
  const bool res_rows = g.res != nullptr && g.res_fmt == 0;  // f32 residual: added to the staged rows
  sp_f4 run = {0.f, 0.f, 0.f, 0.f};  // sum of squares of this lane's columns over the rows of segment run_seg
  int run_seg = 0;
  auto flush = [&]() {
    sp_f4 t = run;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mask = PPS; mask < 64; mask <<= 1) t[e] += __shfl_xor(t[e], mask);
    if (lane < PPS && col_ok) *reinterpret_cast<sp_f4*>(g.grn_part + (unit * g.segmax + run_seg) * g.N + ncol) = t;
    run = sp_f4{0.f, 0.f, 0.f, 0.f};
  };

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = mw0 + i * 32 + r;  // this lane's row in the register stage
    const char* const res_row =
        res_regs ? reinterpret_cast<const char*>(g.res) + ((long)(mrow < g.M ? mrow : g.M - 1) * g.ldr) * 4 : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const sp_f4 wsc = *reinterpret_cast<const sp_f4*>(cst + j * 32 + gq * 8 + 4 * h);
        const sp_f4 bsv = *reinterpret_cast<const sp_f4*>(cst + 32 * TN + j * 32 + gq * 8 + 4 * h);
        sp_f4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)  // wsc is a power of two: the fused form rounds exactly like multiply-then-add
          v[e] = activate(__builtin_fmaf(acc[i][j][4 * gq + e], wsc[e], bsv[e]));
        if (res_regs) {
          // this lane's 4 columns of the residual chunk: hi halves at +8h, lo halves at +16+8h of the 32-byte chunk
          const int nq = nw0 + j * 32 + gq * 8;
          if (nq < g.N) {
            const char* const c = res_row + (long)nq * 4;
            const sp_h4 rh = *reinterpret_cast<const sp_h4*>(c + 8 * h), rl = *reinterpret_cast<const sp_h4*>(c + 16 + 8 * h);
            v = v + (__builtin_convertvector(rh, sp_f4) + __builtin_convertvector(rl, sp_f4));
          }
        }
        sp_f4 piece = v;
        if (g.out_fmt == 1) {
          sp_h4 hi, lo;
          sp8_split4(v, hi, lo);
          typedef unsigned u2 __attribute__((ext_vector_type(2)));
          const u2 a = __builtin_bit_cast(u2, hi), b = __builtin_bit_cast(u2, lo);
          // v_permlane32_swap: lanes 32-63 of the first operand trade places with lanes 0-31 of the second
          const auto s0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
          // h = 0: {own hi, partner's hi} = the chunk's hi piece; h = 1: {partner's lo, own lo} = its lo piece
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          piece = __builtin_bit_cast(sp_f4, u4{s0[0], s1[0], s0[1], s1[1]});
        }
        const int sl = j * 8 + gq * 2 + h;
        *reinterpret_cast<sp_f4*>(stg + r * SROW + ((sl ^ (r & 7)) << 4)) = piece;
        __builtin_amdgcn_sched_barrier(0);  // one register group at a time: interleaving all of them spills on wide tiles
      }
    // the slab is complete in LDS (same wave wrote it; LDS operations of one wave execute in order)
    const int ms0 = mw0 + i * 32;
    int seg_lo = 0, seg_hi = 0;
    if (grn && ms0 < g.M) {
      const int m_last = ms0 + 31 < g.M ? ms0 + 31 : g.M - 1;
      seg_lo = (int)fdiv((uint32_t)ms0, g.d_hw) - img_first;
      seg_hi = (int)fdiv((uint32_t)m_last, g.d_hw) - img_first;
    }
    const bool single = seg_lo == seg_hi;
    if (grn && single && seg_lo != run_seg) {
      flush();
      run_seg = seg_lo;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + lrow;
      const int m = ms0 + row;
      if (col_ok && m < g.M) {
        sp_f4 v = *reinterpret_cast<const sp_f4*>(stg + row * SROW + ((slot ^ (row & 7)) << 4));
        if (grn && single) {
#pragma unroll
          for (int e = 0; e < 4; ++e) run[e] = __builtin_fmaf(v[e], v[e], run[e]);
        }
        const long drow = i * 32 + it * RPI;  // compile-time constant: drow * ld is scalar arithmetic
        if (res_rows) v = v + *reinterpret_cast<const sp_f4*>(reinterpret_cast<const float*>(g.res) + r_lane + drow * g.ldr);
        if (g.remap) {
          const uint32_t img = fdiv((uint32_t)m, g.d_ohw);
          const uint32_t rem = (uint32_t)m - img * (uint32_t)(g.OH * g.OW);
          const uint32_t oh = fdiv(rem, g.d_ow);
          const uint32_t ow = rem - oh * (uint32_t)g.OW;
          const long orow = ((long)img * g.OH2 + oh * g.os + g.oy) * g.OW2 + ow * g.os + g.ox;
          *reinterpret_cast<sp_f4*>(g.Out + orow * g.ldo + g.o_off + ncol) = v;
        } else if (ncol + 4 <= g.N) {
          *reinterpret_cast<sp_f4*>(g.Out + o_lane + drow * g.ldo) = v;
        } else {  // ragged last quad (N % 4 != 0; f32 output without residual only)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ncol + e < g.N) g.Out[o_lane + drow * g.ldo + e] = v[e];
        }
      }
    }
    if (grn && !single) {  // the slab straddles images: one masked pass over the staged slab per image
      for (int sgm = seg_lo; sgm <= seg_hi; ++sgm) {
        if (sgm != run_seg) {
          flush();
          run_seg = sgm;
        }
        for (int it = 0; it < NIT; ++it) {
          const int row = it * RPI + lrow;
          const int m = ms0 + row;
          if (col_ok && m < g.M && (int)fdiv((uint32_t)m, g.d_hw) - img_first == sgm) {
            const sp_f4 v = *reinterpret_cast<const sp_f4*>(stg + row * SROW + ((slot ^ (row & 7)) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) run[e] = __builtin_fmaf(v[e], v[e], run[e]);
          }
        }
      }
    }
  }
  if (grn) flush();
}

}  // namespace mtgv
