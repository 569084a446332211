// The implicit-GEMM kernel template and its tile dispatcher.  Included by exactly two translation units,
// one per operand precision, so that the two sets of instantiations compile in parallel:
//   gemm_f32.hip    PREC 0: f32 operands on v_mfma_f32_32x32x2_f32 (exact products, 157 TFLOP/s ceiling)
//   gemm_f16x3.hip  PREC 1: every f32 operand split on the fly into fp16 hi + lo, three
//                           v_mfma_f32_32x32x16_f16 per product (hi*hi + hi*lo + lo*hi, f32 accumulate)
//
// Tile: 256 threads = 4 waves stacked along M.  Block tile BM = 128*TM rows by
// BN = 32*TN columns; wave w owns rows [w*32*TM, (w+1)*32*TM) x all BN columns
// as TM x TN accumulators of 32x32.  K is consumed in BK-wide steps staged
// through LDS, with the next step's global loads in flight while the current one feeds the MFMAs.
#pragma once
#include "gemm_f32.h"
#include "act.h"

#include <type_traits>

namespace mtgv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct GemmDev {
  GemmArgs a;
  FastDiv d_ohw, d_ow, d_cin, d_kwcin, d_hw, d_cw;
  int tiles_m, tiles_n;
  int remap;     // output rows are not simply m
  int nseg_max;  // APRO: images a 128-row tile can touch (sizes the LDS multiplier tile)
  int off32_ok;  // every operand of the launch spans less than 4 GiB: 32-bit byte offsets are enough
};

// x = hi + lo + O(2^-22 |x|) while lo is a normal fp16 (|x| >= 2^-3); below that lo is subnormal and the error floor is
// 2^-25 ABSOLUTE (fp16 subnormals are kept by the matrix unit).  Registered weights therefore carry a power-of-two scale
// per row (wscale, undone on the f32 accumulator) that puts the row maximum at 2^13..2^14; activations are O(1) per row
// (LayerNorm outputs, activations, pixels), and a launch whose A may exceed the fp16 range passes a_mul (sp8.h).
// A truncated hi (mask off 13 mantissa bits, no conversion back to f32) is 1 % faster and as accurate on one GEMM,
// but its remainder always has the sign of x, so the dropped lo*lo term becomes a systematic bias that adds up over
// the detector's layers (proto error vs the oracle 1.0e-4 instead of 1.7e-5): rounding it is.
// |x| beyond the fp16 range (65504) becomes inf: visible, not silently wrong - such data belongs on PREC 0.
__device__ __forceinline__ void split_f16(const f32x4 x, f16x4& hi, f16x4& lo) {
  hi = __builtin_convertvector(x, f16x4);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), f16x4);
}

// LDS bytes of one launch.  PREC 0: f32 rows padded by 16 B (conflict-free ds_read_b128) + the APRO multiplier
// tile.  PREC 1: two fp16 planes (hi, lo) per operand; 16-half rows are contiguous, 32-half rows padded by 16 B.
template <int TM, int TN, int BK, bool APRO, int PREC>
constexpr size_t gemm_lds_bytes(int nseg_max) {
  if (PREC == 1) return (size_t)2 * 2 * (128 * TM + 32 * TN) * (BK == 16 ? 16 : BK + 8) * sizeof(_Float16);
  return (size_t)2 * (128 * TM + 32 * TN) * (BK + 4) * sizeof(float) + (APRO ? (size_t)2 * nseg_max * BK * sizeof(float) : 0);
}

template <int TM, int TN, int BK, bool CONV, bool APRO, int EPI, int ACT, int PREC>
__global__ __launch_bounds__(256, (TM * TN == 1 ? (BK == 16 ? 6 : 4) : TM * TN == 2 ? (BK == 16 ? 5 : 4) : TM * TN == 3 ? (BK == 16 ? 4 : 3) : TM * TN == 4 ? (BK == 16 ? 3 : 2) : 2)) void gemm_f32_kernel(const GemmDev g) {
#pragma clang fp contract(off)  // loader scaling and epilogue arithmetic identical in every instantiation (act.h)
  constexpr int BM = 128 * TM, BN = 32 * TN, LS = BK + 4;
  constexpr int LSH = BK == 16 ? 16 : BK + 8;  // PREC 1: halves per staged row
  constexpr int KQ = BK / 4;      // float4 per staged row
  constexpr int RPP = 256 / KQ;   // rows staged per pass
  constexpr int AP = BM / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  static_assert(BM % RPP == 0, "A tile must be a whole number of passes");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                // PREC 0: [2][BM][LS]
  float* Bs = smem + 2 * BM * LS;  // PREC 0: [2][BN][LS]
  _Float16* const Ah = reinterpret_cast<_Float16*>(smem);  // PREC 1: [buf 2][plane 2][BM][LSH]
  _Float16* const Bh = Ah + 2 * 2 * BM * LSH;              // PREC 1: [buf 2][plane 2][BN][LSH]

  const GemmArgs& p = g.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.y;
  const float* const Ap = p.A + (long)z * p.strideA;
  const float* const Wp = p.W + (long)z * p.strideW;
  // PREC 1: B already split into fp16 hi/lo halves in memory (registered weights / bank): no conversion work for it
  const bool b_presplit = PREC == 1 && p.W_split != nullptr;
  const float* const Wld = b_presplit ? p.W_split : Wp;
  float* const Op = p.Out + (long)z * p.strideO;
  int M_eff = p.M;
  if (p.m_count != nullptr) {
    const int mc = p.m_count[z];
    M_eff = mc < p.M ? mc : p.M;
  }

  // XCD-aware tile order: blocks that share an XCD (equal blockIdx % 8) walk a
  // contiguous run of tiles, n fastest, so an A row-panel is fetched once per L2.
  int L;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  const int tile_n = L % g.tiles_n, tile_m = L / g.tiles_n;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  if (bm0 >= M_eff) return;  // whole tile beyond this batch's rows (uniform per block)

  // ---- loader state: each thread stages fixed rows, one float4 column ----
  const int lrow = tid / KQ, lk = (tid % KQ) * 4;
  long a_row[AP];   // CONV: pixel index of image start; dense: element offset of row
  int a_ih0[AP], a_iw0[AP];
  bool a_ok[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = bm0 + lrow + i * RPP;
    a_ok[i] = m < M_eff;
    const uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
    if (CONV) {
      const uint32_t img = fdiv(mm, g.d_ohw);
      const uint32_t rem = mm - img * (uint32_t)(p.OH * p.OW);
      const uint32_t oh = fdiv(rem, g.d_ow);
      const uint32_t ow = rem - oh * (uint32_t)p.OW;
      a_row[i] = (long)img * p.H * p.Wd;
      a_ih0[i] = (int)oh * p.stride - p.pad;
      a_iw0[i] = (int)ow * (p.stride_w > 0 ? p.stride_w : p.stride) - p.pad;
    } else {
      a_row[i] = (long)mm * p.c_total + p.c_off;
      a_ih0[i] = a_iw0[i] = 0;
    }
  }
  long b_row[BP];
  bool b_ok[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int rr = lrow + i * RPP;
    const int n = bn0 + rr;
    b_ok[i] = (rr < BN) && (n < p.N);
    b_row[i] = (long)(b_ok[i] ? n : 0) * p.K;
  }

  f32x4 ra[AP], rb[BP];
  // APRO (GRN apply), PREC 0: the multipliers s[img][k] of the images this tile touches are staged per K step
  // into LDS (Ss[2][nseg][BK]) and applied to the A fragments as they are read - the A loads stay plain.
  // PREC 1 has to scale before the fp16 split: each staged float4 of A loads its own multipliers (L2-resident
  // table) and is multiplied on its way into LDS.
  constexpr int SPT = (APRO && PREC == 0) ? (BM * KQ + 255) / 256 : 1;  // multiplier float4s a thread may have to stage
  f32x4 rsl[SPT];
  f32x4 rsa[(APRO && PREC == 1) ? AP : 1];
  long a_srow[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    a_srow[i] = 0;
    if (APRO && PREC == 1) {
      const int m = bm0 + lrow + i * RPP;
      a_srow[i] = (long)fdiv((uint32_t)(m < M_eff ? m : M_eff - 1), g.d_hw) * p.K;
    }
  }
  float* Ss = Bs + 2 * BN * LS;
  const int img_first_t = APRO ? (int)fdiv((uint32_t)bm0, g.d_hw) : 0;
  int nseg_t = 1;
  if (APRO) {
    const int m_end_t = (bm0 + BM < M_eff) ? bm0 + BM : M_eff;
    nseg_t = (int)fdiv((uint32_t)(m_end_t - 1), g.d_hw) - img_first_t + 1;
  }
  // FAST (chosen per block, below): a dense tile that lies completely inside the problem with K a multiple of BK needs
  // no predicates and no zero fill, and every address is a block-uniform base (advanced by the scalar unit) plus a
  // 32-bit per-thread byte offset: 65 instead of 130 VALU instructions per K step of a 128x96 f16x3 tile (9 MFMAs).
  // Measured effect on the whole step: -1 % GEMM time - the loop is not bound by VALU issue alone.
  unsigned a_off[AP], b_off[BP], s_off[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    a_off[i] = (unsigned)((a_row[i] + lk) * (long)sizeof(float));
    s_off[i] = (unsigned)((a_srow[i] + lk) * (long)sizeof(float));
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) b_off[i] = (unsigned)((b_row[i] + lk) * (long)sizeof(float));
  auto load_tile = [&](int kt, auto fast_c) {
    if constexpr (decltype(fast_c)::value) {
      const char* const Ak = reinterpret_cast<const char*>(Ap) + (size_t)kt * BK * sizeof(float);
      const char* const Wk = reinterpret_cast<const char*>(Wld) + (size_t)kt * BK * sizeof(float);
#pragma unroll
      for (int i = 0; i < AP; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ak + a_off[i]);
      if (APRO && PREC == 1) {
        const char* const Sk = reinterpret_cast<const char*>(p.a_scale) + (size_t)kt * BK * sizeof(float);
#pragma unroll
        for (int i = 0; i < AP; ++i) rsa[i] = *reinterpret_cast<const f32x4*>(Sk + s_off[i]);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Wk + b_off[i]);  // rows past BN: row 0, never stored
      if (APRO && PREC == 0) {
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
          const int e = tid + u * 256;
          const int seg = e / KQ, kq = (e % KQ) * 4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (seg < nseg_t) v = *reinterpret_cast<const f32x4*>(p.a_scale + (long)(img_first_t + seg) * p.K + kt * BK + kq);
          rsl[u] = v;
        }
      }
      return;
    }
    const int k = kt * BK + lk;
    const bool kok = k < p.K;
    int kh = 0, kw = 0, c = 0;
    if (CONV) {
      const uint32_t kk = kok ? (uint32_t)k : 0u;
      kh = (int)fdiv(kk, g.d_kwcin);
      const uint32_t r = kk - (uint32_t)kh * (uint32_t)(p.KW * p.Cin);
      kw = (int)fdiv(r, g.d_cin);
      c = (int)(r - (uint32_t)kw * (uint32_t)p.Cin);
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (CONV) {
        const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
        if (a_ok[i] && kok && ih >= 0 && ih < p.H && iw >= 0 && iw < p.Wd)
          v = *reinterpret_cast<const f32x4*>(Ap + (a_row[i] + (long)ih * p.Wd + iw) * p.c_total + p.c_off + c);
      } else {
        if (a_ok[i] && kok) v = *reinterpret_cast<const f32x4*>(Ap + a_row[i] + k);
      }
      ra[i] = v;
      if (APRO && PREC == 1) {
        f32x4 sv = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i] && kok) sv = *reinterpret_cast<const f32x4*>(p.a_scale + a_srow[i] + k);
        rsa[i] = sv;
      }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok[i] && kok) v = *reinterpret_cast<const f32x4*>(Wld + b_row[i] + k);
      rb[i] = v;
    }
    if (APRO && PREC == 0) {
#pragma unroll
      for (int u = 0; u < SPT; ++u) {
        const int e = tid + u * 256;  // (segment, float4 column) of the multiplier tile
        const int seg = e / KQ, kq = (e % KQ) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (seg < nseg_t && kt * BK + kq < p.K)
          v = *reinterpret_cast<const f32x4*>(p.a_scale + (long)(img_first_t + seg) * p.K + kt * BK + kq);
        rsl[u] = v;
      }
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (PREC == 1) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        f16x4 hi, lo;
        split_f16((APRO ? ra[i] * rsa[i] : ra[i]) * p.a_mul, hi, lo);
        const int o = (lrow + i * RPP) * LSH + lk;
        *reinterpret_cast<f16x4*>(&Ah[(buf * 2 + 0) * BM * LSH + o]) = hi;
        *reinterpret_cast<f16x4*>(&Ah[(buf * 2 + 1) * BM * LSH + o]) = lo;
      }
#pragma unroll
      for (int i = 0; i < BP; ++i)
        if (lrow + i * RPP < BN) {
          f16x4 hi, lo;
          if (b_presplit) {
            const f16x8 hl = __builtin_bit_cast(f16x8, rb[i]);
            hi = f16x4{hl[0], hl[1], hl[2], hl[3]}, lo = f16x4{hl[4], hl[5], hl[6], hl[7]};
          } else {
            split_f16(rb[i], hi, lo);
          }
          const int o = (lrow + i * RPP) * LSH + lk;
          *reinterpret_cast<f16x4*>(&Bh[(buf * 2 + 0) * BN * LSH + o]) = hi;
          *reinterpret_cast<f16x4*>(&Bh[(buf * 2 + 1) * BN * LSH + o]) = lo;
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const f32x4 v = ra[i];
      *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + i * RPP) * LS + lk]) = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i)
      if (lrow + i * RPP < BN) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + i * RPP) * LS + lk]) = rb[i];
    if (APRO) {
#pragma unroll
      for (int u = 0; u < SPT; ++u) {
        const int e = tid + u * 256;
        if (e / KQ < nseg_t) *reinterpret_cast<f32x4*>(&Ss[(buf * g.nseg_max + e / KQ) * BK + (e % KQ) * 4]) = rsl[u];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int col = lane & 31, half = lane >> 5;
  const int arow = wave * 32 * TM + col;
  const int kh4 = 4 * half;

  // a wave whose rows all lie beyond M (ragged last tile, tiny-M problems) skips its MFMAs
  const bool wave_active = bm0 + wave * 32 * TM < M_eff;
  int seg_lane[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int sg = 0;
    if (APRO) {
      const int m = bm0 + arow + i * 32;
      sg = (int)fdiv((uint32_t)(m < M_eff ? m : M_eff - 1), g.d_hw) - img_first_t;
    }
    seg_lane[i] = sg;
  }
  const int nk = (p.K + BK - 1) / BK;
  auto k_loop = [&](auto fast_c) {
    load_tile(0, fast_c);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load_tile(kt + 1, fast_c);
      const int cur = kt & 1;
      const float* Ab = As + cur * BM * LS;
      const float* Bb = Bs + cur * BN * LS;
      if constexpr (PREC == 1) {
        if (wave_active) {
          const _Float16* const Ahi = Ah + (cur * 2 + 0) * BM * LSH;
          const _Float16* const Alo = Ah + (cur * 2 + 1) * BM * LSH;
          const _Float16* const Bhi = Bh + (cur * 2 + 0) * BN * LSH;
          const _Float16* const Blo = Bh + (cur * 2 + 1) * BN * LSH;
#pragma unroll
          for (int ks = 0; ks < BK / 16; ++ks) {
            // lane half h holds k = ks*16 + 8h + j in element j of both operands' fragments
            const int ko = ks * 16 + 8 * half;
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              ah[i] = *reinterpret_cast<const f16x8*>(&Ahi[(arow + i * 32) * LSH + ko]);
              al[i] = *reinterpret_cast<const f16x8*>(&Alo[(arow + i * 32) * LSH + ko]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const f16x8 bh = *reinterpret_cast<const f16x8*>(&Bhi[(j * 32 + col) * LSH + ko]);
              const f16x8 bl = *reinterpret_cast<const f16x8*>(&Blo[(j * 32 + col) * LSH + ko]);
#pragma unroll
              for (int i = 0; i < TM; ++i) {  // small cross terms first, the hi*hi product last
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
              }
            }
          }
        }
      } else if (wave_active) {
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        // lane half h holds k = kk*8 + 4h + j in element j; A and B use the same
        // k assignment, so MFMA j multiplies matching k pairs {j, 4+j}.
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          a[i] = *reinterpret_cast<const f32x4*>(&Ab[(arow + i * 32) * LS + kk * 8 + kh4]);
          if (APRO) a[i] = a[i] * *reinterpret_cast<const f32x4*>(&Ss[(cur * g.nseg_max + seg_lane[i]) * BK + kk * 8 + kh4]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bb[(j * 32 + col) * LS + kk * 8 + kh4]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      }
      }
      if (kt + 1 < nk) store_tile(cur ^ 1);
      __syncthreads();
    }
  };
  if constexpr (!CONV) {
    const bool fast_tile = bm0 + BM <= M_eff && bn0 + BN <= p.N && p.K % BK == 0 && g.off32_ok;
    if (fast_tile)
      k_loop(std::true_type{});
    else
      k_loop(std::false_type{});
  } else {
    k_loop(std::false_type{});
  }

  if constexpr (EPI == 1) {
    // ---- fused per-tile top-k (match path): the tile's scores never leave registers.
    // A row's BN scores sit in the 32 lanes of one wave half x TN registers; each
    // round picks the (score desc, id asc) maximum with a 5-step butterfly and
    // retires it.  Candidates go to cand[m][tile_n][kk]; a merge kernel finishes.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool nok = (bn0 + j * 32 + col) < p.N;
      const float ws = (PREC == 1 && p.wscale != nullptr && nok) ? p.wscale[bn0 + j * 32 + col] * p.a_unmul : p.a_unmul;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = nok ? acc[i][j][r] * ws : -INFINITY;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = bm0 + wave * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        for (int kk = 0; kk < p.topk; ++kk) {
          float bs = -INFINITY;
          int bi = 0x7fffffff;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float v = acc[i][j][r];
            if (v > bs) bs = v, bi = bn0 + j * 32 + col;
          }
#pragma unroll
          for (int mask = 16; mask > 0; mask >>= 1) {
            const float os = __shfl_xor(bs, mask);
            const int oi = __shfl_xor(bi, mask);
            if (os > bs || (os == bs && oi < bi)) bs = os, bi = oi;
          }
          if (col == 0 && m < M_eff) {
            const long o = ((long)m * g.tiles_n + tile_n) * p.topk + kk;
            p.cand_s[o] = bs;
            p.cand_i[o] = (bs == -INFINITY) ? -1 : bi;
          }
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (bn0 + j * 32 + col == bi) acc[i][j][r] = -INFINITY;
        }
      }
    }
    return;
  }

  // ---- epilogue: bias, activation, residual, store (C layout: col = lane&31,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  ACT >= 0 fixes the activation at compile time. ----
  auto activate = [&](float x) -> float {
    if constexpr (ACT == ACT_NONE) return x;
    else if constexpr (ACT == ACT_GELU) return act_gelu(x);
    else if constexpr (ACT == ACT_MISH) return act_mish(x);
    else if constexpr (ACT == ACT_SILU) return act_silu(x);
    else return apply_act(x, p.act);
  };
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool interior = (bm0 + BM <= M_eff) && (bn0 + BN <= p.N) && !g.remap && p.crop_boxes == nullptr;
  if (interior) {
    // Whole tile inside the problem, rows map 1:1: no per-element predicates, and every address is
    // (wave-uniform base) + (32-bit lane offset), so stores/loads need no per-element address VALU.
    const long row0 = (long)bm0 + (long)wave_u * 32 * TM;
    float* const obase = Op + row0 * p.ldo + p.o_off + bn0;
    const unsigned loff = (unsigned)(4 * half) * (unsigned)p.ldo + (unsigned)col;
    float bv[TN], wsv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bv[j] = p.bias != nullptr ? p.bias[bn0 + j * 32 + col] : 0.f;
      wsv[j] = (PREC == 1 && p.wscale != nullptr) ? p.wscale[bn0 + j * 32 + col] * p.a_unmul : p.a_unmul;
    }
    if (p.res != nullptr) {
      const float* const rbase = p.res + row0 * p.ldr + bn0;
      const unsigned roff = (unsigned)(4 * half) * (unsigned)p.ldr + (unsigned)col;
      // the residual was written several kernels ago: every load is an HBM / Infinity-Cache round trip.
      // Issue them in two batches of 8 rows ahead of the math and the stores, so the latencies overlap
      // instead of being paid once per output row.
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          float rv[8][TN];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = hb * 8 + q;
            const long rr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) rv[q][j] = (rbase + rr * p.ldr + j * 32)[roff];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = hb * 8 + q;
            const long rr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float v = activate(PREC == 1 ? __builtin_fmaf(acc[i][j][r], wsv[j], bv[j]) : acc[i][j][r] + bv[j]);
              (obase + rr * p.ldo + j * 32)[loff] = v + rv[q][j];
              acc[i][j][r] = v;
            }
          }
        }
    } else if constexpr (EPI == 2) {
      // LayerNorm over the row (the tile holds all N = BN columns of its rows: gemm_ln_fusable): a row's values sit in
      // the 32 lanes of one half-wave x TN column blocks; sums over the blocks in j order, then a butterfly over the
      // lanes (the same two-pass form as ln_rows_kernel: mean, then the squared deviations).
      float lw[TN], lb[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) lw[j] = p.ln_w[j * 32 + col], lb[j] = p.ln_b[j * 32 + col];
      const float inv_n = 1.0f / (float)BN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long rr = i * 32 + (r & 3) + 8 * (r >> 2);
          float v[TN];
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            v[j] = PREC == 1 ? __builtin_fmaf(acc[i][j][r], wsv[j], bv[j]) : acc[i][j][r] + bv[j];
            sum += v[j];
          }
#pragma unroll
          for (int m = 16; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
          const float mean = sum * inv_n;
          float sq = 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            v[j] -= mean;
            sq += v[j] * v[j];
          }
#pragma unroll
          for (int m = 16; m > 0; m >>= 1) sq += __shfl_xor(sq, m);
          const float rstd = 1.0f / sqrtf(sq * inv_n + p.ln_eps);
#pragma unroll
          for (int j = 0; j < TN; ++j) (obase + rr * p.ldo + j * 32)[loff] = v[j] * rstd * lw[j] + lb[j];
        }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long rr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float v = activate(PREC == 1 ? __builtin_fmaf(acc[i][j][r], wsv[j], bv[j]) : acc[i][j][r] + bv[j]);
            (obase + rr * p.ldo + j * 32)[loff] = v;
            acc[i][j][r] = v;
          }
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = bn0 + j * 32 + col;
      const bool nok = n < p.N;
      const float bv = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
      const float ws = (PREC == 1 && p.wscale != nullptr && nok) ? p.wscale[n] * p.a_unmul : p.a_unmul;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = bm0 + wave * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = 0.f;
          if (m < M_eff && nok) {
            v = activate(PREC == 1 ? __builtin_fmaf(acc[i][j][r], ws, bv) : acc[i][j][r] + bv);
            if (p.crop_boxes != nullptr) {
              const float* bx = p.crop_boxes + ((long)z * p.crop_rows + m) * 4;
              const uint32_t py = fdiv((uint32_t)n, g.d_cw);
              const float fx = (float)((uint32_t)n - py * (uint32_t)p.crop_w), fy = (float)py;
              const bool inside = fx >= __fmul_rn(bx[0], p.crop_scale) && fx < __fmul_rn(bx[2], p.crop_scale) &&
                                  fy >= __fmul_rn(bx[1], p.crop_scale) && fy < __fmul_rn(bx[3], p.crop_scale);
              if (!inside) v = 0.f;
            }
            float o = v;
            if (p.res != nullptr) o += p.res[(long)m * p.ldr + n];
            long orow = m;
            if (g.remap) {
              const uint32_t img = fdiv((uint32_t)m, g.d_ohw);
              const uint32_t rem = (uint32_t)m - img * (uint32_t)(p.OH * p.OW);
              const uint32_t oh = fdiv(rem, g.d_ow);
              const uint32_t ow = rem - oh * (uint32_t)p.OW;
              orow = ((long)img * p.OH2 + oh * p.os + p.oy) * p.OW2 + ow * p.os + p.ox;
            }
            Op[orow * p.ldo + p.o_off + n] = o;
          }
          acc[i][j][r] = v;
        }
      }
    }
  }

  // ---- GRN partial sums of squares, segmented by image, fixed summation order ----
  if (p.grn_part != nullptr) {
    float* red = smem;  // [4][BN]; the K loop ended on a barrier, LDS is free
    const int m_end = (bm0 + BM < M_eff) ? bm0 + BM : M_eff;
    const int img_first = (int)fdiv((uint32_t)bm0, g.d_hw);
    const int img_last = (int)fdiv((uint32_t)(m_end - 1), g.d_hw);
    const bool one_image = interior && img_first == img_last;  // every row of the tile in one image: no row tests
    for (int s = 0; s <= img_last - img_first; ++s) {
      const int lo = (img_first + s) * p.hw, hi = lo + p.hw;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float sum = 0.f;
        if (one_image) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][j][r] * acc[i][j][r];
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = bm0 + wave * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              const float v = acc[i][j][r];
              sum += (m >= lo && m < hi) ? v * v : 0.f;
            }
        }
        sum += __shfl_xor(sum, 32);
        if (half == 0) red[wave * BN + j * 32 + col] = sum;
      }
      __syncthreads();
      if (tid < BN) {
        const int n = bn0 + tid;
        if (n < p.N)
          p.grn_part[((long)tile_m * p.segmax + s) * p.N + n] =
              ((red[tid] + red[BN + tid]) + red[2 * BN + tid]) + red[3 * BN + tid];
      }
      __syncthreads();
    }
  }
}

template <int TM, int TN, int BK, bool CONV, bool APRO, int EPI, int ACT, int PREC>
static void launch_one(const GemmDev& g, int grid, hipStream_t s) {
  const size_t lds = gemm_lds_bytes<TM, TN, BK, APRO, PREC>(g.nseg_max);
  static bool attr_done_dev[MTGV_MAX_DEVICES] = {};  // hipFuncSetAttribute is per device
  bool& attr_done = attr_done_dev[current_device()];  // only a launch that needs more than 64 KiB of dynamic LDS has to opt in
  if (lds > 64 * 1024 && !attr_done) {
    HIP_OK(hipFuncSetAttribute((const void*)gemm_f32_kernel<TM, TN, BK, CONV, APRO, EPI, ACT, PREC>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, BK, CONV, APRO, EPI, ACT, PREC>), dim3(grid, g.a.batch), dim3(256), lds, s, g);
}

// activation fixed at compile time for the combinations the path uses; anything else takes the
// runtime-switch instance (ACT = -1)
template <int TM, int TN, int BK, int PREC>
static void launch_variant(const GemmDev& g, bool conv, bool apro, int grid, hipStream_t s) {
  const int act = g.a.act;
  if (apro) {
    launch_one<TM, TN, BK, false, true, 0, ACT_NONE, PREC>(g, grid, s);
  } else if (conv) {
    if (act == ACT_SILU) launch_one<TM, TN, BK, true, false, 0, ACT_SILU, PREC>(g, grid, s);
    else if (act == ACT_NONE) launch_one<TM, TN, BK, true, false, 0, ACT_NONE, PREC>(g, grid, s);
    else launch_one<TM, TN, BK, true, false, 0, -1, PREC>(g, grid, s);
  } else {
    if (act == ACT_NONE) launch_one<TM, TN, BK, false, false, 0, ACT_NONE, PREC>(g, grid, s);
    else if (act == ACT_MISH) launch_one<TM, TN, BK, false, false, 0, ACT_MISH, PREC>(g, grid, s);
    else if (act == ACT_GELU) launch_one<TM, TN, BK, false, false, 0, ACT_GELU, PREC>(g, grid, s);
    else if (act == ACT_SILU) launch_one<TM, TN, BK, false, false, 0, ACT_SILU, PREC>(g, grid, s);
    else launch_one<TM, TN, BK, false, false, 0, -1, PREC>(g, grid, s);
  }
}

// Picks the instantiation for a plan.  top-k launches (match path: scores + per-tile top-k) use one tile
// shape, 128 queries x 64 bank rows, BK 16.  Returns false when no kernel exists for the tile.
template <int PREC>
static bool gemm_dispatch(const GemmDev& g, const GemmPlan& pl, bool conv, bool apro, int grid, hipStream_t s) {
  if (g.a.topk > 0) {
    launch_one<1, 2, 16, false, false, 1, ACT_NONE, PREC>(g, grid, s);
    return true;
  }
  if (g.a.ln_w != nullptr) {  // fused LayerNorm epilogue: the stem's tile only (gemm_ln_fusable)
    if (!(pl.tm == 1 && pl.tn == 3 && pl.bk == 16 && conv && !apro && g.a.act == ACT_NONE)) return false;
    launch_one<1, 3, 16, true, false, 2, ACT_NONE, PREC>(g, grid, s);
    return true;
  }
#define MTGV_CASE(TM_, TN_, BK_)                                \
  if (pl.tm == TM_ && pl.tn == TN_ && pl.bk == BK_) {           \
    launch_variant<TM_, TN_, BK_, PREC>(g, conv, apro, grid, s); \
    return true;                                                \
  }
  MTGV_CASE(1, 1, 16) MTGV_CASE(1, 2, 16) MTGV_CASE(1, 3, 16) MTGV_CASE(1, 4, 16) MTGV_CASE(1, 5, 16) MTGV_CASE(1, 1, 32)
#ifdef MTGV_ALL_TILES  // sweep-only shapes (tools/gemm_sweep.py); never chosen by gemm_plan
  MTGV_CASE(1, 2, 32) MTGV_CASE(1, 3, 32) MTGV_CASE(1, 4, 32) MTGV_CASE(2, 2, 16) MTGV_CASE(2, 2, 32) MTGV_CASE(2, 1, 16)
  MTGV_CASE(2, 3, 16) MTGV_CASE(2, 4, 16) MTGV_CASE(2, 3, 32)
#endif
#undef MTGV_CASE
  return false;
}

}  // namespace mtgv
