// Implicit-GEMM conv / linear on gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Why f32 MFMA: the path's contract is fp32 embeddings within 1e-4 of the
// PyTorch CPU reference and identical top-1 ids; plain fp16 operands miss that
// by 25x.  gfx950 has no TF32, but it does have an exact
// f32-in/f32-acc matrix instruction (157 TFLOP/s dense), so every GEMM-shaped
// op of the path runs on it.
//
// Tile: 256 threads = 4 waves stacked along M.  Block tile BM = 128*TM rows by
// BN = 32*TN columns; wave w owns rows [w*32*TM, (w+1)*32*TM) x all BN columns
// as TM x TN accumulators of 32x32.  K is consumed in BK-wide steps staged
// through LDS (rows padded by 16 B so ds_read_b128 is conflict-free), with the
// next step's global loads in flight while the current one feeds the MFMAs.
#include "gemm_f32.h"
#include "act.h"

#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace mtgv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmDev {
  GemmArgs a;
  FastDiv d_ohw, d_ow, d_cin, d_kwcin, d_hw, d_cw;
  int tiles_m, tiles_n;
  int remap;     // output rows are not simply m
  int nseg_max;  // APRO: images a 128-row tile can touch (sizes the LDS multiplier tile)
};

template <int TM, int TN, int BK, bool CONV, bool APRO, int EPI, int ACT>
__global__ __launch_bounds__(256, (TM * TN == 1 ? (BK == 16 ? 6 : 4) : TM * TN == 2 ? (BK == 16 ? 5 : 4) : TM * TN == 3 ? (BK == 16 ? 4 : 3) : TM * TN == 4 ? (BK == 16 ? 3 : 2) : 2)) void gemm_f32_kernel(const GemmDev g) {
  constexpr int BM = 128 * TM, BN = 32 * TN, LS = BK + 4;
  constexpr int KQ = BK / 4;      // float4 per staged row
  constexpr int RPP = 256 / KQ;   // rows staged per pass
  constexpr int AP = BM / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  static_assert(BM % RPP == 0, "A tile must be a whole number of passes");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                // [2][BM][LS]
  float* Bs = smem + 2 * BM * LS;  // [2][BN][LS]

  const GemmArgs& p = g.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.y;
  const float* const Ap = p.A + (long)z * p.strideA;
  const float* const Wp = p.W + (long)z * p.strideW;
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
  // APRO (GRN apply): the multipliers s[img][k] of the images this tile touches are staged per K step into
  // LDS (Ss[2][nseg][BK]) and applied to the A fragments as they are read - the A loads stay plain.
  constexpr int SPT = APRO ? (BM * KQ + 255) / 256 : 1;  // multiplier float4s a thread may have to stage
  f32x4 rsl[SPT];
  float* Ss = Bs + 2 * BN * LS;
  const int img_first_t = APRO ? (int)fdiv((uint32_t)bm0, g.d_hw) : 0;
  int nseg_t = 1;
  if (APRO) {
    const int m_end_t = (bm0 + BM < M_eff) ? bm0 + BM : M_eff;
    nseg_t = (int)fdiv((uint32_t)(m_end_t - 1), g.d_hw) - img_first_t + 1;
  }
  auto load_tile = [&](int kt) {
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
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok[i] && kok) v = *reinterpret_cast<const f32x4*>(Wp + b_row[i] + k);
      rb[i] = v;
    }
    if (APRO) {
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
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* Ab = As + cur * BM * LS;
    const float* Bb = Bs + cur * BN * LS;
    if (wave_active) {
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

  if constexpr (EPI == 1) {
    // ---- fused per-tile top-k (match path): the tile's scores never leave registers.
    // A row's BN scores sit in the 32 lanes of one wave half x TN registers; each
    // round picks the (score desc, id asc) maximum with a 5-step butterfly and
    // retires it.  Candidates go to cand[m][tile_n][kk]; a merge kernel finishes.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool nok = (bn0 + j * 32 + col) < p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!nok) acc[i][j][r] = -INFINITY;
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
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = p.bias != nullptr ? p.bias[bn0 + j * 32 + col] : 0.f;
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
              const float v = activate(acc[i][j][r] + bv[j]);
              (obase + rr * p.ldo + j * 32)[loff] = v + rv[q][j];
              acc[i][j][r] = v;
            }
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long rr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float v = activate(acc[i][j][r] + bv[j]);
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
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = bm0 + wave * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = 0.f;
          if (m < M_eff && nok) {
            v = activate(acc[i][j][r] + bv);
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
  long launches = 0;
  struct Rec { int M, N, K, KH, stride, batch, act, apro, grn, topk; };
  std::vector<Rec> recs;
} g_prof;
}  // namespace

void gemm_profile_enable(bool on) {
  g_prof.on = on;
  g_prof.used = 0;
  g_prof.flops = 0;
  g_prof.launches = 0;
  g_prof.recs.clear();
}

// per-launch table (shape, ms, TFLOP/s) of everything recorded since enable; tuning aid
void gemm_profile_dump(const char* path) {
  FILE* f = fopen(path, "w");
  MTGV_CHECK(f != nullptr, ERR_RUNTIME, "cannot open %s", path);
  fprintf(f, "idx,M,N,K,KH,stride,batch,act,apro,grn,topk,tm,tn,bk,ms,tflops\n");
  for (size_t i = 0; i + 1 < g_prof.used && i / 2 < g_prof.recs.size(); i += 2) {
    HIP_OK(hipEventSynchronize(g_prof.ev[i + 1]));
    float t = 0.f;
    HIP_OK(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
    const auto& r = g_prof.recs[i / 2];
    const GemmPlan pl = r.topk ? GemmPlan{1, 2, 16, 0, 0} : gemm_plan(r.M, r.N, r.K, r.act != 0, r.apro != 0);
    const double fl = 2.0 * r.M * r.N * r.K * r.batch;
    fprintf(f, "%zu,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.2f\n", i / 2, r.M, r.N, r.K, r.KH, r.stride, r.batch, r.act,
            r.apro, r.grn, r.topk, pl.tm, pl.tn, pl.bk, t, fl / (t * 1e-3) / 1e12);
  }
  fclose(f);
}

bool gemm_profile_enabled() { return g_prof.on; }

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

static void prof_begin(const GemmArgs& a, hipStream_t s) {
  if (!g_prof.on) return;
  while (g_prof.ev.size() < g_prof.used + 2) {
    hipEvent_t e;
    HIP_OK(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  HIP_OK(hipEventRecord(g_prof.ev[g_prof.used], s));
  g_prof.flops += 2.0 * (double)a.M * a.N * a.K * a.batch;
  g_prof.launches += 1;
  g_prof.recs.push_back({a.M, a.N, a.K, a.KH, a.stride, a.batch, a.act, a.a_scale != nullptr, a.grn_part != nullptr, a.topk});
}
static void prof_end(hipStream_t s) {
  if (!g_prof.on) return;
  HIP_OK(hipEventRecord(g_prof.ev[g_prof.used + 1], s));
  g_prof.used += 2;
}

GemmPlan gemm_plan(int M, int N, int K, bool heavy_epilogue, bool scaled_a) {
  GemmPlan pl;
  if (const char* e = getenv("MTGV_GEMM_TILE")) {
    int tm = 0, tn = 0, bk = 0;
    if (sscanf(e, "%d,%d,%d", &tm, &tn, &bk) == 3 && (tm == 1 || (tm == 2 && tn == 2)) && tn >= 1 && tn <= 5 &&
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
  const double ov = 36.0 + (heavy_epilogue ? 48.0 : 0.0);
  static const double eff_plain[6] = {0, 0.88, 0.93, 1.00, 0.80, 0.62};
  static const double eff_heavy[6] = {0, 0.85, 0.93, 0.98, 1.00, 0.85};
  static const double eff_scaled[6] = {0, 0.80, 0.90, 1.00, 0.85, 0.80};  // per-fragment GRN multiply amortises over TN
  int best_tn = 1;
  double best = -1;
  const long tiles_m = ceil_div(M, 128);
  for (int tn = 1; tn <= 5; ++tn) {
    const long tiles = tiles_m * ceil_div(N, 32 * tn);
    const double rounds = (double)((tiles + 255) / 256);
    const double cost = rounds * 128.0 * 32.0 * tn * ((double)K + ov) / (scaled_a ? eff_scaled[tn] : heavy_epilogue ? eff_heavy[tn] : eff_plain[tn]);
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

template <int TM, int TN, int BK, bool CONV, bool APRO, int EPI, int ACT>
static void launch_one(const GemmDev& g, int grid, hipStream_t s) {
  const size_t lds = (size_t)2 * (128 * TM + 32 * TN) * (BK + 4) * sizeof(float) +
                     (APRO ? (size_t)2 * g.nseg_max * BK * sizeof(float) : 0);
  static bool attr_done = false;  // >64 KiB of dynamic LDS must be opted into once per kernel
  if (!attr_done) {
    HIP_OK(hipFuncSetAttribute((const void*)gemm_f32_kernel<TM, TN, BK, CONV, APRO, EPI, ACT>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, BK, CONV, APRO, EPI, ACT>), dim3(grid, g.a.batch), dim3(256), lds, s, g);
}

// activation fixed at compile time for the combinations the path uses; anything else takes the
// runtime-switch instance (ACT = -1)
template <int TM, int TN, int BK>
static void launch_variant(const GemmDev& g, bool conv, bool apro, int grid, hipStream_t s) {
  const int act = g.a.act;
  if (apro) {
    launch_one<TM, TN, BK, false, true, 0, ACT_NONE>(g, grid, s);
  } else if (conv) {
    if (act == ACT_SILU) launch_one<TM, TN, BK, true, false, 0, ACT_SILU>(g, grid, s);
    else if (act == ACT_NONE) launch_one<TM, TN, BK, true, false, 0, ACT_NONE>(g, grid, s);
    else launch_one<TM, TN, BK, true, false, 0, -1>(g, grid, s);
  } else {
    if (act == ACT_NONE) launch_one<TM, TN, BK, false, false, 0, ACT_NONE>(g, grid, s);
    else if (act == ACT_MISH) launch_one<TM, TN, BK, false, false, 0, ACT_MISH>(g, grid, s);
    else if (act == ACT_GELU) launch_one<TM, TN, BK, false, false, 0, ACT_GELU>(g, grid, s);
    else if (act == ACT_SILU) launch_one<TM, TN, BK, false, false, 0, ACT_SILU>(g, grid, s);
    else launch_one<TM, TN, BK, false, false, 0, -1>(g, grid, s);
  }
}

// match path: scores + per-tile top-k, one tile shape (128 queries x 64 bank rows, BK 16)
static void launch_topk(const GemmDev& g, int grid, hipStream_t s) { launch_one<1, 2, 16, false, false, 1, ACT_NONE>(g, grid, s); }

void gemm_launch(const GemmArgs& a, const GemmPlan& pl, hipStream_t s) {
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
  if (a.grn_part) MTGV_CHECK(a.segmax >= gemm_grn_segmax(pl, a.hw), ERR_INVALID, "gemm: segmax too small");

  GemmDev g;
  g.a = a;
  g.d_ohw = make_fastdiv((uint32_t)(a.OH * a.OW));
  g.d_ow = make_fastdiv((uint32_t)a.OW);
  g.d_cin = make_fastdiv((uint32_t)a.Cin);
  g.d_kwcin = make_fastdiv((uint32_t)(a.KW * a.Cin));
  g.d_hw = make_fastdiv((uint32_t)(a.hw > 0 ? a.hw : 1));
  g.d_cw = make_fastdiv((uint32_t)(a.crop_w > 0 ? a.crop_w : 1));
  g.tiles_m = pl.tiles_m;
  g.tiles_n = pl.tiles_n;
  g.nseg_max = a.hw > 0 ? std::min(pl.bm(), (pl.bm() - 1) / a.hw + 2) : 1;
  g.remap = !(a.os == 1 && a.oy == 0 && a.ox == 0 && a.OH2 == a.OH && a.OW2 == a.OW);
  const int grid = pl.tiles_m * pl.tiles_n;

  prof_begin(a, s);
  if (a.topk > 0) {
    MTGV_CHECK(pl.tm == 1 && pl.tn == 2 && pl.bk == 16 && !conv && !apro, ERR_INVALID, "gemm: top-k epilogue needs the 128x64x16 tile");
    MTGV_CHECK(a.cand_s != nullptr && a.cand_i != nullptr && a.topk <= 128, ERR_INVALID, "gemm: bad top-k arguments");
    launch_topk(g, grid, s);
    HIP_OK(hipGetLastError());
    prof_end(s);
    return;
  }

#define MTGV_CASE(TM_, TN_, BK_)                                   \
  if (pl.tm == TM_ && pl.tn == TN_ && pl.bk == BK_) {              \
    launch_variant<TM_, TN_, BK_>(g, conv, apro, grid, s);         \
    HIP_OK(hipGetLastError());                                     \
    prof_end(s);                                                   \
    return;                                                        \
  }
  MTGV_CASE(1, 1, 16) MTGV_CASE(1, 2, 16) MTGV_CASE(1, 3, 16) MTGV_CASE(1, 4, 16) MTGV_CASE(1, 5, 16) MTGV_CASE(1, 1, 32)
#ifdef MTGV_ALL_TILES  // sweep-only shapes (tools/gemm_sweep.py); never chosen by gemm_plan
  MTGV_CASE(1, 2, 32) MTGV_CASE(1, 3, 32) MTGV_CASE(1, 4, 32) MTGV_CASE(2, 2, 16) MTGV_CASE(2, 2, 32)
#endif
#undef MTGV_CASE
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
    for (int t = t_first; t <= t_last; ++t) {
      const int seg = img - (int)fdiv((uint32_t)(t * bm), d_hw);
      sum += part[((long)t * segmax + seg) * N + n];
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

void grn_finalize_launch(const float* part, const GemmPlan& p, int n_img, int hw, int N, const float* gamma, float* scale,
                         hipStream_t s) {
  const size_t lds = (size_t)(N + 256) * sizeof(float);
  MTGV_CHECK(lds <= 160 * 1024, ERR_INVALID, "grn_finalize: N=%d too large", N);
  hipLaunchKernelGGL(grn_finalize_kernel, dim3(n_img), dim3(256), lds, s, part, p.bm(), gemm_grn_segmax(p, hw), hw, N,
                     make_fastdiv((uint32_t)hw), make_fastdiv((uint32_t)p.bm()), gamma, scale);
  HIP_OK(hipGetLastError());
}

}  // namespace mtgv
