// Depthwise 7x7 + LayerNorm, row-streaming form (Block.dwconv + Block.norm, convnextv2.py:198-200, 214-216): the
// input is staged in LDS by LDS-DMA and every input row is fetched from memory ONCE.
//
// A block owns a band of rows of one image at full width and walks down it TH output rows at a time.  The input rows
// live in a ring of R = 2 TH + 6 row slots in LDS, row-major exactly as in memory ([pixel][channel], W * C floats per
// row): the TH + 6 rows the current step reads, plus the TH rows of the next step, which global_load_lds (1 KB per wave
// instruction, no VGPR round trip) fills while the current step computes.  There is no vertical halo re-read at all -
// the launch moves every input byte into LDS exactly once (bands > 1: plus six rows per band) - against 7.5 vector-L1
// requests per output in dwconv7_ln_rows_kernel.
//
// Thread = ONE channel of a TW-pixel strip (lanes run over channels: LDS reads are consecutive words, conflict-free;
// output stores are consecutive words).  With one channel per thread the 49 taps are 49 VGPRs, so the tap table is
// never re-read (the quad-per-thread kernels read it from LDS 12 times per output quad, as much LDS time as their FMAs
// take VALU time), and the kernel's only per-output memory instructions are 7.5 ds_read_b32 (2 clocks each).
// NT = (W / TW) * C threads (768 for every ConvNeXt-V2 tiny stage: W * C = 3072), one block per CU.
//
// Arithmetic is the quad kernels' to the bit: bias, then taps kh ascending, kw ascending, one FMA each; LayerNorm sums
// per pixel as (a0 + a1) + (a2 + a3) per channel quad (two DPP steps), then the quads in ascending order by one
// thread; two-pass variance.
#pragma once
#include <type_traits>

#include "common.h"
#include "sp8.h"

namespace mtgv {

// floats in front of the row ring: LayerNorm partials + statistics, at least 3 C (see the kernel), a multiple of 4
constexpr int dwconv7_ln_stream_scratch_floats(int C, int G, int TW, int TH) {
  const int need = G * (C / 4) * TH * TW + G * TH * TW * 2;
  const int s = need > 3 * C ? need : 3 * C;
  return (s + 3) / 4 * 4;
}

template <int C, int G, int TW, int TH, bool SP8>
__global__ __launch_bounds__(C* G) void dwconv7_ln_stream_kernel(const float* __restrict__ in, const float* __restrict__ w49,
                                                                const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                                const float* __restrict__ ln_b, float* __restrict__ out, int H,
                                                                int bands, int rows_per_band, float eps) {
  constexpr int NT = C * G, NW = NT / 64, W = G * TW, ROWF = W * C, R = 2 * TH + 6, P = TH * TW, C4N = C / 4;
  constexpr int PPR = ROWF * 4 / 1024;  // 1 KB DMA pieces per row
  static_assert(NT % 64 == 0 && NT <= 1024 && ROWF % 256 == 0 && C % 8 == 0, "shape");
  extern __shared__ __attribute__((aligned(1024))) char dws_smem[];
  // The LayerNorm scratch comes first: strip 0's three masked left columns address up to 3 C floats BELOW their row, which
  // for the ring's first slot must still be inside the allocation (padded up to 3 C floats where the scratch is smaller)
  constexpr int SCR = dwconv7_ln_stream_scratch_floats(C, G, TW, TH);
  float* const part = reinterpret_cast<float*>(dws_smem);  // [G * C4N][P]: a pixel's quad partials sit P floats apart
  float* const stat = part + G * C4N * P;                  // [G * P][2] mean, rstd
  float* const ring = part + SCR;                          // [R][ROWF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = tid / C, ch = tid - g * C, c4 = ch >> 2;
  const int w0 = g * TW;
  const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
  const int hb0 = band * rows_per_band;
  const int hb1 = hb0 + rows_per_band < H ? hb0 + rows_per_band : H;
  if (hb0 >= hb1) return;
  const float* const img = in + (long)n * H * ROWF;

  // image rows [r0, r0 + nrows) -> their ring slots (row r lives in slot (r - (hb0 - 3)) % R); rows outside the image
  // are skipped (the compute loop never reads them)
  auto issue_rows = [&](int r0, int nrows) {
    for (int p = wave; p < nrows * PPR; p += NW) {
      const int rr = p / PPR, chunk = p - rr * PPR;
      const int row = r0 + rr;
      if (row < 0 || row >= H) continue;
      const int slot = (row - (hb0 - 3)) % R;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (long)row * ROWF + chunk * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(ring + slot * ROWF + chunk * 256), 16, 0, 0);
    }
  };
  issue_rows(hb0 - 3, TH + 6);

  float wt[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) wt[t] = w49[t * C + ch];
  const float bv = bias[ch], lw = ln_w[ch], lb = ln_b[ch];
  // horizontal zero padding: only the first strip's three left columns and the last strip's three right columns can
  // fall outside the image
  bool jok[TW + 6];
#pragma unroll
  for (int j = 0; j < TW + 6; ++j) jok[j] = (w0 + j - 3) >= 0 && (w0 + j - 3) < W;
  const int lane_off = (w0 - 3) * C + ch;  // float offset of this thread's first column inside a row (may be negative: masked)
  const bool wave_edge = __builtin_amdgcn_ballot_w64(g == 0 || g == G - 1) != 0;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int sbase = 0;  // ring slot of row h0 - 3
  for (int h0 = hb0; h0 < hb1; h0 += TH) {
    if (h0 + TH < hb1) issue_rows(h0 + TH + 3, TH);  // the next step's new rows, into the slots the previous step freed
    float acc[TH][TW];
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[t][j] = bv;
    // MASK: this wave holds lanes of the first or last strip (columns outside the image read as zero); the other waves
    // run the same loop without the selects
    auto conv_rows = [&](auto MASK_T) {
      constexpr bool MASK = decltype(MASK_T)::value;
#pragma unroll
      for (int ir = 0; ir < TH + 6; ++ir) {
        const int ih = h0 - 3 + ir;
        if (ih < 0 || ih >= H) continue;  // (wave-uniform)
        int slot = sbase + ir;
        slot = slot >= R ? slot - R : slot;
        const float* const rowp = ring + slot * ROWF + lane_off;
        float r[TW + 6];
#pragma unroll
        for (int j = 0; j < TW + 6; ++j) {
          const float v = rowp[j * C];  // (an address below the ring for masked columns of strip 0 stays inside LDS: unused)
          r[j] = (!MASK || (j >= 3 && j < TW + 3)) ? v : (jok[j] ? v : 0.f);
        }
#pragma unroll
        for (int t = 0; t < TH; ++t) {
          const int kh = ir - t;  // output row h0 + t sees this input row as its tap row kh
          if (kh < 0 || kh > 6) continue;
#pragma unroll
          for (int kw = 0; kw < 7; ++kw)
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[t][j] = __builtin_fmaf(r[j + kw], wt[kh * 7 + kw], acc[t][j]);
        }
      }
    };
    if (wave_edge) conv_rows(std::true_type{});
    else conv_rows(std::false_type{});
    sbase += TH;
    sbase = sbase >= R ? sbase - R : sbase;

    // ---- LayerNorm over C per pixel ----
    // quad sums (a0 + a1) + (a2 + a3) by two DPP steps; lane 4k of a quad writes the partial
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const float a = acc[t][j];
        const float s1 = a + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, false));
        const float s2 = s1 + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, false));
        if ((ch & 3) == 0) part[(g * C4N + c4) * P + t * TW + j] = s2;
      }
    // this wave's DMA pieces for the next step have landed long ago (issued before the step's FMAs), and so have the
    // previous step's stores; the barrier then says the same of every wave's pieces
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ch < P) {  // thread ch of a strip reduces pixel ch: quads in ascending order
      const float* pp = part + g * C4N * P + ch;
      float sum = 0.f;
      for (int i = 0; i < C4N; ++i) sum += pp[i * P];
      stat[(g * P + ch) * 2] = sum / (float)C;
    }
    __syncthreads();
    float mean[TH][TW];
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        mean[t][j] = stat[(g * P + t * TW + j) * 2];
        const float d = acc[t][j] - mean[t][j];
        // the quad kernels' (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) as the compiler contracts it:
        // fma(d0, d0, d1 * d1) + fma(d2, d2, d3 * d3)
        const float sq = d * d;
        const float nb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sq), 0xB1, 0xF, 0xF, false));
        const float u = __builtin_fmaf(d, d, nb);  // meaningful on even lanes: fma(d_even, d_even, d_odd^2)
        const float v = u + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, u), 0x4E, 0xF, 0xF, false));
        if ((ch & 3) == 0) part[(g * C4N + c4) * P + t * TW + j] = v;
      }
    __syncthreads();
    if (ch < P) {
      const float* pp = part + g * C4N * P + ch;
      float sq = 0.f;
      for (int i = 0; i < C4N; ++i) sq += pp[i * P];
      stat[(g * P + ch) * 2 + 1] = 1.0f / sqrtf(sq / (float)C + eps);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TH; ++t) {
      if (h0 + t >= hb1) continue;
      float* const op = out + (((long)n * H + h0 + t) * W + w0) * C;
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const float rstd = stat[(g * P + t * TW + j) * 2 + 1];
        float o = (acc[t][j] - mean[t][j]) * rstd * lw + lb;
        if (SP8) {
          // (o as an f32 value first: left to itself the compiler converts the un-rounded FMA straight to fp16 -
          // v_fma_mixlo_f16 - and the hi half then differs from the quad kernels' in the double-rounding cases)
          asm volatile("" : "+v"(o));
          // chunk of 8 channels = [8 fp16 hi][8 fp16 lo]: a lane pair (2i, 2i + 1) trades halves so that the even lane
          // holds the pair's two hi halves and the odd lane its two lo halves - one coalesced dword store per pixel
          const _Float16 hi = (_Float16)o;
          const _Float16 lo = (_Float16)(o - (float)hi);
          const uint32_t wv = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
          const uint32_t rv = (uint32_t)__builtin_amdgcn_mov_dpp((int)wv, 0xB1, 0xF, 0xF, false);
          const uint32_t ow = (ch & 1) ? ((rv >> 16) | (wv & 0xffff0000u)) : ((wv & 0xffffu) | (rv << 16));
          char* const cb = reinterpret_cast<char*>(op + (long)j * C) + (ch >> 3) * 32 + (ch & 1) * 16 + ((ch & 7) >> 1) * 4;
          *reinterpret_cast<uint32_t*>(cb) = ow;
        } else {
          op[(long)j * C + ch] = o;
        }
      }
    }
  }
}

// LDS bytes of an instantiation
template <int C, int G, int TW, int TH>
constexpr size_t dwconv7_ln_stream_lds() {
  return (size_t)((2 * TH + 6) * (G * TW * C) + dwconv7_ln_stream_scratch_floats(C, G, TW, TH)) * sizeof(float);
}

template <int C, int G, int TW, int TH, bool SP8>
static void dwconv7_ln_stream_launch(const float* in, const float* w49, const float* bias, const float* ln_w, const float* ln_b, float* out,
                                     int N, int H, int bands, float eps, hipStream_t s) {
  constexpr size_t lds = dwconv7_ln_stream_lds<C, G, TW, TH>();
  static_assert(lds <= 160 * 1024, "the row ring must fit one CU's LDS");
  auto kern = dwconv7_ln_stream_kernel<C, G, TW, TH, SP8>;
  static bool attr[MTGV_MAX_DEVICES] = {};
  const int dev = current_device();
  if (!attr[dev]) {
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr[dev] = true;
  }
  const int rows_per_band = ceil_div(ceil_div(H, bands), TH) * TH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(N * bands)), dim3(C * G), lds, s, in, w49, bias, ln_w, ln_b, out, H, bands, rows_per_band, eps);
  HIP_OK(hipGetLastError());
}

}  // namespace mtgv
