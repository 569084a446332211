// Fused ConvNeXt-V2 MLP for the narrow stages (C = 80 / 96: stage 0 of the nano / tiny encoders).
//
//   Block.forward, convnextv2.py:212-224, after the depthwise conv + LayerNorm:
//     h   = act(x W1^T + b1)                    pwconv1, [M][4C]
//     s   = gamma * Gx / (mean Gx + 1e-6) + 1   GRN, Gx[img][k] = || h[img, :, :, k] ||_2   (convnextv2.py:171-174)
//     out = (h * s) W2^T + b2' + res            pwconv2 (b2' = b2 + W2 beta), residual
//
// In the unfused path the 4C-wide hidden tensor makes a round trip through HBM (stage 0 of AE-tiny at batch 256:
// 604 MB written by pwconv1, read back by pwconv2) and both launches are bound by it.  Here h never leaves the CU:
//
//   PASS 1  GEMM1 + activation, sum of squares per (32-row unit, hidden channel) -> part[M/32][4C]  (no h written)
//   (grn_finalize_kernel turns the partial sums into s)
//   PASS 2  GEMM1 again, chunk by chunk of 32 hidden channels; the chunk's accumulator is activated, scaled by s,
//           split into fp16 hi / lo and used, as it stands in registers, as the activation operand of GEMM2.
//
// One block = 4 waves x 32 rows.  A wave keeps its 32 x C slice of x (SP8: sp8.h) in registers for the whole tile, as
// MFMA fragments; the weights stream through a ring of three LDS slots by LDS-DMA (global_load_lds_dwordx4), one slot
// = the W1 rows of one hidden chunk ([32][C], as C/32 sub-blocks of [32][128 B]) or the W2 columns of one chunk
// ([C][32 k] = [C][128 B]); slots are consumed in the order W1(0), W2(0), W1(1), W2(1), ... with two (PASS 1) or four
// (PASS 2: a ring of six) slots in flight and one s_barrier per chunk.
// Per-channel vectors (row scales of W1, b1, and the GRN multipliers of the tile's <= 2 images) are staged once per tile.
//
// MFMA orientation.  PASS 2 uses "m on lanes" for both GEMMs: v_mfma_f32_32x32x16_f16(W fragment, x fragment) leaves
// row m = lane & 31 on the lane and hidden channels 8 gq + 4 (lane >> 5) + e in register 4 gq + e.  Two register
// groups (gq = 2t, 2t + 1) are exactly the 8 k values lane (m, h) has to supply to the k16 step t of GEMM2 - provided
// W2 is stored with the matching k order inside every group of 16: position p of half h holds
// k = 16 t + 8 (p >> 2) + 4 h + (p & 3)  (mlp_pack_w2p_kernel).  So the hidden tensor goes accumulator -> VALU ->
// operand without touching LDS.  PASS 1 uses the other orientation (x fragment first): a lane then holds ONE hidden
// channel of 16 rows and its sum of squares needs no cross-lane work beyond one xor-32 exchange.
//
// Products: lo*hi + hi*lo + hi*hi per k16 step, k ascending; results do not depend on the batch or the tile.
#pragma once
#include <type_traits>

#include "act.h"
#include "sp8.h"

namespace mtgv {

typedef float mlp_f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* mlp_gptr;
typedef __attribute__((address_space(3))) void* mlp_lptr;

struct MlpDev {
  const char* X = nullptr;      // SP8 rows [M][C] (LayerNorm output)
  const char* W1 = nullptr;     // SP8 [4C][C], rows scaled by 1 / ws1
  const float* ws1 = nullptr;   // [4C]
  const float* b1 = nullptr;    // [4C]
  const char* W2p = nullptr;    // SP8 [C][4C] with the k order of mlp_pack_w2p_kernel, rows scaled by 1 / ws2
  const float* ws2 = nullptr;   // [C]
  const float* b2 = nullptr;    // [C], GRN beta folded in
  const float* scale = nullptr; // [n_img][4C] GRN multipliers (PASS 2)
  const float* res = nullptr;   // [M][C] f32
  float* Out = nullptr;         // [M][C] f32
  float* part = nullptr;        // PASS 1: [M / 32][4C]
  const char* zero = nullptr;   // >= 16 zero bytes
  long* stamps = nullptr;       // tuning aid (MTGV_MLP_STAMPS): [tile][8] clock stamps and wait sums of wave 0
  // PASS 2, last block of a stage: the next layer's LayerNorm over C (the downsample's, convnextv2.py:258-263) in the epilogue -
  // OutLn receives the normalised rows in SP8 form and Out is not written (nothing else reads the stage's last f32 output)
  char* OutLn = nullptr;        // SP8 rows [M][C]; may alias res (a tile reads its residual rows before it stores)
  const float* ln_w = nullptr;  // [C]
  const float* ln_b = nullptr;  // [C]
  float ln_eps = 1e-6f;
  int M = 0, hw = 1, n_img = 1;
  FastDiv d_hw;
};

// C = 16 * C16.  PASS 1: statistics; PASS 2: output.
template <int C16, int ACT, int PASS>
__global__ __launch_bounds__(256, 2) void mlp_fused_kernel(const MlpDev g) {
#pragma clang fp contract(off)
  constexpr int C = 16 * C16, H4 = 4 * C, NCH = H4 / 32;
  constexpr int KB = (C16 + 1) / 2;          // 32-k sub-blocks of a W1 slot = 32-column blocks of the output
  constexpr int NW = 4, BM = 128;
  constexpr int SLOT = KB * 4096, PPW = KB;   // KB * 4 pieces of 1 KB per slot, KB per wave
  constexpr int NV = PASS == 2 ? 4 : 2;       // staged 4C-vectors: ws1, b1, (s of image 0, s of image 1)
  constexpr int VB = H4 * 4;                  // bytes per vector
  constexpr int XB = PASS == 2 ? 2 * C * 4 : 0;  // PASS 2: ws2 and b2 (C floats each) behind them
  constexpr int EXP = (NV * VB + XB + 1023) / 1024, EXB = EXP * 1024;
  constexpr int NS = PASS == 2 ? 2 * NCH : NCH;  // weight slots in consumption order
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const ring = smem + EXB;  // PASS 1: 3 slots, PASS 2: 6

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const bool stamp = g.stamps != nullptr;
  const long st0 = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
  long st1 = 0, wa = 0, wb = 0;
  const int mrow = m0 + wave * 32 + r;
  const int mc = mrow < g.M ? mrow : g.M - 1;
  const bool wave_active = m0 + wave * 32 < g.M;

  // ---- x fragments: k16 step s needs chunks 2 s + h of the row (hi 16 B | lo 16 B) ----
  sp_h8 xh[C16], xl[C16];
  {
    const char* xp = g.X + (long)mc * (C * 4) + h * 32;
#pragma unroll
    for (int s = 0; s < C16; ++s) {
      xh[s] = *reinterpret_cast<const sp_h8*>(xp + s * 64);
      xl[s] = *reinterpret_cast<const sp_h8*>(xp + s * 64 + 16);
    }
  }

  // ---- staged vectors ----
  const int img0 = (int)fdiv((uint32_t)m0, g.d_hw);
  {
    const int img1 = img0 + 1 < g.n_img ? img0 + 1 : g.n_img - 1;
    for (int p = wave; p < EXP; p += NW) {
      const int o = p * 1024 + lane * 16;
      const int v = o / VB, w = o - v * VB;
      const char* s = g.zero;
      if (v == 0) s = reinterpret_cast<const char*>(g.ws1) + w;
      else if (v == 1) s = reinterpret_cast<const char*>(g.b1) + w;
      else if (PASS == 2 && v == 2) s = reinterpret_cast<const char*>(g.scale) + (long)img0 * VB + w;
      else if (PASS == 2 && v == 3) s = reinterpret_cast<const char*>(g.scale) + (long)img1 * VB + w;
      else if (PASS == 2 && v == 4 && w < C * 4) s = reinterpret_cast<const char*>(g.ws2) + w;
      else if (PASS == 2 && v == 4 && w < 2 * C * 4) s = reinterpret_cast<const char*>(g.b2) + (w - C * 4);
      __builtin_amdgcn_global_load_lds((mlp_gptr)s, (mlp_lptr)(smem + p * 1024), 16, 0, 0);
    }
  }

  // ---- weight slots.  Piece p = wave + 4 u of a slot: 8 rows x 128 B; lane -> (row, 16-byte slot ^ swizzle) ----
  // W1 slot of chunk j: sub-block u, rows 8 wave + lane / 8 (hidden channel 32 j + row), k bytes u * 128 + slot * 16
  // W2 slot of chunk j: rows 32 u + 8 wave + lane / 8 (output channel), k bytes j * 128 + slot * 16
  // Pieces are addressed as uniform base (advanced per slot by scalar adds) + a 32-bit lane offset fixed for the whole
  // tile (sp_dma16_saddr).  Lanes whose bytes no MFMA consumes - k beyond C in the last sub-block of a W1 slot, output
  // rows beyond C in a W2 slot (their columns are masked in the epilogue) - fetch a valid neighbour instead.
  const int prow = 8 * wave + (lane >> 3);
  const int pslot = (lane & 7) ^ ((prow >> 1) & 7);
  uint32_t w1_off[PPW], w2_off[PPW];
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    const bool tail = u * 32 + (pslot >> 1) * 8 >= C;
    w1_off[u] = (uint32_t)(prow * (C * 4) + (tail ? (pslot & 1) : pslot) * 16);
    const int n = 32 * u + prow < C ? 32 * u + prow : C - 1;
    w2_off[u] = (uint32_t)(n * (H4 * 4) + pslot * 16);
  }
  auto issue = [&](int sl, int rb) {  // slot sl (consumption order) into ring buffer rb
    const char* const dst = ring + rb * SLOT + wave * 1024;
    if (PASS == 1 || (sl & 1) == 0) {
      const int j = PASS == 1 ? sl : sl >> 1;
      const char* const base = g.W1 + (long)j * (32 * C * 4);
#pragma unroll
      for (int u = 0; u < PPW; ++u) sp_dma16_saddr(base + u * 128, w1_off[u], dst + u * 4096);
    } else {
      const char* const base = g.W2p + (long)(sl >> 1) * 128;
#pragma unroll
      for (int u = 0; u < PPW; ++u) sp_dma16_saddr(base, w2_off[u], dst + u * 4096);
    }
  };
  const unsigned swr = (unsigned)(r >> 1) & 7u;
  const unsigned frag = (unsigned)r * 128u;
  auto activate = [&](float x) -> float {
    if constexpr (ACT == ACT_GELU) return act_gelu(x);
    else return act_mish(x);
  };
  // this wave's pieces of every slot but the N youngest have landed
  auto wait_but = [&](int slots_in_flight) {
    if (slots_in_flight >= 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * PPW) : "memory");
    else if (slots_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
    else if (slots_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };
  // GEMM1 of one hidden chunk out of a W1 slot.  XFIRST: x fragment as the first operand (hidden channel on lanes).
  // Every fragment of the chunk is requested before the first MFMA (the registers are there: two waves per SIMD), so
  // the matrix pipe does not wait on an LDS round trip per k16 step.
  auto gemm1 = [&](const char* sb, auto XFIRST_T) -> mlp_f16v {
    constexpr bool XFIRST = decltype(XFIRST_T)::value;
    sp_h8 wh[C16], wl[C16];
#pragma unroll
    for (int s = 0; s < C16; ++s) {
      const char* const p = sb + (s >> 1) * 4096 + frag;
      wh[s] = *reinterpret_cast<const sp_h8*>(p + (((4 * (s & 1) + 2 * h + 0) ^ swr) << 4));
      wl[s] = *reinterpret_cast<const sp_h8*>(p + (((4 * (s & 1) + 2 * h + 1) ^ swr) << 4));
    }
    mlp_f16v acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int s = 0; s < C16; ++s) {
      if constexpr (XFIRST) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[s], wl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[s], wh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[s], wh[s], acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh[s], acc, 0, 0, 0);
      }
    }
    return acc;
  };

  // Both passes are software-pipelined over the hidden chunks: an iteration holds the matrix work of chunk j + 1
  // (GEMM1) next to the vector work of chunk j (activation, statistics or split) in one basic block, so every wave
  // feeds the matrix pipe and the vector ALU at once instead of in alternating bursts.
  if constexpr (PASS == 1) {
    // slots: W1(0), W1(1), ...; ring of 3, two slots in flight.  lane = hidden channel 32 j + r of rows 8 gq + 4 h + e
    constexpr int NR = 3;
    issue(0, 0);
    if (NS > 1) issue(1, 1);
    if (NS > 2) issue(2, 2);
    wait_but(NS > 2 ? 2 : NS - 1);
    __builtin_amdgcn_s_barrier();
    mlp_f16v acc = gemm1(ring, std::true_type{});
    const long unit = (long)(m0 >> 5) + wave;
    for (int j = 0; j < NCH; ++j) {
      wait_but(j + 2 < NS ? 1 : 0);
      __builtin_amdgcn_s_barrier();
      if (j + 3 < NS) issue(j + 3, j % NR);
      // (the last iteration repeats its own chunk and drops the result: one basic block per iteration)
      const mlp_f16v nxt = gemm1(ring + ((j + 1 < NCH ? j + 1 : j) % NR) * SLOT, std::true_type{});
      const float ws = *reinterpret_cast<const float*>(smem + (32 * j + r) * 4);
      const float bs = *reinterpret_cast<const float*>(smem + VB + (32 * j + r) * 4);
      float ssq = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float a = activate(__builtin_fmaf(acc[q], ws, bs));
        ssq = __builtin_fmaf(a, a, ssq);
      }
      ssq += __shfl_xor(ssq, 32);
      if (h == 0 && wave_active) g.part[unit * H4 + 32 * j + r] = ssq;
#pragma unroll
      for (int i = 0; i < 3 * C16; ++i) {  // one MFMA, then its share of the chunk's vector work
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 10, 0);
      }
      acc = nxt;
    }
  } else {
    // slots: W1(0), then W2(j), W1(j + 1) per chunk; ring of 6, four slots (two iterations) in flight
    constexpr int NR = 6;
#pragma unroll
    for (int sl = 0; sl < 5; ++sl)
      if (sl < NS) issue(sl, sl);
    mlp_f16v acc2[KB];
#pragma unroll
    for (int nb = 0; nb < KB; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[nb][q] = 0.f;
    // this lane's image among the tile's two: row of the staged multipliers
    const unsigned sel = (unsigned)((int)fdiv((uint32_t)mc, g.d_hw) - img0);
    const char* const ex = smem + h * 16;
    const char* const ex_s = ex + (2 + (sel > 1 ? 1 : sel)) * VB;
    static_assert(NS >= 5, "the prologue issues five slots");
    wait_but(4);
    __builtin_amdgcn_s_barrier();
    if (stamp) st1 = (long)__builtin_amdgcn_s_memtime();
    mlp_f16v acc = gemm1(ring, std::false_type{});
    for (int j = 0; j < NCH; ++j) {
      // slots 2 j + 1 (W2 of chunk j) and 2 j + 2 (W1 of chunk j + 1) have landed; 2 j + 3 and 2 j + 4 may still fly
      const long ta = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
      wait_but(j + 3 <= NCH ? 2 : (j + 2 <= NCH ? 1 : 0));
      __builtin_amdgcn_s_barrier();
      if (stamp) wa += (long)__builtin_amdgcn_s_memtime() - ta;
      if (2 * j + 5 < NS) issue(2 * j + 5, (2 * j + 5) % NR);
      if (2 * j + 6 < NS) issue(2 * j + 6, (2 * j + 6) % NR);
      // GEMM1 of the next chunk; the last iteration repeats its own chunk (slot 2 j is still in the ring) and drops the
      // result, so that the iteration stays one basic block the scheduler can interleave
      const mlp_f16v nxt = gemm1(ring + ((j + 1 < NCH ? 2 * j + 2 : 2 * j) % NR) * SLOT, std::false_type{});
      // the chunk's per-channel constants and the W2 fragments of both k16 steps, requested up front
      sp_f4 cw[4], cb[4], cs[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int eo = (32 * j + 8 * gq) * 4;
        cw[gq] = *reinterpret_cast<const sp_f4*>(ex + eo);
        cb[gq] = *reinterpret_cast<const sp_f4*>(ex + VB + eo);
        cs[gq] = *reinterpret_cast<const sp_f4*>(ex_s + eo);
      }
      const char* const sb2 = ring + ((2 * j + 1) % NR) * SLOT;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sp_h8 w2h[KB], w2l[KB];
#pragma unroll
        for (int nb = 0; nb < KB; ++nb) {
          const char* const p = sb2 + nb * 4096 + frag;
          w2h[nb] = *reinterpret_cast<const sp_h8*>(p + (((4 * t + 2 * h + 0) ^ swr) << 4));
          w2l[nb] = *reinterpret_cast<const sp_h8*>(p + (((4 * t + 2 * h + 1) ^ swr) << 4));
        }
        sp_f4 hv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int gq = 2 * t + u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = __builtin_fmaf(acc[4 * gq + e], cw[gq][e], cb[gq][e]);
            if constexpr (ACT == ACT_GELU) hv[u][e] = act_gelu(v) * cs[gq][e];
            else hv[u][e] = act_mish_scaled(v, cs[gq][e]);
          }
        }
        sp_h8 ah, al;
        sp8_split8_mix(hv[0], hv[1], ah, al);
#pragma unroll
        for (int nb = 0; nb < KB; ++nb) {
          acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[nb], ah, acc2[nb], 0, 0, 0);
          acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[nb], al, acc2[nb], 0, 0, 0);
          acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[nb], ah, acc2[nb], 0, 0, 0);
        }
      }
      // Issue order of the iteration: one MFMA, then the vector instructions that fit into its 32 cycles, 36 times over -
      // an in-order wave that issues two MFMAs back to back idles until the matrix pipe takes the second one.
#pragma unroll
      for (int i = 0; i < 12 * C16 / 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 7, 0);  // VALU, TRANS
      }
      acc = nxt;
    }
    const long st2 = stamp ? (long)__builtin_amdgcn_s_memtime() : 0;
    // ---- epilogue: lane (m, h) owns columns 32 nb + 8 gq + 4 h + 0..3 of its row.  Every residual quad is requested
    // before the first store (the compiler may not move a load of res above a store to Out: they could alias), so the
    // tile pays one HBM round trip, not one per quad; row scales and bias come out of the staged vectors.
    if (mrow < g.M) {
      const long ro = (long)mrow * C + 4 * h;
      sp_f4 rs[KB][4];
#pragma unroll
      for (int nb = 0; nb < KB; ++nb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = 32 * nb + 8 * gq;
          rs[nb][gq] = sp_f4{0.f, 0.f, 0.f, 0.f};
          if (n + 4 * h < C) rs[nb][gq] = *reinterpret_cast<const sp_f4*>(g.res + ro + n);  // C % 8 == 0: whole quads
        }
      const char* const ex2 = smem + NV * VB + h * 16;
      if (g.OutLn == nullptr) {
#pragma unroll
      for (int nb = 0; nb < KB; ++nb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = 32 * nb + 8 * gq;
          if (n + 4 * h < C) {
            const sp_f4 ws = *reinterpret_cast<const sp_f4*>(ex2 + n * 4);
            const sp_f4 bs = *reinterpret_cast<const sp_f4*>(ex2 + C * 4 + n * 4);
            sp_f4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(acc2[nb][4 * gq + e], ws[e], bs[e]) + rs[nb][gq][e];
            *reinterpret_cast<sp_f4*>(g.Out + ro + n) = o;
          }
        }
      } else {
        // LayerNorm over the row's C outputs: the row lives in two lanes (h = 0 / 1, 4-column quads alternating), each sums its
        // quads in ascending column order, the pair adds h = 0's part first; two-pass variance as in ln_rows_kernel.  (The
        // values stay where the accumulators were: rs is overwritten.)
        float s0 = 0.f;
#pragma unroll
        for (int nb = 0; nb < KB; ++nb)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * nb + 8 * gq;
            if (n + 4 * h < C) {
              const sp_f4 ws = *reinterpret_cast<const sp_f4*>(ex2 + n * 4);
              const sp_f4 bs = *reinterpret_cast<const sp_f4*>(ex2 + C * 4 + n * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) rs[nb][gq][e] = __builtin_fmaf(acc2[nb][4 * gq + e], ws[e], bs[e]) + rs[nb][gq][e];
              s0 += (rs[nb][gq][0] + rs[nb][gq][1]) + (rs[nb][gq][2] + rs[nb][gq][3]);
            }
          }
        const float so = __shfl_xor(s0, 32);
        const float mean = (h == 0 ? s0 + so : so + s0) / (float)C;
        float q0 = 0.f;
#pragma unroll
        for (int nb = 0; nb < KB; ++nb)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            if (32 * nb + 8 * gq + 4 * h < C) {
              const sp_f4 d = rs[nb][gq] - mean;
              q0 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        const float qo = __shfl_xor(q0, 32);
        const float rstd = 1.0f / sqrtf((h == 0 ? q0 + qo : qo + q0) / (float)C + g.ln_eps);
        char* const orow = g.OutLn + (long)mrow * (C * 4);
#pragma unroll
        for (int nb = 0; nb < KB; ++nb)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * nb + 8 * gq + 4 * h;
            if (n < C) {
              const sp_f4 wv = *reinterpret_cast<const sp_f4*>(g.ln_w + n), bv = *reinterpret_cast<const sp_f4*>(g.ln_b + n);
              const sp_f4 y = (rs[nb][gq] - mean) * rstd * wv + bv;
              sp_h4 hi, lo;
              sp8_split4(y, hi, lo);
              // chunk (n / 8) = [8 hi halves | 8 lo halves]: this lane holds positions 4 h .. 4 h + 3 of each
              char* const cb = orow + (n >> 3) * 32 + h * 8;
              *reinterpret_cast<sp_h4*>(cb) = hi;
              *reinterpret_cast<sp_h4*>(cb + 16) = lo;
            }
          }
      }
    }
    if (stamp && wave == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long st3 = (long)__builtin_amdgcn_s_memtime();
      if (lane == 0) {
        long* d = g.stamps + (long)blockIdx.x * 8;
        d[0] = st0, d[1] = st1, d[2] = st2, d[3] = st3, d[4] = wa, d[5] = wb;  // wb unused since the loop has one barrier
        d[6] = (long)__builtin_amdgcn_s_getreg((15 << 11) | 4);
        d[7] = (long)__builtin_amdgcn_s_memrealtime();
      }
    }
  }
}

// W2 [C][4C] f32 -> SP8 rows in the k order PASS 2 produces its operand in, with a power-of-two scale per row
// (row maximum in [2^13, 2^14), like sp8_pack_rows_kernel).  One wave per row.
// Stored chunk G = 4 j + 2 t + hh (8 values: hi 16 B | lo 16 B), position p: k = 32 j + 16 t + 8 (p >> 2) + 4 hh + (p & 3).
__global__ __launch_bounds__(256) void mlp_pack_w2p_kernel(const float* __restrict__ W2, sp_h8* __restrict__ out,
                                                          float* __restrict__ wscale, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = W2 + (long)row * K;
  float mx = 0.f;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(x[k]));
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
  int e = 0;
  if (mx > 0.f && mx < INFINITY) {
    int ex;
    (void)frexpf(mx, &ex);
    e = 14 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  const float sc = ldexpf(1.0f, e);
  if (lane == 0) wscale[row] = ldexpf(1.0f, -e);
  sp_h8* o = out + (long)row * (K / 4);
  for (int G = lane; G < K / 8; G += 64) {
    const int j = G >> 2, t = (G >> 1) & 1, hh = G & 1;
    const int k0 = 32 * j + 16 * t + 4 * hh;
    const sp_f4 a = *reinterpret_cast<const sp_f4*>(x + k0) * sc, b = *reinterpret_cast<const sp_f4*>(x + k0 + 8) * sc;
    sp_h8 hi, lo;
    sp8_split8(a, b, hi, lo);
    o[2 * G] = hi;
    o[2 * G + 1] = lo;
  }
}

}  // namespace mtgv
