// SP8: the split-precision operand format of the LDS-DMA GEMM (gemm_sp_kernel.h).
//
// A row of K f32 values (K % 8 == 0) is stored as K/8 chunks of 32 bytes: the 8 fp16 "hi" halves (16 B) followed by
// the 8 fp16 "lo" halves (16 B), hi = fp16(x), lo = fp16(x - hi), both round-to-nearest.  Same bytes per element as
// f32, same row pitch, so a tensor keeps its shape and strides; channel offsets must be multiples of 8.
// A 16-byte piece of a chunk is exactly one v_mfma_f32_32x32x16_f16 operand fragment (8 consecutive k), so the
// GEMM moves pieces HBM -> LDS -> MFMA register with no arithmetic in between.
//
// x = hi + lo + O(2^-22 |x|) holds while lo is a normal fp16, i.e. for |x| >= 2^-3; below that lo is subnormal and the
// error floor is 2^-25 absolute.  Activations handed to a GEMM are O(1) per row (LayerNorm outputs, activations,
// pixels), so that floor is below f32 rounding of the row's dot products.  Weights can be uniformly small (a layer
// whose output feeds a LayerNorm is scale-free), so constant B operands are stored with a power-of-two scale per row
// (sp8_pack_rows_kernel): w * 2^e with the row maximum in [2^13, 2^14), undone exactly in the f32 accumulator
// (wscale[n] = 2^-e).  Values beyond +-65504 * 2^-e saturate to inf: visible, never silently wrong.
#pragma once
#include "common.h"

namespace mtgv {

typedef _Float16 sp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp_h4 __attribute__((ext_vector_type(4)));
typedef float sp_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sp8_split4(const sp_f4 x, sp_h4& hi, sp_h4& lo) {
  hi = __builtin_convertvector(x, sp_h4);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, sp_f4), sp_h4);
}

// The same split with the remainder formed by v_fma_mix_f32, which reads an fp16 operand in place: x - float(hi) is one
// instruction per element instead of a conversion and a subtraction (the exact same value: both are exact in f32).
// For VALU-bound producers (mlp_fused_kernel.h).
__device__ __forceinline__ void sp8_split4_mix(const sp_f4 x, sp_h4& hi, sp_h4& lo) {
  hi = __builtin_convertvector(x, sp_h4);
  typedef int i2v __attribute__((ext_vector_type(2)));
  const i2v hb = __builtin_bit_cast(i2v, hi);  // hb[0] = {hi[0], hi[1]}, hb[1] = {hi[2], hi[3]}
  sp_f4 r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hb[0]), "v"(x[0]));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hb[0]), "v"(x[1]));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[2]) : "v"(hb[1]), "v"(x[2]));
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[3]) : "v"(hb[1]), "v"(x[3]));
  lo = __builtin_convertvector(r, sp_h4);
}
__device__ __forceinline__ void sp8_split8_mix(const sp_f4 x0, const sp_f4 x1, sp_h8& hi, sp_h8& lo) {
  sp_h4 h0, l0, h1, l1;
  sp8_split4_mix(x0, h0, l0);
  sp8_split4_mix(x1, h1, l1);
  hi = sp_h8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  lo = sp_h8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

__device__ __forceinline__ void sp8_split8(const sp_f4 x0, const sp_f4 x1, sp_h8& hi, sp_h8& lo) {
  sp_h4 h0, l0, h1, l1;
  sp8_split4(x0, h0, l0);
  sp8_split4(x1, h1, l1);
  hi = sp_h8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  lo = sp_h8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

// LDS-DMA in the saddr form: 16 bytes per lane from (uniform 64-bit base in an SGPR pair) + (32-bit lane offset) to
// (wave-uniform LDS address, through M0) + lane * 16.  hipcc does not select this form for
// __builtin_amdgcn_global_load_lds once its loop passes have folded the base into a per-lane 64-bit pointer; with it a
// stage's pieces cost no vector arithmetic (the stage advance is two scalar adds) and the instruction carries half the
// address bytes.  The compiler does not see the instruction: callers order it by their own s_waitcnt vmcnt(N).
// Two things hipcc does for its own builtin and the assembler does not do inside inline asm: the wait state gfx950
// needs between an SALU write of M0 and the LDS-DMA that reads it (s_nop 0), and making the LDS address wave-uniform
// (readfirstlane here, so that the "s" constraint never receives a value the compiler only believes to be uniform).
__device__ __forceinline__ void sp_dma16_saddr(const char* base, uint32_t off, const char* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)lds);
#if defined(MTGV_SP_EXP) && MTGV_SP_EXP == 5
  // TIMING EXPERIMENT ONLY: the same instruction with a quarter of its lanes - issue cost without the bytes
  asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
#elif defined(MTGV_SP_EXP) && MTGV_SP_EXP == 10
  // EXPERIMENT (results unchanged): every piece twice - twice the instructions, twice the bytes, the same data
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
#elif defined(MTGV_SP_EXP) && MTGV_SP_EXP == 11
  // EXPERIMENT (results unchanged): every piece followed by a copy of itself with a quarter of its lanes - twice the instructions, 1.25 x the bytes
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
#endif
}

// A thread that holds 4 consecutive channels (channel quad c4 of a pixel) turns them into its half of an SP8 chunk
// and trades halves with the neighbouring lane (c4 ^ 1, the other quad of the same chunk), so every lane ends up with
// one whole 16-byte piece: even quads the chunk's hi piece, odd quads its lo piece.  Returns the piece; the caller
// stores it at chunk_base + (c4 & 1) * 16.  All lanes of the quad pair must call it together.
__device__ __forceinline__ sp_h8 sp8_piece_from_quad(const sp_f4 x, int c4) {
  sp_h4 hi, lo;
  sp8_split4(x, hi, lo);
  const bool odd = c4 & 1;
  typedef int i2 __attribute__((ext_vector_type(2)));
  const i2 hi_b = __builtin_bit_cast(i2, hi), lo_b = __builtin_bit_cast(i2, lo);
  const i2 send = odd ? hi_b : lo_b;  // the odd quad's hi goes to the even lane, the even quad's lo to the odd lane
  i2 recv;
  recv[0] = __builtin_amdgcn_mov_dpp(send[0], 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
  recv[1] = __builtin_amdgcn_mov_dpp(send[1], 0xB1, 0xF, 0xF, false);
  typedef int i4 __attribute__((ext_vector_type(4)));
  const i4 out = odd ? i4{recv[0], recv[1], lo_b[0], lo_b[1]} : i4{hi_b[0], hi_b[1], recv[0], recv[1]};
  return __builtin_bit_cast(sp_h8, out);
}

}  // namespace mtgv
