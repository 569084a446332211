// Sustained rate of v_mfma_f32_32x32x16_f16 with operands in registers (no memory traffic):
// CH independent accumulator chains per wave, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 ... && run
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, bool DEP3>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  h8 a, b, a2;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.001f * (threadIdx.x + j)), b[j] = (_Float16)(0.002f * (threadIdx.x - j)), a2[j] = (_Float16)(0.003f * j);
  f32x16 c[CH];
  for (int i = 0; i < CH; ++i)
    for (int q = 0; q < 16; ++q) c[i][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
      if (DEP3) {  // the split GEMM's pattern: three dependent MFMAs per accumulator
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, c[i], 0, 0, 0);
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < CH; ++i)
    for (int q = 0; q < 16; ++q) s += c[i][q];
  if (s == 123.456f) out[0] = s;
}

template <int CH, bool DEP3>
void run(const char* name, int blocks_per_cu) {
  float* d;
  hipMalloc(&d, 64);
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<CH, DEP3><<<blocks, 256>>>(d, iters);
  hipEventRecord(e0);
  k<CH, DEP3><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * iters * CH * (DEP3 ? 3 : 1);
  printf("%-34s waves/SIMD %d: %.0f TFLOP/s issued\n", name, blocks_per_cu, mfmas * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<1, false>("1 chain", w);
    run<4, false>("4 independent chains", w);
    run<4, true>("4 chains x 3 dependent (split GEMM)", w);
  }
  return 0;
}
