// Pure-MFMA microbenchmark: does the chip hold a different clock / rate on the two f32-input MFMA shapes?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shapes.hip -o gpurun_out/mfma_shapes ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(const float* __restrict__ in, float* __restrict__ out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int j = 0; j < 4; ++j) a[j] = in[threadIdx.x * 4 + j], b[j] = in[1024 + threadIdx.x * 4 + j];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + i) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(const float* __restrict__ in, float* __restrict__ out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int j = 0; j < 4; ++j) a[j] = in[threadIdx.x * 4 + j], b[j] = in[1024 + threadIdx.x * 4 + j];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[(j + i) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> double timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 2048 * 256 * 4 * 4);
  std::vector<float> h(4096); for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int blocks : {256, 512, 1024}) {
    // 32x32x2: 4096 flop per instr; 16x16x4: 2048 flop per instr; per wave
    double ms32 = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 5);
    double f32 = (double)blocks * 4 * iters * 16 * 4096.0;
    double ms16 = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters); }, 5);
    double f16 = (double)blocks * 4 * iters * 64 * 2048.0;
    printf("blocks %4d: 32x32x2 %.1f TF (%.2f ms)   16x16x4 %.1f TF (%.2f ms)\n", blocks, f32 / ms32 / 1e9, ms32, f16 / ms16 / 1e9, ms16);
  }
  return 0;
}
