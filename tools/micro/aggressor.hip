// Debug aid: synthetic "aggressor" kernels to co-run beside a library kernel on another stream.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/aggressor.hip -o /tmp/libaggr.so
// mode 0: dense independent v_mfma_f32_32x32x16_f16 chains, no memory traffic
// mode 1: dense independent v_mfma_f32_32x32x2_f32 chains
// mode 2: mode 0 + ds_read_b128 traffic from a 24 KiB LDS tile each iteration
// mode 3: LDS traffic only (no MFMA)
// mode 4: one dependent f16 MFMA chain (the tn = 1 pattern)
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void aggr_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x;
  if (MODE == 2 || MODE == 3) {
    for (int i = lane; i < 6144; i += 256) sm[i] = (float)i * 1e-3f;
    __syncthreads();
  }
  h8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.001f * (lane + j)), b[j] = (_Float16)(0.002f * (lane - j));
  f32x16 c0 = {0}, c1 = {0}, c2 = {0};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    }
    if (MODE == 1) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[1], (float)b[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[2], (float)b[2], c2, 0, 0, 0);
    }
    if (MODE == 4) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    }
    if (MODE == 2 || MODE == 3) {
      acc += *reinterpret_cast<const f32x4*>(&sm[((lane + it) & 1535) * 4]);
    }
  }
  float s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q];
  if (s == 123.456f) out[0] = s;  // keep the work alive
}

extern "C" int aggr_launch(int mode, int blocks, int iters, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (mode == 2 || mode == 3) ? 24576 : 0;
  switch (mode) {
    case 0: hipLaunchKernelGGL(aggr_kernel<0>, dim3(blocks), dim3(256), lds, s, out, iters); break;
    case 1: hipLaunchKernelGGL(aggr_kernel<1>, dim3(blocks), dim3(256), lds, s, out, iters); break;
    case 2: hipLaunchKernelGGL(aggr_kernel<2>, dim3(blocks), dim3(256), lds, s, out, iters); break;
    case 3: hipLaunchKernelGGL(aggr_kernel<3>, dim3(blocks), dim3(256), lds, s, out, iters); break;
    case 4: hipLaunchKernelGGL(aggr_kernel<4>, dim3(blocks), dim3(256), lds, s, out, iters); break;
    default: return 1;
  }
  return (int)hipGetLastError();
}
