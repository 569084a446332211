// Streaming HBM rates on MI355X: pure read, pure write, copy, at several footprints (tuning aid for roofline notes).
// hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_rw.hip -o tools/micro/build/hbm_rw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ src, f4* __restrict__ dst, long n, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (MODE == 0) acc += src[i];
    if (MODE == 1) dst[i] = f4{(float)i, 1.f, 2.f, 3.f};
    if (MODE == 2) dst[i] = src[i];
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

int main() {
  const long maxb = 1536l << 20;
  f4 *a, *b; float* sink;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long mb : {64l, 151l, 302l, 604l, 1208l}) {
    const long n = (mb << 20) / 16;
    for (int grid : {2048, 8192}) {
      float ms[3];
      for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          for (int it = 0; it < 5; ++it) {
            if (mode == 0) stream<0><<<grid, 256>>>(a, b, n, sink);
            if (mode == 1) stream<1><<<grid, 256>>>(a, b, n, sink);
            if (mode == 2) stream<2><<<grid, 256>>>(a, b, n, sink);
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
        ms[mode] /= 5;
      }
      const double gb = (double)(mb << 20) / 1e9;
      printf("%5ld MB grid %5d: read %.2f TB/s  write %.2f TB/s  copy %.2f TB/s (read+write bytes)\n", mb, grid, gb / ms[0], gb / ms[1], 2 * gb / ms[2]);
    }
  }
  return 0;
}
