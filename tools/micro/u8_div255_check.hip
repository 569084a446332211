// conv0_u8_kernel's division-free u / 255 against __fdiv_rn on the device, all 256 bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/u8_div255_check.hip -o tools/micro/build/u8_div255_check
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* bad) {
  const float u = (float)threadIdx.x;
  const float ref = __fdiv_rn(u, 255.0f);
  const float r255 = 1.0f / 255.0f;
  const float q0 = u * r255;
  const float q = __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, u), r255, q0);
  if (__float_as_uint(q) != __float_as_uint(ref)) atomicAdd(bad, 1);
}
int main() {
  int *d, h = -1;
  hipMalloc(&d, 4), hipMemset(d, 0, 4);
  k<<<1, 256>>>(d);
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("bytes whose quotient differs from __fdiv_rn: %d of 256\n", h);
  return h != 0;
}
