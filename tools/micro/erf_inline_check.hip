// act.h: erf_inline vs the device library erff, bit for bit on 4M inputs (hipcc --offload-arch=gfx950 -O3 -w -Imtg-vision_amd/csrc tools/micro/erf_inline_check.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include "act.h"
__global__ void k(const float* x, float* a, float* b, int n) { int i = blockIdx.x*256+threadIdx.x; if (i<n) { a[i] = erff(x[i]); b[i] = mtgv::erf_inline(x[i]); } }
int main(){ const int n=1<<22; float *x,*a,*b; hipMallocManaged(&x,n*4); hipMallocManaged(&a,n*4); hipMallocManaged(&b,n*4);
 for(int i=0;i<n;i++){ x[i] = (i<n/2)? -6.f + 12.f*i/(n/2) : ldexpf((float)(rand()%1000-500)/500.f, rand()%40-30); }
 x[0]=0.f; x[1]=-0.f; x[2]=INFINITY; x[3]=-INFINITY; x[4]=1.0f; x[5]=-1.0f; x[6]=NAN;
 k<<<n/256,256>>>(x,a,b,n); hipDeviceSynchronize(); int bad=0; for(int i=0;i<n;i++){ if (memcmp(&a[i],&b[i],4)!=0 && !(isnan(a[i])&&isnan(b[i]))) { if(bad<5) printf("x=%g erff=%.9g inline=%.9g\n",x[i],a[i],b[i]); bad++; } } printf("mismatches %d of %d\n",bad,n); return 0; }
