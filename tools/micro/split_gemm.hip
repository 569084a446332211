// Micro-benchmark: C[M,N] = A[M,K] * W[N,K]^T with f32 operands split into fp16 hi+lo planes and three
// v_mfma_f32_32x32x16_f16 per product (hi*hi + hi*lo + lo*hi).  Checks (1) that the matrix unit keeps fp16
// subnormals, (2) the error against fp64, (3) the achievable rate on the encoder's pointwise shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/split_gemm.hip -o /tmp/split_gemm && /tmp/split_gemm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__global__ void denorm_probe(float* out) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)9.5367431640625e-07f;  // 2^-20: an fp16 subnormal
    b[j] = (_Float16)1024.0f;
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];  // 16 * 2^-10 = 0.015625 when subnormals are kept
}

__device__ inline void split4(float4 v, h4& hi, h4& lo) {
  f32x4 x = {v.x, v.y, v.z, v.w};
  hi = __builtin_convertvector(x, h4);
  f32x4 r = x - __builtin_convertvector(hi, f32x4);
  lo = __builtin_convertvector(r, h4);
}

__global__ void presplit_kernel(const float* __restrict__ in, _Float16* __restrict__ hi, _Float16* __restrict__ lo, long n4) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  h4 a, b;
  split4(((const float4*)in)[i], a, b);
  ((h4*)hi)[i] = a;
  ((h4*)lo)[i] = b;
}

// 256 threads = 4 waves; block tile 128 x (32*TN); wave w owns rows 32w..32w+31 and all TN column blocks.
// BK = 16: one MFMA k-step per k-tile, LDS rows of 16 halves (32 B) are contiguous so fragment reads are conflict-free.
template <int TN, int PRE_W, int PRE_A, int OCC>
__global__ __launch_bounds__(256, OCC) void split_gemm(const float* __restrict__ A, const _Float16* __restrict__ Ah,
                                                       const _Float16* __restrict__ Al, const float* __restrict__ W,
                                                       const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl,
                                                       float* __restrict__ O, int M, int N, int K) {
  constexpr int BM = 128, BN = 32 * TN, BK = 16;
  constexpr int PB = (BN + 63) / 64;  // B load passes of 64 rows
  __shared__ __attribute__((aligned(16))) _Float16 sA[2][2][BM * BK];
  __shared__ __attribute__((aligned(16))) _Float16 sB[2][2][BN * BK];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
  const int ntn = (N + BN - 1) / BN, ntm = M / BM;
  int id = blockIdx.x;
  const int total = ntn * ntm;
  if (total % 8 == 0) id = (id & 7) * (total >> 3) + (id >> 3);  // blocks of one XCD walk neighbouring tiles
  const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
  const int lrow = t >> 2, kq = t & 3;

  float4 ra[2], rb[PB];
  h4 rah[2], ral[2], rbh[PB], rbl[PB];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long off = (long)(m0 + lrow + 64 * i) * K + k0 + 4 * kq;
      if (PRE_A) {
        rah[i] = *(const h4*)(Ah + off);
        ral[i] = *(const h4*)(Al + off);
      } else {
        ra[i] = *(const float4*)(A + off);
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int row = lrow + 64 * i;
      if ((BN % 64) && row >= BN) continue;
      int n = n0 + row;
      n = n < N ? n : N - 1;
      long off = (long)n * K + k0 + 4 * kq;
      if (PRE_W) {
        rbh[i] = *(const h4*)(Wh + off);
        rbl[i] = *(const h4*)(Wl + off);
      } else {
        rb[i] = *(const float4*)(W + off);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!PRE_A) split4(ra[i], rah[i], ral[i]);
      int o = (lrow + 64 * i) * BK + 4 * kq;
      *(h4*)&sA[buf][0][o] = rah[i];
      *(h4*)&sA[buf][1][o] = ral[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int row = lrow + 64 * i;
      if ((BN % 64) && row >= BN) continue;
      if (!PRE_W) split4(rb[i], rbh[i], rbl[i]);
      int o = row * BK + 4 * kq;
      *(h4*)&sB[buf][0][o] = rbh[i];
      *(h4*)&sB[buf][1][o] = rbl[i];
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  const int nk = K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    const h8 ah = *(const h8*)&sA[cur][0][(32 * wave + r) * BK + 8 * h];
    const h8 al = *(const h8*)&sA[cur][1][(32 * wave + r) * BK + 8 * h];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const h8 bh = *(const h8*)&sB[cur][0][(32 * j + r) * BK + 8 * h];
      const h8 bl = *(const h8*)&sB[cur][1][(32 * j + r) * BK + 8 * h];
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int col = n0 + 32 * j + r;
    if (col >= N) continue;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int row = m0 + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
      O[(long)row * N + col] = acc[j][q];
    }
  }
}

struct Shape {
  const char* name;
  int M, N, K;
};

template <int TN, int PRE_W, int PRE_A, int OCC>
double time_one(const float* A, const _Float16* Ah, const _Float16* Al, const float* W, const _Float16* Wh, const _Float16* Wl,
                float* O, int M, int N, int K, int iters) {
  int ntn = (N + 32 * TN - 1) / (32 * TN), ntm = M / 128;
  dim3 grid(ntn * ntm), block(256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) split_gemm<TN, PRE_W, PRE_A, OCC><<<grid, block>>>(A, Ah, Al, W, Wh, Wl, O, M, N, K);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) split_gemm<TN, PRE_W, PRE_A, OCC><<<grid, block>>>(A, Ah, Al, W, Wh, Wl, O, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters;
}

int main() {
  float* d1;
  CK(hipMalloc(&d1, 4));
  denorm_probe<<<1, 64>>>(d1);
  float v;
  CK(hipMemcpy(&v, d1, 4, hipMemcpyDeviceToHost));
  printf("denorm probe: %.9g (0.015625 = fp16 subnormals kept by the MFMA, 0 = flushed)\n", v);

  // ---- accuracy on a small problem ----
  {
    const int M = 256, N = 256, K = 384;
    std::vector<float> a((size_t)M * K), w((size_t)N * K), o((size_t)M * N);
    srand(1);
    for (auto& x : a) x = (rand() / (float)RAND_MAX * 2 - 1) * 3.0f;
    for (auto& x : w) x = (rand() / (float)RAND_MAX * 2 - 1) * 0.1f;
    float *dA, *dW, *dO;
    _Float16 *dWh, *dWl, *dAh, *dAl;
    CK(hipMalloc(&dA, a.size() * 4));
    CK(hipMalloc(&dW, w.size() * 4));
    CK(hipMalloc(&dO, o.size() * 4));
    CK(hipMalloc(&dWh, w.size() * 2));
    CK(hipMalloc(&dWl, w.size() * 2));
    CK(hipMalloc(&dAh, a.size() * 2));
    CK(hipMalloc(&dAl, a.size() * 2));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    presplit_kernel<<<(w.size() / 4 + 255) / 256, 256>>>(dW, dWh, dWl, w.size() / 4);
    presplit_kernel<<<(a.size() / 4 + 255) / 256, 256>>>(dA, dAh, dAl, a.size() / 4);
    for (int mode = 0; mode < 3; ++mode) {
      CK(hipMemset(dO, 0, o.size() * 4));
      if (mode == 0) split_gemm<4, 0, 0, 3><<<(N / 128) * (M / 128), 256>>>(dA, dAh, dAl, dW, dWh, dWl, dO, M, N, K);
      if (mode == 1) split_gemm<2, 1, 0, 4><<<(N / 64) * (M / 128), 256>>>(dA, dAh, dAl, dW, dWh, dWl, dO, M, N, K);
      if (mode == 2) split_gemm<4, 1, 1, 3><<<(N / 128) * (M / 128), 256>>>(dA, dAh, dAl, dW, dWh, dWl, dO, M, N, K);
      CK(hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost));
      double emax = 0, e32max = 0, ref_abs = 0;
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
          double s = 0;
          float s32 = 0;
          for (int k = 0; k < K; ++k) {
            s += (double)a[(size_t)m * K + k] * w[(size_t)n * K + k];
            s32 = fmaf(a[(size_t)m * K + k], w[(size_t)n * K + k], s32);
          }
          emax = fmax(emax, fabs(o[(size_t)m * N + n] - s));
          e32max = fmax(e32max, fabs(s32 - s));
          ref_abs = fmax(ref_abs, fabs(s));
        }
      printf("accuracy mode %d: max|split - fp64| = %.3e   (f32 fma chain: %.3e, max|ref| = %.2f)\n", mode, emax, e32max, ref_abs);
    }
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dWh); hipFree(dWl); hipFree(dAh); hipFree(dAl);
  }

  // ---- rate on the encoder's pointwise shapes (AE-tiny, 256 crops) and the bank GEMM ----
  Shape shapes[] = {
      {"s0.pw1", 393216, 384, 96},   {"s0.pw2", 393216, 96, 384},   {"s1.pw1", 98304, 768, 192}, {"s1.pw2", 98304, 192, 768},
      {"s2.pw1", 24576, 1536, 384},  {"s2.pw2", 24576, 384, 1536},  {"s3.pw1", 6144, 3072, 768}, {"s3.pw2", 6144, 768, 3072},
      {"bank", 256, 100000, 768},    {"square", 8192, 8192, 4096},
  };
  for (auto& s : shapes) {
    size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, no = (size_t)s.M * s.N;
    float *dA, *dW, *dO;
    _Float16 *dWh, *dWl, *dAh, *dAl;
    CK(hipMalloc(&dA, na * 4));
    CK(hipMalloc(&dW, nw * 4));
    CK(hipMalloc(&dO, no * 4));
    CK(hipMalloc(&dWh, nw * 2));
    CK(hipMalloc(&dWl, nw * 2));
    CK(hipMalloc(&dAh, na * 2));
    CK(hipMalloc(&dAl, na * 2));
    std::vector<float> ha(na), hw(nw);
    for (size_t i = 0; i < na; ++i) ha[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < nw; ++i) hw[i] = ((i * 40503u + 7) >> 4 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hw.data(), nw * 4, hipMemcpyHostToDevice));
    presplit_kernel<<<(nw / 4 + 255) / 256, 256>>>(dW, dWh, dWl, nw / 4);
    presplit_kernel<<<(na / 4 + 255) / 256, 256>>>(dA, dAh, dAl, na / 4);
    double fl = 2.0 * s.M * s.N * s.K;
    double bytes = 4.0 * (na + nw + no);
    int it = 10;
    double t[8];
    t[0] = time_one<2, 0, 0, 4>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[1] = time_one<4, 0, 0, 3>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[2] = time_one<2, 1, 0, 4>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[3] = time_one<4, 1, 0, 3>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[4] = time_one<2, 1, 1, 4>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[5] = time_one<4, 1, 1, 3>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[6] = time_one<3, 1, 0, 3>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    t[7] = time_one<1, 1, 0, 4>(dA, dAh, dAl, dW, dWh, dWl, dO, s.M, s.N, s.K, it);
    printf("%-7s M=%6d N=%6d K=%4d  min-traffic %.2f GB | TF(f32-equivalent): tn2 %.0f  tn4 %.0f | W pre-split: tn1 %.0f tn2 %.0f tn3 %.0f tn4 %.0f | W+A pre-split: tn2 %.0f tn4 %.0f | best %.3f ms = %.2f TB/s\n",
           s.name, s.M, s.N, s.K, bytes / 1e9, fl / t[0] / 1e9, fl / t[1] / 1e9, fl / t[7] / 1e9, fl / t[2] / 1e9, fl / t[6] / 1e9,
           fl / t[3] / 1e9, fl / t[4] / 1e9, fl / t[5] / 1e9, fmin(fmin(t[2], t[3]), fmin(t[4], t[5])),
           bytes / fmin(fmin(t[2], t[3]), fmin(t[4], t[5])) / 1e9);
    fflush(stdout);
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dWh); hipFree(dWl); hipFree(dAh); hipFree(dAl);
  }
  return 0;
}
