// Micro-benchmark 2: split-fp16 GEMM where the A operand never touches LDS.
//   C[M,N] = A[M,K] * W[N,K]^T, A fp32 in memory, W given pre-split (per 4 floats: 4 fp16 hi + 4 fp16 lo, 16 bytes).
// Each of the 4 waves of a block owns 32 rows; lane (r, h) loads its own MFMA fragment A[row r][k = 8h..8h+7] straight
// from global memory (32 contiguous bytes), splits it in registers and feeds the MFMAs.  Only B goes through LDS, in
// K chunks of BKB (one barrier per BKB/16 MFMA steps).  A fragments are prefetched PF steps ahead in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/split_gemm2.hip -o /tmp/split_gemm2 && /tmp/split_gemm2
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__device__ inline void split4(f32x4 x, h4& hi, h4& lo) {
  hi = __builtin_convertvector(x, h4);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), h4);
}

__global__ void presplit_kernel(const float* __restrict__ in, float* __restrict__ out, long n4) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  h4 a, b;
  split4(((const f32x4*)in)[i], a, b);
  h8 o;
  for (int j = 0; j < 4; ++j) o[j] = a[j], o[4 + j] = b[j];
  ((h8*)out)[i] = o;
}

// block: 256 threads, 128 rows x (32*TN) columns.  BKB = K chunk of B per barrier (multiple of 16).
template <int TN, int BKB, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_areg(const float* __restrict__ A, const float* __restrict__ Ws,
                                                      float* __restrict__ O, int M, int N, int K) {
  constexpr int BN = 32 * TN, STEPS = BKB / 16;
  constexpr int LSH = BKB + 8;  // halves per LDS row (+16 B pad)
  __shared__ __attribute__((aligned(16))) _Float16 sB[2][2][BN * LSH];  // [buf][plane][row][k]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
  const int ntn = N / BN;
  int id = blockIdx.x;
  const int total = gridDim.x;
  if (total % 8 == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int m0 = (id / ntn) * 128, n0 = (id % ntn) * BN;

  const float* arow = A + (size_t)(m0 + 32 * wave + r) * K + 8 * h;  // this lane's fragment stream
  // B staging: BN rows x BKB k = BN*BKB/4 groups of 16 bytes, 256 threads
  constexpr int GPR = BKB / 4;                 // 16-byte groups per row
  constexpr int NGC = (BN * GPR + 255) / 256;
  f32x4 rbv[NGC];
  auto loadB = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NGC; ++i) {
      const int gidx = t + i * 256, row = gidx / GPR, gq = gidx % GPR;
      if (gidx < BN * GPR) rbv[i] = *(const f32x4*)(Ws + (size_t)(n0 + row) * K + kc * BKB + gq * 4);
    }
  };
  auto storeB = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NGC; ++i) {
      const int gidx = t + i * 256, row = gidx / GPR, gq = gidx % GPR;
      if (gidx >= BN * GPR) continue;
      const h8 hl = __builtin_bit_cast(h8, rbv[i]);
      *(h4*)&sB[buf][0][row * LSH + gq * 4] = h4{hl[0], hl[1], hl[2], hl[3]};
      *(h4*)&sB[buf][1][row * LSH + gq * 4] = h4{hl[4], hl[5], hl[6], hl[7]};
    }
  };
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  const int nkc = K / BKB;
  // A prefetch: one whole chunk (STEPS fragments) ahead
  f32x4 a0[STEPS], a1[STEPS];
  auto loadA = [&](int kc) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      a0[s] = *(const f32x4*)(arow + kc * BKB + s * 16);
      a1[s] = *(const f32x4*)(arow + kc * BKB + s * 16 + 4);
    }
  };
  loadB(0);
  loadA(0);
  storeB(0);
  __syncthreads();
  for (int kc = 0; kc < nkc; ++kc) {
    const int cur = kc & 1;
    h8 ah[STEPS], al[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {  // split this chunk's A fragments (already in registers)
      h4 h0, l0, h1, l1;
      split4(a0[s], h0, l0);
      split4(a1[s], h1, l1);
      ah[s] = h8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      al[s] = h8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    if (kc + 1 < nkc) {
      loadB(kc + 1);
      loadA(kc + 1);
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const h8 bh = *(const h8*)&sB[cur][0][(32 * j + r) * LSH + s * 16 + 8 * h];
        const h8 bl = *(const h8*)&sB[cur][1][(32 * j + r) * LSH + s * 16 + 8 * h];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc[j], 0, 0, 0);
      }
    }
    if (kc + 1 < nkc) storeB(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = m0 + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
      O[(size_t)row * N + n0 + 32 * j + r] = acc[j][q];
    }
}

template <int TN, int BKB, int OCC>
double run(const float* A, const float* Ws, float* O, int M, int N, int K, int iters) {
  if (N % (32 * TN) != 0 || K % BKB != 0) return 1e30;
  dim3 grid((N / (32 * TN)) * (M / 128)), block(256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) gemm_areg<TN, BKB, OCC><<<grid, block>>>(A, Ws, O, M, N, K);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) gemm_areg<TN, BKB, OCC><<<grid, block>>>(A, Ws, O, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters;
}

int main() {
  {  // accuracy
    const int M = 256, N = 256, K = 384;
    std::vector<float> a((size_t)M * K), w((size_t)N * K), o((size_t)M * N);
    srand(1);
    for (auto& x : a) x = (rand() / (float)RAND_MAX * 2 - 1) * 3.0f;
    for (auto& x : w) x = (rand() / (float)RAND_MAX * 2 - 1) * 0.1f;
    float *dA, *dW, *dWs, *dO;
    CK(hipMalloc(&dA, a.size() * 4)); CK(hipMalloc(&dW, w.size() * 4)); CK(hipMalloc(&dWs, w.size() * 4)); CK(hipMalloc(&dO, o.size() * 4));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    presplit_kernel<<<(w.size() / 4 + 255) / 256, 256>>>(dW, dWs, w.size() / 4);
    gemm_areg<4, 64, 2><<<(N / 128) * (M / 128), 256>>>(dA, dWs, dO, M, N, K);
    CK(hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)a[(size_t)m * K + k] * w[(size_t)n * K + k];
        emax = fmax(emax, fabs(o[(size_t)m * N + n] - s));
      }
    printf("accuracy: max|C - fp64| = %.3e\n", emax);
  }
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"s0.pw1", 393216, 384, 96}, {"s0.pw2", 393216, 96, 384}, {"s1.pw1", 98304, 768, 192}, {"s1.pw2", 98304, 192, 768}, {"s2.pw1", 24576, 1536, 384},
      {"s2.pw2", 24576, 384, 1536}, {"s3.pw1", 6144, 3072, 768}, {"s3.pw2", 6144, 768, 3072}, {"square", 8192, 8192, 4096}};
  for (auto& s : shapes) {
    size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, no = (size_t)s.M * s.N;
    float *dA, *dW, *dWs, *dO;
    CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&dWs, nw * 4)); CK(hipMalloc(&dO, no * 4));
    std::vector<float> ha(na), hw(nw);
    for (size_t i = 0; i < na; ++i) ha[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < nw; ++i) hw[i] = ((i * 40503u + 7) >> 4 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hw.data(), nw * 4, hipMemcpyHostToDevice));
    presplit_kernel<<<(nw / 4 + 255) / 256, 256>>>(dW, dWs, nw / 4);
    const double fl = 2.0 * s.M * s.N * s.K;
    double t[8];
    t[0] = run<4, 32, 3>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[1] = run<4, 16, 3>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[2] = run<3, 16, 4>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[3] = run<3, 32, 4>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[4] = run<3, 32, 3>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[5] = run<5, 32, 2>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[6] = run<6, 32, 2>(dA, dWs, dO, s.M, s.N, s.K, 10);
    t[7] = run<4, 64, 2>(dA, dWs, dO, s.M, s.N, s.K, 10);
    printf("%-7s M=%6d N=%5d K=%4d | TF: tn4/32/o3 %.0f  tn4/16/o3 %.0f  tn3/16/o4 %.0f  tn3/32/o4 %.0f  tn3/32/o3 %.0f  tn5/32/o2 %.0f  tn6/32/o2 %.0f  tn4/64/o2 %.0f\n", s.name, s.M,
           s.N, s.K, fl / t[0] / 1e9, fl / t[1] / 1e9, fl / t[2] / 1e9, fl / t[3] / 1e9, fl / t[4] / 1e9, fl / t[5] / 1e9, fl / t[6] / 1e9, fl / t[7] / 1e9);
    fflush(stdout);
    hipFree(dA); hipFree(dW); hipFree(dWs); hipFree(dO);
  }
  return 0;
}
