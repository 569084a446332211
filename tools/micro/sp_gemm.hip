// Micro-benchmark 3: split-fp16 GEMM with BOTH operands pre-split in memory ("SP8": every aligned group of 8
// floats is stored as 8 fp16 hi (16 B) + 8 fp16 lo (16 B), same 32 bytes) and moved global -> LDS by LDS-DMA
// (global_load_lds_dwordx4), no VGPR staging, no conversion in the loop.
//   C[M,N] = A[M,K] * W[N,K]^T          (three v_mfma_f32_32x32x16_f16 per product: lo*hi + hi*lo + hi*hi)
// Block: WM x WN waves, wave tile (32*TM) x (32*TN), K consumed in stages of 32 (128-byte rows), NST-stage LDS ring,
// one raw s_barrier per stage, counted vmcnt so NST-2 stages stay in flight across the barrier.
// LDS image: [row][8 slots of 16 B], slot' = slot ^ ((row>>1)&7)  (conflict-free ds_read_b128; the swizzle is applied
// to the per-lane SOURCE address because the DMA destination is lane-linear).
// Output orientation: m on lanes (weights are the first MFMA operand), so a lane owns 4 consecutive n per register
// group: 16-byte stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/sp_gemm.hip -o /tmp/sp_gemm && /tmp/sp_gemm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
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

// f32 [rows][K] -> SP8 [rows][K/8][hi8 | lo8]
__global__ void presplit8_kernel(const float* __restrict__ in, h8* __restrict__ out, long n8) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n8) return;
  h8 hi, lo;
  for (int j = 0; j < 8; ++j) {
    const float x = in[i * 8 + j];
    hi[j] = (_Float16)x;
    lo[j] = (_Float16)(x - (float)hi[j]);
  }
  out[2 * i] = hi;
  out[2 * i + 1] = lo;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WM, int WN, int TM, int TN, int NST, int OCC, int KS, int ORI>
__global__ __launch_bounds__(64 * WM * WN, OCC) void sp_gemm(const char* __restrict__ A, const char* __restrict__ B,
                                                            float* __restrict__ O, int M, int N, int K) {
  constexpr int NW = WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int RB = 64 * KS;                      // bytes per staged row (16 k per 64 B)
  constexpr int RPP = 1024 / RB;                   // rows per 1-KiB piece
  constexpr int SPR = RB / 16;                     // 16-byte slots per row
  constexpr int SA = BM * RB, SB = BN * RB, STG = SA + SB;
  constexpr int PA = BM / RPP, NP = (BM + BN) / RPP;  // 1-KiB pieces per stage
  static_assert(NP % NW == 0, "pieces must divide over the waves");
  constexpr int PPW = NP / NW;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const int tiles_n = N / BN;
  int L;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, x = b & 7;
    L = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (b >> 3);
  }
  const int tile_n = L % tiles_n, tile_m = L / tiles_n;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const long rowb = (long)K * 4;

  // loader: piece p = wave + NW*u
  const char* src[PPW];
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    const int p = wave + NW * u;
    const bool isA = p < PA;
    const int pp = isA ? p : p - PA;
    const int row = pp * RPP + lane / SPR;
    const int sw = KS == 2 ? (row >> 1) & 7 : (row >> 2) & 3;
    const int slot = (lane % SPR) ^ sw;
    src[u] = (isA ? A + (m0 + row) * rowb : B + (n0 + row) * rowb) + slot * 16;
  }
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int p = wave + NW * u;
      __builtin_amdgcn_global_load_lds((gptr_t)(src[u] + (long)t * RB), (lptr_t)(smem + buf * STG + p * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int sw = KS == 2 ? (r >> 1) & 7 : (r >> 2) & 3;
  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (unsigned)((wm * TM * 32 + i * 32 + r) * RB);
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = (unsigned)(SA + (wn * TN * 32 + j * 32 + r) * RB);

  const int nk = K / (16 * KS);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);

  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    // stage t landed (this wave's pieces), then everyone's: barrier.  The barrier also says every wave has finished
    // reading buffer (t-1)%NST, which is the one refilled next.
    if (NST > 2 && t + NST - 2 < nk)
      wait_vm<(NST > 2 ? (NST - 2) * PPW : 0)>();
    else
      wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    {
      const int tn = t + NST - 1;
      int nb = buf + NST - 1;
      if (nb >= NST) nb -= NST;
      if (tn < nk) issue(tn, nb);
    }
    const char* const sb = smem + buf * STG;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned shi = (unsigned)(((ks * 4 + h * 2 + 0) ^ sw) << 4);
      const unsigned slo = (unsigned)(((ks * 4 + h * 2 + 1) ^ sw) << 4);
      h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const h8*>(sb + a_off[i] + shi);
        al[i] = *reinterpret_cast<const h8*>(sb + a_off[i] + slo);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const h8*>(sb + b_off[j] + shi);
        bl[j] = *reinterpret_cast<const h8*>(sb + b_off[j] + slo);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (ORI == 1) {  // n on lanes: activations are the first operand
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          }
        }
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
  }

  if (ORI == 2) {  // no stores: main-loop-only rate (accumulators kept live)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[i][j]));
#endif
      }
    return;
  }
  if (ORI == 1) {  // n on lanes: each half wave writes 128 contiguous bytes of one row per register
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const long m = m0 + wm * TM * 32 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          O[m * N + n0 + wn * TN * 32 + j * 32 + r] = acc[i][j][q];
        }
    return;
  }
  // epilogue: lane owns row m = ... + r; register group g holds n = 8g + 4h + 0..3
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = m0 + wm * TM * 32 + i * 32 + r;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long nb = n0 + wn * TN * 32 + j * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(O + m * N + nb + 8 * g) = v;
      }
    }
  }
}

template <int WM, int WN, int TM, int TN, int NST, int OCC, int KS, int ORI>
double run(const char* A, const char* B, float* O, int M, int N, int K, int iters) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int LDS = NST * (BM + BN) * 64 * KS;
  if (M % BM != 0 || N % BN != 0 || K % (16 * KS) != 0 || LDS > 160 * 1024) return 1e30;
  auto kern = sp_gemm<WM, WN, TM, TN, NST, OCC, KS, ORI>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  dim3 grid((N / BN) * (M / BM)), block(64 * WM * WN);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kern<<<grid, block, LDS>>>(A, B, O, M, N, K);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) kern<<<grid, block, LDS>>>(A, B, O, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters;
}

template <int WM, int WN, int TM, int TN, int NST, int OCC, int KS, int ORI>
double check(int M, int N, int K) {
  std::vector<float> a((size_t)M * K), w((size_t)N * K), o((size_t)M * N);
  srand(1);
  for (auto& x : a) x = (rand() / (float)RAND_MAX * 2 - 1) * 3.0f;
  for (auto& x : w) x = (rand() / (float)RAND_MAX * 2 - 1) * 0.1f;
  float *dA, *dW, *dO;
  h8 *sA, *sW;
  CK(hipMalloc(&dA, a.size() * 4)); CK(hipMalloc(&dW, w.size() * 4)); CK(hipMalloc(&dO, o.size() * 4));
  CK(hipMalloc(&sA, a.size() * 4)); CK(hipMalloc(&sW, w.size() * 4));
  CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  presplit8_kernel<<<(a.size() / 8 + 255) / 256, 256>>>(dA, sA, a.size() / 8);
  presplit8_kernel<<<(w.size() / 8 + 255) / 256, 256>>>(dW, sW, w.size() / 8);
  CK(hipMemset(dO, 0xff, o.size() * 4));
  run<WM, WN, TM, TN, NST, OCC, KS, ORI>((const char*)sA, (const char*)sW, dO, M, N, K, 1);
  CK(hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)a[(size_t)m * K + k] * w[(size_t)n * K + k];
      const double e = fabs(o[(size_t)m * N + n] - s);
      if (!(e <= emax)) emax = e;  // NaN-propagating
    }
  hipFree(dA); hipFree(dW); hipFree(dO); hipFree(sA); hipFree(sW);
  return emax;
}

#define CFGS(X)                \
  X(2, 2, 2, 2, 2, 2, 2, 0)    \
  X(2, 2, 2, 2, 2, 2, 2, 1)    \
  X(2, 2, 2, 2, 2, 2, 2, 2)    \
  X(2, 2, 2, 2, 2, 4, 1, 0)    \
  X(2, 2, 2, 2, 2, 4, 1, 1)    \
  X(2, 2, 2, 2, 2, 4, 1, 2)    \
  X(2, 2, 2, 2, 3, 3, 1, 1)    \
  X(2, 2, 2, 2, 4, 2, 1, 1)    \
  X(4, 1, 1, 3, 2, 4, 2, 1)    \
  X(4, 1, 1, 3, 2, 4, 2, 2)    \
  X(2, 2, 2, 3, 2, 2, 2, 1)    \
  X(2, 2, 2, 3, 2, 2, 2, 2)    \
  X(4, 2, 2, 2, 2, 2, 2, 1)    \
  X(4, 2, 2, 2, 2, 2, 2, 2)    \
  X(4, 2, 2, 4, 2, 1, 2, 1)    \
  X(4, 2, 2, 4, 2, 1, 2, 2)

int main(int argc, char** argv) {
#define CHK(WM, WN, TM, TN, NST, OCC, KS, ORI)                                                                       \
  if (ORI != 2)                                                                                                      \
    printf("accuracy %dx%d waves, tile %dx%d, %d stages of %d, ori %d: max|C - fp64| = %.3e\n", WM, WN, 32 * TM * WM, \
           32 * TN * WN, NST, 16 * KS, ORI, check<WM, WN, TM, TN, NST, OCC, KS, ORI>(32 * TM * WM * 2, 32 * TN * WN * 3, 416));
  CFGS(CHK)
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"s0.pw1", 393216, 384, 96}, {"s0.pw2", 393216, 96, 384}, {"s1.pw1", 98304, 768, 192}, {"s1.pw2", 98304, 192, 768}, {"s2.pw1", 24576, 1536, 384},
      {"s2.pw2", 24576, 384, 1536}, {"s3.pw1", 6144, 3072, 768}, {"s3.pw2", 6144, 768, 3072}, {"square", 8192, 8192, 4096}};
  for (auto& s : shapes) {
    size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, no = (size_t)s.M * s.N;
    float *dA, *dW, *dO;
    h8 *sA, *sW;
    CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&dO, no * 4));
    CK(hipMalloc(&sA, na * 4)); CK(hipMalloc(&sW, nw * 4));
    std::vector<float> ha(na), hw(nw);
    for (size_t i = 0; i < na; ++i) ha[i] = ((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < nw; ++i) hw[i] = ((i * 40503u + 7) >> 4 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hw.data(), nw * 4, hipMemcpyHostToDevice));
    presplit8_kernel<<<(na / 8 + 255) / 256, 256>>>(dA, sA, na / 8);
    presplit8_kernel<<<(nw / 8 + 255) / 256, 256>>>(dW, sW, nw / 8);
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("%-7s M=%6d N=%5d K=%4d | TF:", s.name, s.M, s.N, s.K);
#define RUN(WM, WN, TM, TN, NST, OCC, KS, ORI)                                                                      \
  {                                                                                                                 \
    const double t = run<WM, WN, TM, TN, NST, OCC, KS, ORI>((const char*)sA, (const char*)sW, dO, s.M, s.N, s.K, 10); \
    printf(" [%dx%d/%dx%d/s%dx%d/o%d] %.0f", WM, WN, 32 * TM * WM, 32 * TN * WN, NST, 16 * KS, ORI,                    \
           t > 1e20 ? 0.0 : fl / t / 1e9);                                                                          \
  }
    CFGS(RUN)
    printf("\n");
    fflush(stdout);
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(sA); hipFree(sW);
  }
  return 0;
}
