// Micro-benchmark 4: the SP8 split GEMM of sp_gemm.hip with an eight-wave "ping-pong" schedule.
// One block per CU, two waves per SIMD (one of each group); a wave alternates a MEMORY phase (DMA issue for a later
// stage + all ds_read_b128 of its k16 step) and a COMPUTE phase (nothing but the step's MFMAs) with a block barrier after
// each; group 1 starts one phase late, so on every SIMD one wave is always in its MFMA burst while the other fetches.
//   C[M,N] = A[M,K] * W[N,K]^T, both operands SP8, three v_mfma_f32_32x32x16_f16 per product.
// Stages are 16 k (64-byte rows), ring of NST; piece p of a stage (1 KiB = 16 rows) is issued by wave p % 8.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/sp_gemm_pp.hip -o tools/micro/build/sp_gemm_pp
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
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// WM x WN = 8 waves; wave tile (32 TM) x (32 TN); MODE 0: ping-pong, 1: same kernel without the stagger (both groups in phase)
template <int WM, int WN, int TM, int TN, int NST, int MODE, int STORE>
__global__ __launch_bounds__(512, 2) void sp_gemm_pp(const char* __restrict__ A, const char* __restrict__ B, float* __restrict__ O,
                                                      int M, int N, int K) {
  static_assert(WM * WN == 8, "eight waves");
  constexpr int NW = 8, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int RB = 64, RPP = 16, SPR = 4;
  constexpr int SA = BM * RB, SB = BN * RB, STG = SA + SB;
  constexpr int PA = BM / RPP, NP = (BM + BN) / RPP;
  constexpr int PPW = (NP + NW - 1) / NW;        // pieces per wave and stage (the last one may be missing for high waves)
  constexpr int REM = NP % NW;                   // waves < REM own PPW pieces, the others PPW - 1 (REM == 0: all PPW)
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                     // waves 0-3 / 4-7: one of each per SIMD
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
  const bool full = REM == 0 || wave < REM;      // this wave owns PPW pieces (else PPW - 1)

  const char* src[PPW];
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    int p = wave + NW * u;
    if (p >= NP) p = wave;                       // never issued
    const bool isA = p < PA;
    const int pp = isA ? p : p - PA;
    const int row = pp * RPP + lane / SPR;
    const int sw = (row >> 2) & 3;
    const int slot = (lane % SPR) ^ sw;
    src[u] = (isA ? A + (m0 + row) * rowb : B + (n0 + row) * rowb) + slot * 16;
  }
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int p = wave + NW * u;
      if (u < PPW - 1 || full)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[u] + (long)t * RB), (lptr_t)(smem + buf * STG + p * 1024), 16, 0, 0);
    }
  };
  // own pieces of all stages but the newest `keep` have landed
  auto wait_keep = [&](int keep) {  // keep in 0 .. NST-2
    if (full) {
      if (keep >= 3) wait_vm<3 * PPW>();
      else if (keep == 2) wait_vm<2 * PPW>();
      else if (keep == 1) wait_vm<PPW>();
      else wait_vm<0>();
    } else {
      if (keep >= 3) wait_vm<3 * (PPW - 1)>();
      else if (keep == 2) wait_vm<2 * (PPW - 1)>();
      else if (keep == 1) wait_vm<PPW - 1>();
      else wait_vm<0>();
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int sw = (r >> 2) & 3;
  const unsigned shi = (unsigned)(((h * 2 + 0) ^ sw) << 4), slo = (unsigned)(((h * 2 + 1) ^ sw) << 4);
  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (unsigned)((wm * TM * 32 + i * 32 + r) * RB);
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = (unsigned)(SA + (wn * TN * 32 + j * 32 + r) * RB);

  const int nk = K / 16;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);
  wait_keep(nk >= NST - 1 ? NST - 2 : (nk - 1 < 0 ? 0 : nk - 1));
  __builtin_amdgcn_s_barrier();                          // stage 0 is in LDS
  if (MODE == 0 && grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one phase behind

  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    // ---- memory phase: refill the buffer stage t-1 left, read this step's fragments, make sure stage t+1 is in
    {
      const int tn = t + NST - 1;
      int nb = buf - 1;
      if (nb < 0) nb += NST;
      if (tn < nk) issue(tn, nb);
    }
    const char* const sb = smem + buf * STG;
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
    {
      const int left = nk - 1 - (t + 1);         // stages after t+1 already issued: min(NST-2, left)
      wait_keep(left >= NST - 2 ? NST - 2 : (left < 0 ? 0 : left));
    }
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute phase
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
  if (MODE == 0 && grp == 0) __builtin_amdgcn_s_barrier();

  if (STORE == 0) {
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

template <int WM, int WN, int TM, int TN, int NST, int MODE, int STORE>
double run(const char* A, const char* B, float* O, int M, int N, int K, int iters) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int LDS = NST * (BM + BN) * 64;
  if (M % BM != 0 || N % BN != 0 || K % 16 != 0 || LDS > 160 * 1024) return 1e30;
  auto kern = sp_gemm_pp<WM, WN, TM, TN, NST, MODE, STORE>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  dim3 grid((N / BN) * (M / BM)), block(512);
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

template <int WM, int WN, int TM, int TN, int NST, int MODE, int STORE>
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
  run<WM, WN, TM, TN, NST, MODE, STORE>((const char*)sA, (const char*)sW, dO, M, N, K, 1);
  CK(hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)a[(size_t)m * K + k] * w[(size_t)n * K + k];
      const double e = fabs(o[(size_t)m * N + n] - s);
      if (!(e <= emax)) emax = e;
    }
  hipFree(dA); hipFree(dW); hipFree(dO); hipFree(sA); hipFree(sW);
  return emax;
}

//        WM WN TM TN NST MODE STORE
#define CFGS(X)             \
  X(4, 2, 2, 3, 4, 0, 1)    \
  X(4, 2, 2, 3, 4, 1, 1)    \
  X(4, 2, 2, 3, 5, 0, 1)    \
  X(4, 2, 2, 3, 4, 0, 0)    \
  X(4, 2, 2, 4, 4, 0, 1)    \
  X(4, 2, 2, 4, 4, 1, 1)    \
  X(4, 2, 2, 4, 4, 0, 0)    \
  X(2, 4, 2, 3, 4, 0, 1)    \
  X(4, 2, 2, 2, 4, 0, 1)    \
  X(4, 2, 2, 2, 5, 0, 1)    \
  X(4, 2, 1, 3, 4, 0, 1)    \
  X(4, 2, 1, 3, 4, 1, 1)

int main(int argc, char** argv) {
#define CHK(WM, WN, TM, TN, NST, MODE, STORE)                                                                        \
  if (STORE == 1)                                                                                                    \
    printf("accuracy %dx%d waves, tile %dx%d, %d stages, mode %d: max|C - fp64| = %.3e\n", WM, WN, 32 * TM * WM,     \
           32 * TN * WN, NST, MODE, check<WM, WN, TM, TN, NST, MODE, STORE>(32 * TM * WM * 2, 32 * TN * WN * 3, 416)); \
  fflush(stdout);
  CFGS(CHK)
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"s1.pw1", 98304, 768, 192}, {"s1.pw2", 98304, 192, 768}, {"s2.pw1", 24576, 1536, 384},
      {"s2.pw2", 24576, 384, 1536}, {"s3.pw1", 6144, 3072, 768}, {"s3.pw2", 6144, 768, 3072}, {"square", 8192, 6144, 4096}};
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
    printf("%-7s M=%6d N=%5d K=%4d | TF (us):", s.name, s.M, s.N, s.K);
#define RUN(WM, WN, TM, TN, NST, MODE, STORE)                                                                      \
  {                                                                                                                \
    const double t = run<WM, WN, TM, TN, NST, MODE, STORE>((const char*)sA, (const char*)sW, dO, s.M, s.N, s.K, 10); \
    printf(" [%dx%d s%d m%d st%d] %.0f (%.0f)", 32 * TM * WM, 32 * TN * WN, NST, MODE, STORE, t > 1e20 ? 0.0 : fl / t / 1e9, t > 1e20 ? 0.0 : t * 1e3); \
  }
    CFGS(RUN)
    printf("\n");
    fflush(stdout);
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(sA); hipFree(sW);
  }
  return 0;
}
