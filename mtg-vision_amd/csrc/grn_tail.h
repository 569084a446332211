// GRN statistics -> multipliers (convnextv2.py:171-174), shared by the stand-alone grn_finalize_kernel and by the tail of the
// kernels that produce the partial sums (gemm_sp_kernel EPI 8, mlp_fused_kernel pass 1):
//   scale[img][n] = gamma[n] * Gx[n] / (mean_n Gx + 1e-6) + 1,   Gx[n] = sqrt(sum over the image's units of part[unit][seg][n])
//
// The tail ("last block finalizes"): a block that has written all its partial sums fences, counts itself in on every image
// its rows touch, and the block that completes an image's count turns that image's partials into multipliers right there -
// the separate launch between pwconv1 and pwconv2 (18 per encoder pass, 8-10 us each plus its launch gap) disappears.  The
// arithmetic is the stand-alone kernel's (one device function, fixed summation order), so which block finalizes an image does
// not matter; counters return to zero by themselves.
#pragma once
#include "common.h"

namespace mtgv {

struct GrnFin {
  const float* part = nullptr;  // [units][segmax][N], unit = unit_rows consecutive rows
  int unit_rows = 1, segmax = 1, hw = 1, N = 0;
  FastDiv d_hw, d_unit;
  const float* gamma = nullptr;  // [N]
  float* scale = nullptr;        // [n_img][N]
};
struct GrnTail {
  GrnFin fin;
  int* cnt = nullptr;  // [n_img] arrival counters, zero between launches; nullptr: no tail (the caller launches grn_finalize)
};

// One image.  Every thread of the block calls it (barriers inside); threads 0..255 do the work.  sm: N + 256 floats of LDS.
// COHERENT: the partials were written by other workgroups of the SAME launch - read them past the vector L1.
template <bool COHERENT>
__device__ __forceinline__ void grn_finalize_image(const GrnFin& f, int img, float* sm, int tid) {
  float* gx = sm;           // [N]
  float* red = sm + f.N;    // [256]
  const int t_first = (int)fdiv((uint32_t)(img * f.hw), f.d_unit);
  const int t_last = (int)fdiv((uint32_t)((img + 1) * f.hw - 1), f.d_unit);
  float local = 0.f;
  if (tid < 256)
    for (int n = tid; n < f.N; n += 256) {
      float sum = 0.f;
      for (int t0 = t_first; t0 <= t_last; t0 += 8) {  // 8 loads in flight, added in unit order (same sum as one by one)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = t0 + u;
          const int seg = img - (int)fdiv((uint32_t)(t * f.unit_rows), f.d_hw);
          const float* p = f.part + ((long)t * f.segmax + seg) * f.N + n;
          v[u] = 0.f;
          if (t <= t_last) v[u] = COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
      }
      const float gval = sqrtf(sum);
      gx[n] = gval;
      local += gval;
    }
  if (tid < 256) red[tid] = local;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const float denom = red[0] / (float)f.N + 1e-6f;
  if (tid < 256)
    for (int n = tid; n < f.N; n += 256) f.scale[(long)img * f.N + n] = f.gamma[n] * (gx[n] / denom) + 1.0f;
  __syncthreads();  // sm may be reused for the next image
}

// Called by every thread still running in the block, after the block's last partial-sum store.  Rows [m_lo, m_hi] are the
// tile's (m_hi clamped to M - 1); bm = rows per tile, tiles_n = column tiles per row tile.  sm: N + 256 floats of LDS nobody
// else uses any more.
__device__ __forceinline__ void grn_tail(const GrnTail& t, int m_lo, int m_hi, int bm, int tiles_n, float* sm) {
  __shared__ int s_todo[12];
  __shared__ int s_ntodo;
  __threadfence();   // this thread's partial sums are visible device-wide before the block counts itself in
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid == 0) {
    int nt = 0;
    const int img_lo = (int)fdiv((uint32_t)m_lo, t.fin.d_hw), img_hi = (int)fdiv((uint32_t)m_hi, t.fin.d_hw);
    for (int img = img_lo; img <= img_hi && nt < 12; ++img) {
      const int r_first = (img * t.fin.hw) / bm, r_last = ((img + 1) * t.fin.hw - 1) / bm;
      const int expected = (r_last - r_first + 1) * tiles_n;
      if (atomicAdd(&t.cnt[img], 1) == expected - 1) {
        t.cnt[img] = 0;  // every tile of the image has arrived: nobody touches the counter again in this launch
        s_todo[nt++] = img;
      }
    }
    s_ntodo = nt;
  }
  __syncthreads();
  const int ntodo = s_ntodo;
  if (ntodo > 0) __threadfence();
  for (int i = 0; i < ntodo; ++i) grn_finalize_image<true>(t.fin, s_todo[i], sm, tid);
}

}  // namespace mtgv
