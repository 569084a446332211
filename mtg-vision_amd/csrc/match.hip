// Exact cosine top-k over the whole card bank (what Qdrant's HNSW index approximates;
// mtgvision/qdrant.py:76-95).  Bank rows are stored L2-normalised, so cosine = dot:
//   1. l2norm_rows_kernel        q -> q/|q|
//   2. gemm_f32 (EPI=1)          S tile = Q B^T on f32 MFMA, per-tile top-k in registers
//   3. topk_merge_kernel         tiles_n*k candidates per query -> top k (score desc, id asc)
#include "match.h"
#include "gemm_sp.h"
#include "rowops.h"
#include "sp8.h"

#include <stdlib.h>

namespace mtgv {

// One result slot.  out_scores == nullptr selects the exchange format of the sharded match (mtgv/dist.py): out_ids is
// then [slots][2] int64 = (id, the score's float32 bit pattern zero-extended) - one buffer, one all-gather.
__device__ __forceinline__ void store_result(long* __restrict__ out_ids, float* __restrict__ out_scores, long slot, long id, float score) {
  if (out_scores != nullptr) {
    out_scores[slot] = score;
    out_ids[slot] = id;
  } else {
    out_ids[2 * slot] = id;
    out_ids[2 * slot + 1] = (long)__float_as_uint(score);
  }
}

template <typename IdT>
__global__ __launch_bounds__(256) void topk_merge_kernel(float* __restrict__ cs, const IdT* __restrict__ ci, int ncand, int k,
                                                        long id_base, float thr, long* __restrict__ out_ids,
                                                        float* __restrict__ out_scores) {
  __shared__ float rs[256];
  __shared__ long ri[256];
  __shared__ int rp[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* s = cs + (long)b * ncand;
  const IdT* id = ci + (long)b * ncand;
  for (int kk = 0; kk < k; ++kk) {
    float bs = -INFINITY;
    long bi = 0x7fffffffffffffffL;
    int bp = -1;
    for (int p = tid; p < ncand; p += 256) {
      const float v = s[p];
      const long i = (long)id[p];
      if (i >= 0 && v > -INFINITY && (v > bs || (v == bs && i < bi))) bs = v, bi = i, bp = p;
    }
    rs[tid] = bs, ri[tid] = bi, rp[tid] = bp;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) {
        const float os = rs[tid + st];
        const long oi = ri[tid + st];
        const int op = rp[tid + st];
        if (op >= 0 && (rp[tid] < 0 || os > rs[tid] || (os == rs[tid] && oi < ri[tid]))) rs[tid] = os, ri[tid] = oi, rp[tid] = op;
      }
      __syncthreads();
    }
    if (tid == 0) {
      // score_threshold (qdrant.py:83,93): candidates below it are not results - id -1 / score -inf pads, like a
      // bank with fewer than k rows
      const bool ok = rp[0] >= 0 && rs[0] >= thr;
      store_result(out_ids, out_scores, (long)b * k + kk, ok ? ri[0] + id_base : -1, ok ? rs[0] : -INFINITY);
      if (rp[0] >= 0) s[rp[0]] = -INFINITY;  // retire
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Two-pass match for query batches (>= 128 queries, K % 64 == 0, k <= 4).
//   pass 1  gemm_sp_kernel<..., AMODE 6, EPI 16>: fp16 x fp16 products of the hi halves only (one MFMA per product instead
//           of three, half the bank bytes: 154 MB for 100k x 768, which stays in the Infinity Cache); per 96-column wave
//           range the KP = 2 best approximate scores
//   pass 2  rerank_kernel, one block per query: the R = 8 best reported candidates are scored exactly (f32 master rows,
//           float64 accumulation) and the top k of those are the answer - PROVIDED no other column can beat them:
//           every column that was not re-ranked has an approximate score <= B = max(a_R, max over ranges of the range's
//           last reported score), and approximate and exact scores differ by at most EPS (both operands rounded to
//           fp16: 2 * 2^-11 * sum |q_k b_k| <= 2^-10 for unit vectors, plus f32 accumulation).  If the k-th exact score
//           is below B + EPS the block scores exactly every row that could still enter: the reported candidates whose own
//           approximate score comes within EPS of it and all columns of the wave ranges whose cut-off does (typically
//           one or two ranges; banks with many near-duplicates of a query's best match scan more).
// The answer is therefore always the exact top k (score desc, id asc), like the one-pass path's.
// ---------------------------------------------------------------------------
constexpr int PRE_KP = 2;     // candidates per wave range
constexpr int PRE_R = 8;      // candidates re-ranked per query
constexpr float PRE_EPS = 1.1e-3f;

__global__ __launch_bounds__(256) void f32_to_f16_rows_kernel(const float* __restrict__ in, const float* __restrict__ wscale,
                                                             _Float16* __restrict__ out, long rows, int K) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per 8 elements
  const int k8 = K / 8;
  if (i >= rows * k8) return;
  const long row = i / k8;
  const float sc = wscale != nullptr ? 1.0f / wscale[row] : 1.0f;  // exact: a power of two
  const sp_f4 a = *reinterpret_cast<const sp_f4*>(in + i * 8) * sc, b = *reinterpret_cast<const sp_f4*>(in + i * 8 + 4) * sc;
  sp_h8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (_Float16)a[e], o[4 + e] = (_Float16)b[e];
  *reinterpret_cast<sp_h8*>(out + i * 8) = o;
}

// exact score of (query, bank row): float64 sum of the f32 products, 64 lanes over k, fixed reduction order
__device__ __forceinline__ float exact_dot_wave(const float* __restrict__ q, const float* __restrict__ row, int K, int lane) {
  double acc = 0.0;
  for (int k = lane * 4; k < K; k += 256) {
    const sp_f4 a = *reinterpret_cast<const sp_f4*>(q + k), b = *reinterpret_cast<const sp_f4*>(row + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += (double)a[e] * (double)b[e];
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
  return (float)acc;
}

// four rows at once, 16 lanes each (the rescoring scan: independent loads in flight); lane group g = lane >> 4 scores
// rows[g] (< 0: none) and every lane of the group returns the row's score
__device__ __forceinline__ float exact_dot_quad(const float* __restrict__ q, const float* __restrict__ bank, long row, int K, int lane) {
  double acc = 0.0;
  if (row >= 0) {
    const float* r = bank + row * K;
    for (int k = (lane & 15) * 4; k < K; k += 64) {
      const sp_f4 a = *reinterpret_cast<const sp_f4*>(q + k), b = *reinterpret_cast<const sp_f4*>(r + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += (double)a[e] * (double)b[e];
    }
  }
#pragma unroll
  for (int m = 8; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
  return (float)acc;
}

// insert (e, i) into a descending (score desc, id asc) list of k entries held in registers (every lane of the wave alike)
__device__ __forceinline__ void topk_insert(float (&bs)[8], long (&bi)[8], int k, float e, long i) {
  for (int kk = 0; kk < k; ++kk)
    if (bi[kk] < 0 || e > bs[kk] || (e == bs[kk] && i < bi[kk])) {
      const float te = bs[kk];
      const long ti = bi[kk];
      bs[kk] = e, bi[kk] = i, e = te, i = ti;
      if (i < 0) break;
    }
}

__global__ __launch_bounds__(256) void rerank_kernel(const float* __restrict__ qn, const float* __restrict__ bank, long nrows, int K,
                                                    const float* __restrict__ cand_s, const int* __restrict__ cand_i, int slots,
                                                    int range_cols, int k, long id_base, float thr, long* __restrict__ out_ids,
                                                    float* __restrict__ out_scores, int* __restrict__ n_fallback) {
  __shared__ float rs[256];
  __shared__ int ri[256], rp[256];
  __shared__ int sel_p[PRE_R];
  __shared__ float fs[4][8];
  __shared__ long fi[4][8];
  __shared__ float s_cut, s_kth, s_aR;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncand = slots * PRE_KP;
  const float* cs = cand_s + (long)b * ncand;
  const int* ci = cand_i + (long)b * ncand;
  const float* q = qn + (long)b * K;
  // bound of everything a wave range did not report: its last reported score (-inf when the range ran out of columns)
  float cut = -INFINITY;
  for (int w = tid; w < slots; w += 256) cut = fmaxf(cut, cs[w * PRE_KP + PRE_KP - 1]);
  rs[tid] = cut;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) rs[tid] = fmaxf(rs[tid], rs[tid + st]);
    __syncthreads();
  }
  if (tid == 0) s_cut = rs[0];
  if (tid < PRE_R) sel_p[tid] = -1;
  __syncthreads();
  // the R best reported candidates by approximate score (score desc, id asc)
  for (int rr = 0; rr < PRE_R; ++rr) {
    float bs = -INFINITY;
    int bi = 0x7fffffff, bp = -1;
    for (int p = tid; p < ncand; p += 256) {
      const float v = cs[p];
      const int i = ci[p];
      bool taken = false;
      for (int t = 0; t < rr; ++t) taken |= sel_p[t] == p;
      if (!taken && i >= 0 && v > -INFINITY && (v > bs || (v == bs && i < bi))) bs = v, bi = i, bp = p;
    }
    rs[tid] = bs, ri[tid] = bi, rp[tid] = bp;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) {
        const float os = rs[tid + st];
        const int oi = ri[tid + st], op = rp[tid + st];
        if (op >= 0 && (rp[tid] < 0 || os > rs[tid] || (os == rs[tid] && oi < ri[tid]))) rs[tid] = os, ri[tid] = oi, rp[tid] = op;
      }
      __syncthreads();
    }
    if (tid == 0) {
      sel_p[rr] = rp[0];
      if (rr == PRE_R - 1) s_aR = rp[0] >= 0 ? rs[0] : -INFINITY;  // bounds the reported candidates that were not selected
    }
    __syncthreads();
  }
  // exact scores: every wave keeps the k best (score desc, id asc) of the rows it has scored
  float bs[8];
  long bi[8];
  for (int kk = 0; kk < 8; ++kk) bs[kk] = -INFINITY, bi[kk] = -1;
  for (int rr = wave; rr < PRE_R; rr += 4) {
    const int p = sel_p[rr];
    if (p >= 0) topk_insert(bs, bi, k, exact_dot_wave(q, bank + (long)ci[p] * K, K, lane), (long)ci[p]);
  }
  auto publish = [&]() {  // block-wide k-th best so far -> s_kth; the merged list -> fs[0], fi[0]
    if (lane == 0)
      for (int kk = 0; kk < k; ++kk) fs[wave][kk] = bs[kk], fi[wave][kk] = bi[kk];
    __syncthreads();
    if (tid == 0) {
      float ms[8];
      long mi[8];
      for (int kk = 0; kk < k; ++kk) ms[kk] = -INFINITY, mi[kk] = -1;
      for (int w = 0; w < 4; ++w)
        for (int j = 0; j < k; ++j)
          if (fi[w][j] >= 0) topk_insert(ms, mi, k, fs[w][j], fi[w][j]);
      for (int kk = 0; kk < k; ++kk) fs[0][kk] = ms[kk], fi[0][kk] = mi[kk];
      s_kth = mi[k - 1] >= 0 ? ms[k - 1] : -INFINITY;
    }
    __syncthreads();
  };
  publish();
  // Any row not scored yet has an approximate score <= max(a_R, s_cut), hence an exact score <= that + EPS.
  const float bound = fmaxf(s_aR, s_cut);
  if (bound > -INFINITY && !(s_kth > bound + PRE_EPS)) {
    // The bound does not prove the answer (near-duplicate rows, or two strong rows inside one wave range): score exactly
    // every row that could still enter - reported candidates whose own approximate score comes within EPS of the k-th
    // best, and all columns of the ranges whose cut-off does.  Each is scored once: the lists cannot hold duplicates.
    if (tid == 0 && n_fallback != nullptr) atomicAdd(n_fallback, 1);
    const float need = s_kth - PRE_EPS;  // fixed for the scan: scoring more rows only raises the k-th best
    if (wave == 0)  // wave 0 restarts from the merged list, the others from nothing: every scored row lives in one list
      for (int kk = 0; kk < k; ++kk) bs[kk] = fs[0][kk], bi[kk] = fi[0][kk];
    else
      for (int kk = 0; kk < k; ++kk) bs[kk] = -INFINITY, bi[kk] = -1;
    __syncthreads();
    for (int p = wave; p < ncand; p += 4) {
      const int i = ci[p];
      if (i < 0 || !(cs[p] >= need)) continue;
      bool taken = false;
      for (int t = 0; t < PRE_R; ++t) taken |= sel_p[t] == p;
      if (!taken) topk_insert(bs, bi, k, exact_dot_wave(q, bank + (long)i * K, K, lane), (long)i);
    }
    for (int w = 0; w < slots; ++w) {  // every wave walks the ranges; a dangerous range's columns are dealt out 16 at a time
      if (!(cs[w * PRE_KP + PRE_KP - 1] >= need)) continue;
      const long c0 = (long)w * range_cols;
      for (long cb = c0 + wave * 4; cb < c0 + range_cols; cb += 16) {
        long c = cb + (lane >> 4);
        bool skip = c >= nrows || c >= c0 + range_cols;
        for (int j = 0; j < PRE_KP; ++j) skip |= (long)ci[w * PRE_KP + j] == c;
        if (skip) c = -1;
        const float e = exact_dot_quad(q, bank, c, K, lane);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const long cg = __shfl(c, gq * 16);
          const float eg = __shfl(e, gq * 16);
          if (cg >= 0) topk_insert(bs, bi, k, eg, cg);
        }
      }
    }
    publish();
  }
  if (tid == 0)
    for (int kk = 0; kk < k; ++kk) {
      const bool ok = fi[0][kk] >= 0 && fs[0][kk] >= thr;
      store_result(out_ids, out_scores, (long)b * k + kk, ok ? fi[0][kk] + id_base : -1, ok ? fs[0][kk] : -INFINITY);
    }
}

// Merge of the all-gathered per-shard candidates (mtgv/dist.py step 4): gathered[R][b_total][k][2] in the exchange
// format above; block b merges the R * k candidates of query row0 + b (score desc, id asc; ids are global already).
__global__ __launch_bounds__(256) void topk_merge_gathered_kernel(const long* __restrict__ gathered, int R, int b_total, int k, int row0,
                                                                 float thr, long* __restrict__ out_ids, float* __restrict__ out_scores) {
  extern __shared__ __attribute__((aligned(16))) char gm_sm[];
  const int ncand = R * k;
  long* ci = reinterpret_cast<long*>(gm_sm);
  float* cs = reinterpret_cast<float*>(gm_sm + (size_t)ncand * sizeof(long));
  __shared__ float rs[256];
  __shared__ long ri[256];
  __shared__ int rp[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int p = tid; p < ncand; p += 256) {
    const int r = p / k, kk = p - r * k;
    const long* e = gathered + (((long)r * b_total + row0 + b) * k + kk) * 2;
    ci[p] = e[0];
    cs[p] = __uint_as_float((unsigned)e[1]);
  }
  __syncthreads();
  for (int kk = 0; kk < k; ++kk) {
    float bs = -INFINITY;
    long bi = 0x7fffffffffffffffL;
    int bp = -1;
    for (int p = tid; p < ncand; p += 256) {
      const float v = cs[p];
      const long i = ci[p];
      if (i >= 0 && v > -INFINITY && (v > bs || (v == bs && i < bi))) bs = v, bi = i, bp = p;
    }
    rs[tid] = bs, ri[tid] = bi, rp[tid] = bp;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) {
        const float os = rs[tid + st];
        const long oi = ri[tid + st];
        const int op = rp[tid + st];
        if (op >= 0 && (rp[tid] < 0 || os > rs[tid] || (os == rs[tid] && oi < ri[tid]))) rs[tid] = os, ri[tid] = oi, rp[tid] = op;
      }
      __syncthreads();
    }
    if (tid == 0) {
      const bool ok = rp[0] >= 0 && rs[0] >= thr;
      out_scores[(long)b * k + kk] = ok ? rs[0] : -INFINITY;
      out_ids[(long)b * k + kk] = ok ? ri[0] : -1;
      if (rp[0] >= 0) cs[rp[0]] = -INFINITY;  // retire
    }
    __syncthreads();
  }
}

void topk_merge_gathered_launch(const int64_t* gathered, int R, int b_total, int k, int row0, int b, float thr, int64_t* ids,
                                float* scores, hipStream_t s) {
  MTGV_CHECK(R > 0 && k > 0 && b > 0 && row0 >= 0 && row0 + b <= b_total, ERR_INVALID, "topk_merge_gathered: R=%d k=%d rows [%d, %d) of %d", R,
             k, row0, row0 + b, b_total);
  MTGV_CHECK((long)R * k <= 4096, ERR_INVALID, "topk_merge_gathered: %d x %d candidates per query (at most 4096)", R, k);
  const size_t lds = (size_t)R * k * (sizeof(long) + sizeof(float));
  hipLaunchKernelGGL(topk_merge_gathered_kernel, dim3(b), dim3(256), lds, s, (const long*)gathered, R, b_total, k, row0, thr, (long*)ids,
                     scores);
  HIP_OK(hipGetLastError());
}

void topk_merge_launch_i64(float* cs, const int64_t* ci, int b, int ncand, int k, float thr, int64_t* ids, float* scores,
                           hipStream_t s) {
  MTGV_CHECK(b > 0 && ncand > 0 && k > 0, ERR_INVALID, "topk_merge: b=%d ncand=%d k=%d", b, ncand, k);
  hipLaunchKernelGGL((topk_merge_kernel<long>), dim3(b), dim3(256), 0, s, cs, (const long*)ci, ncand, k, 0L, thr, (long*)ids, scores);
  HIP_OK(hipGetLastError());
}

Bank::Bank(int dim, int64_t capacity) : dim_(dim), cap_(capacity) {
  MTGV_CHECK(dim > 0 && dim % 4 == 0, ERR_INVALID, "bank: dim=%d must be a positive multiple of 4", dim);
  MTGV_CHECK(capacity > 0 && capacity < (1ll << 31), ERR_INVALID, "bank: capacity=%lld", (long long)capacity);
  vecs_.alloc((size_t)capacity * dim);
  // the bank is the B operand of the match GEMM; rows of `dim` floats get per-row scaled split copies
  gemm_split_register(vecs_.p, (size_t)capacity * dim, dim % 8 == 0 ? dim : 0);
  // workspace for the usual query batches up front (1024 queries, k <= 8, the candidate layout with the most groups:
  // 64-column tiles), so that topk does not allocate on the hot path; larger requests still grow it once
  stat_.alloc(4);
  HIP_OK(hipMemset(stat_.p, 0, 4 * sizeof(float)));
  const size_t groups = (size_t)ceil_div((int)capacity, 64);
  if (groups * 8 * 1024 * sizeof(float) <= ((size_t)256 << 20)) {
    qn_.ensure((size_t)1024 * dim);
    qhi_.ensure((size_t)1024 * dim / 2 + 8);
    cand_s_.ensure((size_t)1024 * groups * 8);
    cand_i_.ensure((size_t)1024 * groups * 8);
  }
}

Bank::~Bank() { gemm_split_unregister(vecs_.p); }

// fp16 hi halves of rows [row0, row0 + rows) of the (row-scaled) bank: what the SP8 copy holds as its hi pieces, contiguous
void Bank::refresh_hi(int64_t row0, int64_t rows, hipStream_t s) {
  if (dim_ % 64 != 0 || rows <= 0) return;
  hi_.ensure((size_t)cap_ * dim_ / 2 + 8);
  const float* wsc = nullptr;
  if (!sp8_lookup(vecs_.p + (size_t)row0 * dim_, dim_, nullptr, &wsc)) return;
  const long n8 = (long)rows * (dim_ / 8);
  hipLaunchKernelGGL(f32_to_f16_rows_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, vecs_.p + (size_t)row0 * dim_, wsc,
                     reinterpret_cast<_Float16*>(hi_.p) + (size_t)row0 * dim_, (long)rows, dim_);
  HIP_OK(hipGetLastError());
}

int64_t Bank::prepass_fallbacks() const {
  int v = 0;
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(&v, stat_.p, sizeof(int), hipMemcpyDeviceToHost));
  return v;
}

bool Bank::prepass_ok(int b, int k) const {
  const char* e = getenv("MTGV_MATCH_PREPASS");  // read per call: tests and tools compare the two paths in one process
  const bool on = e == nullptr || atoi(e) != 0;
  return on && gemm_sp_active() && topk_sp_on() && b >= 128 && k <= 4 && dim_ % 64 == 0 && size_ >= 4096 && hi_.p != nullptr;
}

void Bank::topk_prepass(const float* q, int b, int k, int64_t id_base, float thr, int64_t* ids, float* scores, hipStream_t s) {
  const int N = (int)size_;
  const int slots = gemm_sp_topk_hi16_slots(N);
  const size_t ncand = (size_t)slots * PRE_KP;
  qn_.ensure((size_t)b * dim_);
  qhi_.ensure((size_t)b * dim_ / 2 + 8);
  cand_s_.ensure((size_t)b * ncand);
  cand_i_.ensure((size_t)b * ncand);
  l2norm_rows_launch(q, qn_.p, b, dim_, s);
  const long n8 = (long)b * (dim_ / 8);
  hipLaunchKernelGGL(f32_to_f16_rows_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, qn_.p, (const float*)nullptr,
                     reinterpret_cast<_Float16*>(qhi_.p), (long)b, dim_);
  HIP_OK(hipGetLastError());
  const float* wsc = nullptr;
  MTGV_CHECK(sp8_lookup(vecs_.p, dim_, nullptr, &wsc), ERR_RUNTIME, "bank: row scales missing");
  int sl = 0;
  {  // recorded for the roofline like the one-pass launch it replaces (sp = 3: one fp16 MFMA per product, fp16 bank bytes)
    GemmArgs rec;
    rec.M = b, rec.N = N, rec.K = dim_, rec.topk = PRE_KP;
    const double tiles = (double)ceil_div(b, 128) * ceil_div(N, 192);
    gemm_profile_begin(rec, s, 3, tiles * (128 + 192) * dim_ * 2.0, 2.0 * ((double)N * dim_ + (double)b * dim_) + 8.0 * b * slots * PRE_KP);
  }
  gemm_sp_topk_hi16_launch(qhi_.p, hi_.p, wsc, b, N, dim_, PRE_KP, cand_s_.p, reinterpret_cast<int*>(cand_i_.p), &sl, s);
  gemm_profile_end(s);
  hipLaunchKernelGGL(rerank_kernel, dim3(b), dim3(256), 0, s, qn_.p, vecs_.p, (long)N, dim_, (const float*)cand_s_.p,
                     (const int*)cand_i_.p, sl, gemm_sp_topk_hi16_range_cols(), k, (long)id_base, thr, (long*)ids, scores,
                     reinterpret_cast<int*>(stat_.p));
  HIP_OK(hipGetLastError());
}

void Bank::append(const float* v, int64_t n, bool is_device, hipStream_t s) {
  MTGV_CHECK(n >= 0 && size_ + n <= cap_, ERR_INVALID, "bank: %lld + %lld rows exceed capacity %lld", (long long)size_,
             (long long)n, (long long)cap_);
  if (n == 0) return;
  float* dst = vecs_.p + (size_t)size_ * dim_;
  HIP_OK(hipMemcpyAsync(dst, v, (size_t)n * dim_ * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  l2norm_rows_launch(dst, dst, n, dim_, s);
  gemm_split_refresh(vecs_.p, (size_t)size_ * dim_, (size_t)n * dim_, s);
  refresh_hi(size_, n, s);
  if (!is_device) HIP_OK(hipStreamSynchronize(s));  // host buffer may be freed by the caller
  size_ += n;
}

void Bank::set_row(int64_t row, const float* v_host, hipStream_t s) {
  MTGV_CHECK(row >= 0 && row < size_, ERR_INVALID, "bank: row %lld outside [0, %lld)", (long long)row, (long long)size_);
  float* dst = vecs_.p + (size_t)row * dim_;
  HIP_OK(hipMemcpyAsync(dst, v_host, (size_t)dim_ * sizeof(float), hipMemcpyHostToDevice, s));
  l2norm_rows_launch(dst, dst, 1, dim_, s);
  gemm_split_refresh(vecs_.p, (size_t)row * dim_, (size_t)dim_, s);
  refresh_hi(row, 1, s);
  HIP_OK(hipStreamSynchronize(s));
}

void Bank::get_rows(int64_t row, int64_t n, float* out_host) const {
  MTGV_CHECK(row >= 0 && n >= 0 && row + n <= size_, ERR_INVALID, "bank: rows [%lld, %lld) outside [0, %lld)", (long long)row,
             (long long)(row + n), (long long)size_);
  if (n == 0) return;
  HIP_OK(hipMemcpy(out_host, vecs_.p + (size_t)row * dim_, (size_t)n * dim_ * sizeof(float), hipMemcpyDeviceToHost));
}

void Bank::topk(const float* q, int b, int k, int64_t id_base, float thr, int64_t* ids, float* scores, hipStream_t s) {
  MTGV_CHECK(b > 0 && k > 0 && k <= 65536, ERR_INVALID, "bank: b=%d k=%d (k must be in [1,65536])", b, k);
  MTGV_CHECK(q != nullptr && ids != nullptr, ERR_INVALID, "bank: null tensor");  // scores == nullptr: exchange format in ids (store_result)
  MTGV_CHECK(size_ > 0, ERR_RUNTIME, "bank is empty");
  if (prepass_ok(b, k)) {
    topk_prepass(q, b, k, id_base, thr, ids, scores, s);
    return;
  }
  GemmPlan pl;
  pl.tm = 1, pl.tn = 2, pl.bk = 16;
  pl.tiles_m = ceil_div(b, pl.bm());
  pl.tiles_n = ceil_div((int)size_, pl.bn());
  qn_.ensure((size_t)b * dim_);
  GemmArgs g = linear_args(qn_.p, dim_, vecs_.p, nullptr, nullptr, 0, b, (int)size_, dim_, ACT_NONE);
  // candidate groups: 64-column tiles of the convert-on-load kernel, or the per-wave column ranges of the LDS-DMA kernel
  int slots = pl.tiles_n, cols = pl.bn();
  g.topk = 1;
  (void)gemm_sp_topk_layout(g, &slots, &cols);
  const int kt = k < cols ? k : cols;  // a group cannot contribute more candidates than it has columns
  const size_t ncand = (size_t)slots * kt;
  cand_s_.ensure((size_t)b * ncand);
  cand_i_.ensure((size_t)b * ncand);
  l2norm_rows_launch(q, qn_.p, b, dim_, s);
  g.cand_s = cand_s_.p;
  g.cand_i = reinterpret_cast<int*>(cand_i_.p);
  g.topk = kt;
  gemm_launch(g, pl, s);
  hipLaunchKernelGGL((topk_merge_kernel<int>), dim3(b), dim3(256), 0, s, cand_s_.p, (const int*)cand_i_.p, (int)ncand, k,
                     (long)id_base, thr, (long*)ids, scores);
  HIP_OK(hipGetLastError());
}

}  // namespace mtgv
