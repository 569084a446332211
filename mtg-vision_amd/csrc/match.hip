// Exact cosine top-k over the whole card bank (what Qdrant's HNSW index approximates;
// mtgvision/qdrant.py:76-95).  Bank rows are stored L2-normalised, so cosine = dot:
//   1. l2norm_rows_kernel        q -> q/|q|
//   2. gemm_f32 (EPI=1)          S tile = Q B^T on f32 MFMA, per-tile top-k in registers
//   3. topk_merge_kernel         tiles_n*k candidates per query -> top k (score desc, id asc)
#include "match.h"
#include "gemm_sp.h"
#include "rowops.h"

namespace mtgv {

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
      out_scores[(long)b * k + kk] = ok ? rs[0] : -INFINITY;
      out_ids[(long)b * k + kk] = ok ? ri[0] + id_base : -1;
      if (rp[0] >= 0) s[rp[0]] = -INFINITY;  // retire
    }
    __syncthreads();
  }
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
  const size_t groups = (size_t)ceil_div((int)capacity, 64);
  if (groups * 8 * 1024 * sizeof(float) <= ((size_t)256 << 20)) {
    qn_.ensure((size_t)1024 * dim);
    cand_s_.ensure((size_t)1024 * groups * 8);
    cand_i_.ensure((size_t)1024 * groups * 8);
  }
}

Bank::~Bank() { gemm_split_unregister(vecs_.p); }

void Bank::append(const float* v, int64_t n, bool is_device, hipStream_t s) {
  MTGV_CHECK(n >= 0 && size_ + n <= cap_, ERR_INVALID, "bank: %lld + %lld rows exceed capacity %lld", (long long)size_,
             (long long)n, (long long)cap_);
  if (n == 0) return;
  float* dst = vecs_.p + (size_t)size_ * dim_;
  HIP_OK(hipMemcpyAsync(dst, v, (size_t)n * dim_ * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  l2norm_rows_launch(dst, dst, n, dim_, s);
  gemm_split_refresh(vecs_.p, (size_t)size_ * dim_, (size_t)n * dim_, s);
  if (!is_device) HIP_OK(hipStreamSynchronize(s));  // host buffer may be freed by the caller
  size_ += n;
}

void Bank::set_row(int64_t row, const float* v_host, hipStream_t s) {
  MTGV_CHECK(row >= 0 && row < size_, ERR_INVALID, "bank: row %lld outside [0, %lld)", (long long)row, (long long)size_);
  float* dst = vecs_.p + (size_t)row * dim_;
  HIP_OK(hipMemcpyAsync(dst, v_host, (size_t)dim_ * sizeof(float), hipMemcpyHostToDevice, s));
  l2norm_rows_launch(dst, dst, 1, dim_, s);
  gemm_split_refresh(vecs_.p, (size_t)row * dim_, (size_t)dim_, s);
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
  MTGV_CHECK(q != nullptr && ids != nullptr && scores != nullptr, ERR_INVALID, "bank: null tensor");
  MTGV_CHECK(size_ > 0, ERR_RUNTIME, "bank is empty");
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
