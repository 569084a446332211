// Card bank: device-resident L2-normalised vectors + exact cosine top-k.
#pragma once
#include "common.h"
#include "encoder.h"

namespace mtgv {

class Bank {
 public:
  Bank(int dim, int64_t capacity);
  ~Bank();
  int dim() const { return dim_; }
  int64_t size() const { return size_; }
  int64_t capacity() const { return cap_; }
  void clear() { size_ = 0; }
  void append(const float* v, int64_t n, bool is_device, hipStream_t s);
  void set_row(int64_t row, const float* v_host, hipStream_t s);
  void get_rows(int64_t row, int64_t n, float* out_host) const;
  // thr: results scoring below it are dropped (id -1, score -inf); -INFINITY keeps everything.
  // scores == nullptr: ids receives [b][k][2] int64 = (id, float32 bits of the score) - the sharded match's exchange format
  void topk(const float* q, int b, int k, int64_t id_base, float thr, int64_t* ids, float* scores, hipStream_t s);
  // queries of two-pass matches so far whose answer the first pass could not prove and that were scanned exactly
  // (synchronises the device)
  int64_t prepass_fallbacks() const;

 private:
  // two-pass match (match.hip): approximate fp16 scores over a hi-only copy of the bank, exact re-rank of the candidates
  bool prepass_ok(int b, int k) const;
  void topk_prepass(const float* q, int b, int k, int64_t id_base, float thr, int64_t* ids, float* scores, hipStream_t s);
  void refresh_hi(int64_t row0, int64_t rows, hipStream_t s);

  int dim_;
  int64_t cap_, size_ = 0;
  DevBuf vecs_, qn_, cand_s_, cand_i_;
  DevBuf hi_, qhi_, stat_;   // fp16 hi halves of the (row-scaled) bank rows / of the normalised queries, 2 bytes per element
};

// merge of all-gathered per-shard candidates in the exchange format (gathered[R][b_total][k][2]: id, score bits) for
// the queries [row0, row0 + b)
void topk_merge_gathered_launch(const int64_t* gathered, int R, int b_total, int k, int row0, int b, float thr, int64_t* ids,
                                float* scores, hipStream_t s);
void topk_merge_launch_i64(float* cs, const int64_t* ci, int b, int ncand, int k, float thr, int64_t* ids, float* scores,
                           hipStream_t s);

}  // namespace mtgv
