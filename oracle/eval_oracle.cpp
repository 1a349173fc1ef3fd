// oracle/eval_oracle.cpp — CPU restatement of the NeuRec native islands.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under neurec_amd/ may include, link or call
// this file; it exists so tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can check (and time) the HIP engine against an independent
// CPU statement of the reference algorithm.
//
// Each function cites the reference code it follows (paths relative to the
// NeuRec tree).  The restatement is pinned two ways (tests/test_oracle_cpu.py):
//   * against the known-answer vectors of SURVEY.md §4 / Appendix A, and
//   * against oracle/_ref — the reference's own headers / Cython module
//     compiled as they are — through the committed fixtures in tests/golden/.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

namespace {

// metric.h:17-28
void m_precision(const int* rank, int k, const int32_t* truth, int t, float* out) {
  int hits = 0;
  for (int i = 0; i < k; ++i) {
    if (std::binary_search(truth, truth + t, rank[i])) hits += 1;
    out[i] = 1.0 * hits / (unsigned)(i + 1);
  }
}
// metric.h:31-43
void m_recall(const int* rank, int k, const int32_t* truth, int t, float* out) {
  int hits = 0;
  size_t truth_len = (size_t)t;
  for (int i = 0; i < k; ++i) {
    if (std::binary_search(truth, truth + t, rank[i])) hits += 1;
    out[i] = 1.0 * hits / truth_len;
  }
}
// metric.h:46-65
void m_ap(const int* rank, int k, const int32_t* truth, int t, float* out) {
  int hits = 0;
  float pre = 0, sum_pre = 0, denominator = 1;
  float truth_len = (float)(size_t)t;
  for (unsigned int i = 0; i < (unsigned)k; ++i) {
    if (std::binary_search(truth, truth + t, rank[i])) {
      hits += 1;
      pre = 1.0 * hits / (i + 1);
      sum_pre += pre;
    }
    denominator = (truth_len < i + 1) ? truth_len : i + 1;
    out[i] = (hits == 0) ? 0.0 : sum_pre / denominator;
  }
}
// metric.h:69-86
void m_ndcg(const int* rank, int k, const int32_t* truth, int t, float* out) {
  float idcg = 0, dcg = 0;
  size_t truth_len = (size_t)t;
  for (unsigned int i = 0; i < (unsigned)k; ++i) {
    if (std::binary_search(truth, truth + t, rank[i])) dcg += 1.0 / log2(i + 2);
    if (i < truth_len) idcg += 1.0 / log2(i + 2);
    out[i] = dcg / idcg;
  }
}
// metric.h:89-109
void m_mrr(const int* rank, int k, const int32_t* truth, int t, float* out) {
  float rr = 0;
  for (int i = 0; i < k; ++i) {
    if (std::binary_search(truth, truth + t, rank[i])) {
      rr = 1.0 / (unsigned)(i + 1);
      for (int j = i; j < k; ++j) out[j] = rr;
      break;
    } else {
      rr = 0.0;
      out[i] = rr;
    }
  }
}

typedef void (*metric_fn)(const int*, int, const int32_t*, int, float*);
metric_fn metric_by_id(int id) {   // metric.h:111-117
  switch (id) {
    case 1: return m_precision;
    case 2: return m_recall;
    case 3: return m_ap;
    case 4: return m_ndcg;
    case 5: return m_mrr;
  }
  return nullptr;
}

// evaluate.h:23-50
void eval_row(const float* ratings, int n, const int32_t* truth, int t, const int* metric, int nm,
              int top_k, float* out, int* topk_out) {
  std::vector<int> index(n);
  std::iota(index.begin(), index.end(), 0);
  int sort_len = std::min(top_k * 2, n);
  std::vector<int> topk_rank(sort_len);
  std::partial_sort_copy(index.begin(), index.end(), topk_rank.begin(), topk_rank.end(),
                         [ratings](int a, int b) { return ratings[a] > ratings[b]; });
  for (int m = 0; m < nm; ++m) metric_by_id(metric[m])(topk_rank.data(), top_k, truth, t,
                                                        out + (size_t)m * top_k);
  if (topk_out) std::memcpy(topk_out, topk_rank.data(), sizeof(int) * top_k);
}

template <class F>
void parallel_rows(int rows, int threads, F f) {
  threads = std::max(1, std::min(threads, rows));
  std::vector<std::thread> pool;
  for (int w = 0; w < threads; ++w)
    pool.emplace_back([=]() { for (int r = w; r < rows; r += threads) f(r); });
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

// cpp_evaluate_matrix (evaluate.h:53-72): one task per row, `threads` workers.
// truth of row r = tidx[tptr[r] .. tptr[r+1]) ascending.
int oracle_eval_matrix(const float* scores, int64_t ld, int rows, int cols, const int64_t* tptr,
                       const int32_t* tidx, const int* metric, int nm, int top_k, int threads,
                       float* out, int* topk_out) {
  for (int m = 0; m < nm; ++m)
    if (!metric_by_id(metric[m])) return 1;
  if (top_k > cols) return 2;
  parallel_rows(rows, threads, [=](int r) {
    eval_row(scores + (size_t)r * ld, cols, tidx + tptr[r], (int)(tptr[r + 1] - tptr[r]), metric,
             nm, top_k, out + (size_t)r * nm * top_k,
             topk_out ? topk_out + (size_t)r * top_k : nullptr);
  });
  return 0;
}

// arg_top_k_2d (arg_topk.h:15-45)
int oracle_arg_topk(const float* scores, int64_t ld, int rows, int cols, int top_k, int threads,
                    int* out) {
  if (top_k > cols) return 2;
  parallel_rows(rows, threads, [=](int r) {
    const float* ratings = scores + (size_t)r * ld;
    std::vector<int> index(cols);
    std::iota(index.begin(), index.end(), 0);
    std::partial_sort_copy(index.begin(), index.end(), out + (size_t)r * top_k,
                           out + (size_t)(r + 1) * top_k,
                           [ratings](int a, int b) { return ratings[a] > ratings[b]; });
  });
  return 0;
}

// uni_evaluator.py:140-143
void oracle_mask_train(float* scores, int64_t ld, int rows, int cols, const int32_t* users,
                       const int64_t* tr_ptr, const int32_t* tr_idx) {
  for (int r = 0; r < rows; ++r) {
    int64_t u = users ? users[r] : r;
    for (int64_t j = tr_ptr[u]; j < tr_ptr[u + 1]; ++j)
      if (tr_idx[j] >= 0 && tr_idx[j] < cols) scores[(size_t)r * ld + tr_idx[j]] = -INFINITY;
  }
}

// Scoring (MF.py:120-122 np.matmul / LightGCN.py:118-119 tf.matmul), stated as
// the k-ascending fused chain the fp32 matrix cores evaluate:
//   acc = 0; acc = fmaf(P[u][k], Q[i][k], acc) for k = 0..d-1.
// (BLAS/TF do not define a summation order; this one is exact to the last bit
// on the GPU and within 2 ulp-of-sum of any other order.)
void oracle_score_gemm(const float* P, int64_t ldp, const int32_t* users, int rows, const float* Q,
                       int64_t ldq, int cols, int d, float* S, int64_t lds, int threads) {
  parallel_rows(rows, threads, [=](int r) {
    const float* p = P + (size_t)(users ? users[r] : r) * ldp;
    for (int i = 0; i < cols; ++i) {
      const float* q = Q + (size_t)i * ldq;
      float acc = 0.f;
      for (int k = 0; k < d; ++k) acc = std::fmaf(p[k], q[k], acc);
      S[(size_t)r * lds + i] = acc;
    }
  });
}

// random_choice.pyx:12-17
static unsigned long long llrand() {
  unsigned long long r = 0;
  for (int i = 0; i < 5; ++i) r = (r << 15) | (unsigned long long)(rand() & 0x7FFF);
  return r & 0xFFFFFFFFFFFFFFFFULL;
}

void oracle_srand(unsigned seed) { srand(seed); }

// randint_choice (random_choice.pyx:20-62); excl need not be sorted.
// returns 0 ok, 1 ValueError(size), 2 ValueError(exclusion >= high), 3 ValueError(not enough)
int oracle_randint_choice(int high, int size, int replace, const int32_t* excl, int n_excl,
                          int32_t* out) {
  if (size <= 0) return 1;
  if (excl && high <= n_excl) return 2;
  if (!replace && (high - n_excl <= size)) return 3;
  std::vector<int32_t> omit(excl, excl + n_excl);
  std::sort(omit.begin(), omit.end());
  int i = 0;
  while (size - i) {
    int a = (int)(llrand() % (unsigned long long)high);
    bool skip = std::binary_search(omit.begin(), omit.end(), a);
    if (!skip) {
      out[i++] = a;
      if (!replace) omit.insert(std::upper_bound(omit.begin(), omit.end(), a), a);
    }
  }
  return 0;
}

}  // extern "C"
