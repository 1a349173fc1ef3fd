// oracle/ref_shim.cpp — extern "C" doors onto the REFERENCE's own evaluator.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no algorithm: it #includes the
// reference headers where they lie under /root/reference (never copied into
// this repository) and forwards to them, so that oracle/_ref/libneurec_ref.so
// *is* the reference's C++ (evaluate.h, metric.h, arg_topk.h, thread_pool.h)
// compiled by g++.  Built by oracle/Makefile when /root/reference exists.
#include <unordered_set>
#include <vector>
#include <cstdint>
#include "evaluate.h"   // reference: evaluator/backend/cpp/include/evaluate.h
#include "arg_topk.h"   // reference: util/cython/include/arg_topk.h

extern "C" {

// cpp_evaluate_matrix(float*, int, vector<unordered_set<int>>&, vector<int>, int, int, float*)
// marshalled the way cpp_evaluator.pyx:28-42 does (lists -> vector<unordered_set<int>>).
void ref_cpp_evaluate_matrix(float* rating_matrix, int rating_len, int rows, const int64_t* tptr,
                             const int32_t* tidx, const int* metric, int nm, int top_k,
                             int thread_num, float* results) {
  std::vector<std::unordered_set<int> > test_items(rows);
  for (int r = 0; r < rows; ++r)
    for (int64_t j = tptr[r]; j < tptr[r + 1]; ++j) test_items[r].insert(tidx[j]);
  std::vector<int> metric_vec(metric, metric + nm);
  cpp_evaluate_matrix(rating_matrix, rating_len, test_items, metric_vec, top_k, thread_num,
                      results);
}

void ref_arg_top_k_2d(float* scores, int columns_num, int rows_num, int top_k, int thread_num,
                      int* results) {
  arg_top_k_2d(scores, columns_num, rows_num, top_k, thread_num, results);
}

}  // extern "C"
