// eval_pipeline.hip — the pruned full-rank evaluation of a user list as ONE native call.
//
// Stands in for the batch loop of UniEvaluator.evaluate (evaluator/backend/cpp/uni_evaluator.py:101-157: per batch
// model.predict -> strike train items -> eval_score_matrix, then the mean over users).  The steps are the library's
// own entry points (score_bf16.hip, score_gemm.hip, eval_select.hip) issued back to back from C: at ~1 ms per
// evaluation of 29,858 users a dozen Python-level calls per batch were a tenth of the wall time.
#include "nr_common.h"
#include "neurec_hip.h"

namespace {

// sums[n_cols] <- the flagged-row count (as a double, next to the column sums: one device->host copy brings both)
__global__ __launch_bounds__(256) void count_flags_kernel(const int32_t* __restrict__ flags, int n,
                                                          double* __restrict__ out) {
  __shared__ int s_part[256];
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) c += flags[i] != 0 ? 1 : 0;
  s_part[threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s_part[threadIdx.x] += s_part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (double)s_part[0];
}

}  // namespace

extern "C" {

int nrhip_eval_pruned(const NrhipEvalPruned* a, void* stream) {
  NR_REQUIRE(a, NR_ERR_ARG, "eval_pruned: null argument block");
  NR_REQUIRE(a->d_P && a->d_Q && a->d_out && a->d_flags && a->d_M && a->d_gemm_ws && a->d_tiles_ws &&
                 a->d_tr_indptr && a->d_tr_indices && a->d_truth_indptr && a->d_truth_indices && a->metric_ids,
             NR_ERR_ARG, "eval_pruned: null pointer argument");
  NR_REQUIRE(a->n_users >= 0 && a->batch_rows >= 1 && a->cols >= 1 && a->d >= 1 && a->top_k >= 1 &&
                 a->n_keep >= a->top_k + 1 && a->n_metric >= 1 && a->n_metric <= 8,
             NR_ERR_ARG, "eval_pruned: bad sizes");
  NR_REQUIRE(a->d_users && a->d_tile_ptr && a->d_row_of, NR_ERR_ARG,
             "eval_pruned: the user list, the strike plan and the user -> row table are part of this form");
  NR_REQUIRE(!a->use_filter || (a->d_filter_ws && a->d_eps), NR_ERR_ARG, "eval_pruned: the filter needs its workspace");
  NR_REQUIRE(a->use_filter || a->n_keep == a->top_k + 1, NR_ERR_ARG,
             "eval_pruned: exact maxima take n_keep = top_k + 1");
  if (a->prepare_items) {
    NR_TRY(nrhip_score_gemm_prepare_items(a->d_Q, a->ldq, a->cols, a->d, a->d_gemm_ws, a->gemm_ws_bytes, stream));
    if (a->use_filter)
      NR_TRY(nrhip_score_filter_prepare_items(a->d_Q, a->ldq, a->cols, a->d, a->d_filter_ws, a->filter_ws_bytes,
                                              a->batch_rows, stream));
  }
  const int out_ld = a->n_metric * a->top_k;
  for (int b = 0; b < a->n_users; b += a->batch_rows) {
    const int rows = a->n_users - b < a->batch_rows ? a->n_users - b : a->batch_rows;
    const int32_t* users = a->d_users + b;
    const float* P = a->d_P;
    if (a->use_filter)
      NR_TRY(nrhip_score_filter_tilemax(P, a->ldp, users, rows, a->cols, a->d, a->d_M, a->mld, a->d_eps,
                                        a->d_filter_ws, a->filter_ws_bytes, a->batch_rows, stream));
    else
      NR_TRY(nrhip_score_tilemax(P, a->ldp, users, rows, a->cols, a->d, nullptr, nullptr, a->d_M, a->mld,
                                 a->d_gemm_ws, a->gemm_ws_bytes, stream));
    // (the plan's users are table rows: the fix-up reads the whole table, row_of maps a user to its evaluation row)
    NR_TRY(nrhip_score_tilemax_fix(a->d_P, a->ldp, a->d, a->cols, a->d_chunk_tile, a->d_chunk_begin, a->n_chunks,
                                   a->d_tile_ptr, a->d_plan_user, a->d_plan_mask, a->d_row_of, b, rows, a->d_M,
                                   a->mld, a->d_gemm_ws, a->gemm_ws_bytes, stream));
    NR_TRY(nrhip_eval_tiles_bounded(a->d_M, a->mld, a->use_filter ? a->d_eps : nullptr, a->n_keep, P, a->ldp,
                                    a->d_gemm_ws, a->d, users, rows, a->cols, a->d_tr_indptr, a->d_tr_indices,
                                    a->d_truth_indptr, a->d_truth_indices, a->metric_ids, a->n_metric, a->top_k,
                                    a->d_out + (int64_t)b * out_ld, a->d_flags + b, a->d_tiles_ws, a->tiles_ws_bytes,
                                    stream));
  }
  if (a->d_sums && a->n_users > 0) {
    NR_REQUIRE(a->d_colsum_ws, NR_ERR_ARG, "eval_pruned: column sums need their workspace");
    NR_TRY(nrhip_colsum_f64(a->d_out, out_ld, a->n_users, out_ld, a->d_sums, a->d_colsum_ws, a->colsum_ws_bytes, stream));
    hipLaunchKernelGGL(count_flags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a->d_flags, a->n_users,
                       a->d_sums + out_ld);
    NR_LAUNCH_CHECK();
  }
  return NR_OK;
}

}  // extern "C"
