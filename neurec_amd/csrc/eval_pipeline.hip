// eval_pipeline.hip — the pruned full-rank evaluation of a user list as ONE native call.
//
// Stands in for the batch loop of UniEvaluator.evaluate (evaluator/backend/cpp/uni_evaluator.py:101-157: per batch
// model.predict -> strike train items -> eval_score_matrix, then the mean over users).  The steps are the library's
// own entry points (score_bf16.hip, score_gemm.hip, eval_select.hip) issued back to back from C: at ~1 ms per
// evaluation of 29,858 users a dozen Python-level calls per batch were a tenth of the wall time.
#include "nr_common.h"
#include "neurec_hip.h"

// eval_select.hip: the column sums with the flagged-row counts formed in their second launch — out[cols] <- rows with
// a flag, out[cols + 1] <- the rows among them whose certificate failed (flag bit 1), as doubles next to the sums
int nr_score_gemm_prepare_items_kmajor(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws, size_t ws_bytes,
                                       void* stream);                                           // score_gemm.hip
int nr_score_gemm_swizzle_items(int cols, int d, void* d_ws, size_t ws_bytes, void* stream);   // score_gemm.hip
int nr_colsum_f64_flags(const float* d_mat, int64_t ld, int rows, int cols, double* d_out, void* d_ws, size_t ws_bytes,
                        const int32_t* d_flags, double* d_flag_out, void* stream);

namespace {

// the flagged rows and their users, in whatever order the waves arrive (every row is redone independently)
__global__ __launch_bounds__(256) void flagged_rows_kernel(const int32_t* __restrict__ flags,
                                                           const int32_t* __restrict__ users, int n, int cap,
                                                           int32_t* __restrict__ count, int32_t* __restrict__ rows_out,
                                                           int32_t* __restrict__ users_out) {
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool f = i < n && flags[i] != 0;
  const unsigned long long m = __ballot(f);
  if (!m) return;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(count, __popcll(m));   // one reservation per wave
  base = __shfl(base, leader, 64);
  if (f) {
    const int at = base + __popcll(m & ((1ull << lane) - 1ull));
    if (at < cap) { rows_out[at] = i; users_out[at] = users[i]; }
  }
}

// out[rows_idx[r]][:] = fixed[r][:]
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ fixed, int cols,
                                                           const int32_t* __restrict__ rows_idx, int rows,
                                                           float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  out[(int64_t)rows_idx[r] * cols + c] = fixed[i];
}

}  // namespace

extern "C" {

int nrhip_eval_redo(const NrhipEvalRedo* r, void* stream) {
  NR_REQUIRE(r && r->ev, NR_ERR_ARG, "eval_redo: null argument block");
  const NrhipEvalPruned* a = r->ev;
  NR_REQUIRE(a->d_P && a->d_Q && a->d_out && a->d_flags && a->d_users && a->d_gemm_ws && a->d_tr_indptr &&
                 a->d_tr_indices && a->d_truth_indptr && a->d_truth_indices && a->metric_ids,
             NR_ERR_ARG, "eval_redo: the evaluation's argument block is incomplete");
  NR_REQUIRE(r->n_flagged >= 0 && r->n_flagged <= a->n_users, NR_ERR_ARG, "eval_redo: n_flagged=%d outside 0..%d",
             r->n_flagged, a->n_users);
  if (r->n_flagged == 0) return NR_OK;
  NR_REQUIRE(r->d_scores && r->d_rows && r->d_row_users && r->d_count && r->d_fixed && r->d_ws, NR_ERR_ARG,
             "eval_redo: null buffer");
  NR_REQUIRE(r->slab_rows >= 1 && r->slab_rows <= a->batch_rows && r->lds >= (a->cols + 63) / 64 * 64, NR_ERR_ARG,
             "eval_redo: slab of %d rows x %lld (1..%d rows, >= %d columns)", r->slab_rows, (long long)r->lds,
             a->batch_rows, (a->cols + 63) / 64 * 64);
  hipStream_t st = (hipStream_t)stream;
  if (r->reload_items == 2)      // the k-major copy is this table's (ev ran with prepare_items = 2): the operand copy only
    NR_TRY(nr_score_gemm_swizzle_items(a->cols, a->d, a->d_gemm_ws, a->gemm_ws_bytes, stream));
  else if (r->reload_items)
    NR_TRY(nrhip_score_gemm_prepare_items(a->d_Q, a->ldq, a->cols, a->d, a->d_gemm_ws, a->gemm_ws_bytes, stream));
  NR_CHECK_HIP(hipMemsetAsync(r->d_count, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(flagged_rows_kernel, dim3((a->n_users + 255) / 256), dim3(256), 0, st, a->d_flags, a->d_users,
                     a->n_users, r->n_flagged, r->d_count, r->d_rows, r->d_row_users);
  NR_LAUNCH_CHECK();
  const int out_ld = a->n_metric * a->top_k;
  for (int lo = 0; lo < r->n_flagged; lo += r->slab_rows) {
    const int rows = r->n_flagged - lo < r->slab_rows ? r->n_flagged - lo : r->slab_rows;
    const int32_t* users = r->d_row_users + lo;
    NR_TRY(nrhip_score_gemm(a->d_P, a->ldp, users, rows, a->cols, a->d, r->d_scores, r->lds, a->d_gemm_ws,
                            a->gemm_ws_bytes, stream));
    NR_TRY(nrhip_mask_train(r->d_scores, r->lds, users, rows, a->cols, a->d_tr_indptr, a->d_tr_indices, stream));
    NR_TRY(nrhip_eval_scores(r->d_scores, r->lds, rows, a->cols, users, a->d_truth_indptr, a->d_truth_indices,
                             a->metric_ids, a->n_metric, a->top_k, r->d_fixed, nullptr, nullptr, r->d_ws, r->ws_bytes,
                             stream));
    const int64_t n = (int64_t)rows * out_ld;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r->d_fixed, out_ld,
                       r->d_rows + lo, rows, a->d_out);
    NR_LAUNCH_CHECK();
  }
  if (a->d_sums) {
    NR_REQUIRE(a->d_colsum_ws, NR_ERR_ARG, "eval_redo: column sums need their workspace");
    NR_TRY(nr_colsum_f64_flags(a->d_out, out_ld, a->n_users, out_ld, a->d_sums, a->d_colsum_ws, a->colsum_ws_bytes,
                               a->d_flags, a->d_sums + out_ld, stream));
  }
  return NR_OK;
}

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
  NR_REQUIRE(a->use_filter >= 0 && a->use_filter <= 2, NR_ERR_ARG, "eval_pruned: use_filter is 0 (exact maxima), 1 (bf16) or 2 (int8)");
  NR_REQUIRE(a->use_filter || a->n_keep == a->top_k + 1, NR_ERR_ARG,
             "eval_pruned: exact maxima take n_keep = top_k + 1");
  NR_REQUIRE(a->prepare_items >= 0 && a->prepare_items <= 2 && (a->prepare_items != 2 || a->use_filter), NR_ERR_ARG,
             "eval_pruned: prepare_items is 0, 1, or 2 (2 with a filter only)");
  if (a->prepare_items) {
    // 2: without the fp32 scoring loop's operand copy — nothing in this call reads it when a filter searches
    if (a->prepare_items == 2)
      NR_TRY(nr_score_gemm_prepare_items_kmajor(a->d_Q, a->ldq, a->cols, a->d, a->d_gemm_ws, a->gemm_ws_bytes, stream));
    else
      NR_TRY(nrhip_score_gemm_prepare_items(a->d_Q, a->ldq, a->cols, a->d, a->d_gemm_ws, a->gemm_ws_bytes, stream));
    if (a->use_filter == 2)
      NR_TRY(nrhip_score_filter_i8_prepare_items(a->d_Q, a->ldq, a->cols, a->d, a->d_filter_ws, a->filter_ws_bytes,
                                                 a->batch_rows, stream));
    else if (a->use_filter)
      NR_TRY(nrhip_score_filter_prepare_items(a->d_Q, a->ldq, a->cols, a->d, a->d_filter_ws, a->filter_ws_bytes,
                                              a->batch_rows, stream));
  }
  const int out_ld = a->n_metric * a->top_k;
  for (int b = 0; b < a->n_users; b += a->batch_rows) {
    const int rows = a->n_users - b < a->batch_rows ? a->n_users - b : a->batch_rows;
    const int32_t* users = a->d_users + b;
    const float* P = a->d_P;
    if (a->use_filter == 2)
      NR_TRY(nrhip_score_filter_i8_tilemax(P, a->ldp, users, rows, a->cols, a->d, a->d_M, a->mld, a->d_eps,
                                           a->d_filter_ws, a->filter_ws_bytes, a->batch_rows, stream));
    else if (a->use_filter)
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
    NR_TRY(nr_colsum_f64_flags(a->d_out, out_ld, a->n_users, out_ld, a->d_sums, a->d_colsum_ws, a->colsum_ws_bytes,
                               a->d_flags, a->d_sums + out_ld, stream));
  }
  return NR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The narrow Mult-VAE step (MultiVAE.py:73-139, one sess.run((loss, optimizer))) as one native call: encode -> decoder
// loss + gradients without the logits slab -> the narrow layers' backward -> dW_q0 -> [loss means, L2 terms] -> dense
// TF-Adam on the eight variables.  The same entry points the Python step issued one by one; issued from C the step no
// longer depends on how fast the host enqueues eight calls with ~90 arguments (0.15 ms of kernels: on a slower host
// the Python-issued step measured 0.23 ms).
// ---------------------------------------------------------------------------------------------------------------
extern "C" int nrhip_vae_step(const NrhipVaeStep* a, const int32_t* d_rows, int batch, float anneal, float keep,
                              float adam_alpha, uint64_t step, int want_loss, int apply, void* stream) {
  NR_REQUIRE(a && d_rows && batch >= 1, NR_ERR_ARG, "vae_step: bad arguments");
  float* const* P = a->P;
  float* const* G = a->G;
  // order of the eight variables: Wq0, bq0, Wq1, bq1, Wp0, bp0, Wp1t, bp1
  NR_TRY(nrhip_vae_encode(a->d_indptr, a->d_indices, d_rows, batch, a->h, a->z, P[0], P[1], P[2], P[3], P[4], P[5],
                          a->act, keep, a->d_drop_given, a->d_eps_given, 1.0f, a->seed, step, a->d_h0val, a->d_H1,
                          a->d_MU, a->d_LOGVAR, a->d_EPSSTD, a->d_ZS, a->d_G1, a->d_KLb, stream));
  NR_TRY(nrhip_vae_decoder_fused(batch, a->n_items, a->h, a->d_G1, P[6], P[7], a->d_indptr, a->d_indices, d_rows,
                                 a->d_nll, G[6], G[7], a->d_dG1, a->d_ws, a->ws_bytes, nullptr, stream));
  NR_TRY(nrhip_vae_mid_backward(batch, a->h, a->z, a->act, anneal, a->d_dG1, a->d_G1, a->d_H1, a->d_MU, a->d_LOGVAR,
                                a->d_EPSSTD, a->d_ZS, P[4], P[2], a->d_DA3, a->d_DH2, a->d_DA1, G[4], G[5], G[2], G[3],
                                G[1], stream));
  NR_TRY(nrhip_vae_dwq0(a->d_indptr, a->d_indices, d_rows, batch, a->h, a->d_h0val, a->d_DA1, G[0], stream));
  if (want_loss) NR_TRY(nrhip_mean2_f32(a->d_nll, a->d_KLb, batch, a->d_stats, stream));
  if (a->reg != 0.0f) {
    const int wi[4] = {0, 2, 4, 6};                            // the four weight matrices (MultiVAE.py:126-131)
    if (want_loss) NR_CHECK_HIP(hipMemsetAsync(a->d_regsum, 0, sizeof(double), (hipStream_t)stream));
    for (int k = 0; k < 4; ++k) {
      if (want_loss) NR_TRY(nrhip_sumsq_accumulate(P[wi[k]], a->sizes[wi[k]], a->d_regsum, stream));
      NR_TRY(nrhip_axpy(2.0f * a->reg, P[wi[k]], G[wi[k]], a->sizes[wi[k]], stream));
    }
  }
  if (!apply) return NR_OK;
  const int32_t clear[8] = {1, 0, 0, 0, 0, 0, 0, 0};            // dW_q0 is accumulated by row: zero again after the step
  return nrhip_adam_dense_tf_multi(8, a->P, a->M, a->V, a->G, a->sizes, clear, adam_alpha, a->beta1, a->beta2,
                                   a->adam_eps, stream);
}
