/* neurec_hip.h — C ABI of libneurec_hip.so, the MI355X (gfx950) engine for the
 * NeuRec embedding hot path.
 *
 * Conventions
 *   - Every pointer named d_* is a DEVICE pointer owned by the caller (e.g. a
 *     PyTorch-ROCm tensor's data_ptr()).  The library never allocates or frees
 *     caller-visible memory; scratch space is passed in as an explicit
 *     workspace whose size comes from the matching *_workspace_bytes query.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it
 *     and the call returns without synchronising.
 *   - Return value: 0 on success, otherwise one of NRHIP_ERR_*; a message is
 *     available from nrhip_last_error() (thread-local).
 *   - int = 32 bit, float = IEEE fp32, exactly as the reference asserts at
 *     import (util/cython/tools.pyx:7-27).
 *   - Item/user id arrays are int32, CSR index pointers are int64 where the
 *     number of interactions may exceed 2^31 (training CSR, adjacency) and
 *     int32 otherwise.
 *
 * Each entry point names the reference interface it stands in for
 * (paths relative to the NeuRec tree, wubinzzu/NeuRec @ v1).
 */
#ifndef NEUREC_HIP_H
#define NEUREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRHIP_OK 0
#define NRHIP_ERR_ARG 1
#define NRHIP_ERR_UNSUPPORTED 2
#define NRHIP_ERR_HIP 3
#define NRHIP_ERR_WORKSPACE 4

#define NRHIP_ABI_VERSION 4
#define NRHIP_MAX_TOPK 128 /* largest top_k the selection kernels accept */

/* ---- library ------------------------------------------------------------ */
int nrhip_abi_version(void);
const char* nrhip_last_error(void);
/* device facts used by bench/roofline reporting */
int nrhip_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* name, int name_len);

/* ---- evaluator ----------------------------------------------------------
 * Replaces: cpp_evaluate_matrix / eval_one_user
 *           (evaluator/backend/cpp/include/evaluate.h:23-72),
 *           metric functions (evaluator/backend/cpp/include/metric.h:17-117),
 *           the -inf train mask loop (evaluator/backend/cpp/uni_evaluator.py:140-143),
 *           arg_top_k_2d (util/cython/include/arg_topk.h:15-45).
 */

/* Scratch needed by nrhip_eval_scores / nrhip_arg_topk for `rows` rows. */
int nrhip_eval_workspace_bytes(int rows, int top_k, size_t* bytes);

/* scores[r][train items of users[r]] = -inf, in place
 * (uni_evaluator.py:140-143).  tr_indptr has one entry per user id + 1. */
int nrhip_mask_train(float* d_scores, int64_t ld, const int32_t* d_users, int rows, int cols,
                     const int64_t* d_tr_indptr, const int32_t* d_tr_indices, void* stream);

/* Per row: rank the `cols` scores exactly like eval_one_user (partial sort of
 * min(2*top_k, cols) indices by descending score, cut to top_k, libstdc++ tie
 * behaviour included), then write n_metric*top_k cumulative metric values,
 * metric-major (evaluate.h:43-47).  Truth of row r is the ascending id list
 * d_truth_indices[d_truth_indptr[t] .. d_truth_indptr[t+1]) with
 * t = d_users ? d_users[r] : r.  metric ids: 1 Precision 2 Recall 3 MAP
 * 4 NDCG 5 MRR (metric.h:111-117).  d_topk_out (optional, may be NULL)
 * receives the top_k ranked item ids per row.  d_n_exact (optional) receives
 * the number of rows that needed the exact tie path. */
int nrhip_eval_scores(const float* d_scores, int64_t ld, int rows, int cols,
                      const int32_t* d_users, const int64_t* d_truth_indptr,
                      const int32_t* d_truth_indices, const int32_t* metric_ids_host,
                      int n_metric, int top_k, float* d_out, int32_t* d_topk_out,
                      int32_t* d_n_exact, void* d_ws, size_t ws_bytes, void* stream);

/* nrhip_eval_scores without the top_k <= 128 limit (cpp_evaluate_matrix / eval_one_user take any K:
 * evaluator/backend/cpp/include/evaluate.h:23-50): one thread per row replays std::partial_sort_copy and the metric
 * loops sequentially.  Same arguments, same output layout; its own workspace query. */
int nrhip_eval_any_k_workspace_bytes(int rows, int cols, int top_k, size_t* bytes);
int nrhip_eval_scores_any_k(const float* d_scores, int64_t ld, int rows, int cols, const int32_t* d_users,
                            const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                            const int32_t* metric_ids_host, int n_metric, int top_k, float* d_out,
                            int32_t* d_topk_out, void* d_ws, size_t ws_bytes, void* stream);

/* Per row arg-top-K with std::partial_sort_copy(K) semantics
 * (arg_topk.h:15-25): d_out[rows][top_k] int32. */
int nrhip_arg_topk(const float* d_scores, int64_t ld, int rows, int cols, int top_k,
                   int32_t* d_out, int32_t* d_n_exact, void* d_ws, size_t ws_bytes,
                   void* stream);

/* Column sums of a [rows][cols] fp32 matrix in fp64 (the np.mean of
 * uni_evaluator.py:150-151 is sum/rows; the division is the caller's). */
int nrhip_colsum_workspace_bytes(int rows, int cols, size_t* bytes);
int nrhip_colsum_f64(const float* d_mat, int64_t ld, int rows, int cols, double* d_out,
                     void* d_ws, size_t ws_bytes, void* stream);

/* Scoring GEMM: S[r][i] = sum_k P[users[r]][k] * Q[i][k], fp32 MFMA, k
 * ascending fused-multiply-add chain.  Replaces np.matmul(user_embed,
 * item_embeddings.T) (model/general_recommender/MF.py:120-122) and the TF
 * matmul of LightGCN.py:118-119.  d_users may be NULL (rows 0..rows-1).
 * S has leading dimension lds >= cols; columns [cols, lds_pad) may be
 * written with zeros where lds_pad = min(lds, roundup(cols,64)). */
int nrhip_score_gemm_workspace_bytes(int rows, int cols, int d, size_t* bytes);
/* Prepare the item side once per evaluation (k-major copy of Q inside ws). */
int nrhip_score_gemm_prepare_items(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws,
                                   size_t ws_bytes, void* stream);
int nrhip_score_gemm(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols,
                     int d, float* d_S, int64_t lds, void* d_ws, size_t ws_bytes, void* stream);

/* ---- pruned full-rank evaluation (no score matrix) --------------------------------
 * Same results as nrhip_score_gemm + nrhip_mask_train + nrhip_eval_scores
 * (uni_evaluator.py:132-151 -> evaluate.h:23-72) without materialising the [rows][cols] scores:
 *   level 1  nrhip_score_tilemax: the scoring loop keeps, per user, only the maximum admissible
 *            score of every 32-item tile (train items and pad columns struck out in registers):
 *            d_M[rows][mld], mld even and >= 2*ceil(cols/64) (32-item tiles).  Item side prepared as
 *            for nrhip_score_gemm.
 *   level 2  nrhip_eval_tiles: the top_k+1 best items of a user lie in the top_k+1 tiles with the
 *            largest maxima; those tiles are rescored (same k-ascending fmaf chain, bit-identical),
 *            ranked and measured.  d_flag_out[r] = 1 marks rows whose ranking may depend on ties
 *            (inside the kept set, or between the two boundary tile maxima): their rows of d_out
 *            are provisional — recompute them through the full score path.
 * Needs 2*ceil(cols/64) >= top_k + 2 and top_k <= 62. */
int nrhip_score_tilemax(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols,
                        int d, const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                        float* d_M, int64_t mld, void* d_ws, size_t ws_bytes, void* stream);
/* Level 1 in two launches (the faster form): nrhip_score_tilemax with d_tr_indptr = d_tr_indices = NULL leaves the
 * train items IN the maxima (the scoring loop carries no cursors and no strike code), and nrhip_score_tilemax_fix
 * recomputes — same MFMA chain, same operand roles, bit-identical values — only the (user, tile) pairs that hold a
 * train item, from a plan built once per train matrix: pairs sorted by 32-item tile (d_tile_ptr[n_tiles32 + 1],
 * d_plan_user[e], d_plan_mask[e]: bit r = item 32*tile + r is a train item of that user), cut into chunks of <= 32
 * pairs of one tile (d_chunk_tile[c], d_chunk_begin[c]).  d_row_of[user] = the user's row in the evaluation order
 * (-1: not evaluated; NULL: row = user); M holds rows [row_lo, row_lo + rows).  Afterwards M equals the one-launch
 * form's, bit for bit (uni_evaluator.py:132-140: ranking_score[train items] = -inf, as a property of the maxima). */
int nrhip_score_tilemax_fix(const float* d_P, int64_t ldp, int d, int cols, const int32_t* d_chunk_tile,
                            const int64_t* d_chunk_begin, int n_chunks, const int64_t* d_tile_ptr,
                            const int32_t* d_plan_user, const uint32_t* d_plan_mask, const int32_t* d_row_of,
                            int row_lo, int rows, float* d_M, int64_t mld, const void* d_ws, size_t ws_bytes,
                            void* stream);

/* The strike plan itself, built on the device (r05: was construction-time torch code): pairs (user, 32-item tile) that
 * hold a train item of the user — uni_evaluator.py:132-140 strikes those items of every batch on the host — grouped by
 * tile, chunks of <= 32 pairs.  Capacities: d_plan_user / d_plan_mask nnz entries; d_tile_ptr n_tiles + 1 with
 * n_tiles = 2 * ceil(cols / 64); d_chunk_tile / d_chunk_begin nnz / 32 + n_tiles + 1; d_counts[2] = {pairs, chunks};
 * d_ws 8 * n_tiles bytes.  The order of the pairs inside a tile is unspecified (each pair is independent). */
int nrhip_tile_strike_plan(const int64_t* d_indptr, const int32_t* d_indices, int n_users, int cols,
                           int32_t* d_plan_user, uint32_t* d_plan_mask, int64_t* d_tile_ptr, int32_t* d_chunk_tile,
                           int64_t* d_chunk_begin, int32_t* d_counts, void* d_ws, size_t ws_bytes, void* stream);
/* Level 1 as a bounded filter on the bf16 matrix cores (csrc/score_bf16.hip; d <= 128).  The fp32 chain stays the
 * definition of every score that is ranked: this only SEARCHES for the tiles worth rescoring, 4-5x faster than the
 * fp32 MFMA loop.  x = hi + lo + r with hi, lo bf16 (round to nearest even), u.i ~= sum_k uh*ih + uh*il + ul*ih in
 * fp32 accumulators, and |approx - chain| <= kappa(d) * ||u|| * max_i ||i|| = d_eps[row] (kappa: nrhip_score_filter_kappa;
 * the derivation stands at the top of score_bf16.hip).  nrhip_score_filter_prepare_items splits the item table once
 * per evaluation (it replaces np.matmul's operand, MF.py:120-122); nrhip_score_filter_tilemax fills d_M like
 * nrhip_score_tilemax without train lists (nrhip_score_tilemax_fix must follow: it writes exact fp32 maxima, error 0)
 * and d_eps[rows].  nrhip_eval_tiles_bounded then certifies each row against its bound or flags it. */
int nrhip_score_filter_workspace_bytes(int rows, int cols, int d, size_t* bytes);
int nrhip_score_filter_kappa(int d, float* kappa);
int nrhip_score_filter_prepare_items(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws, size_t ws_bytes,
                                     int max_rows, void* stream);
int nrhip_score_filter_tilemax(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols, int d,
                               float* d_M, int64_t mld, float* d_eps, void* d_ws, size_t ws_bytes, int max_rows,
                               void* stream);
/* The same bounded filter on the int8 matrix cores (csrc/score_i8.hip; d <= 128): 15-bit fixed point — one scale per
 * user row, one for the whole item table, two int8 planes per entry, three v_mfma_i32_32x32x32_i8 products in exact
 * integer accumulators, the tile maximum taken on the integers.  Same outputs as nrhip_score_filter_tilemax, with a
 * ONE-SIDED contract (what nrhip_eval_tiles_bounded's certificate needs): d_M[r][t] is an upper-bound maximum — the
 * approximate maximum plus the tile's own share of the quantisation error, 0.525*su*sI*max_{i in t} sum|qi| — and
 * fp32 chain maximum of tile t <= d_M[r][t] + d_eps[r],   d_M[r][t] - that maximum <= d_eps[r] + 2 x the tile's share,
 * with d_eps[r] = su*sI*[0.525*sum|qu| + 0.27*d + 64*sum|lu|] + the chain's own rounding, DERIVED from the
 * quantisation (top of score_i8.hip) — no model of the matrix pipe is assumed.  d_eps[r] = NaN for rows (or item
 * tables) whose largest magnitude lies outside [2^-40, 2^40] or is not finite: every certificate fails for them and
 * the caller's fp32 path takes the row.  It replaces the same reference lines (np.matmul's operand, MF.py:120-122;
 * LightGCN.py:187-189). */
int nrhip_score_filter_i8_workspace_bytes(int rows, int cols, int d, size_t* bytes);
int nrhip_score_filter_i8_prepare_items(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws, size_t ws_bytes,
                                        int max_rows, void* stream);
int nrhip_score_filter_i8_tilemax(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols, int d,
                                  float* d_M, int64_t mld, float* d_eps, void* d_ws, size_t ws_bytes, int max_rows,
                                  void* stream);
int nrhip_eval_tiles_workspace_bytes(int rows, int top_k, size_t* bytes);
/* d_gemm_ws: the workspace nrhip_score_gemm_prepare_items filled (the rescoring reads its k-major
 * item copy, one coalesced 256-byte load per k and tile). */
int nrhip_score_gemm_items_kmajor(const void* d_ws, int cols, int d, const float** qt, int* ipad);
int nrhip_eval_tiles(const float* d_M, int64_t mld, const float* d_P, int64_t ldp,
                     const void* d_gemm_ws, int d, const int32_t* d_users, int rows, int cols,
                     const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                     const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                     const int32_t* metric_ids_host, int n_metric, int top_k, float* d_out,
                     int32_t* d_flag_out, void* d_ws, size_t ws_bytes, void* stream);
/* Level 2 for BOUNDED maxima (nrhip_score_filter_tilemax, then nrhip_score_tilemax_fix): the n_keep best tiles
 * (top_k + 1 <= n_keep <= 63) are rescored with the fp32 chain and ranked as in nrhip_eval_tiles; a row stands when its
 * top_k-th rescored score exceeds the largest maximum among the tiles NOT rescored by more than d_eps[row] — no item
 * outside the rescored tiles can then reach or tie the kept set in the fp32 chain — otherwise d_flag_out[row] = 1 and
 * the caller recomputes the row from a full fp32 score row (as for ties).  evaluate.h:23-50 stays the definition.
 * d_eps = NULL: the maxima are exact (nrhip_score_tilemax), n_keep = top_k + 1, nrhip_eval_tiles' rule.  The rescoring
 * runs grouped by tile (32 users of one tile per wave on the fp32 matrix cores: the item tile is read once per 32
 * users, not once per user) where the shape allows; same scores bit for bit. */
int nrhip_eval_tiles_bounded_workspace_bytes(int rows, int cols, int top_k, int n_keep, size_t* bytes);
int nrhip_eval_tiles_bounded(const float* d_M, int64_t mld, const float* d_eps, int n_keep, const float* d_P,
                             int64_t ldp, const void* d_gemm_ws, int d, const int32_t* d_users, int rows, int cols,
                             const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                             const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                             const int32_t* metric_ids_host, int n_metric, int top_k, float* d_out,
                             int32_t* d_flag_out, void* d_ws, size_t ws_bytes, void* stream);

/* The pruned evaluation of a whole user list as one call: the batch loop of UniEvaluator.evaluate
 * (evaluator/backend/cpp/uni_evaluator.py:101-157) over nrhip_score_filter_tilemax / nrhip_score_filter_i8_tilemax
 * (use_filter = 1 / 2) or nrhip_score_tilemax,
 * nrhip_score_tilemax_fix and nrhip_eval_tiles_bounded, then (d_sums != NULL) the fp64 column sums of d_out and the
 * number of flagged rows in d_sums[n_metric*top_k] (and, behind it, how many of them failed their certificate: flag
 * bit 1; bit 0 = ties / bucket overflow) — one device->host copy brings the means and says whether any row
 * must be redone.  Pointers are device pointers except metric_ids (host). */
typedef struct nrhip_eval_pruned_args {
  const float* d_P; int64_t ldp;                 /* user factors [n_table_users][ldp] */
  const float* d_Q; int64_t ldq;                 /* item factors [cols][ldq] */
  int d, cols;
  const int32_t* d_users; int n_users, batch_rows;
  const int64_t* d_tr_indptr; const int32_t* d_tr_indices;          /* train CSR (struck items) */
  const int64_t* d_truth_indptr; const int32_t* d_truth_indices;    /* test CSR */
  const int32_t* d_chunk_tile; const int64_t* d_chunk_begin; int n_chunks;   /* strike plan (nrhip_score_tilemax_fix) */
  const int64_t* d_tile_ptr; const int32_t* d_plan_user; const uint32_t* d_plan_mask;
  const int32_t* d_row_of;                       /* user -> evaluation row (-1: not evaluated) */
  const int32_t* metric_ids; int n_metric, top_k, n_keep;
  int use_filter, prepare_items;                 /* bounded search: 0 none, 1 bf16, 2 int8 (d_filter_ws of that form) / (re)build the item-side copies first:
                                                  * 1 all of them; 2 (with a filter) all but the fp32 scoring loop's operand copy, which this call does not read —
                                                  * nrhip_score_gemm_prepare_items must run before nrhip_score_gemm / nrhip_score_tilemax use that workspace */
  void* d_gemm_ws; size_t gemm_ws_bytes;         /* nrhip_score_gemm_workspace_bytes(batch_rows, cols, d) */
  void* d_filter_ws; size_t filter_ws_bytes;     /* nrhip_score_filter_workspace_bytes(batch_rows, cols, d) */
  void* d_tiles_ws; size_t tiles_ws_bytes;       /* nrhip_eval_tiles_bounded_workspace_bytes(batch_rows, cols, top_k, n_keep) */
  float* d_M; int64_t mld;                       /* [batch_rows][mld] tile maxima */
  float* d_eps;                                  /* [batch_rows] */
  float* d_out;                                  /* [n_users][n_metric*top_k] */
  int32_t* d_flags;                              /* [n_users] */
  double* d_sums;                                /* [n_metric*top_k + 2] or NULL */
  void* d_colsum_ws; size_t colsum_ws_bytes;     /* nrhip_colsum_workspace_bytes(n_users, n_metric*top_k) */
} NrhipEvalPruned;
int nrhip_eval_pruned(const NrhipEvalPruned* args, void* stream);

/* The rows nrhip_eval_pruned flagged (ties that could change a metric, a certificate that did not hold, a full bucket),
 * ranked again from full fp32 score rows — nrhip_score_gemm + nrhip_mask_train + nrhip_eval_scores, the materialised
 * path, exact whatever the cause (uni_evaluator.py:132-151 -> evaluate.h:23-72) — and written over their rows of
 * ev->d_out, then (ev->d_sums != NULL) the column sums again: one call, no host round trip inside.  n_flagged is the
 * count nrhip_eval_pruned left in d_sums[n_metric*top_k] (the caller has read it); ev is that call's argument block,
 * unchanged.  The rows are collected in whatever order the device finds them (every row is independent). */
typedef struct nrhip_eval_redo_args {
  const NrhipEvalPruned* ev;
  int n_flagged;
  int reload_items;                              /* the fp32 scoring loop's item side first: 1 nrhip_score_gemm_prepare_items; 2 its operand
                                                  * copy only (ev ran with prepare_items = 2 on this table: the k-major copy stands) */
  float* d_scores; int64_t lds; int slab_rows;   /* score slab [slab_rows][lds], lds >= roundup(cols, 64), slab_rows <= ev->batch_rows */
  int32_t* d_rows;                               /* [n_flagged] out: the flagged rows */
  int32_t* d_row_users;                          /* [n_flagged] out: their users */
  int32_t* d_count;                              /* one word */
  float* d_fixed;                                /* [slab_rows][n_metric*top_k] */
  void* d_ws; size_t ws_bytes;                   /* nrhip_eval_workspace_bytes(slab_rows, top_k) */
} NrhipEvalRedo;
int nrhip_eval_redo(const NrhipEvalRedo* args, void* stream);

/* ---- sampler ------------------------------------------------------------
 * Replaces: PairwiseSampler.__iter__ = _sampling_negative_items +
 *           DataIterator(shuffle) (data/sampler.py:71-90,198-206;
 *           util/data_iterator.py:58-60,145-152) and the rejection loop of
 *           randint_choice (util/cython/random_choice.pyx:20-62).
 *
 * Train CSR: indptr[n_users+1] (int64), indices ascending per row; d_row_of[E]
 * is the user of each CSR position (users_list of sampler.py:24-39).
 * Output position p of the epoch stream holds triplet perm(p); with
 * shuffle == 0 perm is the identity (user-major order, as the reference with
 * shuffle=False).  d_neg_out has E*neg_num entries, row-major [E][neg_num]. */
int nrhip_sample_bpr_epoch(const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                           const int32_t* d_row_of, int64_t n_inter, int n_items, int neg_num,
                           uint64_t seed, uint64_t epoch, int shuffle, int64_t out_begin,
                           int64_t out_count, int32_t* d_users_out, int32_t* d_pos_out,
                           int32_t* d_neg_out, void* stream);

/* Epoch stream of training INSTANCES on the device — replaces the __iter__ of
 * PointwiseSampler (data/sampler.py:137-149), TimeOrderPointwiseSampler
 * (:269-282) and TimeOrderPairwiseSampler (:339-347): _sampling_negative_items
 * (:71-90) + the Python-list layout (:131-135, :141-143, :259-266) +
 * DataIterator(shuffle) (util/data_iterator.py:58-60,145-152), in one launch.
 *
 * Rows r = 0..R-1 in the iteration order of user_pos_dict; d_row_user[r] = user
 * id; d_seq[d_seq_ptr[r]..) the row's items as the dict holds them (time order
 * for the TimeOrder samplers); d_excl[d_excl_ptr[r]..) the same set ascending
 * (exclusion test).  Instance t = d_inst_ptr[r] + k (k < len_r - high_order,
 * _generative_time_order_positive_items, data/sampler.py:42-68; high_order 0:
 * _generate_positive_items, :24-39), d_inst_row[t] = r.  pointwise == 0: slot =
 * instance, d_items_out = positive, d_neg_out [count][neg_num].  pointwise != 0:
 * n_inst*(neg_num+1) slots, slot s = (c = s / n_inst, t = s % n_inst), c = 0 the
 * positive (label 1), else negative c-1 of instance t (label 0).  d_recent_out
 * [count][high_order] (may be NULL when high_order == 0).  Output position p of
 * the epoch carries slot perm(p); shuffle == 0: identity (the reference's list
 * order). */
int nrhip_sample_instances_epoch(const int64_t* d_seq_ptr, const int32_t* d_seq,
                                 const int64_t* d_excl_ptr, const int32_t* d_excl,
                                 const int64_t* d_inst_ptr, const int32_t* d_inst_row,
                                 const int32_t* d_row_user, int64_t n_inst, int high_order,
                                 int n_items, int neg_num, int pointwise, uint64_t seed,
                                 uint64_t epoch, int shuffle, int64_t out_begin, int64_t out_count,
                                 int32_t* d_users_out, int32_t* d_recent_out, int32_t* d_items_out,
                                 int32_t* d_neg_out, float* d_labels_out, void* stream);

/* batch_randint_choice(high, size, replace, p=None, exclusion)
 * (random_choice.pyx:64-89): request q draws size[q] =
 * d_out_offsets[q+1]-d_out_offsets[q] values into d_out[d_out_offsets[q] ..)
 * (d_out_offsets has n_req+1 entries, total = d_out_offsets[n_req]),
 * excluding the ascending, duplicate-free list
 * d_excl[d_excl_indptr[q] .. d_excl_indptr[q+1]).  replace == 0 additionally
 * forbids repeats inside one request. */
int nrhip_randint_choice_batch(int high, int n_req, int64_t total, const int64_t* d_out_offsets,
                               const int64_t* d_excl_indptr, const int32_t* d_excl, int replace,
                               uint64_t seed, uint64_t call_counter, int32_t* d_out, void* stream);

/* ---- BPR-MF training step -------------------------------------------------
 * Replaces one sess.run((loss, optimizer)) of MF.train_model
 * (model/general_recommender/MF.py:54-76,101; util/learner.py:9-10,19-22;
 * util/tool.py:216-217).
 *   plan:  the occurrences (row, position) of every batch of an epoch stream sorted
 *          by row then position — what makes the row-gradient sums deterministic:
 *          TF adds the IndexedSlices that hit one row in batch order
 *          (unsorted_segment_sum; the positive lookups' slices before the negative
 *          lookups'), and so do the kernels given this order.  d_third = NULL for
 *          pointwise (user, item) instances.  Batch k covers stream elements
 *          [k*batch, min((k+1)*batch, n_total)); its n_cls*len keys are written at
 *          d_plan_out + n_cls*k*batch (n_cls = 3 with d_third, else 2).  Item ids are
 *          offset by n_users in the keys.  Batches of up to (n_cls-1)*batch = 16384 item
 *          occurrences are sorted by one workgroup each in one launch; larger ones by a
 *          segmented multi-workgroup network (a few launches per call).
 *   grad:  gathers p_u,q_i,q_j; x=<p,q_i>-<p,q_j>; loss_b=softplus(-x);
 *          row gradients (duplicates summed in batch order) STORED into the rows of
 *          dense d_GP/d_GQ the batch touches (all other rows are left alone: keep them
 *          zero); d_work is scratch for 8*batch floats (per-triplet loss and l2 terms,
 *          reduced in a fixed order; then room for a plan); d_plan = this batch's slice
 *          of an nrhip_bpr_plan output, or NULL: the plan is then sorted on the spot
 *          (one more launch); d_loss2[0] = sum_b loss_b, d_loss2[1] = reg * sum_b l2_b,
 *          the two addends of MF.py:68-69 (the fetched loss is their sum).
 *   adam:  TF-1.12 sparse Adam = all rows swept (see nr_core.h), and the
 *          gradient buffer is cleared for the next step.               */
int nrhip_bpr_plan(const int32_t* d_users, const int32_t* d_items, const int32_t* d_third,
                   int64_t n_total, int batch, int n_users, uint64_t* d_plan_out, void* stream);
int nrhip_bpr_mf_grad(const float* d_P, const float* d_Q, int d, int n_users,
                      const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                      float reg, float* d_GP, float* d_GQ, float* d_work, float* d_loss2,
                      const uint64_t* d_plan, void* stream);
int nrhip_adam_sparse_tf(float* d_var, float* d_m, float* d_v, float* d_grad, int64_t n,
                         float alpha, float beta1, float beta2, float eps, void* stream);
/* The same optimiser, exactly, without the sweep (SURVEY.md H2 "lazy replay"): d_last[row] = last
 * step applied to the row (int32, zero at step 0); when step t touches a row its missed
 * zero-gradient steps are replayed in registers with the step sizes d_alpha_tab[s] (fp32, 1-based,
 * >= t + 1 entries, each made exactly like the `alpha` of nrhip_adam_sparse_tf), then step t is
 * applied with the row's gradient (cleared afterwards).  d_plan / n_occ: the batch's nrhip_bpr_plan
 * slice — rows must be rows of this [n_rows][d] table (user rows first, item rows offset by the plan's
 * n_users).  Every call also brings rows r = t (mod period) up to step t, so no row is ever more
 * than `period` steps behind.  d_next_plan / n_next_occ (optional): the NEXT step's batch plan — its
 * rows are brought to step t too, so that the next gradient kernel replays nothing.
 * d_plan = NULL, n_occ = 0, period = 1: flush every row to step t — required before the table (or m,
 * v) is read; afterwards the buffers are bit-identical to t sweeps. */
int nrhip_adam_sparse_tf_lazy(float* d_var, float* d_m, float* d_v, float* d_grad, int32_t* d_last,
                              const int32_t* d_stamp, int64_t n_rows, int d, const uint64_t* d_plan,
                              int n_occ, const uint64_t* d_next_plan, int n_next_occ,
                              const float* d_alpha_tab, int t, int period, float beta1, float beta2,
                              float eps, void* stream);
/* The gradient half of a lazy step: nrhip_bpr_mf_grad on the one-allocation table [n_users +
 * n_items][d] (d_G likewise), every gathered row first brought to step t - 1 in registers (nothing
 * written back), the batch's rows stamped d_stamp[row] = t (int32 per row; the optimiser call of the
 * same step takes it to tell batch rows from scheduled rows).  d_plan is required. */
int nrhip_bpr_mf_grad_lazy(const float* d_table, const float* d_m, const float* d_v,
                           const int32_t* d_last, const float* d_alpha_tab, int32_t* d_stamp, int t,
                           float beta1, float beta2, float eps, int d, int n_users,
                           const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                           float reg, float* d_G, float* d_work, float* d_loss2, const uint64_t* d_plan,
                           void* stream);
/* TF-1.12 dense ApplyAdam; d_grad cleared afterwards when clear_grad != 0. */
int nrhip_adam_dense_tf(float* d_var, float* d_m, float* d_v, float* d_grad, int64_t n,
                        float alpha, float beta1, float beta2, float eps, int clear_grad,
                        void* stream);

/* Dense ApplyAdam with gradient = d_grad_a + d_grad_b (added in fp32, one rounding); the
 * gradient buffers are read-only. */
int nrhip_adam_dense_tf2(float* d_var, float* d_m, float* d_v, const float* d_grad_a,
                         const float* d_grad_b, int64_t n, float alpha, float beta1, float beta2,
                         float eps, void* stream);
/* Dense ApplyAdam on up to 32 tensors in one launch (a model's weight matrices and biases:
 * util/learner.py:9-10 applied to every trainable of NGCF.py:91-110 / MultiVAE.py:47-70).  Host
 * arrays of device pointers and lengths; clear_grad_host (optional) as in nrhip_adam_dense_tf. */
int nrhip_adam_dense_tf_multi(int n_tensors, float* const* d_vars_host, float* const* d_ms_host,
                              float* const* d_vs_host, float* const* d_grads_host,
                              const int64_t* sizes_host, const int32_t* clear_grad_host, float alpha,
                              float beta1, float beta2, float eps, void* stream);
/* Row-sparse helpers over a list of row ids of a [*, d] buffer (repeats allowed):
 * dst[row] = src[row] / denom; and zeroing of the listed rows of up to four buffers plus a
 * per-row byte flag (any of them may be NULL). */
int nrhip_rows_div(const int32_t* d_rows, int n_listed, int d, const float* d_src, float denom,
                   float* d_dst, void* stream);
int nrhip_rows_clear(const int32_t* d_rows, int n_listed, int d, float* d_b0, float* d_b1,
                     float* d_b2, float* d_b3, uint8_t* d_flag, void* stream);

/* ---- sparse adjacency x embedding (LightGCN / NGCF propagation) ----------
 * Replaces tf.sparse_tensor_dense_matmul(adj, ego)
 * (model/general_recommender/LightGCN.py:140, NGCF.py:176) and its autodiff
 * transpose.  Y[r] = sum_j vals[j]*X[indices[j]] over the CSR row r in
 * ascending j (product and sum rounded separately), then
 *   Y[r] += addend[r]            if d_addend != NULL
 *   d_sum_out[r] = d_sum_in[r]+Y if d_sum_out != NULL (running layer sum,
 *                                LightGCN.py:146-147)
 * A plan splits long rows into segments; build it once per graph. */
int nrhip_spmm_plan_bytes(int64_t n_rows, int64_t nnz, size_t* bytes);
/* h_indptr is a HOST pointer (n_rows+1 entries): the segment plan is computed
 * on the host, uploaded into the caller's device buffer d_plan_buf, and a
 * host-side handle describing it is returned in *plan_out.  item_rows /
 * item_nnz bound the whole-row work items (0 = tuned defaults; at most 32 rows
 * and 256 non-zeros).  split_row (0 = none) is a locality hint for bipartite graphs: rows
 * below it (user nodes) and from it on (item nodes) gather from disjoint halves of X and are
 * scheduled on disjoint halves of the XCDs. */
int nrhip_spmm_plan_create(const int64_t* h_indptr, int64_t n_rows, int item_rows, int item_nnz,
                           int64_t split_row, void* d_plan_buf, size_t plan_bytes, void* stream,
                           void** plan_out);
int nrhip_spmm_plan_destroy(void* plan);
int nrhip_spmm_plan_info(const void* plan, int64_t* n_work_items, int64_t* n_split_rows);
int nrhip_spmm_workspace_bytes(const void* plan, int d, size_t* bytes);
/* d_Y may be NULL when only the running sum is wanted. */
int nrhip_spmm_csr(const void* plan, const int64_t* d_indptr, const int32_t* d_indices,
                   const float* d_vals, const float* d_X, int d, float* d_Y,
                   const float* d_addend, const float* d_sum_in, float* d_sum_out, void* d_ws,
                   size_t ws_bytes, void* stream);

/* The same product with work skipped (d >= 64; either mask may be NULL, not both):
 *   d_x_row_nonzero[c] == 0 promises X[c][:] == 0: those terms are skipped instead of
 *     gathered (first backward hop of a training step: dLoss/dE* is non-zero on the batch
 *     rows only);
 *   d_y_row_wanted[r] == 0 says output row r is not needed: it is left untouched (last
 *     forward hop of a training step: the loss reads E* on the batch rows only).
 * Produced rows are bit-identical to nrhip_spmm_csr's. */
int nrhip_spmm_csr_masked(const void* plan, const int64_t* d_indptr, const int32_t* d_indices,
                          const float* d_vals, const float* d_X, const uint8_t* d_x_row_nonzero,
                          const uint8_t* d_y_row_wanted, int d, float* d_Y, const float* d_addend,
                          const float* d_sum_in, float* d_sum_out, void* d_ws, size_t ws_bytes,
                          void* stream);
/* Only the listed rows of A·X (+ epilogue) are produced; other rows of the outputs are left
 * untouched.  d_rows may repeat; d_sum_out must not alias d_sum_in.  d in {64,128,256}. */
/* Persistent lane-group schedule for d = 16 / 32 / 64 (128 / 256 on request) (spmm_blocked.hip): one workgroup per CU
 * owns a LIST of rows (4·d-byte accumulators in LDS) — since r05 the rows are dealt to the workgroups by cost
 * (longest first, each to the least-loaded workgroup with a free accumulator), so the balance does not depend on how
 * the nodes are numbered (r01-r04's contiguous runs carried 1.9x the mean cost in their slowest workgroup when
 * popular items cluster in id); a d/4-lane group walks one row with 16-byte
 * loads, so a load instruction moves 4 / 2 / 1 rows; sub-lists longer than seg_len are cut into segments whose
 * partials are added in segment order.  Optional column blocking (block_bytes) cuts the gathered
 * table into L2-sized windows walked phase by phase.  Same contract and masks as nrhip_spmm_csr /
 * _masked; rows of <= seg_len non-zeros are bit-identical to the sequential order.  Attach it to
 * an SpMM plan (per d) and nrhip_spmm_csr / _masked / the step drivers use it for that d.
 * 0 selects a default for block_bytes (no blocking), n_workgroups (one per CU),
 * waves_per_wg (16), seg_len (64), r_max / p_max (LDS accumulators).  NR_ERR_UNSUPPORTED when the
 * matrix does not fit the schedule (the work-item kernel remains). */
int nrhip_spmm_blocked_plan_bytes(int64_t n_rows, int64_t nnz, int d, size_t* bytes);
int nrhip_spmm_blocked_plan_create(const int64_t* h_indptr, const int32_t* h_indices,
                                   int64_t n_rows, int64_t split_row, int d, int64_t block_bytes,
                                   int n_workgroups, int waves_per_wg, int seg_len, int r_max,
                                   int p_max, void* d_plan_buf, size_t plan_bytes, void* stream,
                                   void** plan_out);
int nrhip_spmm_blocked_plan_destroy(void* plan);
int nrhip_spmm_blocked_plan_info(const void* plan, int* n_workgroups, int* n_phases,
                                 int64_t* n_entries, int64_t* n_split);
int nrhip_spmm_blocked_tune(int gathers_in_flight);
/* The plan OWNS a copy of the matrix's (column, value) pairs in its own row order (a workgroup's pairs are one
 * contiguous slice): nrhip_spmm_blocked_pack fills it from the CSR arrays — once per matrix, again whenever the
 * values change (NGCF's node dropout rewrites them every step); a product call that hands in arrays that were not
 * packed packs them first, on its stream.  CONTRACT: freshness is decided by the identity of the two pointers — a
 * caller that rewrites d_vals IN PLACE must call nrhip_spmm_blocked_pack again before the next product
 * (engine.SpmmCSR.values_changed does; NGCF's node dropout is the one such caller). */
int nrhip_spmm_blocked_pack(void* plan, const int32_t* d_indices, const float* d_vals, void* stream);
int nrhip_spmm_blocked(const void* plan, const int32_t* d_indices, const float* d_vals,
                       const float* d_X, float* d_Y, const float* d_addend, const float* d_sum_in,
                       float* d_sum_out, const uint8_t* d_x_row_nonzero,
                       const uint8_t* d_y_row_wanted, void* stream);
int nrhip_spmm_plan_attach_blocked(void* plan, const void* blocked_plan, int d);

/* Last backward hop of a step and the optimiser in one pass (LightGCN.py:130,140 fused):
 * d_var / d_m / d_v <- TF-1.12 ApplyAdam with the dense gradient (A·X + d_addend) + d_grad_b; the
 * product itself is never stored.  Needs the d = 64 lane-group schedule attached to the plan
 * (nrhip_spmm_plan_has_blocked(plan, 64) != 0), NRHIP_ERR_UNSUPPORTED otherwise. */
/* clear_consumed != 0 additionally zeroes the non-zero entries of d_addend / d_grad_b and the set
 * bytes of d_row_flag (may be NULL) as they are read — what nrhip_rows_clear would do after the
 * step (both buffers are row-sparse there).  With clear_consumed and a d_row_flag, the flags are
 * also a promise: rows of d_addend / d_grad_b whose flag byte is 0 are all zero (they are not
 * read). */
int nrhip_spmm_csr_adam(const void* plan, const int32_t* d_indices, const float* d_vals,
                        const float* d_X, int d, float* d_addend, float* d_grad_b, float* d_var,
                        float* d_m, float* d_v, float alpha, float beta1, float beta2, float eps,
                        int clear_consumed, uint8_t* d_row_flag, void* stream);
int nrhip_spmm_blocked_adam(const void* plan, const int32_t* d_indices, const float* d_vals,
                            const float* d_X, float* d_addend, float* d_grad_b, float* d_var,
                            float* d_m, float* d_v, float alpha, float beta1, float beta2,
                            float eps, int clear_consumed, uint8_t* d_row_flag, void* stream);
int nrhip_spmm_plan_has_blocked(const void* plan, int d);   /* 1 / 0, not a status code */

/* Last forward hop of a step with the layer sum completed on the way (LightGCN.py:143-146 on the
 * batch rows): for rows with d_y_row_wanted[r] != 0,
 *   d_sum_out[r] = ((d_sum_in[r] + d_layer_a[r]) + d_layer_b[r]) + (A·X)[r]
 * (d_layer_a / d_layer_b optional, NULL = term absent; other rows of d_sum_out untouched).  The
 * full hops before it then need no running-sum streams.  Same additions in the same order as
 * nrhip_spmm_csr with sum_in / sum_out hop by hop.  Needs the d = 64 wanted-rows schedule
 * (nrhip_spmm_plan_has_wanted(plan, 64) != 0), NRHIP_ERR_UNSUPPORTED otherwise. */
int nrhip_spmm_csr_wanted_layers(const void* plan, const int32_t* d_indices, const float* d_vals,
                                 const float* d_X, int d, const float* d_sum_in,
                                 const float* d_layer_a, const float* d_layer_b, float* d_sum_out,
                                 const uint8_t* d_y_row_wanted, void* stream);
int nrhip_spmm_blocked_wanted_layers(const void* plan, const int32_t* d_indices, const float* d_vals,
                                     const float* d_X, const float* d_sum_in, const float* d_layer_a,
                                     const float* d_layer_b, float* d_sum_out,
                                     const uint8_t* d_y_row_wanted, void* stream);
/* The same hop told the batch instead of flags: wanted rows = d_users | n_users + d_pos |
 * n_users + d_neg.  It does nrhip_lightgcn_mark_batch's job on the way (one launch less per step):
 * d_row_flag (zero on entry) gets 1 on those rows, d_rows_out (optional) the 3*batch rows.  Needs
 * nrhip_spmm_plan_has_wanted(plan, 64) == 2 (a matrix of <= 131072 rows: one bit per row in LDS). */
int nrhip_spmm_csr_wanted_batch(const void* plan, const int32_t* d_indices, const float* d_vals,
                                const float* d_X, int d, const float* d_sum_in, const float* d_layer_a,
                                const float* d_layer_b, float* d_sum_out, const int32_t* d_users,
                                const int32_t* d_pos, const int32_t* d_neg, int batch, int n_users,
                                uint8_t* d_row_flag, int32_t* d_rows_out, void* stream);
int nrhip_spmm_blocked_wanted_batch(const void* plan, const int32_t* d_indices, const float* d_vals,
                                    const float* d_X, const float* d_sum_in, const float* d_layer_a,
                                    const float* d_layer_b, float* d_sum_out, const int32_t* d_users,
                                    const int32_t* d_pos, const int32_t* d_neg, int batch, int n_users,
                                    uint8_t* d_row_flag, int32_t* d_rows_out, void* stream);
/* 0: no wanted-rows schedule, 1: flag form only, 2: flag and batch forms (not status codes) */
int nrhip_spmm_plan_has_wanted(const void* plan, int d);
int nrhip_spmm_blocked_has_wanted(const void* blocked_plan);

/* Chunked propagation hop of the row-sharded engine (neurec_amd/sharded.py; LightGCN.py:132-149 with the operand
 * arriving in rank-ordered chunks — SURVEY 8e "overlap layer-k comm with layer-k local SpMM").  One launch per
 * operand chunk (columns < x_split: the rank's own block d_X; the rest: the received chunk d_X2) carries every
 * (virtual) row's accumulator on, so a row's sum stays ONE ascending-column chain; the
 * finish pass combines a hub row's 256-non-zero segments in segment order and applies nrhip_spmm_csr's epilogue. */
int nrhip_spmm_csr_carry(const void* plan, const int64_t* d_indptr, const int32_t* d_indices, const float* d_vals,
                         const float* d_X, const float* d_X2, int x_split, int d, float* d_Yv, int has_carry,
                         const uint8_t* d_row_mask, void* stream);
int nrhip_spmm_chunks_finish(const int32_t* d_first_vrow, int64_t n_rows, const float* d_Yv, int d, float* d_Y,
                             const float* d_addend, const float* d_sum_in, float* d_sum_out,
                             const uint8_t* d_row_mask, void* stream);

/* Reduced-exchange hop of the row-sharded engine (neurec_amd/sharded.py hop="reduce"; LightGCN.py:132-149 with the
 * item rows' sums formed as per-rank partials over each rank's OWN user rows): d_parts[world][n_rows][d]
 * holds the partial rows an owner received, rank-major; Y[r] = parts[0][r] + parts[1][r] + ... in rank order (one
 * fixed association), then nrhip_spmm_csr's epilogue (addend, running sum).  d_row_mask (optional): 0 = row left
 * untouched.  d a multiple of 4. */
int nrhip_partials_sum_rows(const float* d_parts, int world, int64_t n_rows, int d, float* d_Y, const float* d_addend,
                            const float* d_sum_in, float* d_sum_out, const uint8_t* d_row_mask, void* stream);

int nrhip_spmm_csr_rows(const int64_t* d_indptr, const int32_t* d_indices, const float* d_vals,
                        const float* d_X, int d, const int32_t* d_rows, int n_listed, float* d_Y,
                        const float* d_addend, const float* d_sum_in, float* d_sum_out,
                        void* stream);

/* ---- LightGCN BPR head ----------------------------------------------------
 * Replaces the lookups + create_bpr_loss of LightGCN.py:99-104,156-166 and
 * their gradient.  d_Esum is the running layer sum (E* = Esum/(L+1));
 * user rows first, item rows offset by n_users.  Stores, on the rows the batch touches
 * (duplicates summed in batch order, see nrhip_bpr_plan; other rows are left alone),
 *   d_Gstar  = dLoss/dE*      (dense [N][d], zero elsewhere)
 *   d_Greg   = reg * E0 rows  (dense [N][d], zero elsewhere)
 * and writes the two scalars mf_loss, emb_loss into d_loss2[0..1] (d_loss2 may be NULL:
 * the reference never fetches LightGCN's loss during training, LightGCN.py:178).
 * d_work: 8*batch floats; d_plan: the batch's plan or NULL (sorted on the spot). */
int nrhip_lightgcn_bpr_grad(const float* d_Esum, const float* d_E0, int n_users, int d,
                            int n_layers, const int32_t* d_users, const int32_t* d_pos,
                            const int32_t* d_neg, int batch, float reg, float* d_Gstar,
                            float* d_Greg, float* d_work, float* d_loss2, const uint64_t* d_plan,
                            void* stream);

/* Same head, storing dLoss/dE* already divided by (n_layers+1) into d_H (zero elsewhere) —
 * the H = Gstar/(L+1) the backward hops start from.  Only for n_layers+1 a power of two, where
 * dividing every term is bit-identical to dividing the sum (NRHIP_ERR_ARG otherwise). */
int nrhip_lightgcn_bpr_grad_h(const float* d_Esum, const float* d_E0, int n_users, int d,
                              int n_layers, const int32_t* d_users, const int32_t* d_pos,
                              const int32_t* d_neg, int batch, float reg, float* d_H,
                              float* d_Greg, float* d_work, float* d_loss2, const uint64_t* d_plan,
                              void* stream);

/* Node rows touched by a batch: d_rows_out[3*batch] = users | n_users+pos | n_users+neg and
 * d_row_flag[those rows] = 1 (d_row_flag: n_nodes bytes, zero on entry). */
int nrhip_lightgcn_mark_batch(const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                              int batch, int n_users, int32_t* d_rows_out, uint8_t* d_row_flag,
                              void* stream);

/* ---- whole training steps issued natively -----------------------------------
 * One call enqueues every kernel of a step (the reference's sess.run(opt) is one
 * call into its native runtime too: LightGCN.py:178, MF.py:101).  The context only
 * records caller-owned device pointers; nothing is allocated on the device. */
typedef struct nrhip_lightgcn_buffers {
  const void* plan;          /* SpMM plan of the adjacency A              */
  const void* plan_t;        /* plan of A^T (== plan when A is symmetric) */
  const int64_t* indptr;  const int32_t* indices;  const float* vals;      /* A   */
  const int64_t* indptr_t; const int32_t* indices_t; const float* vals_t;  /* A^T */
  float* E0; float* m; float* v;            /* [n_nodes][d] embeddings + Adam moments */
  float* Ea; float* Eb; float* Esum; float* Esum_rows;   /* layer buffers          */
  float* Gstar; float* Greg; float* H; float* Ga; float* Gb;   /* gradient buffers   */
  int32_t* batch_rows;       /* 3*max_batch                      */
  uint8_t* row_flag;         /* n_nodes bytes, zero between steps */
  float* terms;              /* 8*max_batch floats (loss terms + room for a batch plan) */
  void* spmm_ws; size_t spmm_ws_bytes;
  int n_users; int n_nodes; int d; int n_layers; int max_batch;
  float reg;
} nrhip_lightgcn_buffers;
int nrhip_lightgcn_ctx_create(const nrhip_lightgcn_buffers* bufs, void** ctx_out);
int nrhip_lightgcn_ctx_destroy(void* ctx);
/* alpha = lr*sqrt(1-b2^t)/(1-b1^t) in fp32 (the caller keeps the running powers).
 * d_loss2 may be NULL (loss not fetched).  d_plan: this batch's nrhip_bpr_plan slice (keys built
 * with n_users = the context's) or NULL. */
int nrhip_lightgcn_step(void* ctx, const int32_t* d_users, const int32_t* d_pos,
                        const int32_t* d_neg, int batch, const uint64_t* d_plan, float alpha,
                        float beta1, float beta2, float eps, float* d_loss2, void* stream);

/* Column-sharded tables (LightGCN.py:132-166 when every rank holds d of the D embedding columns — the propagation,
 * the gradient rows and ApplyAdam are column-wise, so the only quantity that needs all columns is the head's inner
 * products): the step cut there.  _fwd: forward hops + this rank's partial products d_partials[3*batch] =
 * (<e_u,e_i>, <e_u,e_j>, l2 term) per triplet; the caller all-gathers them, nrhip_partials_sum adds them in rank
 * order; _bwd: head with the summed d_given, backward hops, ApplyAdam.  Every rank steps on the SAME global batch. */
int nrhip_lightgcn_step_colshard_fwd(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                                     int batch, float* d_partials, void* stream);
int nrhip_lightgcn_step_colshard_bwd(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                                     int batch, const uint64_t* d_plan, const float* d_given, float alpha,
                                     float beta1, float beta2, float eps, float* d_loss2, void* stream);
int nrhip_lightgcn_partial_dots(const float* d_Esum, const float* d_E0, int n_users, int d, int n_layers,
                                const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                                float* d_out, void* stream);
int nrhip_partials_sum(const float* d_parts, int world, int n, float* d_given, void* stream);
int nrhip_lightgcn_bpr_grad_given(const float* d_Esum, const float* d_E0, int n_users, int d, int n_layers,
                                  const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                                  float reg, float* d_Gstar, float* d_Greg, float* d_work, float* d_loss2,
                                  const uint64_t* d_plan, const float* d_given, int divided, void* stream);
/* The same step cut at its one exchange point (multi-GPU): _grad leaves this rank's total
 * dLoss/dE0 in d_grad_out ([n_nodes][d]); the caller sums it over ranks (RCCL all-reduce);
 * _apply runs Adam on the summed gradient. */
int nrhip_lightgcn_step_grad(void* ctx, const int32_t* d_users, const int32_t* d_pos,
                             const int32_t* d_neg, int batch, const uint64_t* d_plan,
                             float* d_loss2, float* d_grad_out, void* stream);
int nrhip_lightgcn_step_apply(void* ctx, float* d_grad, float alpha, float beta1, float beta2,
                              float eps, void* stream);

typedef struct nrhip_mf_buffers {
  float* P; float* Q; float* mP; float* vP; float* mQ; float* vQ; float* GP; float* GQ;
  float* terms;              /* 8*max_batch floats */
  int n_users; int n_items; int d; int max_batch;
  float reg;
  /* lazy sparse Adam (nrhip_adam_sparse_tf_lazy): last != NULL selects it; needs P|Q, mP|mQ, vP|vQ,
   * GP|GQ each one [n_users + n_items][d] allocation */
  int32_t* last; int32_t* stamp; const float* alpha_tab; int alpha_len; int lazy_period;
  /* one-launch step (nrhip_bpr_mf_step_fused): tw != NULL selects it; P / mP / vP then each hold TWO
   * copies of the [n_users + n_items][d] array (copy 1 at + rows * d), tw = int32 [rows][2] ({0, -1} at start),
   * inb = int32 [rows] (zero at start); last / stamp / GP / GQ are not used */
  int32_t* tw; int32_t* inb;
} nrhip_mf_buffers;
int nrhip_mf_ctx_create(const nrhip_mf_buffers* bufs, void** ctx_out);
int nrhip_mf_ctx_destroy(void* ctx);
/* step_index: 1-based index of this optimiser step (lazy mode: alpha must equal alpha_tab[step_index]);
 * d_next_plan / next_batch: the following batch's plan and length, or NULL / 0 (lazy mode: its rows are
 * brought up to date by this step's optimiser launch) */
int nrhip_mf_step(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                  int batch, const uint64_t* d_plan, const uint64_t* d_next_plan, int next_batch,
                  int step_index, float alpha, float beta1, float beta2, float eps, float* d_loss2,
                  void* stream);
/* The batch loop of MF.train_model (model/general_recommender/MF.py:95-103) over the consecutive batches of
 * one epoch stream (n_total triplets, `batch` per step, the last one short): step k runs on triplets
 * [k*batch, ...), plan d_plans + 3*k*batch (layout of nrhip_bpr_plan(n_total, batch); NULL: sorted per step),
 * step index first_step_index + k, step size h_alpha[k] (HOST array), loss pair d_loss2[2k..2k+1].
 * d_terms_steps: NULL, or 2*batch floats per step — the one-launch form then leaves each step's per-triplet
 * loss terms there and one launch after the loop (nrhip_loss_reduce_steps) produces every pair: the same sums
 * bit for bit, no cross-workgroup hand-off inside the steps (MF.py:101 fetches the loss per step but only adds
 * it up, :103,110). */
int nrhip_mf_steps(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                   int64_t n_total, int batch, const uint64_t* d_plans, int first_step_index,
                   const float* h_alpha, float beta1, float beta2, float eps, float* d_loss2,
                   float* d_terms_steps, void* stream);
/* (sum of the BPR terms, reg * sum of the l2 terms) of n_steps steps from their per-triplet terms: step k's lie at
 * d_terms + k*2*batch ([n] terms, then [n] l2 terms; n = batch, the last step n_last). */
int nrhip_loss_reduce_steps(const float* d_terms, int n_steps, int batch, int n_last, float reg, float* d_loss2,
                            void* stream);
/* lazy mode: bring every row of both tables to step `steps_done` (no-op otherwise) */
int nrhip_mf_flush(void* ctx, int steps_done, float beta1, float beta2, float eps, void* stream);

/* ---- NGCF propagation layer, dense half (second-model coverage) ---------------
 * Replaces the per-layer TF ops of NGCF._create_ngcf_embed
 * (model/general_recommender/NGCF.py:181-198) and their gradient; the sparse half
 * (S = A·E, NGCF.py:174-179) is nrhip_spmm_csr.  Width d = 16 (conf/NGCF.properties).
 *   fwd: T1 = S·Wg+bg, T2 = (E⊙S)·Wb+bb, Z = lrelu(T1)+lrelu(T2), E' = Z/keep·mask,
 *        out = l2_normalize(E').  d_mask (n_rows*d bytes) is read when mask_given, else
 *        drawn from (seed, step, layer) and written (the backward pass needs it).
 *        d_out points at this layer's column block of the concatenated output, row
 *        stride ldo (NGCF.py:200).
 *   bwd: from dLoss/d out (d_dout, stride ldo) and dLoss/dE' coming from the next layer
 *        (d_dego_next, may be NULL): d_dS (to be pushed through A^T), d_dego_direct
 *        (= dBi ⊙ S), and the four weight gradients. */
int nrhip_ngcf_workspace_bytes(int64_t n_rows, size_t* bytes);
int nrhip_ngcf_layer_fwd(const float* d_ego, const float* d_S, const float* d_Wg,
                         const float* d_bg, const float* d_Wb, const float* d_bb, int64_t n_rows,
                         int d, float keep, uint8_t* d_mask, int mask_given, uint64_t seed,
                         uint64_t step, int layer, float* d_ego_out, float* d_out, int64_t ldo,
                         void* stream);
int nrhip_ngcf_layer_bwd(const float* d_ego, const float* d_S, const float* d_Wg,
                         const float* d_bg, const float* d_Wb, const float* d_bb, int64_t n_rows,
                         int d, float keep, const uint8_t* d_mask, const float* d_dout, int64_t ldo,
                         const float* d_dego_next, float* d_dS, float* d_dego_direct, float* d_dT1,
                         float* d_dT2, float* d_dWg, float* d_dbg, float* d_dWb, float* d_dbb,
                         void* d_ws, size_t ws_bytes, void* stream);

/* ---- Mult-VAE (second-model coverage) ------------------------------------------
 * Replaces the TF graph of model/general_recommender/MultiVAE.py:73-135 (q_graph, the
 * reparameterised sample, p_graph, log_softmax, neg-ELBO) and its gradient, for the
 * two-layer shape conf/MultiVAE.properties gives (p_dim=[z,h]: I -> h -> 2z | z -> h -> I)
 * with h <= 32 and 2z <= 32.  The multi-hot input batch is never densified: batch row r
 * is the CSR row d_rows[r] of the train matrix (MultiVAE.py:152-165 builds it densely on
 * the host).  W_p1 is kept item-major, [n_items][h] (the transpose of the TF variable).
 *
 *   nrhip_vae_encode   l2-normalise + dropout + h1 = act(x·W_q0+b) + [mu|logvar] + sample
 *                      + g1 = act(z·W_p0+b); per-row KL.  d_drop_given (per CSR position,
 *                      {0,1}) / d_eps_given ([batch][z]) replace the internal draws when
 *                      non-NULL (tests; eps ~ N(0,0.01²) as MultiVAE.py:108).  d_h0val
 *                      (per CSR position, may be NULL) receives the dropped-out input value.
 *   logits             = nrhip_score_gemm(P=g1, Q=W_p1) (+ b_p1, nrhip_add_row_bias or
 *                      fused below)
 *   nrhip_vae_decoder_loss_grad   d_S holds g1·W_p1ᵀ (bias NOT added) on entry (contents on exit
 *                      unspecified); writes nll[b] = -Σ_i log_softmax·x, dW_p1 ([n_items][h]),
 *                      db_p1, dg1 — both products on the fp32 matrix cores, dLoss/dlogits formed in
 *                      registers from the logits (never stored).
 *   nrhip_vae_mid_backward        the 16/32-wide layers' backward and weight gradients.
 *   nrhip_vae_dwq0     scatter of h0ᵀ·da1 into a zeroed dense [n_items][h] gradient.
 * act: 0 tanh, 1 sigmoid, 2 relu, 3 identity (util/tool.py activation_function). */
int nrhip_vae_encode(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows,
                     int batch, int h, int z, const float* d_Wq0, const float* d_bq0,
                     const float* d_Wq1, const float* d_bq1, const float* d_Wp0,
                     const float* d_bp0, int act, float keep, const float* d_drop_given,
                     const float* d_eps_given, float is_training, uint64_t seed, uint64_t step,
                     float* d_h0val, float* d_H1, float* d_MU, float* d_LOGVAR, float* d_EPSSTD,
                     float* d_ZS, float* d_G1, float* d_KLb, void* stream);
int nrhip_add_row_bias(float* d_S, int64_t ld, int batch, int cols, const float* d_bias,
                       void* stream);
int nrhip_vae_workspace_bytes(int batch, int cols, size_t* bytes);
int nrhip_vae_decoder_loss_grad(float* d_S, int64_t ld, int batch, int cols, int h,
                                const float* d_bp1, const int64_t* d_indptr,
                                const int32_t* d_indices, const int32_t* d_rows,
                                const float* d_G1, const float* d_Wp1, float* d_nll,
                                float* d_dWp1, float* d_dbp1, float* d_dG1, void* d_ws,
                                size_t ws_bytes, void* stream);
/* The decoder's loss and gradients with NO [batch][cols] buffer (csrc/vae_fused.hip; MultiVAE.py:104-124):
 * logits tiles are recomputed by MFMA in two passes (statistics; gradients) and never leave the registers.
 * Same arguments as nrhip_vae_decoder_loss_grad minus the slab; d_G1 [batch][h], d_Wp1 [cols][h] item-major.
 * d_dbg_logits: NULL, or (tests) a [batch][cols] buffer that receives pass 1's logits (bias included). */
int nrhip_vae_decoder_fused_workspace_bytes(int batch, int cols, size_t* bytes);
int nrhip_vae_decoder_fused(int batch, int cols, int h, const float* d_G1, const float* d_Wp1,
                            const float* d_bp1, const int64_t* d_indptr, const int32_t* d_indices,
                            const int32_t* d_rows, float* d_nll, float* d_dWp1, float* d_dbp1,
                            float* d_dG1, void* d_ws, size_t ws_bytes, float* d_dbg_logits, void* stream);
int nrhip_vae_mid_backward(int batch, int h, int z, int act, float anneal, const float* d_dG1,
                           const float* d_G1, const float* d_H1, const float* d_MU,
                           const float* d_LOGVAR, const float* d_EPSSTD, const float* d_ZS,
                           const float* d_Wp0, const float* d_Wq1, float* d_DA3, float* d_DH2,
                           float* d_DA1, float* d_dWp0, float* d_dbp0, float* d_dWq1,
                           float* d_dbq1, float* d_dbq0, void* stream);
int nrhip_vae_dwq0(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows,
                   int batch, int h, const float* d_h0val, const float* d_DA1, float* d_dWq0,
                   void* stream);
/* y += a*x ; *d_out += sum(x*x) (fp64) ; *d_out = mean(x) */
int nrhip_axpy(float a, const float* d_x, float* d_y, int64_t n, void* stream);
int nrhip_sumsq_accumulate(const float* d_x, int64_t n, double* d_out, void* stream);
int nrhip_mean_f32(const float* d_x, int n, float* d_out, void* stream);
int nrhip_mean2_f32(const float* d_x, const float* d_y, int n, float* d_out, void* stream);   /* out[0], out[1]: two means, one launch */

/* The narrow Mult-VAE step (one sess.run((loss, optimizer)) of MultiVAE.py:73-139) as one call: nrhip_vae_encode,
 * nrhip_vae_decoder_fused, nrhip_vae_mid_backward, nrhip_vae_dwq0, [nrhip_mean2_f32 -> d_stats = (neg_ll, KL); the L2
 * terms into d_regsum] and nrhip_adam_dense_tf_multi issued back to back.  P / G / M / V / sizes: the eight variables in
 * the order W_q0, b_q0, W_q1, b_q1, W_p0, b_p0, W_p1^T (item-major), b_p1 (host arrays of device pointers).  The
 * per-batch buffers hold max_batch rows; d_drop_given / d_eps_given (tests) may be NULL. */
typedef struct nrhip_vae_step_args {
  const int64_t* d_indptr; const int32_t* d_indices;   /* train CSR: the users' rows are the network's input */
  int n_items, h, z, act;
  float* P[8]; float* G[8]; float* M[8]; float* V[8]; int64_t sizes[8];
  float *d_H1, *d_MU, *d_LOGVAR, *d_EPSSTD, *d_ZS, *d_G1, *d_KLb, *d_h0val;
  float *d_nll, *d_dG1, *d_DA3, *d_DH2, *d_DA1;
  float* d_stats; double* d_regsum;
  void* d_ws; size_t ws_bytes;                        /* nrhip_vae_decoder_fused_workspace_bytes(max_batch, n_items) */
  const float* d_drop_given; const float* d_eps_given;
  float reg, beta1, beta2, adam_eps;
  uint64_t seed;
} NrhipVaeStep;
int nrhip_vae_step(const NrhipVaeStep* args, const int32_t* d_rows, int batch, float anneal, float keep,
                   float adam_alpha, uint64_t step, int want_loss, int apply, void* stream);

/* BPR-MF step in ONE launch: the gradient of MF.py:57-72 + TF-1.12 sparse Adam (util/learner.py:9-10) by
 * exact lazy replay, bit-identical to nrhip_bpr_mf_grad + nrhip_adam_sparse_tf.  Two copies of every table
 * and a stamp per copy (d_tw int32 [rows][2], {0, -1} before step 1): readers of step t take the newer copy
 * older than t, the head of a row writes the other one.  d_inb int32 [rows] (zero before step 1) marks the
 * rows of the coming batch.  d_plan is required; batch_marked != 0 promises that the previous call (step
 * t - 1) received exactly this plan as its d_next_plan. */
int nrhip_bpr_mf_step_fused(float* d_W, float* d_M, float* d_V, int32_t* d_tw, int32_t* d_inb,
                            const float* d_alpha_tab, int t, float beta1, float beta2, float eps, int d,
                            int n_users, int n_items, const int32_t* d_users, const int32_t* d_pos,
                            const int32_t* d_neg, int batch, float reg, float* d_work, float* d_loss2,
                            const uint64_t* d_plan, int batch_marked, const uint64_t* d_next_plan,
                            int n_next_occ, int period, void* stream);
/* every row brought to step t, in copy 0 (copy 1 invalidated) */
int nrhip_bpr_mf_fused_flush(float* d_W, float* d_M, float* d_V, int32_t* d_tw, const float* d_alpha_tab,
                             int t, float beta1, float beta2, float eps, int d, int64_t n_rows, void* stream);
/* ---- the other losses and optimisers of util/learner.py (MF.py:62-76 with is_pairwise /
 * loss_function / learner other than bpr + adam) ----------------------------------------------
 * Same contract as nrhip_bpr_mf_grad (dense d_GP/d_GQ zero outside the batch rows, d_work scratch of
 * 8*batch floats, optional d_plan, d_loss2 = {data loss, reg * l2}).
 *   pairwise loss_kind  (learner.py:19-29):  0 bpr  1 hinge = sum max(y+1, 0)  2 square = sum (1-y)^2
 *   pointwise loss_kind (learner.py:31-41):  0 cross_entropy = tf.losses.sigmoid_cross_entropy (batch
 *                        MEAN)  1 square = sum (label - x)^2;  instances are (user, item, label) as
 *                        PointwiseSampler yields them (data/sampler.py:93-155); regulariser
 *                        l2_loss(p, q) (MF.py:71-72).
 * nrhip_optimizer_rows_tf applies the TF-1.12 *sparse* update of learner.py:2-16 to the rows flagged
 * in d_row_flag (set with nrhip_mark_rows; cleared again here, like the gradient rows):
 *   kind 0 gd; 1 adagrad (slot0 = accumulator, initialise to 1e-8); 2 rmsprop (slot0 = ms
 *   initialised to 1, slot1 = mom initialised to 0, hyper1 = decay 0.9, hyper2 = momentum 0,
 *   eps 1e-10); 3 momentum (slot0 = accumulator, hyper1 = momentum 0.9). */
int nrhip_pairwise_mf_grad(const float* d_P, const float* d_Q, int d, int n_users,
                           const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                           int batch, float reg, int loss_kind, float* d_GP, float* d_GQ,
                           float* d_work, float* d_loss2, const uint64_t* d_plan, void* stream);
int nrhip_pointwise_mf_grad(const float* d_P, const float* d_Q, int d, int n_users,
                            const int32_t* d_users, const int32_t* d_items, const float* d_labels,
                            int batch, float reg, int loss_kind, float* d_GP, float* d_GQ,
                            float* d_work, float* d_loss2, const uint64_t* d_plan, void* stream);
int nrhip_mark_rows(const int32_t* d_ids, int n, int offset, uint8_t* d_flag, void* stream);
/* d_dst[i] = d_src[d_index[i]]: a row flag per VIRTUAL row of the chunked hop (nrhip_spmm_csr_carry) */
int nrhip_gather_u8(const uint8_t* d_src, const int32_t* d_index, int64_t n, uint8_t* d_dst, void* stream);
/* Ordered sums of gradient rows that arrive from other ranks (row-sharded tables): sort the
 * keys (local row << 32 | global occurrence position; any n: one LDS workgroup up to 16384 keys, the
 * segmented multi-workgroup network beyond), then every row's run
 * is added in key order and stored, exactly the order of the single-process head on the global batch.
 * d_index_of_pos[position] = index of that occurrence's row in d_src. */
int nrhip_sort_u64(uint64_t* d_keys, int n, void* stream);
/* Routing of a batch's 3*batch row lookups over block-partitioned tables (neurec_amd/parallel.py:
 * BipartitePartition — rank r owns users [r*bu, (r+1)*bu) and items [r*bi, (r+1)*bi), bu + bi rows per rank):
 *   nrhip_route_batch   requests in owner order, stable in request order (users, then positives, then negatives):
 *                       d_packed[i] = (row in the owner's block, occurrence code = class*code_base + position in this
 *                       rank's batch), d_order[i] = the request (class*batch + b) routed at i, d_inv = its inverse,
 *                       d_counts[world] (optional) = requests per owner.  d_keys: scratch for 3*batch keys.
 *   nrhip_route_owner_keys   on the owner: sorted keys (row << 32 | position of the occurrence in the GLOBAL batch =
 *                       class*G + d_size_off[source rank] + position) of the n rows it was asked for (ordered by
 *                       source rank, d_recv_prefix[world+1] their offsets) and d_index_of_pos[position] = index of
 *                       that occurrence among the n — the inputs of nrhip_rows_sum_sorted, which then adds the
 *                       returning gradient rows in the order TF's unsorted_segment_sum walks the concatenated batch. */
int nrhip_route_batch(const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch, int n_users,
                      int bu, int bi, int code_base, uint64_t* d_keys, int32_t* d_packed, int32_t* d_order,
                      int32_t* d_inv, int32_t* d_counts, int world, void* stream);
int nrhip_route_owner_keys(const int32_t* d_rows, const int32_t* d_codes, int n, const int32_t* d_recv_prefix,
                           const int32_t* d_size_off, int world, int global_batch, int code_base,
                           uint64_t* d_keys_out, int32_t* d_index_of_pos, void* stream);
/* The same two tables for EVERY batch of an epoch stream at once (the ids of an epoch are known when it starts —
 * data/sampler.py:196-213 draws them up front): a step then issues no routing launch at all.
 *   nrhip_sort_u64_segments        n_segs independent segments sorted in one launch (each <= 16384 keys)
 *   nrhip_route_epoch              batch k's requests live in slot [3*batch*k, 3*batch*(k+1)) of d_keys / d_packed /
 *                                  d_order / d_inv (d_seg_off[k] = 3*batch*k, d_seg_len[k] = 3 * that batch's length);
 *                                  d_counts [n_batches][world]
 *   nrhip_route_epoch_owner_keys   the received stream laid out [batch][source rank]: d_asked_off / d_asked_len per
 *                                  batch, d_batch_of[j], per batch d_recv_prefix [world + 1], d_size_off [world],
 *                                  d_global_len; d_index_of_pos has iop_stride entries per batch */
int nrhip_sort_u64_segments(uint64_t* d_keys, const int64_t* d_seg_off, const int32_t* d_seg_len, int n_segs,
                            int max_len, void* stream);
int nrhip_route_epoch(const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int64_t n, int batch,
                      int n_users, int bu, int bi, int code_base, int world, uint64_t* d_keys, int32_t* d_packed,
                      int32_t* d_order, int32_t* d_inv, int32_t* d_counts, const int64_t* d_seg_off,
                      const int32_t* d_seg_len, void* stream);
int nrhip_route_epoch_owner_keys(const int32_t* d_rows, const int32_t* d_codes, const int32_t* d_batch_of, int64_t n,
                                 const int64_t* d_asked_off, const int32_t* d_asked_len, int n_batches, int max_asked,
                                 const int32_t* d_recv_prefix, const int32_t* d_size_off,
                                 const int32_t* d_global_len, int world, int code_base, int64_t iop_stride,
                                 uint64_t* d_keys_out, int32_t* d_index_of_pos, void* stream);
int nrhip_rows_sum_sorted(const uint64_t* d_sorted_keys, int n, const int32_t* d_index_of_pos, int d,
                          const float* d_src, int64_t ld_src, float* d_dst, void* stream);
int nrhip_rows_sum_sorted2(const uint64_t* d_sorted_keys, int n, const int32_t* d_index_of_pos, int d,
                           const float* d_src_a, int64_t ld_a, float* d_dst_a, const float* d_src_b, int64_t ld_b,
                           float* d_dst_b, void* stream);                          /* two tables, same runs, one launch */
int nrhip_optimizer_rows_tf(int kind, float* d_var, float* d_slot0, float* d_slot1, float* d_grad,
                            uint8_t* d_row_flag, int64_t n_rows, int d, float lr, float hyper1,
                            float hyper2, float eps, void* stream);
/* The DENSE updates of learner.py:2-17 — a variable with a consumer that is not a gather (NGCF's and Mult-VAE's
 * tables and weights) gets TF-1.12's Apply* kernels (training_ops.cc): kinds, slots and hyper-parameters as above
 * (rmsprop in ApplyRMSProp's form: ms += (g² - ms)(1 - rho); mom = mom·momentum + g·lr / sqrt(eps + ms)).
 * clear_grad: zero the gradient once consumed. */
int nrhip_optimizer_dense_tf(int kind, float* d_var, float* d_slot0, float* d_slot1, float* d_grad, int64_t n,
                             float lr, float hyper1, float hyper2, float eps, int clear_grad, void* stream);

/* Row lookups of a row-sharded table (BASELINE config 4; the reference's tf.nn.embedding_lookup of
 * LightGCN.py:99-104 and its IndexedSlices gradient when the rows live on another rank):
 * d_dst[w][0..d) = d_src[d_rows[w]][0..d) (d_dst row stride ld_dst), and the reverse
 * d_dst[d_rows[w]] += d_src[w] (fp32 atomics, repeats summed). */
int nrhip_rows_gather(const int32_t* d_rows, int n_listed, int d, const float* d_src, float* d_dst,
                      int64_t ld_dst, void* stream);
int nrhip_rows_gather_ld(const int32_t* d_rows, int n_listed, int d, const float* d_src, int64_t ld_src,
                         float* d_dst, int64_t ld_dst, void* stream);
/* two tables, one row list, one launch: d_dst_a[w] = d_src_a[d_rows[w]], d_dst_b[w] = d_src_b[d_rows[w]] */
int nrhip_rows_gather2(const int32_t* d_rows, int n_listed, int d, const float* d_src_a, int64_t ld_a,
                       const float* d_src_b, int64_t ld_b, float* d_dst_a, int64_t ldd_a, float* d_dst_b,
                       int64_t ldd_b, void* stream);   /* d_src rows ld_src floats apart */
int nrhip_rows_scatter_add(const int32_t* d_rows, int n_listed, int d, const float* d_src,
                           int64_t ld_src, float* d_dst, void* stream);

/* y = a*x (+ y0)  elementwise helpers used between propagation passes. */
int nrhip_scale(const float* d_x, float a, float* d_y, int64_t n, void* stream);
int nrhip_add(const float* d_x, const float* d_y, float* d_out, int64_t n, void* stream);
int nrhip_div_scalar(const float* d_x, float denom, float* d_y, int64_t n, void* stream);
/* the same on row-strided [rows][cols] views (column blocks of a wider matrix) */
int nrhip_add2d(const float* d_x, int64_t ldx, const float* d_y, int64_t ldy, float* d_out,
                int64_t ldo, int64_t rows, int cols, void* stream);
int nrhip_copy2d(const float* d_x, int64_t ldx, float* d_out, int64_t ldo, int64_t rows, int cols,
                 void* stream);

/* ---- NGCF: forward pass and whole training step issued natively ----------------------------------
 * The launch sequence of NGCF._create_ngcf_embed + the BPR head + its gradient + Adam on every trainable
 * (model/general_recommender/NGCF.py:91-110,160-202) as ONE call each: a Python loop issues the ~27
 * launches of a step in ~270 us, more than they take on the GPU.  The context records caller-owned device
 * pointers only.  n_layers <= NRHIP_NGCF_MAX_LAYERS; layer width d = 16. */
#define NRHIP_NGCF_MAX_LAYERS 4
typedef struct nrhip_ngcf_buffers {
  const void* plan; const void* plan_t;                                     /* SpMM plans of A and A^T */
  const int64_t* indptr;  const int32_t* indices;  const float* vals;
  const int64_t* indptr_t; const int32_t* indices_t; const float* vals_t;
  void* spmm_ws; size_t spmm_ws_bytes;
  float* E0; float* mE; float* vE; float* gE0;                              /* [n_nodes][d] */
  float* Out; float* dOut;                                                  /* [n_nodes][d * (n_layers + 1)] */
  float* S[NRHIP_NGCF_MAX_LAYERS]; float* ego[NRHIP_NGCF_MAX_LAYERS + 1];   /* ego[0] == E0 */
  uint8_t* mask[NRHIP_NGCF_MAX_LAYERS];
  float* W[NRHIP_NGCF_MAX_LAYERS][4]; float* gW[NRHIP_NGCF_MAX_LAYERS][4];  /* W_gc, b_gc, W_bi, b_bi */
  float* mW[NRHIP_NGCF_MAX_LAYERS][4]; float* vW[NRHIP_NGCF_MAX_LAYERS][4];
  float* dS; float* dEd; float* dT1; float* dT2; float* dEgo[2];
  float* terms; int32_t* rows; uint8_t* flag; void* ws; size_t ws_bytes;
  int n_users; int n_nodes; int d; int n_layers; int max_batch;
  float reg; float keep;
} nrhip_ngcf_buffers;
int nrhip_ngcf_ctx_create(const nrhip_ngcf_buffers* bufs, void** ctx_out);
int nrhip_ngcf_ctx_destroy(void* ctx);
/* Out = concat(E0, out_1 .. out_L); masks drawn from (seed, step_counter, layer) unless mask_given */
int nrhip_ngcf_forward(void* ctx, uint64_t seed, uint64_t step_counter, int mask_given, void* stream);
/* forward + BPR head on the rows of Out + backward + dense TF Adam on E0 and the 4 * L weights */
int nrhip_ngcf_step(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                    int batch, const uint64_t* d_plan, uint64_t seed, uint64_t step_counter,
                    int mask_given, float alpha, float beta1, float beta2, float eps, float* d_loss2,
                    void* stream);

/* ---- NGCF layers of any width (NGCF.py:31-33,271-286; csrc/ngcf_wide.hip) — the row-wise pieces a layer needs next to
 * nrhip_spmm_csr and nrhip_gemm_kmajor; neurec_amd/ngcf_wide.py strings them (widths 1..256). */
int nrhip_ew_mul(const float* d_a, int64_t lda, const float* d_b, int64_t ldb, int64_t rows, int cols, float* d_out,
                 int64_t ldo, void* stream);                                            /* bi = E .* S (NGCF.py:185) */
int nrhip_ngcf_act_fwd(const float* d_T1, const float* d_T2, int64_t ldt, int64_t n_rows, int w, int w_pad,
                       float keep, uint8_t* d_mask_io, int mask_given, uint64_t seed, uint64_t step, int layer,
                       float* d_ego_out, int64_t lde, float* d_out, int64_t ldo, void* stream);   /* NGCF.py:181-198 */
int nrhip_ngcf_act_bwd(const float* d_dout, int64_t ldo, const float* d_dego_next, int64_t ldn,
                       const float* d_ego_next, int64_t lde, const float* d_T1, const float* d_T2, int64_t ldt,
                       const uint8_t* d_mask, int64_t n_rows, int w, float keep, float* d_dT1, float* d_dT2,
                       void* stream);
int nrhip_ngcf_mix_bwd(const float* d_Y1, const float* d_Y2, int64_t ldy, const float* d_ego, const float* d_S,
                       int64_t lde, int64_t n_rows, int w, int w_pad, float* d_dS, float* d_dego_direct,
                       void* stream);

/* NGCF's other conf-surface branches (r05): alg_type = gcn / gcmc (NGCF.py:204-248 — a layer is leaky_relu(S W_gc + b_gc)
 * with message dropout, gcmc adds a dense layer on top and concatenates only those) and node dropout of the adjacency
 * (NGCF.py:162-164,334-362).  The products are nrhip_spmm_csr / nrhip_gemm_f32; these are the element-wise pieces:
 *   nrhip_lrelu_drop_fwd   y = (flags & 1 ? leaky_relu(T) : T); (flags & 2): y = mask ? y / keep : 0 (mask read when
 *                          mask_given, else drawn from (seed, step, layer) and written); d_out_a [n_rows][lda] padded with
 *                          zeros to w_pad (the next hop's operand), d_out_b [n_rows][ldb] the w real columns; either NULL
 *   nrhip_lrelu_drop_bwd   dT[n_rows][w] = (d_a + d_b) through the same ops backwards (d_b optional)
 *   nrhip_edge_dropout     out[e] = keep_e ? vals[e] * (1 / keep) : 0 over the stored entries (sparse_retain + scale);
 *                          keep_e read (given) or drawn from (seed, step) and written
 *   nrhip_gather_f32       dst[e] = src[index[e]] (the transposed matrix takes the same draw) */
int nrhip_lrelu_drop_fwd(const float* d_T, int64_t ldt, int64_t n_rows, int w, int w_pad, float keep, uint8_t* d_mask_io,
                         int mask_given, uint64_t seed, uint64_t step, int layer, int flags, float* d_out_a, int64_t lda,
                         float* d_out_b, int64_t ldb, void* stream);
int nrhip_lrelu_drop_bwd(const float* d_da, int64_t lda, const float* d_db, int64_t ldb, const float* d_T, int64_t ldt,
                         const uint8_t* d_mask, int64_t n_rows, int w, float keep, int flags, float* d_dT, void* stream);
int nrhip_edge_dropout(const float* d_vals, int64_t n, float keep, uint8_t* d_keep_io, int given, uint64_t seed,
                       uint64_t step, float* d_out, void* stream);
int nrhip_gather_f32(const float* d_src, const int32_t* d_index, int64_t n, float* d_dst, void* stream);

/* The width-generic NGCF forward pass and training step as ONE call each (neurec_amd/ngcf_wide.py strung them from
 * ~70 Python-issued launches: 1.0 ms a step at 64 / [64, 64, 64], of which the GPU was busy 0.45).  The struct
 * records caller-owned device pointers; per layer k: widths w[k] -> w[k+1] (wp = padded to the SpMM's row widths),
 * off[k] = first column of layer k's block in the concatenated output. */
#define NRHIP_NGCF_WIDE_MAX_LAYERS 8
typedef struct {
  const void* plan; const int64_t* indptr; const int32_t* indices; const float* vals;           /* A_hat */
  const void* plan_t; const int64_t* indptr_t; const int32_t* indices_t; const float* vals_t;   /* A_hat^T */
  void* ws_fwd[NRHIP_NGCF_WIDE_MAX_LAYERS]; size_t ws_fwd_bytes[NRHIP_NGCF_WIDE_MAX_LAYERS];     /* SpMM scratch per layer */
  void* ws_bwd[NRHIP_NGCF_WIDE_MAX_LAYERS]; size_t ws_bwd_bytes[NRHIP_NGCF_WIDE_MAX_LAYERS];
  int n_users, n_nodes, n_layers, max_batch, dsum, splits;
  int w[NRHIP_NGCF_WIDE_MAX_LAYERS + 1], wp[NRHIP_NGCF_WIDE_MAX_LAYERS + 1], off[NRHIP_NGCF_WIDE_MAX_LAYERS + 2];
  float *E0p, *mE, *vE, *gE0, *Out, *dOut;
  float* ego[NRHIP_NGCF_WIDE_MAX_LAYERS + 1];
  float *S[NRHIP_NGCF_WIDE_MAX_LAYERS], *X2[NRHIP_NGCF_WIDE_MAX_LAYERS], *T1[NRHIP_NGCF_WIDE_MAX_LAYERS],
      *T2[NRHIP_NGCF_WIDE_MAX_LAYERS];
  uint8_t* mask[NRHIP_NGCF_WIDE_MAX_LAYERS];
  float *W[NRHIP_NGCF_WIDE_MAX_LAYERS][4], *gW[NRHIP_NGCF_WIDE_MAX_LAYERS][4], *mW[NRHIP_NGCF_WIDE_MAX_LAYERS][4],
      *vW[NRHIP_NGCF_WIDE_MAX_LAYERS][4];
  float *dS[NRHIP_NGCF_WIDE_MAX_LAYERS], *dEd[NRHIP_NGCF_WIDE_MAX_LAYERS], *dEgo[NRHIP_NGCF_WIDE_MAX_LAYERS];
  float *dT1, *dT2, *Y1, *Y2, *terms, *cs_ws;
  size_t cs_ws_bytes;
  int32_t* rows; uint8_t* flag;
  void* gemm_ws; size_t gemm_ws_bytes;
  float reg, keep;
} nrhip_ngcf_wide_buffers;
/* forward (NGCF.py:160-202): fills Out; masks_given != 0: mask[k] hold the dropout masks, else they are drawn
 * from (seed, step) and stored there */
int nrhip_ngcf_wide_forward(const nrhip_ngcf_wide_buffers* b, int masks_given, uint64_t seed, uint64_t step,
                            void* stream);
/* forward + BPR head (NGCF.py:91-110) + backward + dense TF-Adam on the ego table and every layer weight */
int nrhip_ngcf_wide_step(const nrhip_ngcf_wide_buffers* b, const int32_t* d_users, const int32_t* d_pos,
                         const int32_t* d_neg, int batch, const uint64_t* d_plan, int masks_given, uint64_t seed,
                         uint64_t step, float alpha, float beta1, float beta2, float eps, float* d_loss2,
                         void* stream);

/* ---- Mult-VAE for any p_dim (conf/MultiVAE.properties:3: [200, 600], [200]; MultiVAE.py:46-135 builds q / p networks
 * of arbitrary depth and width) — width-generic pieces (csrc/vae_wide.hip) + a general fp32-MFMA GEMM (csrc/gemm.hip).
 * act: 0 tanh, 1 sigmoid, 2 relu, 3 identity, -1 none.  neurec_amd/vae_wide.py strings them into a step. */
/* C[m][n] (+)= sum_k A[k][m] * B[k][n]; A [K][lda], B [K][ldb] (contraction index slow), C [M][ldc]; every element the
 * k-ascending fmaf chain (continued from C when accumulate); splits > 1 cuts K into ranges whose partial products
 * (d_ws: nrhip_gemm_workspace_bytes) are added in order.  Epilogue: + d_bias_n[n] (NULL: none), then `act` (-1: none) —
 * a dense layer y = act(x W + b) is one call with A = x^T.  lda, ldb < 2^24 elements (operand rows are addressed through
 * buffer resources with 32-bit byte offsets); K = 0 leaves C (+)= 0 followed by the epilogue. */
int nrhip_gemm_workspace_bytes(int M, int N, int splits, size_t* bytes);
/* the general form: either operand k-major ([K][ld], a_kminor / b_kminor = 0) or k-minor ([M][ld] / [N][ld], the
 * contraction index contiguous, = 1; such an operand must stay below 2 GB) — x W, x W^T, x^T g without a transposed
 * copy of anything */
int nrhip_gemm_f32(const float* d_A, int64_t lda, int a_kminor, const float* d_B, int64_t ldb, int b_kminor, int M,
                   int N, int K, float* d_C, int64_t ldc, int accumulate, const float* d_bias_n, int act, int splits,
                   void* d_ws, size_t ws_bytes, void* stream);
int nrhip_gemm_kmajor(const float* d_A, int64_t lda, const float* d_B, int64_t ldb, int M, int N, int K, float* d_C,
                      int64_t ldc, int accumulate, const float* d_bias_n, int act, int splits, void* d_ws,
                      size_t ws_bytes, void* stream);
int nrhip_transpose2d(const float* d_src, int64_t ld_src, int rows, int cols, float* d_dst, int64_t ld_dst,
                      void* stream);
/* first encoder layer straight from the train CSR (tf.nn.l2_normalize + tf.nn.dropout + the first tf.matmul of
 * q_graph, MultiVAE.py:76-80): Y[b] = act(sum over the row's items of (1/sqrt(n)/keep*mask) * W[item] + bias);
 * d_h0val (per CSR position, optional) keeps the values for nrhip_vae_dwq0_wide */
int nrhip_vae_bag_fwd(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows, int batch,
                      int width, const float* d_W, const float* d_bias, int act, float keep,
                      const float* d_drop_given, uint64_t seed, uint64_t step, float* d_h0val, float* d_Y,
                      void* stream);
int nrhip_act_bwd(const float* d_dY, const float* d_Y, int64_t n, int act, float* d_dA, void* stream);
int nrhip_vae_sample(const float* d_H2, int batch, int z, const float* d_eps_given, float is_training, uint64_t seed,
                     uint64_t step, float* d_EPSSTD, float* d_ZS, float* d_KLb, void* stream);
int nrhip_vae_sample_bwd(const float* d_dZ, const float* d_H2, const float* d_EPSSTD, int batch, int z, float anneal,
                         float* d_dH2, void* stream);
int nrhip_vae_softmax_dlogits(float* d_S, int64_t ld, int batch, int cols, const int64_t* d_indptr,
                              const int32_t* d_indices, const int32_t* d_rows, float* d_nll, void* stream);
/* d_out[c] = sum_r d_X[r][c] (bias gradients), a fixed association; rows > 2048: d_ws of ceil(rows / 512) * cols floats */
int nrhip_colsum_rows(const float* d_X, int64_t ld, int rows, int cols, float* d_out, void* d_ws, size_t ws_bytes,
                      void* stream);
int nrhip_vae_dwq0_wide(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows, int batch,
                        int width, const float* d_h0val, const float* d_DA1, float* d_dWq0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUREC_HIP_H */
