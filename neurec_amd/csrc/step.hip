// step.hip — whole training steps issued from native code.
//
// One C call enqueues every kernel of a step (the launch sequence the Python engines used to
// issue one ctypes call at a time): the reference's `sess.run(opt)` is one call into its
// native runtime too (model/general_recommender/LightGCN.py:178, MF.py:101).  At ~15
// launches per 0.3 ms step the per-call interpreter overhead, not the GPU, was the limit.
// No arithmetic lives here — only the order of the nrhip_* launches declared in
// include/neurec_hip.h.
#include "nr_common.h"
#include <algorithm>
#include "neurec_hip.h"
#include <new>

namespace {

struct LightGCNCtx {
  nrhip_lightgcn_buffers b;
};
struct NGCFCtx {
  nrhip_ngcf_buffers b;
};

struct MFCtx {
  nrhip_mf_buffers b;
  bool alpha_tail_const = false;           // the last kAlphaTail step sizes are one value (see alpha_window)
  const uint64_t* marked_plan = nullptr;   // one-launch step: the plan whose rows the last step marked ...
  int marked_step = 0;                     // ... for this step index
};

constexpr int kAlphaTail = 256;

// Step sizes beyond the table.  TF's lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) stops changing once
// both running fp32 powers vanish against 1 (t > ~17.3 k at beta2 = 0.999): from there on it is one
// value.  A context whose table ends in kAlphaTail equal entries (checked once, at creation) serves any
// later step from that tail: the kernels index the table with the true step numbers (a replay reaches
// back at most lazy_period <= 64 steps, the lane-parallel load 63), so handing them the pointer moved
// back by the excess makes every access land in the constant tail.  *out = NULL when the step cannot be
// served (a table too short for its tail to be constant).
const float* alpha_window(const MFCtx* c, int step_index) {
  const nrhip_mf_buffers& b = c->b;
  if (step_index < b.alpha_len) return b.alpha_tab;
  if (!c->alpha_tail_const || b.lazy_period > 64) return nullptr;
  return b.alpha_tab - (step_index - (b.alpha_len - 1));
}

}  // namespace

extern "C" int nrhip_spmm_csr_adam(const void* plan, const int32_t* d_indices, const float* d_vals,
                                   const float* d_X, int d, float* d_addend, float* d_grad_b,
                                   float* d_var, float* d_m, float* d_v, float alpha, float beta1,
                                   float beta2, float eps, int clear_consumed,
                                   uint8_t* d_row_flag, void* stream);
extern "C" int nrhip_spmm_plan_has_blocked(const void* plan, int d);
extern "C" int nrhip_spmm_plan_has_wanted(const void* plan, int d);
extern "C" int nrhip_spmm_csr_wanted_batch(const void* plan, const int32_t* d_indices,
                                           const float* d_vals, const float* d_X, int d,
                                           const float* d_sum_in, const float* d_layer_a,
                                           const float* d_layer_b, float* d_sum_out,
                                           const int32_t* d_users, const int32_t* d_pos,
                                           const int32_t* d_neg, int batch, int n_users,
                                           uint8_t* d_row_flag, int32_t* d_rows_out, void* stream);
extern "C" int nrhip_spmm_csr_wanted_layers(const void* plan, const int32_t* d_indices,
                                            const float* d_vals, const float* d_X, int d,
                                            const float* d_sum_in, const float* d_layer_a,
                                            const float* d_layer_b, float* d_sum_out,
                                            const uint8_t* d_y_row_wanted, void* stream);

extern "C" {

int nrhip_lightgcn_ctx_create(const nrhip_lightgcn_buffers* bufs, void** ctx_out) {
  NR_REQUIRE(bufs && ctx_out, NR_ERR_ARG, "lightgcn_ctx_create: null argument");
  const nrhip_lightgcn_buffers& b = *bufs;
  NR_REQUIRE(b.plan && b.plan_t && b.indptr && b.indices && b.vals && b.indptr_t && b.indices_t &&
                 b.vals_t && b.E0 && b.m && b.v && b.Ea && b.Eb && b.Esum && b.Esum_rows &&
                 b.Gstar && b.Greg && b.H && b.Ga && b.Gb && b.batch_rows && b.row_flag && b.terms,
             NR_ERR_ARG, "lightgcn_ctx_create: a buffer pointer is null");
  NR_REQUIRE(b.n_users > 0 && b.n_nodes > b.n_users && b.n_layers >= 0 && b.max_batch > 0,
             NR_ERR_ARG, "lightgcn_ctx_create: bad sizes");
  NR_REQUIRE(b.d == 16 || b.d == 32 || b.d == 64 || b.d == 128 || b.d == 256, NR_ERR_UNSUPPORTED,
             "lightgcn_ctx_create: embedding dim %d not built (16, 32, 64, 128, 256)", b.d);
  LightGCNCtx* c = new (std::nothrow) LightGCNCtx();
  NR_REQUIRE(c, NR_ERR_ARG, "lightgcn_ctx_create: out of host memory");
  c->b = b;
  *ctx_out = c;
  return NR_OK;
}

int nrhip_lightgcn_ctx_destroy(void* ctx) {
  delete (LightGCNCtx*)ctx;
  return NR_OK;
}

// forward + backward of one step; *g_out = buffer holding G_0 (dL/dE0 without the reg rows)
struct AdamArgs { float alpha, beta1, beta2, eps; };

// adam != nullptr: the caller wants the optimiser applied too; when the last backward hop can
// carry it as a fused epilogue (d = 64 lane-group schedule, L >= 2) *g_out comes back NULL.
// phase 0: the whole step.  Column-sharded tables cut it at the head's inner products (the one quantity that
// needs all D columns): phase 1 = forward + this rank's partial products into d_partials; phase 2 = head with
// the summed products d_given + backward.
static int lightgcn_fwd_bwd(const nrhip_lightgcn_buffers& b, const int32_t* d_users,
                            const int32_t* d_pos, const int32_t* d_neg, int batch,
                            const uint64_t* d_plan, float* d_loss2, void* stream,
                            const float** g_out, const AdamArgs* adam = nullptr,
                            bool* rearmed = nullptr, int phase = 0, float* d_partials = nullptr,
                            const float* d_given = nullptr) {
  const int L = b.n_layers, d = b.d;
  const bool skip = d >= 64;                 // the work-skipping variants exist for d >= 64
  // 0: no wanted-rows schedule; 1: it takes row flags; 2: it takes the batch itself and publishes
  // the flags / row list on the way (no mark_batch launch)
  const int wanted_form = (skip && L > 0) ? nrhip_spmm_plan_has_wanted(b.plan, d) : 0;
  if (wanted_form != 2 && phase != 2)
    NR_TRY(nrhip_lightgcn_mark_batch(d_users, d_pos, d_neg, batch, b.n_users, b.batch_rows,
                                     b.row_flag, stream));
  // forward: L-1 full hops, the last one only on the batch rows
  const float* esum = b.E0;
  if (L > 0 && phase == 2) esum = b.Esum_rows;
  if (L > 0 && phase != 2) {
    const float* src = b.E0;
    const float* acc_in = b.E0;
    float* ping[2] = {b.Ea, b.Eb};
    // With the wanted-rows schedule the last hop completes the layer sum itself from the last two
    // layers (they are still in the ping-pong buffers), so only hops before those keep the running
    // sum: at L = 3 none does — four 18 MB streams less per step, same additions in the same order.
    const bool chain = wanted_form != 0;
    const float* layer[2] = {nullptr, nullptr};     // outputs of the last two full hops, oldest first
    for (int k = 0; k < L - 1; ++k) {
      const bool keep_sum = !chain || k < L - 3;
      NR_TRY(nrhip_spmm_csr(b.plan, b.indptr, b.indices, b.vals, src, d, ping[k & 1], nullptr,
                            keep_sum ? acc_in : nullptr, keep_sum ? b.Esum : nullptr, b.spmm_ws,
                            b.spmm_ws_bytes, stream));
      src = ping[k & 1];
      if (keep_sum) acc_in = b.Esum;
      layer[0] = layer[1];
      layer[1] = src;
    }
    if (chain) {
      if (L - 1 == 1) { layer[0] = layer[1]; layer[1] = nullptr; }
      if (wanted_form == 2)
        NR_TRY(nrhip_spmm_csr_wanted_batch(b.plan, b.indices, b.vals, src, d, acc_in, layer[0], layer[1],
                                           b.Esum_rows, d_users, d_pos, d_neg, batch, b.n_users,
                                           b.row_flag, b.batch_rows, stream));
      else
        NR_TRY(nrhip_spmm_csr_wanted_layers(b.plan, b.indices, b.vals, src, d, acc_in, layer[0], layer[1],
                                            b.Esum_rows, b.row_flag, stream));
    } else if (skip)
      NR_TRY(nrhip_spmm_csr_masked(b.plan, b.indptr, b.indices, b.vals, src, nullptr, b.row_flag,
                                   d, nullptr, nullptr, acc_in, b.Esum_rows, b.spmm_ws,
                                   b.spmm_ws_bytes, stream));
    else
      NR_TRY(nrhip_spmm_csr(b.plan, b.indptr, b.indices, b.vals, src, d, nullptr, nullptr, acc_in,
                            b.Esum_rows, b.spmm_ws, b.spmm_ws_bytes, stream));
    esum = b.Esum_rows;
  }
  if (phase == 1)
    return nrhip_lightgcn_partial_dots(esum, b.E0, b.n_users, d, L, d_users, d_pos, d_neg, batch, d_partials,
                                       stream);
  // backward: H = Gstar/(L+1) on the batch rows; G_k = H + A^T G_{k+1}
  if (((L + 1) & L) == 0) {
    // L+1 a power of two (the configured L = 3): the head accumulates H directly — exact
    if (d_given)
      NR_TRY(nrhip_lightgcn_bpr_grad_given(esum, b.E0, b.n_users, d, L, d_users, d_pos, d_neg, batch, b.reg, b.H,
                                           b.Greg, b.terms, d_loss2, d_plan, d_given, 1, stream));
    else
      NR_TRY(nrhip_lightgcn_bpr_grad_h(esum, b.E0, b.n_users, d, L, d_users, d_pos, d_neg, batch,
                                       b.reg, b.H, b.Greg, b.terms, d_loss2, d_plan, stream));
  } else {
    if (d_given)
      NR_TRY(nrhip_lightgcn_bpr_grad_given(esum, b.E0, b.n_users, d, L, d_users, d_pos, d_neg, batch, b.reg,
                                           b.Gstar, b.Greg, b.terms, d_loss2, d_plan, d_given, 0, stream));
    else
      NR_TRY(nrhip_lightgcn_bpr_grad(esum, b.E0, b.n_users, d, L, d_users, d_pos, d_neg, batch, b.reg,
                                     b.Gstar, b.Greg, b.terms, d_loss2, d_plan, stream));
    NR_TRY(nrhip_rows_div(b.batch_rows, 3 * batch, d, b.Gstar, (float)(L + 1), b.H, stream));
  }
  const float* g = b.H;
  float* gping[2] = {b.Ga, b.Gb};
  // (the ApplyAdam epilogue exists in the d = 64 lane-group kernel only)
  const bool fuse = adam && rearmed && L >= 2 && d == 64 && nrhip_spmm_plan_has_blocked(b.plan_t, d);
  for (int k = 0; k < L; ++k) {
    if (k == L - 1 && fuse) {
      // last hop: G_0 = H + A^T G_1 is consumed row by row as the Adam gradient (+ reg rows)
      // ... and re-arms H, Greg and the row flags as it reads them (Gstar is untouched when the
      // head wrote H directly), so no rows_clear pass follows
      const bool rearm = ((L + 1) & L) == 0;
      NR_TRY(nrhip_spmm_csr_adam(b.plan_t, b.indices_t, b.vals_t, g, d, b.H, b.Greg, b.E0, b.m, b.v,
                                 adam->alpha, adam->beta1, adam->beta2, adam->eps, rearm ? 1 : 0,
                                 b.row_flag, stream));
      *g_out = nullptr;
      if (rearm) *rearmed = true;
      return NR_OK;
    }
    if (k == 0 && skip)
      NR_TRY(nrhip_spmm_csr_masked(b.plan_t, b.indptr_t, b.indices_t, b.vals_t, g, b.row_flag,
                                   nullptr, d, gping[k & 1], b.H, nullptr, nullptr, b.spmm_ws,
                                   b.spmm_ws_bytes, stream));
    else
      NR_TRY(nrhip_spmm_csr(b.plan_t, b.indptr_t, b.indices_t, b.vals_t, g, d, gping[k & 1], b.H,
                            nullptr, nullptr, b.spmm_ws, b.spmm_ws_bytes, stream));
    g = gping[k & 1];
  }
  *g_out = g;
  return NR_OK;
}

static int lightgcn_check(void* ctx, const int32_t* u, const int32_t* p, const int32_t* n,
                          int batch) {
  NR_REQUIRE(ctx && u && p && n, NR_ERR_ARG, "lightgcn_step: null argument");
  const nrhip_lightgcn_buffers& b = ((LightGCNCtx*)ctx)->b;
  NR_REQUIRE(batch >= 0 && batch <= b.max_batch, NR_ERR_ARG,
             "lightgcn_step: batch %d outside 0..%d", batch, b.max_batch);
  return NR_OK;
}

// One training step (LightGCN.py:178); neurec_amd/trainer.py:LightGCNEngine documents the sequence.
int nrhip_lightgcn_step(void* ctx, const int32_t* d_users, const int32_t* d_pos,
                        const int32_t* d_neg, int batch, const uint64_t* d_plan, float alpha,
                        float beta1, float beta2, float eps, float* d_loss2, void* stream) {
  NR_TRY(lightgcn_check(ctx, d_users, d_pos, d_neg, batch));
  if (batch == 0) return NR_OK;
  const nrhip_lightgcn_buffers& b = ((LightGCNCtx*)ctx)->b;
  const float* g = nullptr;
  const AdamArgs adam{alpha, beta1, beta2, eps};
  bool rearmed = false;
  NR_TRY(lightgcn_fwd_bwd(b, d_users, d_pos, d_neg, batch, d_plan, d_loss2, stream, &g, &adam,
                          &rearmed));
  if (g)   // not folded into the last hop
    NR_TRY(nrhip_adam_dense_tf2(b.E0, b.m, b.v, g, b.Greg, (int64_t)b.n_nodes * b.d, alpha, beta1,
                                beta2, eps, stream));
  if (rearmed) return NR_OK;
  NR_TRY(nrhip_rows_clear(b.batch_rows, 3 * batch, b.d, b.Gstar, b.Greg, b.H, nullptr, b.row_flag,
                          stream));
  return NR_OK;
}

// Column-sharded tables (every rank holds d of the D embedding columns and steps on the WHOLE global batch;
// neurec_amd/colshard.py): the step cut at the head's inner products.  _fwd: forward hops + this rank's partial
// products, d_partials[3*batch]; the caller all-gathers them and sums them in rank order (nrhip_partials_sum);
// _bwd: head with those sums, backward hops, ApplyAdam — all on this rank's columns, no other exchange.
int nrhip_lightgcn_step_colshard_fwd(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                                     int batch, float* d_partials, void* stream) {
  NR_TRY(lightgcn_check(ctx, d_users, d_pos, d_neg, batch));
  NR_REQUIRE(d_partials, NR_ERR_ARG, "lightgcn_step_colshard_fwd: null output");
  if (batch == 0) return NR_OK;
  const float* g = nullptr;
  return lightgcn_fwd_bwd(((LightGCNCtx*)ctx)->b, d_users, d_pos, d_neg, batch, nullptr, nullptr, stream, &g,
                          nullptr, nullptr, 1, d_partials, nullptr);
}

int nrhip_lightgcn_step_colshard_bwd(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                                     int batch, const uint64_t* d_plan, const float* d_given, float alpha,
                                     float beta1, float beta2, float eps, float* d_loss2, void* stream) {
  NR_TRY(lightgcn_check(ctx, d_users, d_pos, d_neg, batch));
  NR_REQUIRE(d_given, NR_ERR_ARG, "lightgcn_step_colshard_bwd: null d_given");
  if (batch == 0) return NR_OK;
  const nrhip_lightgcn_buffers& b = ((LightGCNCtx*)ctx)->b;
  const float* g = nullptr;
  const AdamArgs adam{alpha, beta1, beta2, eps};
  bool rearmed = false;
  NR_TRY(lightgcn_fwd_bwd(b, d_users, d_pos, d_neg, batch, d_plan, d_loss2, stream, &g, &adam, &rearmed, 2, nullptr,
                          d_given));
  if (g)
    NR_TRY(nrhip_adam_dense_tf2(b.E0, b.m, b.v, g, b.Greg, (int64_t)b.n_nodes * b.d, alpha, beta1, beta2, eps,
                                stream));
  if (rearmed) return NR_OK;
  NR_TRY(nrhip_rows_clear(b.batch_rows, 3 * batch, b.d, b.Gstar, b.Greg, b.H, nullptr, b.row_flag, stream));
  return NR_OK;
}

// Multi-GPU form: the same step cut at its one exchange point.  _grad leaves this rank's total
// dL/dE0 (G_0 + reg rows) in d_grad_out; the caller all-reduces it; _apply runs Adam on it.
int nrhip_lightgcn_step_grad(void* ctx, const int32_t* d_users, const int32_t* d_pos,
                             const int32_t* d_neg, int batch, const uint64_t* d_plan,
                             float* d_loss2, float* d_grad_out, void* stream) {
  NR_TRY(lightgcn_check(ctx, d_users, d_pos, d_neg, batch));
  NR_REQUIRE(d_grad_out, NR_ERR_ARG, "lightgcn_step_grad: null output");
  const nrhip_lightgcn_buffers& b = ((LightGCNCtx*)ctx)->b;
  const int64_t n = (int64_t)b.n_nodes * b.d;
  if (batch == 0) {
    NR_CHECK_HIP(hipMemsetAsync(d_grad_out, 0, n * sizeof(float), (hipStream_t)stream));
    return NR_OK;
  }
  const float* g = nullptr;
  NR_TRY(lightgcn_fwd_bwd(b, d_users, d_pos, d_neg, batch, d_plan, d_loss2, stream, &g));
  NR_TRY(nrhip_add(g, b.Greg, d_grad_out, n, stream));
  NR_TRY(nrhip_rows_clear(b.batch_rows, 3 * batch, b.d, b.Gstar, b.Greg, b.H, nullptr, b.row_flag,
                          stream));
  return NR_OK;
}

int nrhip_lightgcn_step_apply(void* ctx, float* d_grad, float alpha, float beta1, float beta2,
                              float eps, void* stream) {
  NR_REQUIRE(ctx && d_grad, NR_ERR_ARG, "lightgcn_step_apply: null argument");
  const nrhip_lightgcn_buffers& b = ((LightGCNCtx*)ctx)->b;
  return nrhip_adam_dense_tf(b.E0, b.m, b.v, d_grad, (int64_t)b.n_nodes * b.d, alpha, beta1, beta2,
                             eps, 0, stream);
}

int nrhip_mf_ctx_create(const nrhip_mf_buffers* bufs, void** ctx_out) {
  NR_REQUIRE(bufs && ctx_out, NR_ERR_ARG, "mf_ctx_create: null argument");
  const nrhip_mf_buffers& b = *bufs;
  NR_REQUIRE(b.P && b.Q && b.mP && b.vP && b.mQ && b.vQ && ((b.tw && b.inb) || (!b.tw && b.GP && b.GQ)) && b.terms, NR_ERR_ARG,
             "mf_ctx_create: a buffer pointer is null");
  NR_REQUIRE(b.n_users > 0 && b.n_items > 0 && b.d >= 1 && b.d <= 256 && b.max_batch > 0,
             NR_ERR_ARG, "mf_ctx_create: bad sizes");
  MFCtx* c = new (std::nothrow) MFCtx();
  NR_REQUIRE(c, NR_ERR_ARG, "mf_ctx_create: out of host memory");
  c->b = b;
  if (b.alpha_tab && b.alpha_len > kAlphaTail) {
    float tail[kAlphaTail];
    if (hipMemcpy(tail, b.alpha_tab + (b.alpha_len - kAlphaTail), sizeof(tail), hipMemcpyDeviceToHost) == hipSuccess) {
      c->alpha_tail_const = true;
      for (int k = 0; k < kAlphaTail; ++k) c->alpha_tail_const = c->alpha_tail_const && tail[k] == tail[kAlphaTail - 1];
    }
  }
  *ctx_out = c;
  return NR_OK;
}

int nrhip_mf_ctx_destroy(void* ctx) {
  delete (MFCtx*)ctx;
  return NR_OK;
}

}  // extern "C"

// d_terms != NULL (one-launch form only): the step leaves its per-triplet loss terms there ([batch] mf, [batch]
// l2) and reduces nothing; the caller reduces many steps at once afterwards (nrhip_loss_reduce_steps)
static int mf_step_impl(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                        int batch, const uint64_t* d_plan, const uint64_t* d_next_plan, int next_batch,
                        int step_index, float alpha, float beta1, float beta2, float eps, float* d_loss2,
                        float* d_terms, void* stream) {
  NR_REQUIRE(ctx && d_users && d_pos && d_neg && d_loss2, NR_ERR_ARG, "mf_step: null argument");
  const nrhip_mf_buffers& b = ((MFCtx*)ctx)->b;
  NR_REQUIRE(batch >= 0 && batch <= b.max_batch, NR_ERR_ARG, "mf_step: batch %d outside 0..%d",
             batch, b.max_batch);
  const int64_t nu = (int64_t)b.n_users * b.d, ni = (int64_t)b.n_items * b.d;
  const bool one_table = b.Q == b.P + nu && b.mQ == b.mP + nu && b.vQ == b.vP + nu &&
                         (b.tw || b.GQ == b.GP + nu);        // the one-launch step keeps no gradient table
  if (b.tw) {
    // gradient + exact lazy Adam in one launch (bpr.hip: mf_fused_step_kernel)
    NR_REQUIRE(one_table, NR_ERR_ARG, "mf_step: the one-launch step needs P|Q (and m, v) as one allocation");
    const float* alpha_tab = alpha_window((const MFCtx*)ctx, step_index);
    NR_REQUIRE(step_index >= 1 && alpha_tab, NR_ERR_ARG,
               "mf_step: step %d outside the step-size table (1..%d) and the table's tail is not constant",
               step_index, b.alpha_len - 1);
    const uint64_t* plan = d_plan;
    if (!plan && batch > 0) {
      uint64_t* own = (uint64_t*)(b.terms + 2 * (size_t)batch);
      NR_TRY(nrhip_bpr_plan(d_users, d_pos, d_neg, batch, batch, b.n_users, own, stream));
      plan = own;
    }
    // the rows of this batch are marked already when the previous step was handed this very plan as
    // its next plan (the sampler's epoch buffer: one address per batch)
    MFCtx* c = (MFCtx*)ctx;
    const int marked = d_plan && c->marked_plan == d_plan && c->marked_step == step_index;
    c->marked_plan = d_next_plan;
    c->marked_step = step_index + 1;
    return nrhip_bpr_mf_step_fused(b.P, b.mP, b.vP, b.tw, b.inb, alpha_tab, step_index, beta1, beta2, eps, b.d,
                                   b.n_users, b.n_items, d_users, d_pos, d_neg, batch, b.reg,
                                   d_terms ? d_terms : b.terms, d_terms ? nullptr : d_loss2,
                                   plan, marked, d_next_plan, d_next_plan ? 3 * next_batch : 0, b.lazy_period,
                                   stream);
  }
  if (b.last) {
    // exact lazy replay instead of the sweep: the head leaves the batch's plan (the caller's, or
    // the one it sorted into the work buffer) — the optimiser walks the same sorted occurrences
    NR_REQUIRE(one_table && b.stamp, NR_ERR_ARG,
               "mf_step: lazy Adam needs P|Q (and m, v, G) as one allocation and a stamp array");
    const float* alpha_tab = alpha_window((const MFCtx*)ctx, step_index);
    NR_REQUIRE(step_index >= 1 && alpha_tab, NR_ERR_ARG,
               "mf_step: step %d outside the step-size table (1..%d) and the table's tail is not constant",
               step_index, b.alpha_len - 1);
    const uint64_t* plan = d_plan;
    if (!plan && batch > 0) {                  // no plan from the sampler: sort it into the work buffer
      uint64_t* own = (uint64_t*)(b.terms + 2 * (size_t)batch);
      NR_TRY(nrhip_bpr_plan(d_users, d_pos, d_neg, batch, batch, b.n_users, own, stream));
      plan = own;
    }
    if (batch > 0)
      NR_TRY(nrhip_bpr_mf_grad_lazy(b.P, b.mP, b.vP, b.last, alpha_tab, b.stamp, step_index, beta1, beta2,
                                    eps, b.d, b.n_users, d_users, d_pos, d_neg, batch, b.reg, b.GP, b.terms,
                                    d_loss2, plan, stream));
    return nrhip_adam_sparse_tf_lazy(b.P, b.mP, b.vP, b.GP, b.last, b.stamp,
                                     (int64_t)b.n_users + b.n_items, b.d, batch ? plan : nullptr,
                                     3 * batch, d_next_plan, d_next_plan ? 3 * next_batch : 0, alpha_tab,
                                     step_index, b.lazy_period, beta1, beta2, eps, stream);
  }
  NR_TRY(nrhip_bpr_mf_grad(b.P, b.Q, b.d, b.n_users, d_users, d_pos, d_neg, batch, b.reg, b.GP, b.GQ,
                           b.terms, d_loss2, d_plan, stream));
  if (one_table) {
    // both tables (and their moments / gradients) are one allocation: one sweep, one launch
    NR_TRY(nrhip_adam_sparse_tf(b.P, b.mP, b.vP, b.GP, nu + ni, alpha, beta1, beta2, eps, stream));
    return NR_OK;
  }
  NR_TRY(nrhip_adam_sparse_tf(b.P, b.mP, b.vP, b.GP, nu, alpha, beta1, beta2, eps, stream));
  NR_TRY(nrhip_adam_sparse_tf(b.Q, b.mQ, b.vQ, b.GQ, ni, alpha, beta1, beta2, eps, stream));
  return NR_OK;
}

extern "C" {

// One BPR-MF step = sess.run((loss, optimizer)) (MF.py:101)
int nrhip_mf_step(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                  int batch, const uint64_t* d_plan, const uint64_t* d_next_plan, int next_batch,
                  int step_index, float alpha, float beta1, float beta2, float eps, float* d_loss2,
                  void* stream) {
  return mf_step_impl(ctx, d_users, d_pos, d_neg, batch, d_plan, d_next_plan, next_batch, step_index, alpha,
                      beta1, beta2, eps, d_loss2, nullptr, stream);
}

/* The batch loop of MF.train_model (MF.py:95-103) over consecutive batches of one epoch stream, in one
 * call: batch k = triplets [k*batch, min((k+1)*batch, n_total)), its plan d_plans + 3*k*batch (the whole
 * stream's plans as nrhip_bpr_plan(n_total, batch) lays them out; NULL: sorted per step), the next batch's
 * plan handed along.  h_alpha[k] = lr_t of step first_step_index + k (HOST array); d_loss2 receives two
 * floats per step.  A Python loop enqueues ~12 us per step — more than the one-launch step takes.
 * d_terms_steps (2 * batch floats per step, or NULL): with the one-launch form the steps then only leave their
 * per-triplet loss terms there and ONE launch after the loop reduces every step's pair — the same fixed-order
 * sums, bit for bit, with no hand-off to a last-arriving workgroup inside the steps (whose memory-model form
 * costs an L2 write-back per workgroup: profiles/r03_exp_loss_handoff.txt). */
int nrhip_mf_steps(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                   int64_t n_total, int batch, const uint64_t* d_plans, int first_step_index,
                   const float* h_alpha, float beta1, float beta2, float eps, float* d_loss2,
                   float* d_terms_steps, void* stream) {
  NR_REQUIRE(ctx && d_users && d_pos && d_neg && h_alpha && d_loss2, NR_ERR_ARG, "mf_steps: null argument");
  NR_REQUIRE(n_total >= 0 && batch >= 1 && first_step_index >= 1, NR_ERR_ARG, "mf_steps: bad sizes");
  const int64_t n_steps = (n_total + batch - 1) / batch;
  const nrhip_mf_buffers& bufs = ((MFCtx*)ctx)->b;
  float* terms = bufs.tw ? d_terms_steps : nullptr;           // deferred reduction: the one-launch form only
  for (int64_t k = 0; k < n_steps; ++k) {
    const int64_t b0 = k * batch;
    const int nb = (int)std::min<int64_t>(batch, n_total - b0);
    const int64_t b1 = b0 + nb;
    const int next = (int)std::min<int64_t>(batch, n_total - b1);
    NR_TRY(mf_step_impl(ctx, d_users + b0, d_pos + b0, d_neg + b0, nb, d_plans ? d_plans + 3 * b0 : nullptr,
                        (d_plans && next > 0) ? d_plans + 3 * b1 : nullptr, next, first_step_index + (int)k,
                        h_alpha[k], beta1, beta2, eps, d_loss2 + 2 * k, terms ? terms + 2 * (size_t)batch * k : nullptr,
                        stream));
  }
  if (terms && n_steps > 0)
    NR_TRY(nrhip_loss_reduce_steps(terms, (int)n_steps, batch, (int)(n_total - (n_steps - 1) * batch), bufs.reg,
                                   d_loss2, stream));
  return NR_OK;
}

int nrhip_mf_flush(void* ctx, int steps_done, float beta1, float beta2, float eps, void* stream) {
  NR_REQUIRE(ctx, NR_ERR_ARG, "mf_flush: null context");
  const nrhip_mf_buffers& b = ((MFCtx*)ctx)->b;
  if (b.tw) {
    if (steps_done <= 0) return NR_OK;
    const float* alpha_tab = alpha_window((const MFCtx*)ctx, steps_done);
    NR_REQUIRE(alpha_tab, NR_ERR_ARG, "mf_flush: step %d outside the step-size table", steps_done);
    return nrhip_bpr_mf_fused_flush(b.P, b.mP, b.vP, b.tw, alpha_tab, steps_done, beta1, beta2, eps, b.d,
                                    (int64_t)b.n_users + b.n_items, stream);
  }
  if (!b.last || steps_done <= 0) return NR_OK;
  const float* alpha_tab = alpha_window((const MFCtx*)ctx, steps_done);
  NR_REQUIRE(alpha_tab, NR_ERR_ARG, "mf_flush: step %d outside the step-size table", steps_done);
  return nrhip_adam_sparse_tf_lazy(b.P, b.mP, b.vP, b.GP, b.last, nullptr, (int64_t)b.n_users + b.n_items,
                                   b.d, nullptr, 0, nullptr, 0, alpha_tab, steps_done, 1, beta1, beta2, eps,
                                   stream);
}

// ---- NGCF ------------------------------------------------------------------------------------
int nrhip_ngcf_ctx_create(const nrhip_ngcf_buffers* bufs, void** ctx_out) {
  NR_REQUIRE(bufs && ctx_out, NR_ERR_ARG, "ngcf_ctx_create: null argument");
  const nrhip_ngcf_buffers& b = *bufs;
  NR_REQUIRE(b.n_layers >= 0 && b.n_layers <= NRHIP_NGCF_MAX_LAYERS && b.d == 16 && b.n_users > 0 &&
                 b.n_nodes > b.n_users && b.max_batch > 0 && b.keep > 0.f && b.keep <= 1.f,
             NR_ERR_ARG, "ngcf_ctx_create: bad sizes (layer width 16, at most %d layers)", NRHIP_NGCF_MAX_LAYERS);
  bool ok = b.plan && b.plan_t && b.indptr && b.indices && b.vals && b.indptr_t && b.indices_t && b.vals_t &&
            b.spmm_ws && b.E0 && b.mE && b.vE && b.gE0 && b.Out && b.dOut && b.dS && b.dEd && b.dT1 && b.dT2 &&
            b.dEgo[0] && b.dEgo[1] && b.terms && b.rows && b.flag && b.ws && b.ego[0] == b.E0;
  for (int k = 0; k < b.n_layers && ok; ++k) {
    ok = b.S[k] && b.ego[k + 1] && b.mask[k];
    for (int j = 0; j < 4 && ok; ++j) ok = b.W[k][j] && b.gW[k][j] && b.mW[k][j] && b.vW[k][j];
  }
  NR_REQUIRE(ok, NR_ERR_ARG, "ngcf_ctx_create: a buffer pointer is null (or ego[0] != E0)");
  NGCFCtx* c = new (std::nothrow) NGCFCtx();
  NR_REQUIRE(c, NR_ERR_ARG, "ngcf_ctx_create: out of host memory");
  c->b = b;
  *ctx_out = c;
  return NR_OK;
}

int nrhip_ngcf_ctx_destroy(void* ctx) {
  delete (NGCFCtx*)ctx;
  return NR_OK;
}

int nrhip_ngcf_forward(void* ctx, uint64_t seed, uint64_t step_counter, int mask_given, void* stream) {
  NR_REQUIRE(ctx, NR_ERR_ARG, "ngcf_forward: null context");
  const nrhip_ngcf_buffers& b = ((NGCFCtx*)ctx)->b;
  const int d = b.d, ldo = d * (b.n_layers + 1);
  NR_TRY(nrhip_copy2d(b.E0, d, b.Out, ldo, b.n_nodes, d, stream));
  for (int k = 0; k < b.n_layers; ++k) {
    NR_TRY(nrhip_spmm_csr(b.plan, b.indptr, b.indices, b.vals, b.ego[k], d, b.S[k], nullptr, nullptr, nullptr,
                          b.spmm_ws, b.spmm_ws_bytes, stream));                                  // NGCF.py:174-179
    NR_TRY(nrhip_ngcf_layer_fwd(b.ego[k], b.S[k], b.W[k][0], b.W[k][1], b.W[k][2], b.W[k][3], b.n_nodes, d,
                                b.keep, b.mask[k], mask_given, seed, step_counter, k, b.ego[k + 1],
                                b.Out + (size_t)(k + 1) * d, ldo, stream));                      // NGCF.py:181-200
  }
  return NR_OK;
}

int nrhip_ngcf_step(void* ctx, const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                    int batch, const uint64_t* d_plan, uint64_t seed, uint64_t step_counter,
                    int mask_given, float alpha, float beta1, float beta2, float eps, float* d_loss2,
                    void* stream) {
  NR_REQUIRE(ctx && d_users && d_pos && d_neg && d_loss2, NR_ERR_ARG, "ngcf_step: null argument");
  const nrhip_ngcf_buffers& b = ((NGCFCtx*)ctx)->b;
  NR_REQUIRE(batch >= 1 && batch <= b.max_batch, NR_ERR_ARG, "ngcf_step: batch %d outside 1..%d", batch, b.max_batch);
  const int d = b.d, L = b.n_layers, ldo = d * (L + 1), U = b.n_users;
  NR_TRY(nrhip_ngcf_forward(ctx, seed, step_counter, mask_given, stream));
  NR_TRY(nrhip_lightgcn_mark_batch(d_users, d_pos, d_neg, batch, U, b.rows, b.flag, stream));
  // the BPR head of NGCF.py:91-100 is the MF head on the rows of the concatenated output
  NR_TRY(nrhip_bpr_mf_grad(b.Out, b.Out + (size_t)U * ldo, ldo, U, d_users, d_pos, d_neg, batch, b.reg, b.dOut,
                           b.dOut + (size_t)U * ldo, b.terms, d_loss2, d_plan, stream));
  const float* dego = nullptr;
  for (int k = L - 1; k >= 0; --k) {
    NR_TRY(nrhip_ngcf_layer_bwd(b.ego[k], b.S[k], b.W[k][0], b.W[k][1], b.W[k][2], b.W[k][3], b.n_nodes, d, b.keep,
                                b.mask[k], b.dOut + (size_t)(k + 1) * d, ldo, dego, b.dS, b.dEd, b.dT1, b.dT2,
                                b.gW[k][0], b.gW[k][1], b.gW[k][2], b.gW[k][3], b.ws, b.ws_bytes, stream));
    float* nxt = b.dEgo[k % 2];
    NR_TRY(nrhip_spmm_csr(b.plan_t, b.indptr_t, b.indices_t, b.vals_t, b.dS, d, nxt, b.dEd, nullptr, nullptr,
                          b.spmm_ws, b.spmm_ws_bytes, stream));                                  // dE_k = dBi*S + A^T dS
    dego = nxt;
  }
  if (!dego) NR_TRY(nrhip_copy2d(b.dOut, ldo, b.gE0, d, b.n_nodes, d, stream));
  else NR_TRY(nrhip_add2d(b.dOut, ldo, dego, d, b.gE0, d, b.n_nodes, d, stream));
  // every trainable in one launch (up to 32 tensors per launch)
  float* vars[1 + 4 * NRHIP_NGCF_MAX_LAYERS]; float* ms[1 + 4 * NRHIP_NGCF_MAX_LAYERS];
  float* vs[1 + 4 * NRHIP_NGCF_MAX_LAYERS]; float* gs[1 + 4 * NRHIP_NGCF_MAX_LAYERS];
  int64_t sizes[1 + 4 * NRHIP_NGCF_MAX_LAYERS];
  int32_t clear[1 + 4 * NRHIP_NGCF_MAX_LAYERS];
  int n = 0;
  vars[n] = b.E0; ms[n] = b.mE; vs[n] = b.vE; gs[n] = b.gE0; sizes[n] = (int64_t)b.n_nodes * d; clear[n] = 0; ++n;
  for (int k = 0; k < L; ++k)
    for (int j = 0; j < 4; ++j) {
      vars[n] = b.W[k][j]; ms[n] = b.mW[k][j]; vs[n] = b.vW[k][j]; gs[n] = b.gW[k][j];
      sizes[n] = (j & 1) ? d : (int64_t)d * d; clear[n] = 0; ++n;
    }
  for (int lo = 0; lo < n; lo += 32) {
    const int m = std::min(32, n - lo);
    NR_TRY(nrhip_adam_dense_tf_multi(m, vars + lo, ms + lo, vs + lo, gs + lo, sizes + lo, clear + lo, alpha, beta1,
                                     beta2, eps, stream));
  }
  NR_TRY(nrhip_rows_clear(b.rows, 3 * batch, ldo, b.dOut, nullptr, nullptr, nullptr, b.flag, stream));
  return NR_OK;
}

}  // extern "C"
