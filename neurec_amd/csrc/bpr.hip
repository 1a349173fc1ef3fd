// bpr.hip — embedding lookup + BPR-triplet forward/backward (MF and LightGCN heads).
//
// Stands in for the TF ops of one training step of
//   MF          model/general_recommender/MF.py:54-72   (lookups :57-58, dots :59,
//               loss util/learner.py:19-22, l2 util/tool.py:216-217)
//   LightGCN    model/general_recommender/LightGCN.py:99-104 (six lookups),
//               :156-166 (create_bpr_loss), util/tool.py:198-200,220-224
// and for their autodiff: IndexedSlices row gradients with duplicates summed.
//
// One wave64 per triplet: the three embedding rows are gathered with one
// coalesced 4·d-byte read each (lane = column), the two inner products are
// wave shuffle reductions, the scalar loss/gradient math runs once per wave,
// and row gradients are scattered with hardware fp32 atomics into dense
// accumulators (duplicate rows inside a batch add up, as TF's
// _apply_sparse_duplicate_indices does).  Per-triplet loss terms are written
// out and reduced in a fixed order by the block that finishes last (finish_loss).
#include "nr_common.h"
#include <atomic>

namespace {

constexpr int kWavesPerBlock = 4;

template <int CPL>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int64_t row, int d,
                                         int lane, float (&out)[CPL]) {
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    out[c] = (k < d) ? base[row * d + k] : 0.f;
  }
}

template <int CPL>
__device__ __forceinline__ float dot_rows(const float (&a)[CPL], const float (&b)[CPL]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; ++c) s = fmaf(a[c], b[c], s);
  return nr_wave_sum_f32(s);
}

// Fixed-order reduction of the per-triplet terms, out2[0] = Σ mf, out2[1] = reg·Σ l2, done by the
// block that finishes last (no second launch: a 1-block reduction kernel cost 4.8 us per step,
// 17 % of an MF step).  `done` is one of a small pool of device counters, zero between launches.
__device__ unsigned g_done_pool[64];

__device__ __forceinline__ void finish_loss(const float* term_mf, const float* term_l2, int batch,
                                            float reg, float* __restrict__ out2, unsigned* done) {
  if (out2 == nullptr) return;
  __shared__ bool s_last;
  __shared__ double s_a[256], s_b[256];
  // The terms were stored with agent scope (write-through); waiting for their completion is all
  // the ordering needed before the counter moves.  A full __threadfence() here costs an L2
  // write-back per block (the row-gradient atomics leave the L2 dirty): +6 us on an MF step.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < batch; i += 256) {   // other blocks' terms: read past the L1
    a += (double)__hip_atomic_load(&term_mf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b += (double)__hip_atomic_load(&term_l2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s_a[threadIdx.x] = a;
  s_b[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) {
      s_a[threadIdx.x] += s_a[threadIdx.x + s];
      s_b[threadIdx.x] += s_b[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = (float)s_a[0];
    out2[1] = reg * (float)s_b[0];
    *done = 0;                                       // re-armed for the next launch that draws it
  }
}

// ---- BPR-MF ------------------------------------------------------------------
template <int CPL>
__device__ __forceinline__ void bpr_mf_grad_body(
    const float* __restrict__ P, const float* __restrict__ Q, int d,
    const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, float reg, float* __restrict__ GP,
    float* __restrict__ GQ, float* __restrict__ term_mf, float* __restrict__ term_l2,
    int loss_kind) {
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int b = blockIdx.x * kWavesPerBlock + wave;
  if (b >= batch) return;
  const int64_t u = __builtin_amdgcn_readfirstlane(users[b]);
  const int64_t i = __builtin_amdgcn_readfirstlane(pos[b]);
  const int64_t j = __builtin_amdgcn_readfirstlane(neg[b]);
  float p[CPL], qi[CPL], qj[CPL];
  load_row<CPL>(P, u, d, lane, p);
  load_row<CPL>(Q, i, d, lane, qi);
  load_row<CPL>(Q, j, d, lane, qj);
  const float x = dot_rows<CPL>(p, qi) - dot_rows<CPL>(p, qj);      // MF.py:59,67
  const float l2 = 0.5f * (dot_rows<CPL>(p, p) + dot_rows<CPL>(qj, qj) + dot_rows<CPL>(qi, qi));
  const float g = nr::pairwise_dloss(loss_kind, x);
  if (lane == 0) {
    // agent-scope stores: written through to where the block that finishes last reads them
    __hip_atomic_store(&term_mf[b], nr::pairwise_loss(loss_kind, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    if (k < d) {
      atomicAdd(&GP[u * d + k], g * (qi[c] - qj[c]) + reg * p[c]);
      atomicAdd(&GQ[i * d + k], g * p[c] + reg * qi[c]);
      atomicAdd(&GQ[j * d + k], -g * p[c] + reg * qj[c]);
    }
  }
}

template <int CPL>
__global__ __launch_bounds__(kWavesPerBlock* NR_WAVE) void bpr_mf_grad_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, int d,
    const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, float reg, float* __restrict__ GP,
    float* __restrict__ GQ, float* __restrict__ term_mf, float* __restrict__ term_l2,
    int loss_kind,
    float* __restrict__ out2, unsigned* done) {
  bpr_mf_grad_body<CPL>(P, Q, d, users, pos, neg, batch, reg, GP, GQ, term_mf, term_l2, loss_kind);
  finish_loss(term_mf, term_l2, batch, reg, out2, done);
}

// ---- pointwise MF (is_pairwise=False, MF.py:70-72): (user, item, label) instances ---------------
template <int CPL>
__device__ __forceinline__ void pointwise_mf_grad_body(
    const float* __restrict__ P, const float* __restrict__ Q, int d,
    const int32_t* __restrict__ users, const int32_t* __restrict__ items,
    const float* __restrict__ labels, int batch, float reg, float scale, float* __restrict__ GP,
    float* __restrict__ GQ, float* __restrict__ term_mf, float* __restrict__ term_l2,
    int loss_kind) {
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int b = blockIdx.x * kWavesPerBlock + wave;
  if (b >= batch) return;
  const int64_t u = __builtin_amdgcn_readfirstlane(users[b]);
  const int64_t i = __builtin_amdgcn_readfirstlane(items[b]);
  const float z = labels[b];
  float p[CPL], q[CPL];
  load_row<CPL>(P, u, d, lane, p);
  load_row<CPL>(Q, i, d, lane, q);
  const float x = dot_rows<CPL>(p, q);                               // MF.py:59
  const float l2 = 0.5f * (dot_rows<CPL>(p, p) + dot_rows<CPL>(q, q));
  const float g = nr::pointwise_dloss(loss_kind, z, x) * scale;
  if (lane == 0) {
    // agent-scope stores: written through to where the block that finishes last reads them
    __hip_atomic_store(&term_mf[b], nr::pointwise_loss(loss_kind, z, x) * scale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    if (k < d) {
      atomicAdd(&GP[u * d + k], g * q[c] + reg * p[c]);
      atomicAdd(&GQ[i * d + k], g * p[c] + reg * q[c]);
    }
  }
}

template <int CPL>
__global__ __launch_bounds__(kWavesPerBlock* NR_WAVE) void pointwise_mf_grad_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, int d,
    const int32_t* __restrict__ users, const int32_t* __restrict__ items,
    const float* __restrict__ labels, int batch, float reg, float scale, float* __restrict__ GP,
    float* __restrict__ GQ, float* __restrict__ term_mf, float* __restrict__ term_l2,
    int loss_kind,
    float* __restrict__ out2, unsigned* done) {
  pointwise_mf_grad_body<CPL>(P, Q, d, users, items, labels, batch, reg, scale, GP, GQ, term_mf, term_l2, loss_kind);
  finish_loss(term_mf, term_l2, batch, reg, out2, done);
}

// ---- LightGCN head --------------------------------------------------------------
template <int CPL>
__device__ __forceinline__ void lightgcn_bpr_grad_body(
    const float* __restrict__ Esum, const float* __restrict__ E0, int n_users, int d,
    float layers_p1, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, float reg, float* __restrict__ Gstar,
    float* __restrict__ Greg, float* __restrict__ term_mf, float* __restrict__ term_l2,
    float grad_div) {
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int b = blockIdx.x * kWavesPerBlock + wave;
  if (b >= batch) return;
  const int64_t u = __builtin_amdgcn_readfirstlane(users[b]);
  const int64_t i = (int64_t)n_users + __builtin_amdgcn_readfirstlane(pos[b]);
  const int64_t j = (int64_t)n_users + __builtin_amdgcn_readfirstlane(neg[b]);
  float eu[CPL], ei[CPL], ej[CPL], zu[CPL], zi[CPL], zj[CPL];
  load_row<CPL>(Esum, u, d, lane, eu);
  load_row<CPL>(Esum, i, d, lane, ei);
  load_row<CPL>(Esum, j, d, lane, ej);
  load_row<CPL>(E0, u, d, lane, zu);
  load_row<CPL>(E0, i, d, lane, zi);
  load_row<CPL>(E0, j, d, lane, zj);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {   // E* = mean over layers = sum / (L+1), LightGCN.py:146-147
    eu[c] = eu[c] / layers_p1;
    ei[c] = ei[c] / layers_p1;
    ej[c] = ej[c] / layers_p1;
  }
  const float x = dot_rows<CPL>(eu, ei) - dot_rows<CPL>(eu, ej);     // LightGCN.py:157-158,162
  const float l2 = 0.5f * (dot_rows<CPL>(zu, zu) + dot_rows<CPL>(zi, zi) + dot_rows<CPL>(zj, zj));
  const float g = nr::bpr_dloss(x);
  if (lane == 0) {
    // agent-scope stores: written through to where the block that finishes last reads them
    __hip_atomic_store(&term_mf[b], nr::bpr_loss(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    if (k < d) {
      // grad_div is 1 or a power of two: dividing each term is then exactly dividing the sum
      atomicAdd(&Gstar[u * d + k], (g * (ei[c] - ej[c])) / grad_div);
      atomicAdd(&Gstar[i * d + k], (g * eu[c]) / grad_div);
      atomicAdd(&Gstar[j * d + k], (-g * eu[c]) / grad_div);
      atomicAdd(&Greg[u * d + k], reg * zu[c]);    // regulariser on layer-0 rows, :160,164
      atomicAdd(&Greg[i * d + k], reg * zi[c]);
      atomicAdd(&Greg[j * d + k], reg * zj[c]);
    }
  }
}

template <int CPL>
__global__ __launch_bounds__(kWavesPerBlock* NR_WAVE) void lightgcn_bpr_grad_kernel(
    const float* __restrict__ Esum, const float* __restrict__ E0, int n_users, int d,
    float layers_p1, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, float reg, float* __restrict__ Gstar,
    float* __restrict__ Greg, float* __restrict__ term_mf, float* __restrict__ term_l2,
    float grad_div,
    float* __restrict__ out2, unsigned* done) {
  lightgcn_bpr_grad_body<CPL>(Esum, E0, n_users, d, layers_p1, users, pos, neg, batch, reg, Gstar, Greg, term_mf, term_l2, grad_div);
  finish_loss(term_mf, term_l2, batch, reg, out2, done);
}

__global__ __launch_bounds__(256) void mark_batch_kernel(const int32_t* __restrict__ users,
                                                         const int32_t* __restrict__ pos,
                                                         const int32_t* __restrict__ neg, int batch,
                                                         int n_users, int32_t* __restrict__ rows,
                                                         uint8_t* __restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int u = users[b], i = n_users + pos[b], j = n_users + neg[b];
  rows[b] = u; rows[batch + b] = i; rows[2 * batch + b] = j;
  flag[u] = 1; flag[i] = 1; flag[j] = 1;
}

unsigned* next_done_counter() {
  static std::atomic<unsigned*> base[64];          // the symbol has one address per device
  static std::atomic<unsigned> next{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  unsigned* b = base[dev].load(std::memory_order_acquire);
  if (!b) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_done_pool)) != hipSuccess) return nullptr;
    b = (unsigned*)p;
    base[dev].store(b, std::memory_order_release);
  }
  return b + (next.fetch_add(1) & 63u);
}

}  // namespace

extern "C" {

static int pairwise_mf_grad(const char* who, const float* d_P, const float* d_Q, int d,
                            const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                            int batch, float reg, int loss_kind, float* d_GP, float* d_GQ,
                            float* d_terms, float* d_loss2, void* stream) {
  NR_REQUIRE(d_P && d_Q && d_users && d_pos && d_neg && d_GP && d_GQ && d_terms && d_loss2,
             NR_ERR_ARG, "%s: null pointer argument", who);
  NR_REQUIRE(d >= 1 && d <= 256, NR_ERR_UNSUPPORTED, "%s: embedding dim %d outside 1..256", who, d);
  NR_REQUIRE(batch >= 0, NR_ERR_ARG, "%s: negative batch", who);
  NR_REQUIRE(loss_kind >= nr::NR_PAIR_BPR && loss_kind <= nr::NR_PAIR_SQUARE, NR_ERR_ARG,
             "%s: unknown pairwise loss %d (0 bpr, 1 hinge, 2 square)", who, loss_kind);
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0) {
    NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
    return NR_OK;
  }
  float* t_mf = d_terms;
  float* t_l2 = d_terms + batch;
  dim3 grid((batch + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * NR_WAVE);
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  if (d <= 64)
    hipLaunchKernelGGL(bpr_mf_grad_kernel<1>, grid, block, 0, st, d_P, d_Q, d, d_users, d_pos,
                       d_neg, batch, reg, d_GP, d_GQ, t_mf, t_l2, loss_kind, d_loss2, done);
  else if (d <= 128)
    hipLaunchKernelGGL(bpr_mf_grad_kernel<2>, grid, block, 0, st, d_P, d_Q, d, d_users, d_pos,
                       d_neg, batch, reg, d_GP, d_GQ, t_mf, t_l2, loss_kind, d_loss2, done);
  else
    hipLaunchKernelGGL(bpr_mf_grad_kernel<4>, grid, block, 0, st, d_P, d_Q, d, d_users, d_pos,
                       d_neg, batch, reg, d_GP, d_GQ, t_mf, t_l2, loss_kind, d_loss2, done);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_bpr_mf_grad(const float* d_P, const float* d_Q, int d, const int32_t* d_users,
                      const int32_t* d_pos, const int32_t* d_neg, int batch, float reg, float* d_GP,
                      float* d_GQ, float* d_terms, float* d_loss2, void* stream) {
  return pairwise_mf_grad("bpr_mf_grad", d_P, d_Q, d, d_users, d_pos, d_neg, batch, reg,
                          nr::NR_PAIR_BPR, d_GP, d_GQ, d_terms, d_loss2, stream);
}

int nrhip_pairwise_mf_grad(const float* d_P, const float* d_Q, int d, const int32_t* d_users,
                           const int32_t* d_pos, const int32_t* d_neg, int batch, float reg,
                           int loss_kind, float* d_GP, float* d_GQ, float* d_terms, float* d_loss2,
                           void* stream) {
  return pairwise_mf_grad("pairwise_mf_grad", d_P, d_Q, d, d_users, d_pos, d_neg, batch, reg,
                          loss_kind, d_GP, d_GQ, d_terms, d_loss2, stream);
}

int nrhip_pointwise_mf_grad(const float* d_P, const float* d_Q, int d, const int32_t* d_users,
                            const int32_t* d_items, const float* d_labels, int batch, float reg,
                            int loss_kind, float* d_GP, float* d_GQ, float* d_terms, float* d_loss2,
                            void* stream) {
  NR_REQUIRE(d_P && d_Q && d_users && d_items && d_labels && d_GP && d_GQ && d_terms && d_loss2,
             NR_ERR_ARG, "pointwise_mf_grad: null pointer argument");
  NR_REQUIRE(d >= 1 && d <= 256, NR_ERR_UNSUPPORTED,
             "pointwise_mf_grad: embedding dim %d outside 1..256", d);
  NR_REQUIRE(batch >= 0, NR_ERR_ARG, "pointwise_mf_grad: negative batch");
  NR_REQUIRE(loss_kind == nr::NR_POINT_CROSS_ENTROPY || loss_kind == nr::NR_POINT_SQUARE, NR_ERR_ARG,
             "pointwise_mf_grad: unknown pointwise loss %d (0 cross_entropy, 1 square)", loss_kind);
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0) {
    NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
    return NR_OK;
  }
  float* t_mf = d_terms;
  float* t_l2 = d_terms + batch;
  // tf.losses.sigmoid_cross_entropy averages over the batch; the squared loss is a plain sum
  const float scale = loss_kind == nr::NR_POINT_CROSS_ENTROPY ? 1.0f / (float)batch : 1.0f;
  dim3 grid((batch + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * NR_WAVE);
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  if (d <= 64)
    hipLaunchKernelGGL(pointwise_mf_grad_kernel<1>, grid, block, 0, st, d_P, d_Q, d, d_users, d_items,
                       d_labels, batch, reg, scale, d_GP, d_GQ, t_mf, t_l2, loss_kind, d_loss2, done);
  else if (d <= 128)
    hipLaunchKernelGGL(pointwise_mf_grad_kernel<2>, grid, block, 0, st, d_P, d_Q, d, d_users, d_items,
                       d_labels, batch, reg, scale, d_GP, d_GQ, t_mf, t_l2, loss_kind, d_loss2, done);
  else
    hipLaunchKernelGGL(pointwise_mf_grad_kernel<4>, grid, block, 0, st, d_P, d_Q, d, d_users, d_items,
                       d_labels, batch, reg, scale, d_GP, d_GQ, t_mf, t_l2, loss_kind, d_loss2, done);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_lightgcn_mark_batch(const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                              int batch, int n_users, int32_t* d_rows_out, uint8_t* d_row_flag,
                              void* stream) {
  NR_REQUIRE(d_users && d_pos && d_neg && d_rows_out && d_row_flag && batch >= 0 && n_users >= 0,
             NR_ERR_ARG, "lightgcn_mark_batch: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(mark_batch_kernel, dim3((batch + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, d_users, d_pos, d_neg, batch, n_users, d_rows_out,
                     d_row_flag);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

static int lightgcn_head(const float* d_Esum, const float* d_E0, int n_users, int d, int n_layers,
                         const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                         int batch, float reg, float* d_Gstar, float* d_Greg, float* d_terms,
                         float* d_loss2, float grad_div, void* stream) {
  NR_REQUIRE(d_Esum && d_E0 && d_users && d_pos && d_neg && d_Gstar && d_Greg && d_terms,
             NR_ERR_ARG, "lightgcn_bpr_grad: null pointer argument");
  NR_REQUIRE(d >= 1 && d <= 256, NR_ERR_UNSUPPORTED,
             "lightgcn_bpr_grad: embedding dim %d outside 1..256", d);
  NR_REQUIRE(batch >= 0 && n_layers >= 0 && n_users >= 0, NR_ERR_ARG,
             "lightgcn_bpr_grad: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0) {
    if (d_loss2) NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
    return NR_OK;
  }
  float* t_mf = d_terms;
  float* t_l2 = d_terms + batch;
  const float lp1 = (float)(n_layers + 1);
  dim3 grid((batch + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * NR_WAVE);
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  if (d <= 64)
    hipLaunchKernelGGL(lightgcn_bpr_grad_kernel<1>, grid, block, 0, st, d_Esum, d_E0, n_users, d,
                       lp1, d_users, d_pos, d_neg, batch, reg, d_Gstar, d_Greg, t_mf, t_l2, grad_div, d_loss2, done);
  else if (d <= 128)
    hipLaunchKernelGGL(lightgcn_bpr_grad_kernel<2>, grid, block, 0, st, d_Esum, d_E0, n_users, d,
                       lp1, d_users, d_pos, d_neg, batch, reg, d_Gstar, d_Greg, t_mf, t_l2, grad_div, d_loss2, done);
  else
    hipLaunchKernelGGL(lightgcn_bpr_grad_kernel<4>, grid, block, 0, st, d_Esum, d_E0, n_users, d,
                       lp1, d_users, d_pos, d_neg, batch, reg, d_Gstar, d_Greg, t_mf, t_l2, grad_div, d_loss2, done);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_lightgcn_bpr_grad(const float* d_Esum, const float* d_E0, int n_users, int d,
                            int n_layers, const int32_t* d_users, const int32_t* d_pos,
                            const int32_t* d_neg, int batch, float reg, float* d_Gstar,
                            float* d_Greg, float* d_terms, float* d_loss2, void* stream) {
  return lightgcn_head(d_Esum, d_E0, n_users, d, n_layers, d_users, d_pos, d_neg, batch, reg,
                       d_Gstar, d_Greg, d_terms, d_loss2, 1.0f, stream);
}

/* Same head, accumulating dLoss/dE* already divided by (n_layers+1) — the H = Gstar/(L+1) of the
 * backward pass — into d_H.  Only when L+1 is a power of two: dividing every term is then
 * bit-identical to dividing the sum, and the rows_div pass disappears. */
int nrhip_lightgcn_bpr_grad_h(const float* d_Esum, const float* d_E0, int n_users, int d,
                              int n_layers, const int32_t* d_users, const int32_t* d_pos,
                              const int32_t* d_neg, int batch, float reg, float* d_H,
                              float* d_Greg, float* d_terms, float* d_loss2, void* stream) {
  NR_REQUIRE(n_layers >= 0 && ((n_layers + 1) & n_layers) == 0, NR_ERR_ARG,
             "lightgcn_bpr_grad_h: n_layers + 1 = %d is not a power of two", n_layers + 1);
  return lightgcn_head(d_Esum, d_E0, n_users, d, n_layers, d_users, d_pos, d_neg, batch, reg, d_H,
                       d_Greg, d_terms, d_loss2, (float)(n_layers + 1), stream);
}

}  // extern "C"
