// bpr.hip — embedding lookup + BPR-triplet forward/backward (MF and LightGCN heads).
//
// Stands in for the TF ops of one training step of
//   MF          model/general_recommender/MF.py:54-72   (lookups :57-58, dots :59,
//               loss util/learner.py:19-22, l2 util/tool.py:216-217)
//   LightGCN    model/general_recommender/LightGCN.py:99-104 (six lookups),
//               :156-166 (create_bpr_loss), util/tool.py:198-200,220-224
// and for their autodiff: IndexedSlices row gradients with duplicates summed.
//
// Row gradients are DETERMINISTIC: TF sums the slices of an IndexedSlices gradient that hit
// the same row with unsorted_segment_sum, i.e. in batch order (the users' slices; for the item
// table the positive lookups' slices, then the negative lookups').  The kernels reproduce that
// order with a *batch plan* (bpr_plan_kernel): the 3·B occurrences (row, position) of a batch
// sorted by row then position — built once per epoch for every batch by the sampler, or per
// call when the caller has none.  One wave64 per sorted occurrence gathers its triplet's rows
// (one coalesced 4·d-byte read each, lane = column), forms the two inner products with wave
// shuffles and leaves its occurrence's gradient row in LDS; the wave of the FIRST occurrence
// of a row then adds the following occurrences in order (from LDS while they are in its
// workgroup, recomputed beyond) and stores the row — no atomics, bit-identical run to run
// and to np.add.at in the oracle.  (r01's one-wave-per-triplet kernels that scattered with fp32
// hardware atomics — 5.0 us against 7.2 at B = 1,024, sums in arbitrary order — left the product
// in r06.)  Per-triplet loss terms
// are written out and reduced in a fixed order by the block that finishes last (finish_loss).
#include "nr_common.h"
#include <atomic>

namespace {

// element k of a d-column row, 0 beyond it: the load is UNCONDITIONAL on a clamped column and the result masked — a
// load behind a branch cannot be counted, the compiler waits for it (s_waitcnt vmcnt(0)) before it issues the next, and
// the heads here are nothing but chains of row loads (r06: mf_fused_step_kernel had 55 such waits for 94 loads)
__device__ __forceinline__ float ld_col(const float* __restrict__ row, int k, int d) {
  const float v = row[min(k, d - 1)];
  return k < d ? v : 0.f;
}

template <int CPL>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int64_t row, int d,
                                         int lane, float (&out)[CPL]) {
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    out[c] = ld_col(base + row * d, k, d);
  }
}

template <int CPL>
__device__ __forceinline__ float dot_rows(const float (&a)[CPL], const float (&b)[CPL]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; ++c) s = fmaf(a[c], b[c], s);
  return nr_wave_sum_f32(s);
}

// Lazy sparse Adam (adam.hip: adam_lazy_kernel): a table row may be several optimiser steps behind.
// The head must see it as TF would — every row moved every step — so it replays the row's missed
// zero-gradient steps in registers (same instruction sequence as the optimiser's own replay; nothing
// is written back here) before using it.  m / v / last are indexed by global row (users, then
// n_users + item): lazy mode keeps P|Q as one table.
struct LazyTables {
  const float* m; const float* v; const int32_t* last; const float* alpha_tab;
  int32_t* stamp;                   // stamp[row] = t on the rows of this batch (for the optimiser launch)
  int t;                            // the step being taken: rows are brought to step t - 1
  float b1, b2, omb1, omb2, eps;
};

template <int CPL>
__device__ __forceinline__ void load_row_lazy(const float* __restrict__ base, int64_t row, int64_t grow,
                                              int d, int lane, const LazyTables& lz, float (&out)[CPL]) {
  const int from = __builtin_amdgcn_readfirstlane(lz.last[grow]) + 1;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    out[c] = ld_col(base + row * d, k, d);
  }
  if (from >= lz.t) return;                      // current (the optimiser was told this batch was coming)
  float mm[CPL], vv[CPL];
  bool quiet = true;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    mm[c] = vv[c] = 0.f;
    if (k < d) {
      mm[c] = lz.m[grow * d + k];
      vv[c] = lz.v[grow * d + k];
    }
    quiet = quiet && mm[c] == 0.f && vv[c] == 0.f;
  }
  if (__all(quiet)) return;
  nr_lazy_replay<CPL>(out, mm, vv, from, lz.t - 1, lz.alpha_tab, lane, lz.b1, lz.b2, lz.omb1, lz.omb2, lz.eps);
}

// Fixed-order reduction of the per-triplet terms, out2[0] = Σ mf, out2[1] = reg·Σ l2, done by the
// block that finishes last (no second launch: a 1-block reduction kernel cost 4.8 us per step,
// 17 % of an MF step).  `done` is one of a small pool of device counters, zero between launches.
__device__ unsigned g_done_pool[64];

// out2[0] = sum mf, out2[1] = reg * sum l2 over `batch` per-triplet terms: the first 256 threads of the
// workgroup (callers launch 256- or 1024-thread blocks) take a fixed strided partition, then a fixed tree —
// the same sum whatever the block shape and whichever kernel runs it
__device__ __forceinline__ void reduce_terms(const float* term_mf, const float* term_l2, int batch, float reg,
                                             float* __restrict__ out2, double* s_a, double* s_b) {
  if (threadIdx.x < 256) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < batch; i += 256) {   // other blocks' terms: read past the L1
      a += (double)__hip_atomic_load(&term_mf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b += (double)__hip_atomic_load(&term_l2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
  }
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
  }
}

// The same reduction for many steps at once, AFTER the launches that wrote the terms (kernel boundary: no
// hand-off inside a step at all).  Step k's terms: d_terms + k * 2 * batch ([batch] mf then [batch] l2; the
// last step may be short: n_last).
__global__ __launch_bounds__(256) void loss_reduce_steps_kernel(const float* __restrict__ terms, int batch,
                                                                int n_last, float reg, float* __restrict__ out2) {
  __shared__ double s_a[256], s_b[256];
  const int k = blockIdx.x;
  const int nb = k == (int)gridDim.x - 1 ? n_last : batch;
  const float* t = terms + (int64_t)k * 2 * batch;
  reduce_terms(t, t + nb, nb, reg, out2 + 2 * k, s_a, s_b);
}

__device__ __forceinline__ void finish_loss(const float* term_mf, const float* term_l2, int batch,
                                            float reg, float* __restrict__ out2, unsigned* done,
                                            unsigned arrivers = 0) {   // 0: every workgroup of the grid
  if (out2 == nullptr) return;
  if (arrivers == 0) arrivers = gridDim.x;
  __shared__ bool s_last;
  __shared__ double s_a[256], s_b[256];
  // Hand-off to the last arriver, by the memory model: every thread's term stores happen-before the
  // workgroup barrier (workgroup-scope release fence + barrier), thread 0's counter increment is an
  // agent-scope RELEASE (cumulative over what the barrier ordered before it), the winner ACQUIRES at
  // agent scope before it reads the other workgroups' terms.  (r01/r02 used a relaxed increment
  // behind s_waitcnt(0) — correct on gfx950 because the terms are write-through agent-scope stores,
  // but formally a race; the release costs an L2 write-back per workgroup, measured in
  // profiles/r03_exp_loss_handoff.txt.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == arrivers - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  reduce_terms(term_mf, term_l2, batch, reg, out2, s_a, s_b);
  if (threadIdx.x == 0) *done = 0;                     // re-armed for the next launch that draws it
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


// =================================================================================
// Deterministic aggregation: the batch plan and the sorted-occurrence kernels
// =================================================================================
// key = (global row << 32) | occurrence position p, with p = class·nb + triplet (class 0 the
// users, class 1 the items / positives, class 2 the negatives; nb = triplets in the batch) and
// global row = user id, or n_users + item id.  Sorted ascending: rows grouped, occurrences of a
// row in batch order, the positive lookups of an item before its negative lookups.
constexpr int kPlanThreads = 1024;
constexpr int kPlanMaxKeys = 16384;          // 128 KB of LDS per sorting workgroup

// grid (n_batches, 2): y = 0 sorts the batch's user occurrences, y = 1 its item occurrences (the
// two key ranges do not overlap, so the concatenation is the sorted whole).
__global__ __launch_bounds__(kPlanThreads) void bpr_plan_kernel(
    const int32_t* __restrict__ users, const int32_t* __restrict__ items,
    const int32_t* __restrict__ third, int64_t n_total, int batch, int n_cls, int n_users,
    int np2_user, int np2_item, uint64_t* __restrict__ skey) {
  extern __shared__ uint64_t s_key[];
  const int64_t first = (int64_t)blockIdx.x * batch;
  const int nb = (int)((n_total - first) < (int64_t)batch ? (n_total - first) : (int64_t)batch);
  const bool item_side = blockIdx.y == 1;
  const int n = item_side ? (n_cls - 1) * nb : nb;
  const int np2 = item_side ? np2_item : np2_user;
  for (int k = threadIdx.x; k < np2; k += kPlanThreads) {
    uint64_t key = ~0ull;
    if (k < n) {
      if (!item_side) {
        key = ((uint64_t)(uint32_t)users[first + k] << 32) | (uint32_t)k;
      } else {
        const int c = k / nb, t = k - c * nb;
        const int32_t it = (c == 0 ? items : third)[first + t];
        key = ((uint64_t)(uint32_t)(n_users + it) << 32) | (uint32_t)((c + 1) * nb + t);
      }
    }
    s_key[k] = key;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (np2 >> 1); i += kPlanThreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const uint64_t a = s_key[lo], b = s_key[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) {
          s_key[lo] = b;
          s_key[hi] = a;
        }
      }
      // steps of stride <= 64 stay inside a wave's own 128-key chunks: no workgroup barrier (see sort_u64_kernel)
      const int j_next = j > 1 ? (j >> 1) : k;
      if (j > NR_WAVE || j_next > NR_WAVE) __syncthreads();
      else __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  uint64_t* out = skey + first * n_cls + (item_side ? nb : 0);
  for (int k = threadIdx.x; k < n; k += kPlanThreads) out[k] = s_key[k];
}

constexpr int kOccWaves = 16;                 // sorted occurrences per workgroup

// plan keys s - 1, s, s + 1 (~0 beyond either end) — the three loads issued TOGETHER on clamped positions: each behind
// its own `if` was a round trip of its own (a load behind a branch is waited for before the next is issued)
__device__ __forceinline__ void plan_keys3(const uint64_t* __restrict__ skey, int s, int n_occ, uint64_t& key,
                                           uint64_t& kprev, uint64_t& knext) {
  const uint64_t k0 = skey[s], km = skey[max(s - 1, 0)], kp = skey[min(s + 1, n_occ - 1)];
  uint32_t a0 = (uint32_t)k0, a1 = (uint32_t)(k0 >> 32), b0 = (uint32_t)km, b1 = (uint32_t)(km >> 32),
           c0 = (uint32_t)kp, c1 = (uint32_t)(kp >> 32);
  // (all six words are "used" here: the compiler may not sink a neighbour's load into the branch that needs it)
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1));
  auto uni = [](uint32_t lo, uint32_t hi) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
  };
  key = uni(a0, a1);
  kprev = s > 0 ? uni(b0, b1) : ~0ull;
  knext = s + 1 < n_occ ? uni(c0, c1) : ~0ull;
}

__device__ __forceinline__ uint64_t plan_key(const uint64_t* __restrict__ skey, int s) {
  const uint64_t k = skey[s];
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(k >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)k);
}

// ---- MF (pairwise: third = negative items; pointwise: third = float labels) ------------------
template <int CPL, bool PAIR, bool LAZY>
__device__ __forceinline__ void mf_occurrence(
    const float* __restrict__ P, const float* __restrict__ Q, int d, int n_users,
    const int32_t* __restrict__ users, const int32_t* __restrict__ items, const void* third,
    int batch, float reg, float scale, int loss_kind, uint32_t p, int lane, float (&out)[CPL],
    float* __restrict__ term_mf, float* __restrict__ term_l2, bool write_terms, const LazyTables& lz) {
  const int cls = (int)(p / (uint32_t)batch), b = (int)(p - (uint32_t)cls * (uint32_t)batch);
  const int64_t u = __builtin_amdgcn_readfirstlane(users[b]);
  const int64_t i = __builtin_amdgcn_readfirstlane(items[b]);
  float pu[CPL], qi[CPL];
  if constexpr (LAZY) {
    load_row_lazy<CPL>(P, u, u, d, lane, lz, pu);
    load_row_lazy<CPL>(Q, i, (int64_t)n_users + i, d, lane, lz, qi);
  } else {
    load_row<CPL>(P, u, d, lane, pu);
    load_row<CPL>(Q, i, d, lane, qi);
  }
  if (PAIR) {
    const int64_t j = __builtin_amdgcn_readfirstlane(((const int32_t*)third)[b]);
    float qj[CPL];
    if constexpr (LAZY) load_row_lazy<CPL>(Q, j, (int64_t)n_users + j, d, lane, lz, qj);
    else load_row<CPL>(Q, j, d, lane, qj);
    const float x = dot_rows<CPL>(pu, qi) - dot_rows<CPL>(pu, qj);      // MF.py:59,67
    const float g = nr::pairwise_dloss(loss_kind, x);
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      out[c] = cls == 0 ? g * (qi[c] - qj[c]) + reg * pu[c]
             : cls == 1 ? g * pu[c] + reg * qi[c]
                        : -g * pu[c] + reg * qj[c];
    if (write_terms && cls == 0) {
      const float l2 = 0.5f * (dot_rows<CPL>(pu, pu) + dot_rows<CPL>(qj, qj) + dot_rows<CPL>(qi, qi));
      if (lane == 0) {
        __hip_atomic_store(&term_mf[b], nr::pairwise_loss(loss_kind, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
    const float z = ((const float*)third)[b];
    const float x = dot_rows<CPL>(pu, qi);                               // MF.py:59
    const float g = nr::pointwise_dloss(loss_kind, z, x) * scale;
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      out[c] = cls == 0 ? g * qi[c] + reg * pu[c] : g * pu[c] + reg * qi[c];
    if (write_terms && cls == 0) {
      const float l2 = 0.5f * (dot_rows<CPL>(pu, pu) + dot_rows<CPL>(qi, qi));
      if (lane == 0) {
        __hip_atomic_store(&term_mf[b], nr::pointwise_loss(loss_kind, z, x) * scale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// Ordered sums over the runs of one row in the sorted occurrence list (one wave = one occurrence, its
// NV value rows in registers).  The first occurrence of a row (the "head") adds the others in list
// order — TF's unsorted_segment_sum order — from LDS.  A run can leave the workgroup only at its end,
// and only the LAST run of a workgroup can: when that run's head is here, ALL waves of the workgroup
// recompute the continuation (kOccWaves occurrences per round, every load of a round in flight at
// once) and the head adds them in order.  (A head that recomputed the continuation alone paid one
// dependent chain of memory round trips per occurrence: 43 us for the 24,576 occurrences of an
// 8,192-triplet global batch, whose hub item has a run of ~30.)  Returns true on the head wave.
template <int CPL, int NV, class F>
__device__ __forceinline__ bool sorted_run_sum(float (&acc)[NV][CPL], float* __restrict__ s_val,
                                               uint32_t* __restrict__ s_row, int* __restrict__ s_edge,
                                               const uint64_t* __restrict__ skey, int n_occ, int s,
                                               uint64_t key, uint64_t kprev, uint64_t knext, int wave,
                                               int lane, F&& occurrence) {
  constexpr int W = CPL * NR_WAVE;
  const bool active = s < n_occ;
  const uint32_t row = active ? (uint32_t)(key >> 32) : ~0u;
  const bool follows = active && s > 0 && (uint32_t)(kprev >> 32) == row;   // not the first of its run
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int c = 0; c < CPL; ++c) s_val[(v * kOccWaves + wave) * W + lane + c * NR_WAVE] = acc[v][c];
  if (lane == 0) {
    s_row[wave] = row;
    if (wave == 0) s_edge[0] = follows;                                        // the workgroup starts mid-run
    if (wave == kOccWaves - 1) s_edge[1] = active && s + 1 < n_occ && (uint32_t)(knext >> 32) == row;
  }
  __syncthreads();
  const bool head = active && !follows;
  if (head) {
    for (int t = 1; wave + t < kOccWaves && s_row[wave + t] == row; ++t)
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[v][c] += s_val[(v * kOccWaves + wave + t) * W + lane + c * NR_WAVE];
  }
  const uint32_t lastrow = s_row[kOccWaves - 1];
  // workgroup-uniform: the last run goes on AND began here
  const bool overflow = s_edge[1] && !(s_edge[0] && s_row[0] == lastrow);
  if (overflow) {
    const bool mine = head && row == lastrow;
    for (int base = (blockIdx.x + 1) * kOccWaves;; base += kOccWaves) {
      __syncthreads();                             // LDS of the previous round has been consumed
      const int idx = base + wave;
      uint64_t k2 = ~0ull;
      if (idx < n_occ) k2 = plan_key(skey, idx);
      const bool m = (uint32_t)(k2 >> 32) == lastrow;
      if (m) {
        float more[NV][CPL];
        occurrence((uint32_t)k2, more);
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int c = 0; c < CPL; ++c) s_val[(v * kOccWaves + wave) * W + lane + c * NR_WAVE] = more[v][c];
      }
      if (lane == 0) s_row[wave] = m ? lastrow : ~0u;
      __syncthreads();
      int cnt = 0;
      while (cnt < kOccWaves && s_row[cnt] == lastrow) ++cnt;                  // sorted: the matches are a prefix
      if (mine)
        for (int t = 0; t < cnt; ++t)
#pragma unroll
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[v][c] += s_val[(v * kOccWaves + t) * W + lane + c * NR_WAVE];
      if (cnt < kOccWaves) break;
    }
  }
  return head;
}

template <int CPL, bool PAIR, bool LAZY = false>
__global__ __launch_bounds__(kOccWaves* NR_WAVE) void mf_grad_sorted_kernel(
    const float* __restrict__ P, const float* __restrict__ Q, int d, int n_users,
    const int32_t* __restrict__ users, const int32_t* __restrict__ items, const void* third,
    int batch, const uint64_t* __restrict__ skey, int n_occ, float reg, float scale, int loss_kind,
    float* __restrict__ GP, float* __restrict__ GQ, float* __restrict__ term_mf,
    float* __restrict__ term_l2, float* __restrict__ out2, unsigned* done, LazyTables lz) {
  __shared__ float s_g[kOccWaves * CPL * NR_WAVE];
  __shared__ uint32_t s_row[kOccWaves];
  __shared__ int s_edge[2];
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int s = blockIdx.x * kOccWaves + wave;
  uint64_t key = 0, kprev = ~0ull, knext = ~0ull;
  float acc[1][CPL] = {};
  if (s < n_occ) {
    // the neighbours in the sorted order decide "first occurrence of its row" and "another one
    // follows": requested now, with everything else, not as a dependent load after the barrier
    plan_keys3(skey, s, n_occ, key, kprev, knext);
    mf_occurrence<CPL, PAIR, LAZY>(P, Q, d, n_users, users, items, third, batch, reg, scale, loss_kind,
                                   (uint32_t)key, lane, acc[0], term_mf, term_l2, true, lz);
  }
  const bool head = sorted_run_sum<CPL, 1>(
      acc, s_g, s_row, s_edge, skey, n_occ, s, key, kprev, knext, wave, lane,
      [&](uint32_t p, float (&out)[1][CPL]) {
        mf_occurrence<CPL, PAIR, LAZY>(P, Q, d, n_users, users, items, third, batch, reg, scale, loss_kind,
                                       p, lane, out[0], term_mf, term_l2, false, lz);
      });
  if (head) {
    const uint32_t row = (uint32_t)(key >> 32);
    if constexpr (LAZY) {
      if (lane == 0) lz.stamp[row] = lz.t;       // "in this step's batch": the optimiser launch reads it
    }
    float* dst = row < (uint32_t)n_users ? GP + (int64_t)row * d
                                         : GQ + (int64_t)(row - (uint32_t)n_users) * d;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int k = lane + c * NR_WAVE;
      if (k < d) dst[k] = acc[0][c];
    }
  }
  finish_loss(term_mf, term_l2, batch, reg, out2, done);
}

// ---- BPR-MF step in ONE launch: gradient + exact lazy TF sparse Adam ---------------------------
// The two-launch form (mf_grad_sorted_kernel<LAZY> then adam_lazy_kernel) needs the launch boundary
// because the optimiser overwrites rows the gradient waves of the same step still read.  Here every
// table (w, m, v) exists TWICE, with a stamp per copy (tw[row][copy] = the step that copy is current
// as of): a reader of step t takes the copy with the larger stamp < t, the ONE writer of a row in
// step t (the head of the row's run, with the summed gradient) writes the other copy and stamps it
// t.  A reader that sees the fresh stamp ignores it (== t), one that sees the old stamp picks the
// same copy anyway, and nobody writes the copy being read: no ordering is needed inside the launch,
// the launch boundary orders the steps.  The gradient rows never leave LDS / registers.
//   Rows outside the batch are maintained as in adam_lazy_kernel — the next batch's rows and the
// scheduled rows r = t (mod period) replay their missed zero-gradient steps — by extra waves of the
// same launch.  They must leave the rows of THIS batch to its heads: inb[row] = the latest batch the
// row is known to be in, written one step ahead by the wave that prepares the row for the next batch
// (or by a small pre-launch when the previous call was not given this batch's plan).  A scheduled wave
// skips inb[row] >= t (in this batch: the head does it; in the next: the preparing wave does it), and
// maintenance writes the other copy too, so that two roles meeting on one row store identical bytes.
struct FusedTables {
  float* W; float* M; float* V;        // [2][rows][d]
  int32_t* tw;                         // [rows][2]
  int32_t* inb;                        // [rows]
  const float* alpha_tab;
  int64_t rows;
  int t;
  float b1, b2, omb1, omb2, eps;
};

// the copy a reader of step t uses, and the first step it has not seen
__device__ __forceinline__ int fused_pick(int2 s, int t, int& from) {
  const int e0 = s.x < t ? s.x : -2, e1 = s.y < t ? s.y : -2;
  const int c = e1 > e0 ? 1 : 0;
  from = (c ? e1 : e0) + 1;
  return c;
}

template <int CPL>
__device__ __forceinline__ void fused_replay(float (&w)[CPL], float (&mm)[CPL], float (&vv)[CPL], int from,
                                             int upto, float a_mine, int a_lo, int lane,
                                             const FusedTables& ft) {
  if (from > upto) return;
  bool quiet = true;
#pragma unroll
  for (int c = 0; c < CPL; ++c) quiet = quiet && mm[c] == 0.f && vv[c] == 0.f;
  if (__all(quiet)) return;                       // 0·b1 = 0 and var - 0 = var: nothing moves
  if (from >= a_lo && from >= 1) {                // the usual case: the step sizes are in the lanes already
    for (int s2 = from; s2 <= upto; ++s2) {
      const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_mine), s2 - a_lo));
#pragma unroll
      for (int c = 0; c < CPL; ++c) nr::adam_sparse_tf(0.f, w[c], mm[c], vv[c], a, ft.b1, ft.b2, ft.omb1, ft.omb2, ft.eps);
    }
  } else {
    nr_lazy_replay<CPL>(w, mm, vv, from, upto, ft.alpha_tab, lane, ft.b1, ft.b2, ft.omb1, ft.omb2, ft.eps);
  }
}

// Row `row` as of step t - 1.  Both copies are requested together with the stamps (one round trip
// instead of two); WITH_MV: the moments as well (the head applies the step to this row).
template <int CPL, bool WITH_MV>
__device__ __forceinline__ int fused_load_row(int64_t row, int d, int lane, const FusedTables& ft,
                                              float a_mine, int a_lo, float (&w)[CPL], float (&mm)[CPL],
                                              float (&vv)[CPL]) {
  const int2 s = ((const int2*)ft.tw)[row];
  const int64_t other = ft.rows * d;
  float w0[CPL], w1[CPL], m0[CPL], m1[CPL], v0[CPL], v1[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    w0[c] = ld_col(ft.W + row * d, k, d);
    w1[c] = ld_col(ft.W + other + row * d, k, d);
    if constexpr (WITH_MV) {
      m0[c] = ld_col(ft.M + row * d, k, d);
      m1[c] = ld_col(ft.M + other + row * d, k, d);
      v0[c] = ld_col(ft.V + row * d, k, d);
      v1[c] = ld_col(ft.V + other + row * d, k, d);
    }
  }
  int from;
  const int cp = __builtin_amdgcn_readfirstlane(fused_pick(s, ft.t, from));
  from = __builtin_amdgcn_readfirstlane(from);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    w[c] = cp ? w1[c] : w0[c];
    if constexpr (WITH_MV) {
      mm[c] = cp ? m1[c] : m0[c];
      vv[c] = cp ? v1[c] : v0[c];
    }
  }
  if (from < ft.t) {                              // behind (the previous launch was not told this row was coming)
    if constexpr (!WITH_MV) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int k = lane + c * NR_WAVE;
        mm[c] = ld_col(ft.M + cp * other + row * d, k, d);
        vv[c] = ld_col(ft.V + cp * other + row * d, k, d);
      }
    }
    fused_replay<CPL>(w, mm, vv, from, ft.t - 1, a_mine, a_lo, lane, ft);
  }
  return cp;
}

// one occurrence's gradient row (MF.py:57-72, BPR loss); HEAD: the class row comes with its moments
template <int CPL, bool HEAD>
__device__ __forceinline__ int fused_occurrence(int d, int n_users, const int32_t* __restrict__ users,
                                                const int32_t* __restrict__ pos,
                                                const int32_t* __restrict__ neg, int batch, float reg,
                                                uint32_t p, int lane, const FusedTables& ft, float a_mine,
                                                int a_lo, float (&out)[CPL], float (&w)[CPL], float (&mm)[CPL],
                                                float (&vv)[CPL], float* __restrict__ term_mf,
                                                float* __restrict__ term_l2, bool write_terms) {
  const int cls = (int)(p / (uint32_t)batch), b = (int)(p - (uint32_t)cls * (uint32_t)batch);
  const int64_t u = __builtin_amdgcn_readfirstlane(users[b]);
  const int64_t i = (int64_t)n_users + __builtin_amdgcn_readfirstlane(pos[b]);
  const int64_t j = (int64_t)n_users + __builtin_amdgcn_readfirstlane(neg[b]);
  float pu[CPL], qi[CPL], qj[CPL];
  int cp = 0;
  {
    // The three rows (stamps + both copies of each) and, for a head, the class row's moments are requested TOGETHER,
    // then picked: fused_load_row one row after the other put a stamp-dependent branch between the rows' loads — three
    // round trips where one does (r06).  Same values: the pick, the rare replay of a row that is behind, in row order.
    const int64_t rows3[3] = {u, i, j};
    const int64_t other = ft.rows * d;
    int2 st[3];
    float w0[3][CPL], w1[3][CPL], m0[CPL], m1[CPL], v0[CPL], v1[CPL];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      st[q] = ((const int2*)ft.tw)[rows3[q]];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int k = lane + c * NR_WAVE;
        w0[q][c] = ld_col(ft.W + rows3[q] * d, k, d);
        w1[q][c] = ld_col(ft.W + other + rows3[q] * d, k, d);
      }
    }
    if (HEAD) {
      const int64_t hr = cls == 0 ? u : cls == 1 ? i : j;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int k = lane + c * NR_WAVE;
        m0[c] = ld_col(ft.M + hr * d, k, d);
        m1[c] = ld_col(ft.M + other + hr * d, k, d);
        v0[c] = ld_col(ft.V + hr * d, k, d);
        v1[c] = ld_col(ft.V + other + hr * d, k, d);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int from;
      const int cq = __builtin_amdgcn_readfirstlane(fused_pick(st[q], ft.t, from));
      from = __builtin_amdgcn_readfirstlane(from);
      const bool is_head = HEAD && cls == q;                    // wave-uniform
      float wq[CPL], mq[CPL], vq[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        wq[c] = cq ? w1[q][c] : w0[q][c];
        mq[c] = is_head ? (cq ? m1[c] : m0[c]) : 0.f;
        vq[c] = is_head ? (cq ? v1[c] : v0[c]) : 0.f;
      }
      if (from < ft.t) {                            // behind (the previous launch was not told this row was coming)
        if (!is_head) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const int k = lane + c * NR_WAVE;
            mq[c] = ld_col(ft.M + cq * other + rows3[q] * d, k, d);
            vq[c] = ld_col(ft.V + cq * other + rows3[q] * d, k, d);
          }
        }
        fused_replay<CPL>(wq, mq, vq, from, ft.t - 1, a_mine, a_lo, lane, ft);
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if (q == 0) pu[c] = wq[c];
        if (q == 1) qi[c] = wq[c];
        if (q == 2) qj[c] = wq[c];
        if (is_head) { mm[c] = mq[c]; vv[c] = vq[c]; }
      }
      if (is_head) cp = cq;
    }
  }
  const float x = dot_rows<CPL>(pu, qi) - dot_rows<CPL>(pu, qj);      // MF.py:59,67
  const float g = nr::pairwise_dloss((int)nr::NR_PAIR_BPR, x);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    out[c] = cls == 0 ? g * (qi[c] - qj[c]) + reg * pu[c]
           : cls == 1 ? g * pu[c] + reg * qi[c]
                      : -g * pu[c] + reg * qj[c];
    if (HEAD) w[c] = cls == 0 ? pu[c] : cls == 1 ? qi[c] : qj[c];
  }
  if (write_terms && cls == 0) {
    const float l2 = 0.5f * (dot_rows<CPL>(pu, pu) + dot_rows<CPL>(qj, qj) + dot_rows<CPL>(qi, qi));
    if (lane == 0) {
      __hip_atomic_store(&term_mf[b], nr::pairwise_loss((int)nr::NR_PAIR_BPR, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return cp;
}

template <int CPL>
__device__ __forceinline__ void fused_store_row(int64_t row, int copy, int d, int lane, const FusedTables& ft,
                                                const float (&w)[CPL], const float (&mm)[CPL],
                                                const float (&vv)[CPL], int stamp) {
  const int64_t base = copy * ft.rows * d + row * d;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    if (k < d) {
      ft.W[base + k] = w[c];
      ft.M[base + k] = mm[c];
      ft.V[base + k] = vv[c];
    }
  }
  if (lane == 0) ft.tw[2 * row + copy] = stamp;
}

#ifdef NR_MF_TIMELINE       // experiment builds only (scripts/exp_mf_timeline.sh): per-workgroup phase stamps
__device__ unsigned long long g_mf_dbg[1024 * 8];
#define NR_MF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_mf_dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define NR_MF_STAMP(i)
#endif

// What bounds this launch (scripts/exp_mf_timeline.py, profiles/r02_exp_mf_timeline.txt): the eight XCDs
// begin a kernel up to 4.3 us apart (0 / 0.4 / 1.6 / 1.6 / 3.7 / 4.2 / 2.6 / 3.1 us, the same order every
// launch); an occurrence workgroup then takes 5-8 us (keys 0.9, rows + gradient 2.9, run sums + Adam +
// store 1.3, loss tail 0.8-2.2), and the zero-gradient replays are real arithmetic: rows / period rows x
// period steps = one exact sqrt + divide chain per table row per step, ~5 us of the whole chip's VALU —
// the same arithmetic as the sweep, without its 145 MB.  One row per maintenance wave spreads it best:
// four rows per wave (a quarter of the waves to dispatch, replays interleaved) took 19 us instead of 12.5.
template <int CPL>
__global__ __launch_bounds__(kOccWaves* NR_WAVE) void mf_fused_step_kernel(
    FusedTables ft, int d, int n_users, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, const uint64_t* __restrict__ skey, int n_occ,
    const uint64_t* __restrict__ skey_next, int n_next, int period, int occ_blocks,
    float reg, float* __restrict__ term_mf, float* __restrict__ term_l2, float* __restrict__ out2,
    unsigned* done) {
  __shared__ float s_g[kOccWaves * CPL * NR_WAVE];
  __shared__ uint32_t s_row[kOccWaves];
  __shared__ int s_edge[2];
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int a_lo = ft.t - (NR_WAVE - 1);
  const float a_mine = ft.alpha_tab[max(a_lo + lane, 0)]; // step sizes of the last 64 steps, lane j: step t - 63 + j
  NR_MF_STAMP(0);
  if ((int)blockIdx.x >= occ_blocks) {
    // maintenance, one independent wave per row (no workgroup barrier): heads of the next batch's plan,
    // then the scheduled rows.  Stamps, batch marks and BOTH copies of the row are requested at once.
    const int64_t mw = (int64_t)(blockIdx.x - occ_blocks) * kOccWaves + wave;
    int64_t row64;
    const bool preparing = mw < n_next;
    if (preparing) {
      row64 = (int64_t)(skey_next[mw] >> 32);
      if (mw > 0 && (int64_t)(skey_next[mw - 1] >> 32) == row64) return;       // a later occurrence of the row
    } else {
      row64 = (int64_t)(ft.t % period) + (mw - n_next) * period;
      if (row64 >= ft.rows) return;
    }
    const int64_t row = __builtin_amdgcn_readfirstlane((int)row64);
    const int2 st = ((const int2*)ft.tw)[row];
    const int mark = ft.inb[row];
    const int64_t other = ft.rows * d;
    float w0[CPL], w1[CPL], m0[CPL], m1[CPL], v0[CPL], v1[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int k = lane + c * NR_WAVE;
      w0[c] = ld_col(ft.W + row * d, k, d);
      w1[c] = ld_col(ft.W + other + row * d, k, d);
      m0[c] = ld_col(ft.M + row * d, k, d);
      m1[c] = ld_col(ft.M + other + row * d, k, d);
      v0[c] = ld_col(ft.V + row * d, k, d);
      v1[c] = ld_col(ft.V + other + row * d, k, d);
    }
    if (preparing && lane == 0) ft.inb[row] = ft.t + 1;              // the next step's scheduled waves leave it alone
    if (mark >= ft.t || st.x == ft.t || st.y == ft.t) return;         // this batch's head / another role has it
    int from;
    const int cp = __builtin_amdgcn_readfirstlane(fused_pick(st, ft.t, from));
    from = __builtin_amdgcn_readfirstlane(from);
    float w[CPL], mm[CPL], vv[CPL];
    bool quiet = true;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      w[c] = cp ? w1[c] : w0[c];
      mm[c] = cp ? m1[c] : m0[c];
      vv[c] = cp ? v1[c] : v0[c];
      quiet = quiet && mm[c] == 0.f && vv[c] == 0.f;
    }
    if (__all(quiet)) {
      if (lane == 0) ft.tw[2 * row + cp] = ft.t;         // nothing to move: the copy in place is current
    } else {
      fused_replay<CPL>(w, mm, vv, from, ft.t, a_mine, a_lo, lane, ft);
      fused_store_row<CPL>(row, 1 - cp, d, lane, ft, w, mm, vv, ft.t);
    }
    NR_MF_STAMP(3);
    return;
  }
  const int s = blockIdx.x * kOccWaves + wave;
  uint64_t key = 0, kprev = ~0ull, knext = ~0ull;
  float acc[1][CPL] = {}, w[CPL] = {}, mm[CPL] = {}, vv[CPL] = {};
  int cp = 0;
  if (s < n_occ) {
    plan_keys3(skey, s, n_occ, key, kprev, knext);
    const bool first = s == 0 || (uint32_t)(kprev >> 32) != (uint32_t)(key >> 32);   // wave-uniform
    NR_MF_STAMP(1);
    if (first)
      cp = fused_occurrence<CPL, true>(d, n_users, users, pos, neg, batch, reg, (uint32_t)key, lane, ft, a_mine,
                                       a_lo, acc[0], w, mm, vv, term_mf, term_l2, true);
    else
      fused_occurrence<CPL, false>(d, n_users, users, pos, neg, batch, reg, (uint32_t)key, lane, ft, a_mine,
                                   a_lo, acc[0], w, mm, vv, term_mf, term_l2, true);
#ifdef NR_MF_TIMELINE
    if (acc[0][0] == 1.2345678e-30f) return;          // (the stamp waits for the gradient)
#endif
    NR_MF_STAMP(2);
  }
  const bool head = sorted_run_sum<CPL, 1>(
      acc, s_g, s_row, s_edge, skey, n_occ, s, key, kprev, knext, wave, lane,
      [&](uint32_t p, float (&out)[1][CPL]) {
        float w2[CPL], m2[CPL], v2[CPL];
        fused_occurrence<CPL, false>(d, n_users, users, pos, neg, batch, reg, p, lane, ft, a_mine, a_lo, out[0],
                                     w2, m2, v2, term_mf, term_l2, false);
      });
  if (head) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_mine), NR_WAVE - 1));   // alpha_tab[t]
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      nr::adam_sparse_tf(acc[0][c], w[c], mm[c], vv[c], a, ft.b1, ft.b2, ft.omb1, ft.omb2, ft.eps);
    fused_store_row<CPL>((int64_t)(uint32_t)(key >> 32), 1 - cp, d, lane, ft, w, mm, vv, ft.t);
  }
  NR_MF_STAMP(3);
  finish_loss(term_mf, term_l2, batch, reg, out2, done, (unsigned)occ_blocks);   // the occurrence workgroups only
  NR_MF_STAMP(4);
}

// rows of a batch whose plan the previous call did not see: marked by a launch of their own
__global__ __launch_bounds__(256) void mf_fused_mark_kernel(const uint64_t* __restrict__ skey, int n_occ,
                                                            int32_t* __restrict__ inb, int t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_occ) return;
  const uint32_t row = (uint32_t)(skey[i] >> 32);
  if (i == 0 || (uint32_t)(skey[i - 1] >> 32) != row) inb[row] = t;
}

// every row brought to step t in copy 0 (what a sweep-based run holds after t steps)
template <int CPL>
__global__ __launch_bounds__(256) void mf_fused_flush_kernel(FusedTables ft, int d) {
  const int lane = nr_lane();
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= ft.rows) return;
  const int a_lo = ft.t - (NR_WAVE - 1);
  const float a_mine = ft.alpha_tab[max(a_lo + lane, 0)];
  const int2 st = ((const int2*)ft.tw)[row];
  int from;
  const int cp = __builtin_amdgcn_readfirstlane(fused_pick(st, ft.t + 1, from));
  from = __builtin_amdgcn_readfirstlane(from);
  if (cp == 0 && from > ft.t) {
    if (lane == 0) ft.tw[2 * row + 1] = -1;
    return;
  }
  float w[CPL], mm[CPL], vv[CPL];
  const int64_t base = cp * ft.rows * d + row * d;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    w[c] = ld_col(ft.W + base, k, d);
    mm[c] = ld_col(ft.M + base, k, d);
    vv[c] = ld_col(ft.V + base, k, d);
  }
  fused_replay<CPL>(w, mm, vv, from, ft.t, a_mine, a_lo, lane, ft);
  fused_store_row<CPL>(row, 0, d, lane, ft, w, mm, vv, ft.t);
  if (lane == 0) ft.tw[2 * row + 1] = -1;
}

// ---- LightGCN head -----------------------------------------------------------------------------
template <int CPL>
__device__ __forceinline__ void lightgcn_occurrence(
    const float* __restrict__ Esum, const float* __restrict__ E0, int n_users, int d,
    float layers_p1, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, float reg, float grad_div, uint32_t p, int lane,
    float (&h)[CPL], float (&r)[CPL], float* __restrict__ term_mf, float* __restrict__ term_l2,
    bool write_terms, const float* __restrict__ given = nullptr) {
  const int cls = (int)(p / (uint32_t)batch), b = (int)(p - (uint32_t)cls * (uint32_t)batch);
  const int64_t u = __builtin_amdgcn_readfirstlane(users[b]);
  const int64_t i = (int64_t)n_users + __builtin_amdgcn_readfirstlane(pos[b]);
  const int64_t j = (int64_t)n_users + __builtin_amdgcn_readfirstlane(neg[b]);
  float eu[CPL], ei[CPL], ej[CPL], z[CPL];
  load_row<CPL>(Esum, u, d, lane, eu);
  load_row<CPL>(Esum, i, d, lane, ei);
  load_row<CPL>(Esum, j, d, lane, ej);
  load_row<CPL>(E0, cls == 0 ? u : cls == 1 ? i : j, d, lane, z);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {   // E* = mean over layers = sum / (L+1), LightGCN.py:146-147
    eu[c] = eu[c] / layers_p1;
    ei[c] = ei[c] / layers_p1;
    ej[c] = ej[c] / layers_p1;
  }
  // column-sharded tables (neurec_amd/colshard.py): this rank holds d of the D columns — the two inner
  // products (and the regulariser's sum of squares) are the sums over the ranks' partial ones, handed in
  const float x = given ? given[3 * b] - given[3 * b + 1]
                        : dot_rows<CPL>(eu, ei) - dot_rows<CPL>(eu, ej);     // LightGCN.py:157-158,162
  const float g = nr::bpr_dloss(x);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    // grad_div is 1 or a power of two: dividing each term is then exactly dividing the sum
    h[c] = cls == 0 ? (g * (ei[c] - ej[c])) / grad_div
         : cls == 1 ? (g * eu[c]) / grad_div
                    : (-g * eu[c]) / grad_div;
    r[c] = reg * z[c];                             // regulariser on layer-0 rows, :160,164
  }
  if (write_terms && cls == 0) {
    float l2;
    if (given) {
      l2 = given[3 * b + 2];
    } else {
      float zi[CPL], zj[CPL];
      load_row<CPL>(E0, i, d, lane, zi);
      load_row<CPL>(E0, j, d, lane, zj);
      l2 = 0.5f * (dot_rows<CPL>(z, z) + dot_rows<CPL>(zi, zi) + dot_rows<CPL>(zj, zj));
    }
    if (lane == 0) {
      __hip_atomic_store(&term_mf[b], nr::bpr_loss(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&term_l2[b], l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int CPL>
__global__ __launch_bounds__(kOccWaves* NR_WAVE) void lightgcn_grad_sorted_kernel(
    const float* __restrict__ Esum, const float* __restrict__ E0, int n_users, int d,
    float layers_p1, const int32_t* __restrict__ users, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int batch, const uint64_t* __restrict__ skey, int n_occ,
    float reg, float* __restrict__ Gstar, float* __restrict__ Greg, float* __restrict__ term_mf,
    float* __restrict__ term_l2, float grad_div, float* __restrict__ out2, unsigned* done,
    const float* __restrict__ given) {
  __shared__ float s_hr[2 * kOccWaves * CPL * NR_WAVE];
  __shared__ uint32_t s_row[kOccWaves];
  __shared__ int s_edge[2];
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int s = blockIdx.x * kOccWaves + wave;
  uint64_t key = 0, kprev = ~0ull, knext = ~0ull;
  float hr[2][CPL] = {};                          // [0]: dLoss/dE* row, [1]: regulariser row
  if (s < n_occ) {
    plan_keys3(skey, s, n_occ, key, kprev, knext);      // neighbours in the sorted order: see mf_grad_sorted_kernel
    lightgcn_occurrence<CPL>(Esum, E0, n_users, d, layers_p1, users, pos, neg, batch, reg, grad_div,
                             (uint32_t)key, lane, hr[0], hr[1], term_mf, term_l2, true, given);
  }
  const bool head = sorted_run_sum<CPL, 2>(
      hr, s_hr, s_row, s_edge, skey, n_occ, s, key, kprev, knext, wave, lane,
      [&](uint32_t p, float (&out)[2][CPL]) {
        lightgcn_occurrence<CPL>(Esum, E0, n_users, d, layers_p1, users, pos, neg, batch, reg, grad_div,
                                 p, lane, out[0], out[1], term_mf, term_l2, false, given);
      });
  if (head) {
    const uint32_t row = (uint32_t)(key >> 32);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int k = lane + c * NR_WAVE;
      if (k < d) {
        Gstar[(int64_t)row * d + k] = hr[0][c];
        Greg[(int64_t)row * d + k] = hr[1][c];
      }
    }
  }
  finish_loss(term_mf, term_l2, batch, reg, out2, done);
}


// Column-sharded tables: the partial inner products of every triplet on this rank's d columns —
// out[b] = (<e_u, e_i>, <e_u, e_j>, (|z_u|^2 + |z_i|^2 + |z_j|^2) / 2) with e = Esum / (L+1), z = E0 rows.
template <int CPL>
__global__ __launch_bounds__(256) void lightgcn_partial_dots_kernel(
    const float* __restrict__ Esum, const float* __restrict__ E0, int n_users, int d, float layers_p1,
    const int32_t* __restrict__ users, const int32_t* __restrict__ pos, const int32_t* __restrict__ neg,
    int batch, float* __restrict__ out) {
  const int lane = nr_lane();
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
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
  for (int c = 0; c < CPL; ++c) {
    eu[c] = eu[c] / layers_p1;
    ei[c] = ei[c] / layers_p1;
    ej[c] = ej[c] / layers_p1;
  }
  const float dp = dot_rows<CPL>(eu, ei), dn = dot_rows<CPL>(eu, ej);
  const float l2 = 0.5f * (dot_rows<CPL>(zu, zu) + dot_rows<CPL>(zi, zi) + dot_rows<CPL>(zj, zj));
  if (lane == 0) {
    out[3 * b] = dp;
    out[3 * b + 1] = dn;
    out[3 * b + 2] = l2;
  }
}

// given[b][c] = sum over the ranks r (ascending) of parts[r][b][c]: every rank adds the same numbers in the same
// order, so all ranks hold bit-identical x_b and loss terms
__global__ __launch_bounds__(256) void partials_sum_kernel(const float* __restrict__ parts, int world, int n,
                                                           float* __restrict__ given) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  float acc = parts[k];
  for (int r = 1; r < world; ++r) acc = acc + parts[(int64_t)r * n + k];
  given[k] = acc;
}


// ---- ordered row sums for gradient rows that arrive from other ranks (row-sharded tables) -------
// keys (row << 32 | global occurrence position) are sorted; the first key of a row's run adds the
// run's source rows in that order (src index = index_of_pos[position]) and STORES the sum — the
// same order the single-process head uses (unsorted_segment_sum over the global batch).
__global__ __launch_bounds__(kPlanThreads) void sort_u64_kernel(uint64_t* __restrict__ keys, int n, int np2) {
  extern __shared__ uint64_t s_key[];
  for (int k = threadIdx.x; k < np2; k += kPlanThreads) s_key[k] = k < n ? keys[k] : ~0ull;
  __syncthreads();
  // A wave's 64 pairs of a pass (i = 64 w .. 64 w + 63, then the same 1,024 pairs further) lie in one aligned
  // 128-key chunk whenever the stride j is <= 64: such a step reads only what the same wave wrote in the step before,
  // and a wave's LDS operations execute in order — no workgroup barrier between two steps of stride <= 64.  20 of
  // the 78 steps of a 4,096-key sort keep theirs (35 -> ~12 us for the 3,072 keys of a routed batch).
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (np2 >> 1); i += kPlanThreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const uint64_t a = s_key[lo], b = s_key[hi];
        if ((a > b) == ((lo & k) == 0)) {
          s_key[lo] = b;
          s_key[hi] = a;
        }
      }
      const int j_next = j > 1 ? (j >> 1) : k;             // the stride of the step that follows
      if (j > NR_WAVE || j_next > NR_WAVE) __syncthreads();
      else __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += kPlanThreads) keys[k] = s_key[k];
}

// The same network over MANY independent segments in one launch (the routing tables of a whole epoch: one segment per
// batch): workgroup s sorts keys[off[s] .. off[s] + len[s]) in its LDS.
__global__ __launch_bounds__(kPlanThreads) void sort_u64_segments_kernel(uint64_t* __restrict__ keys,
                                                                         const int64_t* __restrict__ seg_off,
                                                                         const int32_t* __restrict__ seg_len) {
  extern __shared__ uint64_t s_key[];
  uint64_t* k0 = keys + seg_off[blockIdx.x];
  const int n = seg_len[blockIdx.x];
  if (n <= 1) return;
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  for (int k = threadIdx.x; k < np2; k += kPlanThreads) s_key[k] = k < n ? k0[k] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (np2 >> 1); i += kPlanThreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const uint64_t a = s_key[lo], b = s_key[hi];
        if ((a > b) == ((lo & k) == 0)) {
          s_key[lo] = b;
          s_key[hi] = a;
        }
      }
      const int j_next = j > 1 ? (j >> 1) : k;
      if (j > NR_WAVE || j_next > NR_WAVE) __syncthreads();
      else __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += kPlanThreads) k0[k] = s_key[k];
}

// ---- batches beyond one workgroup's LDS: a segmented multi-workgroup sort -----------------------
// Segments are the plan's (batch, side) key ranges, or one plain array.  The network is the
// all-ascending bitonic form — a merge of size k is one "flip" (partner = mirror position inside
// the k-block) followed by "disperse" steps (partner = i + j) — so positions >= n act as +inf
// without being stored: a compare with a missing partner is skipped.  Steps whose partners lie
// inside one kPlanMaxKeys-chunk run in LDS (seg_sort_local / seg_disperse_local), the others as
// one global pass each.
struct SegLayout {
  int64_t n_total;      // plain: keys in the one segment; plan: triplets in the stream
  int batch, n_cls;     // plan layout (batch == 0: plain array)
};

__device__ __forceinline__ void seg_range(const SegLayout& L, int seg, int64_t& base, int& n) {
  if (L.batch == 0) {
    base = 0;
    n = (int)L.n_total;
    return;
  }
  const int b = seg >> 1;
  const int64_t first = (int64_t)b * L.batch;
  const int nb = (int)((L.n_total - first) < (int64_t)L.batch ? (L.n_total - first) : (int64_t)L.batch);
  base = first * L.n_cls + ((seg & 1) ? nb : 0);
  n = (seg & 1) ? (L.n_cls - 1) * nb : nb;
}

// every occurrence's key, unsorted, at its segment's position (the small-batch kernel makes them in LDS)
__global__ __launch_bounds__(256) void plan_fill_kernel(
    const int32_t* __restrict__ users, const int32_t* __restrict__ items,
    const int32_t* __restrict__ third, int64_t n_total, int batch, int n_cls, int n_users,
    uint64_t* __restrict__ skey) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;           // triplet index in the stream
  if (e >= n_total) return;
  const int64_t first = (e / batch) * batch;
  const int nb = (int)((n_total - first) < (int64_t)batch ? (n_total - first) : (int64_t)batch);
  const int t = (int)(e - first);
  uint64_t* out = skey + first * n_cls;
  out[t] = ((uint64_t)(uint32_t)users[e] << 32) | (uint32_t)t;
  out[nb + t] = ((uint64_t)(uint32_t)(n_users + items[e]) << 32) | (uint32_t)(nb + t);
  if (n_cls == 3) out[2 * nb + t] = ((uint64_t)(uint32_t)(n_users + third[e]) << 32) | (uint32_t)(2 * nb + t);
}

__device__ __forceinline__ void lds_cmpswap(uint64_t* s, int lo, int hi) {
  const uint64_t a = s[lo], b = s[hi];
  if (a > b) {
    s[lo] = b;
    s[hi] = a;
  }
}

// grid (chunks, segments): FULL=true sorts each chunk (all merges up to the chunk size); FULL=false
// runs only the disperse steps j = chunk/2 .. 1 that finish a larger merge
template <bool FULL>
__global__ __launch_bounds__(kPlanThreads) void seg_local_kernel(uint64_t* __restrict__ keys, SegLayout L) {
  extern __shared__ uint64_t s_key[];
  int64_t base;
  int n;
  seg_range(L, blockIdx.y, base, n);
  const int c0 = blockIdx.x * kPlanMaxKeys;
  if (c0 >= n) return;
  uint64_t* seg = keys + base;
  for (int k = threadIdx.x; k < kPlanMaxKeys; k += kPlanThreads) s_key[k] = (c0 + k < n) ? seg[c0 + k] : ~0ull;
  __syncthreads();
  if (FULL) {
    for (int k = 2; k <= kPlanMaxKeys; k <<= 1) {
      for (int i = threadIdx.x; i < (kPlanMaxKeys >> 1); i += kPlanThreads) {       // flip
        const int q = i / (k >> 1), r = i - q * (k >> 1);
        lds_cmpswap(s_key, q * k + r, q * k + (k - 1 - r));
      }
      __syncthreads();
      for (int j = k >> 2; j > 0; j >>= 1) {                                        // disperse
        for (int i = threadIdx.x; i < (kPlanMaxKeys >> 1); i += kPlanThreads) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          lds_cmpswap(s_key, lo, lo | j);
        }
        __syncthreads();
      }
    }
  } else {
    for (int j = kPlanMaxKeys >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (kPlanMaxKeys >> 1); i += kPlanThreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        lds_cmpswap(s_key, lo, lo | j);
      }
      __syncthreads();
    }
  }
  for (int k = threadIdx.x; k < kPlanMaxKeys; k += kPlanThreads)
    if (c0 + k < n) seg[c0 + k] = s_key[k];
}

// one global step over every segment: FLIP: partner = mirror inside the k-block; else partner = i + j
template <bool FLIP>
__global__ __launch_bounds__(256) void seg_global_kernel(uint64_t* __restrict__ keys, SegLayout L, int kj,
                                                         int half_np2) {
  int64_t base;
  int n;
  seg_range(L, blockIdx.y, base, n);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= half_np2) return;
  int lo, hi;
  if (FLIP) {
    const int q = i / (kj >> 1), r = i - q * (kj >> 1);
    lo = q * kj + r;
    hi = q * kj + (kj - 1 - r);
  } else {
    lo = ((i & ~(kj - 1)) << 1) | (i & (kj - 1));
    hi = lo | kj;
  }
  if (hi >= n) return;                          // the partner is +inf: nothing moves
  uint64_t* seg = keys + base;
  const uint64_t a = seg[lo], b = seg[hi];
  if (a > b) {
    seg[lo] = b;
    seg[hi] = a;
  }
}

// src_b / dst_b (optional): a second table summed along the same runs in the same launch (a row-sharded LightGCN step
// receives two gradient rows per occurrence: dLoss/dE* and the regulariser's row)
template <int CPL>
__global__ __launch_bounds__(256) void rows_sum_sorted_kernel(const uint64_t* __restrict__ skey, int n,
                                                              const int32_t* __restrict__ index_of_pos,
                                                              int d, const float* __restrict__ src,
                                                              int64_t ld_src, float* __restrict__ dst,
                                                              const float* __restrict__ src_b, int64_t ld_b,
                                                              float* __restrict__ dst_b) {
  const int lane = nr_lane();
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n) return;
  const uint64_t key = plan_key(skey, s);
  const uint32_t row = (uint32_t)(key >> 32);
  if (s > 0 && (uint32_t)(plan_key(skey, s - 1) >> 32) == row) return;
  float acc[CPL], acb[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = acb[c] = 0.f;
  for (int t = 0; s + t < n; ++t) {
    const uint64_t k2 = t == 0 ? key : plan_key(skey, s + t);
    if ((uint32_t)(k2 >> 32) != row) break;
    const int64_t i = __builtin_amdgcn_readfirstlane(index_of_pos[(uint32_t)k2]);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int k = lane + c * NR_WAVE;
      if (k < d) {
        acc[c] = t == 0 ? src[i * ld_src + k] : acc[c] + src[i * ld_src + k];
        if (src_b) acb[c] = t == 0 ? src_b[i * ld_b + k] : acb[c] + src_b[i * ld_b + k];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    if (k < d) {
      dst[(int64_t)row * d + k] = acc[c];
      if (src_b) dst_b[(int64_t)row * d + k] = acb[c];
    }
  }
}

// np2 >= n, a power of two, at least 2
int plan_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

// 128 KB of dynamic LDS: the attribute is per device (a process may drive several GPUs)
int lds_attr_once(const void* kernel, int slot) {
  static std::atomic<bool> set[4][64];
  int dev = 0;
  NR_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !set[slot][dev].load(std::memory_order_acquire)) {
    NR_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kPlanMaxKeys * (int)sizeof(uint64_t)));
    if (dev >= 0 && dev < 64) set[slot][dev].store(true, std::memory_order_release);
  }
  return NR_OK;
}

// ascending sort of every segment of `keys` (segments of at most n_max keys), any size
int sort_segments(uint64_t* keys, SegLayout L, int n_segs, int n_max, hipStream_t st) {
  if (n_max <= 1) return NR_OK;
  NR_TRY(lds_attr_once((const void*)seg_local_kernel<true>, 2));
  NR_TRY(lds_attr_once((const void*)seg_local_kernel<false>, 3));
  const int np2 = plan_pow2(n_max);
  const unsigned chunks = (unsigned)((n_max + kPlanMaxKeys - 1) / kPlanMaxKeys);
  const size_t lds = (size_t)kPlanMaxKeys * sizeof(uint64_t);
  hipLaunchKernelGGL(seg_local_kernel<true>, dim3(chunks, n_segs), dim3(kPlanThreads), lds, st, keys, L);
  const dim3 ggrid((unsigned)((np2 / 2 + 255) / 256), n_segs);
  for (int k = 2 * kPlanMaxKeys; k <= np2; k <<= 1) {
    hipLaunchKernelGGL(seg_global_kernel<true>, ggrid, dim3(256), 0, st, keys, L, k, np2 / 2);
    for (int j = k >> 2; j >= kPlanMaxKeys; j >>= 1)
      hipLaunchKernelGGL(seg_global_kernel<false>, ggrid, dim3(256), 0, st, keys, L, j, np2 / 2);
    hipLaunchKernelGGL(seg_local_kernel<false>, dim3(chunks, n_segs), dim3(kPlanThreads), lds, st, keys, L);
  }
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int launch_plan(const int32_t* d_users, const int32_t* d_items, const int32_t* d_third,
                int64_t n_total, int batch, int n_cls, int n_users, uint64_t* d_skey,
                hipStream_t st) {
  if ((int64_t)(n_cls - 1) * batch > kPlanMaxKeys) {
    // a batch's keys outgrow one workgroup's LDS: write them unsorted, then the segmented sort
    NR_REQUIRE((int64_t)n_cls * batch < (1ll << 31), NR_ERR_UNSUPPORTED, "bpr_plan: batch %d too large", batch);
    hipLaunchKernelGGL(plan_fill_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, st, d_users,
                       d_items, d_third, n_total, batch, n_cls, n_users, d_skey);
    NR_LAUNCH_CHECK();
    const int64_t n_batches = (n_total + batch - 1) / batch;
    NR_REQUIRE(2 * n_batches <= 65535, NR_ERR_UNSUPPORTED, "bpr_plan: %lld batches of %d in one call",
               (long long)n_batches, batch);
    const int nb_max = (int)(n_total < (int64_t)batch ? n_total : (int64_t)batch);
    return sort_segments(d_skey, SegLayout{n_total, batch, n_cls}, (int)(2 * n_batches), (n_cls - 1) * nb_max, st);
  }
  NR_TRY(lds_attr_once((const void*)bpr_plan_kernel, 0));
  const int nb_max = (int)(n_total < (int64_t)batch ? n_total : (int64_t)batch);
  const int np2_user = plan_pow2(nb_max), np2_item = plan_pow2((n_cls - 1) * nb_max);
  const int64_t n_batches = (n_total + batch - 1) / batch;
  hipLaunchKernelGGL(bpr_plan_kernel, dim3((unsigned)n_batches, 2), dim3(kPlanThreads),
                     (size_t)np2_item * sizeof(uint64_t), st, d_users, d_items, d_third, n_total,
                     batch, n_cls, n_users, np2_user, np2_item, d_skey);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

// the plan to use: the caller's, or one sorted now into the tail of the work buffer
// (d_work = [2·batch loss terms][3·batch keys] as floats: 8·batch floats in all)
int resolve_plan(const uint64_t* d_plan, const int32_t* d_users, const int32_t* d_items,
                 const int32_t* d_third, int batch, int n_cls, int n_users, float* d_work,
                 hipStream_t st, const uint64_t** out) {
  if (d_plan) {
    *out = d_plan;
    return NR_OK;
  }
  uint64_t* own = (uint64_t*)(d_work + 2 * (size_t)batch);
  NR_REQUIRE(((uintptr_t)own & 7u) == 0, NR_ERR_ARG, "work buffer is not 8-byte aligned");
  int rc = launch_plan(d_users, d_items, d_third, batch, batch, n_cls, n_users, own, st);
  *out = own;
  return rc;
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

int nrhip_bpr_plan(const int32_t* d_users, const int32_t* d_items, const int32_t* d_third,
                   int64_t n_total, int batch, int n_users, uint64_t* d_plan_out, void* stream) {
  NR_REQUIRE(d_users && d_items && d_plan_out, NR_ERR_ARG, "bpr_plan: null pointer argument");
  NR_REQUIRE(n_total >= 0 && batch >= 1 && n_users >= 0, NR_ERR_ARG, "bpr_plan: bad sizes");
  if (n_total == 0) return NR_OK;
  return launch_plan(d_users, d_items, d_third, n_total, batch, d_third ? 3 : 2, n_users,
                     d_plan_out, (hipStream_t)stream);
}

/* In-place ascending sort of n 64-bit keys: one workgroup's LDS bitonic network up to 16384 keys, the
 * segmented multi-workgroup network beyond. */
int nrhip_sort_u64(uint64_t* d_keys, int n, void* stream) {
  NR_REQUIRE(d_keys && n >= 0, NR_ERR_ARG, "sort_u64: bad arguments");
  if (n <= 1) return NR_OK;
  if (n > kPlanMaxKeys) return sort_segments(d_keys, SegLayout{n, 0, 0}, 1, n, (hipStream_t)stream);
  NR_TRY(lds_attr_once((const void*)sort_u64_kernel, 1));
  const int np2 = plan_pow2(n);
  hipLaunchKernelGGL(sort_u64_kernel, dim3(1), dim3(kPlanThreads), (size_t)np2 * sizeof(uint64_t),
                     (hipStream_t)stream, d_keys, n, np2);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* In-place ascending sort of n_segs independent segments in ONE launch: segment s = d_keys[d_seg_off[s] ..
 * d_seg_off[s] + d_seg_len[s]), every length <= max_len <= 16384 (one workgroup's LDS per segment). */
int nrhip_sort_u64_segments(uint64_t* d_keys, const int64_t* d_seg_off, const int32_t* d_seg_len, int n_segs,
                            int max_len, void* stream) {
  NR_REQUIRE(n_segs >= 0 && max_len >= 0 && (n_segs == 0 || (d_keys && d_seg_off && d_seg_len)), NR_ERR_ARG,
             "sort_u64_segments: bad arguments");
  NR_REQUIRE(max_len <= kPlanMaxKeys, NR_ERR_UNSUPPORTED, "sort_u64_segments: segments of %d keys (> %d)", max_len,
             kPlanMaxKeys);
  if (n_segs == 0 || max_len <= 1) return NR_OK;
  static std::atomic<bool> attr[64];
  int dev = 0;
  NR_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr[dev].load(std::memory_order_acquire)) {
    NR_CHECK_HIP(hipFuncSetAttribute((const void*)sort_u64_segments_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kPlanMaxKeys * (int)sizeof(uint64_t)));
    if (dev >= 0 && dev < 64) attr[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(sort_u64_segments_kernel, dim3((unsigned)n_segs), dim3(kPlanThreads),
                     (size_t)plan_pow2(max_len) * sizeof(uint64_t), (hipStream_t)stream, d_keys, d_seg_off, d_seg_len);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* d_dst[row] = sum of the source rows of the row's run in the SORTED keys (row << 32 | position),
 * added in key order (the first one stored, the rest added); source row of a key =
 * d_src[d_index_of_pos[position]].  Rows of d_dst outside the keys are left alone. */
static int rows_sum_sorted_launch(const uint64_t* d_sorted_keys, int n, const int32_t* d_index_of_pos, int d,
                                  const float* d_src, int64_t ld_src, float* d_dst, const float* d_src_b, int64_t ld_b,
                                  float* d_dst_b, void* stream) {
  if (n == 0) return NR_OK;
  dim3 grid((n + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (d <= 64)
    hipLaunchKernelGGL(rows_sum_sorted_kernel<1>, grid, block, 0, st, d_sorted_keys, n, d_index_of_pos, d,
                       d_src, ld_src, d_dst, d_src_b, ld_b, d_dst_b);
  else if (d <= 128)
    hipLaunchKernelGGL(rows_sum_sorted_kernel<2>, grid, block, 0, st, d_sorted_keys, n, d_index_of_pos, d,
                       d_src, ld_src, d_dst, d_src_b, ld_b, d_dst_b);
  else
    hipLaunchKernelGGL(rows_sum_sorted_kernel<4>, grid, block, 0, st, d_sorted_keys, n, d_index_of_pos, d,
                       d_src, ld_src, d_dst, d_src_b, ld_b, d_dst_b);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_rows_sum_sorted(const uint64_t* d_sorted_keys, int n, const int32_t* d_index_of_pos, int d,
                          const float* d_src, int64_t ld_src, float* d_dst, void* stream) {
  NR_REQUIRE(d_sorted_keys && d_index_of_pos && d_src && d_dst && n >= 0 && d >= 1 && d <= 256 &&
                 ld_src >= d, NR_ERR_ARG, "rows_sum_sorted: bad arguments");
  return rows_sum_sorted_launch(d_sorted_keys, n, d_index_of_pos, d, d_src, ld_src, d_dst, nullptr, 0, nullptr, stream);
}

/* two tables along the same runs in one launch: d_dst_a[row] from d_src_a, d_dst_b[row] from d_src_b */
int nrhip_rows_sum_sorted2(const uint64_t* d_sorted_keys, int n, const int32_t* d_index_of_pos, int d,
                           const float* d_src_a, int64_t ld_a, float* d_dst_a, const float* d_src_b, int64_t ld_b,
                           float* d_dst_b, void* stream) {
  NR_REQUIRE(d_sorted_keys && d_index_of_pos && d_src_a && d_dst_a && d_src_b && d_dst_b && n >= 0 && d >= 1 &&
                 d <= 256 && ld_a >= d && ld_b >= d, NR_ERR_ARG, "rows_sum_sorted2: bad arguments");
  return rows_sum_sorted_launch(d_sorted_keys, n, d_index_of_pos, d, d_src_a, ld_a, d_dst_a, d_src_b, ld_b, d_dst_b,
                                stream);
}

#define NR_BY_WIDTH(KERNEL, ...)                                                          \
  do {                                                                                    \
    if (d <= 64) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, st, __VA_ARGS__);        \
    else if (d <= 128) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, st, __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, st, __VA_ARGS__);                \
  } while (0)
#define NR_BY_WIDTH2(KERNEL, FLAG, ...)                                                         \
  do {                                                                                          \
    if (d <= 64) hipLaunchKernelGGL((KERNEL<1, FLAG>), grid, block, 0, st, __VA_ARGS__);        \
    else if (d <= 128) hipLaunchKernelGGL((KERNEL<2, FLAG>), grid, block, 0, st, __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<4, FLAG>), grid, block, 0, st, __VA_ARGS__);                \
  } while (0)

static int pairwise_mf_grad(const char* who, const float* d_P, const float* d_Q, int d, int n_users,
                            const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                            int batch, float reg, int loss_kind, float* d_GP, float* d_GQ,
                            float* d_work, float* d_loss2, const uint64_t* d_plan, void* stream) {
  NR_REQUIRE(d_P && d_Q && d_users && d_pos && d_neg && d_GP && d_GQ && d_work && d_loss2,
             NR_ERR_ARG, "%s: null pointer argument", who);
  NR_REQUIRE(d >= 1 && d <= 256, NR_ERR_UNSUPPORTED, "%s: embedding dim %d outside 1..256", who, d);
  NR_REQUIRE(batch >= 0 && n_users >= 0, NR_ERR_ARG, "%s: negative batch / n_users", who);
  NR_REQUIRE(loss_kind >= nr::NR_PAIR_BPR && loss_kind <= nr::NR_PAIR_SQUARE, NR_ERR_ARG,
             "%s: unknown pairwise loss %d (0 bpr, 1 hinge, 2 square)", who, loss_kind);
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0) {
    NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
    return NR_OK;
  }
  float* t_mf = d_work;
  float* t_l2 = d_work + batch;
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  {
    const uint64_t* plan = nullptr;
    int rc = resolve_plan(d_plan, d_users, d_pos, d_neg, batch, 3, n_users, d_work, st, &plan);
    if (rc != NR_OK) return rc;
    const int n_occ = 3 * batch;
    dim3 grid((n_occ + kOccWaves - 1) / kOccWaves), block(kOccWaves * NR_WAVE);
    NR_BY_WIDTH2(mf_grad_sorted_kernel, true, d_P, d_Q, d, n_users, d_users, d_pos,
                 (const void*)d_neg, batch, plan, n_occ, reg, 1.0f, loss_kind, d_GP, d_GQ, t_mf, t_l2,
                 d_loss2, done, LazyTables{});
  }
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_bpr_mf_grad(const float* d_P, const float* d_Q, int d, int n_users,
                      const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                      float reg, float* d_GP, float* d_GQ, float* d_work, float* d_loss2,
                      const uint64_t* d_plan, void* stream) {
  return pairwise_mf_grad("bpr_mf_grad", d_P, d_Q, d, n_users, d_users, d_pos, d_neg, batch, reg,
                          nr::NR_PAIR_BPR, d_GP, d_GQ, d_work, d_loss2, d_plan, stream);
}

/* nrhip_bpr_mf_grad on a table kept by nrhip_adam_sparse_tf_lazy: rows are read as of step t - 1
 * (missed zero-gradient steps replayed in registers) and stamped d_stamp[row] = t.  d_table =
 * [n_users + n_items][d] (users first), d_m / d_v / d_last / d_alpha_tab as in the optimiser call.
 * A plan is required (d_plan != NULL). */
int nrhip_bpr_mf_grad_lazy(const float* d_table, const float* d_m, const float* d_v,
                           const int32_t* d_last, const float* d_alpha_tab, int32_t* d_stamp, int t,
                           float beta1, float beta2, float eps, int d, int n_users,
                           const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                           float reg, float* d_G, float* d_work, float* d_loss2, const uint64_t* d_plan,
                           void* stream) {
  NR_REQUIRE(d_table && d_m && d_v && d_last && d_alpha_tab && d_stamp && d_users && d_pos && d_neg &&
                 d_G && d_work && d_loss2 && d_plan, NR_ERR_ARG, "bpr_mf_grad_lazy: null pointer argument");
  NR_REQUIRE(d >= 1 && d <= 256 && batch >= 0 && n_users >= 0 && t >= 1, NR_ERR_ARG,
             "bpr_mf_grad_lazy: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0) {
    NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
    return NR_OK;
  }
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  const LazyTables lz{d_m, d_v, d_last, d_alpha_tab, d_stamp, t, beta1, beta2, 1.0f - beta1, 1.0f - beta2, eps};
  const float* d_P = d_table;
  const float* d_Q = d_table + (size_t)n_users * d;
  float* d_GP = d_G;
  float* d_GQ = d_G + (size_t)n_users * d;
  float* t_mf = d_work;
  float* t_l2 = d_work + batch;
  const int n_occ = 3 * batch;
  dim3 grid((n_occ + kOccWaves - 1) / kOccWaves), block(kOccWaves * NR_WAVE);
  if (d <= 64)
    hipLaunchKernelGGL((mf_grad_sorted_kernel<1, true, true>), grid, block, 0, st, d_P, d_Q, d, n_users,
                       d_users, d_pos, (const void*)d_neg, batch, d_plan, n_occ, reg, 1.0f,
                       (int)nr::NR_PAIR_BPR, d_GP, d_GQ, t_mf, t_l2, d_loss2, done, lz);
  else if (d <= 128)
    hipLaunchKernelGGL((mf_grad_sorted_kernel<2, true, true>), grid, block, 0, st, d_P, d_Q, d, n_users,
                       d_users, d_pos, (const void*)d_neg, batch, d_plan, n_occ, reg, 1.0f,
                       (int)nr::NR_PAIR_BPR, d_GP, d_GQ, t_mf, t_l2, d_loss2, done, lz);
  else
    hipLaunchKernelGGL((mf_grad_sorted_kernel<4, true, true>), grid, block, 0, st, d_P, d_Q, d, n_users,
                       d_users, d_pos, (const void*)d_neg, batch, d_plan, n_occ, reg, 1.0f,
                       (int)nr::NR_PAIR_BPR, d_GP, d_GQ, t_mf, t_l2, d_loss2, done, lz);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* BPR-MF step in one launch (mf_fused_step_kernel): gradient (MF.py:57-72) + TF-1.12 sparse Adam by
 * exact lazy replay (util/learner.py:9-10).  d_W / d_M / d_V: TWO copies of the [n_rows = n_users +
 * n_items][d] table / moments, copy 1 at + n_rows * d; d_tw: int32 [n_rows][2], the step each copy is
 * current as of ({0, -1} before the first step); d_inb: int32 [n_rows], zero before the first step.
 * d_plan: the batch's sorted occurrences (required); d_next_plan / n_next_occ: the following batch's (or
 * NULL / 0); batch_marked != 0: the previous call (step t - 1) was given THIS batch's plan as its
 * d_next_plan, unchanged since — its rows are marked already; 0: a small launch marks them first.
 * d_alpha_tab[s] = lr_s (>= t + 1 entries). */
int nrhip_bpr_mf_step_fused(float* d_W, float* d_M, float* d_V, int32_t* d_tw, int32_t* d_inb,
                            const float* d_alpha_tab, int t, float beta1, float beta2, float eps, int d,
                            int n_users, int n_items, const int32_t* d_users, const int32_t* d_pos,
                            const int32_t* d_neg, int batch, float reg, float* d_work, float* d_loss2,
                            const uint64_t* d_plan, int batch_marked, const uint64_t* d_next_plan,
                            int n_next_occ, int period, void* stream) {
  // d_loss2 == NULL: the per-triplet terms stay in d_work ([batch] mf, [batch] l2) for nrhip_loss_reduce_steps
  NR_REQUIRE(d_W && d_M && d_V && d_tw && d_inb && d_alpha_tab && d_users && d_pos && d_neg && d_work &&
                 (batch == 0 || d_plan), NR_ERR_ARG, "bpr_mf_step_fused: null pointer argument");
  NR_REQUIRE(d >= 1 && d <= 256 && batch >= 0 && n_users >= 0 && n_items >= 0 && t >= 1 && period >= 1 &&
                 n_next_occ >= 0 && (n_next_occ == 0 || d_next_plan), NR_ERR_ARG, "bpr_mf_step_fused: bad sizes");
  const int64_t rows = (int64_t)n_users + n_items;
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0 && d_loss2) NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  const FusedTables ft{d_W, d_M, d_V, d_tw, d_inb, d_alpha_tab, rows, t, beta1, beta2, 1.0f - beta1,
                       1.0f - beta2, eps};
  const int n_occ = 3 * batch;
  if (n_occ > 0 && !batch_marked) {
    hipLaunchKernelGGL(mf_fused_mark_kernel, dim3((n_occ + 255) / 256), dim3(256), 0, st, d_plan, n_occ, d_inb, t);
    NR_LAUNCH_CHECK();
  }
  const int occ_blocks = (n_occ + kOccWaves - 1) / kOccWaves;
  const int64_t maint = (int64_t)n_next_occ + (rows + period - 1) / period;
  dim3 grid((unsigned)(occ_blocks + (maint + kOccWaves - 1) / kOccWaves)), block(kOccWaves * NR_WAVE);
  if (grid.x == 0) return NR_OK;
  float* loss_out = batch ? d_loss2 : nullptr;          // no batch: nothing to reduce
#define NR_FUSED(CPL)                                                                                   \
  hipLaunchKernelGGL(mf_fused_step_kernel<CPL>, grid, block, 0, st, ft, d, n_users, d_users, d_pos, d_neg,  \
                     batch, d_plan, n_occ, d_next_plan, n_next_occ, period, occ_blocks, reg, d_work,        \
                     d_work + batch, loss_out, done)
  if (d <= 64) NR_FUSED(1); else if (d <= 128) NR_FUSED(2); else NR_FUSED(4);
#undef NR_FUSED
  NR_LAUNCH_CHECK();
  return NR_OK;
}

#ifdef NR_MF_TIMELINE
int nrhip_mf_timeline(unsigned long long* h_out) {
  NR_CHECK_HIP(hipDeviceSynchronize());
  NR_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_mf_dbg), sizeof(unsigned long long) * 1024 * 8));
  return NR_OK;
}
#endif

/* every row of the fused step's tables brought to step t, in copy 0 */
int nrhip_bpr_mf_fused_flush(float* d_W, float* d_M, float* d_V, int32_t* d_tw, const float* d_alpha_tab,
                             int t, float beta1, float beta2, float eps, int d, int64_t n_rows, void* stream) {
  NR_REQUIRE(d_W && d_M && d_V && d_tw && d_alpha_tab && d >= 1 && d <= 256 && n_rows >= 0 && t >= 0, NR_ERR_ARG,
             "bpr_mf_fused_flush: bad arguments");
  if (n_rows == 0) return NR_OK;
  const FusedTables ft{d_W, d_M, d_V, d_tw, nullptr, d_alpha_tab, n_rows, t, beta1, beta2, 1.0f - beta1,
                       1.0f - beta2, eps};
  dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (d <= 64) hipLaunchKernelGGL(mf_fused_flush_kernel<1>, grid, block, 0, st, ft, d);
  else if (d <= 128) hipLaunchKernelGGL(mf_fused_flush_kernel<2>, grid, block, 0, st, ft, d);
  else hipLaunchKernelGGL(mf_fused_flush_kernel<4>, grid, block, 0, st, ft, d);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Loss pairs of n_steps steps whose kernels were launched with d_loss2 = NULL: step k's per-triplet terms lie at
 * d_terms + k * 2 * batch (the last step holds n_last <= batch triplets).  The same fixed-order sums as the
 * in-kernel reduction, bit for bit. */
int nrhip_loss_reduce_steps(const float* d_terms, int n_steps, int batch, int n_last, float reg, float* d_loss2,
                            void* stream) {
  NR_REQUIRE(d_terms && d_loss2 && n_steps >= 0 && batch >= 1 && n_last >= 0 && n_last <= batch, NR_ERR_ARG,
             "loss_reduce_steps: bad arguments");
  if (n_steps == 0) return NR_OK;
  hipLaunchKernelGGL(loss_reduce_steps_kernel, dim3(n_steps), dim3(256), 0, (hipStream_t)stream, d_terms, batch,
                     n_last, reg, d_loss2);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_pairwise_mf_grad(const float* d_P, const float* d_Q, int d, int n_users,
                           const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg,
                           int batch, float reg, int loss_kind, float* d_GP, float* d_GQ,
                           float* d_work, float* d_loss2, const uint64_t* d_plan, void* stream) {
  return pairwise_mf_grad("pairwise_mf_grad", d_P, d_Q, d, n_users, d_users, d_pos, d_neg, batch,
                          reg, loss_kind, d_GP, d_GQ, d_work, d_loss2, d_plan, stream);
}

int nrhip_pointwise_mf_grad(const float* d_P, const float* d_Q, int d, int n_users,
                            const int32_t* d_users, const int32_t* d_items, const float* d_labels,
                            int batch, float reg, int loss_kind, float* d_GP, float* d_GQ,
                            float* d_work, float* d_loss2, const uint64_t* d_plan, void* stream) {
  NR_REQUIRE(d_P && d_Q && d_users && d_items && d_labels && d_GP && d_GQ && d_work && d_loss2,
             NR_ERR_ARG, "pointwise_mf_grad: null pointer argument");
  NR_REQUIRE(d >= 1 && d <= 256, NR_ERR_UNSUPPORTED,
             "pointwise_mf_grad: embedding dim %d outside 1..256", d);
  NR_REQUIRE(batch >= 0 && n_users >= 0, NR_ERR_ARG, "pointwise_mf_grad: negative batch / n_users");
  NR_REQUIRE(loss_kind == nr::NR_POINT_CROSS_ENTROPY || loss_kind == nr::NR_POINT_SQUARE, NR_ERR_ARG,
             "pointwise_mf_grad: unknown pointwise loss %d (0 cross_entropy, 1 square)", loss_kind);
  hipStream_t st = (hipStream_t)stream;
  if (batch == 0) {
    NR_CHECK_HIP(hipMemsetAsync(d_loss2, 0, 2 * sizeof(float), st));
    return NR_OK;
  }
  float* t_mf = d_work;
  float* t_l2 = d_work + batch;
  // tf.losses.sigmoid_cross_entropy averages over the batch; the squared loss is a plain sum
  const float scale = loss_kind == nr::NR_POINT_CROSS_ENTROPY ? 1.0f / (float)batch : 1.0f;
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  {
    const uint64_t* plan = nullptr;
    int rc = resolve_plan(d_plan, d_users, d_items, nullptr, batch, 2, n_users, d_work, st, &plan);
    if (rc != NR_OK) return rc;
    const int n_occ = 2 * batch;
    dim3 grid((n_occ + kOccWaves - 1) / kOccWaves), block(kOccWaves * NR_WAVE);
    NR_BY_WIDTH2(mf_grad_sorted_kernel, false, d_P, d_Q, d, n_users, d_users, d_items,
                 (const void*)d_labels, batch, plan, n_occ, reg, scale, loss_kind, d_GP, d_GQ, t_mf,
                 t_l2, d_loss2, done, LazyTables{});
  }
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
                         int batch, float reg, float* d_Gstar, float* d_Greg, float* d_work,
                         float* d_loss2, float grad_div, const uint64_t* d_plan, void* stream,
                         const float* d_given = nullptr) {
  NR_REQUIRE(d_Esum && d_E0 && d_users && d_pos && d_neg && d_Gstar && d_Greg && d_work,
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
  float* t_mf = d_work;
  float* t_l2 = d_work + batch;
  const float lp1 = (float)(n_layers + 1);
  unsigned* done = next_done_counter();
  NR_REQUIRE(done, NR_ERR_HIP, "loss reduction: device counter pool unavailable");
  {
    const uint64_t* plan = nullptr;
    int rc = resolve_plan(d_plan, d_users, d_pos, d_neg, batch, 3, n_users, d_work, st, &plan);
    if (rc != NR_OK) return rc;
    const int n_occ = 3 * batch;
    dim3 grid((n_occ + kOccWaves - 1) / kOccWaves), block(kOccWaves * NR_WAVE);
    NR_BY_WIDTH(lightgcn_grad_sorted_kernel, d_Esum, d_E0, n_users, d, lp1, d_users, d_pos, d_neg,
                batch, plan, n_occ, reg, d_Gstar, d_Greg, t_mf, t_l2, grad_div, d_loss2, done, d_given);
  }
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_lightgcn_bpr_grad(const float* d_Esum, const float* d_E0, int n_users, int d,
                            int n_layers, const int32_t* d_users, const int32_t* d_pos,
                            const int32_t* d_neg, int batch, float reg, float* d_Gstar,
                            float* d_Greg, float* d_work, float* d_loss2, const uint64_t* d_plan,
                            void* stream) {
  return lightgcn_head(d_Esum, d_E0, n_users, d, n_layers, d_users, d_pos, d_neg, batch, reg,
                       d_Gstar, d_Greg, d_work, d_loss2, 1.0f, d_plan, stream);
}

/* Same head, writing dLoss/dE* already divided by (n_layers+1) — the H = Gstar/(L+1) of the
 * backward pass — into d_H.  Only when L+1 is a power of two: dividing every term is then
 * bit-identical to dividing the sum, and the rows_div pass disappears. */
int nrhip_lightgcn_bpr_grad_h(const float* d_Esum, const float* d_E0, int n_users, int d,
                              int n_layers, const int32_t* d_users, const int32_t* d_pos,
                              const int32_t* d_neg, int batch, float reg, float* d_H,
                              float* d_Greg, float* d_work, float* d_loss2, const uint64_t* d_plan,
                              void* stream) {
  NR_REQUIRE(n_layers >= 0 && ((n_layers + 1) & n_layers) == 0, NR_ERR_ARG,
             "lightgcn_bpr_grad_h: n_layers + 1 = %d is not a power of two", n_layers + 1);
  return lightgcn_head(d_Esum, d_E0, n_users, d, n_layers, d_users, d_pos, d_neg, batch, reg, d_H,
                       d_Greg, d_work, d_loss2, (float)(n_layers + 1), d_plan, stream);
}

/* Column-sharded tables (neurec_amd/colshard.py; LightGCN.py:157-166 when every rank holds d of the D embedding
 * columns): the head in two halves around ONE small exchange.
 *   nrhip_lightgcn_partial_dots   d_out[b] = (<e_u,e_i>, <e_u,e_j>, l2 term) of triplet b on this rank's columns;
 *   nrhip_partials_sum            d_given[k] = sum over ranks r ascending of d_parts[r][k] (n = 3*batch floats per rank):
 *                                 the same additions in the same order on every rank;
 *   nrhip_lightgcn_bpr_grad_given the head of nrhip_lightgcn_bpr_grad(_h) with x_b and the l2 term taken from d_given
 *                                 (divided != 0: rows already divided by n_layers + 1, a power of two). */
int nrhip_lightgcn_partial_dots(const float* d_Esum, const float* d_E0, int n_users, int d, int n_layers,
                                const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                                float* d_out, void* stream) {
  NR_REQUIRE(d_Esum && d_E0 && d_users && d_pos && d_neg && d_out && d >= 1 && d <= 256 && batch >= 0 &&
                 n_layers >= 0, NR_ERR_ARG, "lightgcn_partial_dots: bad arguments");
  if (batch == 0) return NR_OK;
  dim3 grid((batch + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  NR_BY_WIDTH(lightgcn_partial_dots_kernel, d_Esum, d_E0, n_users, d, (float)(n_layers + 1), d_users, d_pos, d_neg,
              batch, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_partials_sum(const float* d_parts, int world, int n, float* d_given, void* stream) {
  NR_REQUIRE(d_parts && d_given && world >= 1 && n >= 0, NR_ERR_ARG, "partials_sum: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(partials_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_parts, world,
                     n, d_given);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_lightgcn_bpr_grad_given(const float* d_Esum, const float* d_E0, int n_users, int d, int n_layers,
                                  const int32_t* d_users, const int32_t* d_pos, const int32_t* d_neg, int batch,
                                  float reg, float* d_Gstar, float* d_Greg, float* d_work, float* d_loss2,
                                  const uint64_t* d_plan, const float* d_given, int divided, void* stream) {
  NR_REQUIRE(d_given, NR_ERR_ARG, "lightgcn_bpr_grad_given: null d_given");
  NR_REQUIRE(!divided || ((n_layers + 1) & n_layers) == 0, NR_ERR_ARG,
             "lightgcn_bpr_grad_given: divided needs n_layers + 1 = %d to be a power of two", n_layers + 1);
  return lightgcn_head(d_Esum, d_E0, n_users, d, n_layers, d_users, d_pos, d_neg, batch, reg, d_Gstar, d_Greg,
                       d_work, d_loss2, divided ? (float)(n_layers + 1) : 1.0f, d_plan, stream, d_given);
}

}  // extern "C"
