// nr_core.h — per-thread arithmetic of the NeuRec hot path, shared by the HIP kernels.
//
// Everything in this header is a pure function of its arguments (no wave-level
// cooperation, no memory side effects beyond the pointers passed in), written
// so it compiles both as device code under hipcc and as host code under g++.
// The host build exists only for tests/hostcheck (a CPU unit-test harness for
// the exact same source the kernels inline); the product never runs it.
//
// Reference behaviour restated here (paths relative to the NeuRec tree):
//   * metric formulas            evaluator/backend/cpp/include/metric.h:17-117
//   * top-K tie behaviour        evaluator/backend/cpp/include/evaluate.h:38-42
//                                (std::partial_sort_copy, libstdc++ [EXT])
//   * rejection sampling         util/cython/random_choice.pyx:20-62
//   * Adam (TF-1.12 semantics)   util/learner.py:9-10, LightGCN.py:130 [EXT]
//   * BPR loss                   util/learner.py:19-22, util/tool.py:220-224
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define NR_HD __host__ __device__ __forceinline__
#else
#define NR_HD inline
#endif

namespace nr {

// ----------------------------------------------------------------------------
// Ordered keys: (score, index) -> one u64 whose unsigned order is
// "higher score first, then lower index first".
// ----------------------------------------------------------------------------
NR_HD uint32_t f32_bits(float f) {
  union { float f; uint32_t u; } c; c.f = f; return c.u;
}
NR_HD float bits_f32(uint32_t u) {
  union { float f; uint32_t u; } c; c.u = u; return c.f;
}
NR_HD uint32_t order_f32(float f) {
  // -0.0 compares equal to +0.0 in the reference comparator; fold it first.
  f = f + 0.0f;
  uint32_t u = f32_bits(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
NR_HD float unorder_f32(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return bits_f32(u);
}
NR_HD uint64_t pack_key(float score, uint32_t idx) {
  return ((uint64_t)order_f32(score) << 32) | (uint64_t)(0xffffffffu - idx);
}
NR_HD uint32_t key_index(uint64_t k) { return 0xffffffffu - (uint32_t)(k & 0xffffffffu); }
NR_HD uint32_t key_order(uint64_t k) { return (uint32_t)(k >> 32); }

// ----------------------------------------------------------------------------
// Exact emulation of libstdc++'s heap-based std::partial_sort_copy on an index
// vector with comparator  comp(a,b) := score[a] > score[b]   (evaluate.h:40,
// arg_topk.h:22).  Used only for rows where ties make the parallel selection
// ambiguous.  The heap stores (score, index) pairs; `comp` looks only at the
// score, exactly like the reference lambda.
// The routines mirror GCC 11 bits/stl_heap.h (__adjust_heap / __push_heap /
// __make_heap / __sort_heap) and bits/stl_algo.h (__partial_sort_copy) [EXT].
// ----------------------------------------------------------------------------
// The heap routines are written against a small view interface — gv(i) / gi(i): score / item of slot i; set(i, v,
// id); mov(dst, src) — so that the same statement sequence runs on a heap in memory (HeapView: host checks, the
// LDS heap of long sort lengths) and on a heap held one slot per lane in registers (eval_select.hip: RegHeap).
struct HeapView {
  float* val;   // heap scores
  int*   idx;   // heap item indices
  NR_HD float gv(int i) const { return val[i]; }
  NR_HD int gi(int i) const { return idx[i]; }
  NR_HD void set(int i, float v, int id) { val[i] = v; idx[i] = id; }
  NR_HD void mov(int dst, int src) { val[dst] = val[src]; idx[dst] = idx[src]; }
};

template <class H>
NR_HD void heap_push(H& h, int hole, int top, float v, int id) {
  int parent = (hole - 1) / 2;
  while (hole > top && h.gv(parent) > v) {           // comp(parent, value)
    h.mov(hole, parent);
    hole = parent;
    parent = (hole - 1) / 2;
  }
  h.set(hole, v, id);
}

template <class H>
NR_HD void heap_adjust(H& h, int hole, int len, float v, int id) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (h.gv(child) > h.gv(child - 1)) child--;       // comp(child, child-1)
    h.mov(hole, child);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    h.mov(hole, child - 1);
    hole = child - 1;
  }
  heap_push(h, hole, top, v, id);
}

template <class H>
NR_HD void heap_make(H& h, int len) {
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    float v = h.gv(parent);
    int id = h.gi(parent);
    heap_adjust(h, parent, len, v, id);
    if (parent == 0) return;
    parent--;
  }
}

// Offer element (v,id) of the input stream to a full heap: replaces the heap
// root iff comp(element, root), i.e. v > root score (strictly).
template <class H>
NR_HD bool heap_offer(H& h, int len, float v, int id) {
  if (v > h.gv(0)) { heap_adjust(h, 0, len, v, id); return true; }
  return false;
}

template <class H>
NR_HD void heap_sort(H& h, int len) {
  int last = len;
  while (last > 1) {
    --last;
    float v = h.gv(last);
    int id = h.gi(last);
    h.mov(last, 0);
    heap_adjust(h, 0, last, v, id);
  }
}

// Whole algorithm on one row, sequentially (host tests + single-thread use).
NR_HD void partial_sort_copy_emul(const float* score, int n, int sort_len,
                                  float* hv, int* hi) {
  HeapView h{hv, hi};
  int m = sort_len < n ? sort_len : n;
  for (int i = 0; i < m; ++i) { hv[i] = score[i]; hi[i] = i; }
  heap_make(h, m);
  for (int i = m; i < n; ++i) heap_offer(h, m, score[i], i);
  heap_sort(h, m);
}

// ----------------------------------------------------------------------------
// Ranking metrics, cumulative @1..K (metric.h:17-109).  `hit(k)` tells whether
// the item at rank k is a test item, `T` is the number of test items of the
// user, inv_log2[k] = 1.0/log2(k+2) computed on the host with libm (the same
// double the reference computes inline).
// Metric ids follow metric.h:111-117: 1 Precision, 2 Recall, 3 MAP, 4 NDCG,
// 5 MRR.  The float/double mix below is the reference's, operation for
// operation, so results are bit-identical.
// ----------------------------------------------------------------------------
template <class HitFn>
NR_HD void metric_eval(int metric_id, HitFn hit, int K, int T,
                       const double* inv_log2, float* out) {
  if (metric_id == 1) {                       // precision, metric.h:17-28
    int hits = 0;
    for (int i = 0; i < K; ++i) {
      if (hit(i)) hits += 1;
      out[i] = (float)(1.0 * hits / (double)(unsigned)(i + 1));
    }
  } else if (metric_id == 2) {                // recall, metric.h:31-43
    int hits = 0;
    for (int i = 0; i < K; ++i) {
      if (hit(i)) hits += 1;
      out[i] = (float)(1.0 * hits / (double)T);
    }
  } else if (metric_id == 3) {                // ap, metric.h:46-65
    int hits = 0;
    float pre = 0.f, sum_pre = 0.f, denom = 1.f;
    const float truth_len = (float)T;
    for (int i = 0; i < K; ++i) {
      if (hit(i)) {
        hits += 1;
        pre = (float)(1.0 * hits / (double)(unsigned)(i + 1));
        sum_pre += pre;
      }
      const float ip1 = (float)(unsigned)(i + 1);
      denom = (truth_len < ip1) ? truth_len : ip1;
      out[i] = (hits == 0) ? 0.0f : sum_pre / denom;
    }
  } else if (metric_id == 4) {                // ndcg, metric.h:69-86
    float idcg = 0.f, dcg = 0.f;
    for (int i = 0; i < K; ++i) {
      if (hit(i)) dcg = (float)((double)dcg + inv_log2[i]);
      if (i < T) idcg = (float)((double)idcg + inv_log2[i]);
      out[i] = dcg / idcg;
    }
  } else if (metric_id == 5) {                // mrr, metric.h:89-109
    float rr = 0.f;
    int i = 0;
    for (; i < K; ++i) {
      if (hit(i)) { rr = (float)(1.0 / (double)(unsigned)(i + 1)); break; }
      out[i] = 0.f;
    }
    for (; i < K; ++i) out[i] = rr;
  }
}

// lower-bound membership test in an ascending int32 list
NR_HD bool sorted_contains(const int32_t* a, int n, int32_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < n && a[lo] == x;
}

// ----------------------------------------------------------------------------
// Counter-based RNG for the sampler: splitmix64 seeding + xorshift64* stream.
// (The reference draws from glibc rand(), random_choice.pyx:12-17; that stream
// is an input, not something a GPU reproduces — see DESIGN.md.)
// ----------------------------------------------------------------------------
NR_HD uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
struct XorShift64s {
  uint64_t s;
  NR_HD void seed(uint64_t key, uint64_t stream, uint64_t counter) {
    uint64_t z = splitmix64(key ^ splitmix64(stream * 0xd1342543de82ef95ull + 0x632be59bd9b4e019ull));
    z = splitmix64(z ^ (counter * 0x9e3779b97f4a7c15ull));
    s = z ? z : 0x2545f4914f6cdd1dull;
  }
  NR_HD uint64_t next() {
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    return s * 0x2545f4914f6cdd1dull;
  }
};

// One negative draw: uniform in [0, high) \ exclusion (ascending list), the
// rejection loop of random_choice.pyx:50-54 with replace=True.
// After kMaxRejects rejections (a user who has interacted with almost every
// item) the draw switches to an equivalent closed form — pick r uniformly among
// the high-n_excl allowed ids and locate the r-th one — so the loop is bounded
// on the device.  Returns -1 when nothing is allowed (the reference raises
// ValueError for that case, random_choice.pyx:32-33; the host wrapper does too).
constexpr int kMaxRejects = 64;
NR_HD int32_t nth_allowed(int32_t r, const int32_t* excl, int n_excl) {
  // k = #{i : excl[i] - i <= r}  (excluded ids below the answer); answer = r + k
  int lo = 0, hi = n_excl;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (excl[mid] - mid <= r) lo = mid + 1; else hi = mid;
  }
  return r + lo;
}
NR_HD int32_t draw_negative(XorShift64s& g, int32_t high, const int32_t* excl, int n_excl) {
  if (n_excl >= high) return -1;
  for (int tries = 0; tries < kMaxRejects; ++tries) {
    int32_t a = (int32_t)(g.next() % (uint64_t)high);
    if (!sorted_contains(excl, n_excl, a)) return a;
  }
  int32_t r = (int32_t)(g.next() % (uint64_t)(high - n_excl));
  return nth_allowed(r, excl, n_excl);
}

// ----------------------------------------------------------------------------
// Keyed bijection on [0, n): 4-round Feistel network over 2*hb bits with
// cycle walking.  Stands in for np.random.permutation(E)
// (util/data_iterator.py:59) without materialising or sorting anything.
// ----------------------------------------------------------------------------
NR_HD uint32_t feistel_round(uint32_t x, uint64_t k) {
  uint64_t z = splitmix64(((uint64_t)x << 32 | (uint32_t)k) ^ (k >> 32) * 0xff51afd7ed558ccdull);
  return (uint32_t)(z >> 32) ^ (uint32_t)z;
}
NR_HD uint64_t permute_index(uint64_t i, uint64_t n, uint64_t key) {
  // half-width in bits so that 2^(2*hb) >= n
  int hb = 1;
  while (((uint64_t)1 << (2 * hb)) < n) hb++;
  const uint32_t mask = (hb >= 32) ? 0xffffffffu : (((uint32_t)1 << hb) - 1u);
  uint64_t x = i;
  do {
    uint32_t l = (uint32_t)(x >> hb) & mask, r = (uint32_t)x & mask;
    for (int rd = 0; rd < 4; ++rd) {
      uint32_t f = feistel_round(r, key + 0x9e3779b97f4a7c15ull * (uint64_t)(rd + 1)) & mask;
      uint32_t nl = r, nr = l ^ f;
      l = nl; r = nr;
    }
    x = ((uint64_t)l << hb) | r;
  } while (x >= n);
  return x;
}

// ----------------------------------------------------------------------------
// BPR pieces.  TF-1.12 softplus (used by log_sigmoid) is thresholded [EXT]:
//   softplus(z) = z             if z > -thr
//               = exp(z)        if z <  thr
//               = log1p(exp(z)) otherwise,      thr = log(eps_f32) + 2
// loss_b = -log_sigmoid(x) = softplus(-x);  d loss_b / d x = -sigmoid(-x).
// ----------------------------------------------------------------------------
NR_HD float tf_softplus(float z) {
  const float thr = -13.942385f;   // logf(FLT_EPSILON) + 2
  if (z > -thr) return z;
  if (z < thr) return expf(z);
  return log1pf(expf(z));
}
NR_HD float bpr_loss(float x) { return tf_softplus(-x); }
NR_HD float bpr_dloss(float x) { return -1.0f / (1.0f + expf(x)); }

// ----------------------------------------------------------------------------
// The other losses of util/learner.py, per element, and their derivative w.r.t. the logit.
//   pairwise  (learner.py:19-29), y = x_pos - x_neg, all summed over the batch:
//       bpr    -log_sigmoid(y)
//       hinge  max(y + 1, 0)        (as written in the reference: margin added, not subtracted)
//       square (1 - y)^2
//   pointwise (learner.py:31-41), z = label, x = logit:
//       cross_entropy  tf.losses.sigmoid_cross_entropy = MEAN over the batch of
//                      max(x,0) - x*z + log1p(exp(-|x|));   d/dx = sigmoid(x) - z
//       square         sum of (z - x)^2
// ----------------------------------------------------------------------------
enum { NR_PAIR_BPR = 0, NR_PAIR_HINGE = 1, NR_PAIR_SQUARE = 2 };
enum { NR_POINT_CROSS_ENTROPY = 0, NR_POINT_SQUARE = 1 };

NR_HD float pairwise_loss(int kind, float y) {
  if (kind == NR_PAIR_BPR) return bpr_loss(y);
  if (kind == NR_PAIR_HINGE) return fmaxf(y + 1.0f, 0.0f);
  return (1.0f - y) * (1.0f - y);
}
NR_HD float pairwise_dloss(int kind, float y) {
  if (kind == NR_PAIR_BPR) return bpr_dloss(y);
  if (kind == NR_PAIR_HINGE) return (y + 1.0f > 0.0f) ? 1.0f : 0.0f;
  return -2.0f * (1.0f - y);
}
NR_HD float pointwise_loss(int kind, float z, float x) {
  if (kind == NR_POINT_CROSS_ENTROPY) return fmaxf(x, 0.0f) - x * z + log1pf(expf(-fabsf(x)));
  return (z - x) * (z - x);
}
NR_HD float pointwise_dloss(int kind, float z, float x) {
  if (kind == NR_POINT_CROSS_ENTROPY) return 1.0f / (1.0f + expf(-x)) - z;
  return -2.0f * (z - x);
}

// ----------------------------------------------------------------------------
// Adam, TF-1.12 arithmetic [EXT].  `alpha` = lr*sqrt(1-b2^t)/(1-b1^t) (fp32,
// computed by the caller from running fp32 powers); eps is added outside the
// bias correction.  No fused multiply-adds: each product/sum is rounded like
// the reference's Eigen expressions.
//   dense  (training_ops ApplyAdam):  m += (g-m)*(1-b1); v += (g*g-v)*(1-b2);
//                                     var -= (m*alpha)/(sqrt(v)+eps)
//   sparse (adam.py _apply_sparse_shared, all rows swept every step):
//          m = m*b1 + g*(1-b1); v = v*b2 + (g*g)*(1-b2);
//          var -= alpha*m/(sqrt(v)+eps)
// ----------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define NR_MUL(a, b) __fmul_rn((a), (b))
#define NR_ADD(a, b) __fadd_rn((a), (b))
#define NR_SUB(a, b) __fsub_rn((a), (b))
#else
// host build is compiled with -ffp-contract=off
#define NR_MUL(a, b) ((a) * (b))
#define NR_ADD(a, b) ((a) + (b))
#define NR_SUB(a, b) ((a) - (b))
#endif

NR_HD void adam_dense_tf(float g, float& var, float& m, float& v, float alpha,
                         float one_minus_b1, float one_minus_b2, float eps) {
  m = NR_ADD(m, NR_MUL(NR_SUB(g, m), one_minus_b1));
  v = NR_ADD(v, NR_MUL(NR_SUB(NR_MUL(g, g), v), one_minus_b2));
  var = NR_SUB(var, NR_MUL(m, alpha) / NR_ADD(sqrtf(v), eps));
}
NR_HD void adam_sparse_tf(float g, float& var, float& m, float& v, float alpha,
                          float b1, float b2, float one_minus_b1,
                          float one_minus_b2, float eps) {
  m = NR_ADD(NR_MUL(m, b1), NR_MUL(g, one_minus_b1));
  v = NR_ADD(NR_MUL(v, b2), NR_MUL(NR_MUL(g, g), one_minus_b2));
  var = NR_SUB(var, NR_MUL(alpha, m) / NR_ADD(sqrtf(v), eps));
}

}  // namespace nr
