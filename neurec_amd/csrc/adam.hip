// adam.hip — TF-1.12 Adam sweeps and the small elementwise helpers between
// propagation passes; plus the library-level entry points (error string,
// ABI version, device facts).
//
// Stands in for tf.train.AdamOptimizer(lr).minimize(loss)
// (util/learner.py:9-10 for MF, LightGCN.py:130): see nr_core.h for the exact
// arithmetic of the dense (ApplyAdam) and sparse (_apply_sparse_shared)
// variants.  Both sweep every element of the table each step — that is the
// reference's semantics even for embedding gradients — so the
// kernel is a pure HBM stream: 4 reads + 3 (or 4, with the gradient clear)
// writes of n·4 bytes, 16 B per lane per access.
#include "nr_common.h"
#include <algorithm>
#include <string.h>

namespace {

template <bool SPARSE, bool CLEAR>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ var, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ grad,
                                                   int64_t n, float alpha, float b1, float b2,
                                                   float omb1, float omb2, float eps) {
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 w = reinterpret_cast<float4*>(var)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    if (SPARSE) {
      nr::adam_sparse_tf(g.x, w.x, mm.x, vv.x, alpha, b1, b2, omb1, omb2, eps);
      nr::adam_sparse_tf(g.y, w.y, mm.y, vv.y, alpha, b1, b2, omb1, omb2, eps);
      nr::adam_sparse_tf(g.z, w.z, mm.z, vv.z, alpha, b1, b2, omb1, omb2, eps);
      nr::adam_sparse_tf(g.w, w.w, mm.w, vv.w, alpha, b1, b2, omb1, omb2, eps);
    } else {
      nr::adam_dense_tf(g.x, w.x, mm.x, vv.x, alpha, omb1, omb2, eps);
      nr::adam_dense_tf(g.y, w.y, mm.y, vv.y, alpha, omb1, omb2, eps);
      nr::adam_dense_tf(g.z, w.z, mm.z, vv.z, alpha, omb1, omb2, eps);
      nr::adam_dense_tf(g.w, w.w, mm.w, vv.w, alpha, omb1, omb2, eps);
    }
    reinterpret_cast<float4*>(var)[i] = w;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (CLEAR) reinterpret_cast<float4*>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // tail (n not a multiple of 4)
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float w = var[i], mm = m[i], vv = v[i];
    const float g = grad[i];
    if (SPARSE) nr::adam_sparse_tf(g, w, mm, vv, alpha, b1, b2, omb1, omb2, eps);
    else nr::adam_dense_tf(g, w, mm, vv, alpha, omb1, omb2, eps);
    var[i] = w; m[i] = mm; v[i] = vv;
    if (CLEAR) grad[i] = 0.f;
  }
}


// ---- exact lazy replay of TF-1.12's sparse Adam -----------------------------------
// _apply_sparse_shared decays m and v of EVERY row and moves EVERY row each step; a row the batch
// did not touch sees g = 0:  m <- m·b1 (+0),  v <- v·b2 (+0),  var <- var - lr_s·m/(sqrt(v)+eps).
// That recurrence only involves the row itself, so it can be run late: last[row] remembers the
// last step applied; when step t touches the row, the missed steps last+1..t-1 are replayed in
// registers with the very same instruction sequence (nr::adam_sparse_tf with g = 0 and that step's
// lr_s from a host-made table), then step t is applied with the row's summed gradient.  Identical
// bits, 2.4 MB of touched rows per step instead of a 145 MB sweep.  To bound the replay (a row
// untouched for 10^4 steps would hold one wave for 10^4 iterations), every step also brings the
// rows r = t (mod period) up to date: no row is ever more than `period` steps behind.  A flush
// (plan = NULL, period = 1) brings every row to step t — before tables are read.  Given the NEXT
// batch's plan, the rows that batch will gather are brought to step t as well (zero-gradient replay),
// so that the next gradient kernel finds them current and replays nothing.
// The gradient kernel of step t must see every row it gathers as of step t - 1: it replays the
// same missed steps in registers (bpr.hip: load_row_lazy) and stamps the batch's rows with t, which
// is how a scheduled-row wave here knows to leave a row to the batch wave that owns it.
// One wave per sorted batch occurrence (the first occurrence of a row does the row) + one per
// scheduled row; rows whose m and v are still all zero replay nothing (0·b1 = 0, var - 0 = var).
template <int CPL>
__global__ __launch_bounds__(256) void adam_lazy_kernel(
    float* __restrict__ var, float* __restrict__ m, float* __restrict__ v, float* __restrict__ grad,
    int32_t* __restrict__ last, const int32_t* __restrict__ stamp, int d, int64_t n_rows,
    const uint64_t* __restrict__ skey, int n_occ, const uint64_t* __restrict__ skey_next, int n_next,
    const float* __restrict__ alpha_tab, int t, int period, float b1, float b2, float omb1, float omb2,
    float eps) {
  const int lane = nr_lane();
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int64_t row;
  bool has_grad = false;
  if (w < n_occ) {                                                 // rows of this step's batch
    row = (int64_t)(skey[w] >> 32);
    if (w > 0 && (int64_t)(skey[w - 1] >> 32) == row) return;      // a later occurrence of the row
    has_grad = true;
  } else if (w < (int64_t)n_occ + n_next) {                        // rows the NEXT step's batch will gather:
    const int64_t k = w - n_occ;                                   // brought to step t now, so that its
    row = (int64_t)(skey_next[k] >> 32);                           // gradient kernel finds them current
    if (k > 0 && (int64_t)(skey_next[k - 1] >> 32) == row) return;
  } else {                                                         // scheduled rows: r = t (mod period)
    row = (int64_t)(t % period) + (w - n_occ - n_next) * period;
    if (row >= n_rows) return;
  }
  row = __builtin_amdgcn_readfirstlane((int)row);
  // everything that depends only on the row is requested at once: the claim / stamp, the row's three
  // vectors, and the step sizes of the last 64 steps (lane j: step t - 63 + j; a replay never reaches
  // further back than `period` <= 64 steps unless the caller flushes rarely — then the table is read)
  float wv[CPL], mm[CPL], vv[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + c * NR_WAVE;
    wv[c] = mm[c] = vv[c] = 0.f;
    if (k < d) {
      wv[c] = var[row * d + k];
      mm[c] = m[row * d + k];
      vv[c] = v[row * d + k];
    }
  }
  const int a_lo = t - (NR_WAVE - 1);
  const float a_mine = alpha_tab[max(a_lo + lane, 0)];
  int from;
  if (has_grad) {
    from = __builtin_amdgcn_readfirstlane(last[row]) + 1;
  } else {
    const int st = stamp ? stamp[row] : -1;
    // a row may be both scheduled and in the next batch: whoever raises last[row] to t first owns it
    int old = 0;
    if (lane == 0 && st != t) old = atomicMax(&last[row], t);
    old = __builtin_amdgcn_readfirstlane(old);
    if (st == t || old >= t) return;               // in this step's batch (done above) / already claimed
    from = old + 1;
  }
  const int upto = has_grad ? t - 1 : t;                           // steps replayed with g = 0
  bool quiet = true;
#pragma unroll
  for (int c = 0; c < CPL; ++c) quiet = quiet && mm[c] == 0.f && vv[c] == 0.f;
  const bool replayed = !__all(quiet) && from <= upto;
  if (replayed) {
    if (from >= a_lo && from >= 1) {               // the usual case: the steps are in the lanes already
      for (int s2 = from; s2 <= upto; ++s2) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_mine), s2 - a_lo));
#pragma unroll
        for (int c = 0; c < CPL; ++c) nr::adam_sparse_tf(0.f, wv[c], mm[c], vv[c], a, b1, b2, omb1, omb2, eps);
      }
    } else {
      nr_lazy_replay<CPL>(wv, mm, vv, from, upto, alpha_tab, lane, b1, b2, omb1, omb2, eps);
    }
  }
  if (has_grad) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_mine), NR_WAVE - 1));   // alpha_tab[t]
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int k = lane + c * NR_WAVE;
      float g = 0.f;
      if (k < d) {
        g = grad[row * d + k];
        grad[row * d + k] = 0.f;                                   // re-armed, as the sweep does
      }
      nr::adam_sparse_tf(g, wv[c], mm[c], vv[c], a, b1, b2, omb1, omb2, eps);
    }
  }
  if (has_grad || replayed) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int k = lane + c * NR_WAVE;
      if (k < d) {
        var[row * d + k] = wv[c];
        m[row * d + k] = mm[c];
        v[row * d + k] = vv[c];
      }
    }
  }
  if (has_grad && lane == 0) last[row] = t;
}

// dense ApplyAdam whose gradient is the sum of two buffers (g = g1 + g2, one rounding — the
// same value a separate elementwise add would have produced); neither buffer is modified.
__global__ __launch_bounds__(256) void adam_dense2_kernel(float* __restrict__ var,
                                                          float* __restrict__ m,
                                                          float* __restrict__ v,
                                                          const float* __restrict__ g1,
                                                          const float* __restrict__ g2, int64_t n,
                                                          float alpha, float omb1, float omb2,
                                                          float eps) {
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 w = reinterpret_cast<float4*>(var)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    const float4 a = reinterpret_cast<const float4*>(g1)[i];
    const float4 b = reinterpret_cast<const float4*>(g2)[i];
    nr::adam_dense_tf(__fadd_rn(a.x, b.x), w.x, mm.x, vv.x, alpha, omb1, omb2, eps);
    nr::adam_dense_tf(__fadd_rn(a.y, b.y), w.y, mm.y, vv.y, alpha, omb1, omb2, eps);
    nr::adam_dense_tf(__fadd_rn(a.z, b.z), w.z, mm.z, vv.z, alpha, omb1, omb2, eps);
    nr::adam_dense_tf(__fadd_rn(a.w, b.w), w.w, mm.w, vv.w, alpha, omb1, omb2, eps);
    reinterpret_cast<float4*>(var)[i] = w;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float w = var[i], mm = m[i], vv = v[i];
    nr::adam_dense_tf(__fadd_rn(g1[i], g2[i]), w, mm, vv, alpha, omb1, omb2, eps);
    var[i] = w; m[i] = mm; v[i] = vv;
  }
}

// Row-sparse helpers over a list of row ids (repeats allowed: every operation is idempotent).
// One wave per listed row, lane = column.
__global__ __launch_bounds__(256) void rows_div_kernel(const int32_t* __restrict__ rows,
                                                       int n_listed, int d,
                                                       const float* __restrict__ src, float denom,
                                                       float* __restrict__ dst) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n_listed) return;
  const int64_t r = rows[w];
  for (int k = lane; k < d; k += 64) dst[r * d + k] = src[r * d + k] / denom;
}
// ---- the other optimisers of util/learner.py:2-16, TF-1.12 sparse application ------------------
// (GradientDescent scatter_sub, SparseApplyAdagrad, SparseApplyRMSProp, SparseApplyMomentum):
// only the rows a batch touched move.  grad is a dense [n_rows][d] buffer holding the summed
// row gradients (zero elsewhere); flag[row] != 0 marks the touched rows.  One wave per row;
// the row's gradient and flag are cleared on the way out.
//   gd        var -= lr*g
//   adagrad   a += g*g;                       var -= (lr*g) * rsqrt(a)          (a starts at 1e-8)
//   rmsprop   ms = ms*rho + (g*g)*(1-rho);    mom = mom*momentum + rsqrt(ms+eps)*lr*g;  var -= mom
//             (ms starts at 1, rho = 0.9, momentum = 0, eps = 1e-10: tf.train.RMSPropOptimizer(lr))
//   momentum  a = a*momentum + g;             var -= a*lr
enum { OPT_GD = 0, OPT_ADAGRAD = 1, OPT_RMSPROP = 2, OPT_MOMENTUM = 3 };
template <int KIND>
__global__ __launch_bounds__(256) void optimizer_rows_kernel(float* __restrict__ var,
                                                             float* __restrict__ s0,
                                                             float* __restrict__ s1,
                                                             float* __restrict__ grad,
                                                             uint8_t* __restrict__ flag,
                                                             int64_t n_rows, int d, float lr,
                                                             float h1, float h2, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += stride) {
    if (flag[r] == 0) continue;
    for (int k = lane; k < d; k += 64) {
      const int64_t o = r * d + k;
      const float g = grad[o];
      float v = var[o];
      if (KIND == OPT_GD) {
        v = __fsub_rn(v, __fmul_rn(lr, g));
      } else if (KIND == OPT_ADAGRAD) {
        const float a = __fadd_rn(s0[o], __fmul_rn(g, g));
        s0[o] = a;
        v = __fsub_rn(v, __fmul_rn(__fmul_rn(lr, g), 1.0f / sqrtf(a)));
      } else if (KIND == OPT_RMSPROP) {
        const float ms = __fadd_rn(__fmul_rn(s0[o], h1), __fmul_rn(__fmul_rn(g, g), 1.0f - h1));
        s0[o] = ms;
        const float mom = __fadd_rn(__fmul_rn(s1[o], h2),
                                    __fmul_rn(__fmul_rn(1.0f / sqrtf(__fadd_rn(ms, eps)), lr), g));
        s1[o] = mom;
        v = __fsub_rn(v, mom);
      } else {
        const float a = __fadd_rn(__fmul_rn(s0[o], h1), g);
        s0[o] = a;
        v = __fsub_rn(v, __fmul_rn(a, lr));
      }
      var[o] = v;
      grad[o] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) flag[r] = 0;
  }
}

// The DENSE updates of util/learner.py:2-17 (a variable whose consumers are not all gathers gets TF-1.12's Apply*
// kernels, core/kernels/training_ops.cc [EXT]; NGCF's and Mult-VAE's tables and weights):
//   gd        var -= g * lr
//   adagrad   accum += g * g;                          var -= (g * lr) * rsqrt(accum)
//   rmsprop   ms += (g * g - ms) * (1 - rho);          mom = mom * momentum + (g * lr) / sqrt(eps + ms);  var -= mom
//   momentum  accum = accum * momentum + g;            var -= accum * lr
// (ApplyRMSProp's arithmetic is not SparseApplyRMSProp's above: the moving average is an increment, the step a division.)
template <int KIND>
__global__ __launch_bounds__(256) void optimizer_dense_kernel(float* __restrict__ var, float* __restrict__ s0,
                                                              float* __restrict__ s1, float* __restrict__ grad,
                                                              int64_t n, float lr, float h1, float h2, float eps,
                                                              int clear_grad) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
    const float g = grad[o];
    float v = var[o];
    if (KIND == OPT_GD) {
      v = __fsub_rn(v, __fmul_rn(g, lr));
    } else if (KIND == OPT_ADAGRAD) {
      const float a = __fadd_rn(s0[o], __fmul_rn(g, g));
      s0[o] = a;
      v = __fsub_rn(v, __fmul_rn(__fmul_rn(g, lr), 1.0f / sqrtf(a)));
    } else if (KIND == OPT_RMSPROP) {
      const float ms0 = s0[o];
      const float ms = __fadd_rn(ms0, __fmul_rn(__fsub_rn(__fmul_rn(g, g), ms0), 1.0f - h1));
      s0[o] = ms;
      const float mom = __fadd_rn(__fmul_rn(s1[o], h2), __fdiv_rn(__fmul_rn(g, lr), sqrtf(__fadd_rn(eps, ms))));
      s1[o] = mom;
      v = __fsub_rn(v, mom);
    } else {
      const float a = __fadd_rn(__fmul_rn(s0[o], h1), g);
      s0[o] = a;
      v = __fsub_rn(v, __fmul_rn(a, lr));
    }
    var[o] = v;
    if (clear_grad) grad[o] = 0.f;
  }
}

__global__ __launch_bounds__(256) void mark_rows_kernel(const int32_t* __restrict__ ids, int n,
                                                        int offset, uint8_t* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[(int64_t)ids[i] + offset] = 1;
}

// dst[w][:] = src[rows[w]][:]  /  dst[rows[w]][:] += src[w][:]  (row lookups of a sharded table)
__global__ __launch_bounds__(256) void rows_gather_kernel(const int32_t* __restrict__ rows,
                                                          int n_listed, int d,
                                                          const float* __restrict__ src, int64_t ld_src,
                                                          float* __restrict__ dst, int64_t ld_dst) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n_listed) return;
  const int64_t r = rows[w];
  for (int k = lane; k < d; k += 64) dst[(int64_t)w * ld_dst + k] = src[r * ld_src + k];
}
// the same for two tables with one row list (a row-sharded step looks its rows up in Esum AND E0, and sends two
// gradient rows per occurrence back: one launch per pair)
__global__ __launch_bounds__(256) void rows_gather2_kernel(const int32_t* __restrict__ rows, int n_listed, int d,
                                                           const float* __restrict__ src_a, int64_t ld_a,
                                                           const float* __restrict__ src_b, int64_t ld_b,
                                                           float* __restrict__ dst_a, int64_t ldd_a,
                                                           float* __restrict__ dst_b, int64_t ldd_b) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n_listed) return;
  const int64_t r = rows[w];
  for (int k = lane; k < d; k += 64) {
    const float a = src_a[r * ld_a + k], b = src_b[r * ld_b + k];
    dst_a[(int64_t)w * ldd_a + k] = a;
    dst_b[(int64_t)w * ldd_b + k] = b;
  }
}
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const int32_t* __restrict__ rows,
                                                               int n_listed, int d,
                                                               const float* __restrict__ src,
                                                               int64_t ld_src,
                                                               float* __restrict__ dst) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n_listed) return;
  const int64_t r = rows[w];
  for (int k = lane; k < d; k += 64) atomicAdd(&dst[r * d + k], src[(int64_t)w * ld_src + k]);
}
__global__ __launch_bounds__(256) void rows_clear_kernel(const int32_t* __restrict__ rows,
                                                         int n_listed, int d, float* b0, float* b1,
                                                         float* b2, float* b3,
                                                         uint8_t* __restrict__ flag) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n_listed) return;
  const int64_t r = rows[w];
  for (int k = lane; k < d; k += 64) {
    if (b0) b0[r * d + k] = 0.f;
    if (b1) b1[r * d + k] = 0.f;
    if (b2) b2[r * d + k] = 0.f;
    if (b3) b3[r * d + k] = 0.f;
  }
  if (flag && lane == 0) flag[r] = 0;
}

enum { OP_SCALE = 0, OP_ADD = 1, OP_DIV = 2 };
template <int OP>
__global__ __launch_bounds__(256) void ewise_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ y, float a,
                                                    float* __restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float r;
    if (OP == OP_SCALE) r = __fmul_rn(a, x[i]);
    else if (OP == OP_ADD) r = __fadd_rn(x[i], y[i]);
    else r = x[i] / a;
    out[i] = r;
  }
}

// out[r][c] = x[r][c] + y[r][c] on row-strided views (column blocks of wider matrices)
__global__ __launch_bounds__(256) void add2d_kernel(const float* __restrict__ x, int64_t ldx,
                                                    const float* __restrict__ y, int64_t ldy,
                                                    float* __restrict__ out, int64_t ldo,
                                                    int64_t rows, int cols) {
  const int64_t n = rows * cols;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    out[r * ldo + c] = __fadd_rn(x[r * ldx + c], y[r * ldy + c]);
  }
}
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ x, int64_t ldx,
                                                     float* __restrict__ out, int64_t ldo,
                                                     int64_t rows, int cols) {
  const int64_t n = rows * cols;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    out[r * ldo + c] = x[r * ldx + c];
  }
}

inline unsigned sweep_blocks(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;   // 256 CUs x 16 resident blocks, grid-stride beyond
  if (b < 1) b = 1;
  return (unsigned)b;
}

thread_local char g_err[512] = "";

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Several dense TF-Adam updates in one launch (a model's weight matrices and biases: sixteen
// 5-us launches per NGCF step otherwise).  blockIdx.y = tensor.
constexpr int kMaxMulti = 32;        // NGCF with two layers has 17 trainables: one launch
struct MultiAdam {
  float* var[kMaxMulti]; float* m[kMaxMulti]; float* v[kMaxMulti]; float* grad[kMaxMulti];
  int64_t n[kMaxMulti];
  int clear[kMaxMulti];
};
__global__ __launch_bounds__(256) void adam_multi_kernel(MultiAdam t, float alpha, float omb1, float omb2,
                                                         float eps) {
  const int k = blockIdx.y;
  float* __restrict__ var = t.var[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  float* __restrict__ grad = t.grad[k];
  const int64_t n = t.n[k], n4 = n / 4;
  const bool clear = t.clear[k] != 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 w = reinterpret_cast<float4*>(var)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    nr::adam_dense_tf(g.x, w.x, mm.x, vv.x, alpha, omb1, omb2, eps);
    nr::adam_dense_tf(g.y, w.y, mm.y, vv.y, alpha, omb1, omb2, eps);
    nr::adam_dense_tf(g.z, w.z, mm.z, vv.z, alpha, omb1, omb2, eps);
    nr::adam_dense_tf(g.w, w.w, mm.w, vv.w, alpha, omb1, omb2, eps);
    reinterpret_cast<float4*>(var)[i] = w;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (clear) reinterpret_cast<float4*>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float w = var[i], mm = m[i], vv = v[i];
    nr::adam_dense_tf(grad[i], w, mm, vv, alpha, omb1, omb2, eps);
    var[i] = w; m[i] = mm; v[i] = vv;
    if (clear) grad[i] = 0.f;
  }
}

}  // namespace

// dst[i] = src[index[i]] (row flags of the chunked hop's virtual rows: neurec_amd/sharded.py)
__global__ __launch_bounds__(256) static void gather_u8_kernel(const uint8_t* __restrict__ src,
                                                               const int32_t* __restrict__ index, int64_t n,
                                                               uint8_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[index[i]];
}

extern "C" {

void nrhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* nrhip_last_error(void) { return g_err; }

int nrhip_abi_version(void) { return 4; }

int nrhip_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* name, int name_len) {
  int dev = 0;
  NR_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  NR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (name && name_len > 0) {
    snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return NR_OK;
}

int nrhip_adam_sparse_tf(float* d_var, float* d_m, float* d_v, float* d_grad, int64_t n,
                         float alpha, float beta1, float beta2, float eps, void* stream) {
  NR_REQUIRE(d_var && d_m && d_v && d_grad && n >= 0, NR_ERR_ARG, "adam_sparse_tf: bad arguments");
  NR_REQUIRE(aligned16(d_var) && aligned16(d_m) && aligned16(d_v) && aligned16(d_grad), NR_ERR_ARG,
             "adam_sparse_tf: buffers must be 16-byte aligned");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL((adam_kernel<true, true>), dim3(sweep_blocks(n / 4 + 1)), dim3(256), 0,
                     (hipStream_t)stream, d_var, d_m, d_v, d_grad, n, alpha, beta1, beta2,
                     1.0f - beta1, 1.0f - beta2, eps);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Exact lazy form of nrhip_adam_sparse_tf on a [n_rows][d] table: see adam_lazy_kernel.  d_last
 * (int32 per row, zero at step 0) = last step applied; d_alpha_tab[s] = lr_s of step s (1-based,
 * fp32, made by the caller exactly as the per-step `alpha` it would pass to the sweep), at least
 * t + 1 entries; d_plan / n_occ = the batch's sorted occurrences (rows are table rows), t = this
 * step.  d_plan = NULL, n_occ = 0, period = 1: flush — every row brought to step t. */
int nrhip_adam_sparse_tf_lazy(float* d_var, float* d_m, float* d_v, float* d_grad, int32_t* d_last,
                              const int32_t* d_stamp, int64_t n_rows, int d, const uint64_t* d_plan,
                              int n_occ, const uint64_t* d_next_plan, int n_next_occ,
                              const float* d_alpha_tab, int t, int period, float beta1, float beta2,
                              float eps, void* stream) {
  NR_REQUIRE(d_var && d_m && d_v && d_grad && d_last && d_alpha_tab && (n_occ == 0 || d_stamp), NR_ERR_ARG,
             "adam_sparse_tf_lazy: null pointer argument");
  NR_REQUIRE(n_rows >= 0 && d >= 1 && d <= 256 && n_occ >= 0 && (n_occ == 0 || d_plan) && n_next_occ >= 0 &&
                 (n_next_occ == 0 || d_next_plan) && t >= 0 && period >= 1,
             NR_ERR_ARG, "adam_sparse_tf_lazy: bad sizes");
  const int64_t scheduled = (n_rows + period - 1) / period;
  const int64_t waves = (int64_t)n_occ + n_next_occ + scheduled;
  if (waves == 0) return NR_OK;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NR_LAZY(CPL)                                                                                  \
  hipLaunchKernelGGL(adam_lazy_kernel<CPL>, grid, block, 0, st, d_var, d_m, d_v, d_grad, d_last, d_stamp, d, n_rows, \
                     d_plan, n_occ, d_next_plan, n_next_occ, d_alpha_tab, t, period, beta1, beta2,          \
                     1.0f - beta1, 1.0f - beta2, eps)
  if (d <= 64) NR_LAZY(1); else if (d <= 128) NR_LAZY(2); else NR_LAZY(4);
#undef NR_LAZY
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_adam_dense_tf(float* d_var, float* d_m, float* d_v, float* d_grad, int64_t n,
                        float alpha, float beta1, float beta2, float eps, int clear_grad,
                        void* stream) {
  NR_REQUIRE(d_var && d_m && d_v && d_grad && n >= 0, NR_ERR_ARG, "adam_dense_tf: bad arguments");
  NR_REQUIRE(aligned16(d_var) && aligned16(d_m) && aligned16(d_v) && aligned16(d_grad), NR_ERR_ARG,
             "adam_dense_tf: buffers must be 16-byte aligned");
  if (n == 0) return NR_OK;
  if (clear_grad)
    hipLaunchKernelGGL((adam_kernel<false, true>), dim3(sweep_blocks(n / 4 + 1)), dim3(256), 0,
                       (hipStream_t)stream, d_var, d_m, d_v, d_grad, n, alpha, beta1, beta2,
                       1.0f - beta1, 1.0f - beta2, eps);
  else
    hipLaunchKernelGGL((adam_kernel<false, false>), dim3(sweep_blocks(n / 4 + 1)), dim3(256), 0,
                       (hipStream_t)stream, d_var, d_m, d_v, d_grad, n, alpha, beta1, beta2,
                       1.0f - beta1, 1.0f - beta2, eps);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_adam_dense_tf2(float* d_var, float* d_m, float* d_v, const float* d_grad_a,
                         const float* d_grad_b, int64_t n, float alpha, float beta1, float beta2,
                         float eps, void* stream) {
  NR_REQUIRE(d_var && d_m && d_v && d_grad_a && d_grad_b && n >= 0, NR_ERR_ARG,
             "adam_dense_tf2: bad arguments");
  NR_REQUIRE(aligned16(d_var) && aligned16(d_m) && aligned16(d_v) && aligned16(d_grad_a) &&
                 aligned16(d_grad_b),
             NR_ERR_ARG, "adam_dense_tf2: buffers must be 16-byte aligned");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(adam_dense2_kernel, dim3(sweep_blocks(n / 4 + 1)), dim3(256), 0,
                     (hipStream_t)stream, d_var, d_m, d_v, d_grad_a, d_grad_b, n, alpha,
                     1.0f - beta1, 1.0f - beta2, eps);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Dense TF-1.12 ApplyAdam on up to 32 tensors in one launch (host arrays of device pointers and
 * lengths; clear_grad[k] != 0 zeroes tensor k's gradient as nrhip_adam_dense_tf does). */
int nrhip_adam_dense_tf_multi(int n_tensors, float* const* d_vars, float* const* d_ms,
                              float* const* d_vs, float* const* d_grads, const int64_t* sizes,
                              const int32_t* clear_grad, float alpha, float beta1, float beta2,
                              float eps, void* stream) {
  NR_REQUIRE(n_tensors >= 0 && n_tensors <= kMaxMulti && (n_tensors == 0 || (d_vars && d_ms && d_vs && d_grads && sizes)),
             NR_ERR_ARG, "adam_dense_tf_multi: 0..%d tensors", kMaxMulti);
  if (n_tensors == 0) return NR_OK;
  MultiAdam t{};
  int64_t biggest = 0;
  for (int k = 0; k < n_tensors; ++k) {
    NR_REQUIRE(d_vars[k] && d_ms[k] && d_vs[k] && d_grads[k] && sizes[k] >= 0, NR_ERR_ARG,
               "adam_dense_tf_multi: tensor %d: bad arguments", k);
    NR_REQUIRE(aligned16(d_vars[k]) && aligned16(d_ms[k]) && aligned16(d_vs[k]) && aligned16(d_grads[k]),
               NR_ERR_ARG, "adam_dense_tf_multi: tensor %d: buffers must be 16-byte aligned", k);
    t.var[k] = d_vars[k]; t.m[k] = d_ms[k]; t.v[k] = d_vs[k]; t.grad[k] = d_grads[k];
    t.n[k] = sizes[k];
    t.clear[k] = clear_grad ? clear_grad[k] : 0;
    biggest = std::max<int64_t>(biggest, sizes[k]);
  }
  hipLaunchKernelGGL(adam_multi_kernel, dim3(sweep_blocks(biggest / 4 + 1), n_tensors), dim3(256), 0,
                     (hipStream_t)stream, t, alpha, 1.0f - beta1, 1.0f - beta2, eps);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_rows_div(const int32_t* d_rows, int n_listed, int d, const float* d_src, float denom,
                   float* d_dst, void* stream) {
  NR_REQUIRE(d_rows && d_src && d_dst && n_listed >= 0 && d >= 1, NR_ERR_ARG,
             "rows_div: bad arguments");
  if (n_listed == 0) return NR_OK;
  hipLaunchKernelGGL(rows_div_kernel, dim3((n_listed + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     d_rows, n_listed, d, d_src, denom, d_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_gather_u8(const uint8_t* d_src, const int32_t* d_index, int64_t n, uint8_t* d_dst, void* stream) {
  NR_REQUIRE(d_src && d_index && d_dst && n >= 0, NR_ERR_ARG, "gather_u8: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(gather_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src,
                     d_index, n, d_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_mark_rows(const int32_t* d_ids, int n, int offset, uint8_t* d_flag, void* stream) {
  NR_REQUIRE(d_ids && d_flag && n >= 0, NR_ERR_ARG, "mark_rows: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(mark_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     d_ids, n, offset, d_flag);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_optimizer_rows_tf(int kind, float* d_var, float* d_slot0, float* d_slot1, float* d_grad,
                            uint8_t* d_row_flag, int64_t n_rows, int d, float lr, float hyper1,
                            float hyper2, float eps, void* stream) {
  NR_REQUIRE(d_var && d_grad && d_row_flag && n_rows >= 0 && d >= 1, NR_ERR_ARG,
             "optimizer_rows_tf: bad arguments");
  NR_REQUIRE(kind >= OPT_GD && kind <= OPT_MOMENTUM, NR_ERR_ARG,
             "optimizer_rows_tf: unknown optimiser %d (0 gd, 1 adagrad, 2 rmsprop, 3 momentum)", kind);
  NR_REQUIRE(kind == OPT_GD || d_slot0, NR_ERR_ARG, "optimizer_rows_tf: slot buffer missing");
  NR_REQUIRE(kind != OPT_RMSPROP || d_slot1, NR_ERR_ARG, "optimizer_rows_tf: rmsprop needs two slots");
  if (n_rows == 0) return NR_OK;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NR_OPT(K)                                                                                  \
  hipLaunchKernelGGL(optimizer_rows_kernel<K>, grid, block, 0, st, d_var, d_slot0, d_slot1, d_grad, \
                     d_row_flag, n_rows, d, lr, hyper1, hyper2, eps)
  if (kind == OPT_GD) NR_OPT(OPT_GD);
  else if (kind == OPT_ADAGRAD) NR_OPT(OPT_ADAGRAD);
  else if (kind == OPT_RMSPROP) NR_OPT(OPT_RMSPROP);
  else NR_OPT(OPT_MOMENTUM);
#undef NR_OPT
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_optimizer_dense_tf(int kind, float* d_var, float* d_slot0, float* d_slot1, float* d_grad, int64_t n,
                             float lr, float hyper1, float hyper2, float eps, int clear_grad, void* stream) {
  NR_REQUIRE(d_var && d_grad && n >= 0, NR_ERR_ARG, "optimizer_dense_tf: bad arguments");
  NR_REQUIRE(kind >= OPT_GD && kind <= OPT_MOMENTUM, NR_ERR_ARG,
             "optimizer_dense_tf: unknown optimiser %d (0 gd, 1 adagrad, 2 rmsprop, 3 momentum)", kind);
  NR_REQUIRE(kind == OPT_GD || d_slot0, NR_ERR_ARG, "optimizer_dense_tf: slot buffer missing");
  NR_REQUIRE(kind != OPT_RMSPROP || d_slot1, NR_ERR_ARG, "optimizer_dense_tf: rmsprop needs two slots");
  if (n == 0) return NR_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NR_OPTD(K)                                                                                          \
  hipLaunchKernelGGL(optimizer_dense_kernel<K>, grid, block, 0, st, d_var, d_slot0, d_slot1, d_grad, n, lr, \
                     hyper1, hyper2, eps, clear_grad)
  if (kind == OPT_GD) NR_OPTD(OPT_GD);
  else if (kind == OPT_ADAGRAD) NR_OPTD(OPT_ADAGRAD);
  else if (kind == OPT_RMSPROP) NR_OPTD(OPT_RMSPROP);
  else NR_OPTD(OPT_MOMENTUM);
#undef NR_OPTD
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_rows_gather(const int32_t* d_rows, int n_listed, int d, const float* d_src, float* d_dst,
                      int64_t ld_dst, void* stream) {
  NR_REQUIRE(d_rows && d_src && d_dst && n_listed >= 0 && d >= 1 && ld_dst >= d, NR_ERR_ARG,
             "rows_gather: bad arguments");
  if (n_listed == 0) return NR_OK;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((n_listed + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, d_rows, n_listed, d, d_src, (int64_t)d, d_dst, ld_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* the same with a source row stride: d_src may be a column block of a wider row-major buffer */
int nrhip_rows_gather_ld(const int32_t* d_rows, int n_listed, int d, const float* d_src, int64_t ld_src,
                         float* d_dst, int64_t ld_dst, void* stream) {
  NR_REQUIRE(d_rows && d_src && d_dst && n_listed >= 0 && d >= 1 && ld_dst >= d && ld_src >= d, NR_ERR_ARG,
             "rows_gather_ld: bad arguments");
  if (n_listed == 0) return NR_OK;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((n_listed + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, d_rows, n_listed, d, d_src, ld_src, d_dst, ld_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* d_dst_a[w] = d_src_a[d_rows[w]] and d_dst_b[w] = d_src_b[d_rows[w]] in one launch (row strides as given) */
int nrhip_rows_gather2(const int32_t* d_rows, int n_listed, int d, const float* d_src_a, int64_t ld_a,
                       const float* d_src_b, int64_t ld_b, float* d_dst_a, int64_t ldd_a, float* d_dst_b,
                       int64_t ldd_b, void* stream) {
  NR_REQUIRE(d_rows && d_src_a && d_src_b && d_dst_a && d_dst_b && n_listed >= 0 && d >= 1 && ld_a >= d && ld_b >= d &&
                 ldd_a >= d && ldd_b >= d, NR_ERR_ARG, "rows_gather2: bad arguments");
  if (n_listed == 0) return NR_OK;
  hipLaunchKernelGGL(rows_gather2_kernel, dim3((n_listed + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_rows,
                     n_listed, d, d_src_a, ld_a, d_src_b, ld_b, d_dst_a, ldd_a, d_dst_b, ldd_b);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_rows_scatter_add(const int32_t* d_rows, int n_listed, int d, const float* d_src,
                           int64_t ld_src, float* d_dst, void* stream) {
  NR_REQUIRE(d_rows && d_src && d_dst && n_listed >= 0 && d >= 1 && ld_src >= d, NR_ERR_ARG,
             "rows_scatter_add: bad arguments");
  if (n_listed == 0) return NR_OK;
  hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((n_listed + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, d_rows, n_listed, d, d_src, ld_src, d_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_rows_clear(const int32_t* d_rows, int n_listed, int d, float* d_b0, float* d_b1,
                     float* d_b2, float* d_b3, uint8_t* d_flag, void* stream) {
  NR_REQUIRE(d_rows && n_listed >= 0 && d >= 1, NR_ERR_ARG, "rows_clear: bad arguments");
  if (n_listed == 0) return NR_OK;
  hipLaunchKernelGGL(rows_clear_kernel, dim3((n_listed + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, d_rows, n_listed, d, d_b0, d_b1, d_b2, d_b3, d_flag);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_add2d(const float* d_x, int64_t ldx, const float* d_y, int64_t ldy, float* d_out,
                int64_t ldo, int64_t rows, int cols, void* stream) {
  NR_REQUIRE(d_x && d_y && d_out && rows >= 0 && cols >= 1 && ldx >= cols && ldy >= cols &&
                 ldo >= cols,
             NR_ERR_ARG, "add2d: bad arguments");
  if (rows == 0) return NR_OK;
  hipLaunchKernelGGL(add2d_kernel, dim3(sweep_blocks(rows * cols)), dim3(256), 0,
                     (hipStream_t)stream, d_x, ldx, d_y, ldy, d_out, ldo, rows, cols);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_copy2d(const float* d_x, int64_t ldx, float* d_out, int64_t ldo, int64_t rows, int cols,
                 void* stream) {
  NR_REQUIRE(d_x && d_out && rows >= 0 && cols >= 1 && ldx >= cols && ldo >= cols, NR_ERR_ARG,
             "copy2d: bad arguments");
  if (rows == 0) return NR_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(sweep_blocks(rows * cols)), dim3(256), 0,
                     (hipStream_t)stream, d_x, ldx, d_out, ldo, rows, cols);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_scale(const float* d_x, float a, float* d_y, int64_t n, void* stream) {
  NR_REQUIRE(d_x && d_y && n >= 0, NR_ERR_ARG, "scale: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(ewise_kernel<OP_SCALE>, dim3(sweep_blocks(n)), dim3(256), 0,
                     (hipStream_t)stream, d_x, (const float*)nullptr, a, d_y, n);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_add(const float* d_x, const float* d_y, float* d_out, int64_t n, void* stream) {
  NR_REQUIRE(d_x && d_y && d_out && n >= 0, NR_ERR_ARG, "add: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(ewise_kernel<OP_ADD>, dim3(sweep_blocks(n)), dim3(256), 0,
                     (hipStream_t)stream, d_x, d_y, 0.f, d_out, n);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_div_scalar(const float* d_x, float denom, float* d_y, int64_t n, void* stream) {
  NR_REQUIRE(d_x && d_y && n >= 0, NR_ERR_ARG, "div_scalar: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(ewise_kernel<OP_DIV>, dim3(sweep_blocks(n)), dim3(256), 0,
                     (hipStream_t)stream, d_x, (const float*)nullptr, denom, d_y, n);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
