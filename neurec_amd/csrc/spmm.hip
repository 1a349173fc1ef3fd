// spmm.hip — sparse adjacency × dense embedding propagation  Y = Â · X.
//
// Stands in for tf.sparse_tensor_dense_matmul(adj, ego)
// (model/general_recommender/LightGCN.py:140; NGCF.py:176 and its 100-slab
// split NGCF.py:320-332) and, with the transposed CSR, for its gradient.
//
// Â is CSR (row-major, ascending columns — what the reference hands to TF,
// LightGCN.py:151-154).  X, Y are row-major [N][d] fp32.
//
// Work decomposition (d >= 64): the host cuts the row list into *work items* —
// a run of up to 32 whole rows holding at most 256 non-zeros, or one 256-nnz
// segment of a longer (hub) row — and one wave64 owns one item.  Inside an item
// the wave streams the item's (column, value) pairs 64 at a time with one
// coalesced load (the next chunk is prefetched while the current one is
// consumed), broadcasts them lane by lane with v_readlane, and keeps 8 row
// gathers X[col] (one coalesced 4·d-byte read each, lanes = columns) in flight
// at all times, *across row boundaries*: a row end only retires the accumulator
// into an LDS tile, it never waits on memory.  When the item's non-zeros are
// exhausted the tile is drained through the fused epilogue with batched loads.
// (A first version used one wave per row; at ~23 nnz per row it was bound by
// three dependent memory round trips per row, not by bandwidth.)
//
// Numerics: products and sums are rounded separately, in ascending column
// order within a row — the order of the reference's CPU kernel — so rows of up
// to 256 non-zeros are bit-identical to oracle/.  Longer rows are split; their
// partial sums are combined in segment order by a second tiny kernel
// (deterministic, no atomics).
//
// Fused epilogue (saves a pass over [N][d] per layer):
//     y = Σ_j a_j x_j ; y += addend[r]  (backward: the dE*/(L+1) term)
//     sum_out[r] = sum_in[r] + y          (forward: running layer sum for the
//                                          mean over layers, LightGCN.py:146-147)
//
// Roofline: HBM.  Algorithmic bytes per pass = nnz·8 + (N+1)·4 + 2·N·d·4
// (SURVEY.md §8d); the gathers themselves are served by L2 / Infinity Cache.
#include "nr_common.h"
#include <limits.h>
#include <algorithm>
#include <stdlib.h>
#include <new>
#include <vector>

// spmm_blocked.hip
extern "C" int nrhip_spmm_blocked(const void* plan, const int32_t* d_indices, const float* d_vals,
                                  const float* d_X, float* d_Y, const float* d_addend,
                                  const float* d_sum_in, float* d_sum_out,
                                  const uint8_t* d_x_row_nonzero, const uint8_t* d_y_row_wanted,
                                  void* stream);

extern "C" int nrhip_spmm_blocked_wanted_layers(const void* plan, const int32_t* d_indices,
                                                const float* d_vals, const float* d_X,
                                                const float* d_sum_in, const float* d_layer_a,
                                                const float* d_layer_b, float* d_sum_out,
                                                const uint8_t* d_y_row_wanted, void* stream);
extern "C" int nrhip_spmm_blocked_wanted_batch(const void* plan, const int32_t* d_indices,
                                               const float* d_vals, const float* d_X,
                                               const float* d_sum_in, const float* d_layer_a,
                                               const float* d_layer_b, float* d_sum_out,
                                               const int32_t* d_users, const int32_t* d_pos,
                                               const int32_t* d_neg, int batch, int n_users,
                                               uint8_t* d_row_flag, int32_t* d_rows_out, void* stream);
extern "C" int nrhip_spmm_blocked_has_wanted(const void* plan);
extern "C" int nrhip_spmm_blocked_adam(const void* plan, const int32_t* d_indices,
                                       const float* d_vals, const float* d_X, float* d_addend,
                                       float* d_grad_b, float* d_var, float* d_m, float* d_v,
                                       float alpha, float beta1, float beta2, float eps,
                                       int clear_consumed, uint8_t* d_row_flag, void* stream);

namespace {

constexpr int kSegLen = 256;     // non-zeros per segment of a split (hub) row
constexpr int kItemRows = 32;    // most whole rows a work item may hold (LDS tile height)
constexpr int kLegacyWaves = 4;
constexpr int kGather = 16;      // row gathers in flight per wave (8/16/32 measured equal)

struct SpmmPlan {
  int64_t n_rows, nnz;
  const void* blocked[5];    // optional lane-group schedules for d = 16 / 32 / 64 / 128 / 256 (spmm_blocked.hip); not owned
  // --- work items (d >= 64 path)
  int64_t n_items, n_hub_items, n_a_items, n_b_items;   // items = [hub | class A | class B]
  int item_rows, item_nnz;   // limits the items were cut with
  int32_t* item_row0;   // first row of the run (whole-row items) or the row (segment items)
  int32_t* item_nrows;  // rows in the run (1..item_rows); 0 marks a segment of a split row
  int32_t* item_slot;   // segment items: index into the partial buffer; else -1
  int64_t* item_begin;  // first non-zero
  int32_t* item_len;    // number of non-zeros
  // --- per-row segments (d < 64 path)
  int64_t n_seg;
  int32_t* seg_row;
  int64_t* seg_begin;
  int32_t* seg_len;
  int32_t* seg_slot;
  // --- split rows
  int64_t n_multi_seg, n_multi_row;
  int32_t* multi_row;
  int32_t* multi_first;
  int32_t* multi_nseg;
};

inline int blocked_slot(int d) {
  return d == 16 ? 0 : d == 32 ? 1 : d == 64 ? 2 : d == 128 ? 3 : d == 256 ? 4 : -1;
}

struct PlanSizes {
  size_t max_seg, max_multi, max_item;
};
PlanSizes plan_sizes(int64_t n_rows, int64_t nnz) {
  PlanSizes s;
  s.max_seg = (size_t)(n_rows + nnz / kSegLen + 1);
  s.max_multi = (size_t)(nnz / kSegLen + 1);
  s.max_item = (size_t)(n_rows + nnz / kSegLen + 1);
  return s;
}
size_t plan_bytes_for(int64_t n_rows, int64_t nnz) {
  const PlanSizes s = plan_sizes(n_rows, nnz);
  return nr_align_up(s.max_seg * 4, 256) * 3 + nr_align_up(s.max_seg * 8, 256) +
         nr_align_up(s.max_multi * 4, 256) * 3 + nr_align_up(s.max_item * 4, 256) * 4 +
         nr_align_up(s.max_item * 8, 256);
}

// Add Σ a_j·X[col_j] over the contiguous non-zeros [b, b+len) to acc, in order.  G row
// gathers are issued back to back (indices clamped, so every load is unconditional) and
// then consumed in order: memory-level parallelism comes from the batch, the summation
// order stays the sequential one.  MASKED: terms whose X row is flagged all-zero are
// skipped instead of gathered.
template <int D, int G, bool MASKED>
__device__ __forceinline__ void gather_accumulate(int64_t b, int len, int lane,
                                                  const int32_t* __restrict__ indices,
                                                  const float* __restrict__ vals,
                                                  const float* __restrict__ X,
                                                  const uint8_t* __restrict__ col_mask,
                                                  float (&acc)[D / NR_WAVE]) {
  constexpr int CPL = D / NR_WAVE;
  for (int k0 = 0; k0 < len; k0 += NR_WAVE) {
    const int n = min(NR_WAVE, len - k0);
    int my_idx = 0;
    float my_val = 0.f;
    if (lane < n) {
      my_idx = indices[b + k0 + lane];
      my_val = vals[b + k0 + lane];
    }
    if constexpr (!MASKED) {                    // dense walk: positions t0..t0+G-1
      for (int t0 = 0; t0 < n; t0 += G) {
        float a[G];
        float x[G][CPL];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const int tt = min(t0 + u, n - 1);
          const int col = __builtin_amdgcn_readlane(my_idx, tt);
          a[u] = __builtin_bit_cast(
              float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), tt));
          const float* xr = X + (int64_t)col * D + lane;
#pragma unroll
          for (int c = 0; c < CPL; ++c) x[u][c] = xr[c * NR_WAVE];
        }
#pragma unroll
        for (int u = 0; u < G; ++u)
          if (t0 + u < n) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(a[u], x[u][c]));
          }
      }
      continue;
    }
    uint64_t km = ~0ull >> (NR_WAVE - n);
    if (col_mask) km = __ballot(lane < n && col_mask[my_idx] != 0);
    while (km) {
      bool live[G];
      float a[G];
      float x[G][CPL];
      int pos = 0;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        live[u] = km != 0;
        if (live[u]) { pos = __builtin_ctzll(km); km &= km - 1; }
        const int col = __builtin_amdgcn_readlane(my_idx, pos);
        a[u] = __builtin_bit_cast(float,
                                  __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), pos));
        const float* xr = X + (int64_t)col * D + lane;
#pragma unroll
        for (int c = 0; c < CPL; ++c) x[u][c] = xr[c * NR_WAVE];
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (live[u]) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(a[u], x[u][c]));
        }
    }
  }
}

// One split (hub) row handled by a whole block: its waves take consecutive 256-nnz
// segments, the segment partials are added in segment order through LDS (deterministic,
// no atomics, no second kernel), then the fused epilogue runs once.
template <int D, int WPB, int G, bool MASKED>
__device__ __forceinline__ void hub_row_block(int row, float (*part)[D],
                                              const int64_t* __restrict__ indptr,
                                              const int32_t* __restrict__ indices,
                                              const float* __restrict__ vals,
                                              const float* __restrict__ X, float* __restrict__ Y,
                                              const float* __restrict__ addend,
                                              const float* sum_in, float* sum_out,
                                              const uint8_t* __restrict__ col_mask) {
  constexpr int CPL = D / NR_WAVE;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int64_t rb64 = indptr[row], re64 = indptr[row + 1];
  const int64_t rb = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rb64 >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(rb64 & 0xffffffff));
  const int rlen = __builtin_amdgcn_readfirstlane((int)(re64 - rb64));
  const int nseg = rlen > 0 ? (rlen + kSegLen - 1) / kSegLen : 1;
  float total[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) total[c] = 0.f;
  for (int s0 = 0; s0 < nseg; s0 += WPB) {
    const int seg = s0 + wave;
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    if (seg < nseg)
      gather_accumulate<D, G, MASKED>(rb + (int64_t)seg * kSegLen,
                                      min(kSegLen, rlen - seg * kSegLen), lane, indices, vals, X,
                                      col_mask, acc);
#pragma unroll
    for (int c = 0; c < CPL; ++c) part[wave][lane + c * NR_WAVE] = acc[c];
    __syncthreads();
    if (wave == 0) {
      const int cnt = min(WPB, nseg - s0);
      for (int w = 0; w < cnt; ++w) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float pv = part[w][lane + c * NR_WAVE];
          total[c] = (s0 == 0 && w == 0) ? pv : __fadd_rn(total[c], pv);
        }
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int64_t o = (int64_t)row * D + lane + c * NR_WAVE;
      float y = total[c];
      if (addend) y = __fadd_rn(y, addend[o]);
      if (Y) Y[o] = y;
      if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
    }
  }
}

// ----------------------------------------------------------------------------
// d >= 64: one wave per work item (a run of whole rows, or one segment of a split row
// whose partial sum goes to `partial` and is combined by spmm_fix_kernel).
// MASKED enables the two work-skipping filters:
//   col_mask[c] == 0  promises X[c][:] == 0            -> term skipped
//   row_mask[r] == 0  says output row r is not needed   -> row left untouched
// ----------------------------------------------------------------------------
// CARRY: a row's accumulator starts from carry[row] instead of 0 — the chunked (pipelined all-gather) hop of the
// row-sharded engine walks a row's non-zeros in several launches, one per operand chunk, and the sum must stay
// ONE ascending chain ((carry + a1·x1) + a2·x2 ...).  The carried rows are requested together with the item's first
// indices (same round trip), parked in the LDS tile and picked up at every row boundary.
// CHUNK: 0 = the one-launch hop; 1 = first chunk of a chunked hop; 2 = a later chunk (CARRY).  Chunk launches read
// the operand from TWO buffers: columns < x_split index X (this rank's own block), the others X2 (the received chunk).
template <int D, int WPB, int TR, int G, bool MASKED, int CHUNK = 0>
__global__ __launch_bounds__(WPB* NR_WAVE) void spmm_item_kernel(
    const int32_t* __restrict__ item_row0, const int32_t* __restrict__ item_nrows,
    const int64_t* __restrict__ item_begin, const int32_t* __restrict__ item_len,
    const int32_t* __restrict__ item_slot, float* __restrict__ partial, int n_hub, int n_a,
    int n_b, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ addend,
    const float* sum_in, float* sum_out, const uint8_t* __restrict__ col_mask,
    const uint8_t* __restrict__ row_mask, const float* carry = nullptr, const float* X2 = nullptr,
    int x_split = 0) {
  constexpr int CPL = D / NR_WAVE;
  constexpr bool CARRY = CHUNK == 2;
  __shared__ float s_tile[WPB][TR][D];
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  // Item order is [hub segments | class A | class B].  Classes are a locality hint: in a
  // bipartite graph user rows gather only item rows and vice versa, so running class A on
  // XCDs 0-3 and class B on XCDs 4-7 (block b lands on XCD b % 8 — observed, used for speed
  // only) halves the table each XCD's private L2 has to hold.
  const int hub_blocks = ((n_hub + WPB - 1) / WPB + 7) / 8 * 8;
  int64_t item;
  if ((int)blockIdx.x < hub_blocks) {
    item = (int64_t)blockIdx.x * WPB + wave;
    if (item >= n_hub) return;
  } else {
    const int bb = blockIdx.x - hub_blocks;
    if (n_b == 0) {
      item = (int64_t)bb * WPB + wave;
      if (item >= n_a) return;
      item += n_hub;
    } else {
      const int xcd = bb & 7, j = bb >> 3;
      const int64_t local = ((int64_t)j * 4 + (xcd & 3)) * WPB + wave;
      if (xcd < 4) {
        if (local >= n_a) return;
        item = n_hub + local;
      } else {
        if (local >= n_b) return;
        item = (int64_t)n_hub + n_a + local;
      }
    }
  }
  // wave-uniform descriptor -> SGPRs (threadIdx-derived values look divergent to hipcc)
  const int r0 = __builtin_amdgcn_readfirstlane(item_row0[item]);
  const int nr = __builtin_amdgcn_readfirstlane(item_nrows[item]);
  const int len = __builtin_amdgcn_readfirstlane(item_len[item]);
  const int64_t b64 = item_begin[item];
  const int64_t b = ((int64_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 & 0xffffffff));

  if (nr == 0) {                                // one 256-nnz segment of a split (hub) row
    if (MASKED && row_mask && row_mask[r0] == 0) return;
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    gather_accumulate<D, G, MASKED>(b, len, lane, indices, vals, X, MASKED ? col_mask : nullptr,
                                    acc);
    const int slot = __builtin_amdgcn_readfirstlane(item_slot[item]);
#pragma unroll
    for (int c = 0; c < CPL; ++c) partial[(int64_t)slot * D + lane + c * NR_WAVE] = acc[c];
    return;
  }
  uint64_t rows_wanted = ~0ull;                 // bit r: row r0+r is to be produced
  if constexpr (MASKED) {
    if (row_mask) {
      rows_wanted = __ballot(lane < nr && row_mask[(int64_t)r0 + lane] != 0);
      if (rows_wanted == 0) return;             // nothing in this item is needed
    }
  }
  // row ends relative to the item's first non-zero, one per lane
  int my_end = INT_MAX;
  if (lane < nr) my_end = (int)(indptr[(int64_t)r0 + lane + 1] - b);
  int cur = 0;
  int cur_end = __builtin_amdgcn_readlane(my_end, 0);

  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
  float (*tile)[D] = s_tile[wave];

  int nxt_idx = 0;
  float nxt_val = 0.f;
  if (lane < min(NR_WAVE, len)) {
    nxt_idx = indices[b + lane];
    nxt_val = vals[b + lane];
  }
  if constexpr (CARRY) {                        // behind the index loads: the same memory round trip
    for (int i0 = 0; i0 < nr; i0 += 8) {
      float cv[8][CPL];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = (int64_t)r0 + min(i0 + i, nr - 1);
#pragma unroll
        for (int c = 0; c < CPL; ++c) cv[i][c] = carry[row * D + lane + c * NR_WAVE];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i0 + i < nr) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) tile[i0 + i][lane + c * NR_WAVE] = cv[i][c];
        }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = tile[0][lane + c * NR_WAVE];   // a lane reads what it wrote itself
  }
  for (int k0 = 0; k0 < len; k0 += NR_WAVE) {
    const int n = min(NR_WAVE, len - k0);
    const int my_idx = nxt_idx;
    const float my_val = nxt_val;
    if (k0 + NR_WAVE < len && lane < min(NR_WAVE, len - k0 - NR_WAVE)) {   // prefetch next chunk
      nxt_idx = indices[b + k0 + NR_WAVE + lane];
      nxt_val = vals[b + k0 + NR_WAVE + lane];
    }
    if constexpr (!MASKED) {                    // dense walk: positions t0..t0+G-1
      for (int t0 = 0; t0 < n; t0 += G) {
        float a[G];
        float x[G][CPL];
#pragma unroll
        for (int u = 0; u < G; ++u) {           // all gathers unconditional (index clamped)
          const int tt = min(t0 + u, n - 1);
          const int col = __builtin_amdgcn_readlane(my_idx, tt);
          a[u] = __builtin_bit_cast(
              float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), tt));
          const float* xr = ((CHUNK != 0 && col >= x_split) ? X2 + (int64_t)(col - x_split) * D
                                                                : X + (int64_t)col * D) + lane;
#pragma unroll
          for (int c = 0; c < CPL; ++c) x[u][c] = xr[c * NR_WAVE];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (t0 + u < n) {
            const int t = k0 + t0 + u;
            while (t == cur_end) {              // row boundary (possibly several empty rows)
#pragma unroll
              for (int c = 0; c < CPL; ++c) {
                tile[cur][lane + c * NR_WAVE] = acc[c];
                acc[c] = (CARRY && cur + 1 < nr) ? tile[cur + 1][lane + c * NR_WAVE] : 0.f;
              }
              ++cur;
              cur_end = cur < nr ? __builtin_amdgcn_readlane(my_end, cur) : INT_MAX;
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(a[u], x[u][c]));
          }
        }
      }
      continue;
    }
    bool keep = lane < n;
    if (col_mask) keep = keep && col_mask[my_idx] != 0;
    if (row_mask) {                             // does this lane's non-zero sit in a wanted row?
      const int t = k0 + lane;
      bool in_wanted = false;
      int beg = 0;
      for (int r = 0; r < nr; ++r) {
        const int end = __builtin_amdgcn_readlane(my_end, r);
        if ((rows_wanted >> r) & 1) in_wanted = in_wanted || (t >= beg && t < end);
        beg = end;
      }
      keep = keep && in_wanted;
    }
    uint64_t km = __ballot(keep);
    while (km) {
      bool live[G];
      int post[G];
      float a[G];
      float x[G][CPL];
      int pos = 0;
#pragma unroll
      for (int u = 0; u < G; ++u) {             // all gathers unconditional (position clamped)
        live[u] = km != 0;
        if (live[u]) { pos = __builtin_ctzll(km); km &= km - 1; }
        post[u] = pos;
        const int col = __builtin_amdgcn_readlane(my_idx, pos);
        a[u] = __builtin_bit_cast(float,
                                  __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), pos));
        const float* xr = ((CHUNK != 0 && col >= x_split) ? X2 + (int64_t)(col - x_split) * D
                                                                : X + (int64_t)col * D) + lane;
#pragma unroll
        for (int c = 0; c < CPL; ++c) x[u][c] = xr[c * NR_WAVE];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (live[u]) {
          const int t = k0 + post[u];
          while (t >= cur_end) {                // row boundary (possibly several empty rows)
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              tile[cur][lane + c * NR_WAVE] = acc[c];
              acc[c] = (CARRY && cur + 1 < nr) ? tile[cur + 1][lane + c * NR_WAVE] : 0.f;
            }
            ++cur;
            cur_end = cur < nr ? __builtin_amdgcn_readlane(my_end, cur) : INT_MAX;
          }
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(a[u], x[u][c]));
        }
      }
    }
  }
  while (cur < nr) {                            // last row + trailing empty rows
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      tile[cur][lane + c * NR_WAVE] = acc[c];
      acc[c] = (CARRY && cur + 1 < nr) ? tile[cur + 1][lane + c * NR_WAVE] : 0.f;
    }
    ++cur;
  }
  // Drain the tile (each lane reads back only what it wrote itself: no LDS hazard).  The
  // epilogue operands of 4 rows are loaded together before any is used.
  constexpr int kDrain = 4;
  for (int rr = 0; rr < nr; rr += kDrain) {
    float ad[kDrain][CPL], si[kDrain][CPL];
#pragma unroll
    for (int i = 0; i < kDrain; ++i) {
      const int64_t row = (int64_t)r0 + min(rr + i, nr - 1);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int64_t o = row * D + lane + c * NR_WAVE;
        ad[i][c] = addend ? addend[o] : 0.f;
        si[i][c] = sum_out ? sum_in[o] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < kDrain; ++i) {
      if (rr + i < nr && ((rows_wanted >> (rr + i)) & 1)) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int64_t o = ((int64_t)r0 + rr + i) * D + lane + c * NR_WAVE;
          float y = tile[rr + i][lane + c * NR_WAVE];
          if (addend) y = __fadd_rn(y, ad[i][c]);
          if (Y) Y[o] = y;
          if (sum_out) sum_out[o] = __fadd_rn(si[i][c], y);
        }
      }
    }
  }
}

// Listed rows only, one block per listed row (repeats allowed).  Same association as the
// full kernel: 256-nnz segments added in order.
template <int D>
__global__ __launch_bounds__(4 * NR_WAVE) void spmm_rows_kernel(
    const int32_t* __restrict__ rows_list, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ addend,
    const float* __restrict__ sum_in, float* __restrict__ sum_out) {
  __shared__ float s_part[4][D];
  const int row = __builtin_amdgcn_readfirstlane(rows_list[blockIdx.x]);
  hub_row_block<D, 4, kGather, false>(row, s_part, indptr, indices, vals, X, Y, addend, sum_in,
                                      sum_out, nullptr);
}

// combine the partial sums of split rows, in segment order (LPR = lanes per row)
template <int D>
__global__ __launch_bounds__(kLegacyWaves* NR_WAVE) void spmm_fix_kernel(
    const int32_t* __restrict__ multi_row, const int32_t* __restrict__ multi_first,
    const int32_t* __restrict__ multi_nseg, int64_t n_multi, const float* __restrict__ partial,
    float* __restrict__ Y, const float* __restrict__ addend, const float* sum_in,
    float* sum_out, const uint8_t* __restrict__ row_mask) {
  constexpr int LPR = D < NR_WAVE ? D : NR_WAVE;
  constexpr int CPL = D / LPR;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int64_t m = (int64_t)blockIdx.x * kLegacyWaves + wave;
  if (m >= n_multi || lane >= LPR) return;
  const int row = multi_row[m], first = multi_first[m], nseg = multi_nseg[m];
  if (row_mask && row_mask[row] == 0) return;
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = partial[(int64_t)first * D + lane + c * LPR];
  for (int s = 1; s < nseg; ++s) {
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      acc[c] = __fadd_rn(acc[c], partial[(int64_t)(first + s) * D + lane + c * LPR]);
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int64_t o = (int64_t)row * D + lane + c * LPR;
    float y = acc[c];
    if (addend) y = __fadd_rn(y, addend[o]);
    if (Y) Y[o] = y;
    if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
  }
}

// ----------------------------------------------------------------------------
// d < 64 (NGCF's 16-wide layers): 64/d row segments per wave, one lane group each
// ----------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kLegacyWaves* NR_WAVE) void spmm_seg_kernel(
    const int32_t* __restrict__ seg_row, const int64_t* __restrict__ seg_begin,
    const int32_t* __restrict__ seg_len, const int32_t* __restrict__ seg_slot, int64_t n_seg,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ addend,
    const float* sum_in, float* sum_out, float* __restrict__ partial) {
  constexpr int RPW = NR_WAVE / D;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int64_t wave_id = (int64_t)blockIdx.x * kLegacyWaves + wave;
  const int g = lane / D, col = lane % D;
  const int64_t seg = wave_id * RPW + g;
  const bool live = seg < n_seg;
  const int row = live ? seg_row[seg] : 0;
  const int len = live ? seg_len[seg] : 0;
  const int slot = live ? seg_slot[seg] : -1;
  const int64_t b = live ? seg_begin[seg] : 0;
  int max_len = len;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_len = max(max_len, __shfl_xor(max_len, m, NR_WAVE));
  float acc = 0.f;
  for (int t0 = 0; t0 < max_len; t0 += 4) {
    float a[4], x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = min(t0 + u, len - 1);
      a[u] = 0.f; x[u] = 0.f;
      if (tt >= 0) {
        a[u] = vals[b + tt];
        x[u] = X[(int64_t)indices[b + tt] * D + col];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (t0 + u < len) acc = __fadd_rn(acc, __fmul_rn(a[u], x[u]));
  }
  if (!live) return;
  if (slot < 0) {
    const int64_t o = (int64_t)row * D + col;
    float y = acc;
    if (addend) y = __fadd_rn(y, addend[o]);
    if (Y) Y[o] = y;
    if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
  } else {
    partial[(int64_t)slot * D + col] = acc;
  }
}

template <int D>
int launch_fix(const SpmmPlan* p, float* Y, const float* addend, const float* sum_in,
               float* sum_out, const float* partial, const uint8_t* row_mask, hipStream_t st) {
  if (p->n_multi_row > 0) {
    const int64_t fb = (p->n_multi_row + kLegacyWaves - 1) / kLegacyWaves;
    hipLaunchKernelGGL(spmm_fix_kernel<D>, dim3((unsigned)fb), dim3(kLegacyWaves * NR_WAVE), 0, st,
                       p->multi_row, p->multi_first, p->multi_nseg, p->n_multi_row, partial, Y,
                       addend, sum_in, sum_out, row_mask);
    NR_LAUNCH_CHECK();
  }
  return NR_OK;
}

template <int D, int WPB>
int launch_items(const SpmmPlan* p, const int64_t* indptr, const int32_t* indices,
                 const float* vals, const float* X, float* Y, const float* addend,
                 const float* sum_in, float* sum_out, float* partial, const uint8_t* col_mask,
                 const uint8_t* row_mask, hipStream_t st) {
  const int64_t hub_blocks = ((p->n_hub_items + WPB - 1) / WPB + 7) / 8 * 8;
  const int64_t ba = (p->n_a_items + WPB - 1) / WPB, bb = (p->n_b_items + WPB - 1) / WPB;
  const int64_t blocks = hub_blocks + (p->n_b_items == 0 ? ba : 8 * ((std::max(ba, bb) + 3) / 4));
  if (p->n_items > 0) {
    dim3 grid((unsigned)blocks), block(WPB * NR_WAVE);
#define NR_ITEM_LAUNCH(TR, M)                                                                  \
  hipLaunchKernelGGL((spmm_item_kernel<D, WPB, TR, kGather, M>), grid, block, 0, st,            \
                     p->item_row0, p->item_nrows, p->item_begin, p->item_len, p->item_slot,     \
                     partial, (int)p->n_hub_items, (int)p->n_a_items, (int)p->n_b_items, indptr,   \
                     indices, vals, X, Y, addend, sum_in, sum_out,                               \
                     col_mask, row_mask)
    if (col_mask || row_mask) {
      if (p->item_rows <= 16) NR_ITEM_LAUNCH(16, true); else NR_ITEM_LAUNCH(32, true);
    } else {
      if (p->item_rows <= 16) NR_ITEM_LAUNCH(16, false); else NR_ITEM_LAUNCH(32, false);
    }
#undef NR_ITEM_LAUNCH
    NR_LAUNCH_CHECK();
  }
  return launch_fix<D>(p, Y, addend, sum_in, sum_out, partial, row_mask, st);
}

template <int D>
int launch_segs(const SpmmPlan* p, const int32_t* indices, const float* vals, const float* X,
                float* Y, const float* addend, const float* sum_in, float* sum_out, float* partial,
                hipStream_t st) {
  constexpr int RPW = NR_WAVE / D;
  const int64_t waves = (p->n_seg + RPW - 1) / RPW;
  const int64_t blocks = (waves + kLegacyWaves - 1) / kLegacyWaves;
  if (blocks > 0) {
    hipLaunchKernelGGL(spmm_seg_kernel<D>, dim3((unsigned)blocks), dim3(kLegacyWaves * NR_WAVE), 0,
                       st, p->seg_row, p->seg_begin, p->seg_len, p->seg_slot, p->n_seg, indices,
                       vals, X, Y, addend, sum_in, sum_out, partial);
    NR_LAUNCH_CHECK();
  }
  return launch_fix<D>(p, Y, addend, sum_in, sum_out, partial, nullptr, st);
}


// ----------------------------------------------------------------------------
// Chunked hop, last act: the virtual rows (a real row's 256-non-zero segments, each summed as its own row over
// all chunk launches) combined in segment order — the association of spmm_fix_kernel — and the fused epilogue.
// ----------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kLegacyWaves* NR_WAVE) void spmm_chunks_finish_kernel(
    const int32_t* __restrict__ first_vrow, int64_t n_rows, const float* __restrict__ Yv, float* __restrict__ Y,
    const float* __restrict__ addend, const float* sum_in, float* sum_out, const uint8_t* __restrict__ row_mask) {
  constexpr int CPL = D / NR_WAVE;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int64_t row = (int64_t)blockIdx.x * kLegacyWaves + wave;
  if (row >= n_rows) return;
  if (row_mask && row_mask[row] == 0) return;
  const int v0 = first_vrow[row], v1 = first_vrow[row + 1];
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = Yv[(int64_t)v0 * D + lane + c * NR_WAVE];
  for (int v = v0 + 1; v < v1; ++v) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = __fadd_rn(acc[c], Yv[(int64_t)v * D + lane + c * NR_WAVE]);
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int64_t o = row * D + lane + c * NR_WAVE;
    float y = acc[c];
    if (addend) y = __fadd_rn(y, addend[o]);
    if (Y) Y[o] = y;
    if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
  }
}

// Reduced-exchange hop of the row-sharded engine (neurec_amd/sharded.py, hop="reduce"): every rank multiplied ITS
// user rows into partial item rows; the owner of an item block has received the W partials of its rows
// (parts[q][row][*], rank-major) and adds them in RANK ORDER — one fixed association, the same on every run —
// then the fused epilogue of nrhip_spmm_csr.  Streams: W + 2..4 reads and 1..2 writes of [n_rows][d] fp32; float4 lanes.
__global__ __launch_bounds__(256) void partials_sum_rows_kernel(
    const float4* __restrict__ parts, int world, int64_t n_rows, int d4, float4* __restrict__ Y,
    const float4* __restrict__ addend, const float4* sum_in, float4* sum_out, const uint8_t* __restrict__ row_mask) {
  const int64_t n4 = n_rows * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    if (row_mask && row_mask[i / d4] == 0) continue;
    float4 acc = parts[i];
    for (int q = 1; q < world; ++q) {
      const float4 t = parts[(int64_t)q * n4 + i];
      acc.x = __fadd_rn(acc.x, t.x); acc.y = __fadd_rn(acc.y, t.y);
      acc.z = __fadd_rn(acc.z, t.z); acc.w = __fadd_rn(acc.w, t.w);
    }
    if (addend) {
      const float4 a = addend[i];
      acc.x = __fadd_rn(acc.x, a.x); acc.y = __fadd_rn(acc.y, a.y);
      acc.z = __fadd_rn(acc.z, a.z); acc.w = __fadd_rn(acc.w, a.w);
    }
    if (Y) Y[i] = acc;
    if (sum_out) {
      const float4 s = sum_in[i];
      sum_out[i] = make_float4(__fadd_rn(s.x, acc.x), __fadd_rn(s.y, acc.y), __fadd_rn(s.z, acc.z),
                               __fadd_rn(s.w, acc.w));
    }
  }
}

template <int D, int WPB>
int launch_items_carry(const SpmmPlan* p, const int64_t* indptr, const int32_t* indices, const float* vals,
                       const float* X, const float* X2, int x_split, float* Yv, bool has_carry,
                       const uint8_t* row_mask, hipStream_t st) {
  const int64_t ba = (p->n_a_items + WPB - 1) / WPB, bb = (p->n_b_items + WPB - 1) / WPB;
  const int64_t blocks = p->n_b_items == 0 ? ba : 8 * ((std::max(ba, bb) + 3) / 4);
  if (p->n_items == 0) return NR_OK;
  dim3 grid((unsigned)blocks), block(WPB * NR_WAVE);
#define NR_CARRY_LAUNCH(TR, M, CY)                                                                            \
  hipLaunchKernelGGL((spmm_item_kernel<D, WPB, TR, kGather, M, CY>), grid, block, 0, st, p->item_row0,        \
                     p->item_nrows, p->item_begin, p->item_len, p->item_slot, (float*)nullptr, 0,             \
                     (int)p->n_a_items, (int)p->n_b_items, indptr, indices, vals, X, Yv, (const float*)nullptr, \
                     (const float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr, row_mask,               \
                     (const float*)Yv, X2, x_split)
  const bool small = p->item_rows <= 16;
  if (row_mask) {
    if (has_carry) { if (small) NR_CARRY_LAUNCH(16, true, 2); else NR_CARRY_LAUNCH(32, true, 2); }
    else { if (small) NR_CARRY_LAUNCH(16, true, 1); else NR_CARRY_LAUNCH(32, true, 1); }
  } else {
    if (has_carry) { if (small) NR_CARRY_LAUNCH(16, false, 2); else NR_CARRY_LAUNCH(32, false, 2); }
    else { if (small) NR_CARRY_LAUNCH(16, false, 1); else NR_CARRY_LAUNCH(32, false, 1); }
  }
#undef NR_CARRY_LAUNCH
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // namespace

extern "C" {

int nrhip_spmm_plan_bytes(int64_t n_rows, int64_t nnz, size_t* bytes) {
  NR_REQUIRE(bytes && n_rows >= 0 && nnz >= 0, NR_ERR_ARG, "spmm_plan_bytes: bad arguments");
  *bytes = plan_bytes_for(n_rows, nnz);
  return NR_OK;
}

int nrhip_spmm_plan_create(const int64_t* h_indptr, int64_t n_rows, int item_rows, int item_nnz,
                           int64_t split_row, void* d_plan_buf, size_t plan_bytes, void* stream,
                           void** plan_out) {
  if (split_row <= 0 || split_row >= n_rows) split_row = 0;
  if (item_rows <= 0) item_rows = 16;       // defaults tuned on the gowalla-shaped graph
  if (item_nnz <= 0) item_nnz = 256;
  if (item_rows > kItemRows) item_rows = kItemRows;
  if (item_nnz > kSegLen) item_nnz = kSegLen;
  NR_REQUIRE(h_indptr && d_plan_buf && plan_out && n_rows >= 0, NR_ERR_ARG,
             "spmm_plan_create: bad arguments");
  NR_REQUIRE(n_rows < (int64_t)0x7fffffff, NR_ERR_UNSUPPORTED, "spmm: more than 2^31 rows");
  const int64_t nnz = h_indptr[n_rows] - h_indptr[0];
  NR_REQUIRE(plan_bytes >= plan_bytes_for(n_rows, nnz), NR_ERR_WORKSPACE,
             "spmm_plan_create: plan buffer %zu < %zu bytes", plan_bytes,
             plan_bytes_for(n_rows, nnz));
  std::vector<int32_t> seg_row, seg_len, seg_slot, multi_row, multi_first, multi_nseg;
  std::vector<int64_t> seg_begin;
  std::vector<int32_t> it_row0, it_nrows, it_len, it_slot;
  std::vector<int64_t> it_begin;
  // split (hub) rows first: their segments are the longest-running work
  int32_t slot = 0;
  for (int64_t r = 0; r < n_rows; ++r) {
    const int64_t b = h_indptr[r], e = h_indptr[r + 1];
    NR_REQUIRE(e >= b, NR_ERR_ARG, "spmm_plan_create: indptr not monotone at row %lld",
               (long long)r);
    const int64_t len = e - b;
    if (len > kSegLen) {
      const int ns = (int)((len + kSegLen - 1) / kSegLen);
      multi_row.push_back((int32_t)r);
      multi_first.push_back(slot);
      multi_nseg.push_back(ns);
      for (int s = 0; s < ns; ++s) {
        const int64_t sb = b + (int64_t)s * kSegLen;
        const int32_t sl = (int32_t)std::min<int64_t>(kSegLen, len - (int64_t)s * kSegLen);
        seg_row.push_back((int32_t)r); seg_begin.push_back(sb); seg_len.push_back(sl);
        seg_slot.push_back(slot);
        it_row0.push_back((int32_t)r); it_nrows.push_back(0); it_begin.push_back(sb);
        it_len.push_back(sl); it_slot.push_back(slot);
        ++slot;
      }
    }
  }
  const int64_t n_multi_seg = (int64_t)seg_row.size();
  for (int64_t r = 0; r < n_rows; ++r) {
    const int64_t b = h_indptr[r], len = h_indptr[r + 1] - b;
    if (len <= kSegLen) {
      seg_row.push_back((int32_t)r); seg_begin.push_back(b); seg_len.push_back((int32_t)len);
      seg_slot.push_back(-1);
    }
  }
  const int64_t n_hub_items = (int64_t)it_row0.size();
  int64_t n_a_items = 0;
  // whole-row work items: greedy runs of consecutive short rows (rows < split_row = class A)
  for (int64_t r = 0; r < n_rows;) {
    if (split_row && r == split_row) n_a_items = (int64_t)it_row0.size() - n_hub_items;
    const int64_t len_r = h_indptr[r + 1] - h_indptr[r];
    if (len_r > kSegLen) { ++r; continue; }
    const int64_t r0 = r;
    int64_t tot = 0;
    int nr = 0;
    while (r < n_rows && nr < item_rows) {
      const int64_t l = h_indptr[r + 1] - h_indptr[r];
      if (l > kSegLen) break;                       // a split row ends the run
      if (nr > 0 && tot + l > item_nnz) break;
      if (nr > 0 && split_row && r == split_row) break;   // runs do not straddle the classes
      tot += l; ++nr; ++r;
    }
    it_row0.push_back((int32_t)r0); it_nrows.push_back(nr); it_begin.push_back(h_indptr[r0]);
    it_len.push_back((int32_t)tot); it_slot.push_back(-1);
  }

  SpmmPlan* p = new (std::nothrow) SpmmPlan();
  NR_REQUIRE(p, NR_ERR_ARG, "spmm_plan_create: out of host memory");
  p->n_rows = n_rows;
  p->nnz = nnz;
  for (int i = 0; i < 5; ++i) p->blocked[i] = nullptr;
  p->item_rows = item_rows;
  p->item_nnz = item_nnz;
  p->n_hub_items = n_hub_items;
  if (!split_row) n_a_items = (int64_t)it_row0.size() - n_hub_items;
  p->n_a_items = n_a_items;
  p->n_b_items = (int64_t)it_row0.size() - n_hub_items - n_a_items;
  p->n_seg = (int64_t)seg_row.size();
  p->n_items = (int64_t)it_row0.size();
  p->n_multi_seg = n_multi_seg;
  p->n_multi_row = (int64_t)multi_row.size();
  const PlanSizes sz = plan_sizes(n_rows, nnz);
  char* q = (char*)d_plan_buf;
  auto carve4 = [&](size_t n) { int32_t* r = (int32_t*)q; q += nr_align_up(n * 4, 256); return r; };
  auto carve8 = [&](size_t n) { int64_t* r = (int64_t*)q; q += nr_align_up(n * 8, 256); return r; };
  p->seg_row = carve4(sz.max_seg);
  p->seg_len = carve4(sz.max_seg);
  p->seg_slot = carve4(sz.max_seg);
  p->seg_begin = carve8(sz.max_seg);
  p->multi_row = carve4(sz.max_multi);
  p->multi_first = carve4(sz.max_multi);
  p->multi_nseg = carve4(sz.max_multi);
  p->item_row0 = carve4(sz.max_item);
  p->item_nrows = carve4(sz.max_item);
  p->item_len = carve4(sz.max_item);
  p->item_slot = carve4(sz.max_item);
  p->item_begin = carve8(sz.max_item);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  auto up = [&](void* dst, const void* src, size_t n) {
    if (n && e == hipSuccess) e = hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st);
  };
  up(p->seg_row, seg_row.data(), seg_row.size() * 4);
  up(p->seg_begin, seg_begin.data(), seg_begin.size() * 8);
  up(p->seg_len, seg_len.data(), seg_len.size() * 4);
  up(p->seg_slot, seg_slot.data(), seg_slot.size() * 4);
  up(p->multi_row, multi_row.data(), multi_row.size() * 4);
  up(p->multi_first, multi_first.data(), multi_first.size() * 4);
  up(p->multi_nseg, multi_nseg.data(), multi_nseg.size() * 4);
  up(p->item_row0, it_row0.data(), it_row0.size() * 4);
  up(p->item_nrows, it_nrows.data(), it_nrows.size() * 4);
  up(p->item_begin, it_begin.data(), it_begin.size() * 8);
  up(p->item_len, it_len.data(), it_len.size() * 4);
  up(p->item_slot, it_slot.data(), it_slot.size() * 4);
  if (e == hipSuccess) e = hipStreamSynchronize(st);   // host vectors die at return
  if (e != hipSuccess) {
    delete p;
    nrhip_set_error("spmm_plan_create: upload failed: %s", hipGetErrorString(e));
    return NR_ERR_HIP;
  }
  *plan_out = p;
  return NR_OK;
}

int nrhip_spmm_plan_destroy(void* plan) {
  delete (SpmmPlan*)plan;
  return NR_OK;
}

int nrhip_spmm_plan_attach_blocked(void* plan, const void* blocked_plan, int d) {
  NR_REQUIRE(plan && blocked_slot(d) >= 0, NR_ERR_ARG,
             "spmm_plan_attach_blocked: null plan or dim %d not in (16, 32, 64, 128, 256)", d);
  ((SpmmPlan*)plan)->blocked[blocked_slot(d)] = blocked_plan;       // NULL detaches
  return NR_OK;
}

/* Last backward hop + optimiser in one pass: var/m/v <- TF-Adam with gradient
 * (A·X + addend) + grad_b.  Needs the d = 64 lane-group schedule attached to the plan;
 * NR_ERR_UNSUPPORTED otherwise (callers then run nrhip_spmm_csr + nrhip_adam_dense_tf2). */
int nrhip_spmm_csr_adam(const void* plan, const int32_t* d_indices, const float* d_vals,
                        const float* d_X, int d, float* d_addend, float* d_grad_b, float* d_var,
                        float* d_m, float* d_v, float alpha, float beta1, float beta2, float eps,
                        int clear_consumed, uint8_t* d_row_flag, void* stream) {
  NR_REQUIRE(plan, NR_ERR_ARG, "spmm_csr_adam: null plan");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  NR_REQUIRE(d == 64 && p->blocked[blocked_slot(64)], NR_ERR_UNSUPPORTED,
             "spmm_csr_adam: needs the d = 64 lane-group schedule");
  return nrhip_spmm_blocked_adam(p->blocked[blocked_slot(64)], d_indices, d_vals, d_X, d_addend, d_grad_b, d_var,
                                 d_m, d_v, alpha, beta1, beta2, eps, clear_consumed, d_row_flag,
                                 stream);
}

int nrhip_spmm_csr_wanted_layers(const void* plan, const int32_t* d_indices, const float* d_vals,
                                 const float* d_X, int d, const float* d_sum_in,
                                 const float* d_layer_a, const float* d_layer_b, float* d_sum_out,
                                 const uint8_t* d_y_row_wanted, void* stream) {
  NR_REQUIRE(plan, NR_ERR_ARG, "spmm_csr_wanted_layers: null plan");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  NR_REQUIRE(d == 64 && p->blocked[blocked_slot(64)], NR_ERR_UNSUPPORTED,
             "spmm_csr_wanted_layers: needs the d = 64 lane-group schedule");
  return nrhip_spmm_blocked_wanted_layers(p->blocked[blocked_slot(64)], d_indices, d_vals, d_X, d_sum_in,
                                          d_layer_a, d_layer_b, d_sum_out, d_y_row_wanted, stream);
}

int nrhip_spmm_csr_wanted_batch(const void* plan, const int32_t* d_indices, const float* d_vals,
                                const float* d_X, int d, const float* d_sum_in, const float* d_layer_a,
                                const float* d_layer_b, float* d_sum_out, const int32_t* d_users,
                                const int32_t* d_pos, const int32_t* d_neg, int batch, int n_users,
                                uint8_t* d_row_flag, int32_t* d_rows_out, void* stream) {
  NR_REQUIRE(plan, NR_ERR_ARG, "spmm_csr_wanted_batch: null plan");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  NR_REQUIRE(d == 64 && p->blocked[blocked_slot(64)], NR_ERR_UNSUPPORTED,
             "spmm_csr_wanted_batch: needs the d = 64 lane-group schedule");
  return nrhip_spmm_blocked_wanted_batch(p->blocked[blocked_slot(64)], d_indices, d_vals, d_X, d_sum_in,
                                         d_layer_a, d_layer_b, d_sum_out, d_users, d_pos, d_neg, batch,
                                         n_users, d_row_flag, d_rows_out, stream);
}

int nrhip_spmm_plan_has_wanted(const void* plan, int d) {
  if (!plan || d != 64) return 0;
  const SpmmPlan* p = (const SpmmPlan*)plan;
  return p->blocked[blocked_slot(64)] ? nrhip_spmm_blocked_has_wanted(p->blocked[blocked_slot(64)]) : 0;
}

int nrhip_spmm_plan_has_blocked(const void* plan, int d) {
  if (!plan) return 0;
  const SpmmPlan* p = (const SpmmPlan*)plan;
  const int slot = blocked_slot(d);
  return slot >= 0 && p->blocked[slot] != nullptr;
}

int nrhip_spmm_plan_info(const void* plan, int64_t* n_work_items, int64_t* n_split_rows) {
  NR_REQUIRE(plan, NR_ERR_ARG, "spmm_plan_info: null plan");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  if (n_work_items) *n_work_items = p->n_items;
  if (n_split_rows) *n_split_rows = p->n_multi_row;
  return NR_OK;
}

int nrhip_spmm_workspace_bytes(const void* plan, int d, size_t* bytes) {
  NR_REQUIRE(plan && bytes && d >= 1, NR_ERR_ARG, "spmm_workspace_bytes: bad arguments");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  *bytes = nr_align_up((size_t)(p->n_multi_seg + 1) * (size_t)d * sizeof(float), 256);
  return NR_OK;
}

static int spmm_dispatch(const char* who, const void* plan, const int64_t* d_indptr,
                         const int32_t* d_indices, const float* d_vals, const float* d_X, int d,
                         float* d_Y, const float* d_addend, const float* d_sum_in,
                         float* d_sum_out, const uint8_t* d_col_mask, const uint8_t* d_row_mask,
                         void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(plan && d_indptr && d_indices && d_vals && d_X && (d_Y || d_sum_out), NR_ERR_ARG,
             "%s: null pointer argument", who);
  NR_REQUIRE((d_sum_out == nullptr) || (d_sum_in != nullptr), NR_ERR_ARG,
             "%s: sum_out needs sum_in", who);
  const SpmmPlan* p = (const SpmmPlan*)plan;
  hipStream_t st = (hipStream_t)stream;
  const int bslot = blocked_slot(d);
  if (bslot >= 0 && p->blocked[bslot])   // persistent lane-group kernel (same contract, same masks)
    return nrhip_spmm_blocked(p->blocked[bslot], d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in,
                              d_sum_out, d_col_mask, d_row_mask, stream);
  if (d < 64)   // the 16/32-wide path has no work-skipping variants
    NR_REQUIRE(d_row_mask == nullptr, NR_ERR_UNSUPPORTED,
               "%s: row mask needs an embedding dim >= 64", who);
  if (p->n_multi_seg > 0)
    NR_REQUIRE(d_ws && ws_bytes >= (size_t)p->n_multi_seg * d * sizeof(float), NR_ERR_WORKSPACE,
               "%s: workspace too small for %lld split-row partials", who,
               (long long)p->n_multi_seg);
  float* part = (float*)d_ws;
  switch (d) {
    case 16: return launch_segs<16>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    case 32: return launch_segs<32>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    case 64: return launch_items<64, 4>(p, d_indptr, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, d_col_mask, d_row_mask, st);
    case 128: return launch_items<128, 2>(p, d_indptr, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, d_col_mask, d_row_mask, st);
    case 256: return launch_items<256, 1>(p, d_indptr, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, d_col_mask, d_row_mask, st);
    default:
      NR_REQUIRE(false, NR_ERR_UNSUPPORTED,
                 "%s: embedding dim %d not built (16, 32, 64, 128, 256)", who, d);
  }
  return NR_OK;
}

int nrhip_spmm_csr(const void* plan, const int64_t* d_indptr, const int32_t* d_indices,
                   const float* d_vals, const float* d_X, int d, float* d_Y,
                   const float* d_addend, const float* d_sum_in, float* d_sum_out, void* d_ws,
                   size_t ws_bytes, void* stream) {
  return spmm_dispatch("spmm_csr", plan, d_indptr, d_indices, d_vals, d_X, d, d_Y, d_addend,
                       d_sum_in, d_sum_out, nullptr, nullptr, d_ws, ws_bytes, stream);
}

int nrhip_spmm_csr_masked(const void* plan, const int64_t* d_indptr, const int32_t* d_indices,
                          const float* d_vals, const float* d_X, const uint8_t* d_x_row_nonzero,
                          const uint8_t* d_y_row_wanted, int d, float* d_Y, const float* d_addend,
                          const float* d_sum_in, float* d_sum_out, void* d_ws, size_t ws_bytes,
                          void* stream) {
  NR_REQUIRE(d_x_row_nonzero || d_y_row_wanted, NR_ERR_ARG, "spmm_csr_masked: no mask given");
  NR_REQUIRE(!(d_y_row_wanted && d_sum_out && d_sum_out == d_sum_in) || true, NR_ERR_ARG, "");
  return spmm_dispatch("spmm_csr_masked", plan, d_indptr, d_indices, d_vals, d_X, d, d_Y,
                       d_addend, d_sum_in, d_sum_out, d_x_row_nonzero, d_y_row_wanted, d_ws,
                       ws_bytes, stream);
}

int nrhip_spmm_csr_rows(const int64_t* d_indptr, const int32_t* d_indices, const float* d_vals,
                        const float* d_X, int d, const int32_t* d_rows, int n_listed, float* d_Y,
                        const float* d_addend, const float* d_sum_in, float* d_sum_out,
                        void* stream) {
  NR_REQUIRE(d_indptr && d_indices && d_vals && d_X && d_rows && (d_Y || d_sum_out), NR_ERR_ARG,
             "spmm_csr_rows: null pointer argument");
  NR_REQUIRE((d_sum_out == nullptr) || (d_sum_in != nullptr && d_sum_in != d_sum_out), NR_ERR_ARG,
             "spmm_csr_rows: sum_out needs a distinct sum_in (listed rows may repeat)");
  NR_REQUIRE(n_listed >= 0, NR_ERR_ARG, "spmm_csr_rows: negative count");
  if (n_listed == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)n_listed), block(4 * NR_WAVE);
  switch (d) {
    case 64: hipLaunchKernelGGL(spmm_rows_kernel<64>, grid, block, 0, st, d_rows, d_indptr, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out); break;
    case 128: hipLaunchKernelGGL(spmm_rows_kernel<128>, grid, block, 0, st, d_rows, d_indptr, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out); break;
    case 256: hipLaunchKernelGGL(spmm_rows_kernel<256>, grid, block, 0, st, d_rows, d_indptr, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out); break;
    default:
      NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "spmm_csr_rows: embedding dim %d not built (64, 128, 256)", d);
  }
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* One chunk of a chunked hop (neurec_amd/sharded.py: the row-sharded propagation with the all-gather received in
 * rank-ordered chunks): Yv[v] = (has_carry ? Yv[v] : 0) + Σ_j a_j·X[col_j] over the non-zeros this chunk's CSR holds
 * for virtual row v, ONE ascending chain across the launches of a hop.  Column c < x_split is row c of d_X (the
 * rank's own block), any other column row c - x_split of d_X2 (the received chunk).  The CSR must not contain a row
 * of more than 256 non-zeros (virtual rows: a real row's 256-non-zero segments).  d_row_mask (optional, per virtual row): 0 =
 * row not wanted, left untouched. */
int nrhip_spmm_csr_carry(const void* plan, const int64_t* d_indptr, const int32_t* d_indices, const float* d_vals,
                         const float* d_X, const float* d_X2, int x_split, int d, float* d_Yv, int has_carry,
                         const uint8_t* d_row_mask, void* stream) {
  NR_REQUIRE(plan && d_indptr && d_indices && d_vals && d_X && d_X2 && d_Yv && x_split >= 0, NR_ERR_ARG,
             "spmm_csr_carry: null pointer argument");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  NR_REQUIRE(p->n_multi_seg == 0 && p->n_hub_items == 0, NR_ERR_UNSUPPORTED,
             "spmm_csr_carry: a row of more than 256 non-zeros (chunk CSRs hold virtual rows)");
  hipStream_t st = (hipStream_t)stream;
  switch (d) {
    case 64: return launch_items_carry<64, 4>(p, d_indptr, d_indices, d_vals, d_X, d_X2, x_split, d_Yv, has_carry != 0, d_row_mask, st);
    case 128: return launch_items_carry<128, 2>(p, d_indptr, d_indices, d_vals, d_X, d_X2, x_split, d_Yv, has_carry != 0, d_row_mask, st);
    case 256: return launch_items_carry<256, 1>(p, d_indptr, d_indices, d_vals, d_X, d_X2, x_split, d_Yv, has_carry != 0, d_row_mask, st);
    default:
      NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "spmm_csr_carry: embedding dim %d not built (64, 128, 256)", d);
  }
  return NR_OK;
}

/* After the last chunk: y[r] = Yv[first_vrow[r]] + Yv[first_vrow[r]+1] + ... (segment order), then the fused
 * epilogue of nrhip_spmm_csr (y += addend; sum_out = sum_in + y); d_row_mask per REAL row (0: row left untouched). */
int nrhip_spmm_chunks_finish(const int32_t* d_first_vrow, int64_t n_rows, const float* d_Yv, int d, float* d_Y,
                             const float* d_addend, const float* d_sum_in, float* d_sum_out,
                             const uint8_t* d_row_mask, void* stream) {
  NR_REQUIRE(d_first_vrow && d_Yv && (d_Y || d_sum_out) && n_rows >= 0, NR_ERR_ARG, "spmm_chunks_finish: bad arguments");
  NR_REQUIRE((d_sum_out == nullptr) || (d_sum_in != nullptr), NR_ERR_ARG, "spmm_chunks_finish: sum_out needs sum_in");
  if (n_rows == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((n_rows + kLegacyWaves - 1) / kLegacyWaves)), block(kLegacyWaves * NR_WAVE);
  switch (d) {
    case 64: hipLaunchKernelGGL(spmm_chunks_finish_kernel<64>, grid, block, 0, st, d_first_vrow, n_rows, d_Yv, d_Y, d_addend, d_sum_in, d_sum_out, d_row_mask); break;
    case 128: hipLaunchKernelGGL(spmm_chunks_finish_kernel<128>, grid, block, 0, st, d_first_vrow, n_rows, d_Yv, d_Y, d_addend, d_sum_in, d_sum_out, d_row_mask); break;
    case 256: hipLaunchKernelGGL(spmm_chunks_finish_kernel<256>, grid, block, 0, st, d_first_vrow, n_rows, d_Yv, d_Y, d_addend, d_sum_in, d_sum_out, d_row_mask); break;
    default:
      NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "spmm_chunks_finish: embedding dim %d not built (64, 128, 256)", d);
  }
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_partials_sum_rows(const float* d_parts, int world, int64_t n_rows, int d, float* d_Y, const float* d_addend,
                            const float* d_sum_in, float* d_sum_out, const uint8_t* d_row_mask, void* stream) {
  NR_REQUIRE(d_parts && (d_Y || d_sum_out) && world >= 1 && n_rows >= 0, NR_ERR_ARG,
             "partials_sum_rows: null pointer / bad count");
  NR_REQUIRE((d_sum_out == nullptr) || (d_sum_in != nullptr), NR_ERR_ARG, "partials_sum_rows: sum_out needs sum_in");
  NR_REQUIRE(d > 0 && d % 4 == 0, NR_ERR_UNSUPPORTED, "partials_sum_rows: width %d is not a multiple of 4", d);
  if (n_rows == 0) return NR_OK;
  const int64_t n4 = n_rows * (d / 4);
  const int64_t blocks = std::min<int64_t>((n4 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(partials_sum_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)d_parts, world, n_rows, d / 4, (float4*)d_Y, (const float4*)d_addend,
                     (const float4*)d_sum_in, (float4*)d_sum_out, d_row_mask);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
