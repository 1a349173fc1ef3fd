// spmm.hip — sparse adjacency × dense embedding propagation  Y = Â · X.
//
// Stands in for tf.sparse_tensor_dense_matmul(adj, ego)
// (model/general_recommender/LightGCN.py:140; NGCF.py:176 and its 100-slab
// split NGCF.py:320-332) and, with the transposed CSR, for its gradient.
//
// Â is CSR (row-major, ascending columns — what the reference hands to TF,
// LightGCN.py:151-154).  X, Y are row-major [N][d] fp32.  One wave64 owns one
// row *segment* (<= 256 non-zeros): the lanes span the d columns, so each
// gathered X row is one coalesced 4·d-byte read, the 64 column indices/values
// of a chunk are fetched with one coalesced load and then broadcast lane by
// lane through v_readlane (no LDS).  Products and sums are rounded separately
// in ascending column order — the order the reference's CPU kernel uses — so
// single-segment rows are bit-identical to oracle/.  Rows longer than one
// segment (hub items/users) are split so no wave serialises thousands of
// dependent gathers; their partial sums are combined in segment order by a
// second tiny kernel (deterministic, not atomics).
//
// Fused epilogue (saves a pass over [N][d] per layer):
//     y = Σ_j a_j x_j ; y += addend[r]  (backward: the dE*/(L+1) term)
//     sum_out[r] = sum_in[r] + y          (forward: running layer sum for
//                                          mean over layers, LightGCN.py:146-147)
//
// Roofline: HBM.  Algorithmic bytes per pass = nnz·8 + (N+1)·8 + 2·N·d·4
// (SURVEY.md §8d); the gathers themselves are served by L2 / Infinity Cache.
#include "nr_common.h"
#include <vector>
#include <new>

namespace {

constexpr int kSegLen = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kGather = 8;   // row gathers in flight per wave

struct SpmmPlan {
  int64_t n_rows;
  int64_t nnz;
  int64_t n_seg;        // all segments
  int64_t n_multi_seg;  // segments that belong to multi-segment rows (stored first)
  int64_t n_multi_row;
  // device arrays, carved from the caller's buffer
  int32_t* seg_row;
  int64_t* seg_begin;
  int32_t* seg_len;
  int32_t* seg_slot;    // -1: single-segment row; else index into the partial buffer
  int32_t* multi_row;
  int32_t* multi_first; // first partial slot of the row
  int32_t* multi_nseg;
};

size_t plan_bytes_for(int64_t n_rows, int64_t nnz) {
  const int64_t max_seg = n_rows + nnz / kSegLen + 1;
  const int64_t max_multi = nnz / kSegLen + 1;
  return nr_align_up((size_t)max_seg * 4, 256) + nr_align_up((size_t)max_seg * 8, 256) +
         nr_align_up((size_t)max_seg * 4, 256) * 2 + nr_align_up((size_t)max_multi * 4, 256) * 3;
}

template <int D>
struct Shape {
  static constexpr int LPR = D < 64 ? D : 64;   // lanes per row
  static constexpr int RPW = 64 / LPR;          // row segments per wave
  static constexpr int CPL = D / LPR;           // columns per lane
};

template <int D>
__device__ __forceinline__ void epilogue_store(float (&acc)[Shape<D>::CPL], int64_t row, int col0,
                                               float* __restrict__ Y,
                                               const float* __restrict__ addend,
                                               const float* sum_in, float* sum_out) {
#pragma unroll
  for (int c = 0; c < Shape<D>::CPL; ++c) {
    const int64_t o = row * D + col0 + c * Shape<D>::LPR;
    float y = acc[c];
    if (addend) y = __fadd_rn(y, addend[o]);
    if (Y) Y[o] = y;
    if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
  }
}

template <int D>
__global__ __launch_bounds__(kWavesPerBlock* NR_WAVE) void spmm_seg_kernel(
    const int32_t* __restrict__ seg_row, const int64_t* __restrict__ seg_begin,
    const int32_t* __restrict__ seg_len, const int32_t* __restrict__ seg_slot, int64_t n_seg,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ X, float* __restrict__ Y, const float* __restrict__ addend,
    const float* sum_in, float* sum_out, float* __restrict__ partial) {   // sum_in may alias sum_out
  using S = Shape<D>;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  float acc[S::CPL];
#pragma unroll
  for (int c = 0; c < S::CPL; ++c) acc[c] = 0.f;

  if constexpr (S::RPW == 1) {
    const int64_t seg_v = wave_id;
    if (seg_v >= n_seg) return;
    // wave-uniform descriptor -> SGPRs (threadIdx-derived values look divergent to hipcc)
    const int row = __builtin_amdgcn_readfirstlane(seg_row[seg_v]);
    const int len = __builtin_amdgcn_readfirstlane(seg_len[seg_v]);
    const int slot = __builtin_amdgcn_readfirstlane(seg_slot[seg_v]);
    const int64_t b64 = seg_begin[seg_v];
    const int64_t b = ((int64_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 & 0xffffffff));
    for (int k0 = 0; k0 < len; k0 += NR_WAVE) {
      const int n = min(NR_WAVE, len - k0);
      int my_idx = 0;
      float my_val = 0.f;
      if (lane < n) {
        my_idx = indices[b + k0 + lane];
        my_val = vals[b + k0 + lane];
      }
      // kGather gathers are issued back to back (indices clamped so every load is
      // unconditional), then consumed in order: memory-level parallelism comes from
      // the batch, the summation order stays the sequential one.
      for (int t0 = 0; t0 < n; t0 += kGather) {
        float a[kGather];
        float x[kGather][S::CPL];
#pragma unroll
        for (int u = 0; u < kGather; ++u) {
          const int tt = min(t0 + u, n - 1);
          const int col = __builtin_amdgcn_readlane(my_idx, tt);
          a[u] = __builtin_bit_cast(
              float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), tt));
          const float* xr = X + (int64_t)col * D + lane;
#pragma unroll
          for (int c = 0; c < S::CPL; ++c) x[u][c] = xr[c * 64];
        }
#pragma unroll
        for (int u = 0; u < kGather; ++u) {
          if (t0 + u < n) {
#pragma unroll
            for (int c = 0; c < S::CPL; ++c)
              acc[c] = __fadd_rn(acc[c], __fmul_rn(a[u], x[u][c]));
          }
        }
      }
    }
    if (slot < 0) {
      epilogue_store<D>(acc, (int64_t)row, lane, Y, addend, sum_in, sum_out);
    } else {
#pragma unroll
      for (int c = 0; c < S::CPL; ++c) partial[(int64_t)slot * D + lane + c * 64] = acc[c];
    }
  } else {
    // several short rows per wave: each LPR-lane group walks its own segment
    const int g = lane / S::LPR, col = lane % S::LPR;
    const int64_t seg = wave_id * S::RPW + g;
    const bool live = seg < n_seg;
    const int row = live ? seg_row[seg] : 0;
    const int len = live ? seg_len[seg] : 0;
    const int slot = live ? seg_slot[seg] : -1;
    const int64_t b = live ? seg_begin[seg] : 0;
    int max_len = len;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) max_len = max(max_len, __shfl_xor(max_len, m, NR_WAVE));
    for (int t = 0; t < max_len; ++t) {
      if (t < len) {
        const int cidx = indices[b + t];
        const float a = vals[b + t];
        acc[0] = __fadd_rn(acc[0], __fmul_rn(a, X[(int64_t)cidx * D + col]));
      }
    }
    if (live) {
      if (slot < 0) {
        epilogue_store<D>(acc, (int64_t)row, col, Y, addend, sum_in, sum_out);
      } else {
        partial[(int64_t)slot * D + col] = acc[0];
      }
    }
  }
}

// combine the partial sums of multi-segment rows, in segment order
template <int D>
__global__ __launch_bounds__(kWavesPerBlock* NR_WAVE) void spmm_fix_kernel(
    const int32_t* __restrict__ multi_row, const int32_t* __restrict__ multi_first,
    const int32_t* __restrict__ multi_nseg, int64_t n_multi, const float* __restrict__ partial,
    float* __restrict__ Y, const float* __restrict__ addend, const float* sum_in,
    float* sum_out) {
  using S = Shape<D>;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int64_t m = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (m >= n_multi) return;
  if (lane >= S::LPR) return;
  const int row = multi_row[m], first = multi_first[m], nseg = multi_nseg[m];
  float acc[S::CPL];
#pragma unroll
  for (int c = 0; c < S::CPL; ++c) acc[c] = partial[(int64_t)first * D + lane + c * S::LPR];
  for (int s = 1; s < nseg; ++s) {
#pragma unroll
    for (int c = 0; c < S::CPL; ++c)
      acc[c] = __fadd_rn(acc[c], partial[(int64_t)(first + s) * D + lane + c * S::LPR]);
  }
  epilogue_store<D>(acc, (int64_t)row, lane, Y, addend, sum_in, sum_out);
}

template <int D>
int launch_spmm(const SpmmPlan* p, const int32_t* indices, const float* vals, const float* X,
                float* Y, const float* addend, const float* sum_in, float* sum_out, float* partial,
                hipStream_t st) {
  using S = Shape<D>;
  const int64_t waves = (p->n_seg + S::RPW - 1) / S::RPW;
  const int64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 0) {
    hipLaunchKernelGGL(spmm_seg_kernel<D>, dim3((unsigned)blocks), dim3(kWavesPerBlock * NR_WAVE),
                       0, st, p->seg_row, p->seg_begin, p->seg_len, p->seg_slot, p->n_seg, indices,
                       vals, X, Y, addend, sum_in, sum_out, partial);
    NR_LAUNCH_CHECK();
  }
  if (p->n_multi_row > 0) {
    const int64_t fb = (p->n_multi_row + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(spmm_fix_kernel<D>, dim3((unsigned)fb), dim3(kWavesPerBlock * NR_WAVE), 0,
                       st, p->multi_row, p->multi_first, p->multi_nseg, p->n_multi_row, partial, Y,
                       addend, sum_in, sum_out);
    NR_LAUNCH_CHECK();
  }
  return NR_OK;
}

}  // namespace

extern "C" {

int nrhip_spmm_plan_bytes(int64_t n_rows, int64_t nnz, size_t* bytes) {
  NR_REQUIRE(bytes && n_rows >= 0 && nnz >= 0, NR_ERR_ARG, "spmm_plan_bytes: bad arguments");
  *bytes = plan_bytes_for(n_rows, nnz);
  return NR_OK;
}

int nrhip_spmm_plan_create(const int64_t* h_indptr, int64_t n_rows, void* d_plan_buf,
                           size_t plan_bytes, void* stream, void** plan_out) {
  NR_REQUIRE(h_indptr && d_plan_buf && plan_out && n_rows >= 0, NR_ERR_ARG,
             "spmm_plan_create: bad arguments");
  NR_REQUIRE(n_rows < (int64_t)0x7fffffff, NR_ERR_UNSUPPORTED, "spmm: more than 2^31 rows");
  const int64_t nnz = h_indptr[n_rows] - h_indptr[0];
  NR_REQUIRE(plan_bytes >= plan_bytes_for(n_rows, nnz), NR_ERR_WORKSPACE,
             "spmm_plan_create: plan buffer %zu < %zu bytes", plan_bytes,
             plan_bytes_for(n_rows, nnz));
  std::vector<int32_t> seg_row, seg_len, seg_slot, multi_row, multi_first, multi_nseg;
  std::vector<int64_t> seg_begin;
  seg_row.reserve(n_rows);
  // long rows first (their segments take longest; start them before the short tail)
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
        seg_row.push_back((int32_t)r);
        seg_begin.push_back(b + (int64_t)s * kSegLen);
        seg_len.push_back((int32_t)std::min<int64_t>(kSegLen, len - (int64_t)s * kSegLen));
        seg_slot.push_back(slot++);
      }
    }
  }
  const int64_t n_multi_seg = (int64_t)seg_row.size();
  for (int64_t r = 0; r < n_rows; ++r) {
    const int64_t b = h_indptr[r], len = h_indptr[r + 1] - b;
    if (len <= kSegLen) {
      seg_row.push_back((int32_t)r);
      seg_begin.push_back(b);
      seg_len.push_back((int32_t)len);
      seg_slot.push_back(-1);
    }
  }
  SpmmPlan* p = new (std::nothrow) SpmmPlan();
  NR_REQUIRE(p, NR_ERR_ARG, "spmm_plan_create: out of host memory");
  p->n_rows = n_rows;
  p->nnz = nnz;
  p->n_seg = (int64_t)seg_row.size();
  p->n_multi_seg = n_multi_seg;
  p->n_multi_row = (int64_t)multi_row.size();
  const int64_t max_seg = n_rows + nnz / kSegLen + 1;
  const int64_t max_multi = nnz / kSegLen + 1;
  char* q = (char*)d_plan_buf;
  p->seg_row = (int32_t*)q;     q += nr_align_up((size_t)max_seg * 4, 256);
  p->seg_begin = (int64_t*)q;   q += nr_align_up((size_t)max_seg * 8, 256);
  p->seg_len = (int32_t*)q;     q += nr_align_up((size_t)max_seg * 4, 256);
  p->seg_slot = (int32_t*)q;    q += nr_align_up((size_t)max_seg * 4, 256);
  p->multi_row = (int32_t*)q;   q += nr_align_up((size_t)max_multi * 4, 256);
  p->multi_first = (int32_t*)q; q += nr_align_up((size_t)max_multi * 4, 256);
  p->multi_nseg = (int32_t*)q;
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

int nrhip_spmm_plan_info(const void* plan, int64_t* n_segments, int64_t* n_split_rows) {
  NR_REQUIRE(plan, NR_ERR_ARG, "spmm_plan_info: null plan");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  if (n_segments) *n_segments = p->n_seg;
  if (n_split_rows) *n_split_rows = p->n_multi_row;
  return NR_OK;
}

int nrhip_spmm_workspace_bytes(const void* plan, int d, size_t* bytes) {
  NR_REQUIRE(plan && bytes && d >= 1, NR_ERR_ARG, "spmm_workspace_bytes: bad arguments");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  *bytes = nr_align_up((size_t)(p->n_multi_seg + 1) * (size_t)d * sizeof(float), 256);
  return NR_OK;
}

int nrhip_spmm_csr(const void* plan, const int32_t* d_indices, const float* d_vals,
                   const float* d_X, int d, float* d_Y, const float* d_addend,
                   const float* d_sum_in, float* d_sum_out, void* d_ws, size_t ws_bytes,
                   void* stream) {
  NR_REQUIRE(plan && d_indices && d_vals && d_X && (d_Y || d_sum_out), NR_ERR_ARG,
             "spmm_csr: null pointer argument");
  NR_REQUIRE((d_sum_out == nullptr) || (d_sum_in != nullptr), NR_ERR_ARG,
             "spmm_csr: sum_out needs sum_in");
  const SpmmPlan* p = (const SpmmPlan*)plan;
  if (p->n_multi_seg > 0)
    NR_REQUIRE(d_ws && ws_bytes >= (size_t)p->n_multi_seg * d * sizeof(float), NR_ERR_WORKSPACE,
               "spmm_csr: workspace too small for %lld split-row partials",
               (long long)p->n_multi_seg);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)d_ws;
  switch (d) {
    case 16: return launch_spmm<16>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    case 32: return launch_spmm<32>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    case 64: return launch_spmm<64>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    case 128: return launch_spmm<128>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    case 256: return launch_spmm<256>(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, part, st);
    default:
      NR_REQUIRE(false, NR_ERR_UNSUPPORTED,
                 "spmm_csr: embedding dim %d not built (16, 32, 64, 128, 256)", d);
  }
  return NR_OK;
}

}  // extern "C"
