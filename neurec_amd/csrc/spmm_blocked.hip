// spmm_blocked.hip — persistent-workgroup, lane-group SpMM  Y = Â·X  for d = 64 / 128 / 256
// (optionally cache-blocked by column windows).
//
// Why: at the gowalla shape one pass gathers 1.62 M rows of 256 B (415 MB) out of an 18 MB
// table.  A pure random-gather micro-benchmark of that pattern (scripts/exp_gather.py) tops out
// at ≈9 TB/s — the L2-miss path into the Infinity Cache — and the work-item kernel of spmm.hip
// already sits on that ceiling.  The same gathers run at 16-17 TB/s when (a) every XCD only
// touches a ≤2.6 MB window of the table at a time, so its private 4 MB L2 holds it, and (b) a load
// instruction moves four rows (16 lanes × 16 B each) instead of one
// (scripts/exp_gather_blocked.py).  This kernel is built on those two facts.
//
// Schedule (host, once per matrix):
//   * one workgroup per CU (16 waves), workgroup b on XCD b % 8.  With a bipartite split the
//     user rows go to XCDs 0-3 and the item rows to XCDs 4-7 (they gather from disjoint halves
//     of the table); each workgroup owns a contiguous run of rows, balanced by non-zeros;
//   * the column range a class gathers from is cut into K blocks of ≤ block_bytes; a row's
//     non-zeros (ascending columns) fall into ≤ K contiguous sub-lists, one per block;
//   * phase k of the kernel walks the sub-lists of block k.  All workgroups of an XCD move
//     through the phases at about the same pace (equal work, no barrier needed), so the XCD's
//     L2 holds one block at a time.
// Kernel: every row of the workgroup has a 256-byte accumulator in LDS that lives across the
// phases.  A 16-lane group owns one sub-list: it loads the accumulator, adds a_j·X[col_j] in
// ascending column order (products and sums rounded separately — the order of the reference's
// CPU kernel, so rows are bit-identical to oracle/), and stores it back.  Sub-lists longer
// than 64 are cut into segments with their own partial accumulators, added in segment order
// after the phase (deterministic; only such hub rows deviate from the sequential order).
// Epilogue as in spmm.hip: y += addend, sum_out = sum_in + y.
#include "nr_common.h"
#include <algorithm>
#include <new>
#include <cstdlib>
#include <queue>
#include <functional>
#include <utility>
#include <vector>

namespace {

constexpr int kSegDefault = 64;     // longest sub-list one lane group walks alone
constexpr int kRMaxDefault = 416;   // row accumulators per workgroup (104 KB)
constexpr int kPMaxDefault = 192;   // segment partial slots per workgroup and phase (48 KB)
constexpr int kMaxPhases = 32;
constexpr int kMaxLdsBytes = 160 * 1024;

struct BlockedPlan {
  int64_t n_rows, nnz, n_ent, n_cmb;
  int n_wg, n_phases;
  int seg, r_max, p_max, waves, d;
  int32_t* wg_row0;      // first position of the workgroup's rows in row_of
  int32_t* wg_nrows;
  // r05: the rows of a workgroup are a LIST, not a run: row_of[wg_row0[w] + slot] (see "dealt rows" at the plan
  // builder) — and the plan owns the (column, value) pairs in that order, so a workgroup's pairs stay one
  // contiguous slice (the staged masked kernel reads it in bulk).  ent[].z / wg_nnz index the packed arrays.
  int32_t* row_of;       // [n_rows]
  uint32_t* pk_src;      // [n_rows] first CSR position of row_of[k]
  uint32_t* pk_dst;      // [n_rows + 1] first packed position of row_of[k]
  int32_t* pk_idx;       // [nnz]
  float* pk_val;         // [nnz]
  const int32_t* pk_from_idx;   // the CSR arrays the packed copy was made from (nullptr: not yet)
  const float* pk_from_val;
  int32_t* wg_ent_off;   // [n_wg][n_phases + 1]
  int32_t* wg_cmb_off;   // [n_wg][n_phases + 1]
  int4* ent;             // {accumulator slot, length, first non-zero (packed), owning row (global id)}
  int4* cmb;             // {row slot, first partial slot, segments, row (global id)}
  // masked hops of a training step (d = 64, one phase): dedicated kernels below
  uint32_t* wg_nnz;      // [n_wg][2] first non-zero of the workgroup's rows, count
  int nnz_cap, ent_cap;  // largest slice / descriptor list of a workgroup
  int colmask_ok;
  // wanted-rows schedule (row-masked hop): the same sub-lists dealt to the workgroups by descending
  // row length, descriptors carry global rows
  int4* w_ent;           // {0 | r_max + partial slot, length, first non-zero, row}
  int4* w_cmb;           // {row, first partial slot, segments, 0}
  int32_t* w_ent_off;    // [n_wg][2]
  int32_t* w_cmb_off;    // [n_wg][2]
  int wanted_ok, w_ent_cap, w_nnz_cap, w_bitmap_words;
  // wave-cooperative row-masked hop (spmm_wanted_wave_kernel): units dealt by descending length; a row
  // of > 64 non-zeros is cut into 64-segments grouped in chunks of <= 8 consecutive segments, each
  // chunk a unit of its own (a hub's bytes spread over several CUs); partial sums in global memory
  int ww_ok, ww_ent_cap;
  int32_t* ww_off;       // [n_wg + 1] entries
  int32_t* ww_choff;     // [n_wg + 1] chunks
  int4* ww_ent;          // {0 | 1 + global partial slot | -(1 + LDS slot), length, first non-zero, row}
  int32_t* ww_gch;       // hub index of every chunk
  int4* ww_hub;          // {row, first partial slot, segments, chunks}
  int32_t* ww_lcoff;     // [n_wg + 1] one-chunk rows (segment sums stay in LDS)
  int4* ww_lcmb;         // {row, first LDS slot, segments, 0}
  int ww_lds_slots;      // LDS partial slots per workgroup
  float* ww_part;        // [segments of multi-chunk rows][64]
  unsigned* ww_cnt;      // [multi-chunk rows] chunks finished (zero between launches)
};

size_t blocked_plan_bytes(int64_t n_rows, int64_t nnz) {
  const size_t max_ent = (size_t)std::min<int64_t>(nnz, n_rows * (int64_t)kMaxPhases) + (size_t)(nnz / 16) + 64;
  const size_t max_cmb = (size_t)(nnz / 16) + 64;
  const size_t wg = 4096 + (size_t)(n_rows / 32);   // generous bound on workgroups
  const size_t w_ent = (size_t)n_rows + (size_t)(nnz / 16) + 64;        // wanted-rows schedule (one phase)
  // wave-cooperative wanted-rows schedule: an entry per row or 64-segment, a partial row per segment
  const size_t ww_seg = (size_t)(nnz / 32) + 64, ww_ents = (size_t)n_rows + ww_seg;
  const size_t ww = 3 * nr_align_up((wg + 1) * 4, 256) + nr_align_up(ww_ents * 16 + 16, 256) +
                    nr_align_up(ww_seg * 4 + 4, 256) + 2 * nr_align_up(max_cmb * 16 + 16, 256) +
                    nr_align_up(ww_seg * 256 + 256, 256) + nr_align_up(max_cmb * 4 + 4, 256);
  const size_t dealt = 3 * nr_align_up(((size_t)n_rows + 1) * 4, 256) + 2 * nr_align_up((size_t)nnz * 4 + 4, 256);
  return ww + dealt + nr_align_up(max_ent * 16, 256) + nr_align_up(max_cmb * 16, 256) +
         3 * nr_align_up(wg * 8, 256) + 2 * nr_align_up(wg * (kMaxPhases + 1) * 4, 256) +
         nr_align_up(w_ent * 16, 256) + nr_align_up(max_cmb * 16, 256) + 2 * nr_align_up(wg * 8, 256);
}

// Optional fused optimiser epilogue (last backward hop of a LightGCN step): instead of storing
// y, treat g = y + grad_b[row] as the dense gradient of `var` and apply TF-1.12 ApplyAdam to the
// row (training_ops: m += (g-m)(1-b1); v += (g²-v)(1-b2); var -= m·alpha/(sqrt(v)+eps)).
struct AdamEpilogue {
  float4* var; float4* m; float4* v; const float4* grad_b;
  float alpha, omb1, omb2, eps;
  // optional: re-arm the step's sparse buffers on the way (what rows_clear would do afterwards):
  // zero the addend / grad_b entries that were non-zero and the row flags
  float4* addend_rw; float4* grad_b_rw; uint8_t* flag_rw;
};


// Epilogue shared by the full-pass kernels: the workgroup's finished row sums sit in LDS
// (s_acc[slot][LPR]); y += addend, then either the ApplyAdam epilogue or the stores of y and of the
// running layer sum — a row is LPR lanes x 16 B (whole cache lines at d >= 32), so a row LIST
// streams as well as a row run.
template <bool MASKED, bool ADAM, int LPR>
__device__ __forceinline__ void blocked_epilogue(const float4* s_acc, int r0, int nr, int tid, int nthreads,
                                                 float4* __restrict__ Y, const float4* __restrict__ addend,
                                                 const float4* sum_in, float4* sum_out,
                                                 const uint8_t* __restrict__ row_mask, const AdamEpilogue& ad,
                                                 const int32_t* s_rows = nullptr) {
  for (int i = tid; i < nr * LPR; i += nthreads) {
    // s_rows: the workgroup's row list (LDS; dealt rows), else its rows are the run [r0, r0 + nr)
    const int row = s_rows ? s_rows[i / LPR] : r0 + i / LPR;
    if constexpr (MASKED) {
      if (row_mask && row_mask[row] == 0) continue;
    }
    const int64_t o = (int64_t)row * LPR + (i & (LPR - 1));
    float4 y = s_acc[i];
    // with row flags (re-arming step buffers) the addend / grad_b rows of unflagged rows are all
    // zero by contract and are not read: two of the seven epilogue streams touch batch rows only
    bool flagged = true;
    if constexpr (ADAM) {
      if (ad.flag_rw) flagged = ad.flag_rw[row] != 0;
    }
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (addend) {
      if (flagged) hv = addend[o];
      y.x = __fadd_rn(y.x, hv.x); y.y = __fadd_rn(y.y, hv.y);
      y.z = __fadd_rn(y.z, hv.z); y.w = __fadd_rn(y.w, hv.w);
    }
    if constexpr (ADAM) {
      float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (flagged) gb = ad.grad_b[o];
      if (ad.addend_rw) {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hv.x != 0.f || hv.y != 0.f || hv.z != 0.f || hv.w != 0.f) ad.addend_rw[o] = zero;
        if (gb.x != 0.f || gb.y != 0.f || gb.z != 0.f || gb.w != 0.f) ad.grad_b_rw[o] = zero;
        if ((i & (LPR - 1)) == 0 && ad.flag_rw && flagged) ad.flag_rw[row] = 0;
      }
      float4 w = ad.var[o], mm = ad.m[o], vv = ad.v[o];
      nr::adam_dense_tf(__fadd_rn(y.x, gb.x), w.x, mm.x, vv.x, ad.alpha, ad.omb1, ad.omb2, ad.eps);
      nr::adam_dense_tf(__fadd_rn(y.y, gb.y), w.y, mm.y, vv.y, ad.alpha, ad.omb1, ad.omb2, ad.eps);
      nr::adam_dense_tf(__fadd_rn(y.z, gb.z), w.z, mm.z, vv.z, ad.alpha, ad.omb1, ad.omb2, ad.eps);
      nr::adam_dense_tf(__fadd_rn(y.w, gb.w), w.w, mm.w, vv.w, ad.alpha, ad.omb1, ad.omb2, ad.eps);
      ad.var[o] = w; ad.m[o] = mm; ad.v[o] = vv;
      continue;
    }
    if (Y) Y[o] = y;
    if (sum_out) {
      const float4 si = sum_in[o];
      sum_out[o] = make_float4(__fadd_rn(si.x, y.x), __fadd_rn(si.y, y.y), __fadd_rn(si.z, y.z),
                               __fadd_rn(si.w, y.w));
    }
  }
}

// per-workgroup phase stamps (experiments: build with NEUREC_HIPCC_EXTRA=-DNR_BLK_TIMELINE, scripts/exp_blk_timeline.py)
#ifdef NR_BLK_TIMELINE
__device__ unsigned long long g_blk_dbg[4096 * 24];
#define NR_BLK_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_blk_dbg[blockIdx.x * 24 + (i)] = wall_clock64(); } while (0)
#define NR_BLK_WSTAMP() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_blk_dbg[blockIdx.x * 24 + 8 + (threadIdx.x >> 6)] = wall_clock64(); } while (0)
extern "C" int nrhip_exp_blk_timeline(unsigned long long* h_out) {
  if (hipDeviceSynchronize() != hipSuccess) return NR_ERR_HIP;
  return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_blk_dbg), sizeof(unsigned long long) * 4096 * 24) == hipSuccess ? NR_OK : NR_ERR_HIP;
}
#else
#define NR_BLK_STAMP(i) do {} while (0)
#define NR_BLK_WSTAMP() do {} while (0)
#endif

template <bool MASKED, int kWaves, int kG, int D, bool ADAM = false>
__global__ __launch_bounds__(kWaves* NR_WAVE) void spmm_blocked_kernel(
    const int32_t* __restrict__ wg_row0, const int32_t* __restrict__ wg_nrows,
    const int32_t* __restrict__ wg_ent_off, const int32_t* __restrict__ wg_cmb_off,
    const int4* __restrict__ ent, const int4* __restrict__ cmb, int n_phases,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float4* __restrict__ X, float4* __restrict__ Y, const float4* __restrict__ addend,
    const float4* sum_in, float4* sum_out, const uint8_t* __restrict__ col_mask,
    const uint8_t* __restrict__ row_mask, int kRMax, AdamEpilogue ad, const int32_t* __restrict__ row_of,
    int kPMax) {
  constexpr int LPR = D / 4;                   // lanes per row: 16-byte pieces of a d-float row
  constexpr int GPW = NR_WAVE / LPR;           // lane groups (rows in flight) per wave: 4 / 2 / 1
  constexpr int kGroups = kWaves * GPW;
  // a 16-entry (column, value) chunk of a sub-list is carried by CL lanes of the group, IPL
  // entries each (narrow rows have fewer than 16 lanes per group)
  constexpr int CL = LPR < 16 ? LPR : 16;
  constexpr int IPL = 16 / CL;
  static_assert(kG % IPL == 0, "broadcast component must be a compile-time index");
  extern __shared__ float4 s_acc[];          // [(r_max + p_max)][LPR], then the row list [r_max]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & (LPR - 1), g = lane / LPR, gbase = lane & ~(LPR - 1);
  const int wg = blockIdx.x;
  NR_BLK_STAMP(0);
  const int r0 = wg_row0[wg], nr = wg_nrows[wg];
  int32_t* s_rows = (int32_t*)(s_acc + (size_t)(kRMax + kPMax) * LPR);
  for (int i = tid; i < nr; i += kWaves * NR_WAVE) s_rows[i] = row_of[r0 + i];   // read back in the epilogue
  for (int i = tid; i < nr * LPR; i += kWaves * NR_WAVE) s_acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  NR_BLK_STAMP(1);
  const int32_t* eoff = wg_ent_off + (int64_t)wg * (n_phases + 1);
  const int32_t* coff = wg_cmb_off + (int64_t)wg * (n_phases + 1);
  for (int k = 0; k < n_phases; ++k) {
    const int e0 = eoff[k], e1 = eoff[k + 1];
    // Software pipeline over this lane group's sub-lists (entries ei, ei+64, ...): the next
    // entry's descriptor is requested before, and its first 16 (column, value) pairs right
    // after, the current entry's gathers are issued — so in steady state a sub-list costs one
    // gather round trip, not three dependent ones.
    int ei = e0 + wave * GPW + g;
    int4 cur = make_int4(0, 0, 0, 0);
    if (ei < e1) cur = ent[ei];
    int cur_idx[IPL];
    float cur_val[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
      cur_idx[i] = 0;
      cur_val[i] = 0.f;
      if (c < CL && c * IPL + i < cur.y) {
        cur_idx[i] = indices[(uint32_t)cur.z + c * IPL + i];
        cur_val[i] = vals[(uint32_t)cur.z + c * IPL + i];
      }
    }
    int cur_want = 1;                                // row filter of the current entry (prefetched)
    if constexpr (MASKED) {
      if (row_mask && ei < e1) cur_want = row_mask[cur.w];
    }
    for (int base = e0 + wave * GPW; base < e1; base += kGroups) {
      const bool live = ei < e1;
      const int ein = ei + kGroups;
      int4 nxt = make_int4(0, 0, 0, 0);
      if (ein < e1) nxt = ent[ein];
      const int slot = cur.x;
      int len = cur.y;
      const uint32_t begin = (uint32_t)cur.z;
      if constexpr (MASKED) {
        if (cur_want == 0) len = 0;                                    // output row not wanted
      }
      int nxt_want = 1;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && slot < kRMax) acc = s_acc[slot * LPR + c];           // segments start from zero
      int maxlen = len;
#pragma unroll
      for (int m = LPR; m < NR_WAVE; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m, NR_WAVE));
      maxlen = __builtin_amdgcn_readfirstlane(maxlen);
      int nxt_idx[IPL];
      float nxt_val[IPL];
#pragma unroll
      for (int i = 0; i < IPL; ++i) { nxt_idx[i] = 0; nxt_val[i] = 0.f; }
      int ch_idx[IPL];                               // the NEXT 16-entry chunk of this sub-list, requested while
      float ch_val[IPL];                             // the current chunk's gathers are in flight
#pragma unroll
      for (int i = 0; i < IPL; ++i) { ch_idx[i] = 0; ch_val[i] = 0.f; }
      for (int k0 = 0; k0 < maxlen || k0 == 0; k0 += 16) {
        int my_idx[IPL];
        float my_val[IPL];
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
          my_idx[i] = k0 == 0 ? cur_idx[i] : ch_idx[i];
          my_val[i] = k0 == 0 ? cur_val[i] : ch_val[i];
          if constexpr (MASKED) {
            if (col_mask && c < CL && k0 + c * IPL + i < len && col_mask[my_idx[i]] == 0)
              my_idx[i] = -1;                        // X row all zero
          }
        }
        const int nn = min(16, maxlen - k0);
        for (int t0 = 0; t0 < nn || (k0 == 0 && t0 == 0); t0 += kG) {
          float a[kG];
          float4 x[kG];
          bool on[kG];
#pragma unroll
          for (int u = 0; u < kG; ++u) {
            // entry t0+u of the chunk sits in lane (t0+u)/IPL of the group, component u % IPL
            // (t0 is a multiple of kG, kG of IPL)
            const int src = gbase | (((t0 + u) & 15) / IPL);
            const int col = __shfl(my_idx[u % IPL], src, NR_WAVE);
            a[u] = __shfl(my_val[u % IPL], src, NR_WAVE);
            on[u] = k0 + t0 + u < len;
            if constexpr (MASKED) {
              on[u] = on[u] && col >= 0;
              x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (on[u]) x[u] = X[(int64_t)col * LPR + c];
            } else {
              x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (t0 + u < nn) x[u] = X[(int64_t)max(col, 0) * LPR + c];   // wave-uniform guard
            }
          }
          if (t0 == 0 && k0 + 16 < maxlen) {         // a long sub-list: its next chunk (wave-uniform guard) — a chunk
#pragma unroll                                       // used to cost index round trip + gather round trips in sequence
            for (int i = 0; i < IPL; ++i) {
              ch_idx[i] = 0;
              ch_val[i] = 0.f;
              if (c < CL && k0 + 16 + c * IPL + i < len) {
                ch_idx[i] = indices[begin + k0 + 16 + c * IPL + i];
                ch_val[i] = vals[begin + k0 + 16 + c * IPL + i];
              }
            }
          }
          if (k0 == 0 && t0 == 0) {                  // prefetch for the next sub-list: first chunk, row filter
#pragma unroll
            for (int i = 0; i < IPL; ++i)
              if (c < CL && c * IPL + i < nxt.y) {
                nxt_idx[i] = indices[(uint32_t)nxt.z + c * IPL + i];
                nxt_val[i] = vals[(uint32_t)nxt.z + c * IPL + i];
              }
            if constexpr (MASKED) {
              if (row_mask && ein < e1) nxt_want = row_mask[nxt.w];
            }
          }
#pragma unroll
          for (int u = 0; u < kG; ++u)
            if (on[u]) {
              acc.x = __fadd_rn(acc.x, __fmul_rn(a[u], x[u].x));
              acc.y = __fadd_rn(acc.y, __fmul_rn(a[u], x[u].y));
              acc.z = __fadd_rn(acc.z, __fmul_rn(a[u], x[u].z));
              acc.w = __fadd_rn(acc.w, __fmul_rn(a[u], x[u].w));
            }
        }
      }
      if (live) s_acc[slot * LPR + c] = acc;
      cur = nxt;
#pragma unroll
      for (int i = 0; i < IPL; ++i) { cur_idx[i] = nxt_idx[i]; cur_val[i] = nxt_val[i]; }
      cur_want = nxt_want;
      ei = ein;
    }
    NR_BLK_WSTAMP();
    const int c0 = coff[k], c1 = coff[k + 1];
    if (c1 > c0) {                                  // workgroup-uniform
      __syncthreads();
      NR_BLK_STAMP(2);
      for (int ci = c0 + wave * GPW + g; ci < c1; ci += kGroups) {
        const int4 cm = cmb[ci];
        float4 acc = s_acc[cm.x * LPR + c];
        for (int s = 0; s < cm.z; ++s) {
          const float4 p = s_acc[(cm.y + s) * LPR + c];
          acc.x = __fadd_rn(acc.x, p.x); acc.y = __fadd_rn(acc.y, p.y);
          acc.z = __fadd_rn(acc.z, p.z); acc.w = __fadd_rn(acc.w, p.w);
        }
        s_acc[cm.x * LPR + c] = acc;
      }
    }
    __syncthreads();
  }
  NR_BLK_STAMP(3);
  blocked_epilogue<MASKED, ADAM, LPR>(s_acc, r0, nr, tid, kWaves * NR_WAVE, Y, addend, sum_in, sum_out,
                                      row_mask, ad, s_rows);
  NR_BLK_STAMP(4);
}


// ---------------------------------------------------------------------------------------------
// The masked hops of a LightGCN step (d = 64).  Column-masked (first backward hop): only columns of
// batch rows contribute.  In the general masked kernel above it costs 27-29 us: every lane group still walks
// all its sub-lists through the descriptor -> columns -> mask -> rows chain.  Here the workgroup's
// (column, value) slice is staged in LDS with one bulk read, the mask is applied there, every
// sub-list is compacted in place, and the walk only sees the survivors, several sub-lists in
// flight per lane group (no accumulators in LDS: a finished row goes straight to memory).
// Same sub-lists, same order of additions, same combine: bit-identical to spmm_blocked_kernel.
// (The mirror-image kernel for the row-masked forward hop — a wave per wanted sub-list — was built
// too and removed: a real batch is dominated by hub rows, whose gathers are ~45 % of a full pass,
// so that hop is bound by gather throughput, not by the walk: profiles/r01_exp_masked_hops.txt.)

struct LayerChain { const float4* a; const float4* b; };
__device__ __forceinline__ int4 zero_int4() { return make_int4(0, 0, 0, 0); }   // optional further terms of the running sum

__device__ __forceinline__ float4 chain_sum(float4 si, const LayerChain& ch, int64_t o) {
  if (ch.a) {
    const float4 t = ch.a[o];
    si = make_float4(__fadd_rn(si.x, t.x), __fadd_rn(si.y, t.y), __fadd_rn(si.z, t.z), __fadd_rn(si.w, t.w));
  }
  if (ch.b) {
    const float4 t = ch.b[o];
    si = make_float4(__fadd_rn(si.x, t.x), __fadd_rn(si.y, t.y), __fadd_rn(si.z, t.z), __fadd_rn(si.w, t.w));
  }
  return si;
}

__device__ __forceinline__ void masked_row_out(float4 y, int64_t o, const float4* __restrict__ addend,
                                               bool addend_row_nonzero, float4* __restrict__ Y,
                                               const float4* sum_in, float4* sum_out,
                                               const LayerChain* chain = nullptr) {
  if (addend) {
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);       // a row promised zero is not read
    if (addend_row_nonzero) av = addend[o];
    y.x = __fadd_rn(y.x, av.x); y.y = __fadd_rn(y.y, av.y);
    y.z = __fadd_rn(y.z, av.z); y.w = __fadd_rn(y.w, av.w);
  }
  if (Y) Y[o] = y;
  if (sum_out) {
    float4 si = sum_in[o];
    if (chain) si = chain_sum(si, *chain, o);
    sum_out[o] = make_float4(__fadd_rn(si.x, y.x), __fadd_rn(si.y, y.y), __fadd_rn(si.z, y.z),
                             __fadd_rn(si.w, y.w));
  }
}

// Walk of LDS-staged sub-lists by 16-lane groups (d = 64): rounds of kQ gathers per lane, three
// rounds in flight (the staged sub-lists are short, so depth comes from running several of them
// ahead); rounds retire in issue order.  s_ent[i] = {accumulator slot, length, offset into s_iv,
// row - r0 | bit 30: addend row may be non-zero | bit 29: row not wanted}.
// Every round issues exactly kQ + 1 loads, unconditionally (idle lanes and rounds read row 0 of X):
// only then can the compiler count the loads younger than the round it retires and wait with
// vmcnt(N > 0); with predicated loads it must assume none were issued and waits for all of them
// (vmcnt(0)), which silently turns the ring into one round in flight (measured: 14 us -> see
// profiles/r01_exp_masked_hops.txt).  The "+ 1" is the epilogue operand of the row a round
// completes (addend, else sum_in), requested with the round's gathers for the same reason.
__device__ __forceinline__ void staged_walk(const int4* s_ent, const int2* s_iv, int ne, int r0,
                                            const float4* __restrict__ X, float4* __restrict__ Y,
                                            const float4* __restrict__ addend, const float4* sum_in,
                                            float4* sum_out, float4* s_part, int kRMax, int wave,
                                            int g, int c, LayerChain chain = LayerChain{nullptr, nullptr}) {
  constexpr int RS = 16, LPR = 16, GPW = 4, kGroups = 64;
  constexpr int kQ = 4;
  struct Round { float4 x[kQ]; float4 pre; int off, n, slot, owner; bool first, last, live, valid; };
  const float4* pre_src = addend ? addend : (sum_in ? sum_in : X);   // workgroup-uniform, never null
  int ibase = wave * GPW, it0 = 0, imax = 0;      // issue cursor: batch of sub-lists, position
  int4 ids = make_int4(0, 0, 0, 0);
  bool ihas = ibase < ne;                         // wave-uniform: rounds left to issue
  auto open = [&]() {
    ids = make_int4(0, 0, 0, 0);
    if (ibase + g < ne) ids = s_ent[ibase + g];
    int m = ids.y;
#pragma unroll
    for (int sh = LPR; sh < NR_WAVE; sh <<= 1) m = max(m, __shfl_xor(m, sh, NR_WAVE));
    imax = __builtin_amdgcn_readfirstlane(m);
  };
  auto issue = [&](Round& r) {
    r.valid = ihas;
    r.off = ids.z + it0;
    r.n = ihas ? ids.y - it0 : 0;
    r.slot = ids.x;
    r.owner = ids.w;
    r.first = it0 == 0;
    r.live = ihas && ibase + g < ne;
    r.last = ihas && it0 + kQ >= imax;
#pragma unroll
    for (int u = 0; u < kQ; ++u) {
      int col = s_iv[u < r.n ? r.off + u : 0].x;
      if (u >= r.n) col = 0;
      r.x[u] = X[(int64_t)col * RS + c];
    }
    const bool want_pre = r.last && r.live && r.slot < kRMax && !((r.owner >> 29) & 1) &&
                          (!addend || ((r.owner >> 30) & 1));
    r.pre = pre_src[(want_pre ? ((int64_t)r0 + (r.owner & 0xFFFFFF)) * RS : 0) + c];
    if (r.last) {
      ibase += kGroups;
      it0 = 0;
      ihas = ibase < ne;
      if (ihas) open();
    } else if (ihas) {
      it0 += kQ;
    }
  };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto retire = [&](const Round& r) {
    if (r.first) acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float a[kQ];
#pragma unroll
    for (int u = 0; u < kQ; ++u) a[u] = __int_as_float(s_iv[u < r.n ? r.off + u : 0].y);
#pragma unroll
    for (int u = 0; u < kQ; ++u) {
      const float4 t = make_float4(__fadd_rn(acc.x, __fmul_rn(a[u], r.x[u].x)),
                                   __fadd_rn(acc.y, __fmul_rn(a[u], r.x[u].y)),
                                   __fadd_rn(acc.z, __fmul_rn(a[u], r.x[u].z)),
                                   __fadd_rn(acc.w, __fmul_rn(a[u], r.x[u].w)));
      if (u < r.n) acc = t;
    }
    if (r.last && r.live && !((r.owner >> 29) & 1)) {
      if (r.slot < kRMax) {
        const int64_t o = ((int64_t)r0 + (r.owner & 0xFFFFFF)) * RS + c;
        float4 y = acc;
        if (addend) {
          float4 av = make_float4(0.f, 0.f, 0.f, 0.f);     // a row promised zero was not read
          if ((r.owner >> 30) & 1) av = r.pre;
          y.x = __fadd_rn(y.x, av.x); y.y = __fadd_rn(y.y, av.y);
          y.z = __fadd_rn(y.z, av.z); y.w = __fadd_rn(y.w, av.w);
        }
        if (Y) Y[o] = y;
        if (sum_out) {
          float4 si = r.pre;
          if (addend) si = sum_in[o];                      // both operands: this one is read late
          si = chain_sum(si, chain, o);
          sum_out[o] = make_float4(__fadd_rn(si.x, y.x), __fadd_rn(si.y, y.y), __fadd_rn(si.z, y.z),
                                   __fadd_rn(si.w, y.w));
        }
      } else {
        s_part[(size_t)(r.slot - kRMax) * RS + c] = acc;
      }
    }
  };
  Round q0, q1, q2;                               // three rounds in flight: what 128 VGPRs hold
  if (ihas) open();
  issue(q0);
  issue(q1);
  while (true) {
    issue(q2);
    if (!q0.valid) break;
    retire(q0);
    issue(q0);
    if (!q1.valid) break;
    retire(q1);
    issue(q1);
    if (!q2.valid) break;
    retire(q2);
  }
}

template <bool COLMASK, bool ROWMASK>
__global__ __launch_bounds__(16 * NR_WAVE) void spmm_staged_masked_kernel(
    const int32_t* __restrict__ wg_row0, const int32_t* __restrict__ wg_ent_off,
    const int32_t* __restrict__ wg_cmb_off, const uint32_t* __restrict__ wg_nnz,
    const int4* __restrict__ ent, const int4* __restrict__ cmb,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float4* __restrict__ X, float4* __restrict__ Y, const float4* __restrict__ addend,
    const float4* sum_in, float4* sum_out, const uint8_t* __restrict__ col_mask,
    const uint8_t* __restrict__ row_mask, int addend_masked, int kRMax, int p_max, int ent_cap) {
  constexpr int RS = 16, LPR = 16, GPW = 4, kGroups = 64;
  extern __shared__ float4 s_mem[];
  float4* s_part = s_mem;                                          // [p_max][16]
  int4* s_ent = (int4*)(s_mem + (size_t)p_max * RS);               // [ent_cap]
  int2* s_iv = (int2*)(s_ent + ent_cap);                           // [non-zeros of the workgroup]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & (LPR - 1), g = lane / LPR;
  const int wg = blockIdx.x;
  const uint32_t nz0 = wg_nnz[2 * wg];
  const int nz = (int)wg_nnz[2 * wg + 1];
  const int e0 = wg_ent_off[2 * wg], ne = wg_ent_off[2 * wg + 1] - e0;
  const int c0 = wg_cmb_off[2 * wg], c1 = wg_cmb_off[2 * wg + 1];
  {
    constexpr int kStage = 8;                                      // loads in flight per lane while staging
    int4 e = make_int4(0, 0, 0, 0);
    if (tid < ne) e = ent[e0 + tid];                               // first 1024 descriptors ride along
    for (int i0 = 0; i0 < nz; i0 += kStage * 16 * NR_WAVE) {
      int col[kStage];
      float v[kStage];
      uint8_t keep[kStage];
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        const int i = i0 + k * 16 * NR_WAVE + tid;
        col[k] = 0;
        v[k] = 0.f;
        if (i < nz) {
          col[k] = indices[nz0 + i];
          v[k] = vals[nz0 + i];
        }
      }
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        keep[k] = 1;
        // (r02 A/B: replacing these per-pair mask bytes by a hash — no memory — takes the hop from 20.3 to
        // 18.6 us: an LDS bit set instead of the byte gathers could save at most ~1.5 us)
        if constexpr (COLMASK) keep[k] = col_mask[col[k]];
      }
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        const int i = i0 + k * 16 * NR_WAVE + tid;
        if (i < nz) s_iv[i] = make_int2(keep[k] ? col[k] : -1, __float_as_int(v[k]));
      }
    }
    for (int i = tid; i < ne; i += 16 * NR_WAVE) {
      if (i >= 16 * NR_WAVE) e = ent[e0 + i];
      e.z = (int)((uint32_t)e.z - nz0);                            // offset inside the staged slice
      // bit 30 of the owner: the addend row of this output row may be non-zero;
      // bit 29: the output row is not wanted (nothing is gathered, nothing is written)
      const int owner = e.w;                                       // the row itself (global id < 2^24)
      bool addend_on = true;
      if constexpr (COLMASK) addend_on = !addend_masked || col_mask[owner] != 0;
      if (addend_on) e.w |= 1 << 30;
      if constexpr (ROWMASK) {
        if (row_mask[owner] == 0) {
          e.y = 0;
          e.w |= 1 << 29;
        }
      }
      s_ent[i] = e;
    }
  }
  __syncthreads();
  // compact every sub-list to its surviving (column, value) pairs, in place and in order
  for (int base = wave * GPW; COLMASK && base < ne; base += kGroups) {
    const int ei = base + g;
    int4 ds = make_int4(0, 0, 0, 0);
    if (ei < ne) ds = s_ent[ei];
    int maxlen = ds.y;
#pragma unroll
    for (int sh = LPR; sh < NR_WAVE; sh <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, sh, NR_WAVE));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    int w = 0;
    for (int k0 = 0; k0 < maxlen; k0 += LPR) {
      const bool in = k0 + c < ds.y;
      int2 iv = make_int2(-1, 0);
      if (in) iv = s_iv[ds.z + k0 + c];
      const bool keep = iv.x >= 0;
      const unsigned long long m = __ballot(keep);
      const uint32_t gm = (uint32_t)(m >> (g * LPR)) & 0xFFFFu;
      if (keep) s_iv[ds.z + w + __popc(gm & ((1u << c) - 1u))] = iv;
      w += __popc(gm);
    }
    if (ei < ne && c == 0) s_ent[ei].y = w;
  }
  staged_walk(s_ent, s_iv, ne, 0, X, Y, addend, sum_in, sum_out, s_part, kRMax, wave, g, c);
  if (c1 > c0) {                                           // workgroup-uniform
    __syncthreads();
    for (int ci = c0 + wave * GPW + g; ci < c1; ci += kGroups) {
      const int4 cm = cmb[ci];
      if constexpr (ROWMASK) {
        if (row_mask[cm.w] == 0) continue;
      }
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sgm = 0; sgm < cm.z; ++sgm) {
        const float4 q = s_part[(size_t)(cm.y - kRMax + sgm) * RS + c];
        acc.x = __fadd_rn(acc.x, q.x); acc.y = __fadd_rn(acc.y, q.y);
        acc.z = __fadd_rn(acc.z, q.z); acc.w = __fadd_rn(acc.w, q.w);
      }
      bool addend_on = true;
      if constexpr (COLMASK) addend_on = !addend_masked || col_mask[cm.w] != 0;
      masked_row_out(acc, (int64_t)cm.w * RS + c, addend, addend_on, Y, sum_in, sum_out);
    }
  }
}

struct BatchLists {        // a step's triplets; rows = users | n_users + pos | n_users + neg
  const int32_t* users; const int32_t* pos; const int32_t* neg;
  int batch, n_users;
  uint8_t* row_flag;       // out: 1 on the batch rows (must be zero elsewhere on entry)
  int32_t* rows_out;       // out (optional): the 3*batch rows, list order users, pos, neg
};

// Row-masked hop (last forward hop: only the batch rows are produced).  A real batch always holds
// the hub rows (positives are drawn in proportion to degree): ~45 % of all non-zeros belong to
// wanted rows, and in the row-run schedule they sit in a few workgroups that then do a full
// pass's work while the rest idle.  This kernel runs on a second schedule of the same sub-lists,
// dealt to the workgroups by descending row length (descriptors carry global rows): the
// workgroup collects its wanted sub-lists, stages their (column, value) pairs in LDS (a wave per
// sub-list, in chunks if they outgrow the buffer) and walks them with staged_walk.
__global__ __launch_bounds__(16 * NR_WAVE) void spmm_wanted_rows_kernel(
    const int32_t* __restrict__ w_ent_off, const int32_t* __restrict__ w_cmb_off,
    const int4* __restrict__ w_ent, const int4* __restrict__ w_cmb,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float4* __restrict__ X, float4* __restrict__ Y, const float4* __restrict__ addend,
    const float4* sum_in, float4* sum_out, const uint8_t* __restrict__ row_mask, int kRMax,
    int p_max, int ent_cap, int nnz_cap, LayerChain chain, BatchLists bl, int bitmap_words) {
  constexpr int RS = 16;
  extern __shared__ float4 s_mem[];
  float4* s_part = s_mem;                                          // [p_max][16]
  int4* s_now = (int4*)(s_mem + (size_t)p_max * RS);               // sub-lists of this chunk
  int4* s_later = s_now + ent_cap;                                 // sub-lists that did not fit
  uint32_t* s_bits = (uint32_t*)(s_later + ent_cap);               // [bitmap_words] wanted rows (batch form)
  int2* s_iv = (int2*)(s_bits + bitmap_words);                     // [nnz_cap]
  __shared__ int s_n_now, s_n_later, s_cursor;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int wg = blockIdx.x;
  const int e0 = w_ent_off[2 * wg], ne = w_ent_off[2 * wg + 1] - e0;
  const int c0 = w_cmb_off[2 * wg], c1 = w_cmb_off[2 * wg + 1];
  if (tid == 0) { s_n_now = 0; s_n_later = 0; s_cursor = 0; }
  const bool by_batch = bl.users != nullptr;                       // workgroup-uniform
  if (by_batch) {
    // the batch itself says which rows are wanted: every workgroup builds the bit set in LDS
    // (no mark_batch launch before this kernel, no flag gather per descriptor); the workgroups
    // share the work of publishing the flags / row list the later kernels of the step read
    for (int i = tid; i < bitmap_words; i += 16 * NR_WAVE) s_bits[i] = 0u;
    __syncthreads();
    for (int i = tid; i < 3 * bl.batch; i += 16 * NR_WAVE) {
      const int which = i / bl.batch, b = i - which * bl.batch;
      const int row = which == 0 ? bl.users[b] : bl.n_users + (which == 1 ? bl.pos[b] : bl.neg[b]);
      atomicOr(&s_bits[row >> 5], 1u << (row & 31));
      if (i % (int)gridDim.x == wg) {
        bl.row_flag[row] = 1;
        if (bl.rows_out) bl.rows_out[i] = row;
      }
    }
  }
  __syncthreads();
  auto is_wanted = [&](int row) -> bool {
    return by_batch ? ((s_bits[row >> 5] >> (row & 31)) & 1u) != 0u : row_mask[row] != 0;
  };
  // wanted sub-lists of this workgroup, each with room reserved in the staging buffer (.x keeps
  // the slot, .z becomes the staging offset; the first non-zero moves to a register copy)
  for (int i = tid; i < ne; i += 16 * NR_WAVE) {
    const int4 e = w_ent[e0 + i];
    if (is_wanted(e.w)) {
      const int off = atomicAdd(&s_cursor, e.y);
      if (off + e.y <= nnz_cap) s_now[atomicAdd(&s_n_now, 1)] = make_int4(e.x, e.y | (off << 8), e.z, e.w);
      else s_later[atomicAdd(&s_n_later, 1)] = e;
    }
  }
  __syncthreads();
  while (true) {
    const int n_now = s_n_now, n_later = s_n_later;
    // stage: a wave per sub-list (<= 64 pairs, one coalesced read each); .y = length | offset << 8
    for (int k = wave; k < n_now; k += 16) {
      const int4 e = s_now[k];
      const int len = e.y & 0xFF, off = e.y >> 8;
      if (lane < len)
        s_iv[off + lane] = make_int2(indices[(uint32_t)e.z + lane], __float_as_int(vals[(uint32_t)e.z + lane]));
    }
    __syncthreads();
    for (int k = tid; k < n_now; k += 16 * NR_WAVE) {              // descriptors in staged_walk's format
      const int4 e = s_now[k];
      s_now[k] = make_int4(e.x, e.y & 0xFF, e.y >> 8, e.w | (1 << 30));
    }
    __syncthreads();
    staged_walk(s_now, s_iv, n_now, 0, X, Y, addend, sum_in, sum_out, s_part, kRMax, wave, g, c, chain);
    if (n_later == 0) break;                                       // workgroup-uniform
    __syncthreads();
    if (tid == 0) { s_n_now = 0; s_n_later = 0; s_cursor = 0; }
    __syncthreads();
    // next chunk: re-deal the deferred sub-lists (s_later -> s_now, overflow back into s_later's
    // already consumed prefix: entry k is read before any slot <= k is written)
    for (int k0 = 0; k0 < n_later; k0 += 16 * NR_WAVE) {
      const int k = k0 + tid;
      int4 e = make_int4(0, 0, 0, 0);
      if (k < n_later) e = s_later[k];
      __syncthreads();
      if (k < n_later) {
        const int off = atomicAdd(&s_cursor, e.y);
        if (off + e.y <= nnz_cap) s_now[atomicAdd(&s_n_now, 1)] = make_int4(e.x, e.y | (off << 8), e.z, e.w);
        else s_later[atomicAdd(&s_n_later, 1)] = e;
      }
      __syncthreads();
    }
  }
  if (c1 > c0) {                                                   // workgroup-uniform
    __syncthreads();
    for (int ci = c0 + wave * 4 + g; ci < c1; ci += 64) {
      const int4 cm = w_cmb[ci];
      if (!is_wanted(cm.x)) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sgm = 0; sgm < cm.z; ++sgm) {
        const float4 q = s_part[(size_t)(cm.y - kRMax + sgm) * RS + c];
        acc.x = __fadd_rn(acc.x, q.x); acc.y = __fadd_rn(acc.y, q.y);
        acc.z = __fadd_rn(acc.z, q.z); acc.w = __fadd_rn(acc.w, q.w);
      }
      masked_row_out(acc, (int64_t)cm.x * RS + c, addend, true, Y, sum_in, sum_out, &chain);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-masked hop, wave-cooperative form.  Two things bound the staged walk on a real batch (which
// always holds the hub rows: positives are drawn by degree):
//  (1) a sub-list given to ONE 16-lane group is up to sixteen dependent rounds of four gathers —
//      here a sub-list belongs to a whole wave: lane L holds pair L, lane group g gathers the rows
//      of pairs 16g..16g+15, so all 64 gathers of a segment are in flight at once, and the sums
//      still run in the strict order used everywhere else: group 0 adds its 16 products, hands
//      the running sum to group 1, ... (products and sums rounded separately: bit-identical);
//  (2) a hub's 3,077 x 256 B = 0.8 MB of gathers through ONE CU's vector-memory path (64 B/clk)
//      is >= 5 us whatever the waves do — here a row of > 64 non-zeros is cut into chunks of <= 8
//      consecutive 64-segments and every chunk is dealt on its own, so the biggest hub runs on 7 CUs.
//      Segment sums go to a global buffer (agent-scope stores); a chunk ends with ONE counter update
//      per row (not one per segment: 1,250 same-address atomics cost more than they saved,
//      profiles/r02_exp_wanted_hop.txt), and the chunk that finishes a row last adds ALL its segment
//      sums one by one in segment order — the same association as the full pass.
struct WantedEpi {
  float4* Y; const float4* addend; const float4* sum_in; float4* sum_out; LayerChain chain;
};
#ifdef NR_WW_TIMELINE      // experiment builds only (scripts/exp_wanted_timeline.sh): per-workgroup phase stamps
__device__ unsigned long long g_ww_dbg[256 * 8];
#define NR_WW_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_ww_dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define NR_WW_STAMP_END(i) do { __syncthreads(); NR_WW_STAMP(i); } while (0)
#else
#define NR_WW_STAMP(i)
#define NR_WW_STAMP_END(i)
#endif

__global__ __launch_bounds__(16 * NR_WAVE) void spmm_wanted_wave_kernel(
    const int32_t* __restrict__ ww_off, const int32_t* __restrict__ ww_choff,
    const int4* __restrict__ ww_ent, const int32_t* __restrict__ ww_gch, const int4* __restrict__ ww_hub,
    const int32_t* __restrict__ ww_lcoff, const int4* __restrict__ ww_lcmb,
    float* ww_part, unsigned* ww_cnt, const int32_t* __restrict__ indices,
    const float* __restrict__ vals, const float4* __restrict__ X, WantedEpi ep,
    const uint8_t* __restrict__ row_mask, BatchLists bl, int bitmap_words, int ent_cap, int lds_slots) {
  constexpr int RS = 16;
  extern __shared__ float4 s_mem[];
  float4* s_part = s_mem;                                          // [lds_slots][16] segment sums of one-chunk rows
  int4* s_want = (int4*)(s_mem + (size_t)lds_slots * RS);          // [ent_cap] wanted sub-lists
  uint32_t* s_bits = (uint32_t*)(s_want + ent_cap);                // [bitmap_words] wanted rows (batch form)
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int wg = blockIdx.x;
  const int e0 = ww_off[wg], ne = ww_off[wg + 1] - e0;
  const int h0 = ww_choff[wg], h1 = ww_choff[wg + 1];
  const int l0 = ww_lcoff[wg], l1 = ww_lcoff[wg + 1];
  NR_WW_STAMP(0);
  if (tid == 0) s_n = 0;
  const bool by_batch = bl.users != nullptr;                       // workgroup-uniform
  if (by_batch) {
    for (int i = tid; i < bitmap_words; i += 16 * NR_WAVE) s_bits[i] = 0u;
    __syncthreads();
    // every workgroup reads the whole batch (an 8,192-triplet global batch: 24 ids per thread).  A triplet's three ids
    // are requested together, unconditionally inside the loop (r06: three lists one after the other, each id behind its
    // own `i < batch` test, were waited for one by one — 3 round trips at B = 1,024, 24 at 8,192)
    for (int i = tid; i < bl.batch; i += 16 * NR_WAVE) {
      const int u = bl.users[i], p = bl.pos[i], q = bl.neg[i];
      const int ru = u, rp = bl.n_users + p, rq = bl.n_users + q;
      if (u >= 0) atomicOr(&s_bits[ru >> 5], 1u << (ru & 31));
      if (p >= 0) atomicOr(&s_bits[rp >> 5], 1u << (rp & 31));
      if (q >= 0) atomicOr(&s_bits[rq >> 5], 1u << (rq & 31));
    }
    // publishing is shared out: a contiguous piece of the 3 * batch list positions per workgroup
    const int total = 3 * bl.batch, per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int i = wg * per + tid; i < min(total, (wg + 1) * per); i += 16 * NR_WAVE) {
      const int which = i / bl.batch, b = i - which * bl.batch;
      const int row = which == 0 ? bl.users[b] : bl.n_users + (which == 1 ? bl.pos[b] : bl.neg[b]);
      bl.row_flag[row] = 1;
      if (bl.rows_out) bl.rows_out[i] = row;
    }
  }
  __syncthreads();
  auto is_wanted = [&](int row) -> bool {
    return by_batch ? ((s_bits[row >> 5] >> (row & 31)) & 1u) != 0u : row_mask[row] != 0;
  };
  NR_WW_STAMP(1);
  for (int i = tid; i < ne; i += 16 * NR_WAVE) {
    const int4 e = ww_ent[e0 + i];
    if (is_wanted(e.w)) s_want[atomicAdd(&s_n, 1)] = e;
  }
  __syncthreads();
  NR_WW_STAMP(2);
  const int n = s_n;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // this wave's sub-lists: k = wave, wave + 16, ...; the next one's pairs are requested while the
  // current one's gathers are in flight
  int k = wave;
  int4 e = zero_int4();
  int col = 0;
  float val = 0.f;
  if (k < n) {
    e = s_want[k];
    if (lane < e.y) {
      col = indices[(uint32_t)e.z + lane];
      val = vals[(uint32_t)e.z + lane];
    }
  }
  while (k < n) {
    const int len = e.y, slot = e.x, row = e.w;
    float4 x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cj = __shfl(col, (g << 4) | j, NR_WAVE);           // lanes past the end hold column 0
      x[j] = X[(int64_t)cj * RS + c];
    }
    // the row's epilogue operands ride with the gathers (requested after the sum they would be one
    // more memory round trip per sub-list); rows that end in a partial slot read element 0
    const int64_t o = (int64_t)row * RS + c;
    const int64_t oe = slot == 0 ? o : (int64_t)c;
    float4 pre_si = zero, pre_a = zero, pre_b = zero;               // (an addend is read late: 128 VGPRs)
    if (ep.sum_out) {
      pre_si = ep.sum_in[oe];
      if (ep.chain.a) pre_a = ep.chain.a[oe];
      if (ep.chain.b) pre_b = ep.chain.b[oe];
    }
    const int kn = k + 16;
    int4 en = zero_int4();
    int coln = 0;
    float valn = 0.f;
    if (kn < n) {
      en = s_want[kn];
      if (lane < en.y) {
        coln = indices[(uint32_t)en.z + lane];
        valn = vals[(uint32_t)en.z + lane];
      }
    }
    // products first (pair j of this lane group: value broadcast from lane 16g + j), in place
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float aj = __shfl(val, (g << 4) | j, NR_WAVE);
      x[j] = make_float4(__fmul_rn(aj, x[j].x), __fmul_rn(aj, x[j].y), __fmul_rn(aj, x[j].z), __fmul_rn(aj, x[j].w));
    }
    float4 acc = zero, carry = zero;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      if (gg * 16 < len) {                                         // wave-uniform
        if (g == gg) {
          acc = carry;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float4 t = make_float4(__fadd_rn(acc.x, x[j].x), __fadd_rn(acc.y, x[j].y),
                                         __fadd_rn(acc.z, x[j].z), __fadd_rn(acc.w, x[j].w));
            if (gg * 16 + j < len) acc = t;
          }
        }
        const int src = (gg << 4) | c;
        carry = make_float4(__shfl(acc.x, src, NR_WAVE), __shfl(acc.y, src, NR_WAVE),
                            __shfl(acc.z, src, NR_WAVE), __shfl(acc.w, src, NR_WAVE));
      }
    }
    if (g == 0) {
      if (slot == 0) {                                             // masked_row_out on the prefetched operands
        float4 y = carry;
        if (ep.addend) {
          const float4 pre_add = ep.addend[o];
          y = make_float4(__fadd_rn(y.x, pre_add.x), __fadd_rn(y.y, pre_add.y), __fadd_rn(y.z, pre_add.z),
                          __fadd_rn(y.w, pre_add.w));
        }
        if (ep.Y) ep.Y[o] = y;
        if (ep.sum_out) {
          float4 si = pre_si;
          if (ep.chain.a)
            si = make_float4(__fadd_rn(si.x, pre_a.x), __fadd_rn(si.y, pre_a.y), __fadd_rn(si.z, pre_a.z),
                             __fadd_rn(si.w, pre_a.w));
          if (ep.chain.b)
            si = make_float4(__fadd_rn(si.x, pre_b.x), __fadd_rn(si.y, pre_b.y), __fadd_rn(si.z, pre_b.z),
                             __fadd_rn(si.w, pre_b.w));
          ep.sum_out[o] = make_float4(__fadd_rn(si.x, y.x), __fadd_rn(si.y, y.y), __fadd_rn(si.z, y.z),
                                      __fadd_rn(si.w, y.w));
        }
      } else if (slot < 0) {                                       // a segment of a one-chunk row: LDS
        s_part[(size_t)(-slot - 1) * RS + c] = carry;
      } else {                                                     // a segment of a multi-chunk row: global,
        float* mine = ww_part + (size_t)(slot - 1) * 64 + c * 4;   // agent scope (written through)
        __hip_atomic_store(mine + 0, carry.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, carry.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2, carry.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 3, carry.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    k = kn;
    e = en;
    col = coln;
    val = valn;
  }
  if (l1 > l0) {                                                   // workgroup-uniform: rows whose segments all ran here
    __syncthreads();
    for (int li = l0 + wave * 4 + g; li < l1; li += 64) {
      const int4 cm = ww_lcmb[li];                                 // {row, first LDS slot, segments}
      if (!is_wanted(cm.x)) continue;
      float4 acc = zero;
      for (int sgm = 0; sgm < cm.z; ++sgm) {
        const float4 q = s_part[(size_t)(cm.y + sgm) * RS + c];
        acc.x = __fadd_rn(acc.x, q.x); acc.y = __fadd_rn(acc.y, q.y);
        acc.z = __fadd_rn(acc.z, q.z); acc.w = __fadd_rn(acc.w, q.w);
      }
      masked_row_out(acc, (int64_t)cm.x * RS + c, ep.addend, true, ep.Y, ep.sum_in, ep.sum_out, &ep.chain);
    }
  }
  NR_WW_STAMP_END(3);
  if (tid == 0 && blockIdx.x < 256) {
#ifdef NR_WW_TIMELINE
    g_ww_dbg[blockIdx.x * 8 + 6] = (unsigned long long)n;
    g_ww_dbg[blockIdx.x * 8 + 7] = (unsigned long long)(h1 - h0);
#endif
  }
  if (h1 > h0) {                                                   // workgroup-uniform: this workgroup holds chunks
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                                 // segment sums have left before a count moves
    __syncthreads();
    for (int hi = h0 + wave; hi < h1; hi += 16) {
      const int hub = ww_gch[hi];
      const int4 hd = ww_hub[hub];                                 // {row, first partial slot, segments, chunks}
      if (!is_wanted(hd.x)) continue;
      unsigned old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(&ww_cnt[hub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old != (unsigned)hd.w - 1u) continue;                    // another chunk of the row is still out
      // all segment sums of the row, 16 at a time in flight, added in segment order
      float4 sum = zero;
      for (int s0 = 0; s0 < hd.z; s0 += 16) {
        float4 q[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float* qp = ww_part + ((size_t)hd.y + min(s0 + j, hd.z - 1)) * 64 + c * 4;
          q[j].x = __hip_atomic_load(qp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q[j].y = __hip_atomic_load(qp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q[j].z = __hip_atomic_load(qp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q[j].w = __hip_atomic_load(qp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (s0 + j < hd.z)
            sum = make_float4(__fadd_rn(sum.x, q[j].x), __fadd_rn(sum.y, q[j].y), __fadd_rn(sum.z, q[j].z),
                              __fadd_rn(sum.w, q[j].w));
      }
      if (g == 0)
        masked_row_out(sum, (int64_t)hd.x * RS + c, ep.addend, true, ep.Y, ep.sum_in, ep.sum_out, &ep.chain);
      if (lane == 0) __hip_atomic_store(&ww_cnt[hub], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    }
  }
  NR_WW_STAMP_END(4);
}

size_t wanted_lds_bytes(const BlockedPlan* p) {
  return (size_t)p->p_max * 256 + 2 * (size_t)p->w_ent_cap * 16 + (size_t)p->w_bitmap_words * 4 +
         (size_t)p->w_nnz_cap * 8;
}
int ww_bitmap_words(const BlockedPlan* p) { return (int)((p->n_rows + 127) / 128 * 4); }
size_t ww_lds_bytes(const BlockedPlan* p) {
  return (size_t)p->ww_lds_slots * 256 + (size_t)p->ww_ent_cap * 16 + (size_t)ww_bitmap_words(p) * 4 + 16;
}
size_t colmask_lds_bytes(const BlockedPlan* p) {
  return (size_t)p->p_max * 256 + (size_t)p->ent_cap * 16 + (size_t)p->nnz_cap * 8;
}

// launch of the wave-cooperative wanted-rows kernel (all three entrances of the row-masked hop)
int launch_wanted_wave(const BlockedPlan* p, const int32_t* d_indices, const float* d_vals, const float* d_X,
                       float* d_Y, const float* d_addend, const float* d_sum_in, float* d_sum_out,
                       const uint8_t* d_y_row_wanted, LayerChain chain, BatchLists bl, hipStream_t st) {
  hipLaunchKernelGGL(spmm_wanted_wave_kernel, dim3((unsigned)p->n_wg), dim3(16 * NR_WAVE), ww_lds_bytes(p), st,
                     p->ww_off, p->ww_choff, p->ww_ent, p->ww_gch, p->ww_hub, p->ww_lcoff, p->ww_lcmb, p->ww_part,
                     p->ww_cnt, d_indices, d_vals, (const float4*)d_X,
                     WantedEpi{(float4*)d_Y, (const float4*)d_addend, (const float4*)d_sum_in,
                               (float4*)d_sum_out, chain},
                     d_y_row_wanted, bl, ww_bitmap_words(p), p->ww_ent_cap, p->ww_lds_slots);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

struct HostEnt { int32_t slot, len; uint32_t begin; int32_t owner; };
int s_gathers_in_flight = 8;     // tuning knob (nrhip_spmm_blocked_tune)

// (column, value) pairs copied into the plan's row order: row k of the order is CSR positions
// [src[k], src[k] + dst[k+1] - dst[k]) -> packed positions [dst[k], dst[k+1]).  Once per matrix.
__global__ __launch_bounds__(256) void row_order_pack_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst,
                                                             int64_t n_rows, const int32_t* __restrict__ indices,
                                                             const float* __restrict__ vals, int32_t* __restrict__ pk_idx,
                                                             float* __restrict__ pk_val) {
  const int c = threadIdx.x & 15;
  for (int64_t k = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); k < n_rows; k += (int64_t)gridDim.x * 16) {
    const uint32_t s0 = src[k], d0 = dst[k], len = dst[k + 1] - d0;
    for (uint32_t j = c; j < len; j += 16) {
      pk_idx[d0 + j] = indices[s0 + j];
      pk_val[d0 + j] = vals[s0 + j];
    }
  }
}

}  // namespace

extern "C" {

int nrhip_spmm_blocked_plan_bytes(int64_t n_rows, int64_t nnz, int d, size_t* bytes) {
  NR_REQUIRE(bytes && n_rows >= 0 && nnz >= 0, NR_ERR_ARG, "spmm_blocked_plan_bytes: bad arguments");
  (void)d;
  *bytes = blocked_plan_bytes(n_rows, nnz);
  return NR_OK;
}

int nrhip_spmm_blocked_plan_create(const int64_t* h_indptr, const int32_t* h_indices,
                                   int64_t n_rows, int64_t split_row, int d, int64_t block_bytes,
                                   int n_workgroups, int waves_per_wg, int seg_len, int r_max,
                                   int p_max, void* d_plan_buf, size_t plan_bytes, void* stream,
                                   void** plan_out) {
  const int kWaves = waves_per_wg > 0 ? waves_per_wg : 16;
  const int kSeg = seg_len > 0 ? seg_len : kSegDefault;
  const int kD = d;
  const int kRMax = r_max > 0 ? r_max : kRMaxDefault * kWaves / 16 * 64 / (d > 0 ? d : 64);
  const int kPMax = p_max > 0 ? p_max : kPMaxDefault * kWaves / 16 * 64 / (d > 0 ? d : 64);
  NR_REQUIRE(d == 16 || d == 32 || d == 64 || d == 128 || d == 256, NR_ERR_UNSUPPORTED,
             "spmm_blocked: embedding dim %d not built (16, 32, 64, 128, 256)", d);
  NR_REQUIRE(kWaves == 16 || kWaves == 8, NR_ERR_UNSUPPORTED, "spmm_blocked: waves per workgroup %d (8, 16)", kWaves);
  NR_REQUIRE(kSeg >= 16 && (size_t)(kRMax + kPMax) * kD * 4 + (size_t)kRMax * 4 <= (size_t)kMaxLdsBytes, NR_ERR_UNSUPPORTED,
             "spmm_blocked: seg %d / accumulators %d+%d do not fit", kSeg, kRMax, kPMax);
  NR_REQUIRE(h_indptr && h_indices && d_plan_buf && plan_out && n_rows > 0, NR_ERR_ARG,
             "spmm_blocked_plan_create: bad arguments");
  const int64_t nnz = h_indptr[n_rows] - h_indptr[0];
  NR_REQUIRE(h_indptr[0] == 0 && nnz < ((int64_t)1 << 32), NR_ERR_UNSUPPORTED,
             "spmm_blocked: indptr must start at 0 and hold < 2^32 non-zeros");
  const size_t need_bytes = blocked_plan_bytes(n_rows, nnz);
  NR_REQUIRE(plan_bytes >= need_bytes, NR_ERR_WORKSPACE,
             "spmm_blocked_plan_create: plan buffer %zu < %zu bytes", plan_bytes, need_bytes);
  if (block_bytes <= 0) block_bytes = (int64_t)1 << 40;     // measured: phases cost more than they save
  if (split_row <= 0 || split_row >= n_rows) split_row = 0;
  int n_wg = n_workgroups;
  if (n_wg <= 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    NR_CHECK_HIP(hipGetDevice(&dev));
    NR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    n_wg = prop.multiProcessorCount * (16 / kWaves);
  }
  n_wg = n_wg / 8 * 8;
  if (n_workgroups <= 0 && n_wg >= 8) {
    // more rows than one workgroup per CU can hold accumulators for: launch a multiple of the CU
    // count (the extra workgroups queue behind the resident ones; without column blocking the
    // rounds are independent)
    const int64_t per_class = split_row ? n_wg / 2 : n_wg;
    const int64_t biggest = split_row ? std::max<int64_t>(split_row, n_rows - split_row) : n_rows;
    const int64_t cap = (int64_t)kRMax * 9 / 10;
    const int64_t mult = (biggest + per_class * cap - 1) / (per_class * cap);
    if (mult > 1) n_wg = (int)std::min<int64_t>((int64_t)n_wg * mult, 4096 + n_rows / 32) / 8 * 8;
  }
  NR_REQUIRE(n_wg >= 8 && n_wg <= 4096 + n_rows / 32, NR_ERR_UNSUPPORTED, "spmm_blocked: %d workgroups",
             n_wg);

  // cost of a row for balancing: its non-zeros plus a fixed cost per sub-list (descriptor, first
  // index chunk and the short last gather round; fitted on the per-workgroup timeline,
  // profiles/r01_exp_spmm_timeline.txt)
  const int ent_cost = 4;     // r05 (rows dealt): 0 .. 16 all within 1 us of each other, 2-4 best (profiles/r05_exp_entcost.txt)
  auto row_cost = [&](int64_t l) { return l + (int64_t)ent_cost * std::max<int64_t>(1, (l + kSeg - 1) / kSeg); };
  struct ClassDesc { int64_t ra, rb; std::vector<int> wgs; int32_t cmin; int64_t width, K; };
  std::vector<ClassDesc> classes;
  if (split_row) {
    ClassDesc a{0, split_row, {}, 0, 1, 1}, b{split_row, n_rows, {}, 0, 1, 1};
    // workgroups per class in proportion to the class's cost; class A fills XCDs 0.. first
    // (workgroup w runs on XCD w % 8), so at most one XCD serves both halves of the table
    int64_t cost_a = 0, cost_b = 0;
    for (int64_t r = 0; r < n_rows; ++r) (r < split_row ? cost_a : cost_b) += row_cost(h_indptr[r + 1] - h_indptr[r]);
    int n_a = (int)((double)n_wg * (double)cost_a / (double)std::max<int64_t>(cost_a + cost_b, 1) + 0.5);
    n_a = std::min(std::max(n_a, 1), n_wg - 1);
    {
      // every class must still fit its rows into its workgroups' accumulators
      const int64_t cap = (int64_t)kRMax * 9 / 10;
      const int need_a = (int)((split_row + cap - 1) / cap), need_b = (int)((n_rows - split_row + cap - 1) / cap);
      n_a = std::max(n_a, std::min(need_a, n_wg - 1));
      n_a = std::min(n_a, std::max(n_wg - need_b, 1));
    }
    std::vector<int> order;                    // workgroup ids, XCD-major
    for (int x = 0; x < 8; ++x)
      for (int w = x; w < n_wg; w += 8) order.push_back(w);
    for (int i = 0; i < n_wg; ++i) (i < n_a ? a : b).wgs.push_back(order[i]);
    std::sort(a.wgs.begin(), a.wgs.end());
    std::sort(b.wgs.begin(), b.wgs.end());
    classes.push_back(a);
    classes.push_back(b);
  } else {
    ClassDesc a{0, n_rows, {}, 0, 1, 1};
    for (int w = 0; w < n_wg; ++w) a.wgs.push_back(w);
    classes.push_back(a);
  }
  std::vector<int32_t> wg_row0(n_wg, 0), wg_nrows(n_wg, 0);
  std::vector<std::vector<std::vector<HostEnt>>> wg_ent(n_wg);     // [wg][phase][entries]
  std::vector<std::vector<std::vector<int4>>> wg_cmb(n_wg);
  int n_phases = 1;
  std::vector<std::vector<int32_t>> wg_rows((size_t)n_wg);         // [wg] its rows, slot order
  for (ClassDesc& cl : classes) {
    const int64_t cb = h_indptr[cl.ra], ce = h_indptr[cl.rb];
    int32_t cmin = INT32_MAX, cmax = -1;
    for (int64_t t = cb; t < ce; ++t) {
      cmin = std::min(cmin, h_indices[t]);
      cmax = std::max(cmax, h_indices[t]);
    }
    if (cmax < cmin) { cmin = 0; cmax = 0; }
    const int64_t span = (int64_t)cmax + 1 - cmin;
    int64_t K = (span * kD * 4 + block_bytes - 1) / block_bytes;
    K = std::max<int64_t>(K, 1);
    NR_REQUIRE(K <= kMaxPhases, NR_ERR_UNSUPPORTED,
               "spmm_blocked: gathered table of %lld rows needs %lld column blocks (max %d) — "
               "use the work-item kernel", (long long)span, (long long)K, kMaxPhases);
    const int64_t width = (span + K - 1) / K;
    n_phases = std::max<int>(n_phases, (int)K);
    cl.cmin = cmin; cl.width = width; cl.K = K;
    // Which rows a workgroup owns.  r01-r04: contiguous runs balanced by cost — fine while a row's length is
    // independent of its id (the first synthetic twin shuffled the popularity ranks), but in real interaction data
    // (and in the r05 twin, whose item popularity follows the real test split) popular items cluster in id: the
    // run of hub rows then holds few rows and the runs of tail rows hit the accumulator cap (kRMax rows of ~8
    // non-zeros = 0.7 of the cost target), which pushes the excess onto the other runs — the slowest workgroup of
    // the item class carried 1.9x the mean cost and the pass took 50 us instead of 34.  r05: rows are DEALT —
    // sorted by cost, each to the least-loaded workgroup that still has an accumulator (LPT) — so every
    // workgroup gets the same cost whatever the numbering; the plan lists a workgroup's rows (row_of) and owns
    // the (column, value) pairs in that order.  (The r01-r04 contiguous runs left the product in r06:
    // profiles/r05_exp_entcost.txt has the A/B.)
    const int64_t n_cl = cl.rb - cl.ra;
    const int64_t nw = (int64_t)cl.wgs.size();
    std::vector<std::vector<int32_t>> lists((size_t)nw);
    {
      NR_REQUIRE(n_cl <= nw * (int64_t)kRMax, NR_ERR_UNSUPPORTED,
                 "spmm_blocked: %lld rows do not fit %zu workgroups x %d accumulators — use the work-item kernel",
                 (long long)n_cl, cl.wgs.size(), kRMax);
      std::vector<int32_t> by_cost((size_t)n_cl);
      for (int64_t q = 0; q < n_cl; ++q) by_cost[(size_t)q] = (int32_t)(cl.ra + q);
      std::stable_sort(by_cost.begin(), by_cost.end(), [&](int32_t x, int32_t y) {
        return h_indptr[x + 1] - h_indptr[x] > h_indptr[y + 1] - h_indptr[y];
      });
      // min-heap of (cost so far, workgroup); a workgroup whose accumulators are all taken leaves the heap
      typedef std::pair<int64_t, int> Load;
      std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
      for (int i = 0; i < (int)nw; ++i) heap.push(Load(0, i));
      // rows that cannot be placed freely any more (as many rows left as free accumulators) are not an issue:
      // every workgroup in the heap has a free accumulator and n_cl <= nw * kRMax
      for (int32_t row : by_cost) {
        Load top = heap.top();
        heap.pop();
        lists[(size_t)top.second].push_back(row);
        top.first += row_cost(h_indptr[row + 1] - h_indptr[row]);
        if ((int64_t)lists[(size_t)top.second].size() < kRMax) heap.push(top);
      }
    }
    for (size_t wi = 0; wi < cl.wgs.size(); ++wi) {
      const int w = cl.wgs[wi];
      wg_rows[(size_t)w].swap(lists[wi]);
    }
  }
  // run order: workgroup by workgroup; the packed (column, value) arrays follow it
  std::vector<int32_t> row_of((size_t)n_rows);
  std::vector<uint32_t> pk_src((size_t)n_rows), pk_dst((size_t)n_rows + 1, 0);
  {
    int64_t k = 0;
    for (int w = 0; w < n_wg; ++w) {
      wg_row0[(size_t)w] = (int32_t)k;
      wg_nrows[(size_t)w] = (int32_t)wg_rows[(size_t)w].size();
      for (int32_t row : wg_rows[(size_t)w]) {
        row_of[(size_t)k] = row;
        pk_src[(size_t)k] = (uint32_t)h_indptr[row];
        pk_dst[(size_t)k + 1] = pk_dst[(size_t)k] + (uint32_t)(h_indptr[row + 1] - h_indptr[row]);
        ++k;
      }
    }
    NR_REQUIRE(k == n_rows, NR_ERR_ARG, "spmm_blocked: internal: %lld of %lld rows scheduled", (long long)k,
               (long long)n_rows);
  }
  for (const ClassDesc& cl : classes) {
    const int32_t cmin = cl.cmin;
    const int64_t width = cl.width, K = cl.K;
    for (size_t wi = 0; wi < cl.wgs.size(); ++wi) {
      const int w = cl.wgs[wi];
      wg_ent[w].assign((size_t)K, {});
      wg_cmb[w].assign((size_t)K, {});
      std::vector<int> pcount((size_t)K, 0);
      for (size_t si = 0; si < wg_rows[(size_t)w].size(); ++si) {
        const int64_t row = wg_rows[(size_t)w][si];
        const int32_t slot = (int32_t)si;
        // position of the row's first pair in the packed arrays
        const int64_t shift = (int64_t)pk_dst[(size_t)wg_row0[(size_t)w] + si] - h_indptr[row];
        int64_t t = h_indptr[row];
        const int64_t te = h_indptr[row + 1];
        if (t == te) wg_ent[w][0].push_back(HostEnt{slot, 0, (uint32_t)(t + shift), (int32_t)row});   // empty row
        while (t < te) {
          const int64_t k = ((int64_t)h_indices[t] - cmin) / width;
          int64_t t2 = t + 1;
          const int64_t col_end = cmin + (k + 1) * width;       // first column of the next block
          while (t2 < te && h_indices[t2] < col_end) ++t2;
          const int64_t len = t2 - t;
          if (len <= kSeg) {
            wg_ent[w][(size_t)k].push_back(HostEnt{slot, (int32_t)len, (uint32_t)(t + shift), (int32_t)row});
          } else {
            const int ns = (int)((len + kSeg - 1) / kSeg);
            NR_REQUIRE(pcount[(size_t)k] + ns <= kPMax, NR_ERR_UNSUPPORTED,
                       "spmm_blocked: more than %d hub segments in one workgroup phase — use the "
                       "work-item kernel", kPMax);
            const int first = kRMax + pcount[(size_t)k];
            for (int sg = 0; sg < ns; ++sg)
              wg_ent[w][(size_t)k].push_back(
                  HostEnt{first + sg, (int32_t)std::min<int64_t>(kSeg, len - (int64_t)sg * kSeg),
                          (uint32_t)(t + shift + (int64_t)sg * kSeg), (int32_t)row});
            wg_cmb[w][(size_t)k].push_back(make_int4(slot, first, ns, (int)row));
            pcount[(size_t)k] += ns;
          }
          t = t2;
        }
      }
      for (auto& v : wg_ent[w])
        std::stable_sort(v.begin(), v.end(), [](const HostEnt& a, const HostEnt& b) { return a.len > b.len; });
    }
  }
  // flatten
  std::vector<int32_t> ent_off((size_t)n_wg * (n_phases + 1), 0), cmb_off((size_t)n_wg * (n_phases + 1), 0);
  std::vector<int4> ent, cmb;
  for (int w = 0; w < n_wg; ++w) {
    for (int k = 0; k <= n_phases; ++k) {
      ent_off[(size_t)w * (n_phases + 1) + k] = (int32_t)ent.size();
      cmb_off[(size_t)w * (n_phases + 1) + k] = (int32_t)cmb.size();
      if (k < n_phases && (size_t)k < wg_ent[w].size()) {
        for (const HostEnt& e : wg_ent[w][(size_t)k])
          ent.push_back(make_int4(e.slot, e.len, (int)e.begin, e.owner));
        for (const int4& cm : wg_cmb[w][(size_t)k]) cmb.push_back(cm);
      }
    }
  }
  std::vector<uint32_t> wg_nnz((size_t)n_wg * 2, 0);
  int64_t nnz_cap = 0, ent_cap = 0;
  for (int w = 0; w < n_wg; ++w) {
    const int64_t b = pk_dst[(size_t)wg_row0[w]], en = pk_dst[(size_t)wg_row0[w] + (size_t)wg_nrows[w]];
    wg_nnz[2 * (size_t)w] = (uint32_t)b;
    wg_nnz[2 * (size_t)w + 1] = (uint32_t)(en - b);
    nnz_cap = std::max(nnz_cap, en - b);
    ent_cap = std::max<int64_t>(ent_cap, ent_off[(size_t)w * (n_phases + 1) + n_phases] -
                                             ent_off[(size_t)w * (n_phases + 1)]);
  }
  BlockedPlan* p = new (std::nothrow) BlockedPlan();
  NR_REQUIRE(p, NR_ERR_ARG, "spmm_blocked_plan_create: out of host memory");
  p->nnz_cap = (int)std::min<int64_t>(nnz_cap, INT32_MAX / 16);
  p->ent_cap = (int)std::min<int64_t>((ent_cap + 15) / 16 * 16, INT32_MAX / 32);
  {
    const char* off = getenv("NEUREC_SPMM_MASKED_FAST");     // "0": A/B against the general kernel
    const bool on = d == 64 && kWaves == 16 && n_phases == 1 && !(off && off[0] == '0');
    const size_t base = (size_t)kPMax * 256 + (size_t)p->ent_cap * 16;
    p->colmask_ok = on && nnz_cap < ((int64_t)1 << 24) && n_rows < ((int64_t)1 << 24) &&
                    base + (size_t)nnz_cap * 8 <= (size_t)kMaxLdsBytes;
    p->wanted_ok = on && kSeg <= 255 && n_rows < ((int64_t)1 << 24);
  }
  std::vector<int4> w_ent, w_cmb;
  std::vector<int32_t> w_ent_off((size_t)n_wg * 2, 0), w_cmb_off((size_t)n_wg * 2, 0);
  p->w_ent_cap = 0;
  p->w_nnz_cap = 0;
  p->w_bitmap_words = 0;
  if (p->wanted_ok) {
    std::vector<int32_t> order((size_t)n_rows);
    for (int64_t r = 0; r < n_rows; ++r) order[(size_t)r] = (int32_t)r;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
      return h_indptr[a + 1] - h_indptr[a] > h_indptr[b + 1] - h_indptr[b];
    });
    std::vector<std::vector<int4>> we((size_t)n_wg), wc((size_t)n_wg);
    std::vector<int> wp((size_t)n_wg, 0);
    for (int64_t k = 0; k < n_rows && p->wanted_ok; ++k) {
      const int32_t row = order[(size_t)k];
      const size_t w = (size_t)(k % n_wg);
      const int64_t b = h_indptr[row], len = h_indptr[row + 1] - b;
      if (len <= kSeg) {
        we[w].push_back(make_int4(0, (int)len, (int)(uint32_t)b, row));
      } else {
        const int ns = (int)((len + kSeg - 1) / kSeg);
        if (wp[w] + ns > kPMax) { p->wanted_ok = 0; break; }
        for (int sg = 0; sg < ns; ++sg)
          we[w].push_back(make_int4(kRMax + wp[w] + sg, (int)std::min<int64_t>(kSeg, len - (int64_t)sg * kSeg),
                                    (int)(uint32_t)(b + (int64_t)sg * kSeg), row));
        wc[w].push_back(make_int4(row, kRMax + wp[w], ns, 0));
        wp[w] += ns;
      }
    }
    for (int w = 0; w < n_wg && p->wanted_ok; ++w) {
      w_ent_off[2 * (size_t)w] = (int32_t)w_ent.size();
      w_cmb_off[2 * (size_t)w] = (int32_t)w_cmb.size();
      w_ent.insert(w_ent.end(), we[(size_t)w].begin(), we[(size_t)w].end());
      w_cmb.insert(w_cmb.end(), wc[(size_t)w].begin(), wc[(size_t)w].end());
      w_ent_off[2 * (size_t)w + 1] = (int32_t)w_ent.size();
      w_cmb_off[2 * (size_t)w + 1] = (int32_t)w_cmb.size();
      p->w_ent_cap = std::max<int>(p->w_ent_cap, (int)we[(size_t)w].size());
    }
    p->w_ent_cap = (p->w_ent_cap + 15) / 16 * 16;
    // LDS: partial slots + two descriptor lists + whatever is left for staged (column, value) pairs
    // a bit per row for the batch form of the kernel, when the matrix is small enough to afford it
    p->w_bitmap_words = n_rows <= 131072 ? (int)((n_rows + 127) / 128 * 4) : 0;
    const int64_t left = (int64_t)kMaxLdsBytes - 256 - (int64_t)kPMax * 256 - 2 * (int64_t)p->w_ent_cap * 16 -
                         (int64_t)p->w_bitmap_words * 4;
    p->w_nnz_cap = (int)std::min<int64_t>(left / 8, (int64_t)1 << 22);
    if (const char* cap = getenv("NEUREC_SPMM_WANTED_NNZ_CAP"))     // tests: force the chunked path
      p->w_nnz_cap = std::min(p->w_nnz_cap, std::max(atoi(cap), 4 * kSeg));
    if (p->w_nnz_cap < 4 * kSeg) p->wanted_ok = 0;
  }
  // wave-cooperative row-masked hop (spmm_wanted_wave_kernel): its own schedule — whole rows of <= 64
  // non-zeros and CHUNKS of <= kChunk consecutive 64-segments of longer rows are the units, dealt to
  // the workgroups by descending length.  NEUREC_SPMM_WANTED_WAVE=0 keeps the staged lane-group
  // walker (A/B runs).
  std::vector<int4> ww_ent, ww_hub, ww_lcmb;
  std::vector<int32_t> ww_off((size_t)n_wg + 1, 0), ww_choff((size_t)n_wg + 1, 0), ww_lcoff((size_t)n_wg + 1, 0), ww_gch;
  int64_t ww_segments = 0;
  p->ww_ok = 0;
  p->ww_ent_cap = 0;
  p->ww_lds_slots = 0;
  {
    const char* off = getenv("NEUREC_SPMM_WANTED_WAVE");
    const bool on = d == 64 && kWaves == 16 && kSeg <= 64 && n_rows <= 131072 * 4 && !(off && off[0] == '0');
    if (on) {
      constexpr int kChunk = 8;
      // hub >= 0: chunk of a multi-chunk row (global partials); hub == -2: a one-chunk row of > 64
      // non-zeros (segment sums in LDS); hub == -1: a whole row of <= 64
      struct Unit { int64_t len; int32_t row; int hub, seg0, nseg; };
      std::vector<Unit> units;
      for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t len = h_indptr[r + 1] - h_indptr[r];
        if (len <= kSeg) {
          units.push_back(Unit{len, (int32_t)r, -1, 0, 0});
          continue;
        }
        const int ns = (int)((len + kSeg - 1) / kSeg), nch = (ns + kChunk - 1) / kChunk;
        if (nch == 1) {
          units.push_back(Unit{len, (int32_t)r, -2, 0, ns});
          continue;
        }
        const int hub = (int)ww_hub.size();
        ww_hub.push_back(make_int4((int)r, (int)ww_segments, ns, nch));
        for (int ch = 0; ch < nch; ++ch) {
          const int s0 = ch * kChunk, s1 = std::min(ns, s0 + kChunk);
          units.push_back(Unit{std::min<int64_t>(len - (int64_t)s0 * kSeg, (int64_t)(s1 - s0) * kSeg), (int32_t)r,
                               hub, s0, s1 - s0});
        }
        ww_segments += ns;
      }
      std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.len > b.len; });
      std::vector<std::vector<int4>> per((size_t)n_wg), perl((size_t)n_wg);
      std::vector<std::vector<int32_t>> perch((size_t)n_wg);
      std::vector<int> lds_used((size_t)n_wg, 0);
      for (size_t k = 0; k < units.size(); ++k) {
        const Unit& u = units[k];
        const size_t w = k % (size_t)n_wg;
        const int64_t b = h_indptr[u.row], len = h_indptr[u.row + 1] - b;
        if (u.hub == -1) {
          per[w].push_back(make_int4(0, (int)len, (int)(uint32_t)b, u.row));
        } else if (u.hub == -2) {
          perl[w].push_back(make_int4(u.row, lds_used[w], u.nseg, 0));
          for (int sg = 0; sg < u.nseg; ++sg)
            per[w].push_back(make_int4(-(1 + lds_used[w] + sg), (int)std::min<int64_t>(kSeg, len - (int64_t)sg * kSeg),
                                       (int)(uint32_t)(b + (int64_t)sg * kSeg), u.row));
          lds_used[w] += u.nseg;
        } else {
          for (int sg = u.seg0; sg < u.seg0 + u.nseg; ++sg)
            per[w].push_back(make_int4(1 + ww_hub[(size_t)u.hub].y + sg,
                                       (int)std::min<int64_t>(kSeg, len - (int64_t)sg * kSeg),
                                       (int)(uint32_t)(b + (int64_t)sg * kSeg), u.row));
          perch[w].push_back(u.hub);
        }
      }
      p->ww_lds_slots = 0;
      for (int w = 0; w < n_wg; ++w) {
        ww_off[(size_t)w] = (int32_t)ww_ent.size();
        ww_choff[(size_t)w] = (int32_t)ww_gch.size();
        ww_lcoff[(size_t)w] = (int32_t)ww_lcmb.size();
        ww_ent.insert(ww_ent.end(), per[(size_t)w].begin(), per[(size_t)w].end());
        ww_gch.insert(ww_gch.end(), perch[(size_t)w].begin(), perch[(size_t)w].end());
        ww_lcmb.insert(ww_lcmb.end(), perl[(size_t)w].begin(), perl[(size_t)w].end());
        p->ww_ent_cap = std::max<int>(p->ww_ent_cap, (int)per[(size_t)w].size());
        p->ww_lds_slots = std::max(p->ww_lds_slots, lds_used[(size_t)w]);
      }
      ww_off[(size_t)n_wg] = (int32_t)ww_ent.size();
      ww_choff[(size_t)n_wg] = (int32_t)ww_gch.size();
      ww_lcoff[(size_t)n_wg] = (int32_t)ww_lcmb.size();
      p->ww_ent_cap = (p->ww_ent_cap + 15) / 16 * 16;
      p->ww_ok = ww_segments < ((int64_t)1 << 30) &&
                 (size_t)p->ww_lds_slots * 256 + (size_t)p->ww_ent_cap * 16 +
                         (size_t)((n_rows + 127) / 128 * 16) + 1024 <= (size_t)kMaxLdsBytes;
    }
  }
  p->n_rows = n_rows; p->nnz = nnz; p->n_wg = n_wg; p->n_phases = n_phases;
  p->seg = kSeg; p->r_max = kRMax; p->p_max = kPMax; p->waves = kWaves; p->d = d;
  p->n_ent = (int64_t)ent.size(); p->n_cmb = (int64_t)cmb.size();
  char* q = (char*)d_plan_buf;
  auto carve = [&](size_t bytes) { void* r = q; q += nr_align_up(bytes, 256); return r; };
  p->ent = (int4*)carve(ent.size() * 16 + 16);
  p->cmb = (int4*)carve(cmb.size() * 16 + 16);
  p->wg_row0 = (int32_t*)carve((size_t)n_wg * 4);
  p->wg_nrows = (int32_t*)carve((size_t)n_wg * 4);
  p->row_of = (int32_t*)carve(row_of.size() * 4);
  p->pk_src = (uint32_t*)carve(pk_src.size() * 4);
  p->pk_dst = (uint32_t*)carve(pk_dst.size() * 4);
  p->pk_idx = (int32_t*)carve((size_t)nnz * 4 + 4);
  p->pk_val = (float*)carve((size_t)nnz * 4 + 4);
  p->pk_from_idx = nullptr;
  p->pk_from_val = nullptr;
  p->wg_ent_off = (int32_t*)carve(ent_off.size() * 4);
  p->wg_cmb_off = (int32_t*)carve(cmb_off.size() * 4);
  p->wg_nnz = (uint32_t*)carve(wg_nnz.size() * 4);
  p->w_ent = (int4*)carve(w_ent.size() * 16 + 16);
  p->w_cmb = (int4*)carve(w_cmb.size() * 16 + 16);
  p->w_ent_off = (int32_t*)carve(w_ent_off.size() * 4);
  p->w_cmb_off = (int32_t*)carve(w_cmb_off.size() * 4);
  p->ww_off = p->ww_choff = p->ww_gch = p->ww_lcoff = nullptr; p->ww_ent = p->ww_hub = p->ww_lcmb = nullptr;
  p->ww_part = nullptr; p->ww_cnt = nullptr;
  if (p->ww_ok) {
    p->ww_off = (int32_t*)carve(ww_off.size() * 4);
    p->ww_choff = (int32_t*)carve(ww_choff.size() * 4);
    p->ww_ent = (int4*)carve(ww_ent.size() * 16 + 16);
    p->ww_gch = (int32_t*)carve(ww_gch.size() * 4 + 4);
    p->ww_hub = (int4*)carve(ww_hub.size() * 16 + 16);
    p->ww_lcoff = (int32_t*)carve(ww_lcoff.size() * 4);
    p->ww_lcmb = (int4*)carve(ww_lcmb.size() * 16 + 16);
    p->ww_part = (float*)carve((size_t)ww_segments * 256 + 256);
    p->ww_cnt = (unsigned*)carve(ww_hub.size() * 4 + 4);
  }
  if ((size_t)(q - (char*)d_plan_buf) > plan_bytes) {
    delete p;
    nrhip_set_error("spmm_blocked_plan_create: plan needs %zu bytes, buffer has %zu",
                    (size_t)(q - (char*)d_plan_buf), plan_bytes);
    return NR_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  auto up = [&](void* dst, const void* src, size_t n) {
    if (n && e == hipSuccess) e = hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st);
  };
  up(p->ent, ent.data(), ent.size() * 16);
  up(p->cmb, cmb.data(), cmb.size() * 16);
  up(p->wg_row0, wg_row0.data(), wg_row0.size() * 4);
  up(p->wg_nrows, wg_nrows.data(), wg_nrows.size() * 4);
  up(p->row_of, row_of.data(), row_of.size() * 4);
  up(p->pk_src, pk_src.data(), pk_src.size() * 4);
  up(p->pk_dst, pk_dst.data(), pk_dst.size() * 4);
  up(p->wg_ent_off, ent_off.data(), ent_off.size() * 4);
  up(p->wg_cmb_off, cmb_off.data(), cmb_off.size() * 4);
  up(p->wg_nnz, wg_nnz.data(), wg_nnz.size() * 4);
  up(p->w_ent, w_ent.data(), w_ent.size() * 16);
  up(p->w_cmb, w_cmb.data(), w_cmb.size() * 16);
  up(p->w_ent_off, w_ent_off.data(), w_ent_off.size() * 4);
  up(p->w_cmb_off, w_cmb_off.data(), w_cmb_off.size() * 4);
  if (p->ww_ok) {
    up(p->ww_off, ww_off.data(), ww_off.size() * 4);
    up(p->ww_choff, ww_choff.data(), ww_choff.size() * 4);
    up(p->ww_ent, ww_ent.data(), ww_ent.size() * 16);
    up(p->ww_gch, ww_gch.data(), ww_gch.size() * 4);
    up(p->ww_hub, ww_hub.data(), ww_hub.size() * 16);
    up(p->ww_lcoff, ww_lcoff.data(), ww_lcoff.size() * 4);
    up(p->ww_lcmb, ww_lcmb.data(), ww_lcmb.size() * 16);
    if (e == hipSuccess) e = hipMemsetAsync(p->ww_cnt, 0, ww_hub.size() * 4 + 4, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  const int lds = (kRMax + kPMax) * kD * 4 + kRMax * 4;           // accumulators + the workgroup's row list
  auto allow = [&](const void* fn) {
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  };
#define NR_ALLOW(DD)                                                                             \
  allow((const void*)spmm_blocked_kernel<false, 16, 8, DD>); allow((const void*)spmm_blocked_kernel<true, 16, 8, DD>); \
  allow((const void*)spmm_blocked_kernel<false, 16, 4, DD>); allow((const void*)spmm_blocked_kernel<true, 16, 4, DD>); \
  allow((const void*)spmm_blocked_kernel<false, 8, 8, DD>); allow((const void*)spmm_blocked_kernel<true, 8, 8, DD>);   \
  allow((const void*)spmm_blocked_kernel<false, 8, 4, DD>); allow((const void*)spmm_blocked_kernel<true, 8, 4, DD>)
  if (d == 16) { NR_ALLOW(16); } else if (d == 32) { NR_ALLOW(32); }
  else if (d == 64) {
    NR_ALLOW(64);
    allow((const void*)spmm_blocked_kernel<false, 16, 8, 64, true>);
  }
  else if (d == 128) { NR_ALLOW(128); } else { NR_ALLOW(256); }
#undef NR_ALLOW
  if (p->wanted_ok && e == hipSuccess)
    e = hipFuncSetAttribute((const void*)spmm_wanted_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)wanted_lds_bytes(p));
  if (p->ww_ok && e == hipSuccess)
    e = hipFuncSetAttribute((const void*)spmm_wanted_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ww_lds_bytes(p));
  if (p->colmask_ok)
    for (const void* fn : {(const void*)spmm_staged_masked_kernel<true, false>,
                           (const void*)spmm_staged_masked_kernel<false, true>,
                           (const void*)spmm_staged_masked_kernel<true, true>})
      if (e == hipSuccess)
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)colmask_lds_bytes(p));
  if (e != hipSuccess) {
    delete p;
    nrhip_set_error("spmm_blocked_plan_create: %s", hipGetErrorString(e));
    return NR_ERR_HIP;
  }
  *plan_out = p;
  return NR_OK;
}

/* Copy the matrix's (column, value) pairs into the plan's row order (once per matrix; again if the values change:
 * engine.SpmmCSR.values_changed).  The lane-group kernels read this private copy. */
int nrhip_spmm_blocked_pack(void* plan, const int32_t* d_indices, const float* d_vals, void* stream) {
  NR_REQUIRE(plan && d_indices && d_vals, NR_ERR_ARG, "spmm_blocked_pack: null pointer argument");
  BlockedPlan* p = (BlockedPlan*)plan;
  if (p->nnz > 0) {
    // the (column, value) pairs in the plan's row order: a 16-lane group per row
    const unsigned blocks = (unsigned)std::min<int64_t>((p->n_rows + 15) / 16, 8192);
    hipLaunchKernelGGL(row_order_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p->pk_src, p->pk_dst,
                       p->n_rows, d_indices, d_vals, p->pk_idx, p->pk_val);
    NR_LAUNCH_CHECK();
  }
  p->pk_from_idx = d_indices;
  p->pk_from_val = d_vals;
  return NR_OK;
}

// The kernels that walk the plan's own schedule read the plan's packed pairs; a call that hands in other CSR
// arrays than the ones packed (or none packed yet) packs first, on the caller's stream.
static int ensure_packed(const BlockedPlan* p, const int32_t* d_indices, const float* d_vals, void* stream) {
  if (p->pk_from_idx == d_indices && p->pk_from_val == d_vals) return NR_OK;
  return nrhip_spmm_blocked_pack((void*)p, d_indices, d_vals, stream);
}

#ifdef NR_WW_TIMELINE
int nrhip_ww_timeline(unsigned long long* h_out) {
  NR_CHECK_HIP(hipDeviceSynchronize());
  NR_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_ww_dbg), sizeof(unsigned long long) * 256 * 8));
  return NR_OK;
}
#endif

int nrhip_spmm_blocked_tune(int gathers_in_flight) {
  NR_REQUIRE(gathers_in_flight == 4 || gathers_in_flight == 8, NR_ERR_UNSUPPORTED,
             "spmm_blocked_tune: gathers in flight %d (4, 8)", gathers_in_flight);
  s_gathers_in_flight = gathers_in_flight;
  return NR_OK;
}

int nrhip_spmm_blocked_plan_destroy(void* plan) {
  delete (BlockedPlan*)plan;
  return NR_OK;
}

int nrhip_spmm_blocked_plan_info(const void* plan, int* n_workgroups, int* n_phases,
                                 int64_t* n_entries, int64_t* n_split) {
  NR_REQUIRE(plan, NR_ERR_ARG, "spmm_blocked_plan_info: null plan");
  const BlockedPlan* p = (const BlockedPlan*)plan;
  if (n_workgroups) *n_workgroups = p->n_wg;
  if (n_phases) *n_phases = p->n_phases;
  if (n_entries) *n_entries = p->n_ent;
  if (n_split) *n_split = p->n_cmb;
  return NR_OK;
}

int nrhip_spmm_blocked(const void* plan, const int32_t* d_indices, const float* d_vals,
                       const float* d_X, float* d_Y, const float* d_addend, const float* d_sum_in,
                       float* d_sum_out, const uint8_t* d_x_row_nonzero,
                       const uint8_t* d_y_row_wanted, void* stream) {
  NR_REQUIRE(plan && d_indices && d_vals && d_X && (d_Y || d_sum_out), NR_ERR_ARG,
             "spmm_blocked: null pointer argument");
  NR_REQUIRE((d_sum_out == nullptr) || (d_sum_in != nullptr), NR_ERR_ARG,
             "spmm_blocked: sum_out needs sum_in");
  const BlockedPlan* p = (const BlockedPlan*)plan;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)p->n_wg), block(p->waves * NR_WAVE);
  const size_t lds = (size_t)(p->r_max + p->p_max) * p->d * 4 + (size_t)p->r_max * 4;
  const bool masked = d_x_row_nonzero || d_y_row_wanted;
  const int gif = s_gathers_in_flight;
  if (p->ww_ok && d_y_row_wanted && !d_x_row_nonzero)
    return launch_wanted_wave(p, d_indices, d_vals, d_X, d_Y, d_addend, d_sum_in, d_sum_out, d_y_row_wanted,
                              LayerChain{nullptr, nullptr}, BatchLists{}, st);
  if (p->wanted_ok && d_y_row_wanted && !d_x_row_nonzero) {
    hipLaunchKernelGGL(spmm_wanted_rows_kernel, grid, block, wanted_lds_bytes(p), st, p->w_ent_off,
                       p->w_cmb_off, p->w_ent, p->w_cmb, d_indices, d_vals, (const float4*)d_X,
                       (float4*)d_Y, (const float4*)d_addend, (const float4*)d_sum_in,
                       (float4*)d_sum_out, d_y_row_wanted, p->r_max, p->p_max, p->w_ent_cap,
                       p->w_nnz_cap, LayerChain{nullptr, nullptr}, BatchLists{}, p->w_bitmap_words);
    NR_LAUNCH_CHECK();
    return NR_OK;
  }
  NR_TRY(ensure_packed(p, d_indices, d_vals, stream));
  if (p->colmask_ok && masked) {
    // an addend that is the operand itself shares its promise (zero rows where the mask is 0)
    const int addend_masked = d_x_row_nonzero && d_addend == d_X ? 1 : 0;
#define NR_STAGED(CM, RM)                                                                          \
  hipLaunchKernelGGL((spmm_staged_masked_kernel<CM, RM>), grid, block, colmask_lds_bytes(p), st,   \
                     p->wg_row0, p->wg_ent_off, p->wg_cmb_off, p->wg_nnz, p->ent, p->cmb, p->pk_idx, \
                     p->pk_val, (const float4*)d_X, (float4*)d_Y, (const float4*)d_addend,          \
                     (const float4*)d_sum_in, (float4*)d_sum_out, d_x_row_nonzero, d_y_row_wanted,  \
                     addend_masked, p->r_max, p->p_max, p->ent_cap)
    if (d_x_row_nonzero && d_y_row_wanted) NR_STAGED(true, true);
    else if (d_x_row_nonzero) NR_STAGED(true, false);
    else NR_STAGED(false, true);
#undef NR_STAGED
    NR_LAUNCH_CHECK();
    return NR_OK;
  }
#define NR_BLK(M, W, GG, DD)                                                                       \
  hipLaunchKernelGGL((spmm_blocked_kernel<M, W, GG, DD>), grid, block, lds, st, p->wg_row0,        \
                     p->wg_nrows, p->wg_ent_off, p->wg_cmb_off, p->ent, p->cmb, p->n_phases,        \
                     p->pk_idx, p->pk_val, (const float4*)d_X, (float4*)d_Y, (const float4*)d_addend, \
                     (const float4*)d_sum_in, (float4*)d_sum_out, d_x_row_nonzero, d_y_row_wanted,  \
                     p->r_max, AdamEpilogue{}, p->row_of, p->p_max)
#define NR_BLK_D(DD)                                                                               \
  if (p->waves == 16) {                                                                            \
    if (masked) { if (gif == 4) NR_BLK(true, 16, 4, DD); else NR_BLK(true, 16, 8, DD); }           \
    else { if (gif == 4) NR_BLK(false, 16, 4, DD); else NR_BLK(false, 16, 8, DD); }                \
  } else {                                                                                         \
    if (masked) { if (gif == 4) NR_BLK(true, 8, 4, DD); else NR_BLK(true, 8, 8, DD); }             \
    else { if (gif == 4) NR_BLK(false, 8, 4, DD); else NR_BLK(false, 8, 8, DD); }                  \
  }
  if (p->d == 16) { NR_BLK_D(16) } else if (p->d == 32) { NR_BLK_D(32) } else if (p->d == 64) { NR_BLK_D(64) }
  else if (p->d == 128) { NR_BLK_D(128) } else { NR_BLK_D(256) }
#undef NR_BLK_D
#undef NR_BLK
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Row-masked hop with the running layer sum completed on the way (LightGCN.py:143-146 on the batch
 * rows): d_sum_out[r] = ((d_sum_in[r] + d_layer_a[r]) + d_layer_b[r]) + (A·X)[r] for the rows with
 * d_y_row_wanted[r] != 0 (d_layer_a / d_layer_b optional) — the full hops before it then need no
 * running-sum streams at all.  Needs the wanted-rows schedule (d = 64). */
int nrhip_spmm_blocked_wanted_layers(const void* plan, const int32_t* d_indices, const float* d_vals,
                                     const float* d_X, const float* d_sum_in, const float* d_layer_a,
                                     const float* d_layer_b, float* d_sum_out,
                                     const uint8_t* d_y_row_wanted, void* stream) {
  NR_REQUIRE(plan && d_indices && d_vals && d_X && d_sum_in && d_sum_out && d_y_row_wanted, NR_ERR_ARG,
             "spmm_blocked_wanted_layers: null pointer argument");
  NR_REQUIRE(d_layer_a || !d_layer_b, NR_ERR_ARG, "spmm_blocked_wanted_layers: layer_b without layer_a");
  const BlockedPlan* p = (const BlockedPlan*)plan;
  if (p->ww_ok)
    return launch_wanted_wave(p, d_indices, d_vals, d_X, nullptr, nullptr, d_sum_in, d_sum_out, d_y_row_wanted,
                              LayerChain{(const float4*)d_layer_a, (const float4*)d_layer_b}, BatchLists{},
                              (hipStream_t)stream);
  NR_REQUIRE(p->wanted_ok, NR_ERR_UNSUPPORTED, "spmm_blocked_wanted_layers: no wanted-rows schedule");
  hipLaunchKernelGGL(spmm_wanted_rows_kernel, dim3((unsigned)p->n_wg), dim3(p->waves * NR_WAVE),
                     wanted_lds_bytes(p), (hipStream_t)stream, p->w_ent_off, p->w_cmb_off, p->w_ent,
                     p->w_cmb, d_indices, d_vals, (const float4*)d_X, (float4*)nullptr,
                     (const float4*)nullptr, (const float4*)d_sum_in, (float4*)d_sum_out,
                     d_y_row_wanted, p->r_max, p->p_max, p->w_ent_cap, p->w_nnz_cap,
                     LayerChain{(const float4*)d_layer_a, (const float4*)d_layer_b}, BatchLists{},
                     p->w_bitmap_words);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* The same hop told the batch instead of flags: wanted rows = d_users | n_users + d_pos |
 * n_users + d_neg.  It also does nrhip_lightgcn_mark_batch's job on the way: d_row_flag (zero on
 * entry) gets 1 on those rows, d_rows_out (optional) the 3*batch rows.  Needs
 * nrhip_spmm_blocked_has_wanted(plan) == 2 (wanted-rows schedule and a matrix of <= 131072 rows). */
int nrhip_spmm_blocked_wanted_batch(const void* plan, const int32_t* d_indices, const float* d_vals,
                                    const float* d_X, const float* d_sum_in, const float* d_layer_a,
                                    const float* d_layer_b, float* d_sum_out, const int32_t* d_users,
                                    const int32_t* d_pos, const int32_t* d_neg, int batch, int n_users,
                                    uint8_t* d_row_flag, int32_t* d_rows_out, void* stream) {
  NR_REQUIRE(plan && d_indices && d_vals && d_X && d_sum_in && d_sum_out && d_users && d_pos && d_neg &&
                 d_row_flag && batch >= 0 && n_users >= 0,
             NR_ERR_ARG, "spmm_blocked_wanted_batch: bad arguments");
  NR_REQUIRE(d_layer_a || !d_layer_b, NR_ERR_ARG, "spmm_blocked_wanted_batch: layer_b without layer_a");
  const BlockedPlan* p = (const BlockedPlan*)plan;
  if (batch == 0) return NR_OK;
  if (p->ww_ok)
    return launch_wanted_wave(p, d_indices, d_vals, d_X, nullptr, nullptr, d_sum_in, d_sum_out, nullptr,
                              LayerChain{(const float4*)d_layer_a, (const float4*)d_layer_b},
                              BatchLists{d_users, d_pos, d_neg, batch, n_users, d_row_flag, d_rows_out},
                              (hipStream_t)stream);
  NR_REQUIRE(p->wanted_ok && p->w_bitmap_words > 0, NR_ERR_UNSUPPORTED,
             "spmm_blocked_wanted_batch: no wanted-rows schedule with a row bit set");
  hipLaunchKernelGGL(spmm_wanted_rows_kernel, dim3((unsigned)p->n_wg), dim3(p->waves * NR_WAVE),
                     wanted_lds_bytes(p), (hipStream_t)stream, p->w_ent_off, p->w_cmb_off, p->w_ent,
                     p->w_cmb, d_indices, d_vals, (const float4*)d_X, (float4*)nullptr,
                     (const float4*)nullptr, (const float4*)d_sum_in, (float4*)d_sum_out,
                     (const uint8_t*)nullptr, p->r_max, p->p_max, p->w_ent_cap, p->w_nnz_cap,
                     LayerChain{(const float4*)d_layer_a, (const float4*)d_layer_b},
                     BatchLists{d_users, d_pos, d_neg, batch, n_users, d_row_flag, d_rows_out},
                     p->w_bitmap_words);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_spmm_blocked_has_wanted(const void* plan) {      // 0 no, 1 flag form, 2 flag and batch forms
  if (plan && ((const BlockedPlan*)plan)->ww_ok) return 2;
  if (!plan || !((const BlockedPlan*)plan)->wanted_ok) return 0;
  return ((const BlockedPlan*)plan)->w_bitmap_words > 0 ? 2 : 1;
}

/* Y is not stored: the rows' results (plus d_addend) are consumed as the dense gradient
 * g = y + d_grad_b and TF-Adam is applied to d_var / d_m / d_v in the same pass (d = 64 only).
 * clear_consumed: also zero the non-zero entries of d_addend / d_grad_b and the set bytes of
 * d_row_flag (may be NULL) once read — the re-arming nrhip_rows_clear would do afterwards. */
int nrhip_spmm_blocked_adam(const void* plan, const int32_t* d_indices, const float* d_vals,
                            const float* d_X, float* d_addend, float* d_grad_b, float* d_var,
                            float* d_m, float* d_v, float alpha, float beta1, float beta2,
                            float eps, int clear_consumed, uint8_t* d_row_flag, void* stream) {
  NR_REQUIRE(plan && d_indices && d_vals && d_X && d_grad_b && d_var && d_m && d_v, NR_ERR_ARG,
             "spmm_blocked_adam: null pointer argument");
  const BlockedPlan* p = (const BlockedPlan*)plan;
  NR_REQUIRE(p->d == 64 && p->waves == 16, NR_ERR_UNSUPPORTED,
             "spmm_blocked_adam: built for d = 64 schedules with 16 waves");
  NR_REQUIRE(d_X != d_var, NR_ERR_ARG, "spmm_blocked_adam: the operand must not be the updated table");
  const size_t lds = (size_t)(p->r_max + p->p_max) * p->d * 4 + (size_t)p->r_max * 4;   // allowed at plan creation
  NR_TRY(ensure_packed(p, d_indices, d_vals, stream));
  NR_REQUIRE(!clear_consumed || d_addend, NR_ERR_ARG, "spmm_blocked_adam: clear_consumed needs an addend");
  AdamEpilogue ad{(float4*)d_var, (float4*)d_m, (float4*)d_v, (const float4*)d_grad_b,
                  alpha, 1.0f - beta1, 1.0f - beta2, eps,
                  clear_consumed ? (float4*)d_addend : nullptr,
                  clear_consumed ? (float4*)d_grad_b : nullptr, clear_consumed ? d_row_flag : nullptr};
  hipLaunchKernelGGL((spmm_blocked_kernel<false, 16, 8, 64, true>), dim3((unsigned)p->n_wg),
                     dim3(16 * NR_WAVE), lds, (hipStream_t)stream, p->wg_row0, p->wg_nrows,
                     p->wg_ent_off, p->wg_cmb_off, p->ent, p->cmb, p->n_phases, p->pk_idx, p->pk_val,
                     (const float4*)d_X, (float4*)nullptr, (const float4*)d_addend,
                     (const float4*)nullptr, (float4*)nullptr, (const uint8_t*)nullptr,
                     (const uint8_t*)nullptr, p->r_max, ad, p->row_of, p->p_max);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
