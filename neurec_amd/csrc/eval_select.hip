// eval_select.hip — full-rank top-K selection + ranking metrics on a score matrix.
//
// Stands in for the reference's native evaluator:
//   cpp_evaluate_matrix / eval_one_user  evaluator/backend/cpp/include/evaluate.h:23-72
//   precision/recall/ap/ndcg/mrr        evaluator/backend/cpp/include/metric.h:17-117
//   train-item -inf mask                evaluator/backend/cpp/uni_evaluator.py:140-143
//   arg_top_k_2d                        util/cython/include/arg_topk.h:15-45
//
// Launch shape: one wave64 per user row (the reference: one thread-pool task
// per row).  HBM-bound: each score is read exactly once with coalesced loads;
// only scores that beat the wave's running threshold are staged in an LDS
// candidate ring, which is periodically reduced to the current top sort_len.
// Rows whose ranking is ambiguous under ties are re-ranked by an exact
// emulation of libstdc++'s heap partial_sort_copy (nr_core.h) so that the
// item order is the reference's, bit for bit.
#include "nr_common.h"
#include <vector>
#include <atomic>
#include <stdlib.h>

extern "C" int nrhip_score_gemm_items_kmajor(const void* d_ws, int cols, int d, const float** qt,
                                             int* ipad);   // score_gemm.hip

namespace {

constexpr int kSelWaves = 4;          // waves (rows) per block
constexpr int kSelSlots = 512;        // candidate slots per wave (4 KiB)
constexpr int kMaxSort = 2 * 128;     // NRHIP_MAX_TOPK * 2
constexpr int kRankStride = kMaxSort; // ints per row in the rank workspace

struct InvLog2Table { double v[128]; };
struct MetricIds { int n; int id[8]; };

__device__ __forceinline__ void wave_lds_sync() {
  // DS operations of one wave execute in order; this only pins the compiler.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Pull the `want` largest keys out of keys[0..cnt) into top[0..take) in
// descending order; returns take and the largest key left behind (0 if none).
__device__ int wave_extract_top(uint64_t* keys, int cnt, uint64_t* top, int want,
                                uint64_t& next_best) {
  const int lane = nr_lane();
  const int take = want < cnt ? want : cnt;
  for (int r = 0; r <= take; ++r) {
    uint64_t best = 0;
    int slot = -1;
    for (int i = lane; i < cnt; i += NR_WAVE) {
      uint64_t k = keys[i];
      if (k > best) { best = k; slot = i; }
    }
    uint64_t w = nr_wave_max_u64(best);
    if (r == take) { next_best = w; break; }
    if (best == w && slot >= 0) keys[slot] = 0;   // keys are unique: one lane clears
    top[r] = w;                                    // every lane writes the same value
  }
  return take;
}


// ----------------------------------------------------------------------------
// Cheap ring maintenance.  wave_extract_top above costs ~100 VALU instructions per extracted key
// (PMC: 13.7 k VALU per 41 k-element row, all of it refreshes); what a refresh needs is only the
// `want`-th largest score and the keys at or above it, in any order:
//   * a 32-pass radix select over the keys' order words (held in registers, MAXPL per lane),
//   * an in-place stable compaction of the keys with order >= that value.
// The select stops at the first prefix that leaves at most `slack` keys more than wanted: that
// prefix is a lower bound of the want-th largest score, which is all a threshold has to be.
// Every key at or above the threshold is kept (ties included), so no tie can straddle the kept /
// dropped boundary here.  Returns the number kept (>= want); T receives the threshold order word.
// ----------------------------------------------------------------------------
constexpr int kMaxPerLane = kSelSlots / NR_WAVE;

__device__ int wave_radix_compact(uint64_t* keys, int cnt, int want, int slack, uint32_t& T) {
  const int lane = nr_lane();
  uint32_t ord[kMaxPerLane];
  bool valid[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int idx = lane + NR_WAVE * i;
    valid[i] = idx < cnt;
    ord[i] = valid[i] ? nr::key_order(keys[idx]) : 0u;
  }
  uint32_t prefix = 0;
  int need = want, cand = cnt;                   // candidates = keys that match `prefix` so far
  for (int b = 31; b >= 0; --b) {
    int c1 = 0;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const uint32_t x = ord[i] ^ prefix;
      const bool hi_match = (b == 31) ? true : ((x >> (b + 1)) == 0u);
      c1 += __popcll(__ballot(valid[i] && hi_match && ((ord[i] >> b) & 1u)));
    }
    if (c1 >= need) { prefix |= (1u << b); cand = c1; }
    else { need -= c1; cand -= c1; }
    // prefix (low bits zero) is already a valid threshold: (want - need) + cand keys lie at or
    // above it.  Stop as soon as that is at most `slack` more than wanted.
    if (cand <= need + slack) break;
  }
  T = prefix;
  int out = 0;
  const int chunks = (cnt + NR_WAVE - 1) / NR_WAVE;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    if (i >= chunks) break;
    const int idx = lane + NR_WAVE * i;
    const uint64_t k = valid[i] ? keys[idx] : 0ull;          // whole chunk read before any write
    const bool keep = valid[i] && ord[i] >= prefix;
    const uint64_t mask = __ballot(keep);
    wave_lds_sync();
    if (keep) keys[out + nr_mbcnt(mask)] = k;                // out + rank <= idx: never ahead of a read
    out += __popcll(mask);
    wave_lds_sync();
  }
  return out;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const uint32_t lo = __shfl_xor((uint32_t)v, m, NR_WAVE);
  const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), m, NR_WAVE);
  return ((uint64_t)hi << 32) | lo;
}
// the key of lane + 1 (0 on the last lane)
__device__ __forceinline__ uint64_t shfl_down1_u64(uint64_t v) {
  const uint32_t lo = __shfl_down((uint32_t)v, 1, NR_WAVE);
  const uint32_t hi = __shfl_down((uint32_t)(v >> 32), 1, NR_WAVE);
  return nr_lane() == NR_WAVE - 1 ? 0ull : (((uint64_t)hi << 32) | lo);
}
// A/B knob of the short-row selection (NEUREC_SELECT_FAST=0 keeps the streaming ring for every row)
__device__ int g_select_fast = 1;
__device__ __forceinline__ bool select_fast_path() { return g_select_fast != 0; }

// descending bitonic sort of one key per lane (0 = empty sorts last)
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t v) {
  const int lane = nr_lane();
#pragma unroll
  for (int k = 2; k <= NR_WAVE; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint64_t o = shfl_xor_u64(v, j);
      const bool desc_block = (lane & k) == 0;
      const bool lower = (lane & j) == 0;
      const bool take_max = lower == desc_block;
      v = take_max ? (v > o ? v : o) : (v < o ? v : o);
    }
  }
  return v;
}

// ----------------------------------------------------------------------------
// Selection kernel.  VEC = floats per lane per load (4 needs 16-byte aligned
// rows).  rank[row][0..cut) <- item ids in rank order; flag[row] <- 1 when
// ties make the parallel answer ambiguous w.r.t. the reference's heap order.
// ----------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kSelWaves* NR_WAVE) void select_rows_kernel(
    const float* __restrict__ scores, int64_t ld, int rows, int cols, int sort_len, int cut,
    int32_t* __restrict__ rank, int32_t* __restrict__ flag, uint64_t* __restrict__ tie_mask, int64_t rank_ld, int fill,
    int32_t* __restrict__ zero, int zero_n) {
  // zero (optional): zero_n words the NEXT kernel wants cleared (the tile counters of the bucket pass) — the grid
  // clears them on its way in instead of a hipMemsetAsync launch between the two kernels
  if (zero)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += gridDim.x * blockDim.x) zero[i] = 0;
  // rank_ld: the stride of `rank`'s rows.  fill != 0: the slots [ranks written, cut) of the row are set to 0 — the
  // pruned evaluation's tile lists: a row without `cut` comparable scores (a NaN factor row) would otherwise keep what
  // a previous call left in the workspace, and every later kernel indexes M / the item copy / the buckets with them.
  // tie_mask (optional): bit k <=> the scores at ranks k and k + 1 are equal, k < cut (bit cut - 1 = the pair that
  // straddles the cut); all ones when only "some tie" is known (the streaming path)
  __shared__ uint64_t s_keys[kSelWaves][kSelSlots];
  __shared__ uint64_t s_top[kSelWaves][kMaxSort];
  const int wave = threadIdx.x / NR_WAVE;
  const int lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;

  uint64_t* keys = s_keys[wave];
  uint64_t* top = s_top[wave];
  const float* srow = scores + (int64_t)row * ld;

  // Short rows (the pruned evaluation ranks 1,282 tile maxima, then 672 rescored items per user): the whole
  // row fits in registers and the streaming ring below — several radix refreshes of ~700 VALU instructions
  // each, ~2,000+ per row — is the wrong tool.  The outputs depend on the exact order of the cut + 1 best
  // keys only; the (cut+1)-th largest of the 64 LANE maxima is a lower bound of the (cut+1)-th largest
  // score, the scores at or above it are a few dozen, and one register bitonic sort orders them.
  // Same keys, same total order, same tie rule as the general path (a tie group too large for one key
  // per lane falls through to it).
  if constexpr (VEC == 4) {
    constexpr int FR = 8;                                     // float4 registers per lane: <= 2,048 scores
    const int need = cut + 1;
    if (cols <= NR_WAVE * 4 * FR && need <= NR_WAVE && cut >= 1 && select_fast_path()) {
      float v[FR][4];
#pragma unroll
      for (int u = 0; u < FR; ++u) {
        const int e0 = (u * NR_WAVE + lane) * 4;
        if (e0 + 3 < cols) {
          const float4 t = *reinterpret_cast<const float4*>(srow + e0);
          v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[u][c] = (e0 + c < cols) ? srow[e0 + c] : NAN;   // NaN: never a key
        }
      }
      uint32_t ord[FR][4], best = 0u;                         // order words of real scores are never 0
      bool has_nan = false;                                   // a NaN among the row's own columns (see the flag below)
#pragma unroll
      for (int u = 0; u < FR; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ord[u][c] = v[u][c] >= -INFINITY ? nr::order_f32(v[u][c]) : 0u;
          best = max(best, ord[u][c]);
          has_nan = has_nan || (ord[u][c] == 0u && (u * NR_WAVE + lane) * 4 + c < cols);
        }
      const uint64_t sb = wave_sort_desc((uint64_t)best);
      // fewer than `need` lanes hold a score: 0 — everything is a candidate
      const uint32_t tau_ord = __builtin_amdgcn_readlane((uint32_t)sb, need - 1);
      int c_n = 0;
#pragma unroll
      for (int u = 0; u < FR; ++u) {
        const uint32_t m = max(max(ord[u][0], ord[u][1]), max(ord[u][2], ord[u][3]));
        if (__ballot(m != 0u && m >= tau_ord) == 0) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool pass = ord[u][c] != 0u && ord[u][c] >= tau_ord;
          const uint64_t mask = __ballot(pass);
          if (mask) {
            const int at = c_n + nr_mbcnt(mask);
            if (pass && at < NR_WAVE)
              keys[at] = ((uint64_t)ord[u][c] << 32) | (uint64_t)(0xffffffffu - (uint32_t)((u * NR_WAVE + lane) * 4 + c));
            c_n += __popcll(mask);
          }
        }
      }
      if (c_n <= NR_WAVE) {                                   // wave-uniform
        wave_lds_sync();
        const uint64_t mine = wave_sort_desc(lane < c_n ? keys[lane] : 0ull);
        const uint64_t next = shfl_down1_u64(mine);           // lane 63 and lanes past the keys: 0 = none
        const int n_out = min(cut, min(sort_len, c_n));
        const bool tie = lane < n_out && next != 0ull && nr::key_order(mine) == nr::key_order(next);
        if (lane < n_out) rank[(int64_t)row * rank_ld + lane] = (int32_t)nr::key_index(mine);
        else if (fill && lane < cut) rank[(int64_t)row * rank_ld + lane] = 0;
        const uint64_t tmask = __ballot(tie);
        // A NaN score is no key here, but in the reference it sits in the heap like any other value and every
        // comparison with it is false: what std::partial_sort_copy then returns is a function of the row's whole
        // history.  Such a row is flagged — exact_rows_kernel replays that heap, NaNs included — instead of ranking
        // the comparable scores only (and leaving the slots beyond them as the workspace held them).
        const bool nan_any = __ballot(has_nan) != 0;
        if (lane == 0) {
          flag[row] = (tmask || nan_any) ? 1 : 0;
          if (tie_mask) tie_mask[row] = nan_any ? ~0ull : tmask;
        }
        return;
      }
      wave_lds_sync();                                        // a huge tie group: the general path
    }
  }

  int cnt = 0;                        // wave-uniform
  float tau = -INFINITY;              // wave-uniform: score of the current sort_len-th best
  bool btie = false;                  // a tie straddled the kept/dropped boundary
  uint32_t btie_order = 0;

  int limit = max(128, 2 * sort_len);             // ring fill that triggers the next refresh
  if (limit > kSelSlots - NR_WAVE * VEC) limit = kSelSlots - NR_WAVE * VEC;

  auto refresh = [&]() {
    wave_lds_sync();
    if (cnt >= sort_len) {                       // cheap path: threshold + compaction, ties all kept
      uint32_t T;
      const int kept = wave_radix_compact(keys, cnt, sort_len, 16, T);
      if (kept <= limit / 2 + sort_len) {
        tau = nr::unorder_f32(T);
        cnt = kept;
        wave_lds_sync();
        return;
      }
      cnt = kept;                                // a huge tie group: fall through to the exact cut
    }
    uint64_t nb = 0;
    int take = wave_extract_top(keys, cnt, top, sort_len, nb);
    wave_lds_sync();
    if (take == sort_len) {
      uint32_t last = nr::key_order(top[take - 1]);
      tau = nr::unorder_f32(last);
      if (nb != 0 && nr::key_order(nb) == last) { btie = true; btie_order = last; }
    }
    for (int r = lane; r < take; r += NR_WAVE) keys[r] = top[r];
    cnt = take;
    wave_lds_sync();
  };

  constexpr int UNITS = 8;                        // loads in flight per lane
  constexpr int UNIT_ELEMS = NR_WAVE * VEC;
  const int step = UNITS * UNIT_ELEMS;

  bool nan_seen = false;                          // a NaN among the row's own columns: the row is flagged (see the short-row path)
  for (int base = 0; base < cols; base += step) {
    float v[UNITS][VEC];
    if (base + step <= cols) {                 // interior of the row: unconditional back-to-back loads
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int e0 = base + u * UNIT_ELEMS + lane * VEC;
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(srow + e0);
          v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
        } else {
          v[u][0] = srow[e0];
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) nan_seen = nan_seen || v[u][c] != v[u][c];
      }
    } else {
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int e0 = base + u * UNIT_ELEMS + lane * VEC;
        if constexpr (VEC == 4) {
          if (e0 + 3 < cols) {
            const float4 t = *reinterpret_cast<const float4*>(srow + e0);
            v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
          } else {
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[u][c] = (e0 + c < cols) ? srow[e0 + c] : NAN;
          }
        } else {
          v[u][0] = (e0 < cols) ? srow[e0] : NAN;   // NaN never passes `>= tau`
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) nan_seen = nan_seen || (e0 + c < cols && v[u][c] != v[u][c]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      // the threshold is refreshed early and then less and less often (128, 256, ... slots): a
      // finite tau after the first few hundred elements is what keeps the ring quiet afterwards
      if (cnt > limit) { refresh(); limit = min(2 * limit, kSelSlots - UNIT_ELEMS); }
      const int e0 = base + u * UNIT_ELEMS + lane * VEC;
      if constexpr (VEC == 4) {
        // one vote per 16-byte word first: most words hold nothing >= tau once tau is finite
        const float m = fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3]));
        if (__ballot(m >= tau) == 0) continue;
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const bool pass = v[u][c] >= tau;
        const uint64_t mask = __ballot(pass);
        if (mask) {
          if (pass) keys[cnt + nr_mbcnt(mask)] = nr::pack_key(v[u][c], (uint32_t)(e0 + c));
          cnt += __popcll(mask);
        }
      }
    }
  }

  // final ranking: compact to the keys at or above the sort_len-th score, then (when they fit one
  // per lane) a register bitonic sort; otherwise the selection loop
  wave_lds_sync();
  uint64_t nb = 0;
  int take;
  bool sorted_in_regs = false;
  if (cnt >= sort_len && sort_len < NR_WAVE) {
    uint32_t T;
    cnt = wave_radix_compact(keys, cnt, sort_len, NR_WAVE - sort_len < 16 ? NR_WAVE - sort_len : 16, T);
    wave_lds_sync();
    if (cnt <= NR_WAVE) {
      const uint64_t mine = wave_sort_desc(lane < cnt ? keys[lane] : 0ull);
      take = sort_len;                           // cnt >= sort_len here
      if (lane < take) top[lane] = mine;
      const uint32_t nlo = __builtin_amdgcn_readlane((uint32_t)mine, take < NR_WAVE ? take : 0);
      const uint32_t nhi = __builtin_amdgcn_readlane((uint32_t)(mine >> 32), take < NR_WAVE ? take : 0);
      nb = (take < cnt) ? (((uint64_t)nhi << 32) | nlo) : 0ull;   // best key left behind, if any
      sorted_in_regs = true;
    }
  }
  if (!sorted_in_regs) take = wave_extract_top(keys, cnt, top, sort_len, nb);
  wave_lds_sync();

  bool tie = false;
  for (int r = lane; r < cut && r < take; r += NR_WAVE) {
    const uint64_t a = top[r];
    const uint64_t b = (r + 1 < take) ? top[r + 1] : nb;
    if (b != 0 && nr::key_order(a) == nr::key_order(b)) tie = true;
    rank[(int64_t)row * rank_ld + r] = (int32_t)nr::key_index(a);
  }
  if (fill)
    for (int r = max(0, min(cut, take)) + lane; r < cut; r += NR_WAVE) rank[(int64_t)row * rank_ld + r] = 0;
  if (btie && take >= cut && cut >= 1 && nr::key_order(top[cut - 1]) == btie_order) tie = true;
  const bool any_tie = __ballot(tie || nan_seen) != 0;
  if (lane == 0) {
    flag[row] = any_tie ? 1 : 0;
    if (tie_mask) tie_mask[row] = any_tie ? ~0ull : 0ull;
  }
}

// ----------------------------------------------------------------------------
// The pruned evaluation's FIRST selection as a kernel of its own (r06): the `cut` largest of a row of <= 256 FR tile
// maxima, written straight into the row's tile list.  select_rows_kernel's short-row path, word for word — same keys
// (ordered score, -column), same total order — without the streaming ring next to it: that path's registers (94) held
// the launch at 5 waves per SIMD, and the kernel is one round trip for the row plus two register sorts, so the waves
// in flight are its throughput (54-64 registers: 8 waves).  Ties among maxima do not matter here (no flag is kept):
// a tie group beyond one key per lane — a zero factor row: every maximum equal — is ranked by repeated maximum
// extraction over the registers instead of the ring.  Slots beyond the comparable maxima (a NaN row) are set to 0.
// ----------------------------------------------------------------------------
template <int FR>
__global__ __launch_bounds__(kSelWaves* NR_WAVE) void select_tiles_kernel(
    const float* __restrict__ scores, int64_t ld, int rows, int cols, int cut, int32_t* __restrict__ out, int64_t out_ld,
    int32_t* __restrict__ zero, int zero_n) {
  __shared__ uint64_t s_keys[kSelWaves][NR_WAVE];
  if (zero)                                                    // (the bucket pass's counters: see select_rows_kernel)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += gridDim.x * blockDim.x) zero[i] = 0;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  const float* srow = scores + (int64_t)row * ld;
  uint32_t ord[FR][4], best = 0u;                              // order words of real scores are never 0
#pragma unroll
  for (int u = 0; u < FR; ++u) {
    const int e0 = (u * NR_WAVE + lane) * 4;
    // one float4 per (lane, u), unconditional: the row's stride ld is a multiple of 4 and >= cols, so a word that
    // starts below ld lies inside the row (its pad) — a word past it is read at the row's start and masked.  (Loads
    // behind the `e0 + 3 < cols` branch were waited for one by one: the last, partial word cost four round trips.)
    const float4 t = *reinterpret_cast<const float4*>(srow + (e0 + 3 < ld ? e0 : 0));
    float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (e0 + c < cols) ? v[c] : NAN;                // NaN: never a key
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ord[u][c] = v[c] >= -INFINITY ? nr::order_f32(v[c]) : 0u;
      best = max(best, ord[u][c]);
    }
  }
  const int need = cut + 1;
  const uint64_t sb = wave_sort_desc((uint64_t)best);
  const uint32_t tau_ord = __builtin_amdgcn_readlane((uint32_t)sb, need - 1);     // fewer than `need` lanes with a score: 0
  uint64_t* keys = s_keys[wave];
  int c_n = 0;
#pragma unroll
  for (int u = 0; u < FR; ++u) {
    const uint32_t m = max(max(ord[u][0], ord[u][1]), max(ord[u][2], ord[u][3]));
    if (__ballot(m != 0u && m >= tau_ord) == 0) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool pass = ord[u][c] != 0u && ord[u][c] >= tau_ord;
      const uint64_t mask = __ballot(pass);
      if (mask) {
        const int at = c_n + nr_mbcnt(mask);
        if (pass && at < NR_WAVE)
          keys[at] = ((uint64_t)ord[u][c] << 32) | (uint64_t)(0xffffffffu - (uint32_t)((u * NR_WAVE + lane) * 4 + c));
        c_n += __popcll(mask);
      }
    }
  }
  int32_t* orow = out + (int64_t)row * out_ld;
  if (c_n <= NR_WAVE) {                                        // wave-uniform
    wave_lds_sync();
    const uint64_t mine = wave_sort_desc(lane < c_n ? keys[lane] : 0ull);
    const int n_out = min(cut, c_n);
    if (lane < cut) orow[lane] = lane < n_out ? (int32_t)nr::key_index(mine) : 0;
    return;
  }
  // more than 64 candidates at or above the threshold (equal maxima): the `cut` best keys one after the other
  uint64_t last = ~0ull;                                       // every key is below it
  for (int k = 0; k < cut; ++k) {
    uint64_t m = 0ull;
#pragma unroll
    for (int u = 0; u < FR; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint64_t key = ord[u][c] ? ((uint64_t)ord[u][c] << 32) | (uint64_t)(0xffffffffu - (uint32_t)((u * NR_WAVE + lane) * 4 + c)) : 0ull;
        if (key < last && key > m) m = key;
      }
#pragma unroll
    for (int sh = 1; sh < NR_WAVE; sh <<= 1) {
      const uint64_t o = shfl_xor_u64(m, sh);
      m = o > m ? o : m;
    }
    if (lane == 0) orow[k] = m ? (int32_t)nr::key_index(m) : 0;
    last = m ? m : 0ull;                                       // (no key left: the remaining slots take 0)
  }
}

// ----------------------------------------------------------------------------
// Exact path for flagged rows: wave-cooperative replay of
// std::partial_sort_copy's heap (see nr_core.h).  The heap lives in LDS and
// every lane executes the (uniform) heap code; lanes only differ while
// scanning the row 64 scores at a time for elements that beat the heap root.
// ----------------------------------------------------------------------------
// The heap of a sort length <= 64, one slot per lane: slot reads are v_readlane with a scalar index, slot writes
// a lane compare + select — no LDS round trip in the sift chains (the LDS heap made a flagged 41 k-column row cost 0.31 ms, most
// of it waiting for one dependent LDS access after the other).  Every lane runs the same (uniform) control flow.
struct RegHeap {
  float v;      // this lane's slot: score
  int id;       //                   item
  __device__ __forceinline__ float gv(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
  }
  __device__ __forceinline__ int gi(int i) const { return __builtin_amdgcn_readlane(id, i); }
  __device__ __forceinline__ void set(int i, float nv, int nid) {
    const bool me = (int)nr_lane() == i;                       // (no v_writelane builtin here: compare + select)
    v = me ? nv : v;
    id = me ? nid : id;
  }
  __device__ __forceinline__ void mov(int dst, int src) { set(dst, gv(src), gi(src)); }
};

__global__ __launch_bounds__(kSelWaves* NR_WAVE) void exact_rows_kernel(
    const float* __restrict__ scores, int64_t ld, int rows, int cols, int sort_len, int cut,
    int32_t* __restrict__ rank, const int32_t* __restrict__ flag, int32_t* __restrict__ n_exact) {
  __shared__ float s_val[kSelWaves][kMaxSort];
  __shared__ int s_idx[kSelWaves][kMaxSort];
  const int wave = threadIdx.x / NR_WAVE;
  const int lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows || flag[row] == 0) return;

  const float* srow = scores + (int64_t)row * ld;
  const int m = sort_len < cols ? sort_len : cols;
  if (m <= NR_WAVE) {                                          // wave-uniform: the register heap
    RegHeap h{lane < m ? srow[lane] : 0.f, lane};
    nr::heap_make(h, m);
    constexpr int kAheadR = 16;
    for (int base0 = m; base0 < cols; base0 += kAheadR * NR_WAVE) {
      float vv[kAheadR];
#pragma unroll
      for (int g = 0; g < kAheadR; ++g) {
        const int e = base0 + g * NR_WAVE + lane;
        vv[g] = e < cols ? srow[e] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < kAheadR; ++g) {
        const int base = base0 + g * NR_WAVE;
        if (base >= cols) break;
        bool live = base + lane < cols;
        const float v = vv[g];
        for (;;) {
          const float root = h.gv(0);
          const uint64_t mask = __ballot(live && v > root);
          if (!mask) break;
          const int t = __builtin_ctzll(mask);
          const float vt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
          nr::heap_adjust(h, 0, m, vt, base + t);
          live = live && lane > t;
        }
      }
    }
    nr::heap_sort(h, m);
    if (lane < cut && lane < m) rank[(int64_t)row * kRankStride + lane] = h.id;
    if (lane == 0 && n_exact) atomicAdd(n_exact, 1);
    return;
  }
  nr::HeapView h{s_val[wave], s_idx[wave]};
  for (int i = lane; i < m; i += NR_WAVE) { h.val[i] = srow[i]; h.idx[i] = i; }
  wave_lds_sync();
  nr::heap_make(h, m);
  wave_lds_sync();
  // the scan is a chain of dependent loads unless the row is fetched ahead: 8 groups of 64 scores
  // in flight, offered to the heap in index order
  constexpr int kAhead = 32;                                   // (8: 320 us for a 41 k row, one round trip per 512 scores)
  for (int base0 = m; base0 < cols; base0 += kAhead * NR_WAVE) {
    float vv[kAhead];
#pragma unroll
    for (int g = 0; g < kAhead; ++g) {
      const int e = base0 + g * NR_WAVE + lane;
      vv[g] = e < cols ? srow[e] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < kAhead; ++g) {
      const int base = base0 + g * NR_WAVE;
      if (base >= cols) break;
      bool live = base + lane < cols;
      const float v = vv[g];
      for (;;) {
        const float root = h.val[0];
        const uint64_t mask = __ballot(live && v > root);
        if (!mask) break;
        const int t = __builtin_ctzll(mask);
        const float vt = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
        nr::heap_adjust(h, 0, m, vt, base + t);
        wave_lds_sync();
        live = live && lane > t;
      }
    }
  }
  nr::heap_sort(h, m);
  wave_lds_sync();
  for (int r = lane; r < cut && r < m; r += NR_WAVE)
    rank[(int64_t)row * kRankStride + r] = h.idx[r];
  if (lane == 0 && n_exact) atomicAdd(n_exact, 1);
}

// ----------------------------------------------------------------------------
// Metrics: one wave per row; lane k tests rank k against the user's ascending
// test list, then lane m evaluates metric m sequentially over k (the float /
// double sequence of metric.h, see nr::metric_eval).
// ----------------------------------------------------------------------------
// One row's metrics by one wave (metrics_kernel's body; rank_compact_kernel ends with it).  item0 = the item at rank
// `lane` (lanes >= the number of ranks: anything); rank_row = the row's ranks in memory for cut-offs beyond one per lane
// (may be null when top_k + 1 <= 64).  s_hit / s_pre [128], s_truth [kMetricsTruthLds]: the wave's LDS.
constexpr int kMetricsTruthLds = 256;
__device__ __forceinline__ void metrics_row(
    const int32_t* __restrict__ rank_row, int32_t item0, int row, int lane, int top_k, bool with_tie, uint64_t tm,
    const int32_t* __restrict__ users, const int64_t* __restrict__ t_indptr, const int32_t* __restrict__ t_indices,
    const MetricIds& mids, const InvLog2Table& tbl, float* __restrict__ out, int32_t* __restrict__ topk_out,
    int32_t* __restrict__ flag_io, unsigned char* s_hit, float* s_pre, int32_t* s_truth) {
  const int64_t t = users ? (int64_t)users[row] : (int64_t)row;
  const int64_t tb = t_indptr[t], te = t_indptr[t + 1];
  const int T = (int)(te - tb);
  // the user's test list into LDS (one coalesced load, in flight together with the ranks) and the membership searches
  // there: a binary search in global memory was ~5 dependent round trips per row, 46 us for 29,858 users
  constexpr int kTruthLds = kMetricsTruthLds;
  const int n_rank = top_k + (with_tie ? 1 : 0);               // (with tie masks: the first item outside too)
  const bool in_lds = T <= kTruthLds;                          // wave-uniform
  if (in_lds)
    for (int i = lane; i < T; i += NR_WAVE) s_truth[i] = t_indices[tb + i];
  wave_lds_sync();
  const int32_t* truth = in_lds ? s_truth : t_indices + tb;
  for (int k = lane; k < n_rank; k += NR_WAVE) {
    const int32_t item = k == lane ? item0 : rank_row[k];      // (item0 only: n_rank <= 64, rank_row may be null)
    s_hit[k] = nr::sorted_contains(truth, T, item) ? 1 : 0;
    if (topk_out && k < top_k) topk_out[(int64_t)row * top_k + k] = item;
  }
  wave_lds_sync();
  if (with_tie) {
    // Equal scores at neighbouring ranks: the reference's order among them is its heap's history (exact_rows_kernel
    // replays it from a full row).  Every metric here is a function of the hit / miss sequence over the ranks alone,
    // so a tie between two items that are both test items or both not cannot change any of them — the pair across
    // the cut included (rank holds top_k + 1 items here).  Only a tie whose two sides differ needs the replay, and a
    // tie group that runs on beyond the first item outside (its other members are not known here).
    bool bad = false;
    if (lane < top_k && ((tm >> lane) & 1ull) && s_hit[lane] != s_hit[lane + 1]) bad = true;
    if (((tm >> (top_k - 1)) & 1ull) && ((tm >> top_k) & 1ull)) bad = true;
    if (__ballot(bad) != 0 && lane == 0) flag_io[row] |= 1;    // (bit 1 of a pruned row: its certificate failed)
  }
  // A row without a test item among its top_k (most rows: hits are rare) has every metric 0 at every cut-off as long as
  // the user has test items at all (0 / T, 0 / idcg with idcg > 0, no reciprocal rank): written directly, the float /
  // double recurrences below are skipped.  (They are ~800 VALU instructions a row: 47 -> ~15 us for 29,858 users.)
  {
    bool h = false;
    for (int k = lane; k < top_k; k += NR_WAVE) h = h || s_hit[k] != 0;
    if (T > 0 && __ballot(h) == 0) {
      for (int e = lane; e < mids.n * top_k; e += NR_WAVE) out[(int64_t)row * mids.n * top_k + e] = 0.f;
      return;
    }
  }
  // nr::metric_eval (the float / double sequence of metric.h) with the lanes across the cut-offs k
  // instead of across the metrics: the running quantities (hit count, DCG, IDCG, the sum of the
  // precisions at the hits) are order-dependent float recurrences — every lane steps through them and
  // keeps the state at its own k — and the divisions, which are most of the instructions, are done for
  // all k at once.  (One lane per metric ran five divergent 20-step loops of fp64 divisions one after
  // the other: ~1,800 instructions a row, 88 us for 29,858 users.)  top_k <= 128: two cut-offs per lane.
  const unsigned char* hit = s_hit;
  int hits = 0, first = -1;
  float dcg = 0.f, idcg = 0.f;
  int hits_k[2] = {0, 0};
  float dcg_k[2] = {0.f, 0.f}, idcg_k[2] = {0.f, 0.f};
  for (int i = 0; i < top_k; ++i) {
    const bool h = hit[i] != 0;
    if (h) {
      hits += 1;
      dcg = (float)((double)dcg + tbl.v[i]);                 // metric.h:69-86
      if (first < 0) first = i;
    }
    if (i < T) idcg = (float)((double)idcg + tbl.v[i]);
    if ((i & (NR_WAVE - 1)) == lane) {
      const int j = i >> 6;
      hits_k[j] = hits; dcg_k[j] = dcg; idcg_k[j] = idcg;
    }
  }
  float pre_k[2], sum_k[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = lane + NR_WAVE * j;
    pre_k[j] = (float)(1.0 * hits_k[j] / (double)(unsigned)(k + 1));   // precision, metric.h:17-28
    if (k < top_k) s_pre[k] = pre_k[j];
  }
  bool want_ap = false;
  for (int m = 0; m < mids.n; ++m) want_ap = want_ap || mids.id[m] == 3;
  if (want_ap) {                                               // ap, metric.h:46-65: sum of the precisions at the hits, in order
    wave_lds_sync();
    float sum_pre = 0.f;
    for (int i = 0; i < top_k; ++i) {
      if (hit[i] != 0) sum_pre += s_pre[i];
      if ((i & (NR_WAVE - 1)) == lane) sum_k[i >> 6] = sum_pre;
    }
  }
  const float rr = first >= 0 ? (float)(1.0 / (double)(unsigned)(first + 1)) : 0.f;   // mrr, metric.h:89-109
  const float truth_len = (float)T;
  for (int m = 0; m < mids.n; ++m) {
    const int id = mids.id[m];
    float* o = out + ((int64_t)row * mids.n + m) * top_k;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lane + NR_WAVE * j;
      if (k >= top_k) continue;
      float val;
      if (id == 1) {
        val = pre_k[j];
      } else if (id == 2) {
        val = (float)(1.0 * hits_k[j] / (double)T);            // recall, metric.h:31-43
      } else if (id == 3) {
        const float ip1 = (float)(unsigned)(k + 1);
        const float denom = (truth_len < ip1) ? truth_len : ip1;
        val = (hits_k[j] == 0) ? 0.0f : sum_k[j] / denom;
      } else if (id == 4) {
        val = dcg_k[j] / idcg_k[j];
      } else {
        val = (first >= 0 && k >= first) ? rr : 0.f;
      }
      o[k] = val;
    }
  }
}


__global__ __launch_bounds__(kSelWaves* NR_WAVE) void metrics_kernel(
    const int32_t* __restrict__ rank, int rows, int top_k, const int32_t* __restrict__ users,
    const int64_t* __restrict__ t_indptr, const int32_t* __restrict__ t_indices, MetricIds mids,
    InvLog2Table tbl, float* __restrict__ out, int32_t* __restrict__ topk_out,
    const uint64_t* __restrict__ tie_mask, int32_t* __restrict__ flag_io) {
  __shared__ unsigned char s_hit[kSelWaves][128];
  __shared__ float s_pre[kSelWaves][128];
  __shared__ int32_t s_truth[kSelWaves][kMetricsTruthLds];
  const int wave = threadIdx.x / NR_WAVE;
  const int lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  const int n_rank = top_k + (tie_mask ? 1 : 0);
  const int32_t* rank_row = rank + (int64_t)row * kRankStride;
  const int32_t item0 = lane < n_rank ? rank_row[lane] : 0;
  metrics_row(rank_row, item0, row, lane, top_k, tie_mask != nullptr, tie_mask ? tie_mask[row] : 0ull, users, t_indptr,
              t_indices, mids, tbl, out, topk_out, flag_io, s_hit[wave], s_pre[wave], s_truth[wave]);
}

// limit > 0: ids outside [0, limit) become 0.  The pruned evaluation's tile ids: a row without a single comparable
// score (a NaN factor row) leaves its rank slots as the workspace held them — whatever a previous call wrote there,
// item ids of another table included — and every later kernel indexes M / the item copy / the buckets with them.
__global__ void copy_rank_kernel(const int32_t* __restrict__ rank, int rows, int top_k,
                                 int32_t* __restrict__ out, int limit) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * top_k) return;
  const int r = (int)(i / top_k), k = (int)(i % top_k);
  const int v = rank[(int64_t)r * kRankStride + k];
  out[i] = (limit > 0 && (unsigned)v >= (unsigned)limit) ? 0 : v;
}

__global__ __launch_bounds__(kSelWaves* NR_WAVE) void mask_train_kernel(
    float* __restrict__ scores, int64_t ld, const int32_t* __restrict__ users, int rows, int cols,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices) {
  const int wave = threadIdx.x / NR_WAVE;
  const int lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  const int64_t u = users ? (int64_t)users[row] : (int64_t)row;
  const int64_t b = indptr[u], e = indptr[u + 1];
  float* srow = scores + (int64_t)row * ld;
  for (int64_t j = b + lane; j < e; j += NR_WAVE) {
    const int32_t it = indices[j];
    if (it >= 0 && it < cols) srow[it] = -INFINITY;
  }
}


// ----------------------------------------------------------------------------------------------
// Pruned evaluation, level 2 (nrhip_eval_tiles).  Level 1 (nrhip_score_tilemax) left, per user,
// the maximum admissible score of every 64-item tile.  The K+1 best items of a user lie in the
// K+1 tiles with the largest maxima, and if the (K+1)-th and (K+2)-th tile maxima differ no
// item outside those tiles can reach or tie the kept set.  So: rank the tile maxima, rescore
// only the K+1 chosen tiles (the k-ascending fmaf chain the MFMA evaluates — bit-identical),
// strike the train items, select on the compact row, map the columns back to item ids.  Rows
// whose answer could depend on ties (inside the compact row or between the two boundary tile
// maxima) are flagged; the caller re-ranks those from a full score row.
// ----------------------------------------------------------------------------------------------
constexpr int kTileItems = 32;   // one MFMA row block of the scoring loop

// one wave per user row: sort the chosen tile ids, rescore them, strike train items / pad columns
__global__ __launch_bounds__(kSelWaves* NR_WAVE) void rescore_tiles_kernel(
    const float* __restrict__ P, int64_t ldp, const float* __restrict__ QT, int64_t ipad, int d,
    const int32_t* __restrict__ users, int rows, int cols, const int32_t* __restrict__ tiles,
    int tiles_ld, int n_keep, const int64_t* __restrict__ tr_indptr,
    const int32_t* __restrict__ tr_indices, float* __restrict__ C, int64_t cld,
    int32_t* __restrict__ tilemap) {
  __shared__ __attribute__((aligned(16))) float s_p[kSelWaves][128];
  __shared__ int32_t s_map[kSelWaves][64];
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  const int64_t u = users ? (int64_t)users[row] : (int64_t)row;
  // ascending tile order keeps "compact column order == item id order" (ties are broken by index)
  const int mine = lane < n_keep ? tiles[(int64_t)row * tiles_ld + lane] : INT_MAX;
  int pos = 0;
  for (int i = 0; i < n_keep; ++i) {                           // (equal ids — a NaN row's clamped garbage — keep lane order:
    const int o = __builtin_amdgcn_readlane(mine, i);          //  every slot of s_map is written)
    pos += (o < mine || (o == mine && i < lane)) ? 1 : 0;
  }
  if (lane < n_keep) { s_map[wave][pos] = mine; tilemap[(int64_t)row * n_keep + pos] = mine; }
  for (int k = lane; k < d; k += NR_WAVE) s_p[wave][k] = P[u * ldp + k];
  wave_lds_sync();
  float* crow = C + (int64_t)row * cld;
  for (int s = 0; s < n_keep; s += 2) {
    // two 32-item tiles per pass: lanes 0-31 take tile s, lanes 32-63 tile s+1.  k-major item
    // copy: a coalesced 128-byte load per half wave and k, 8 in flight; the chain is the
    // k-ascending fmaf sequence of the scoring MFMA (MF.py:120-122)
    const int sl = s + (lane >> 5);
    const bool have = sl < n_keep;
    const int item = (have ? s_map[wave][sl] : 0) * kTileItems + (lane & 31);
    const float* q = QT + item;
    float acc = 0.f;
    if (d == 64) {
      // all 64 k-rows of the pass requested at once: the loop below asks for 8 and waits, eight
      // dependent round trips per pass (the kernel was 0.44 ms of a 2.85 ms evaluation)
      float v[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) v[i] = q[(int64_t)i * ipad];
#pragma unroll
      for (int i = 0; i < 64; ++i) acc = fmaf(s_p[wave][i], v[i], acc);
    } else {
      for (int k0 = 0; k0 < d; k0 += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = q[(int64_t)min(k0 + i, d - 1) * ipad];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (k0 + i < d) acc = fmaf(s_p[wave][k0 + i], v[i], acc);
      }
    }
    if (have) crow[sl * kTileItems + (lane & 31)] = item < cols ? acc : -INFINITY;
  }
  wave_lds_sync();
  // strike the user's train items that fall in a chosen tile (uni_evaluator.py:140-143)
  const int64_t tb = tr_indptr[u], te = tr_indptr[u + 1];
  for (int64_t t = tb + lane; t < te; t += NR_WAVE) {
    const int item = tr_indices[t], tile = item / kTileItems;
    int lo = 0, hi = n_keep;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_map[wave][mid] < tile) lo = mid + 1; else hi = mid; }
    if (lo < n_keep && s_map[wave][lo] == tile) crow[lo * kTileItems + (item % kTileItems)] = -INFINITY;
  }
}

// ----------------------------------------------------------------------------------------------
// Rescoring grouped by TILE (the form nrhip_eval_tiles_bounded runs).  rescore_tiles_kernel above reads, per user,
// n_keep item tiles of 8 KB each from the L2s — 5.6 GB at the gowalla shape, the largest kernel of an evaluation
// (0.43 ms) although the arithmetic is 2.8 GFLOP.  Here the (row, tile) pairs are bucketed by tile first and a wave
// takes 32 pairs of ONE tile: the item tile is loaded once per 32 users and the scores come from the same
// v_mfma_f32_32x32x2_f32 chain as the scoring loop (bit for bit the k-ascending fmaf chain, as tilemax_fix_kernel).
//   tile_pairs_kernel     the (row, slot) pairs counted per tile in LDS, ONE global reservation per (workgroup, tile)
//                         — a tile that every user chose costs a workgroup one atomic, not one per row — then
//                         written to the tile's bucket as (row << 6 | slot)
//   chunk_scan / _list    buckets cut into chunks of <= 32 pairs
//   rescore_pairs_kernel  wave per chunk: gather the 32 user rows, 32 x 32 scores, float4 stores into C[row][slot]
//   rank_compact_kernel   strikes (uni_evaluator.py:140-143), ranking, item ids, certificate: one wave per row
// The order of the pairs inside a bucket depends on the atomics; no result does (every pair is independent).
// ----------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kPairRows = 256;                                 // rows per workgroup of tile_pairs_kernel
constexpr int kPairThreads = 1024;                             // ... and its threads: the launch is rows / 256 workgroups (117 at gowalla, half
                                                               // the CUs) of dependent LDS / global atomics — 16 waves each hide what 4 could not
constexpr int kMaxGroupedTiles = 12288;                        // LDS histogram: 4 bytes per 32-item tile

// A thread per (row, slot) pair; slot = the tile's place in the selection's order (descending maxima).  The compact
// row C[row][slot][32] is therefore NOT in item order — which only matters for the index tie-break of the second
// selection, and every row with a tie among its cut + 1 best compact scores is flagged and redone from a full row.
// (Measured on the way: a wave per row with the slots sorted by tile id: 57 us at 64 rows per workgroup — 600 k global
// reservations — and 99 us at 256 — 64 dependent iterations per wave; this form: one reservation per (workgroup, tile).)
__global__ __launch_bounds__(kPairThreads) void tile_pairs_kernel(const int32_t* __restrict__ tiles, int tiles_ld, int n_keep,
                                                         int rows, int n_tiles, int32_t* __restrict__ tilemap,
                                                         int32_t* __restrict__ gcnt, uint32_t* __restrict__ bucket,
                                                         int32_t* __restrict__ overflow) {
  extern __shared__ int32_t s_hist[];                          // [n_tiles]: counts, then the bucket cursors
  const int tid = threadIdx.x;
  for (int i = tid; i < n_tiles; i += kPairThreads) s_hist[i] = 0;
  __syncthreads();
  const int r0 = blockIdx.x * kPairRows;
  const int n_pairs = min(kPairRows, rows - r0) * n_keep;
  for (int p = tid; p < n_pairs; p += kPairThreads) {
    int t = tiles[(int64_t)(r0 + p / n_keep) * tiles_ld + p % n_keep];
    if ((unsigned)t >= (unsigned)n_tiles) t = 0;               // NaN tables: garbage ids stay in range
    atomicAdd(&s_hist[t], 1);
  }
  __syncthreads();
  for (int t = tid; t < n_tiles; t += kPairThreads) {
    const int c = s_hist[t];
    if (c) s_hist[t] = atomicAdd(&gcnt[t], c);                 // ONE reservation per (workgroup, tile)
  }
  __syncthreads();
  for (int p = tid; p < n_pairs; p += kPairThreads) {
    const int row = r0 + p / n_keep, slot = p % n_keep;
    int t = tiles[(int64_t)row * tiles_ld + slot];
    if ((unsigned)t >= (unsigned)n_tiles) t = 0;
    tilemap[(int64_t)row * n_keep + slot] = t;
    const int at = atomicAdd(&s_hist[t], 1);
    // a bucket holds one pair per row; only rows with garbage tile ids (NaN tables: the same tile several times) can
    // ask for more — a pair that does not fit is dropped and its row flagged (redone from a full score row)
    if (at < rows) bucket[(int64_t)t * rows + at] = ((uint32_t)row << 6) | (uint32_t)slot;
    else overflow[row] = 1;
  }
}

// chunk_begin[t] = sum over t' < t of ceil(cnt[t'] / 32); chunk_begin[n_tiles] = the number of chunks.  One workgroup.
// pair_begin (the compact form): the packed bucket starts, pair_begin[t] = sum over t' < t of cnt[t'].
__global__ __launch_bounds__(1024) void chunk_scan_kernel(int32_t* __restrict__ gcnt, int n_tiles, int cap,
                                                          int32_t* __restrict__ chunk_begin,
                                                          int32_t* __restrict__ pair_begin) {
  __shared__ int32_t s_sum[1024];
  __shared__ int32_t s_pairs[1024];
  const int tid = threadIdx.x;
  const int per = (n_tiles + 1023) / 1024;
  const int b = tid * per, e = min(n_tiles, b + per);
  int local = 0, pairs = 0;
  for (int t = b; t < e; ++t) {
    const int c = min(gcnt[t], cap);
    local += (c + 31) >> 5;
    pairs += c;
  }
  s_sum[tid] = local;
  s_pairs[tid] = pairs;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? s_sum[tid - off] : 0;
    const int q = tid >= off ? s_pairs[tid - off] : 0;
    __syncthreads();
    s_sum[tid] += v;
    s_pairs[tid] += q;
    __syncthreads();
  }
  int run = s_sum[tid] - local, prun = s_pairs[tid] - pairs;
  for (int t = b; t < e; ++t) {
    const int c = min(gcnt[t], cap);
    chunk_begin[t] = run;
    run += (c + 31) >> 5;
    if (pair_begin) pair_begin[t] = prun;
    prun += c;
  }
  if (tid == 1023) {
    chunk_begin[n_tiles] = s_sum[1023];
    if (pair_begin) pair_begin[n_tiles] = s_pairs[1023];
  }
  __syncthreads();
  for (int t = b; t < e; ++t) gcnt[t] = min(gcnt[t], cap);     // (the later kernels read the clamped counts)
}
// chunks[c] = {tile, first bucket entry of the chunk (absolute), pairs in the chunk, 0}: everything rescore_pairs_kernel
// needs besides the entries themselves in ONE load (r06: it read chunks -> gcnt / pair_begin -> bucket in sequence)
__global__ __launch_bounds__(256) void chunk_list_kernel(const int32_t* __restrict__ gcnt,
                                                         const int32_t* __restrict__ chunk_begin, int n_tiles,
                                                         const int32_t* __restrict__ pair_begin, int rows,
                                                         int4* __restrict__ chunks) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n_tiles) return;
  const int cnt = gcnt[t], n = (cnt + 31) >> 5, base = chunk_begin[t];
  const int64_t first = pair_begin ? (int64_t)pair_begin[t] : (int64_t)t * rows;      // packed / strided buckets
  for (int c = lane; c < n; c += 64) {
    const int64_t at = first + 32 * c;
    chunks[base + c] = make_int4(t, (int)(at & 0xffffffff), min(32, cnt - 32 * c), (int)(at >> 32));
  }
}

// ---- the COMPACT bucket form: any number of tiles (r06; BASELINE configs[3] has 31,250 32-item tiles) -------------
// The strided form above gives every tile a bucket of `rows` slots (n_tiles x rows x 4 bytes: 153 MB at gowalla, 4 GB
// at 10^6 items x 32,768 rows) and an LDS histogram of all tiles.  Here the buckets are packed (rows x n_keep slots in
// all) and the LDS histogram covers a WINDOW of kPairWindow tiles at a time, so nothing depends on the item count:
//   tile_count_kernel     pairs per tile (per window: LDS counts, one global atomic per (workgroup, tile))
//   chunk_scan_kernel<1>  pair_begin[t] = packed bucket starts next to chunk_begin[t]
//   tile_fill_kernel      the same windows again: one reservation per (workgroup, tile) in the tile's bucket, LDS
//                         cursors hand the slots out
// The order of the pairs inside a bucket depends on the atomics; no result does (every pair is independent).
constexpr int kPairWindow = 12288;                             // tiles per LDS window (48 KB)

__device__ __forceinline__ int pair_tile(const int32_t* __restrict__ tiles, int tiles_ld, int n_keep, int r0, int p,
                                         int n_tiles) {
  const int t = tiles[(int64_t)(r0 + p / n_keep) * tiles_ld + p % n_keep];
  return (unsigned)t >= (unsigned)n_tiles ? 0 : t;             // NaN tables: garbage ids stay in range
}

__global__ __launch_bounds__(256) void tile_count_kernel(const int32_t* __restrict__ tiles, int tiles_ld, int n_keep,
                                                         int rows, int n_tiles, int32_t* __restrict__ gcnt) {
  __shared__ int32_t s_hist[kPairWindow];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * kPairRows;
  const int n_pairs = min(kPairRows, rows - r0) * n_keep;
  for (int w0 = 0; w0 < n_tiles; w0 += kPairWindow) {
    const int wn = min(kPairWindow, n_tiles - w0);
    for (int i = tid; i < wn; i += 256) s_hist[i] = 0;
    __syncthreads();
    for (int p = tid; p < n_pairs; p += 256) {
      const int t = pair_tile(tiles, tiles_ld, n_keep, r0, p, n_tiles) - w0;
      if ((unsigned)t < (unsigned)wn) atomicAdd(&s_hist[t], 1);
    }
    __syncthreads();
    for (int t = tid; t < wn; t += 256)
      if (s_hist[t]) atomicAdd(&gcnt[w0 + t], s_hist[t]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void tile_fill_kernel(const int32_t* __restrict__ tiles, int tiles_ld, int n_keep,
                                                        int rows, int n_tiles, const int32_t* __restrict__ pair_begin,
                                                        int32_t* __restrict__ cursor, int32_t* __restrict__ tilemap,
                                                        uint32_t* __restrict__ bucket) {
  __shared__ int32_t s_hist[kPairWindow];                      // counts, then the bucket cursors
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * kPairRows;
  const int n_pairs = min(kPairRows, rows - r0) * n_keep;
  for (int p = tid; p < n_pairs; p += 256)
    tilemap[(int64_t)(r0 + p / n_keep) * n_keep + p % n_keep] = pair_tile(tiles, tiles_ld, n_keep, r0, p, n_tiles);
  for (int w0 = 0; w0 < n_tiles; w0 += kPairWindow) {
    const int wn = min(kPairWindow, n_tiles - w0);
    for (int i = tid; i < wn; i += 256) s_hist[i] = 0;
    __syncthreads();
    for (int p = tid; p < n_pairs; p += 256) {
      const int t = pair_tile(tiles, tiles_ld, n_keep, r0, p, n_tiles) - w0;
      if ((unsigned)t < (unsigned)wn) atomicAdd(&s_hist[t], 1);
    }
    __syncthreads();
    for (int t = tid; t < wn; t += 256) {
      const int c = s_hist[t];
      if (c) s_hist[t] = pair_begin[w0 + t] + atomicAdd(&cursor[w0 + t], c);   // ONE reservation per (workgroup, tile)
    }
    __syncthreads();
    for (int p = tid; p < n_pairs; p += 256) {
      const int t = pair_tile(tiles, tiles_ld, n_keep, r0, p, n_tiles) - w0;
      if ((unsigned)t < (unsigned)wn)
        bucket[atomicAdd(&s_hist[t], 1)] = ((uint32_t)(r0 + p / n_keep) << 6) | (uint32_t)(p % n_keep);
    }
    __syncthreads();
  }
}

template <int KS>
__global__ __launch_bounds__(256) void rescore_pairs_kernel(
    const float* __restrict__ P, int64_t ldp, int d, const float* __restrict__ QT, int64_t ipad, int cols,
    const int32_t* __restrict__ users, int rows, const int32_t* __restrict__ chunk_begin, int n_tiles,
    const int4* __restrict__ chunks, const uint32_t* __restrict__ bucket, float* __restrict__ C, int64_t cld) {
  constexpr int DP = 2 * KS;
  __shared__ float sB[4][32][DP + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int c = blockIdx.x * 4 + wave;
  if (c >= chunk_begin[n_tiles]) return;
  const int4 ck = chunks[c];
  const int t = ck.x, n = ck.z;
  uint32_t pair = 0u;
  if (j < n) pair = bucket[(((int64_t)ck.w << 32) | (uint32_t)ck.y) + j];
  const int row = (int)(pair >> 6), slot = (int)(pair & 63u);
  const int64_t u = users ? (int64_t)users[row] : (int64_t)row;
  float a[KS], b[KS];
  const float* q = QT + (int64_t)h * ipad + (int64_t)t * 32 + j;
#pragma unroll
  for (int s = 0; s < KS; ++s) a[s] = q[(int64_t)2 * s * ipad];
  {
    // the 32 factor rows, one coalesced row per load, all in flight, into this wave's LDS slice
    float v0[32], v1[DP > 64 ? 32 : 1];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const float* pr = P + __shfl(u, r, 64) * ldp;
      v0[r] = lane < d ? pr[lane] : 0.f;
      if (DP > 64) v1[r] = lane + 64 < d ? pr[lane + 64] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (lane < DP) sB[wave][r][lane] = v0[r];
      if (DP > 64) sB[wave][r][lane + 64] = v1[r];
    }
  }
  __builtin_amdgcn_wave_barrier();
  wave_lds_sync();
#pragma unroll
  for (int s = 0; s < KS; ++s) b[s] = sB[wave][j][2 * s + h];
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  if (j >= n) return;
  // items of this lane: 32 t + 8 g + 4 h + 0..3 for g = 0..3 (accumulator registers 4 g .. 4 g + 3)
  float* crow = C + (int64_t)row * cld + slot * kTileItems + 4 * h;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int item = t * kTileItems + 8 * g + 4 * h;
    float4 v;
    v.x = item + 0 < cols ? acc[4 * g + 0] : -INFINITY;
    v.y = item + 1 < cols ? acc[4 * g + 1] : -INFINITY;
    v.z = item + 2 < cols ? acc[4 * g + 2] : -INFINITY;
    v.w = item + 3 < cols ? acc[4 * g + 3] : -INFINITY;
    *reinterpret_cast<float4*>(crow + 8 * g) = v;
  }
}

// compact column -> item id; boundary check on the tile maxima -> flag_out
__global__ __launch_bounds__(256) void remap_rank_kernel(int32_t* __restrict__ rank,
                                                         const int32_t* __restrict__ sel_flag,
                                                         const int32_t* __restrict__ tilemap,
                                                         const int32_t* __restrict__ tiles,
                                                         int tiles_ld, const float* __restrict__ M,
                                                         int64_t mld, int rows, int n_keep, int cut, int top_k,
                                                         const float* __restrict__ C, int64_t cld,
                                                         const float* __restrict__ eps,
                                                         const int32_t* __restrict__ overflow,
                                                         int32_t* __restrict__ flag_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int col_k = rank[(int64_t)row * kRankStride + top_k - 1];     // compact column of the top_k-th best (read before the remap)
  for (int k = lane; k < cut; k += NR_WAVE) {
    const int col = rank[(int64_t)row * kRankStride + k];
    rank[(int64_t)row * kRankStride + k] =
        tilemap[(int64_t)row * n_keep + col / kTileItems] * kTileItems + col % kTileItems;
  }
  if (lane == 0) {
    const float inside = M[(int64_t)row * mld + tiles[(int64_t)row * tiles_ld + n_keep - 1]];
    const float outside = M[(int64_t)row * mld + tiles[(int64_t)row * tiles_ld + n_keep]];
    if (eps) {
      // bounded maxima (nrhip_score_filter_tilemax): every item of a tile that was not rescored scores at most
      // outside + eps[row] in the fp32 chain; the row stands only if its cut-th rescored score is strictly above
      const float s_k = C[(int64_t)row * cld + col_k];
      // (the sum rounded UP: the comparison itself must not eat into the bound; a NaN anywhere flags the row)
      // (2: the bound did not certify the row — what a search arithmetic can be blamed for; 1: ties, bucket overflow)
      flag_out[row] = !(s_k > nr_add_up(outside, eps[row])) ? 2 : 0;
    } else {
      flag_out[row] = !(inside > outside) ? 1 : 0;
    }
    if (overflow && overflow[row]) flag_out[row] |= 1;         // a pair of this row did not fit its tile's bucket
    // (ties inside the compact row: metrics_kernel decides from the tie mask whether they can change a metric)
  }
}

// ----------------------------------------------------------------------------------------------
// Level 2, the ranking of a compact row as ONE launch (r06; the grouped forms): what strike_compact_kernel +
// select_rows_kernel<4> + remap_rank_kernel did in three passes over C (103 MB at gowalla) and two launch gaps.  One
// wave per row:
//   * the user's train items that fall in a rescored tile become bits of an LDS bitmap over the compact row
//     (uni_evaluator.py:140-143's strikes) — C itself is only read;
//   * the whole row (<= 63 x 32 scores) sits in registers, struck entries become -inf as they arrive, and the
//     selection is select_rows_kernel's short-row path: same keys (ordered score, -compact column), same total order,
//     same tie mask;
//   * compact columns -> item ids, and the boundary certificate on the tile maxima, as remap_rank_kernel.
// A tie group too large for one key per lane (that path falls back to the streaming ring in select_rows_kernel) flags
// the row here: it is then ranked from a full score row, which is exact whatever the cause.
// ----------------------------------------------------------------------------------------------
template <int FR>                                              // float4 registers per lane: the compact row holds <= 256 FR scores
__global__ __launch_bounds__(kSelWaves* NR_WAVE) void rank_compact_kernel(
    const float* __restrict__ C, int64_t cld, int rows, int n_keep, int top_k, const int32_t* __restrict__ users,
    const int64_t* __restrict__ tr_indptr, const int32_t* __restrict__ tr_indices,
    const int32_t* __restrict__ tilemap, const int32_t* __restrict__ tiles, int tiles_ld,
    const float* __restrict__ M, int64_t mld, const float* __restrict__ eps, const int32_t* __restrict__ overflow,
    int32_t* __restrict__ flag_out, const int64_t* __restrict__ t_indptr, const int32_t* __restrict__ t_indices,
    MetricIds mids, InvLog2Table tbl, float* __restrict__ out) {
  // (the ranks and the tie mask never leave the wave: the row's metrics are formed here, metrics_row)
  __shared__ unsigned char s_hit[kSelWaves][128];
  __shared__ float s_pre[kSelWaves][128];
  __shared__ int32_t s_truth[kSelWaves][kMetricsTruthLds];
  __shared__ int32_t s_map[kSelWaves][NR_WAVE];
  __shared__ uint32_t s_strike[kSelWaves][NR_WAVE];
  __shared__ uint64_t s_keys[kSelWaves][NR_WAVE];
  __shared__ uint32_t s_bits[kSelWaves][NR_WAVE];             // 2,048-bit filter of the chosen tiles (tile mod 2,048)
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int row = blockIdx.x * kSelWaves + wave;
  if (row >= rows) return;
  const int my_tile = lane < n_keep ? tilemap[(int64_t)row * n_keep + lane] : -1;
  s_map[wave][lane] = my_tile;
  s_strike[wave][lane] = 0u;
  s_bits[wave][lane] = 0u;
  // (FR = 4 for n_keep <= 32 — the usual 22-27 tiles: half the registers of the 2,048-score form, 8 waves per SIMD
  //  instead of 5 in a kernel that is nothing but dependent round trips: 85 -> us per 29,858 rows)
  const int cols = n_keep * kTileItems, cut = top_k + 1, need = cut + 1;
  const float* srow = C + (int64_t)row * cld;
  float v[FR][4];
#pragma unroll
  for (int u = 0; u < FR; ++u) {                              // the row is requested before the train list is walked
    const int e0 = (u * NR_WAVE + lane) * 4;
    // (unconditional, on a clamped position, masked afterwards: a load behind a branch is waited for before the next
    //  one is issued — FR round trips for the row instead of one; cols is a multiple of 32 and >= 32)
    const float4 t = *reinterpret_cast<const float4*>(srow + min(e0, cols - 4));
    const bool in = e0 < cols;
    v[u][0] = in ? t.x : NAN; v[u][1] = in ? t.y : NAN; v[u][2] = in ? t.z : NAN; v[u][3] = in ? t.w : NAN;   // NaN: never a key
  }
  wave_lds_sync();
  if (my_tile >= 0) atomicOr(&s_bits[wave][(my_tile & 2047) >> 5], 1u << (my_tile & 31));
  wave_lds_sync();
  const int64_t usr = users ? (int64_t)users[row] : (int64_t)row;
  const int64_t tb = tr_indptr[usr], te = tr_indptr[usr + 1];
  for (int64_t t = tb + lane; t < te; t += NR_WAVE) {
    const int item = tr_indices[t], tile = item / kTileItems;
    // most train items lie in tiles that were not rescored: one LDS word says so (the slots are in selection order,
    // not sorted — without the filter every train item walked all n_keep of them: 73 -> us per 29,858 rows)
    if (!((s_bits[wave][(tile & 2047) >> 5] >> (tile & 31)) & 1u)) continue;
    for (int k = 0; k < n_keep; ++k)
      if (s_map[wave][k] == tile) atomicOr(&s_strike[wave][k], 1u << (item % kTileItems));
  }
  wave_lds_sync();
  uint32_t ord[FR][4], best = 0u;                             // order words of real scores are never 0
#pragma unroll
  for (int u = 0; u < FR; ++u) {
    const int e0 = (u * NR_WAVE + lane) * 4;
    const uint32_t nib = e0 < cols ? (s_strike[wave][e0 / kTileItems] >> (e0 % kTileItems)) & 0xFu : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = ((nib >> c) & 1u) ? -INFINITY : v[u][c];
      ord[u][c] = x >= -INFINITY ? nr::order_f32(x) : 0u;
      best = max(best, ord[u][c]);
    }
  }
  const uint64_t sb = wave_sort_desc((uint64_t)best);
  const uint32_t tau_ord = __builtin_amdgcn_readlane((uint32_t)sb, need - 1);   // fewer than `need` lanes with a score: 0
  uint64_t* keys = s_keys[wave];
  int c_n = 0;
#pragma unroll
  for (int u = 0; u < FR; ++u) {
    const uint32_t m = max(max(ord[u][0], ord[u][1]), max(ord[u][2], ord[u][3]));
    if (__ballot(m != 0u && m >= tau_ord) == 0) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool pass = ord[u][c] != 0u && ord[u][c] >= tau_ord;
      const uint64_t mask = __ballot(pass);
      if (mask) {
        const int at = c_n + nr_mbcnt(mask);
        if (pass && at < NR_WAVE)
          keys[at] = ((uint64_t)ord[u][c] << 32) | (uint64_t)(0xffffffffu - (uint32_t)((u * NR_WAVE + lane) * 4 + c));
        c_n += __popcll(mask);
      }
    }
  }
  if (c_n > NR_WAVE) {                                        // wave-uniform: a tie group beyond one key per lane
    if (lane == 0) flag_out[row] = 1;                         // (the row is redone from a full score row)
    for (int e = lane; e < mids.n * top_k; e += NR_WAVE) out[(int64_t)row * mids.n * top_k + e] = 0.f;
    return;
  }
  wave_lds_sync();
  const uint64_t mine = wave_sort_desc(lane < c_n ? keys[lane] : 0ull);
  const uint64_t next = shfl_down1_u64(mine);                 // lane 63 and lanes past the keys: 0 = none
  const int n_out = min(cut, c_n);
  const bool tie = lane < n_out && next != 0ull && nr::key_order(mine) == nr::key_order(next);
  const uint64_t tmask = __ballot(tie);
  int32_t item = -1;                                          // (no item: never a test item)
  if (lane < n_out) {
    const int col = (int)nr::key_index(mine);
    item = s_map[wave][col / kTileItems] * kTileItems + col % kTileItems;
  }
  // the score of the top_k-th best, from its key (the order word is a bijection on the scores; -0 reads as +0)
  const uint32_t ok_lo = __builtin_amdgcn_readlane((uint32_t)(mine >> 32), top_k - 1);
  if (lane == 0) {
    const float inside = M[(int64_t)row * mld + tiles[(int64_t)row * tiles_ld + n_keep - 1]];
    const float outside = M[(int64_t)row * mld + tiles[(int64_t)row * tiles_ld + n_keep]];
    int flag;
    if (n_out < top_k) {
      flag = 1;                                               // fewer than top_k rankable scores in the rescored tiles
    } else if (eps) {
      const float s_k = nr::unorder_f32(ok_lo);
      flag = !(s_k > nr_add_up(outside, eps[row])) ? 2 : 0;   // (2: the bound did not certify the row; see remap_rank_kernel)
    } else {
      flag = !(inside > outside) ? 1 : 0;
    }
    if (overflow && overflow[row]) flag |= 1;
    flag_out[row] = flag;
  }
  // 5. the metrics of the row (lane 0 ORs bit 0 into the flag it has just written when a tie could change one)
  metrics_row(nullptr, item, row, lane, top_k, true, tmask, users, t_indptr, t_indices, mids, tbl, out, nullptr,
              flag_out, s_hit[wave], s_pre[wave], s_truth[wave]);
}

// column sums in fp64: stage 1 sums 256-row slabs, stage 2 adds the slabs in
// slab order (fixed order => deterministic).
constexpr int kSlabRows = 256;
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ mat, int64_t ld,
                                                     int rows, int cols,
                                                     double* __restrict__ partial) {
  __shared__ double s_part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  const int r0 = blockIdx.y * kSlabRows;
  const int r1 = min(rows, r0 + kSlabRows);
  double acc = 0.0;
  if (c < cols) {
    // the thread's 64 rows (r0 + g, + 4, ...) requested at once, added in row order: 936 waves is all this launch has,
    // and with 8 loads in flight each it was eight memory round trips long (18 us for 29,858 x 100; one load per add: 64)
    constexpr int kMine = kSlabRows / 4;
    float v[kMine];
    // (unconditional loads, the row clamped: a load behind a branch cannot be counted, and the compiler then waits for
    //  each one before it issues the next — 64 round trips, which is what this kernel's 16-20 us were)
#pragma unroll
    for (int i = 0; i < kMine; ++i) v[i] = mat[(int64_t)min(r0 + g + 4 * i, r1 - 1) * ld + c];
#pragma unroll
    for (int i = 0; i < kMine; ++i)
      if (r0 + g + 4 * i < r1) acc += (double)v[i];
  }
  s_part[g][threadIdx.x & 63] = acc;
  __syncthreads();
  if (g == 0 && c < cols)
    partial[(int64_t)blockIdx.y * cols + c] =
        ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) +
        s_part[3][threadIdx.x];
}
__global__ void colsum_stage2(const double* __restrict__ partial, int n_slabs, int cols,
                              double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  double acc = 0.0;
  int s = 0;
  for (; s + 16 <= n_slabs; s += 16) {             // 16 loads in flight, added in slab order (one load per add was
    double v[16];                                  // a chain of n_slabs memory round trips: 29 us for 117 slabs)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = partial[(int64_t)(s + i) * cols + c];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i];
  }
  for (; s < n_slabs; ++s) acc += partial[(int64_t)s * cols + c];
  out[c] = acc;
}

// colsum_stage2 with one more workgroup (the last) that counts the flagged rows next to the sums: flag_out[0] <- rows
// with a flag, flag_out[1] <- rows among them whose certificate failed (flag bit 1), as doubles — one device->host copy
// brings sums and counts (r06: a count_flags launch did this after the sums, 6 us + a launch gap)
__global__ __launch_bounds__(1024) void colsum_stage2_flags(const double* __restrict__ partial, int n_slabs, int cols,
                                                            double* __restrict__ out, const int32_t* __restrict__ flags,
                                                            int n, double* __restrict__ flag_out) {
  if (blockIdx.x + 1 < gridDim.x) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double acc = 0.0;
    int s = 0;
    for (; s + 16 <= n_slabs; s += 16) {             // (colsum_stage2's order: slab after slab)
      double v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = partial[(int64_t)(s + i) * cols + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += v[i];
    }
    for (; s < n_slabs; ++s) acc += partial[(int64_t)s * cols + c];
    out[c] = acc;
    return;
  }
  __shared__ int s_part[1024], s_cert[1024];
  int c = 0, k = 0;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {         // 8 loads in flight per thread
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 1024 < n ? flags[i0 + u * 1024] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { c += v[u] != 0 ? 1 : 0; k += (v[u] & 2) ? 1 : 0; }
  }
  s_part[threadIdx.x] = c;
  s_cert[threadIdx.x] = k;
  __syncthreads();
  for (int st = 512; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) { s_part[threadIdx.x] += s_part[threadIdx.x + st]; s_cert[threadIdx.x] += s_cert[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { flag_out[0] = (double)s_part[0]; flag_out[1] = (double)s_cert[0]; }
}

struct EvalWs {
  int32_t* rank;
  int32_t* flag;
  int32_t* n_exact;
};
size_t eval_ws_bytes(int rows) {
  return nr_align_up((size_t)rows * kRankStride * sizeof(int32_t), 256) +
         nr_align_up((size_t)rows * sizeof(int32_t), 256) + 256;
}
EvalWs carve_ws(void* ws, int rows) {
  char* p = (char*)ws;
  EvalWs w;
  w.rank = (int32_t*)p;
  p += nr_align_up((size_t)rows * kRankStride * sizeof(int32_t), 256);
  w.flag = (int32_t*)p;
  p += nr_align_up((size_t)rows * sizeof(int32_t), 256);
  w.n_exact = (int32_t*)p;
  return w;
}

// NEUREC_SELECT_FAST=0 (read once per device) switches the short-row register selection off
int select_knob_once() {
  static std::atomic<int> done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return NR_OK;
  if (done[dev].load(std::memory_order_acquire)) return NR_OK;
  const char* e = getenv("NEUREC_SELECT_FAST");
  const int v = (e && e[0] == '0') ? 0 : 1;
  NR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_select_fast), &v, sizeof(v)));
  done[dev].store(1, std::memory_order_release);
  return NR_OK;
}

// the same knob on the host: NEUREC_SELECT_FAST=0 also keeps the pruned evaluation's first selection on the general kernel
static bool select_fast_host() {
  static const int v = [] { const char* e = getenv("NEUREC_SELECT_FAST"); return (e && e[0] == '0') ? 0 : 1; }();
  return v != 0;
}

int run_selection(const float* d_scores, int64_t ld, int rows, int cols, int sort_len, int cut,
                  const EvalWs& w, hipStream_t st) {
  { const int rc = select_knob_once(); if (rc != NR_OK) return rc; }
  const int blocks = (rows + kSelWaves - 1) / kSelWaves;
  NR_CHECK_HIP(hipMemsetAsync(w.n_exact, 0, sizeof(int32_t), st));
  const bool vec4 = (ld % 4 == 0) && (((uintptr_t)d_scores) % 16 == 0);
  if (vec4)
    hipLaunchKernelGGL(select_rows_kernel<4>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st,
                       d_scores, ld, rows, cols, sort_len, cut, w.rank, w.flag, (uint64_t*)nullptr,
                       (int64_t)kRankStride, 0, (int32_t*)nullptr, 0);
  else
    hipLaunchKernelGGL(select_rows_kernel<1>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st,
                       d_scores, ld, rows, cols, sort_len, cut, w.rank, w.flag, (uint64_t*)nullptr,
                       (int64_t)kRankStride, 0, (int32_t*)nullptr, 0);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(exact_rows_kernel, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, d_scores,
                     ld, rows, cols, sort_len, cut, w.rank, w.flag, w.n_exact);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

// ----------------------------------------------------------------------------
// Any top_k (the reference's evaluate.h:23-50 takes any K; the selection kernels above are built for K <= 128):
// one thread per score row runs the libstdc++ partial_sort_copy replay itself (nr::partial_sort_copy_emul — the
// definition the parallel selection is checked against) over its row, heap in global memory, then the metrics with
// the reference's float / double sequence (nr::metric_eval).  Slow and simple: a row is ~cols + 2K·ln(cols/2K)·log2K
// sequential steps; the common cut-offs (<= 128) never come here.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void eval_bigk_kernel(
    const float* __restrict__ S, int64_t ld, int rows, int cols, int sort_len, int top_k,
    const int32_t* __restrict__ users, const int64_t* __restrict__ t_indptr, const int32_t* __restrict__ t_indices,
    MetricIds mids, const double* __restrict__ inv_log2, float* __restrict__ heap_val, int* __restrict__ heap_idx,
    float* __restrict__ out, int32_t* __restrict__ topk_out) {
  const int row = blockIdx.x * 64 + threadIdx.x;
  if (row >= rows) return;
  float* hv = heap_val + (int64_t)row * sort_len;
  int* hi = heap_idx + (int64_t)row * sort_len;
  nr::partial_sort_copy_emul(S + (int64_t)row * ld, cols, sort_len, hv, hi);
  const int64_t t = users ? (int64_t)users[row] : (int64_t)row;
  const int64_t tb = t_indptr[t];
  const int T = (int)(t_indptr[t + 1] - tb);
  if (topk_out)
    for (int k = 0; k < top_k; ++k) topk_out[(int64_t)row * top_k + k] = hi[k];
  for (int m = 0; m < mids.n; ++m)
    nr::metric_eval(mids.id[m], [&](int k) { return nr::sorted_contains(t_indices + tb, T, hi[k]); }, top_k, T,
                    inv_log2, out + ((int64_t)row * mids.n + m) * top_k);
}


}  // namespace

extern "C" {

int nrhip_eval_workspace_bytes(int rows, int top_k, size_t* bytes) {
  NR_REQUIRE(rows >= 0 && top_k >= 1 && bytes, NR_ERR_ARG,
             "eval_workspace_bytes: rows=%d top_k=%d", rows, top_k);
  NR_REQUIRE(top_k <= 128, NR_ERR_UNSUPPORTED, "eval: top_k=%d outside the built range 1..128",
             top_k);
  *bytes = eval_ws_bytes(rows > 0 ? rows : 1);
  return NR_OK;
}

int nrhip_mask_train(float* d_scores, int64_t ld, const int32_t* d_users, int rows, int cols,
                     const int64_t* d_tr_indptr, const int32_t* d_tr_indices, void* stream) {
  NR_REQUIRE(d_scores && d_tr_indptr && d_tr_indices && ld >= cols && rows >= 0, NR_ERR_ARG,
             "mask_train: bad arguments");
  if (rows == 0) return NR_OK;
  hipLaunchKernelGGL(mask_train_kernel, dim3((rows + kSelWaves - 1) / kSelWaves),
                     dim3(kSelWaves * NR_WAVE), 0, (hipStream_t)stream, d_scores, ld, d_users, rows,
                     cols, d_tr_indptr, d_tr_indices);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_eval_scores(const float* d_scores, int64_t ld, int rows, int cols,
                      const int32_t* d_users, const int64_t* d_truth_indptr,
                      const int32_t* d_truth_indices, const int32_t* metric_ids_host, int n_metric,
                      int top_k, float* d_out, int32_t* d_topk_out, int32_t* d_n_exact, void* d_ws,
                      size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_scores && d_truth_indptr && d_truth_indices && metric_ids_host && d_out && d_ws,
             NR_ERR_ARG, "eval_scores: null pointer argument");
  NR_REQUIRE(top_k >= 1 && top_k <= 128, NR_ERR_UNSUPPORTED,
             "eval_scores: top_k=%d outside 1..128", top_k);
  NR_REQUIRE(cols >= top_k, NR_ERR_ARG,
             "eval_scores: %d columns < top_k=%d (the reference reads past its buffer here)", cols,
             top_k);
  NR_REQUIRE(n_metric >= 1 && n_metric <= 8, NR_ERR_ARG, "eval_scores: n_metric=%d outside 1..8",
             n_metric);
  NR_REQUIRE(ld >= cols && rows >= 0, NR_ERR_ARG, "eval_scores: ld=%lld < cols=%d",
             (long long)ld, cols);
  MetricIds mids;
  mids.n = n_metric;
  for (int i = 0; i < n_metric; ++i) {
    NR_REQUIRE(metric_ids_host[i] >= 1 && metric_ids_host[i] <= 5, NR_ERR_ARG,
               "eval_scores: metric id %d is not one of 1..5", metric_ids_host[i]);
    mids.id[i] = metric_ids_host[i];
  }
  if (rows == 0) return NR_OK;
  NR_REQUIRE(ws_bytes >= eval_ws_bytes(rows), NR_ERR_WORKSPACE,
             "eval_scores: workspace %zu < %zu bytes", ws_bytes, eval_ws_bytes(rows));
  hipStream_t st = (hipStream_t)stream;
  EvalWs w = carve_ws(d_ws, rows);
  const int sort_len = (2 * top_k < cols) ? 2 * top_k : cols;   // evaluate.h:37
  int rc = run_selection(d_scores, ld, rows, cols, sort_len, top_k, w, st);
  if (rc != NR_OK) return rc;
  InvLog2Table tbl;
  for (int i = 0; i < 128; ++i) tbl.v[i] = 1.0 / log2((double)(unsigned)(i + 2));  // metric.h:78
  hipLaunchKernelGGL(metrics_kernel, dim3((rows + kSelWaves - 1) / kSelWaves),
                     dim3(kSelWaves * NR_WAVE), 0, st, w.rank, rows, top_k, d_users,
                     d_truth_indptr, d_truth_indices, mids, tbl, d_out, d_topk_out, (const uint64_t*)nullptr,
                     (int32_t*)nullptr);
  NR_LAUNCH_CHECK();
  if (d_n_exact)
    NR_CHECK_HIP(hipMemcpyAsync(d_n_exact, w.n_exact, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return NR_OK;
}

/* nrhip_eval_scores for ANY top_k (evaluate.h:23-50 has no limit): same arguments and output; workspace =
 * nrhip_eval_any_k_workspace_bytes.  One thread per row (see eval_bigk_kernel). */
static size_t eval_any_k_ws(int rows, int cols, int top_k) {
  const int sort_len = (2 * top_k < cols) ? 2 * top_k : cols;
  const size_t r = (size_t)(rows > 0 ? rows : 1);
  return nr_align_up((size_t)top_k * sizeof(double), 256) + 2 * nr_align_up(r * (size_t)sort_len * 4, 256);
}

int nrhip_eval_any_k_workspace_bytes(int rows, int cols, int top_k, size_t* bytes) {
  NR_REQUIRE(bytes && rows >= 0 && cols >= 1 && top_k >= 1, NR_ERR_ARG, "eval_any_k_workspace_bytes: bad arguments");
  *bytes = eval_any_k_ws(rows, cols, top_k);
  return NR_OK;
}

int nrhip_eval_scores_any_k(const float* d_scores, int64_t ld, int rows, int cols, const int32_t* d_users,
                            const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                            const int32_t* metric_ids_host, int n_metric, int top_k, float* d_out,
                            int32_t* d_topk_out, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_scores && d_truth_indptr && d_truth_indices && metric_ids_host && d_out && d_ws, NR_ERR_ARG,
             "eval_scores_any_k: null pointer argument");
  NR_REQUIRE(top_k >= 1 && cols >= top_k, NR_ERR_ARG,
             "eval_scores_any_k: %d columns < top_k=%d (the reference reads past its buffer here)", cols, top_k);
  NR_REQUIRE(n_metric >= 1 && n_metric <= 8, NR_ERR_ARG, "eval_scores_any_k: n_metric=%d outside 1..8", n_metric);
  NR_REQUIRE(ld >= cols && rows >= 0, NR_ERR_ARG, "eval_scores_any_k: ld=%lld < cols=%d", (long long)ld, cols);
  MetricIds mids;
  mids.n = n_metric;
  for (int i = 0; i < n_metric; ++i) {
    NR_REQUIRE(metric_ids_host[i] >= 1 && metric_ids_host[i] <= 5, NR_ERR_ARG,
               "eval_scores_any_k: metric id %d is not one of 1..5", metric_ids_host[i]);
    mids.id[i] = metric_ids_host[i];
  }
  if (rows == 0) return NR_OK;
  NR_REQUIRE(ws_bytes >= eval_any_k_ws(rows, cols, top_k), NR_ERR_WORKSPACE, "eval_scores_any_k: workspace %zu < %zu",
             ws_bytes, eval_any_k_ws(rows, cols, top_k));
  hipStream_t st = (hipStream_t)stream;
  const int sort_len = (2 * top_k < cols) ? 2 * top_k : cols;   // evaluate.h:37
  char* ws = (char*)d_ws;
  double* tbl = (double*)ws;
  float* hv = (float*)(ws + nr_align_up((size_t)top_k * sizeof(double), 256));
  int* hi = (int*)((char*)hv + nr_align_up((size_t)rows * sort_len * 4, 256));
  std::vector<double> h((size_t)top_k);
  for (int i = 0; i < top_k; ++i) h[(size_t)i] = 1.0 / log2((double)(unsigned)(i + 2));           // metric.h:78
  NR_CHECK_HIP(hipMemcpyAsync(tbl, h.data(), (size_t)top_k * sizeof(double), hipMemcpyHostToDevice, st));
  NR_CHECK_HIP(hipStreamSynchronize(st));                       // the host table goes out of scope
  hipLaunchKernelGGL(eval_bigk_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, d_scores, ld, rows, cols, sort_len,
                     top_k, d_users, d_truth_indptr, d_truth_indices, mids, tbl, hv, hi, d_out, d_topk_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Workspace of nrhip_eval_tiles for `rows` rows (includes the selection scratch). */
// tile-grouped rescoring (rescore_pairs_kernel): extra workspace behind the per-row form's; 0 = not used at this shape
// (more tiles than the LDS histogram holds, or buckets beyond 1 GB: the per-row kernel runs)
static int g_rescore_grouped = -1;
// NEUREC_RESCORE_GROUPED: 0 = the per-row rescoring kernel, 2 = the compact bucket form at every shape (tests: the form
// large item counts take, exercised on small ones); default: strided buckets where they fit, compact buckets beyond
static int rescore_grouped_mode() {
  if (g_rescore_grouped < 0) {
    const char* e = getenv("NEUREC_RESCORE_GROUPED");
    g_rescore_grouped = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : 1;
  }
  return g_rescore_grouped;
}
enum GroupedForm { kGroupedNone = 0, kGroupedStrided = 1, kGroupedCompact = 2 };
static size_t grouped_form_bytes(int form, int rows, int cols, int n_keep) {
  const size_t r = (size_t)(rows > 0 ? rows : 1);
  const size_t n_tiles = 2 * (((size_t)cols + 63) / 64);
  const size_t max_chunks = r * (size_t)n_keep / 32 + n_tiles + 1;
  if (form == kGroupedStrided) {
    if (n_tiles > (size_t)kMaxGroupedTiles || n_tiles * r * 4 > ((size_t)1 << 30) || r >= ((size_t)1 << 26)) return 0;
    return nr_align_up((n_tiles + r) * 4, 256) + nr_align_up((n_tiles + 1) * 4, 256) + nr_align_up(max_chunks * 16, 256) +
           nr_align_up(n_tiles * r * 4, 256);                   // counts + overflow flags | chunk starts | chunks | buckets
  }
  if (form == kGroupedCompact) {
    if (r >= ((size_t)1 << 26) || r * (size_t)n_keep >= ((size_t)1 << 31)) return 0;
    // counts + bucket cursors | chunk starts | bucket starts | chunks | packed buckets
    return nr_align_up(2 * n_tiles * 4, 256) + 2 * nr_align_up((n_tiles + 1) * 4, 256) + nr_align_up(max_chunks * 16, 256) +
           nr_align_up(r * (size_t)n_keep * 4, 256);
  }
  return 0;
}
// the form a batch of `rows` rows takes (0: the per-row kernel) and its workspace behind the per-row form's
static int grouped_form(int rows, int cols, int n_keep, size_t* bytes) {
  const int mode = rescore_grouped_mode();
  *bytes = 0;
  if (mode == 0) return kGroupedNone;
  if (mode == 1 && (*bytes = grouped_form_bytes(kGroupedStrided, rows, cols, n_keep)) != 0) return kGroupedStrided;
  if ((*bytes = grouped_form_bytes(kGroupedCompact, rows, cols, n_keep)) != 0) return kGroupedCompact;
  return kGroupedNone;
}
static size_t eval_tiles_ws_bytes(int rows, int n_keep) {
  const size_t r = (size_t)(rows > 0 ? rows : 1);
  // (+ the tie masks of the compact selection, 8 bytes per row, behind the compact rows)
  return eval_ws_bytes((int)r) + nr_align_up(r * (size_t)(n_keep + 1) * 4, 256) +
         nr_align_up(r * (size_t)n_keep * 4, 256) +
         nr_align_up(r * (size_t)n_keep * kTileItems * sizeof(float), 256) + nr_align_up(r * 8, 256);
}

int nrhip_eval_tiles_workspace_bytes(int rows, int top_k, size_t* bytes) {
  NR_REQUIRE(bytes && rows >= 0 && top_k >= 1 && top_k <= 62, NR_ERR_ARG,
             "eval_tiles_workspace_bytes: rows=%d top_k=%d (1..62)", rows, top_k);
  *bytes = eval_tiles_ws_bytes(rows, top_k + 1);
  return NR_OK;
}

int nrhip_eval_tiles_bounded_workspace_bytes(int rows, int cols, int top_k, int n_keep, size_t* bytes) {
  NR_REQUIRE(bytes && rows >= 0 && cols >= 1 && top_k >= 1 && n_keep >= top_k + 1 && n_keep <= 63, NR_ERR_ARG,
             "eval_tiles_bounded_workspace_bytes: rows=%d cols=%d top_k=%d n_keep=%d (top_k + 1 .. 63)", rows, cols,
             top_k, n_keep);
  // a workspace sized for `rows` serves every batch of up to `rows` rows (a loop sizes it once for its
  // full batches and a SHORT last batch may take the tile-grouped path the full ones were too large for): the
  // grouped part is the largest any row count up to `rows` can ask for
  // the strided form serves row counts up to r_group (its buckets grow with rows x tiles), the compact form any
  const size_t n_tiles = 2 * (((size_t)cols + 63) / 64);
  const size_t r_group = std::min<size_t>((size_t)(rows > 0 ? rows : 1),
                                          std::min<size_t>(((size_t)1 << 30) / (n_tiles * 4), ((size_t)1 << 26) - 1));
  size_t extra = 0;
  if (rescore_grouped_mode() != 0) {
    if (r_group) extra = grouped_form_bytes(kGroupedStrided, (int)r_group, cols, n_keep);
    extra = std::max(extra, grouped_form_bytes(kGroupedCompact, rows, cols, n_keep));
  }
  *bytes = eval_tiles_ws_bytes(rows, n_keep) + extra;
  return NR_OK;
}

/* Pruned full-rank evaluation of `rows` users from the tile maxima d_M[rows][mld] of
 * nrhip_score_tilemax (same users, same cols): writes the metric rows like nrhip_eval_scores and
 * d_flag_out[r] = 1 for the rows whose ranking may depend on ties — those rows of d_out are
 * provisional and must be recomputed from a full score row (nrhip_score_gemm + nrhip_mask_train +
 * nrhip_eval_scores).  Needs ceil(cols/64) >= top_k + 2 tiles. */
static int eval_tiles_impl(const float* d_M, int64_t mld, const float* d_P, int64_t ldp,
                           const void* d_gemm_ws, int d, const int32_t* d_users, int rows, int cols,
                           const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                           const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                           const int32_t* metric_ids_host, int n_metric, int top_k, int n_keep,
                           const float* d_eps, bool grouped, float* d_out, int32_t* d_flag_out, void* d_ws,
                           size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_M && d_P && d_gemm_ws && d_tr_indptr && d_tr_indices && d_truth_indptr && d_truth_indices &&
                 metric_ids_host && d_out && d_flag_out && d_ws,
             NR_ERR_ARG, "eval_tiles: null pointer argument");
  NR_REQUIRE(top_k >= 1 && top_k <= 62, NR_ERR_UNSUPPORTED, "eval_tiles: top_k=%d outside 1..62", top_k);
  NR_REQUIRE(d >= 1 && d <= 128 && ldp >= d, NR_ERR_UNSUPPORTED,
             "eval_tiles: embedding dim %d outside 1..128", d);
  const float* qt = nullptr;
  int ipad = 0;
  {
    const int rc0 = nrhip_score_gemm_items_kmajor(d_gemm_ws, cols, d, &qt, &ipad);
    if (rc0 != NR_OK) return rc0;
  }
  const int n_tiles = 2 * ((cols + 63) / 64);      // 32-item tiles, as nrhip_score_tilemax writes them
  NR_REQUIRE(n_keep >= top_k + 1 && n_keep <= 63, NR_ERR_ARG, "eval_tiles: n_keep=%d outside top_k + 1 .. 63", n_keep);
  NR_REQUIRE(n_tiles >= n_keep + 1 && mld >= n_tiles, NR_ERR_ARG,
             "eval_tiles: %d tiles < n_keep + 1 (use the full score path)", n_tiles);
  NR_REQUIRE(n_metric >= 1 && n_metric <= 8 && rows >= 0, NR_ERR_ARG, "eval_tiles: bad sizes");
  MetricIds mids;
  mids.n = n_metric;
  for (int i = 0; i < n_metric; ++i) {
    NR_REQUIRE(metric_ids_host[i] >= 1 && metric_ids_host[i] <= 5, NR_ERR_ARG,
               "eval_tiles: metric id %d is not one of 1..5", metric_ids_host[i]);
    mids.id[i] = metric_ids_host[i];
  }
  if (rows == 0) return NR_OK;
  size_t extra = 0;
  int form = grouped ? grouped_form(rows, cols, n_keep, &extra) : kGroupedNone;
  // a workspace without room for the form's buckets takes the packed form, then the per-row rescoring (same results)
  if (form == kGroupedStrided && ws_bytes < eval_tiles_ws_bytes(rows, n_keep) + extra) {
    extra = grouped_form_bytes(kGroupedCompact, rows, cols, n_keep);
    form = extra ? kGroupedCompact : kGroupedNone;
  }
  if (form != kGroupedNone && ws_bytes < eval_tiles_ws_bytes(rows, n_keep) + extra) { form = kGroupedNone; extra = 0; }
  NR_REQUIRE(ws_bytes >= eval_tiles_ws_bytes(rows, n_keep) + extra, NR_ERR_WORKSPACE,
             "eval_tiles: workspace %zu < %zu bytes", ws_bytes, eval_tiles_ws_bytes(rows, n_keep) + extra);
  hipStream_t st = (hipStream_t)stream;
  { const int rc = select_knob_once(); if (rc != NR_OK) return rc; }
  const int tiles_ld = n_keep + 1;
  EvalWs w = carve_ws(d_ws, rows);
  char* p = (char*)d_ws + eval_ws_bytes(rows);
  int32_t* tiles = (int32_t*)p;   p += nr_align_up((size_t)rows * tiles_ld * 4, 256);
  int32_t* tilemap = (int32_t*)p; p += nr_align_up((size_t)rows * n_keep * 4, 256);
  float* C = (float*)p;           p += nr_align_up((size_t)rows * n_keep * kTileItems * sizeof(float), 256);
  uint64_t* tmask = (uint64_t*)p;
  const int64_t cld = (int64_t)n_keep * kTileItems;
  // 1. the top_k + 2 largest tile maxima per user (their order among equal maxima is irrelevant)
  const int blocks = (rows + kSelWaves - 1) / kSelWaves;
  // (the grouped forms' tile counters lead their part of the workspace: this launch clears them on its way in)
  int32_t* const gcnt0 = form != kGroupedNone ? (int32_t*)((char*)d_ws + eval_tiles_ws_bytes(rows, n_keep)) : nullptr;
  const int gcnt0_n = form == kGroupedStrided ? n_tiles + rows : form == kGroupedCompact ? 2 * n_tiles : 0;
  const bool vec4 = (mld % 4 == 0) && (((uintptr_t)d_M) % 16 == 0);
  if (vec4 && n_tiles <= 2048 && tiles_ld + 1 <= NR_WAVE && select_fast_host()) {
    const int fr = (n_tiles + 255) / 256;                       // float4 registers per lane
#define NR_TILES_CASE(FR)                                                                                         \
  hipLaunchKernelGGL(select_tiles_kernel<FR>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, d_M, mld, rows, n_tiles, \
                     tiles_ld, tiles, (int64_t)tiles_ld, gcnt0, gcnt0_n)
    if (fr <= 2) NR_TILES_CASE(2);
    else if (fr <= 4) NR_TILES_CASE(4);
    else if (fr <= 6) NR_TILES_CASE(6);
    else NR_TILES_CASE(8);
#undef NR_TILES_CASE
  } else if (vec4)
    hipLaunchKernelGGL(select_rows_kernel<4>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, d_M,
                       mld, rows, n_tiles, tiles_ld, tiles_ld, tiles, w.flag, (uint64_t*)nullptr, (int64_t)tiles_ld, 1,
                       gcnt0, gcnt0_n);
  else
    hipLaunchKernelGGL(select_rows_kernel<1>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, d_M,
                       mld, rows, n_tiles, tiles_ld, tiles_ld, tiles, w.flag, (uint64_t*)nullptr, (int64_t)tiles_ld, 1,
                       gcnt0, gcnt0_n);
  NR_LAUNCH_CHECK();
  // (straight into the tile lists, unfilled slots zeroed: r06 — a copy_rank launch did both before)
  // 2. rescore the chosen tiles, train items struck out
  int32_t* overflow = nullptr;
  if (form != kGroupedNone) {
    char* x = (char*)d_ws + eval_tiles_ws_bytes(rows, n_keep);
    const size_t max_chunks = (size_t)rows * n_keep / 32 + n_tiles + 1;
    int32_t *gcnt, *chunk_begin, *pair_begin = nullptr;
    int4* chunks;
    uint32_t* bucket;
    if (form == kGroupedStrided) {
      gcnt = (int32_t*)x;        x += nr_align_up(((size_t)n_tiles + rows) * 4, 256);
      overflow = gcnt + n_tiles;
      chunk_begin = (int32_t*)x; x += nr_align_up(((size_t)n_tiles + 1) * 4, 256);
      chunks = (int4*)x;         x += nr_align_up(max_chunks * 16, 256);
      bucket = (uint32_t*)x;
      hipLaunchKernelGGL(tile_pairs_kernel, dim3((rows + kPairRows - 1) / kPairRows), dim3(kPairThreads),
                         (size_t)n_tiles * 4, st, tiles, tiles_ld, n_keep, rows, n_tiles, tilemap,
                         gcnt, bucket, overflow);
      NR_LAUNCH_CHECK();
      hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, st, gcnt, n_tiles, rows, chunk_begin,
                         (int32_t*)nullptr);
      NR_LAUNCH_CHECK();
    } else {
      gcnt = (int32_t*)x;        x += nr_align_up(2 * (size_t)n_tiles * 4, 256);
      int32_t* cursor = gcnt + n_tiles;
      chunk_begin = (int32_t*)x; x += nr_align_up(((size_t)n_tiles + 1) * 4, 256);
      pair_begin = (int32_t*)x;  x += nr_align_up(((size_t)n_tiles + 1) * 4, 256);
      chunks = (int4*)x;         x += nr_align_up(max_chunks * 16, 256);
      bucket = (uint32_t*)x;
      const dim3 wg((rows + kPairRows - 1) / kPairRows);
      hipLaunchKernelGGL(tile_count_kernel, wg, dim3(256), 0, st, tiles, tiles_ld, n_keep, rows, n_tiles, gcnt);
      NR_LAUNCH_CHECK();
      // (a packed bucket holds every pair of its tile: no cap, no overflow rows)
      hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, st, gcnt, n_tiles, INT_MAX, chunk_begin, pair_begin);
      NR_LAUNCH_CHECK();
      hipLaunchKernelGGL(tile_fill_kernel, wg, dim3(256), 0, st, tiles, tiles_ld, n_keep, rows, n_tiles, pair_begin,
                         cursor, tilemap, bucket);
      NR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(chunk_list_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, st, gcnt, chunk_begin, n_tiles,
                       pair_begin, rows, chunks);
    NR_LAUNCH_CHECK();
    const dim3 pgrid((unsigned)((max_chunks + 3) / 4));
    const int dp = d <= 16 ? 16 : d <= 32 ? 32 : d <= 48 ? 48 : d <= 64 ? 64 : 128;
#define NR_PAIRS_CASE(KS)                                                                                        \
  hipLaunchKernelGGL(rescore_pairs_kernel<KS>, pgrid, dim3(256), 0, st, d_P, ldp, d, qt, (int64_t)ipad, cols,     \
                     d_users, rows, chunk_begin, n_tiles, chunks, bucket, C, cld)
    switch (dp) {
      case 16: NR_PAIRS_CASE(8); break;
      case 32: NR_PAIRS_CASE(16); break;
      case 48: NR_PAIRS_CASE(24); break;
      case 64: NR_PAIRS_CASE(32); break;
      default: NR_PAIRS_CASE(64); break;
    }
#undef NR_PAIRS_CASE
    NR_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(rescore_tiles_kernel, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, d_P, ldp,
                       qt, (int64_t)ipad, d, d_users, rows, cols, tiles, tiles_ld, n_keep, d_tr_indptr,
                       d_tr_indices, C, cld, tilemap);
    NR_LAUNCH_CHECK();
  }
  InvLog2Table tbl;
  for (int i = 0; i < 128; ++i) tbl.v[i] = 1.0 / log2((double)(unsigned)(i + 2));
  if (form != kGroupedNone) {
    // 3 + 4 + 5 in one launch: strikes, ranking of the compact rows, columns -> item ids, boundary check, flags, and the
    // metrics from the ranks while they are still in the wave's registers (r06: metrics_kernel was 21 us + a launch gap)
    if (n_keep * kTileItems <= 1024)
      hipLaunchKernelGGL(rank_compact_kernel<4>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, C, cld, rows, n_keep,
                         top_k, d_users, d_tr_indptr, d_tr_indices, tilemap, tiles, tiles_ld, d_M, mld, d_eps, overflow,
                         d_flag_out, d_truth_indptr, d_truth_indices, mids, tbl, d_out);
    else
      hipLaunchKernelGGL(rank_compact_kernel<8>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, C, cld, rows, n_keep,
                         top_k, d_users, d_tr_indptr, d_tr_indices, tilemap, tiles, tiles_ld, d_M, mld, d_eps, overflow,
                         d_flag_out, d_truth_indptr, d_truth_indices, mids, tbl, d_out);
    NR_LAUNCH_CHECK();
    return NR_OK;
  } else {
    // 3. rank the compact rows (no in-place exact path: a tie needs the full row)
    const int ccols = (int)cld;
    const int sort_len = (2 * top_k < ccols) ? 2 * top_k : ccols;
    hipLaunchKernelGGL(select_rows_kernel<4>, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, C, cld,
                       rows, ccols, sort_len, top_k + 1, w.rank, w.flag, tmask,    // (one item beyond the cut: the tie rule)
                       (int64_t)kRankStride, 1, (int32_t*)nullptr, 0);    // (unranked slots: column 0, a NaN row's)
    NR_LAUNCH_CHECK();
    // 4. columns -> item ids, boundary check, flags
    hipLaunchKernelGGL(remap_rank_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, w.rank, w.flag,
                       tilemap, tiles, tiles_ld, d_M, mld, rows, n_keep, top_k + 1, top_k, C, cld, d_eps, overflow,
                       d_flag_out);
    NR_LAUNCH_CHECK();
  }
  // 5. metrics
  hipLaunchKernelGGL(metrics_kernel, dim3(blocks), dim3(kSelWaves * NR_WAVE), 0, st, w.rank, rows,
                     top_k, d_users, d_truth_indptr, d_truth_indices, mids, tbl, d_out,
                     (int32_t*)nullptr, tmask, d_flag_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_eval_tiles(const float* d_M, int64_t mld, const float* d_P, int64_t ldp,
                     const void* d_gemm_ws, int d, const int32_t* d_users, int rows, int cols,
                     const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                     const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                     const int32_t* metric_ids_host, int n_metric, int top_k, float* d_out,
                     int32_t* d_flag_out, void* d_ws, size_t ws_bytes, void* stream) {
  return eval_tiles_impl(d_M, mld, d_P, ldp, d_gemm_ws, d, d_users, rows, cols, d_tr_indptr, d_tr_indices,
                         d_truth_indptr, d_truth_indices, metric_ids_host, n_metric, top_k, top_k + 1, nullptr,
                         false, d_out, d_flag_out, d_ws, ws_bytes, stream);
}

/* nrhip_eval_tiles for maxima that are only bounded (nrhip_score_filter_tilemax + nrhip_score_tilemax_fix):
 * the n_keep (>= top_k + 1) best tiles are rescored with the fp32 chain; a row stands when its top_k-th rescored
 * score exceeds the largest maximum among the other tiles by more than d_eps[row], else it is flagged.
 * d_eps = NULL: exact maxima (nrhip_score_tilemax), nrhip_eval_tiles' rule, n_keep = top_k + 1.  Either way the
 * rescoring runs grouped by tile (rescore_pairs_kernel) where the shape allows. */
int nrhip_eval_tiles_bounded(const float* d_M, int64_t mld, const float* d_eps, int n_keep, const float* d_P,
                             int64_t ldp, const void* d_gemm_ws, int d, const int32_t* d_users, int rows, int cols,
                             const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                             const int64_t* d_truth_indptr, const int32_t* d_truth_indices,
                             const int32_t* metric_ids_host, int n_metric, int top_k, float* d_out,
                             int32_t* d_flag_out, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_eps || n_keep == top_k + 1, NR_ERR_ARG,
             "eval_tiles_bounded: without a bound the maxima are exact and n_keep is top_k + 1");
  return eval_tiles_impl(d_M, mld, d_P, ldp, d_gemm_ws, d, d_users, rows, cols, d_tr_indptr, d_tr_indices,
                         d_truth_indptr, d_truth_indices, metric_ids_host, n_metric, top_k, n_keep, d_eps, true,
                         d_out, d_flag_out, d_ws, ws_bytes, stream);
}

int nrhip_arg_topk(const float* d_scores, int64_t ld, int rows, int cols, int top_k,
                   int32_t* d_out, int32_t* d_n_exact, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_scores && d_out && d_ws, NR_ERR_ARG, "arg_topk: null pointer argument");
  NR_REQUIRE(top_k >= 1 && top_k <= 256, NR_ERR_UNSUPPORTED, "arg_topk: top_k=%d outside 1..256",
             top_k);
  NR_REQUIRE(cols >= top_k && ld >= cols && rows >= 0, NR_ERR_ARG,
             "arg_topk: cols=%d top_k=%d ld=%lld", cols, top_k, (long long)ld);
  if (rows == 0) return NR_OK;
  NR_REQUIRE(ws_bytes >= eval_ws_bytes(rows), NR_ERR_WORKSPACE,
             "arg_topk: workspace %zu < %zu bytes", ws_bytes, eval_ws_bytes(rows));
  hipStream_t st = (hipStream_t)stream;
  EvalWs w = carve_ws(d_ws, rows);
  int rc = run_selection(d_scores, ld, rows, cols, top_k, top_k, w, st);   // arg_topk.h:22
  if (rc != NR_OK) return rc;
  const int64_t n = (int64_t)rows * top_k;
  hipLaunchKernelGGL(copy_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.rank,
                     rows, top_k, d_out, 0);
  NR_LAUNCH_CHECK();
  if (d_n_exact)
    NR_CHECK_HIP(hipMemcpyAsync(d_n_exact, w.n_exact, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return NR_OK;
}

int nrhip_colsum_workspace_bytes(int rows, int cols, size_t* bytes) {
  NR_REQUIRE(rows >= 0 && cols >= 0 && bytes, NR_ERR_ARG, "colsum_workspace_bytes: bad arguments");
  *bytes = (size_t)((rows + kSlabRows - 1) / kSlabRows + 1) * (size_t)cols * sizeof(double);
  return NR_OK;
}

int nrhip_colsum_f64(const float* d_mat, int64_t ld, int rows, int cols, double* d_out,
                     void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_mat && d_out && d_ws && ld >= cols, NR_ERR_ARG, "colsum_f64: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0 || cols == 0) {
    if (cols) NR_CHECK_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * cols, st));
    return NR_OK;
  }
  const int n_slabs = (rows + kSlabRows - 1) / kSlabRows;
  NR_REQUIRE(ws_bytes >= (size_t)n_slabs * cols * sizeof(double), NR_ERR_WORKSPACE,
             "colsum_f64: workspace too small");
  hipLaunchKernelGGL(colsum_stage1, dim3((cols + 63) / 64, n_slabs), dim3(256), 0, st, d_mat, ld,
                     rows, cols, (double*)d_ws);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_stage2, dim3((cols + 255) / 256), dim3(256), 0, st,
                     (const double*)d_ws, n_slabs, cols, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"

// nrhip_colsum_f64 + the flagged-row counts of the pruned evaluation in its second launch (eval_pipeline.hip)
__attribute__((visibility("hidden"))) int nr_colsum_f64_flags(const float* d_mat, int64_t ld, int rows, int cols,
                                                              double* d_out, void* d_ws, size_t ws_bytes,
                                                              const int32_t* d_flags, double* d_flag_out, void* stream) {
  NR_REQUIRE(d_mat && d_out && d_ws && ld >= cols && rows >= 1 && cols >= 1 && d_flags && d_flag_out, NR_ERR_ARG,
             "colsum_f64_flags: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int n_slabs = (rows + kSlabRows - 1) / kSlabRows;
  NR_REQUIRE(ws_bytes >= (size_t)n_slabs * cols * sizeof(double), NR_ERR_WORKSPACE,
             "colsum_f64_flags: workspace too small");
  hipLaunchKernelGGL(colsum_stage1, dim3((cols + 63) / 64, n_slabs), dim3(256), 0, st, d_mat, ld, rows, cols,
                     (double*)d_ws);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_stage2_flags, dim3((cols + 1023) / 1024 + 1), dim3(1024), 0, st, (const double*)d_ws, n_slabs,
                     cols, d_out, d_flags, rows, d_flag_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

