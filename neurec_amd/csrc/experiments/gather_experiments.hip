// spmm_slab.hip — Y = Â·X on column-slab-major embeddings.
//
// Layout: an [N][d] fp32 table is stored as S = d/SW slabs, slab s holding columns
// [s·SW, (s+1)·SW) of every row contiguously: Xs[s][n][SW].  With SW = 16 a slab of the
// gowalla-shaped table is 70,839 × 64 B = 4.5 MB, and the half of it one class of rows
// gathers from (user rows read item rows and vice versa) is 2.6 / 1.9 MB — it fits the 4 MB
// L2 of one XCD.  The launch pins (slab, row class) pairs to XCDs (block b runs on XCD b % 8),
// so every gather after the first touch of a row is an L2 hit.
//
// Work decomposition: one lane group of SW lanes owns one *entry* (a whole row, or one
// segment of a split hub row); a wave64 carries 64/SW entries at once, so every vector
// instruction (index broadcast, gather, multiply, add) advances 64/SW non-zeros — the
// row-major kernel spent ~5 VALU instructions per non-zero on a single row.  Entries are
// sorted by length within their class at plan time, so the groups of a wave finish together.
// Products and sums are rounded separately, in ascending column order within an entry.
#include "nr_common.h"
#include <algorithm>

namespace {

template <int SW, int WPB, int G>
__global__ __launch_bounds__(WPB* NR_WAVE) void spmm_slab_kernel(
    const int32_t* __restrict__ ent_row, const int64_t* __restrict__ ent_begin,
    const int32_t* __restrict__ ent_len, const int32_t* __restrict__ ent_slot, int n_a, int n_b,
    const int32_t* __restrict__ indices, const float* __restrict__ vals,
    const float* __restrict__ Xs, float* __restrict__ Ys, const float* __restrict__ addend,
    const float* sum_in, float* sum_out, float* __restrict__ partial, int64_t slab_stride, int S) {
  constexpr int GROUPS = NR_WAVE / SW;
  const int wave = threadIdx.x / NR_WAVE, lane = nr_lane();
  const int c = lane % SW, g = lane / SW;
  const int P = n_b > 0 ? 2 * S : S;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  int pair, bip;                                   // (slab, class) pair; block index inside it
  if (P <= 8) {
    const int rep = 8 / P;
    pair = xcd % P;
    bip = j * rep + xcd / P;
  } else {
    const int per = P / 8;
    pair = xcd + 8 * (j % per);
    bip = j / per;
  }
  const int slab = pair % S, cls = pair / S;
  const int n_cls = cls ? n_b : n_a, base = cls ? n_a : 0;
  const int64_t e0 = ((int64_t)bip * WPB + wave) * GROUPS;
  if (e0 >= n_cls) return;
  const bool live = e0 + g < n_cls;
  const int64_t e = base + e0 + (live ? g : 0);
  const int row = ent_row[e];
  const int len = live ? ent_len[e] : 0;
  const int slot = ent_slot[e];
  const int64_t rb = ent_begin[e];
  int maxlen = len;
#pragma unroll
  for (int m = SW; m < NR_WAVE; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m, NR_WAVE));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);

  const float* __restrict__ X = Xs + (int64_t)slab * slab_stride;
  const int gbase = lane & ~(SW - 1);
  float acc = 0.f;
  for (int k0 = 0; k0 < maxlen; k0 += SW) {
    int my_idx = 0;
    float my_val = 0.f;
    if (k0 + c < len) {
      my_idx = indices[rb + k0 + c];
      my_val = vals[rb + k0 + c];
    }
    const int nleft = min(SW, maxlen - k0);
    for (int t0 = 0; t0 < nleft; t0 += G) {
      float a[G], x[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int src = gbase | ((t0 + u) & (SW - 1));
        const int col = __shfl(my_idx, src, NR_WAVE);
        a[u] = __shfl(my_val, src, NR_WAVE);
        x[u] = X[(uint32_t)col * SW + c];           // past the end: col 0, a 0 (row 0 is L2-hot)
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (k0 + t0 + u < len) acc = __fadd_rn(acc, __fmul_rn(a[u], x[u]));
    }
  }
  if (!live) return;
  if (slot >= 0) {
    partial[((int64_t)slot * S + slab) * SW + c] = acc;
    return;
  }
  const int64_t o = (int64_t)slab * slab_stride + (int64_t)row * SW + c;
  float y = acc;
  if (addend) y = __fadd_rn(y, addend[o]);
  if (Ys) Ys[o] = y;
  if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
}

// split rows: add the segment partials in segment order
template <int SW>
__global__ __launch_bounds__(256) void spmm_slab_fix_kernel(
    const int32_t* __restrict__ multi_row, const int32_t* __restrict__ multi_first,
    const int32_t* __restrict__ multi_nseg, int n_multi, const float* __restrict__ partial,
    float* __restrict__ Ys, const float* __restrict__ addend, const float* sum_in, float* sum_out,
    int64_t slab_stride, int S) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per = S * SW;
  if (t >= (int64_t)n_multi * per) return;
  const int m = (int)(t / per), sc = (int)(t % per), slab = sc / SW, c = sc % SW;
  const int first = multi_first[m], nseg = multi_nseg[m];
  float acc = partial[((int64_t)first * S + slab) * SW + c];
  for (int s = 1; s < nseg; ++s)
    acc = __fadd_rn(acc, partial[((int64_t)(first + s) * S + slab) * SW + c]);
  const int64_t o = (int64_t)slab * slab_stride + (int64_t)multi_row[m] * SW + c;
  float y = acc;
  if (addend) y = __fadd_rn(y, addend[o]);
  if (Ys) Ys[o] = y;
  if (sum_out) sum_out[o] = __fadd_rn(sum_in[o], y);
}

template <int SW, int WPB, int G>
int launch_slab(const int32_t* ent_row, const int64_t* ent_begin, const int32_t* ent_len,
                const int32_t* ent_slot, int n_a, int n_b, const int32_t* multi_row,
                const int32_t* multi_first, const int32_t* multi_nseg, int n_multi,
                const int32_t* indices, const float* vals, const float* Xs, int64_t n_rows, int d,
                float* Ys, const float* addend, const float* sum_in, float* sum_out, float* partial,
                hipStream_t st) {
  constexpr int GROUPS = NR_WAVE / SW;
  const int S = d / SW;
  const int P = n_b > 0 ? 2 * S : S;
  const int64_t ents = std::max(n_a, n_b);
  const int64_t bpp = (ents + (int64_t)GROUPS * WPB - 1) / ((int64_t)GROUPS * WPB);   // blocks per pair
  int64_t blocks;
  if (P <= 8) {
    const int rep = 8 / P;
    blocks = 8 * ((bpp + rep - 1) / rep);
  } else {
    blocks = 8 * bpp * (P / 8);
  }
  if (blocks > 0) {
    hipLaunchKernelGGL((spmm_slab_kernel<SW, WPB, G>), dim3((unsigned)blocks), dim3(WPB * NR_WAVE), 0,
                       st, ent_row, ent_begin, ent_len, ent_slot, n_a, n_b, indices, vals, Xs, Ys,
                       addend, sum_in, sum_out, partial, n_rows * SW, S);
    NR_LAUNCH_CHECK();
  }
  if (n_multi > 0) {
    const int64_t threads = (int64_t)n_multi * d;
    hipLaunchKernelGGL(spmm_slab_fix_kernel<SW>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       st, multi_row, multi_first, multi_nseg, n_multi, partial, Ys, addend, sum_in,
                       sum_out, n_rows * SW, S);
    NR_LAUNCH_CHECK();
  }
  return NR_OK;
}

}  // namespace

extern "C" {

/* Experimental entry (round 1): entries and split-row tables are built by the caller. */
int nrhip_spmm_slab(const int32_t* d_ent_row, const int64_t* d_ent_begin, const int32_t* d_ent_len,
                    const int32_t* d_ent_slot, int n_a, int n_b, const int32_t* d_multi_row,
                    const int32_t* d_multi_first, const int32_t* d_multi_nseg, int n_multi,
                    const int32_t* d_indices, const float* d_vals, const float* d_Xs,
                    int64_t n_rows, int d, int slab_width, int waves_per_block, float* d_Ys,
                    const float* d_addend, const float* d_sum_in, float* d_sum_out,
                    float* d_partial, void* stream) {
  NR_REQUIRE(d_ent_row && d_ent_begin && d_ent_len && d_ent_slot && d_indices && d_vals && d_Xs &&
                 (d_Ys || d_sum_out),
             NR_ERR_ARG, "spmm_slab: null pointer argument");
  NR_REQUIRE(d % slab_width == 0 && n_rows * (int64_t)slab_width < (int64_t)1 << 31, NR_ERR_UNSUPPORTED,
             "spmm_slab: d %% slab_width != 0 or slab larger than 2^31 floats");
  hipStream_t st = (hipStream_t)stream;
#define NR_SLAB(SW, WPB, G)                                                                        \
  return launch_slab<SW, WPB, G>(d_ent_row, d_ent_begin, d_ent_len, d_ent_slot, n_a, n_b,          \
                                 d_multi_row, d_multi_first, d_multi_nseg, n_multi, d_indices,      \
                                 d_vals, d_Xs, n_rows, d, d_Ys, d_addend, d_sum_in, d_sum_out,      \
                                 d_partial, st)
  if (slab_width == 8 && waves_per_block == 4) NR_SLAB(8, 4, 8);
  if (slab_width == 16 && waves_per_block == 4) NR_SLAB(16, 4, 16);
  if (slab_width == 16 && waves_per_block == 8) NR_SLAB(16, 8, 16);
  if (slab_width == 16 && waves_per_block == 2) NR_SLAB(16, 2, 16);
  if (slab_width == 32 && waves_per_block == 4) NR_SLAB(32, 4, 16);
  if (slab_width == 64 && waves_per_block == 4) NR_SLAB(64, 4, 16);
#undef NR_SLAB
  NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "spmm_slab: (slab_width %d, waves %d) not built", slab_width,
             waves_per_block);
  return NR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Micro-benchmark (scripts/exp_gather.py): what bounds random row gathers of a [N][64] fp32
// table?  Each wave sums `per_wave` listed rows; VEC floats per lane => 64/(64/VEC) ... i.e.
// VEC = 1: one 256 B row per load instruction; VEC = 2: two rows; VEC = 4: four rows (dwordx4).
// ---------------------------------------------------------------------------------------------
namespace {
template <int VEC, int G>
__global__ __launch_bounds__(256) void exp_gather_kernel(const int32_t* __restrict__ ids,
                                                         int64_t n, int per_wave,
                                                         const float* __restrict__ T,
                                                         float* __restrict__ out) {
  constexpr int LPR = 64 / VEC;             // lanes per row
  constexpr int RPI = NR_WAVE / LPR;        // rows per load instruction
  const int lane = nr_lane();
  const int64_t w = (int64_t)blockIdx.x * 4 + threadIdx.x / NR_WAVE;
  const int64_t b = w * per_wave;
  if (b >= n) return;
  const int len = (int)min((int64_t)per_wave, n - b);
  const int g = lane / LPR, c = lane % LPR;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  for (int k0 = 0; k0 < len; k0 += NR_WAVE) {
    const int my = (k0 + lane < len) ? ids[b + k0 + lane] : 0;
    const int nn = min(NR_WAVE, len - k0);
    for (int t0 = 0; t0 < nn; t0 += G * RPI) {
      float x[G][VEC];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int src = min(t0 + u * RPI + g, nn - 1);
        const int row = (RPI == 1) ? __builtin_amdgcn_readlane(my, min(t0 + u, nn - 1))
                                   : __shfl(my, src, NR_WAVE);
        const float* p = T + (int64_t)row * 64 + c * VEC;
        if constexpr (VEC == 4) {
          const float4 q = *(const float4*)p;
          x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
        } else if constexpr (VEC == 2) {
          const float2 q = *(const float2*)p;
          x[u][0] = q.x; x[u][1] = q.y;
        } else {
          x[u][0] = *p;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += x[u][v];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) s += acc[v];
  out[w * NR_WAVE + lane] = s;
}
}  // namespace

extern "C" int nrhip_exp_gather(const int32_t* d_ids, int64_t n, int per_wave, const float* d_T,
                                int vec, int in_flight, float* d_out, void* stream) {
  NR_REQUIRE(d_ids && d_T && d_out && n > 0 && per_wave > 0, NR_ERR_ARG, "exp_gather: bad arguments");
  const int64_t waves = (n + per_wave - 1) / per_wave;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NR_EXP(V, GG)                                                                             \
  if (vec == V && in_flight == GG) {                                                              \
    hipLaunchKernelGGL((exp_gather_kernel<V, GG>), grid, block, 0, st, d_ids, n, per_wave, d_T, d_out); \
    NR_LAUNCH_CHECK();                                                                            \
    return NR_OK;                                                                                 \
  }
  NR_EXP(1, 16) NR_EXP(1, 8) NR_EXP(2, 8) NR_EXP(2, 16) NR_EXP(4, 4) NR_EXP(4, 8) NR_EXP(4, 16)
#undef NR_EXP
  NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "exp_gather: (vec %d, in_flight %d) not built", vec, in_flight);
  return NR_OK;
}

// Same question for the slab-major layout T[4][N][16]: (slab, class) pairs pinned to XCDs, 64-byte
// row pieces, VEC floats per lane (16/VEC lanes per piece, 64·VEC/16 pieces per load instruction).
// ids[0, half) are the gathers of class A (values in [split, N)), ids[half, 2·half) of class B.
namespace {
template <int VEC, int G>
__global__ __launch_bounds__(256) void exp_gather_slab_kernel(const int32_t* __restrict__ ids,
                                                              int64_t half, int per_wave,
                                                              const float* __restrict__ T, int64_t N,
                                                              float* __restrict__ out) {
  constexpr int LPP = 16 / VEC;             // lanes per 64-byte piece
  constexpr int PPI = NR_WAVE / LPP;        // pieces per load instruction
  const int lane = nr_lane();
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slab = xcd & 3, cls = xcd >> 2;
  const int64_t chunk = (int64_t)j * 4 + threadIdx.x / NR_WAVE;
  if (chunk * per_wave >= half) return;
  const int64_t b = cls * half + chunk * per_wave;
  const int len = (int)min((int64_t)per_wave, half - chunk * per_wave);
  const int g = lane / LPP, c = lane % LPP;
  const float* __restrict__ Ts = T + (int64_t)slab * N * 16;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  for (int k0 = 0; k0 < len; k0 += NR_WAVE) {
    const int my = (k0 + lane < len) ? ids[b + k0 + lane] : 0;
    const int nn = min(NR_WAVE, len - k0);
    for (int t0 = 0; t0 < nn; t0 += G * PPI) {
      float x[G][VEC];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int src = min(t0 + u * PPI + g, nn - 1);
        const int row = __shfl(my, src, NR_WAVE);
        const float* p = Ts + (uint32_t)row * 16 + c * VEC;
        if constexpr (VEC == 4) {
          const float4 q = *(const float4*)p;
          x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
        } else if constexpr (VEC == 2) {
          const float2 q = *(const float2*)p;
          x[u][0] = q.x; x[u][1] = q.y;
        } else {
          x[u][0] = *p;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += x[u][v];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) s += acc[v];
  out[((int64_t)blockIdx.x * 4 + threadIdx.x / NR_WAVE) * NR_WAVE + lane] = s;
}
}  // namespace

extern "C" int nrhip_exp_gather_slab(const int32_t* d_ids, int64_t half, int per_wave,
                                     const float* d_T, int64_t n_rows, int vec, int in_flight,
                                     float* d_out, void* stream) {
  NR_REQUIRE(d_ids && d_T && d_out && half > 0 && per_wave > 0, NR_ERR_ARG, "exp_gather_slab: bad arguments");
  const int64_t chunks = (half + per_wave - 1) / per_wave;
  dim3 grid((unsigned)(8 * ((chunks + 3) / 4))), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NR_EXP(V, GG)                                                                             \
  if (vec == V && in_flight == GG) {                                                              \
    hipLaunchKernelGGL((exp_gather_slab_kernel<V, GG>), grid, block, 0, st, d_ids, half, per_wave, \
                       d_T, n_rows, d_out);                                                       \
    NR_LAUNCH_CHECK();                                                                            \
    return NR_OK;                                                                                 \
  }
  NR_EXP(1, 16) NR_EXP(1, 8) NR_EXP(2, 8) NR_EXP(4, 4) NR_EXP(4, 8) NR_EXP(4, 2)
#undef NR_EXP
  NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "exp_gather_slab: (vec %d, in_flight %d) not built", vec, in_flight);
  return NR_OK;
}

// Column-blocked persistent variant on the ROW-MAJOR [N][64] table: 256 workgroups (one per CU),
// workgroup b on XCD b % 8; XCDs 0-3 gather from rows [split, N), XCDs 4-7 from rows [0, split).
// Every workgroup walks K phases; in phase k all its gathers fall in the k-th column block of its
// class's table, so the XCD's 4 MB L2 only has to hold one block at a time.
// ids layout: [wg][k][per_phase] (per_phase ids per workgroup and phase).
namespace {
template <int VEC, int G, int WAVES>
__global__ __launch_bounds__(WAVES* NR_WAVE) void exp_gather_blocked_kernel(
    const int32_t* __restrict__ ids, int K, int per_phase, const float* __restrict__ T,
    float* __restrict__ out) {
  constexpr int LPR = 64 / VEC;
  constexpr int RPI = NR_WAVE / LPR;
  const int lane = nr_lane(), wave = threadIdx.x / NR_WAVE;
  const int g = lane / LPR, c = lane % LPR;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  const int share = (per_phase + WAVES - 1) / WAVES;
  for (int k = 0; k < K; ++k) {
    const int64_t base = ((int64_t)blockIdx.x * K + k) * per_phase;
    const int lo = wave * share, hi = min(per_phase, lo + share);
    for (int k0 = lo; k0 < hi; k0 += NR_WAVE) {
      const int my = (k0 + lane < hi) ? ids[base + k0 + lane] : 0;
      const int nn = min(NR_WAVE, hi - k0);
      for (int t0 = 0; t0 < nn; t0 += G * RPI) {
        float x[G][VEC];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const int src = min(t0 + u * RPI + g, nn - 1);
          const int row = __shfl(my, src, NR_WAVE);
          const float* p = T + (int64_t)row * 64 + c * VEC;
          if constexpr (VEC == 4) {
            const float4 q = *(const float4*)p;
            x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
          } else {
            x[u][0] = *p;
          }
        }
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += x[u][v];
      }
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) s += acc[v];
  out[((int64_t)blockIdx.x * WAVES + wave) * NR_WAVE + lane] = s;
}
}  // namespace

extern "C" int nrhip_exp_gather_blocked(const int32_t* d_ids, int n_wg, int K, int per_phase,
                                        const float* d_T, int vec, int in_flight, int waves,
                                        float* d_out, void* stream) {
  NR_REQUIRE(d_ids && d_T && d_out && K > 0 && per_phase > 0, NR_ERR_ARG, "exp_gather_blocked: bad arguments");
  hipStream_t st = (hipStream_t)stream;
#define NR_EXP(V, GG, W)                                                                          \
  if (vec == V && in_flight == GG && waves == W) {                                                \
    hipLaunchKernelGGL((exp_gather_blocked_kernel<V, GG, W>), dim3(n_wg), dim3(W * NR_WAVE), 0, st, \
                       d_ids, K, per_phase, d_T, d_out);                                          \
    NR_LAUNCH_CHECK();                                                                            \
    return NR_OK;                                                                                 \
  }
  NR_EXP(1, 16, 16) NR_EXP(1, 8, 16) NR_EXP(4, 4, 16) NR_EXP(4, 8, 16) NR_EXP(4, 4, 8) NR_EXP(1, 16, 8)
  NR_EXP(4, 2, 16)
#undef NR_EXP
  NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "exp_gather_blocked: variant not built");
  return NR_OK;
}

// Micro-benchmark: read-back bandwidth of a [rows][ld] fp32 slab by access pattern.
// WPR waves share one row (WPR = 1: one row stream per wave, the select kernel's pattern;
// WPR = 4: a 256-thread block walks one row).  U 16-byte loads in flight per lane.
namespace {
template <int WPR, int U>
__global__ __launch_bounds__(256) void exp_rowmax_kernel(const float* __restrict__ S, int64_t ld,
                                                         int rows, int cols, float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 4 + wave) / WPR, part = (blockIdx.x * 4 + wave) % WPR;
  if (row >= rows) return;
  const float4* p = reinterpret_cast<const float4*>(S + (int64_t)row * ld);
  const int n4 = cols / 4;
  float m = -INFINITY;
  for (int base = part * U * 64; base < n4; base += WPR * U * 64) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 64 + lane;
      v[u] = i < n4 ? p[i] : make_float4(m, m, m, m);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) m = fmaxf(fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)), m);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64)); m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 8, 64));
  m = fmaxf(m, __shfl_xor(m, 4, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 1, 64));
  if (lane == 0) out[(int64_t)row * WPR + part] = m;
}
}  // namespace

extern "C" int nrhip_exp_rowmax(const float* d_S, int64_t ld, int rows, int cols, int waves_per_row,
                                int in_flight, float* d_out, void* stream) {
  NR_REQUIRE(d_S && d_out && rows > 0 && cols > 0, NR_ERR_ARG, "exp_rowmax: bad arguments");
  const int64_t waves = (int64_t)rows * waves_per_row;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define NR_EXP(W, UU)                                                                             \
  if (waves_per_row == W && in_flight == UU) {                                                    \
    hipLaunchKernelGGL((exp_rowmax_kernel<W, UU>), grid, block, 0, st, d_S, ld, rows, cols, d_out); \
    NR_LAUNCH_CHECK();                                                                            \
    return NR_OK;                                                                                 \
  }
  NR_EXP(1, 4) NR_EXP(1, 8) NR_EXP(1, 16) NR_EXP(4, 4) NR_EXP(4, 8) NR_EXP(2, 8) NR_EXP(4, 2)
#undef NR_EXP
  NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "exp_rowmax: variant not built");
  return NR_OK;
}

// Does a stream of 128-byte gathers that only ever touches the FIRST half of 256-byte rows use all
// L2 channels?  mode 0: always half 0; mode 1: half = row parity (both halves, same line count);
// mode 2: half 1 only.  8 lanes x 16 bytes per gather, 8 gathers per lane group in flight.
namespace {
__global__ __launch_bounds__(256) void exp_halfline_kernel(const int32_t* __restrict__ ids, int64_t n,
                                                           int per_wave, const float4* __restrict__ T,
                                                           int mode, float4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t b = wave * per_wave;
  if (b >= n) return;
  const int len = (int)min((int64_t)per_wave, n - b);
  const int g = lane >> 3, c = lane & 7;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < len; k0 += 64) {
    const int my = (k0 + lane < len) ? ids[b + k0 + lane] : 0;
    float4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = __shfl(my, u * 8 + g, 64);
      const int half = mode == 0 ? 0 : (mode == 1 ? (row & 1) : 1);
      x[u] = T[(int64_t)row * 16 + half * 8 + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
  }
  if (acc.x == 123.456f) out[wave] = acc;
}
}  // namespace

extern "C" int nrhip_exp_halfline(const int32_t* d_ids, int64_t n, int per_wave, const float* d_T, int mode,
                                  float* d_out, void* stream) {
  const int64_t waves = (n + per_wave - 1) / per_wave;
  hipLaunchKernelGGL(exp_halfline_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     d_ids, n, per_wave, (const float4*)d_T, mode, (float4*)d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

// What the fp32 matrix pipe sustains when nothing else happens: `iters` rounds of 128 independent-
// enough v_mfma_f32_32x32x2_f32 (4 accumulators, operands in registers) per wave.
namespace {
typedef float exp_f32x16 __attribute__((ext_vector_type(16)));
template <int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void exp_mfma_peak_kernel(float* out, int iters, float seed) {
  float a[32], b[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) { a[s] = seed + threadIdx.x * 0.001f + s; b[s] = seed - s; }
  exp_f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], a[s], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], b[s], c3, 0, 0, 0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
}  // namespace

extern "C" int nrhip_exp_mfma_peak(int blocks, int waves_per_simd, int iters, float* d_out, void* stream) {
  if (waves_per_simd == 2)
    hipLaunchKernelGGL(exp_mfma_peak_kernel<2>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, d_out, iters, 1.0f);
  else
    hipLaunchKernelGGL(exp_mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_out, iters, 1.0f);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

// ---------------------------------------------------------------------------------------------
// Micro-benchmark (scripts/exp_grid_barrier.py): what does a device-wide barrier cost on this part?  The BPR-MF
// step is one launch per step (12 us, 4.3 us of it the XCDs' launch skew); a persistent epoch kernel would trade the
// launch for a grid barrier per step.  `n_wg` resident workgroups of 256 threads cross `iters` barriers:
//   mode 0  one global counter: thread 0 of every workgroup adds 1 (release), spins until it reads iter·n_wg (acquire)
//   mode 1  two levels: a counter per XCD (blockIdx % 8), the last arriver of an XCD adds 1 to the global one; all
//           spin on the global counter
// `work` = dependent fma steps per thread between barriers (0: the bare barrier).
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void exp_grid_barrier_kernel(unsigned* cnt, int n_wg, int iters, int mode, int work,
                                                               float* out) {
  const int wg = blockIdx.x, xcd = wg & 7;
  const int per_xcd = (n_wg - xcd + 7) / 8;
  float x = threadIdx.x * 1e-3f;
  for (int it = 1; it <= iters; ++it) {
    for (int k = 0; k < work; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (mode == 0) {
        atomicAdd(&cnt[0], 1u);
        while (__hip_atomic_load(&cnt[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it * n_wg) {}
      } else {
        const unsigned prev = atomicAdd(&cnt[16 * (1 + xcd)], 1u);
        if (prev + 1 == (unsigned)it * per_xcd) atomicAdd(&cnt[0], 1u);
        while (__hip_atomic_load(&cnt[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it * 8u) {}
      }
    }
    __syncthreads();
  }
  if (x == 12345.f) out[wg] = x;
}
}  // namespace

extern "C" int nrhip_exp_grid_barrier(unsigned* d_cnt, int n_wg, int iters, int mode, int work, float* d_out,
                                      void* stream) {
  NR_REQUIRE(d_cnt && d_out && n_wg >= 8 && iters >= 1, NR_ERR_ARG, "exp_grid_barrier: bad arguments");
  NR_CHECK_HIP(hipMemsetAsync(d_cnt, 0, 16 * 9 * sizeof(unsigned), (hipStream_t)stream));
  hipLaunchKernelGGL(exp_grid_barrier_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, d_cnt, n_wg, iters, mode,
                     work, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

// ---------------------------------------------------------------------------------------------
// Micro-benchmark (scripts/exp_mfma_valu_overlap.py): do fp32 MFMAs of one wave and VALU work of ANOTHER wave of the
// same SIMD overlap?  (csrc/vae_fused.hip: MFMA time and exp time ADD.)  A workgroup = 8 waves = 2 per SIMD; waves
// 0-3 take role A, waves 4-7 role B:   role 1 = `iters` x 16 dependent v_mfma_f32_32x32x2_f32 (one accumulator);
// role 2 = `iters` x 256 dependent v_fma_f32; role 3 = `iters` x 64 dependent v_exp_f32 (+ 1 fma each); role 0 = idle.
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ float exp_role_work(int role, int iters, float seed) {
  float r = seed;
  if (role == 1) {
    exp_f32x16 c = {0};
    float a = seed, b = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) r += c[i];
  } else if (role == 2) {
    float x = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 256; ++s) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    }
    r = x;
  } else if (role == 3) {
    float x = seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 64; ++s) x = __builtin_fmaf(__builtin_amdgcn_exp2f(x), 1e-3f, -0.5f);
    }
    r = x;
  } else if (role == 4 || role == 5) {
    // 16 bf16 MFMAs 32x32x16 over four independent accumulators; role 5: four independent fmas after every MFMA,
    // issued by the SAME wave (does a wave's VALU work run in the shadow of its own MFMAs?)
    typedef __attribute__((ext_vector_type(8))) __bf16 exp_bf16x8;
    exp_f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    uint4 ua = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    ua.x += (uint32_t)(seed * 3.f);
    uint4 ub = make_uint4(0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u);
    ub.y += (uint32_t)(seed * 5.f);
    if (iters & 1) {                                           // odd iteration count: operands with random mantissa / sign bits
      uint32_t hsh = (uint32_t)(seed * 1e4f) * 2654435761u + threadIdx.x * 40503u;
      auto nxt = [&]() { hsh ^= hsh << 13; hsh ^= hsh >> 17; hsh ^= hsh << 5; return (hsh & 0x807f807fu) | 0x3f003f00u; };
      ua = make_uint4(nxt(), nxt(), nxt(), nxt()); ub = make_uint4(nxt(), nxt(), nxt(), nxt());
    }
    const exp_bf16x8 a = __builtin_bit_cast(exp_bf16x8, ua), b = __builtin_bit_cast(exp_bf16x8, ub);   // distinct registers
    float x0 = seed, x1 = seed + 1.f, x2 = seed + 2.f, x3 = seed + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        if (role == 5) { x0 = __builtin_fmaf(x0, 1.0000001f, 1e-9f); x1 = __builtin_fmaf(x1, 1.0000001f, 1e-9f); x2 = __builtin_fmaf(x2, 1.0000001f, 1e-9f); x3 = __builtin_fmaf(x3, 1.0000001f, 1e-9f); }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        if (role == 5) { x0 = __builtin_fmaf(x0, 1.0000001f, 1e-9f); x1 = __builtin_fmaf(x1, 1.0000001f, 1e-9f); x2 = __builtin_fmaf(x2, 1.0000001f, 1e-9f); x3 = __builtin_fmaf(x3, 1.0000001f, 1e-9f); }
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        if (role == 5) { x0 = __builtin_fmaf(x0, 1.0000001f, 1e-9f); x1 = __builtin_fmaf(x1, 1.0000001f, 1e-9f); x2 = __builtin_fmaf(x2, 1.0000001f, 1e-9f); x3 = __builtin_fmaf(x3, 1.0000001f, 1e-9f); }
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        if (role == 5) { x0 = __builtin_fmaf(x0, 1.0000001f, 1e-9f); x1 = __builtin_fmaf(x1, 1.0000001f, 1e-9f); x2 = __builtin_fmaf(x2, 1.0000001f, 1e-9f); x3 = __builtin_fmaf(x3, 1.0000001f, 1e-9f); }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
    r += x0 + x1 + x2 + x3;
  } else if (role == 7) {
    typedef __attribute__((ext_vector_type(8))) __bf16 exp_bf16x8;
    typedef __attribute__((ext_vector_type(4))) float exp_f32x4;
    exp_f32x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = exp_f32x4{0, 0, 0, 0};
    uint4 ua = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), ub = make_uint4(0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u);
    ua.x += (uint32_t)(seed * 3.f); ub.y += (uint32_t)(seed * 5.f);
    if (iters & 1) {                                           // odd iteration count: operands with random mantissa / sign bits
      uint32_t hsh = (uint32_t)(seed * 1e4f) * 2654435761u + threadIdx.x * 40503u;
      auto nxt = [&]() { hsh ^= hsh << 13; hsh ^= hsh >> 17; hsh ^= hsh << 5; return (hsh & 0x807f807fu) | 0x3f003f00u; };
      ua = make_uint4(nxt(), nxt(), nxt(), nxt()); ub = make_uint4(nxt(), nxt(), nxt(), nxt());
    }
    const exp_bf16x8 a = __builtin_bit_cast(exp_bf16x8, ua), b = __builtin_bit_cast(exp_bf16x8, ub);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  } else if (role == 8 || role == 9) {
    // int8 MFMAs: role 8 = 16 x v_mfma_i32_32x32x32_i8 over four accumulators, role 9 = 32 x v_mfma_i32_16x16x64_i8
    // over eight (twice the MACs of roles 4 / 7 per iteration); operands with random bits when iters is odd
    typedef __attribute__((ext_vector_type(4))) int exp_i32x4;
    typedef __attribute__((ext_vector_type(16))) int exp_i32x16;
    exp_i32x4 a = {0x01020304, 0x05060708, 0x090a0b0c, 0x0d0e0f10}, b = {0x11121314, 0x15161718, 0x191a1b1c, 0x1d1e1f20};
    a[0] += (int)(seed * 3.f); b[1] += (int)(seed * 5.f);
    if (iters & 1) {
      uint32_t hsh = (uint32_t)(seed * 1e4f) * 2654435761u + threadIdx.x * 40503u;
      auto nxt = [&]() { hsh ^= hsh << 13; hsh ^= hsh >> 17; hsh ^= hsh << 5; return (int)hsh; };
      a = exp_i32x4{nxt(), nxt(), nxt(), nxt()}; b = exp_i32x4{nxt(), nxt(), nxt(), nxt()};
    }
    if (role == 8) {
      exp_i32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) r += (float)(c0[i] + c1[i] + c2[i] + c3[i]);
    } else {
      exp_i32x4 c[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = exp_i32x4{0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) r += (float)(c[i][0] + c[i][1] + c[i][2] + c[i][3]);
    }
  } else if (role == 6) {
    float x0 = seed, x1 = seed + 1.f, x2 = seed + 2.f, x3 = seed + 3.f;   // the 64 fmas of role 5 alone
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { x0 = __builtin_fmaf(x0, 1.0000001f, 1e-9f); x1 = __builtin_fmaf(x1, 1.0000001f, 1e-9f); x2 = __builtin_fmaf(x2, 1.0000001f, 1e-9f); x3 = __builtin_fmaf(x3, 1.0000001f, 1e-9f); }
    }
    r = x0 + x1 + x2 + x3;
  }
  return r;
}

__global__ __launch_bounds__(512, 1) void exp_overlap_kernel(int role_a, int role_b, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const float r = exp_role_work(wave < 4 ? role_a : role_b, iters, 1.0f + threadIdx.x * 1e-4f);
  if (r == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = r;
}
}  // namespace

extern "C" int nrhip_exp_overlap(int blocks, int role_a, int role_b, int iters, float* d_out, void* stream) {
  hipLaunchKernelGGL(exp_overlap_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, role_a, role_b, iters, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

// ---------------------------------------------------------------------------------------------
// r05: a persistent BPR-MF epoch kernel confined to ONE XCD — what would its per-step barrier
// cost, and how long does the step's WORK take on an eighth of the chip?  (scripts/exp_one_xcd.py)
//   nrhip_exp_xcd_barrier   8·G blocks are launched; only those that find themselves on XCD `xcd` (HW_REG_XCC_ID)
//                           take part, the others return at once: G workgroups of one XCD cross `iters` barriers
//                           on one counter (release add, acquire spin).  us per barrier = kernel time / iters.
//   nrhip_exp_xcc_histogram which XCD do the blocks of a launch run on (d_hist[8]) — reads back what a CU-masked
//                           stream (hipExtStreamCreateWithCUMask) really selects
//   nrhip_exp_cumask_stream_create / _destroy: a stream whose kernels run on the masked CUs only
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int exp_xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}
__global__ __launch_bounds__(256) void exp_xcd_barrier_kernel(unsigned* cnt, int xcd, int group, int iters, float* out) {
  if (exp_xcc_id() != xcd) return;
  __shared__ int s_member;
  if (threadIdx.x == 0) s_member = (int)atomicAdd(&cnt[1], 1u);           // the first `group` arrivals take part
  __syncthreads();
  if (s_member >= group) return;
  float x = threadIdx.x * 1e-3f;
  for (int it = 1; it <= iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(&cnt[0], 1u);
      // (bounded spin: if fewer than `group` blocks landed on this XCD the kernel must end, not hang the box;
      //  cnt[2] != 0 afterwards says the numbers are void)
      long spins = 0;
      while (__hip_atomic_load(&cnt[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it * group)
        if (++spins > (1L << 22)) { cnt[2] = 1u; break; }
      if (cnt[2]) it = iters;
    }
    __syncthreads();
  }
  if (x == 12345.f) out[blockIdx.x] = x;
}
__global__ void exp_xcc_histogram_kernel(int* hist) {
  if (threadIdx.x == 0) atomicAdd(&hist[exp_xcc_id() & 7], 1);
}
}  // namespace

extern "C" int nrhip_exp_xcd_barrier(unsigned* d_cnt, int xcd, int group, int iters, float* d_out, void* stream) {
  NR_REQUIRE(d_cnt && d_out && xcd >= 0 && xcd < 8 && group >= 1 && group <= 32 && iters >= 1, NR_ERR_ARG,
             "exp_xcd_barrier: bad arguments (group <= 32: one workgroup per CU of the XCD)");
  NR_CHECK_HIP(hipMemsetAsync(d_cnt, 0, 16 * sizeof(unsigned), (hipStream_t)stream));
  hipLaunchKernelGGL(exp_xcd_barrier_kernel, dim3(8 * group), dim3(256), 0, (hipStream_t)stream, d_cnt, xcd, group, iters,
                     d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

extern "C" int nrhip_exp_xcc_histogram(int n_blocks, int* d_hist, void* stream) {
  NR_REQUIRE(d_hist && n_blocks >= 1, NR_ERR_ARG, "exp_xcc_histogram: bad arguments");
  NR_CHECK_HIP(hipMemsetAsync(d_hist, 0, 8 * sizeof(int), (hipStream_t)stream));
  hipLaunchKernelGGL(exp_xcc_histogram_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, d_hist);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

extern "C" int nrhip_exp_cumask_stream_create(const uint32_t* mask, int words, void** stream_out) {
  NR_REQUIRE(mask && words >= 1 && stream_out, NR_ERR_ARG, "exp_cumask_stream_create: bad arguments");
  hipStream_t s = nullptr;
  NR_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *stream_out = (void*)s;
  return NR_OK;
}

extern "C" int nrhip_exp_cumask_stream_destroy(void* stream) {
  if (stream) NR_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return NR_OK;
}
