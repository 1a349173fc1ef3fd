// score_i8.hip — level 1 of the pruned evaluation as a BOUNDED FILTER on the int8 matrix cores (d <= 128).
//
// Same contract as score_bf16.hip: the evaluation's answer is the fp32 k-ascending fmaf chain of every (user, item)
// score (score_gemm.hip; MF.py:120-122, LightGCN.py:187-189); this file only SEARCHES for the tiles worth rescoring
// and hands every approximate tile maximum over together with a bound eps[u] on its distance from the chain's value.
// nrhip_eval_tiles_bounded accepts a row only if that bound certifies the choice; no value computed here is ranked.
//
// Why a second arithmetic: the bf16 filter runs at the package power limit (profiles/r04_exp_score_filter.txt), and
// v_mfma_i32_32x32x32_i8 sustains 2.6 x the MACs/s of v_mfma_f32_32x32x16_bf16 under the same load
// (profiles/r04_exp_mfma_valu_overlap.txt).  Fixed point, 15 bits per entry:
//     u_k = su (qu_k + du_k),  qu = rint(u_k / su) in [-16256, 16256],  su = max_k |u_k| / 16256   (one scale per USER row)
//     i_k = sI (qi_k + di_k),  sI = max over the WHOLE item table / 16256                           (one scale for all items)
//     q = 128 h + l,  h in [-127, 127],  l in [-64, 63]        (two int8 planes)
//     approx(u, i) = su sI 128 · V,   V = 128 Σ hu hi + Σ (hu li + lu hi)      (three int8 products; Σ lu li dropped)
// One item scale, so that the maximum over a tile's items can be taken on the INTEGERS V (v_max3_i32 on the
// accumulators) with one conversion and one fused multiply-add per stored maximum.
// The integer matrix pipe is exact, so the bound is derived, not measured.  For user u and item i
//     |approx - exact dot| <= su sI [ 0.503 Σ|qu| + 0.503 Σ|qi| + 0.2531 d + 64 Σ|lu| ] + 3·2^-24 |approx|
//       (|du|, |di| <= 0.5 + 16256·3·2^-24: the quotient is formed as x · fl(16256 / max);  |lu li| <= 64 |lu|;
//        Qu1 = Σ|qu|, Lu1 = Σ|lu|, Qi1 = Σ|qi| are integers formed while splitting; the int -> float conversion of V,
//        the rounding of su sI and of the final product are the 3·2^-24)
//     |exact dot - fp32 chain| <= 1.5 · d · 2^-24 · ||u||₂ · max_i ||i||₂        (the chain's own rounding, as score_bf16)
// and the item-dependent part goes INTO the stored maximum, per 32-item tile (item norms are heavy-tailed: the
// table-wide max of Qi1 is 4-5 x a typical tile's, and the tiles that decide a certificate are the unpopular ones):
//     M[u][t]  = su sI 128 · max_{i in t} V(u, i)  +  0.525 su sI · max_{i in t} Qi1(i)       an UPPER-bound maximum
//     eps[u]   = su sI [ 0.525 Qu1 + 0.27 d + 64 Lu1 ] + 1.5 d 2^-24 ||u|| max||i|| + score_bf16's absolute term
//     =>  fp32 chain(u, i) <= M[u][t] + eps[u] for every item i of tile t        (what the certificate needs), and
//         M[u][t] - (chain's tile maximum) <= eps[u] + 1.05 su sI max_{i in t} Qi1(i)         (how loose it can be)
//     (0.525 - 0.503 covers the three float roundings of M.)
// Rows (or an item table) whose largest magnitude is outside [2^-40, 2^40], or not finite, get eps = NaN: the
// certificate fails and they take the fp32 path, as rows with ties do.
//
// Operand layout (split_rows_i8_kernel, items once per evaluation, users per batch): block b of 32 rows, plane
// (0 = h, 1 = l), k-step s (32 k each): the wave's 64 lanes read ONE uint4 each — lane 32 g + j holds row 32 b + j,
// k = 32 s + 16 g .. + 15 — its v_mfma_i32_32x32x32_i8 operand.  (Which k a byte of the operand stands for inside the
// instruction does not matter: both operands are built by the same kernel and integer sums do not depend on order.)
#include "nr_common.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

constexpr int kQMax = 16256;                                  // 127 · 128
constexpr uint32_t kBitsLo = 0x2B800000u;                     // 2^-40
constexpr uint32_t kBitsHi = 0x53800000u;                     // 2^40
// scalars of an item table (uint32 slots of the workspace): [0] bits of max |entry|, [2] bits of max_i ||i||₂
constexpr int kScAmax = 0, kScNorm = 2;

// max |entry| of a table as float bits (NaN / inf sort above every finite value).  flat != 0: the rows lie back to
// back (ld == d) on a 16-byte boundary and n·d is a multiple of 4 — one float4 per thread and step, no index arithmetic.
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ src, int64_t ld, int n, int d,
                                                          int flat, uint32_t* __restrict__ out) {
  uint32_t m = 0u;
  const int64_t step = (int64_t)gridDim.x * 256;
  if (flat) {
    const int64_t total4 = (int64_t)n * d / 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += step) {
      const float4 t = s4[e];
      m = max(max(m, __float_as_uint(t.x) & 0x7fffffffu), __float_as_uint(t.y) & 0x7fffffffu);
      m = max(max(m, __float_as_uint(t.z) & 0x7fffffffu), __float_as_uint(t.w) & 0x7fffffffu);
    }
  } else {
    const int64_t total = (int64_t)n * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += step) {
      const int64_t r = e / d;
      m = max(m, __float_as_uint(src[r * ld + (e - r * d)]) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, NR_WAVE));
  // one atomic per BLOCK at most, and only if it can raise the maximum: same-address device atomics cost ~12 ns each
  // (8,192 of them were 95 us of a 10 MB pass); a stale read only costs an atomic that changes nothing
  __shared__ uint32_t s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
  }
}

__device__ __forceinline__ bool scale_bad(uint32_t bits) { return bits != 0u && (bits < kBitsLo || bits > kBitsHi); }

// dst[((b·2 + plane)·KS + s)·64 + lane] = the 16 int8 of row 32·b + (lane & 31), k = 32·s + 16·(lane >> 5) .. +15.
// items (is_items != 0): the scale is the table's (sc[kScAmax], formed before); qblk[b] = max over the block's rows of
//   Σ_k |q_k| (as a float: an integer below 2^21), sc[kScNorm] takes the largest row norm.
// users: per-row scale; cu[2r] = su·sI·128, cu[2r + 1] = 0.525·su·sI rounded up (both 0 for rows that cannot be
//   bounded), eps[r] = the user-side bound (NaN for those).
__global__ void split_rows_i8_kernel(const float* __restrict__ src, int64_t ld, const int32_t* __restrict__ ids, int n,
                                     int d, int ks32, uint4* __restrict__ dst, uint32_t* __restrict__ sc,
                                     float* __restrict__ qblk, float* __restrict__ cu, float* __restrict__ eps,
                                     int is_items) {
  __shared__ uint32_t s_max[4][64];                           // (ks32 <= 4 k-steps of 32: d <= 128)
  __shared__ int s_q1[4][64], s_l1[4][64];
  __shared__ float s_sq[4][64];
  const int lane = threadIdx.x, s = threadIdx.y, b = blockIdx.x;
  const int j = lane & 31, g = lane >> 5;
  const int r = b * 32 + j, k0 = 32 * s + 16 * g;
  const bool have = r < n;
  const float* p = src + (have ? (ids ? (int64_t)ids[r] : (int64_t)r) : 0) * ld;
  float x[16];
  if (have && k0 + 16 <= d && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(p + k0 + 4 * w);
      x[4 * w] = t.x; x[4 * w + 1] = t.y; x[4 * w + 2] = t.z; x[4 * w + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (have && k0 + e < d) ? p[k0 + e] : 0.f;
  }
  uint32_t mb = 0u;
#pragma unroll
  for (int e = 0; e < 16; ++e) mb = max(mb, __float_as_uint(x[e]) & 0x7fffffffu);
  uint32_t bits;
  if (is_items) {
    bits = sc[kScAmax];
  } else {
    s_max[s][lane] = mb;
    __syncthreads();
    bits = 0u;
    for (int q = 0; q < ks32; ++q) bits = max(bits, max(s_max[q][j], s_max[q][j + 32]));
  }
  const bool bad = scale_bad(bits), off = bad || bits == 0u;
  const float amax = __uint_as_float(bits);
  const float inv = off ? 0.f : (float)kQMax / amax;
  int q1 = 0, l1 = 0;
  float sq = 0.f;
  uint32_t hw[4] = {0u, 0u, 0u, 0u}, lw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    int q = off ? 0 : (int)rintf(x[e] * inv);
    q = min(kQMax, max(-kQMax, q));
    const int h = (q + 64) >> 7;                              // floor((q + 64) / 128): |h| <= 127
    const int l = q - (h << 7);                               // in [-64, 63]
    q1 += abs(q);
    l1 += abs(l);
    sq = fmaf(x[e], x[e], sq);
    hw[e >> 2] |= ((uint32_t)h & 0xffu) << (8 * (e & 3));
    lw[e >> 2] |= ((uint32_t)l & 0xffu) << (8 * (e & 3));
  }
  dst[(((int64_t)b * 2 + 0) * ks32 + s) * 64 + lane] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  dst[(((int64_t)b * 2 + 1) * ks32 + s) * 64 + lane] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  s_q1[s][lane] = q1;
  s_l1[s][lane] = l1;
  s_sq[s][lane] = sq;
  __syncthreads();
  if (s != 0) return;                                         // wave 0 finishes the block's 32 rows
  int Q1 = 0, L1 = 0;
  float SQ = 0.f;
  for (int q = 0; q < ks32; ++q) {
    Q1 += s_q1[q][j] + s_q1[q][j + 32];
    L1 += s_l1[q][j] + s_l1[q][j + 32];
    SQ += s_sq[q][j] + s_sq[q][j + 32];
  }
  const float nv = sqrtf(SQ);
  if (is_items) {
    // the block's maxima: Σ|q| to its own slot, the norm through ONE atomic per block, not per row (same-address
    // atomics were 0.1 ms of the item pass)
    int mq = have ? Q1 : 0, mn = have ? __float_as_int(nv) : 0;                  // nv >= 0 (or NaN: stays on top)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      mq = max(mq, __shfl_xor(mq, o, NR_WAVE));
      mn = max(mn, __shfl_xor(mn, o, NR_WAVE));
    }
    if (lane == 0) {
      qblk[b] = (float)mq;
      int* nmp = reinterpret_cast<int*>(&sc[kScNorm]);        // (only if it can raise the maximum: see absmax_bits_kernel)
      if (mn > __hip_atomic_load(nmp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nmp, mn);
    }
    return;
  }
  if (lane >= 32 || !have) return;
  const uint32_t ibits = sc[kScAmax];
  const float inorm = __uint_as_float(sc[kScNorm]);
  if (bad || scale_bad(ibits) || !(nv < __builtin_huge_valf()) || !(inorm < __builtin_huge_valf())) {
    cu[2 * r] = 0.f;
    cu[2 * r + 1] = 0.f;
    eps[r] = __builtin_nanf("");
    return;
  }
  const float su = bits ? amax / (float)kQMax : 0.f;
  const float sI = ibits ? __uint_as_float(ibits) / (float)kQMax : 0.f;
  const float susi = nr_mul_up(su, sI);
  cu[2 * r] = su * sI * 128.f;
  cu[2 * r + 1] = nr_mul_up(0.525f, susi);
  const float bracket = nr_add_up(nr_add_up(nr_mul_up(0.525f, (float)Q1), 0.27f * (float)d), 64.f * (float)L1);
  const float e_fixed = nr_mul_up(susi, bracket);
  const float e_chain = nr_mul_up(nr_mul_up(1.5f * (float)d * 5.9604644775390625e-08f, nv), inorm);
  const float e_abs = nr_mul_up(7.70371978e-34f * (float)d, nr_add_up(1.0f, nr_add_up(nv, inorm)));
  eps[r] = nr_add_up(nr_add_up(e_fixed, e_chain), e_abs);
}

__device__ __forceinline__ int max_halves_i(int m) {
  const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)m, (uint32_t)m, false, false);
  return max((int)r[0], (int)r[1]);
}
// max over a lane's 16 accumulator pairs of V = 128·HH + X
__device__ __forceinline__ int vmax16(const i32x16& hh, const i32x16& xx) {
  int m = max(max((hh[0] << 7) + xx[0], (hh[1] << 7) + xx[1]), (hh[2] << 7) + xx[2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) m = max(max(m, (hh[i] << 7) + xx[i]), (hh[i + 1] << 7) + xx[i + 1]);
  return max(m, (hh[15] << 7) + xx[15]);
}
__device__ __forceinline__ int vmax16_skip(const i32x16& hh, const i32x16& xx, uint32_t skip) {
  int m = INT_MIN;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = ((skip >> i) & 1u) ? m : max(m, (hh[i] << 7) + xx[i]);
  return m;
}

template <int KS>
struct BSetI8 {
  i32x4 v[2][2][KS];                                          // [item block X][plane][k-step]
};

// One wave: 64 users (two 32-user blocks, both planes in registers for the whole chunk) against the chunk's 64-item
// tiles — the structure of tilemax_bf16_kernel: per-wave operand loads, three item sets (tile t + 3 requested before
// tile t + 1's MFMAs), two accumulator sets (tile t reduced on the VALU under the MFMAs of tile t + 1), branch-free
// stores.  A tile is 12 KS MFMAs: per k-step and (item block, user block) one h·h into HH and l·h, h·l into X.
// The instruction order is the compiler's (about seven VALU instructions of the other tile's reduction behind each
// MFMA); a hand-laid order through sched_group_barrier (one MFMA, one item load every third, six VALU) measured
// slower: 0.139 against 0.117 ms per 16,384 users (profiles/r05_exp_filter_i8.txt).
template <int KS>
__global__ __launch_bounds__(256, 1) void tilemax_i8_kernel(const uint4* __restrict__ PB, const uint4* __restrict__ QB,
                                                            const float* __restrict__ cu,
                                                            const float* __restrict__ qblk, int bpad, int rows, int cols,
                                                            int n_tiles, float* __restrict__ M, int64_t mld,
                                                            int tiles_per_chunk, float* __restrict__ sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int ub0 = (blockIdx.x * 4 + wave) * 2;
  if (ub0 * 32 >= bpad) return;
  const int t_begin = blockIdx.y * tiles_per_chunk;
  const int t_end = min(n_tiles, t_begin + tiles_per_chunk);
  if (t_begin >= t_end) return;
  const int t_stop = min(t_end, cols / 64);                   // full tiles: the pipelined loop

  i32x4 ah[2][KS], al[2][KS];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ah[y][s] = __builtin_bit_cast(i32x4, PB[(((int64_t)(ub0 + y) * 2 + 0) * KS + s) * 64 + lane]);
      al[y][s] = __builtin_bit_cast(i32x4, PB[(((int64_t)(ub0 + y) * 2 + 1) * KS + s) * 64 + lane]);
    }
  const int my_row = ub0 * 32 + 32 * h + j;                   // the row this lane stores
  const bool row_ok = my_row < rows;
  const float my_scale = row_ok ? cu[2 * my_row] : 0.f;       // su sI 128
  const float my_tile = row_ok ? cu[2 * my_row + 1] : 0.f;    // 0.525 su sI: times the tile's largest Σ|qi|
  float* const my_sink = sink + 2 * lane;
  float* const my_M = M + (int64_t)(row_ok ? my_row : 0) * mld;

  auto load_b = [&](int t, BSetI8<KS>& b) __attribute__((always_inline)) {
    const uint4* q = QB + (int64_t)min(t, t_end - 1) * (4 * KS * 64) + lane;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < KS; ++s) b.v[x][pl][s] = __builtin_bit_cast(i32x4, q[((x * 2 + pl) * KS + s) * 64]);
  };
  auto mfma_tile = [&](const BSetI8<KS>& b, i32x16 (&hh)[2][2], i32x16 (&xx)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          xx[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.v[x][1][s], ah[y][s], xx[x][y], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          hh[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.v[x][0][s], ah[y][s], hh[x][y], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          xx[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.v[x][0][s], al[y][s], xx[x][y], 0, 0, 0);
    }
  };
  auto zero = [&](i32x16 (&hh)[2][2], i32x16 (&xx)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) { hh[x][y][i] = 0; xx[x][y][i] = 0; }
  };
  auto reduce_store = [&](int t, const i32x16 (&hh)[2][2], const i32x16 (&xx)[2][2]) __attribute__((always_inline)) {
    const int m00 = max_halves_i(vmax16(hh[0][0], xx[0][0])), m10 = max_halves_i(vmax16(hh[1][0], xx[1][0]));
    const int m01 = max_halves_i(vmax16(hh[0][1], xx[0][1])), m11 = max_halves_i(vmax16(hh[1][1], xx[1][1]));
    const int tq = min(t, n_tiles - 1);                        // (tiles past the end are computed into the sink)
    const float2 v = make_float2(fmaf(qblk[2 * tq], my_tile, (float)(h ? m01 : m00) * my_scale),
                                 fmaf(qblk[2 * tq + 1], my_tile, (float)(h ? m11 : m10) * my_scale));
    float* dst = (row_ok && t < t_stop) ? my_M + 2 * t : my_sink;
    *reinterpret_cast<float2*>(dst) = v;
  };

  if (t_begin < t_stop) {
    constexpr int NB = 3;                                     // item sets in flight
    BSetI8<KS> b[NB];
    i32x16 hh[2][2][2], xx[2][2][2];
#pragma unroll
    for (int k = 0; k < NB; ++k) load_b(t_begin + k, b[k]);
    zero(hh[0], xx[0]);
    mfma_tile(b[0], hh[0], xx[0]);
    // phase ph: tile t + ph is finished in set ph & 1; tile t + ph + 1 is computed from b[(ph + 1) % NB] into the other
    // set while this one is reduced; b[ph % NB] is free again and takes tile t + ph + NB
    for (int t = t_begin; t < t_stop; t += 6) {
#pragma unroll
      for (int ph = 0; ph < 6; ++ph) {
        load_b(t + ph + NB, b[ph % NB]);
        zero(hh[(ph + 1) & 1], xx[(ph + 1) & 1]);
        mfma_tile(b[(ph + 1) % NB], hh[(ph + 1) & 1], xx[(ph + 1) & 1]);
        reduce_store(t + ph, hh[ph & 1], xx[ph & 1]);
      }
    }
  }
  if (t_stop < t_end) {                                       // the partial last tile: pad columns excluded
    const int t = t_stop, it = t * 64;
    BSetI8<KS> b;
    i32x16 hh[2][2], xx[2][2];
    load_b(t, b);
    zero(hh, xx);
    mfma_tile(b, hh, xx);
    float m[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      uint32_t skip = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (it + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * h >= cols) skip |= 1u << reg;
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int mi = max_halves_i(vmax16_skip(hh[x][y], xx[x][y], skip));
        m[x][y] = mi == INT_MIN ? -INFINITY : fmaf(qblk[2 * t + x], my_tile, (float)mi * my_scale);   // (-inf: pad columns only)
      }
    }
    if (row_ok) *reinterpret_cast<float2*>(my_M + 2 * t) = h ? make_float2(m[0][1], m[1][1]) : make_float2(m[0][0], m[1][0]);
  }
}

// 64 < d <= 128: the k range of a tile in TWO halves ("units": (tile, half)), as tilemax_bf16_wide_kernel — the user
// operands (64 registers) and two accumulator sets (256) stay resident, an item set is one unit's 32 registers; with
// whole-tile item sets the KS = 4 form spilled (three sets: 119 registers, two: 69).  A tile's accumulators are zeroed
// before its first unit and reduced after its second, while the first unit of the next tile runs into the other set.
__global__ __launch_bounds__(256, 1) void tilemax_i8_wide_kernel(const uint4* __restrict__ PB, const uint4* __restrict__ QB,
                                                                 const float* __restrict__ cu,
                                                                 const float* __restrict__ qblk, int bpad, int rows,
                                                                 int cols, int n_tiles, float* __restrict__ M, int64_t mld,
                                                                 int tiles_per_chunk, float* __restrict__ sink) {
  constexpr int KS = 2, KT = 4;                               // k-steps of 32 per unit / per tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int ub0 = (blockIdx.x * 4 + wave) * 2;
  if (ub0 * 32 >= bpad) return;
  const int t_begin = blockIdx.y * tiles_per_chunk;
  const int t_end = min(n_tiles, t_begin + tiles_per_chunk);
  if (t_begin >= t_end) return;
  const int t_stop = min(t_end, cols / 64);

  i32x4 ah[2][KT], al[2][KT];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int s = 0; s < KT; ++s) {
      ah[y][s] = __builtin_bit_cast(i32x4, PB[(((int64_t)(ub0 + y) * 2 + 0) * KT + s) * 64 + lane]);
      al[y][s] = __builtin_bit_cast(i32x4, PB[(((int64_t)(ub0 + y) * 2 + 1) * KT + s) * 64 + lane]);
    }
  const int my_row = ub0 * 32 + 32 * h + j;
  const bool row_ok = my_row < rows;
  const float my_scale = row_ok ? cu[2 * my_row] : 0.f;
  const float my_tile = row_ok ? cu[2 * my_row + 1] : 0.f;
  float* const my_sink = sink + 2 * lane;
  float* const my_M = M + (int64_t)(row_ok ? my_row : 0) * mld;

  // unit q of the chunk = (tile t_begin + q / 2, k half q & 1)
  auto load_u = [&](int q, BSetI8<KS>& b) __attribute__((always_inline)) {
    const uint4* p = QB + (int64_t)min(t_begin + (q >> 1), t_end - 1) * (4 * KT * 64) + lane;
    const int kh = q & 1;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < KS; ++s) b.v[x][pl][s] = __builtin_bit_cast(i32x4, p[((x * 2 + pl) * KT + kh * KS + s) * 64]);
  };
  auto mfma_unit = [&](int kh, const BSetI8<KS>& b, i32x16 (&hh)[2][2], i32x16 (&xx)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          xx[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.v[x][1][s], ah[y][kh * KS + s], xx[x][y], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          hh[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.v[x][0][s], ah[y][kh * KS + s], hh[x][y], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          xx[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.v[x][0][s], al[y][kh * KS + s], xx[x][y], 0, 0, 0);
    }
  };
  auto zero = [&](i32x16 (&hh)[2][2], i32x16 (&xx)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) { hh[x][y][i] = 0; xx[x][y][i] = 0; }
  };
  auto reduce_store = [&](int t, const i32x16 (&hh)[2][2], const i32x16 (&xx)[2][2]) __attribute__((always_inline)) {
    const int m00 = max_halves_i(vmax16(hh[0][0], xx[0][0])), m10 = max_halves_i(vmax16(hh[1][0], xx[1][0]));
    const int m01 = max_halves_i(vmax16(hh[0][1], xx[0][1])), m11 = max_halves_i(vmax16(hh[1][1], xx[1][1]));
    const int tq = min(t, n_tiles - 1);
    const float2 v = make_float2(fmaf(qblk[2 * tq], my_tile, (float)(h ? m01 : m00) * my_scale),
                                 fmaf(qblk[2 * tq + 1], my_tile, (float)(h ? m11 : m10) * my_scale));
    float* dst = (row_ok && t < t_stop) ? my_M + 2 * t : my_sink;
    *reinterpret_cast<float2*>(dst) = v;
  };

  if (t_begin < t_stop) {
    BSetI8<KS> b[2];
    i32x16 hh[2][2][2], xx[2][2][2];
    load_u(0, b[0]);
    load_u(1, b[1]);
    zero(hh[0], xx[0]);
    mfma_unit(0, b[0], hh[0], xx[0]);
    const int n_units = 2 * (t_stop - t_begin);
    // phase ph: unit q = q0 + ph is done; unit q + 1 runs from b[(ph + 1) & 1]; if q closed its tile (odd), that tile
    // is reduced meanwhile; b[ph & 1] takes unit q + 2.  q0 is a multiple of 4: every index below is static.
    for (int q0 = 0; q0 < n_units; q0 += 4) {
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        load_u(q0 + ph + 2, b[ph & 1]);
        if (((ph + 1) & 1) == 0) zero(hh[((ph + 1) >> 1) & 1], xx[((ph + 1) >> 1) & 1]);
        mfma_unit((ph + 1) & 1, b[(ph + 1) & 1], hh[((ph + 1) >> 1) & 1], xx[((ph + 1) >> 1) & 1]);
        if (ph & 1) reduce_store(t_begin + ((q0 + ph) >> 1), hh[(ph >> 1) & 1], xx[(ph >> 1) & 1]);
      }
    }
  }
  if (t_stop < t_end) {                                       // the partial last tile: pad columns excluded
    const int t = t_stop, it = t * 64;
    BSetI8<KS> b;
    i32x16 hh[2][2], xx[2][2];
    zero(hh, xx);
    load_u(2 * (t - t_begin), b);
    mfma_unit(0, b, hh, xx);
    load_u(2 * (t - t_begin) + 1, b);
    mfma_unit(1, b, hh, xx);
    float m[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      uint32_t skip = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (it + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * h >= cols) skip |= 1u << reg;
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int mi = max_halves_i(vmax16_skip(hh[x][y], xx[x][y], skip));
        m[x][y] = mi == INT_MIN ? -INFINITY : fmaf(qblk[2 * t + x], my_tile, (float)mi * my_scale);
      }
    }
    if (row_ok) *reinterpret_cast<float2*>(my_M + 2 * t) = h ? make_float2(m[0][1], m[1][1]) : make_float2(m[0][0], m[1][0]);
  }
}

inline int padded_dim32(int d) { return d <= 32 ? 32 : (d <= 64 ? 64 : (d <= 128 ? 128 : -1)); }
inline int round_up64(int x) { return (x + 63) / 64 * 64; }

struct FilterWsI8 {
  uint4* QB;
  uint4* PB;
  float* cu;                                                  // [rows][2]
  float* qblk;                                                // [item blocks of 32]
  uint32_t* sc;
  float* sink;                                                // 64 float2: where stores of rows / tiles that do not exist go
  size_t q_bytes, p_bytes, n_bytes, total;
};
FilterWsI8 carve(void* ws, int rows, int cols, int dp) {
  FilterWsI8 f;
  f.q_bytes = nr_align_up((size_t)round_up64(cols) * dp * 2, 256);          // h + l planes = 2 bytes per element
  f.p_bytes = nr_align_up((size_t)round_up64(rows > 0 ? rows : 1) * dp * 2, 256);
  f.n_bytes = nr_align_up((size_t)round_up64(rows > 0 ? rows : 1) * 8, 256);
  const size_t b_bytes = nr_align_up((size_t)(round_up64(cols) / 32) * 4, 256);
  f.QB = (uint4*)ws;
  f.PB = (uint4*)((char*)ws + f.q_bytes);
  f.cu = (float*)((char*)ws + f.q_bytes + f.p_bytes);
  f.qblk = (float*)((char*)ws + f.q_bytes + f.p_bytes + f.n_bytes);
  f.sc = (uint32_t*)((char*)ws + f.q_bytes + f.p_bytes + f.n_bytes + b_bytes);
  f.sink = (float*)(f.sc + 64);
  f.total = f.q_bytes + f.p_bytes + f.n_bytes + b_bytes + 256 + 512;
  return f;
}

}  // namespace

extern "C" {

int nrhip_score_filter_i8_workspace_bytes(int rows, int cols, int d, size_t* bytes) {
  NR_REQUIRE(bytes && rows >= 0 && cols >= 1 && d >= 1, NR_ERR_ARG, "score_filter_i8_workspace_bytes: bad arguments");
  const int dp = padded_dim32(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter_i8: embedding dim %d > 128 not built (use nrhip_score_tilemax)", d);
  *bytes = carve(nullptr, rows, cols, dp).total;
  return NR_OK;
}

int nrhip_score_filter_i8_prepare_items(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws, size_t ws_bytes,
                                        int max_rows, void* stream) {
  NR_REQUIRE(d_Q && d_ws && cols >= 1 && d >= 1 && ldq >= d && max_rows >= 0, NR_ERR_ARG,
             "score_filter_i8_prepare_items: bad arguments");
  const int dp = padded_dim32(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter_i8: embedding dim %d > 128 not built", d);
  FilterWsI8 f = carve(d_ws, max_rows, cols, dp);
  NR_REQUIRE(ws_bytes >= f.total, NR_ERR_WORKSPACE, "score_filter_i8_prepare_items: workspace %zu < %zu", ws_bytes,
             f.total);
  hipStream_t st = (hipStream_t)stream;
  NR_CHECK_HIP(hipMemsetAsync(f.sc, 0, 64 * sizeof(uint32_t), st));
  const int64_t total = (int64_t)cols * d;
  const int flat = ldq == d && ((uintptr_t)d_Q & 15) == 0 && total % 4 == 0;
  const int64_t want = (total / (flat ? 4 : 1) + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 512 ? 512 : want));
  hipLaunchKernelGGL(absmax_bits_kernel, dim3(blocks), dim3(256), 0, st, d_Q, ldq, cols, d, flat, f.sc + kScAmax);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(split_rows_i8_kernel, dim3(round_up64(cols) / 32), dim3(64, dp / 32), 0, st, d_Q, ldq,
                     (const int32_t*)nullptr, cols, d, dp / 32, f.QB, f.sc, f.qblk, (float*)nullptr, (float*)nullptr, 1);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* nrhip_score_filter_tilemax on the int8 matrix cores: M[rows][2*ceil(cols/64)] UPPER-bound tile maxima (32-item
 * tiles, pad columns excluded, train items NOT struck; the tile's share of the quantisation error is inside) and
 * d_eps[rows] with  fp32 chain maximum of tile t <= M[r][t] + d_eps[r]  and  M[r][t] - that maximum <= d_eps[r] +
 * 2 x the tile's share (top of this file); d_eps[r] is NaN for a row the fixed-point form cannot bound (magnitudes
 * outside 2^-40 .. 2^40, non-finite entries) — such rows fail every certificate.
 * nrhip_score_filter_i8_prepare_items with the same workspace, cols, d and max_rows >= rows must have run. */
int nrhip_score_filter_i8_tilemax(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols, int d,
                                  float* d_M, int64_t mld, float* d_eps, void* d_ws, size_t ws_bytes, int max_rows,
                                  void* stream) {
  NR_REQUIRE(d_P && d_M && d_eps && d_ws && cols >= 1 && d >= 1 && ldp >= d && rows >= 0 && max_rows >= rows &&
                 mld >= 2 * ((cols + 63) / 64) && mld % 2 == 0,
             NR_ERR_ARG, "score_filter_i8_tilemax: bad arguments (mld must be even and >= 2*ceil(cols/64))");
  const int dp = padded_dim32(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter_i8: embedding dim %d > 128 not built", d);
  if (rows == 0) return NR_OK;
  FilterWsI8 f = carve(d_ws, max_rows, cols, dp);
  NR_REQUIRE(ws_bytes >= f.total, NR_ERR_WORKSPACE, "score_filter_i8_tilemax: workspace %zu < %zu", ws_bytes, f.total);
  hipStream_t st = (hipStream_t)stream;
  const int ks = dp / 32, bpad = round_up64(rows);
  hipLaunchKernelGGL(split_rows_i8_kernel, dim3(bpad / 32), dim3(64, ks), 0, st, d_P, ldp, d_users, rows, d, ks, f.PB,
                     f.sc, (float*)nullptr, f.cu, d_eps, 0);
  NR_LAUNCH_CHECK();
  const int bx = (bpad / 64 + 3) / 4;
  const int n_tiles = round_up64(cols) / 64;
  // chunks per user panel: as nrhip_score_filter_tilemax — minimise rounds x (tiles per chunk + 3) over the 256 CUs
  int by = 1, tpc = 0;
  {
    int64_t best = -1;
    for (int cand = 1; cand <= 64 && cand <= n_tiles; ++cand) {
      int t = (n_tiles + cand - 1) / cand;
      t = (t + 5) / 6 * 6;                                    // the pipelined body covers 6 tiles
      const int chunks = (n_tiles + t - 1) / t;
      const int64_t rounds = ((int64_t)bx * chunks + 255) / 256;
      const int64_t cost = rounds * (t + 3);
      if (best < 0 || cost < best) { best = cost; by = chunks; tpc = t; }
    }
  }
  dim3 grid(bx, by), block(256);
  if (ks == 1)
    hipLaunchKernelGGL(tilemax_i8_kernel<1>, grid, block, 0, st, f.PB, f.QB, f.cu, f.qblk, bpad, rows, cols, n_tiles, d_M, mld,
                       tpc, f.sink);
  else if (ks == 2)
    hipLaunchKernelGGL(tilemax_i8_kernel<2>, grid, block, 0, st, f.PB, f.QB, f.cu, f.qblk, bpad, rows, cols, n_tiles, d_M, mld,
                       tpc, f.sink);
  else
    hipLaunchKernelGGL(tilemax_i8_wide_kernel, grid, block, 0, st, f.PB, f.QB, f.cu, f.qblk, bpad, rows, cols, n_tiles, d_M,
                       mld, tpc, f.sink);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
