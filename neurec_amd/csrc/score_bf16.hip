// score_bf16.hip — level 1 of the pruned evaluation as a BOUNDED FILTER on the bf16 matrix cores.
//
// The evaluation's answer is defined by the fp32 k-ascending fmaf chain of every (user, item) score
// (score_gemm.hip; MF.py:120-122, LightGCN.py:187-189) and stays that: the chosen tiles are rescored with that
// chain and ranked from those values (eval_select.hip, nrhip_eval_tiles_bounded).  What this file replaces is only
// the SEARCH for the tiles worth rescoring: v_mfma_f32_32x32x2_f32 issues 64 FLOP / clk / SIMD, the bf16 MFMA 1,024,
// so the tile maxima are taken over a three-term bf16 expansion of the same products
//     x = hi + lo + r,   hi = bf16_rne(x),  lo = bf16_rne(x - hi),  |r| <= 2^-18 |x|
//     u·i ~= Σ_k  uh·ih + uh·il + ul·ih                     (48 MFMAs of 32 clk per 64 x 64 tile instead of 128 of 64)
// and every maximum comes with a bound on its distance from the fp32 chain's value:
//     |approx(u, i) - chain(u, i)| <= kappa(d) · ||u||₂ · max_i ||i||₂ = eps[u]
//     kappa(d) = 1.5 · (3.2 · 2^-18  +  3 d · 2^-23  +  d · 2^-24)
//       dropped terms (ul·il, r·x, x·r)   products of bf16 are exact in fp32; <= 192 fp32 accumulations,   the chain's
//                                          each within 2^-23 of its partial sum (truncation allowed)        own rounding
//     with Σ_k |u_k i_k| <= ||u||₂ ||i||₂.  tests/test_eval_gpu.py measures the left side against this bound.
//     ASSUMED MODEL of the matrix pipe (the one thing here that is measured, not derived): each of the <= 3 d
//     accumulate steps of v_mfma_f32_*_bf16 leaves its partial sum within 2^-23 relative (round-to-nearest or
//     truncation of an exact product-sum) — the 1.5 and the tests' 0.6-of-the-bound ceiling are its margin; values
//     below 2^-126 may flush, which the absolute term of eps covers (r05).
// The ranking kernel accepts a row only if its K-th rescored (exact) score exceeds the largest approximate maximum
// among the tiles it did NOT rescore by more than eps[u]; any other row is flagged and redone from a full fp32 score
// row, exactly as rows with ties are.  No reduced-precision value reaches a ranking or a metric.
//
// Operand layout (built by split_rows_kernel, items once per evaluation, users per batch): for a block b of 32 rows,
// term (0 = hi, 1 = lo), k-step s (16 k each) the wave's 64 lanes read ONE uint4 each — lane 32·g + j holds
// row 32·b + j, k = 16·s + 8·g .. +7 — i.e. exactly its v_mfma_f32_32x32x16_bf16 operand: 1 KB coalesced per load.
// Train items: the planned (user, tile) pairs are recomputed in fp32 by nrhip_score_tilemax_fix afterwards (error 0).
#include "nr_common.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {
  const uint32_t b = __float_as_uint(x);
  if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;        // NaN stays NaN
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

// dst[((b·2 + term)·KS + s)·64 + lane] = the 8 bf16 of row 32·b + (lane & 31), k = 16·s + 8·(lane >> 5) .. +7;
// norm[row] = ||row||₂ (fp32); *max_norm = max over rows (optional); eps[row] = kappa · norm[row] · *other_max (optional)
__global__ void split_rows_kernel(const float* __restrict__ src, int64_t ld, const int32_t* __restrict__ ids, int n,
                                  int d, int ks16, uint4* __restrict__ dst, float* __restrict__ norm,
                                  float* __restrict__ max_norm, const float* __restrict__ other_max, float kappa,
                                  float* __restrict__ eps) {
  __shared__ float s_sq[8][64];
  const int lane = threadIdx.x, s = threadIdx.y, b = blockIdx.x;
  // blocks of 32 rows, k-steps of 16: the operands of v_mfma_f32_32x32x16_bf16
  constexpr int brows = 32, kgroups = 2, rows_log2 = 5;
  const int r = b * brows + (lane & (brows - 1)), k0 = 8 * kgroups * s + 8 * (lane >> rows_log2);
  float x[8];
  const bool have = r < n;
  const float* p = src + (have ? (ids ? (int64_t)ids[r] : (int64_t)r) : 0) * ld;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (have && k0 + e < d) ? p[k0 + e] : 0.f;
  uint32_t hi[8], lo[8];
  float sq = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = bf16_rne_bits(x[e]);
    lo[e] = bf16_rne_bits(x[e] - __uint_as_float(hi[e] << 16));       // x - hi is exact in fp32
    sq = fmaf(x[e], x[e], sq);
  }
  uint4 vh, vl;
  vh.x = hi[0] | (hi[1] << 16); vh.y = hi[2] | (hi[3] << 16); vh.z = hi[4] | (hi[5] << 16); vh.w = hi[6] | (hi[7] << 16);
  vl.x = lo[0] | (lo[1] << 16); vl.y = lo[2] | (lo[3] << 16); vl.z = lo[4] | (lo[5] << 16); vl.w = lo[6] | (lo[7] << 16);
  dst[(((int64_t)b * 2 + 0) * ks16 + s) * 64 + lane] = vh;
  dst[(((int64_t)b * 2 + 1) * ks16 + s) * 64 + lane] = vl;
  s_sq[s][lane] = sq;
  __syncthreads();
  if (s == 0 && lane < brows) {
    float t = 0.f;
    for (int q = 0; q < ks16; ++q)
      for (int g = 0; g < kgroups; ++g) t += s_sq[q][lane + brows * g];
    const float nv = sqrtf(t);
    if (have && norm) norm[r] = nv;
    // the row's bound: kappa ||u|| max ||i||, rounded UP, plus an absolute term for what a relative bound cannot
    // cover — operands / products / partial sums below 2^-126 (a bf16 `lo` part of |x| < 2^-117, a product of two
    // tiny entries) may be flushed to zero by the matrix pipe: each of the <= 6 d flushed quantities is below
    // 2^-126 (times an operand magnitude <= the norms), so 2^-110 · d · (1 + ||u|| + max ||i||) is a generous
    // ceiling; rows of such magnitude cannot be certified against it and take the fp32 path, as they should
    if (have && eps)
      eps[r] = nr_add_up(nr_mul_up(nr_mul_up(kappa, nv), other_max[0]),
                         nr_mul_up(7.70371978e-34f * (float)d, nr_add_up(1.0f, nr_add_up(nv, other_max[0]))));
    if (have && max_norm) atomicMax(reinterpret_cast<int*>(max_norm), __float_as_int(nv));   // nv >= 0 (or NaN: stays)
  }
}

// fmaxf() costs two instructions per call here (v_max_f32 x, x, x to quiet a possible signalling NaN, then the max).
// The bare instructions; a NaN score is dropped by v_max (IEEE mode returns the other operand) as fmaxf would drop it.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max16(const f32x16& c) {
  float m = vmax3(c[0], c[1], c[2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) m = vmax3(m, c[i], c[i + 1]);
  return vmax(m, c[15]);
}
__device__ __forceinline__ float max16_skip(const f32x16& c, uint32_t skip) {
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = ((skip >> i) & 1u) ? m : fmaxf(m, c[i]);
  return m;
}
// max over the two lane halves (rows (reg&3) + 8 (reg>>2) + 4 h of the same 32-row block)
__device__ __forceinline__ float max_halves(float m) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int KS16>
struct BSet {
  uint4 v[2][2][KS16];                                        // [item block X][term][k-step]
};

// One wave: 64 users (two 32-user column blocks, operands in registers for the whole chunk) against the chunk's
// 64-item tiles; the four waves of a workgroup take four user panels against the same tiles (shared in L1 / L2).
// Software pipeline, branch-free: while the 48 MFMAs of tile t + 1 run into one accumulator set, the VALU reduces the
// other set (tile t: 64 accumulator reads, the maxima, the half exchange) and stores it — a wave issues its VALU work
// in the shadow of its own MFMAs (32 clk each), there is no second wave on the SIMD to do it.  Three B register sets
// (loads of tile t + 2 issued before tile t's MFMAs), two accumulator sets: the body is unrolled over 6 phases.
// Stores without a branch: every lane stores one float2 (half 0: user block 0, half 1: user block 1), rows / tiles
// that do not exist go to a per-lane sink.  The partial last tile (pad columns) runs after the loop, unpipelined.
template <int KS16>
__global__ __launch_bounds__(256, 1) void tilemax_bf16_kernel(const uint4* __restrict__ PB,
                                                              const uint4* __restrict__ QB, int bpad, int rows,
                                                              int cols, int n_tiles, float* __restrict__ M,
                                                              int64_t mld, int tiles_per_chunk,
                                                              float* __restrict__ sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int ub0 = (blockIdx.x * 4 + wave) * 2;
  if (ub0 * 32 >= bpad) return;
  const int t_begin = blockIdx.y * tiles_per_chunk;
  const int t_end = min(n_tiles, t_begin + tiles_per_chunk);
  if (t_begin >= t_end) return;
  const int t_stop = min(t_end, cols / 64);                   // full tiles: the pipelined loop

  bf16x8 ah[2][KS16], al[2][KS16];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
      ah[y][s] = __builtin_bit_cast(bf16x8, PB[(((int64_t)(ub0 + y) * 2 + 0) * KS16 + s) * 64 + lane]);
      al[y][s] = __builtin_bit_cast(bf16x8, PB[(((int64_t)(ub0 + y) * 2 + 1) * KS16 + s) * 64 + lane]);
    }
  const int my_row = ub0 * 32 + 32 * h + j;                   // the row this lane stores
  float* const my_sink = sink + 2 * lane;
  float* const my_M = M + (int64_t)my_row * mld;
  const bool row_ok = my_row < rows;

  auto load_b = [&](int t, BSet<KS16>& b) {
    const uint4* q = QB + (int64_t)min(t, t_end - 1) * (4 * KS16 * 64) + lane;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int s = 0; s < KS16; ++s) b.v[x][term][s] = q[((x * 2 + term) * KS16 + s) * 64];
  };
  // MFMA group g of a tile (g = 0 .. 3 KS16 - 1): the four accumulators, one k-step of one term; small terms first
  auto mfma_group = [&](int g, const BSet<KS16>& b, f32x16 (&c)[2][2]) {
    const int term = g / KS16, s = g % KS16;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        if (term == 0)
          c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b.v[x][1][s]), ah[y][s], c[x][y], 0, 0, 0);
        else if (term == 1)
          c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b.v[x][0][s]), al[y][s], c[x][y], 0, 0, 0);
        else
          c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b.v[x][0][s]), ah[y][s], c[x][y], 0, 0, 0);
      }
  };
  auto zero = [&](f32x16 (&c)[2][2]) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[x][y][i] = 0.f;
  };
  // the finished accumulator set of tile t: maxima, half exchange, store
  auto reduce_store = [&](int t, const f32x16 (&c)[2][2]) {
    float m00 = max_halves(max16(c[0][0])), m10 = max_halves(max16(c[1][0]));
    float m01 = max_halves(max16(c[0][1])), m11 = max_halves(max16(c[1][1]));
    const float2 v = h ? make_float2(m01, m11) : make_float2(m00, m10);
    float* dst = (row_ok && t < t_stop) ? my_M + 2 * t : my_sink;
    *reinterpret_cast<float2*>(dst) = v;
  };

  if (t_begin < t_stop) {
    BSet<KS16> b[3];
    f32x16 c[2][2][2];
    load_b(t_begin, b[0]);
    load_b(t_begin + 1, b[1]);
    load_b(t_begin + 2, b[2]);
    zero(c[0]);
#pragma unroll
    for (int g = 0; g < 3 * KS16; ++g) mfma_group(g, b[0], c[0]);
    // phase ph: tile t + ph is finished in c[ph & 1]; tile t + ph + 1 is computed from b[(ph + 1) % 3] into the
    // other set while this one is reduced; b[ph % 3] is free again and takes tile t + ph + 3
    for (int t = t_begin; t < t_stop; t += 6) {
#pragma unroll
      for (int ph = 0; ph < 6; ++ph) {
        load_b(t + ph + 3, b[ph % 3]);
        zero(c[(ph + 1) & 1]);
#pragma unroll
        for (int g = 0; g < 3 * KS16; ++g) mfma_group(g, b[(ph + 1) % 3], c[(ph + 1) & 1]);
        reduce_store(t + ph, c[ph & 1]);
      }
    }
  }
  if (t_stop < t_end) {                                       // the partial last tile: pad columns excluded
    const int t = t_stop, it = t * 64;
    BSet<KS16> b;
    f32x16 c[2][2];
    load_b(t, b);
    zero(c);
#pragma unroll
    for (int g = 0; g < 3 * KS16; ++g) mfma_group(g, b, c);
    float m[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      uint32_t skip = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (it + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * h >= cols) skip |= 1u << reg;
      m[x][0] = max_halves(max16_skip(c[x][0], skip));
      m[x][1] = max_halves(max16_skip(c[x][1], skip));
    }
    if (row_ok) *reinterpret_cast<float2*>(my_M + 2 * t) = h ? make_float2(m[0][1], m[1][1]) : make_float2(m[0][0], m[1][0]);
  }
}

// 64 < d <= 128: the same loop with the k range of a tile in TWO halves ("units": (tile, half)), so that the operand
// sets keep the size of the d = 64 kernel — 128 user-operand registers stay resident, an item set is 64.  A tile's
// accumulators are zeroed before its first unit and reduced after its second, while the first unit of the next tile
// runs into the other set.  Two item sets (the next unit is loaded while this one runs), unrolled over 4 units.
__global__ __launch_bounds__(256, 1) void tilemax_bf16_wide_kernel(const uint4* __restrict__ PB,
                                                                   const uint4* __restrict__ QB, int bpad, int rows,
                                                                   int cols, int n_tiles, float* __restrict__ M,
                                                                   int64_t mld, int tiles_per_chunk,
                                                                   float* __restrict__ sink) {
  constexpr int KS = 4, KT = 8;                               // k-steps of 16 per unit / per tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int ub0 = (blockIdx.x * 4 + wave) * 2;
  if (ub0 * 32 >= bpad) return;
  const int t_begin = blockIdx.y * tiles_per_chunk;
  const int t_end = min(n_tiles, t_begin + tiles_per_chunk);
  if (t_begin >= t_end) return;
  const int t_stop = min(t_end, cols / 64);

  bf16x8 ah[2][KT], al[2][KT];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int s = 0; s < KT; ++s) {
      ah[y][s] = __builtin_bit_cast(bf16x8, PB[(((int64_t)(ub0 + y) * 2 + 0) * KT + s) * 64 + lane]);
      al[y][s] = __builtin_bit_cast(bf16x8, PB[(((int64_t)(ub0 + y) * 2 + 1) * KT + s) * 64 + lane]);
    }
  const int my_row = ub0 * 32 + 32 * h + j;
  float* const my_sink = sink + 2 * lane;
  float* const my_M = M + (int64_t)my_row * mld;
  const bool row_ok = my_row < rows;

  // unit q of the chunk = (tile t_begin + q / 2, k half q & 1)
  auto load_u = [&](int q, BSet<KS>& b) __attribute__((always_inline)) {
    const uint4* p = QB + (int64_t)min(t_begin + (q >> 1), t_end - 1) * (4 * KT * 64) + lane;
    const int kh = q & 1;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int s = 0; s < KS; ++s) b.v[x][term][s] = p[((x * 2 + term) * KT + kh * KS + s) * 64];
  };
  auto mfma_unit = [&](int kh, const BSet<KS>& b, f32x16 (&c)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            if (term == 0)
              c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b.v[x][1][s]), ah[y][kh * KS + s], c[x][y], 0, 0, 0);
            else if (term == 1)
              c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b.v[x][0][s]), al[y][kh * KS + s], c[x][y], 0, 0, 0);
            else
              c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b.v[x][0][s]), ah[y][kh * KS + s], c[x][y], 0, 0, 0);
          }
  };
  auto zero = [&](f32x16 (&c)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[x][y][i] = 0.f;
  };
  auto reduce_store = [&](int t, const f32x16 (&c)[2][2]) __attribute__((always_inline)) {
    float m00 = max_halves(max16(c[0][0])), m10 = max_halves(max16(c[1][0]));
    float m01 = max_halves(max16(c[0][1])), m11 = max_halves(max16(c[1][1]));
    const float2 v = h ? make_float2(m01, m11) : make_float2(m00, m10);
    float* dst = (row_ok && t < t_stop) ? my_M + 2 * t : my_sink;
    *reinterpret_cast<float2*>(dst) = v;
  };

  if (t_begin < t_stop) {
    BSet<KS> b[2];                                            // (three sets spilled: 128 user + 192 item registers)
    f32x16 c[2][2][2];
    load_u(0, b[0]);
    load_u(1, b[1]);
    zero(c[0]);
    mfma_unit(0, b[0], c[0]);
    const int n_units = 2 * (t_stop - t_begin);
    // phase ph: unit q = q0 + ph is done; unit q + 1 runs from b[(ph + 1) & 1]; if q closed its tile (odd), that
    // tile is reduced meanwhile; b[ph & 1] takes unit q + 2.  q0 is a multiple of 4: every index below is static.
    for (int q0 = 0; q0 < n_units; q0 += 4) {
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        load_u(q0 + ph + 2, b[ph & 1]);
        if (((ph + 1) & 1) == 0) zero(c[((ph + 1) >> 1) & 1]);
        mfma_unit((ph + 1) & 1, b[(ph + 1) & 1], c[((ph + 1) >> 1) & 1]);
        if (ph & 1) reduce_store(t_begin + ((q0 + ph) >> 1), c[(ph >> 1) & 1]);
      }
    }
  }
  if (t_stop < t_end) {                                       // the partial last tile: pad columns excluded
    const int t = t_stop, it = t * 64;
    BSet<KS> b;
    f32x16 c[2][2];
    zero(c);
    load_u(2 * (t - t_begin), b);
    mfma_unit(0, b, c);
    load_u(2 * (t - t_begin) + 1, b);
    mfma_unit(1, b, c);
    float m[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      uint32_t skip = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (it + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * h >= cols) skip |= 1u << reg;
      m[x][0] = max_halves(max16_skip(c[x][0], skip));
      m[x][1] = max_halves(max16_skip(c[x][1], skip));
    }
    if (row_ok) *reinterpret_cast<float2*>(my_M + 2 * t) = h ? make_float2(m[0][1], m[1][1]) : make_float2(m[0][0], m[1][0]);
  }
}

inline int padded_dim16(int d) {                              // widths the bf16 filter is built for (128: two k halves)
  const int opts[5] = {16, 32, 48, 64, 128};
  for (int i = 0; i < 5; ++i)
    if (d <= opts[i]) return opts[i];
  return -1;
}
inline int round_up64(int x) { return (x + 63) / 64 * 64; }

struct FilterWs {
  uint4* QB;
  uint4* PB;
  float* unorm;
  float* inorm_max;
  float* sink;                                                // 64 float2: where stores of rows / tiles that do not exist go
  size_t q_bytes, p_bytes, n_bytes, total;
};
FilterWs carve(void* ws, int rows, int cols, int dp) {
  FilterWs f;
  f.q_bytes = nr_align_up((size_t)round_up64(cols) * dp * 4, 256);          // hi + lo bf16 = 4 bytes per element
  f.p_bytes = nr_align_up((size_t)round_up64(rows > 0 ? rows : 1) * dp * 4, 256);
  f.n_bytes = nr_align_up((size_t)round_up64(rows > 0 ? rows : 1) * 4, 256);
  f.QB = (uint4*)ws;
  f.PB = (uint4*)((char*)ws + f.q_bytes);
  f.unorm = (float*)((char*)ws + f.q_bytes + f.p_bytes);
  f.inorm_max = (float*)((char*)ws + f.q_bytes + f.p_bytes + f.n_bytes);
  f.sink = f.inorm_max + 64;
  f.total = f.q_bytes + f.p_bytes + f.n_bytes + 256 + 512;
  return f;
}
inline float kappa_of(int dp) {
  return 1.5f * (3.2f * 3.814697265625e-06f + 3.0f * dp * 1.1920928955078125e-07f + dp * 5.9604644775390625e-08f);
}

}  // namespace

extern "C" {

int nrhip_score_filter_workspace_bytes(int rows, int cols, int d, size_t* bytes) {
  NR_REQUIRE(bytes && rows >= 0 && cols >= 1 && d >= 1, NR_ERR_ARG, "score_filter_workspace_bytes: bad arguments");
  const int dp = padded_dim16(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter: embedding dim %d > 128 not built (use nrhip_score_tilemax)", d);
  *bytes = carve(nullptr, rows, cols, dp).total;
  return NR_OK;
}

int nrhip_score_filter_kappa(int d, float* kappa) {
  NR_REQUIRE(kappa && d >= 1, NR_ERR_ARG, "score_filter_kappa: bad arguments");
  const int dp = padded_dim16(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter: embedding dim %d > 128 not built", d);
  *kappa = kappa_of(dp);
  return NR_OK;
}

int nrhip_score_filter_prepare_items(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws, size_t ws_bytes,
                                     int max_rows, void* stream) {
  NR_REQUIRE(d_Q && d_ws && cols >= 1 && d >= 1 && ldq >= d && max_rows >= 0, NR_ERR_ARG,
             "score_filter_prepare_items: bad arguments");
  const int dp = padded_dim16(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter: embedding dim %d > 128 not built", d);
  FilterWs f = carve(d_ws, max_rows, cols, dp);
  NR_REQUIRE(ws_bytes >= f.total, NR_ERR_WORKSPACE, "score_filter_prepare_items: workspace %zu < %zu", ws_bytes,
             f.total);
  hipStream_t st = (hipStream_t)stream;
  NR_CHECK_HIP(hipMemsetAsync(f.inorm_max, 0, sizeof(float), st));
  const int ks = dp / 16;
  hipLaunchKernelGGL(split_rows_kernel, dim3(round_up64(cols) / 32), dim3(64, ks), 0, st, d_Q, ldq,
                     (const int32_t*)nullptr, cols, d, ks, f.QB, (float*)nullptr, f.inorm_max, (const float*)nullptr,
                     0.f, (float*)nullptr);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Approximate tile maxima M[rows][2*ceil(cols/64)] (32-item tiles, pad columns excluded, train items NOT struck:
 * nrhip_score_tilemax_fix follows) and d_eps[rows]: |M[r][t] - (fp32 chain maximum of that tile)| <= d_eps[r].
 * nrhip_score_filter_prepare_items with the same workspace, cols, d and max_rows >= rows must have run. */
int nrhip_score_filter_tilemax(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols, int d,
                               float* d_M, int64_t mld, float* d_eps, void* d_ws, size_t ws_bytes, int max_rows,
                               void* stream) {
  NR_REQUIRE(d_P && d_M && d_eps && d_ws && cols >= 1 && d >= 1 && ldp >= d && rows >= 0 && max_rows >= rows &&
                 mld >= 2 * ((cols + 63) / 64) && mld % 2 == 0,
             NR_ERR_ARG, "score_filter_tilemax: bad arguments (mld must be even and >= 2*ceil(cols/64))");
  const int dp = padded_dim16(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_filter: embedding dim %d > 128 not built", d);
  if (rows == 0) return NR_OK;
  FilterWs f = carve(d_ws, max_rows, cols, dp);
  NR_REQUIRE(ws_bytes >= f.total, NR_ERR_WORKSPACE, "score_filter_tilemax: workspace %zu < %zu", ws_bytes, f.total);
  hipStream_t st = (hipStream_t)stream;
  const int ks16 = dp / 16, bpad = round_up64(rows);
  hipLaunchKernelGGL(split_rows_kernel, dim3(bpad / 32), dim3(64, ks16), 0, st, d_P, ldp,
                     d_users, rows, d, ks16, f.PB, f.unorm, (float*)nullptr, f.inorm_max, kappa_of(dp), d_eps);
  NR_LAUNCH_CHECK();
  const int bx = (bpad / 64 + 3) / 4;
  const int n_tiles = round_up64(cols) / 64;
  // chunks per user panel: a workgroup's start (64 operand registers per lane + the first tiles: ~3 tile times) and the
  // last, partly filled round of workgroups are what a launch loses; pick the split that minimises
  // rounds x (tiles per chunk + 3) over the 256 CUs (one workgroup per CU).  2,048 small workgroups cost 13 %.
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
#define NR_FILTER_CASE(K)                                                                                       \
  hipLaunchKernelGGL(tilemax_bf16_kernel<K>, grid, block, 0, st, f.PB, f.QB, bpad, rows, cols, n_tiles, d_M, mld, tpc, \
                     f.sink)
  switch (ks16) {
    case 1: NR_FILTER_CASE(1); break;
    case 2: NR_FILTER_CASE(2); break;
    case 3: NR_FILTER_CASE(3); break;
    case 4: NR_FILTER_CASE(4); break;
    default:
      hipLaunchKernelGGL(tilemax_bf16_wide_kernel, grid, block, 0, st, f.PB, f.QB, bpad, rows, cols, n_tiles, d_M, mld,
                         tpc, f.sink);
  }
#undef NR_FILTER_CASE
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
