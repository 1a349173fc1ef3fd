// vae.hip — Mult-VAE encoder/decoder kernels (second-model coverage, config 5).
//
// Stands in for the TF graph of model/general_recommender/MultiVAE.py:73-135:
//   q-network  h0 = dropout(l2_normalize(x));  h1 = act(h0·W_q0 + b);  [mu | logvar] = h1·W_q1 + b
//   sample     z = mu + is_training · eps · exp(logvar/2),  eps ~ N(0, 0.01²)
//   p-network  g1 = act(z·W_p0 + b);  logits = g1·W_p1 + b;  log_softmax
//   loss       -mean_b Σ_i logsm·x  +  anneal · mean_b KL_b  (+ 2·reg-term, MultiVAE.py:126-135)
// and for its gradient.  Sizes at the configured p_dim=[16,32]: hidden h = 32, latent z = 16,
// batch B = 512, I = 40,981 items (gowalla).
//
// What is a GEMM and what is not.  x is a multi-hot user row, so h0·W_q0 is NOT a dense
// [B,I]×[I,32] product: it is a bag-sum of ~27 rows of W_q0 per user (one wave per user row,
// straight from the train CSR, no dense [B,I] input is ever built — the reference fills 84 MB
// of it on the host per batch, MultiVAE.py:152-165).  The only dense product with a large
// dimension on both sides is logits = g1·W_p1ᵀ ([B,32]×[32,I]): that one runs on the fp32
// matrix cores through nrhip_score_gemm (W_p1 is stored item-major, [I][32]).  Its two
// gradients read the [B,I] logits once each and run on the matrix cores too (dLoss/dlogits is
// formed in registers, never stored: vae_dwp1_mfma_kernel / vae_dg1_mfma_kernel; r01's VALU
// design — dlogits in place, then one pass per gradient, 160 us per step — left the product in r06).  The 16/32-wide middle layers are register-resident per-row math,
// as in dense.hip.
#include "nr_common.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>

namespace {

constexpr int kMaxD = 32;       // hidden / 2*latent widths up to 32

enum { ACT_TANH = 0, ACT_SIGMOID = 1, ACT_RELU = 2, ACT_IDENTITY = 3 };

__device__ __forceinline__ float act_fwd(int a, float x) {
  if (a == ACT_TANH) return tanhf(x);
  if (a == ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
  if (a == ACT_RELU) return fmaxf(x, 0.f);
  return x;
}
// derivative w.r.t. the pre-activation, from the output y
__device__ __forceinline__ float act_bwd(int a, float y) {
  if (a == ACT_TANH) return 1.0f - y * y;
  if (a == ACT_SIGMOID) return y * (1.0f - y);
  if (a == ACT_RELU) return y > 0.f ? 1.0f : 0.f;
  return 1.0f;
}

__device__ __forceinline__ float uniform01(uint64_t h) {
  return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

// ----------------------------------------------------------------------------
// Encoder + sampling + first decoder layer: one workgroup per batch row.
// Row r of the batch is the item list indices[indptr[rows[r]] .. indptr[rows[r]+1]).
// Lane j < h owns hidden column j.  Saves what the backward pass needs.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_encode_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const int32_t* __restrict__ rows, int batch, int h, int z, const float* __restrict__ Wq0,
    const float* __restrict__ bq0, const float* __restrict__ Wq1, const float* __restrict__ bq1,
    const float* __restrict__ Wp0, const float* __restrict__ bp0, int act, float keep,
    const float* __restrict__ drop_given /* per CSR position, {0,1}, or NULL */,
    const float* __restrict__ eps_given /* [batch][z] or NULL */, float is_training, uint64_t seed,
    uint64_t step, float* __restrict__ h0val /* per CSR position */, float* __restrict__ H1,
    float* __restrict__ MU, float* __restrict__ LOGVAR, float* __restrict__ EPSSTD,
    float* __restrict__ ZS, float* __restrict__ G1, float* __restrict__ KLb) {
  // One workgroup (4 waves) per batch row.  A row's item list is cut into segments of kEncSeg items; wave w sums
  // segments w, w+4, .. (each in ascending item order), the segment sums are added in segment order — a typical row
  // (27 items) is one segment and one wave, exactly the sequential sum; a hub row of thousands of items no longer
  // keeps the whole batch waiting on one wave's chain (r04: 20 -> 9 us at the gowalla shape).
  constexpr int kEncSeg = 256, kEncMaxSeg = 64;
  __shared__ float s_vec[kMaxD];
  __shared__ float s_seg[kEncMaxSeg][kMaxD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x;
  const int64_t u = rows[r];
  const int64_t b = indptr[u], e = indptr[u + 1];
  const int n = (int)(e - b);
  // l2_normalize of a 0/1 row: every non-zero becomes 1/sqrt(max(n, 1e-12))
  const float inv = 1.0f / sqrtf(fmaxf((float)n, 1e-12f));
  const uint64_t drop_key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull));
  // segments longer rows would need beyond kEncMaxSeg are merged into longer ones (same order)
  const int seg_len = max(kEncSeg, (n + kEncMaxSeg - 1) / kEncMaxSeg);
  const int n_seg = max(1, (n + seg_len - 1) / seg_len);
  for (int sg = wave; sg < n_seg; sg += 4) {
  float a1 = 0.f;
  const int64_t sb = b + (int64_t)sg * seg_len, se = min(e, sb + seg_len);
  // 64 (item, value) pairs per chunk, one per lane; then 32 row gathers of W_q0 in flight at a time
  // (the sum stays in ascending item order)
  for (int64_t t0 = sb; t0 < se; t0 += NR_WAVE) {
    const int nn = (int)min((int64_t)NR_WAVE, se - t0);
    int my_item = 0;
    float my_val = 0.f;
    if (lane < nn) {
      const int64_t t = t0 + lane;
      float kp;
      if (drop_given) kp = drop_given[t];
      else kp = (keep >= 1.0f || uniform01(nr::splitmix64(drop_key ^ (uint64_t)t)) < keep) ? 1.f : 0.f;
      my_val = (inv / keep) * kp;                         // x/keep_prob * mask (tf.nn.dropout)
      my_item = indices[t];
      if (h0val) h0val[t] = my_val;
    }
    constexpr int kRowsInFlight = 32;                      // a typical user's whole list in one round trip
    for (int s0 = 0; s0 < nn; s0 += kRowsInFlight) {
      float w[kRowsInFlight];
#pragma unroll
      for (int u = 0; u < kRowsInFlight; ++u) {
        const int item = __shfl(my_item, min(s0 + u, nn - 1), NR_WAVE);
        w[u] = Wq0[(int64_t)item * h + min(lane, h - 1)];    // unconditional; lanes >= h discard it
      }
#pragma unroll
      for (int u = 0; u < kRowsInFlight; ++u) {
        const float v = __shfl(my_val, min(s0 + u, nn - 1), NR_WAVE);
        if (s0 + u < nn) a1 = fmaf(v, w[u], a1);
      }
    }
  }
  if (lane < kMaxD) s_seg[sg][lane] = a1;
  }
  __syncthreads();
  if (wave != 0) return;
  float a1 = 0.f;
  if (lane < kMaxD) {
    a1 = s_seg[0][lane];
    for (int sg = 1; sg < n_seg; ++sg) a1 += s_seg[sg][lane];
  }
  float h1 = 0.f;
  if (lane < h) { h1 = act_fwd(act, a1 + bq0[lane]); H1[(int64_t)r * h + lane] = h1; }
  float* sv = s_vec;
  if (lane < h) sv[lane] = h1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // h2 = h1·W_q1 + b_q1   (2z columns)
  float h2 = 0.f;
  if (lane < 2 * z) {
    h2 = bq1[lane];
    for (int k = 0; k < h; ++k) h2 = fmaf(sv[k], Wq1[k * 2 * z + lane], h2);
  }
  // lanes [0,z): mu ; lanes [z,2z): logvar
  const float logvar = __shfl(h2, lane + z, 64);          // valid for lane < z
  float zs = 0.f, klt = 0.f;
  if (lane < z) {
    const float mu = h2;
    const float sd = expf(0.5f * logvar);
    float eps;
    if (eps_given) eps = eps_given[(int64_t)r * z + lane];
    else {                                                // Box-Muller, N(0, 0.01²)
      const uint64_t k0 = nr::splitmix64(nr::splitmix64(seed ^ 0xabcdull ^ (step << 20)) ^ ((uint64_t)r * z + lane));
      const float u1 = uniform01(k0), u2 = uniform01(nr::splitmix64(k0));
      eps = 0.01f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    }
    const float es = eps * sd;
    zs = mu + is_training * es;
    MU[(int64_t)r * z + lane] = mu;
    LOGVAR[(int64_t)r * z + lane] = logvar;
    EPSSTD[(int64_t)r * z + lane] = es;
    ZS[(int64_t)r * z + lane] = zs;
    klt = 0.5f * (-logvar + expf(logvar) + mu * mu - 1.0f);
  }
  klt = nr_wave_sum_f32(klt);
  if (lane == 0) KLb[r] = klt;
  __builtin_amdgcn_wave_barrier();
  if (lane < z) sv[lane] = zs;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane < h) {
    float a3 = bp0[lane];
    for (int k = 0; k < z; ++k) a3 = fmaf(sv[k], Wp0[k * h + lane], a3);
    G1[(int64_t)r * h + lane] = act_fwd(act, a3);
  }
}

// logits[b][i] += bias[i]  (in place; the GEMM produced g1·W_p1ᵀ)
__global__ __launch_bounds__(256) void add_row_bias_kernel(float* __restrict__ S, int64_t ld,
                                                           int batch, int cols,
                                                           const float* __restrict__ bias) {
  const int64_t n = (int64_t)batch * cols;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    S[r * ld + c] += bias[c];
  }
}

// ----------------------------------------------------------------------------------------------
// Decoder gradient on the matrix cores.  The first design turned the logits into dLoss/dlogits in
// place (read + write 84 MB), then streamed that slab once for dW_p1 and once for dg1 with VALU
// FMAs: 160 us per step.  Here the logits are only READ:
//   vae_softmax_stats_kernel   per row: the positive-item bitmap, (lse, n_b), nll
//   vae_dwp1_mfma_kernel       dW_p1[32-item tile] = Gᵀ·g1 — G = (softmax·n_b − x)/B formed in registers
//                              from the logits as they arrive (lanes over items: the natural, coalesced
//                              layout IS the A operand of v_mfma_f32_32x32x2_f32), rows = contraction
//   vae_dg1_mfma_kernel        dg1[32-row tile] += G·W_p1 over a range of item tiles — the same loads, G
//                              transposed through a wave-local LDS tile (lanes over rows), items = contraction
// Both without workgroup barriers in their loops (a fused kernel — 16 waves = 16 row tiles sharing each
// item tile, dW summed across the waves in LDS per tile — spent its time at its three barriers per tile:
// 115 us).  Partial sums meet in a fixed order: deterministic.
// ----------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kDecTile = 32;
constexpr int kDecTileLd = 33;                   // padded transpose tile
constexpr int kDecWaves = 4;                     // waves per workgroup in both kernels
constexpr int kDg1Ranges = 40;                   // item ranges of the dg1 kernel (= partial sums per output)

__global__ __launch_bounds__(256) void vae_softmax_stats_kernel(
    const float* __restrict__ S, int64_t ld, int cols, const float* __restrict__ bias,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const int32_t* __restrict__ rows, float2* __restrict__ stat_out, float* __restrict__ nll,
    uint32_t* __restrict__ bitmap_ws, int bitmap_words) {
  __shared__ float s_red[256];
  __shared__ float s_lse;
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* srow = S + (int64_t)r * ld;
  const int64_t u = rows[r];
  const int64_t b = indptr[u], e = indptr[u + 1];
  uint32_t* bm = bitmap_ws + (int64_t)r * bitmap_words;
  for (int w = tid; w < bitmap_words; w += 256) bm[w] = 0u;
  __syncthreads();
  for (int64_t t = b + tid; t < e; t += 256) atomicOr(&bm[indices[t] >> 5], 1u << (indices[t] & 31));
  const int cols4 = cols & ~3;
  const float4* srow4 = reinterpret_cast<const float4*>(srow);
  const float4* bias4 = reinterpret_cast<const float4*>(bias);
  float mx = -INFINITY, sum = 0.f;
  auto fold = [&](float x) {
    if (x > mx) { sum = sum * expf(mx - x) + 1.0f; mx = x; }
    else sum += expf(x - mx);
  };
  constexpr int kUn = 8;                                     // independent 16-byte loads per round
  const int n4 = cols4 / 4;
  for (int i0 = tid; i0 < n4; i0 += 256 * kUn) {
    float4 a[kUn], bb[kUn];
#pragma unroll
    for (int k = 0; k < kUn; ++k) {
      const int i = min(i0 + k * 256, n4 - 1);
      a[k] = srow4[i];
      bb[k] = bias4[i];
    }
#pragma unroll
    for (int k = 0; k < kUn; ++k)
      if (i0 + k * 256 < n4) {
        fold(a[k].x + bb[k].x); fold(a[k].y + bb[k].y); fold(a[k].z + bb[k].z); fold(a[k].w + bb[k].w);
      }
  }
  for (int i = cols4 + tid; i < cols; i += 256) fold(srow[i] + bias[i]);
  s_red[tid] = mx;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) s_red[tid] = fmaxf(s_red[tid], s_red[tid + s]); __syncthreads(); }
  const float gmx = s_red[0];
  __syncthreads();
  s_red[tid] = (mx == -INFINITY) ? 0.f : sum * expf(mx - gmx);
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) s_red[tid] += s_red[tid + s]; __syncthreads(); }
  if (tid == 0) s_lse = gmx + logf(s_red[0]);
  __syncthreads();
  const float lse = s_lse;
  float ll = 0.f;                                            // Σ over the row's items of log-softmax
  for (int64_t t = b + tid; t < e; t += 256) {
    const int i = indices[t];
    ll += (srow[i] + bias[i]) - lse;
  }
  s_red[tid] = ll;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) s_red[tid] += s_red[tid + s]; __syncthreads(); }
  if (tid == 0) {
    nll[r] = -s_red[0];
    stat_out[r] = make_float2(lse, (float)(e - b));
  }
}

// G of the 32 x 32 tile (rows row0.., items item0..) in the layout the loads give: lane (hlf, i), slot s
// holds G[row0 + 2s + hlf][item0 + i]
template <class StatOf>      // stat_of(s) = (lse, n_b) of row row0 + 2s + hlf
__device__ __forceinline__ void vae_grad_tile(const float* __restrict__ S, int64_t ld, int batch, int cols,
                                              int row0, int item0, int tile_index, float bias_i,
                                              StatOf&& stat_of, const uint32_t* __restrict__ bitmap,
                                              int bitmap_words, float inv_batch, int lane, float (&v)[16]) {
  const int hlf = lane >> 5, i = lane & 31, item = item0 + i;
  const bool interior = row0 + kDecTile <= batch && item0 + kDecTile <= cols;    // wave-uniform
  float x[16];
  if (interior) {                                    // one base address, constant strides (no clamps)
    const float* p = S + (int64_t)(row0 + hlf) * ld + item;
#pragma unroll
    for (int s = 0; s < 16; ++s) x[s] = p[(int64_t)(2 * s) * ld];    // two 128-byte row segments per instruction
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      x[s] = S[(int64_t)min(row0 + 2 * s + hlf, batch - 1) * ld + min(item, cols - 1)];
  }
  const int brow = row0 + i;                         // lanes j and j + 32 hold row0 + j's bitmap word of the tile
  const uint32_t myword = brow < batch ? bitmap[(int64_t)brow * bitmap_words + tile_index] : 0u;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int row = row0 + 2 * s + hlf;
    const uint32_t word = __shfl(myword, 2 * s + hlf, NR_WAVE);
    const float bit = (float)((word >> i) & 1u);
    const float2 rs = stat_of(s);
    const float l = (x[s] + bias_i) - rs.x;          // log-softmax
    const float g = (expf(l) * rs.y - bit) * inv_batch;
    v[s] = (interior || (row < batch && item < cols)) ? g : 0.f;
  }
}

// dW_p1 tile [32 items][h] and db_p1: workgroup = one item tile, wave q takes row tiles q, q+4, ...
__global__ __launch_bounds__(kDecWaves* NR_WAVE) void vae_dwp1_mfma_kernel(
    const float* __restrict__ S, int64_t ld, int batch, int cols, int h, const float* __restrict__ bias,
    const float2* __restrict__ stat, float inv_batch, const uint32_t* __restrict__ bitmap, int bitmap_words,
    const float* __restrict__ G1, float* __restrict__ dWp1, float* __restrict__ dbp1) {
  __shared__ float s_c[kDecWaves][16 * NR_WAVE];
  __shared__ float s_cs[kDecWaves][kDecTile];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hlf = lane >> 5, i = lane & 31;
  const int t = blockIdx.x, item0 = t * kDecTile;
  const float b = item0 + i < cols ? bias[item0 + i] : 0.f;
  f32x16 c2 = {0};
  float cs = 0.f;
  const int n_rt = (batch + kDecTile - 1) / kDecTile;
  for (int rt = wave; rt < n_rt; rt += kDecWaves) {
    const int row0 = rt * kDecTile;
    float2 st[16];
    float hb[16];
    if (row0 + kDecTile <= batch && h == kDecTile) {         // full row tile, full width: constant strides
      const float2* sp = stat + row0 + hlf;
      const float* gp = G1 + (int64_t)(row0 + hlf) * kDecTile + i;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        st[s] = sp[2 * s];
        hb[s] = gp[2 * s * kDecTile];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int row = row0 + 2 * s + hlf;
        st[s] = stat[min(row, batch - 1)];
        hb[s] = (row < batch && i < h) ? G1[(int64_t)min(row, batch - 1) * h + min(i, h - 1)] : 0.f;
      }
    }
    float v[16];
    vae_grad_tile(S, ld, batch, cols, row0, item0, t, b, [&](int s2) { return st[s2]; }, bitmap, bitmap_words,
                  inv_batch, lane, v);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      cs += v[s];
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[s], hb[s], c2, 0, 0, 0);   // A = G(item, row), B = g1(row, col)
    }
  }
  cs += __shfl_xor(cs, 32, NR_WAVE);
  if (hlf == 0) s_cs[wave][i] = cs;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) s_c[wave][reg * NR_WAVE + lane] = c2[reg];
  __syncthreads();
  for (int e = tid; e < 16 * NR_WAVE; e += kDecWaves * NR_WAVE) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < kDecWaves; ++w) sum += s_c[w][e];
    const int reg = e >> 6, ln = e & 63;
    const int m = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), n = ln & 31;   // C/D map: row m (item), col n
    if (item0 + m < cols && n < h) dWp1[(int64_t)(item0 + m) * h + n] = sum;
  }
  if (tid < kDecTile && item0 + tid < cols) {
    float bs = 0.f;
#pragma unroll
    for (int w = 0; w < kDecWaves; ++w) bs += s_cs[w][tid];
    dbp1[item0 + tid] = bs;
  }
}

// dg1 partial [32 rows][h] of one item range: workgroup = (row tile, item range), wave q takes the
// range's tiles q, q+4, ...
__global__ __launch_bounds__(kDecWaves* NR_WAVE) void vae_dg1_mfma_kernel(
    const float* __restrict__ S, int64_t ld, int batch, int cols, int h, const float* __restrict__ bias,
    const float2* __restrict__ stat, float inv_batch, const uint32_t* __restrict__ bitmap, int bitmap_words,
    const float* __restrict__ Wp1, float* __restrict__ part, int tiles_per_range, int rows_pad) {
  __shared__ float s_tile[kDecWaves][kDecTile * kDecTileLd];
  __shared__ float s_c[kDecWaves][16 * NR_WAVE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hlf = lane >> 5, i = lane & 31;
  const int row0 = blockIdx.x * kDecTile;
  const int n_tiles = (cols + kDecTile - 1) / kDecTile;
  const int t_begin = blockIdx.y * tiles_per_range, t_end = min(n_tiles, t_begin + tiles_per_range);
  __shared__ float2 s_stat[kDecTile];                          // (lse, n_b) of the workgroup's row tile
  if (tid < kDecTile) s_stat[tid] = stat[min(row0 + tid, batch - 1)];
  __syncthreads();
  float* tile = s_tile[wave];
  f32x16 c1 = {0};
  for (int t = t_begin + wave; t < t_end; t += kDecWaves) {
    const int item0 = t * kDecTile;
    const float b = item0 + i < cols ? bias[item0 + i] : 0.f;
    float wb[16];                                   // W_p1 rows of the tile: 128-byte rows, two per instruction
    if (item0 + kDecTile <= cols && h == kDecTile) {
      const float* wp = Wp1 + (int64_t)(item0 + hlf) * kDecTile + i;
#pragma unroll
      for (int k = 0; k < 16; ++k) wb[k] = wp[2 * k * kDecTile];
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int it = item0 + 2 * k + hlf;
        wb[k] = (it < cols && i < h) ? Wp1[(int64_t)min(it, cols - 1) * h + min(i, h - 1)] : 0.f;
      }
    }
    float v[16];
    vae_grad_tile(S, ld, batch, cols, row0, item0, t, b, [&](int s2) { return s_stat[2 * s2 + hlf]; }, bitmap,
                  bitmap_words, inv_batch, lane, v);
    // transpose through the wave's LDS tile: lanes over rows, items become the contraction index
#pragma unroll
    for (int s = 0; s < 16; ++s) tile[(2 * s + hlf) * kDecTileLd + i] = v[s];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float gt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) gt[k] = tile[i * kDecTileLd + 2 * k + hlf];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k)
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[k], wb[k], c1, 0, 0, 0);   // A = G(row, item), B = W(item, col)
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) s_c[wave][reg * NR_WAVE + lane] = c1[reg];
  __syncthreads();
  for (int e = tid; e < 16 * NR_WAVE; e += kDecWaves * NR_WAVE) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < kDecWaves; ++w) sum += s_c[w][e];
    const int reg = e >> 6, ln = e & 63;
    const int m = (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5), n = ln & 31;   // row m, col n
    part[((int64_t)blockIdx.y * rows_pad + row0 + m) * kMaxD + n] = sum;
  }
}

// the item ranges' partial sums added in range order; 8 loads in flight per thread
__global__ __launch_bounds__(256) void vae_dg1_reduce_n_kernel(const float* __restrict__ part, int n_parts,
                                                               int rows_pad, int batch, int h,
                                                               float* __restrict__ dG1) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= batch * h) return;
  const int r = idx / h, j = idx % h;
  const float* p = part + (int64_t)r * kMaxD + j;
  const int64_t stride = (int64_t)rows_pad * kMaxD;
  float sum = 0.f;
  for (int k0 = 0; k0 < n_parts; k0 += 8) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = p[(int64_t)min(k0 + k, n_parts - 1) * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k0 + k < n_parts) sum += t[k];
  }
  dG1[idx] = sum;
}

// ----------------------------------------------------------------------------
// Middle of the backward pass, one wave per batch row:
//   da3 = dg1·act'(g1);  dz = da3·W_p0ᵀ;  dmu = dz + anneal·mu/B;
//   dlogvar = dz·(eps·std)/2 + anneal·(exp(logvar)−1)/(2B);  dh2 = [dmu | dlogvar];
//   da1 = (dh2·W_q1ᵀ)·act'(h1)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_mid_bwd_kernel(
    int batch, int h, int z, int act, float anneal, float inv_batch, const float* __restrict__ dG1,
    const float* __restrict__ G1, const float* __restrict__ H1, const float* __restrict__ MU,
    const float* __restrict__ LOGVAR, const float* __restrict__ EPSSTD,
    const float* __restrict__ Wp0, const float* __restrict__ Wq1, float* __restrict__ DA3,
    float* __restrict__ DH2, float* __restrict__ DA1) {
  __shared__ float s_vec[4][kMaxD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= batch) return;
  float* sv = s_vec[wave];
  float da3 = 0.f;
  if (lane < h) {
    da3 = dG1[(int64_t)r * h + lane] * act_bwd(act, G1[(int64_t)r * h + lane]);
    DA3[(int64_t)r * h + lane] = da3;
    sv[lane] = da3;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float dh2 = 0.f;
  if (lane < 2 * z) {
    const int k = lane < z ? lane : lane - z;
    float dz = 0.f;
    for (int j = 0; j < h; ++j) dz = fmaf(sv[j], Wp0[k * h + j], dz);       // da3·W_p0ᵀ
    if (lane < z) {
      dh2 = dz + anneal * MU[(int64_t)r * z + k] * inv_batch;
    } else {
      dh2 = dz * EPSSTD[(int64_t)r * z + k] * 0.5f +
            anneal * 0.5f * (expf(LOGVAR[(int64_t)r * z + k]) - 1.0f) * inv_batch;
    }
    DH2[(int64_t)r * 2 * z + lane] = dh2;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < 2 * z) sv[lane] = dh2;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane < h) {
    float acc = 0.f;
    for (int j = 0; j < 2 * z; ++j) acc = fmaf(sv[j], Wq1[lane * 2 * z + j], acc);   // dh2·W_q1ᵀ
    DA1[(int64_t)r * h + lane] = acc * act_bwd(act, H1[(int64_t)r * h + lane]);
  }
}

// Small weight gradients, reduced over the batch: out[k][j] = Σ_b X[b][k]·G[b][j] and
// bias[j] = Σ_b G[b][j].  One wave per output element (K·J + J <= 1056), lanes stride the batch.
__global__ __launch_bounds__(256) void vae_small_wgrad_kernel(const float* __restrict__ X, int K,
                                                              const float* __restrict__ G, int J,
                                                              int batch, float* __restrict__ dW,
                                                              float* __restrict__ db) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= K * J + J) return;
  float acc = 0.f;
  if (o < K * J) {
    const int k = o / J, j = o % J;
    for (int b = lane; b < batch; b += NR_WAVE) acc = fmaf(X[(int64_t)b * K + k], G[(int64_t)b * J + j], acc);
    acc = nr_wave_sum_f32(acc);
    if (lane == 0) dW[o] = acc;
  } else {
    const int j = o - K * J;
    for (int b = lane; b < batch; b += NR_WAVE) acc += G[(int64_t)b * J + j];
    acc = nr_wave_sum_f32(acc);
    if (lane == 0) db[j] = acc;
  }
}

// the three small weight gradients of the middle of the backward pass in ONE launch (they depend on the same
// producer and on nothing of each other; three launches cost three boundaries for ~1,600 wave-sized jobs)
struct SmallWgradJob { const float* X; int K; const float* G; int J; float* dW; float* db; int first_block; };
__global__ __launch_bounds__(256) void vae_small_wgrad3_kernel(SmallWgradJob j0, SmallWgradJob j1, SmallWgradJob j2,
                                                               int batch) {
  const SmallWgradJob& jb = (int)blockIdx.x >= j2.first_block ? j2 : (int)blockIdx.x >= j1.first_block ? j1 : j0;
  const int o = ((int)blockIdx.x - jb.first_block) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int K = jb.K, J = jb.J;
  if (o >= K * J + J) return;
  float acc = 0.f;
  if (o < K * J) {
    const int k = o / J, j = o % J;
    for (int b = lane; b < batch; b += NR_WAVE) acc = fmaf(jb.X[(int64_t)b * K + k], jb.G[(int64_t)b * J + j], acc);
    acc = nr_wave_sum_f32(acc);
    if (lane == 0) jb.dW[o] = acc;
  } else {
    const int j = o - K * J;
    for (int b = lane; b < batch; b += NR_WAVE) acc += jb.G[(int64_t)b * J + j];
    acc = nr_wave_sum_f32(acc);
    if (lane == 0) jb.db[j] = acc;
  }
}

// dW_q0[item][:] += h0[b][item]·da1[b][:] over the batch's CSR entries (scatter, fp32 atomics).
// grid (batch row, 16 chunk slots): a wave takes 16-item chunks of ITS row, slot + 64 apart — a long row is spread over
// up to 64 waves instead of one wave issuing its atomics one item after the other (the first form: 512 waves, the
// longest row's ~1,000 atomics in a chain, 25.6 us) — and with h <= 32 the two wave halves take two items at a time.
constexpr int kDwq0Chunk = 16, kDwq0SlotsY = 16;
__global__ __launch_bounds__(256) void vae_dwq0_kernel(const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const int32_t* __restrict__ rows, int batch,
                                                       int h, const float* __restrict__ h0val,
                                                       const float* __restrict__ DA1,
                                                       float* __restrict__ dWq0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x;
  const int64_t u = rows[r];
  const int64_t b = indptr[u], e = indptr[u + 1];
  const int n = (int)(e - b);
  const int per = h <= 32 ? 2 : 1;                          // items per wave iteration
  const int col = per == 2 ? (lane & 31) : lane, half = per == 2 ? (lane >> 5) : 0;
  const float g = col < h ? DA1[(int64_t)r * h + col] : 0.f;
  for (int c = blockIdx.y * 4 + wave; c * kDwq0Chunk < n; c += 4 * kDwq0SlotsY) {
    const int64_t t0 = b + (int64_t)c * kDwq0Chunk;
    const int nn = (int)min((int64_t)kDwq0Chunk, e - t0);
    int my_item = 0;
    float my_val = 0.f;
    if (lane < nn) { my_item = indices[t0 + lane]; my_val = h0val[t0 + lane]; }
    for (int s = 0; s < nn; s += per) {
      const int idx = s + half;
      const int item = __shfl(my_item, idx & 63, NR_WAVE);
      const float val = __shfl(my_val, idx & 63, NR_WAVE);
      if (idx < nn && col < h) atomicAdd(&dWq0[(int64_t)item * h + col], val * g);
    }
  }
}

// block 0: mean of x -> out[0]; block 1: mean of y -> out[1] (same arithmetic as mean_kernel)
__global__ __launch_bounds__(256) void mean2_kernel(const float* __restrict__ x, const float* __restrict__ y, int n,
                                                    float* __restrict__ out) {
  __shared__ double s[256];
  const float* src = blockIdx.x == 0 ? x : y;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)src[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(s[0] / (double)n);
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, int n,
                                                   float* __restrict__ out) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)x[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) *out = (float)(s[0] / (double)n);
}

// y += a·x  (the 2·reg·W term of the regulariser's gradient)
__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x,
                                                   float* __restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = fmaf(a, x[i], y[i]);
}

// out += Σ x²  (double accumulation; one atomic per block)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n,
                                                    double* __restrict__ out) {
  __shared__ double s[256];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    acc += (double)x[i] * (double)x[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(out, s[0]);
}

}  // namespace

extern "C" {

int nrhip_vae_encode(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows,
                     int batch, int h, int z, const float* d_Wq0, const float* d_bq0,
                     const float* d_Wq1, const float* d_bq1, const float* d_Wp0,
                     const float* d_bp0, int act, float keep, const float* d_drop_given,
                     const float* d_eps_given, float is_training, uint64_t seed, uint64_t step,
                     float* d_h0val, float* d_H1, float* d_MU, float* d_LOGVAR, float* d_EPSSTD,
                     float* d_ZS, float* d_G1, float* d_KLb, void* stream) {
  NR_REQUIRE(d_indptr && d_indices && d_rows && d_Wq0 && d_bq0 && d_Wq1 && d_bq1 && d_Wp0 &&
                 d_bp0 && d_H1 && d_MU && d_LOGVAR && d_EPSSTD && d_ZS && d_G1 && d_KLb,
             NR_ERR_ARG, "vae_encode: null pointer argument");
  NR_REQUIRE(h >= 1 && h <= kMaxD && z >= 1 && 2 * z <= kMaxD, NR_ERR_UNSUPPORTED,
             "vae_encode: hidden %d / latent %d outside the built range (h <= 32, 2z <= 32)", h, z);
  NR_REQUIRE(act >= 0 && act <= 3 && keep > 0.f && keep <= 1.f && batch >= 0, NR_ERR_ARG,
             "vae_encode: bad activation / keep / batch");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_encode_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream,
                     d_indptr, d_indices, d_rows, batch, h, z, d_Wq0, d_bq0, d_Wq1, d_bq1, d_Wp0,
                     d_bp0, act, keep, d_drop_given, d_eps_given, is_training, seed, step, d_h0val,
                     d_H1, d_MU, d_LOGVAR, d_EPSSTD, d_ZS, d_G1, d_KLb);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_add_row_bias(float* d_S, int64_t ld, int batch, int cols, const float* d_bias,
                       void* stream) {
  NR_REQUIRE(d_S && d_bias && ld >= cols && batch >= 0 && cols >= 1, NR_ERR_ARG,
             "add_row_bias: bad arguments");
  if (batch == 0) return NR_OK;
  int64_t blocks = ((int64_t)batch * cols + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_row_bias_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, d_S, ld, batch, cols, d_bias);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_vae_workspace_bytes(int batch, int cols, size_t* bytes) {
  NR_REQUIRE(bytes && batch >= 0 && cols >= 1, NR_ERR_ARG, "vae_workspace_bytes: bad arguments");
  const size_t rows = (size_t)(batch > 0 ? batch : 1);
  // positive-item bit rows of the softmax gradient, then the dg1 partial sums (per item range) and the per-row
  // softmax statistics
  const size_t rows_pad = (rows + kDecTile - 1) / kDecTile * kDecTile;
  const size_t parts = (size_t)kDg1Ranges * rows_pad;
  *bytes = nr_align_up(rows * (size_t)((cols + 31) / 32) * sizeof(uint32_t), 256) +
           parts * kMaxD * sizeof(float) + nr_align_up(rows * sizeof(float2), 256);
  return NR_OK;
}

/* d_S holds g1·W_p1ᵀ (no bias) on entry and is only read. */
int nrhip_vae_decoder_loss_grad(float* d_S, int64_t ld, int batch, int cols, int h,
                                const float* d_bp1, const int64_t* d_indptr,
                                const int32_t* d_indices, const int32_t* d_rows,
                                const float* d_G1, const float* d_Wp1, float* d_nll,
                                float* d_dWp1, float* d_dbp1, float* d_dG1, void* d_ws,
                                size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_S && d_bp1 && d_indptr && d_indices && d_rows && d_G1 && d_Wp1 && d_nll && d_dWp1 &&
                 d_dbp1 && d_dG1 && d_ws && ld >= cols && batch >= 1 && cols >= 1,
             NR_ERR_ARG, "vae_decoder_loss_grad: bad arguments");
  NR_REQUIRE(h >= 1 && h <= kMaxD, NR_ERR_UNSUPPORTED, "vae_decoder: hidden %d > 32", h);
  const int words = (cols + 31) / 32;
  const size_t bits_bytes = nr_align_up((size_t)batch * words * sizeof(uint32_t), 256);
  float* part = (float*)((char*)d_ws + bits_bytes);
  hipStream_t st = (hipStream_t)stream;
  {
    // statistics, then one read of the logits per gradient
    const int rows_pad = (batch + kDecTile - 1) / kDecTile * kDecTile;
    const size_t parts_bytes = (size_t)kDg1Ranges * rows_pad * kMaxD * sizeof(float);
    NR_REQUIRE(ws_bytes >= bits_bytes + parts_bytes + nr_align_up((size_t)batch * sizeof(float2), 256),
               NR_ERR_WORKSPACE, "vae_decoder_loss_grad: workspace too small (nrhip_vae_workspace_bytes)");
    float2* stat = (float2*)((char*)part + parts_bytes);
    hipLaunchKernelGGL(vae_softmax_stats_kernel, dim3(batch), dim3(256), 0, st, d_S, ld, cols, d_bp1, d_indptr,
                       d_indices, d_rows, stat, d_nll, (uint32_t*)d_ws, words);
    NR_LAUNCH_CHECK();
    const int n_tiles = (cols + kDecTile - 1) / kDecTile;
    const float inv_b = 1.0f / (float)batch;
    hipLaunchKernelGGL(vae_dwp1_mfma_kernel, dim3(n_tiles), dim3(kDecWaves * NR_WAVE), 0, st, d_S, ld, batch, cols,
                       h, d_bp1, stat, inv_b, (const uint32_t*)d_ws, words, d_G1, d_dWp1, d_dbp1);
    NR_LAUNCH_CHECK();
    const int per_range = (n_tiles + kDg1Ranges - 1) / kDg1Ranges;
    const int n_ranges = (n_tiles + per_range - 1) / per_range;
    hipLaunchKernelGGL(vae_dg1_mfma_kernel, dim3(rows_pad / kDecTile, n_ranges), dim3(kDecWaves * NR_WAVE), 0, st,
                       d_S, ld, batch, cols, h, d_bp1, stat, inv_b, (const uint32_t*)d_ws, words, d_Wp1, part,
                       per_range, rows_pad);
    NR_LAUNCH_CHECK();
    hipLaunchKernelGGL(vae_dg1_reduce_n_kernel, dim3((batch * h + 255) / 256), dim3(256), 0, st, part, n_ranges,
                       rows_pad, batch, h, d_dG1);
    NR_LAUNCH_CHECK();
    return NR_OK;
  }
}

int nrhip_vae_mid_backward(int batch, int h, int z, int act, float anneal, const float* d_dG1,
                           const float* d_G1, const float* d_H1, const float* d_MU,
                           const float* d_LOGVAR, const float* d_EPSSTD, const float* d_ZS,
                           const float* d_Wp0, const float* d_Wq1, float* d_DA3, float* d_DH2,
                           float* d_DA1, float* d_dWp0, float* d_dbp0, float* d_dWq1,
                           float* d_dbq1, float* d_dbq0, void* stream) {
  NR_REQUIRE(d_dG1 && d_G1 && d_H1 && d_MU && d_LOGVAR && d_EPSSTD && d_ZS && d_Wp0 && d_Wq1 &&
                 d_DA3 && d_DH2 && d_DA1 && d_dWp0 && d_dbp0 && d_dWq1 && d_dbq1 && d_dbq0 &&
                 batch >= 1,
             NR_ERR_ARG, "vae_mid_backward: bad arguments");
  NR_REQUIRE(h >= 1 && h <= kMaxD && z >= 1 && 2 * z <= kMaxD, NR_ERR_UNSUPPORTED,
             "vae_mid_backward: widths outside the built range");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vae_mid_bwd_kernel, dim3((batch + 3) / 4), dim3(256), 0, st, batch, h, z, act,
                     anneal, 1.0f / (float)batch, d_dG1, d_G1, d_H1, d_MU, d_LOGVAR, d_EPSSTD,
                     d_Wp0, d_Wq1, d_DA3, d_DH2, d_DA1);
  NR_LAUNCH_CHECK();
  // dW_p0 = zsᵀ·da3 [z][h];  dW_q1 = h1ᵀ·dh2 [h][2z];  db_q0 = Σ da1 — one launch, three block ranges
  const int b0 = (z * h + h + 3) / 4, b1 = (h * 2 * z + 2 * z + 3) / 4, b2 = (h + 3) / 4;
  const SmallWgradJob j0{d_ZS, z, d_DA3, h, d_dWp0, d_dbp0, 0};
  const SmallWgradJob j1{d_H1, h, d_DH2, 2 * z, d_dWq1, d_dbq1, b0};
  const SmallWgradJob j2{d_H1, 0, d_DA1, h, nullptr, d_dbq0, b0 + b1};
  hipLaunchKernelGGL(vae_small_wgrad3_kernel, dim3(b0 + b1 + b2), dim3(256), 0, st, j0, j1, j2, batch);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* dW_q0 (dense [n_items][h], zero on entry) += scatter of the batch. */
int nrhip_vae_dwq0(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows,
                   int batch, int h, const float* d_h0val, const float* d_DA1, float* d_dWq0,
                   void* stream) {
  NR_REQUIRE(d_indptr && d_indices && d_rows && d_h0val && d_DA1 && d_dWq0 && batch >= 0 &&
                 h >= 1 && h <= kMaxD,
             NR_ERR_ARG, "vae_dwq0: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_dwq0_kernel, dim3(batch, kDwq0SlotsY), dim3(256), 0, (hipStream_t)stream,
                     d_indptr, d_indices, d_rows, batch, h, d_h0val, d_DA1, d_dWq0);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_axpy(float a, const float* d_x, float* d_y, int64_t n, void* stream) {
  NR_REQUIRE(d_x && d_y && n >= 0, NR_ERR_ARG, "axpy: bad arguments");
  if (n == 0) return NR_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, d_x,
                     d_y, n);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* *d_out += sum(x*x) in fp64 (zero it first for a plain sum of squares). */
int nrhip_sumsq_accumulate(const float* d_x, int64_t n, double* d_out, void* stream) {
  NR_REQUIRE(d_x && d_out && n >= 0, NR_ERR_ARG, "sumsq: bad arguments");
  if (n == 0) return NR_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, n,
                     d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* d_out[0] = mean(d_x[0..n)), d_out[1] = mean(d_y[0..n)) in one launch (the step's two loss terms) */
int nrhip_mean2_f32(const float* d_x, const float* d_y, int n, float* d_out, void* stream) {
  NR_REQUIRE(d_x && d_y && d_out && n >= 1, NR_ERR_ARG, "mean2_f32: bad arguments");
  hipLaunchKernelGGL(mean2_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, d_x, d_y, n, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_mean_f32(const float* d_x, int n, float* d_out, void* stream) {
  NR_REQUIRE(d_x && d_out && n >= 1, NR_ERR_ARG, "mean_f32: bad arguments");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_x, n, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
