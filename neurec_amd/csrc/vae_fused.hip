// vae_fused.hip — Mult-VAE decoder loss + gradients WITHOUT a logits slab (MultiVAE.py:104-124).
//
//   logits = g1·W_p1ᵀ + b_p1            [B][I]   (B = 512, I = 40,981, contraction 32)
//   loss   = -mean_b Σ_i log_softmax(logits)_bi · x_bi
//   G      = dLoss/dlogits = (softmax · n_b − x) / B
//   dW_p1  = Gᵀ·g1   [I][32],   db_p1 = colsum(G),   dg1 = G·W_p1   [B][32]
//
// The first form (vae.hip: nrhip_vae_decoder_loss_grad) wrote the 84 MB slab once and read it three times
// (statistics, dW_p1, dg1): 336 MB of HBM traffic for operands that are 5.3 MB.  The contraction is 32 deep,
// so a 32 x 32 logits tile costs 17 fp32 MFMAs — cheaper to recompute than to move.  Two passes over the
// (row tile, item tile) grid, both holding every logit in accumulator registers only:
//
//   pass 1  vae_dec_stats_kernel   logits tile -> running (max, Σexp) per row; per-workgroup partials ->
//                                  vae_dec_stat_kernel -> (lse, n_b) and nll per row  (vae_dec_rows_kernel before
//                                  it: positives' bitmap and positives' logit sum, a wave per row)
//   pass 2  vae_dec_grad_kernel    the same tile again -> G in registers -> dg1 += G·W (the accumulator layout IS
//                                  the A operand: no transpose), G through a wave-local LDS tile -> dW_p1 += Gᵀ·g1,
//                                  db_p1 += colsum;  vae_dg1_reduce_wg_kernel adds the workgroups' dg1 partials
//
// Orientation: the tile is C[item][row] = W_tile · g1ᵀ, so a lane holds ONE batch row (lane & 31) and 16 items
// — row statistics are in-lane folds, and the C registers feed the dg1 MFMA directly (contraction index =
// the item a register holds).  The bias rides as a 17th k-step (A = b_i, B = 1): fmaf(1, b, dot) = dot + b
// rounded once, bit for bit the `matmul + bias` of the slab form; logits are the same k-ascending fmaf chain
// as nrhip_score_gemm's (tests/test_multivae_gpu.py::test_fused_decoder_equals_the_slab_form: torch.equal).
//
// Work split: a workgroup = 8 waves x a contiguous range of item tiles, whose W tiles are staged in LDS in ONE
// round of loads per group.  Pass 1: 8 waves = 8 row tiles, no barrier inside a group.  Pass 2: 8 waves x RT row
// tiles = all rows of a 256·RT-row chunk; g1 sits in LDS (both operand layouts are read from the one copy); dW_p1
// of an item tile is the sum of the 8 waves' partial tiles, added in wave order (every wave adds two accumulator
// registers' worth) behind the ONE barrier per item tile; dg1 accumulates in registers over the workgroup's item
// range and leaves as one partial per workgroup.  No atomics on floats: every sum has a fixed order.
//
// Measured (MI355X, B = 512, I = 40,981, h = 32; profiles/r04_exp_vae_fused.txt): MFMA time and VALU time ADD in
// these kernels (knock-outs: pass 1 per item tile 2.6 us MFMA-only + 2.0 us exp-only vs 3.95 us together), so the
// exps were cut to the bone where the analysis allows (pass 1) and the MFMA count is the floor of pass 2.
#include "nr_common.h"
#include <atomic>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kT = 32;                 // tile edge
constexpr int kLd = 33;                // padded LDS row
constexpr int kWaves = 8;              // waves per workgroup
constexpr int kThreads = kWaves * NR_WAVE;
constexpr int kScratch = kT * kLd;     // floats per wave-local tile (1056 = 16·64 partial + 32 column sums)

// item (pass-2 contraction index) held by accumulator register j of a lane in half `hlf`
__device__ __forceinline__ int c_row(int j, int hlf) { return (j & 3) + 8 * (j >> 2) + 4 * hlf; }

// exp(x) for x <= 0 (x - max, log-softmax): ocml's expf algorithm — x·log2(e) in two pieces, v_exp_f32 on the
// fraction, ldexp by the integer part — without its overflow / underflow selects (two compares into SGPR pairs
// and two selects per call: with 32 calls per tile the compiler spilled SGPRs into VGPR lanes around them).
// Arguments below -104 (result under the smallest denormal) are clamped there: the result is 0.
__device__ __forceinline__ float exp_nonpos(float x) {
  const float xc = fmaxf(x, -104.0f);
  const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;       // log2(e) head and tail
  const float ph = xc * c;
  const float pl = __builtin_fmaf(xc, cc, __builtin_fmaf(xc, c, -ph));
  const float e = __builtin_rintf(ph);
  const float a = (ph - e) + pl;
  return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}

// e^(x - ref) = 2^(x·log2e - ref·log2e): v_exp_f32 on one fma (both passes).  Pass 1's Σ exp:  The single rounding of the
// exponent costs a term a relative error of 6e-8·|x - ref|·log2e — weighted by the term itself, |t|·e^t <= 1/e, so
// the SUM (>= 1: the maximum's own term) keeps a relative error under 1e-7 and lse an absolute one under 1e-7; the
// full-precision form above costs 11 instructions per logit against 2, and MFMA and VALU time add up here.
__device__ __forceinline__ float exp2_scaled(float x_log2e_minus_ref) {
  return __builtin_amdgcn_exp2f(x_log2e_minus_ref);
}
constexpr float kLog2e = 0x1.715476p+0f;

struct FusedArgs {
  const float* G1;        // [batch][h]
  const float* Wp1;       // [cols][h]  (item-major)
  const float* bp1;       // [cols]
  const uint32_t* bitmap; // [batch][words]   positives of every batch row (pass 2)
  int batch, cols, h, words, n_tiles, tiles_per_wg, rows_pad;
  int main_tiles, n_rem, groups_per_tile;   // whole tiles dealt to workgroups; left-over tiles cut by rows (see pass 2)
  float* xpart;                             // [n_rem·groups_per_tile][kScratch] partial dW tiles of the left-over tiles
  // pass 1 out: per-workgroup partial statistics, SoA [wg][2][rows_pad]
  float* pstat;
  // pass 2 in
  const float2* stat;     // (lse, n_b) per row
  float inv_batch;
  // pass 2 out
  float* dWp1; float* dbp1;       // [cols][h], [cols]
  float* part;                    // [wg][rows_pad][32]   dg1 partials
  float* dbg_logits;              // tests only: [batch][cols] copy of pass 1's logits (NULL in the product path)
  // vae_dec_rows_kernel: positives' bitmap and logit sum
  const int64_t* indptr; const int32_t* indices; const int32_t* rows;
  uint32_t* bitmap_out; float* posll;
  int n_wg;
};

// W tiles [n][kT][kLd] (+ bias in column 32; pad items: bias -inf so their logit is -inf) of the item tiles
// t0 .. t0+n-1, every load of the group in flight before the first LDS store: ONE memory round trip per group
// (W was last written by the optimiser: the first touch comes from HBM/MALL, ~2 us — a per-tile fetch a tile ahead
// left that latency exposed on every tile).
template <int MAXT, int BATCH>
__device__ __forceinline__ void stage_w_tiles(const FusedArgs& a, int t0, int n, float* __restrict__ Wt, int tid) {
  static_assert((2 * MAXT) % BATCH == 0, "whole batches");
#pragma unroll
  for (int i0 = 0; i0 < 2 * MAXT; i0 += BATCH) {
    float v[BATCH];
    // (loads unconditional on clamped positions, masked at the store: a load behind a condition is waited for before
    //  the next one is issued, and the "one round trip per group" above was BATCH of them)
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int e = (i0 + i) * kThreads + tid, tt = e >> 10, it = (t0 + tt) * kT + ((e >> 5) & 31), k = e & 31;
      v[i] = a.Wp1[(int64_t)min(it, a.cols - 1) * a.h + min(k, a.h - 1)];
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int e = (i0 + i) * kThreads + tid, tt = e >> 10, it = (t0 + tt) * kT + ((e >> 5) & 31), k = e & 31;
      Wt[(e >> 10) * kScratch + ((e >> 5) & 31) * kLd + (e & 31)] = (tt < n && it < a.cols && k < a.h) ? v[i] : 0.f;
    }
  }
  if (tid < MAXT * kT) {
    const int tt = tid >> 5, it = (t0 + tt) * kT + (tid & 31);
    Wt[tt * kScratch + (tid & 31) * kLd + 32] = (tt < n && it < a.cols) ? a.bp1[it] : -INFINITY;
  }
}

// Per batch row: its positives as bits (bitmap[r][w] bit b <=> item 32·w + b is in the row's CSR list, pass 2's
// `x`) and the sum of its positives' logits — each the k-ascending fmaf chain + bias of the MFMA tiles, so pass 1
// carries neither the bitmap nor a select per logit.  Workgroup per row, thread per positive; the row's sum is a
// fixed tree.  (Measured alternatives: a wave per row, 21 us — rows with hundreds of positives serialise; riding
// in pass 1's grid as extra blocks lengthened that kernel by 11 us.)
__global__ __launch_bounds__(256) void vae_dec_rows_kernel(const FusedArgs a) {
  __shared__ float s_g[kT];
  __shared__ float s_l[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  uint32_t* bm = a.bitmap_out + (int64_t)r * a.words;
  for (int w = tid; w < a.words; w += 256) bm[w] = 0u;
  if (tid < kT) s_g[tid] = tid < a.h ? a.G1[(int64_t)r * a.h + tid] : 0.f;
  __syncthreads();                                             // (orders the zeros before the ORs: one workgroup)
  const int64_t u = a.rows[r];
  const int64_t b = a.indptr[u], e = a.indptr[u + 1];
  float total = 0.f;
  for (int64_t t0 = b; t0 < e; t0 += 1024) {                   // four positives per thread and round
    float l = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int64_t t = t0 + k4 * 256 + tid;
      if (t < e) {
        const int it = a.indices[t];
        atomicOr(&bm[it >> 5], 1u << (it & 31));
        const float* w = a.Wp1 + (int64_t)it * a.h;
        const float bi = a.bp1[it];
        float wk[kT];                                          // the row's loads in flight, then the chain in k order
#pragma unroll
        for (int k = 0; k < kT; ++k) wk[k] = k < a.h ? w[k] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kT; ++k) acc = __builtin_fmaf(wk[k], s_g[k], acc);
        l += acc + bi;
      }
    }
    s_l[tid] = l;                                              // fixed tree: the same sum every run
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
      if (tid < st) s_l[tid] += s_l[tid + st];
      __syncthreads();
    }
    total += s_l[0];
    __syncthreads();
  }
  if (tid == 0) a.posll[r] = total;
}

// ---------------------------------------------------------------------------------------------------------------
// Pass 1: per row the running (max, Σ exp) over this workgroup's item range.  8 waves = 8 row tiles, the group's W
// tiles in LDS, no barrier inside a group: the waves drift apart and one wave's exps run under another's MFMAs.
// MODE 0: statistics; 2: also copies the logits out (tests).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGroup1 = 8;
template <int MODE>
__global__ __launch_bounds__(kThreads, 2) void vae_dec_stats_kernel(const FusedArgs a) {
  __shared__ float Wt[kGroup1 * kScratch];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hlf = lane >> 5, j32 = lane & 31;
  const int row = (blockIdx.y * kWaves + wave) * kT + j32;
  const bool active = row - j32 < a.rows_pad;                  // wave-uniform
  const int t_begin = blockIdx.x * a.tiles_per_wg;
  const int t_end = min(a.main_tiles, t_begin + a.tiles_per_wg);
  float gA[17];                                                // logits' B operand: lane = row, k = 2·step + hlf
#pragma unroll
  for (int s = 0; s < 16; ++s) gA[s] = a.G1[(int64_t)min(row, a.batch - 1) * a.h + min(2 * s + hlf, a.h - 1)];   // (unconditional,
#pragma unroll                                                                                                  //  then masked)
  for (int s = 0; s < 16; ++s) gA[s] = (row < a.batch && 2 * s + hlf < a.h) ? gA[s] : 0.f;
  gA[16] = hlf == 0 ? 1.0f : 0.f;                              // step 16: bias · 1
  float mx = -INFINITY, sm = 0.f;
  auto fold_tile = [&](const float* W, int t) {
    float wA[17];                                              // A operand: lane = item, k = 2·step + hlf
#pragma unroll
    for (int s = 0; s < 16; ++s) wA[s] = W[j32 * kLd + 2 * s + hlf];
    wA[16] = hlf == 0 ? W[j32 * kLd + 32] : 0.f;
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 17; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[s], gA[s], c, 0, 0, 0);
    // c[j] = logit(row, item 32·t + c_row(j, hlf)), bias included
    if (MODE == 2 && row < a.batch) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int it = t * kT + c_row(j, hlf);
        if (it < a.cols) a.dbg_logits[(int64_t)row * a.cols + it] = c[j];
      }
    }
    float m16 = c[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) m16 = fmaxf(m16, c[j]);
    const float mnew = fmaxf(mx, m16);
    const float ref = mnew == -INFINITY ? 0.f : mnew;
    const float nref = -ref * kLog2e;
    float s16 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s16 += exp2_scaled(__builtin_fmaf(c[j], kLog2e, nref));
    sm = sm * exp_nonpos(mx - ref) + s16;
    mx = mnew;
  };
  for (int g0 = t_begin; g0 < t_end; g0 += kGroup1) {
    const int n = min(kGroup1, t_end - g0);
    __syncthreads();                                           // the previous group's readers are done
    stage_w_tiles<kGroup1, 16>(a, g0, n, Wt, tid);
    __syncthreads();
    if (!active) continue;
    for (int tt = 0; tt < n; ++tt) fold_tile(Wt + tt * kScratch, g0 + tt);
  }
  // the left-over tiles (see pass 2): the (tile, this workgroup's 8 row tiles) groups dealt round robin over x
  for (int j = 0; j < a.n_rem; ++j) {
    if ((j * (int)gridDim.y + (int)blockIdx.y) % (int)gridDim.x != (int)blockIdx.x) continue;
    __syncthreads();
    stage_w_tiles<kGroup1, 16>(a, a.main_tiles + j, 1, Wt, tid);
    __syncthreads();
    if (active) fold_tile(Wt, a.main_tiles + j);
  }
  if (!active) return;
  const float omx = __shfl_xor(mx, 32, NR_WAVE), osm = __shfl_xor(sm, 32, NR_WAVE);
  const float m = fmaxf(mx, omx), ref = m == -INFINITY ? 0.f : m;
  const float s0 = hlf == 0 ? sm : osm, m0 = hlf == 0 ? mx : omx;      // half 0 first, then half 1
  const float s1 = hlf == 0 ? osm : sm, m1 = hlf == 0 ? omx : mx;
  const float ssum = s0 * exp_nonpos(m0 - ref) + s1 * exp_nonpos(m1 - ref);
  if (hlf == 0) {
    float* o = a.pstat + (int64_t)blockIdx.x * 2 * a.rows_pad + row;
    o[0] = m; o[a.rows_pad] = ssum;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pass 2: the gradients.  8 waves x RT row tiles = a chunk of 256·RT rows; one barrier per item tile (the exchange
// of the waves' partial dW tiles).
//
// Tail.  I = 40,981 is 1,280.7 item tiles = 5.003 per CU: with whole tiles dealt to whole workgroups every one ran
// SIX tile steps (the ceiling), a sixth of the kernel for 0.06 % of the work.  So every workgroup takes
// floor(n_tiles / n_wg) tiles and the R tiles left over are cut by ROWS into groups of 8 row tiles (one per wave) dealt
// round robin: a group is one wave-tile per wave, its dW partial goes to a small buffer (xpart) that
// vae_dec_xreduce (the last blocks of the dg1 reduce launch) adds over the groups of a tile, in row order.  dg1 needs nothing extra: the group's
// contribution lands in the wave's accumulators like any other tile's.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGroup2 = 6;
template <int RT>
__global__ __launch_bounds__(kThreads, 1) void vae_dec_grad_kernel(const FusedArgs a) {
  extern __shared__ float lds[];
  float* Wt = lds;                                   // [kGroup2][kT][kLd]      W tiles of the group (+ bias column)
  float* scratch = Wt + kGroup2 * kScratch;          // [2][kWaves][kScratch]   transposes, then the partial dW tiles
  float* gS = scratch + 2 * kWaves * kScratch;       // [RT·256][kLd]           g1 of the row chunk
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hlf = lane >> 5, j32 = lane & 31;
  const int t_begin = blockIdx.x * a.tiles_per_wg;
  const int t_end = min(a.main_tiles, t_begin + a.tiles_per_wg);
  constexpr int kChunk = RT * kWaves * kT;
  int nbar = 0;                                      // tile steps so far: the parity of the scratch buffers

  for (int chunk0 = 0; chunk0 < a.batch; chunk0 += kChunk) {
    __syncthreads();                                 // the previous chunk's readers of gS are done
    for (int e0 = tid; e0 < kChunk * kT; e0 += 8 * kThreads) {       // 8 loads in flight per thread
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + i * kThreads, r = e >> 5, k = e & 31;
        v[i] = a.G1[(int64_t)min(chunk0 + r, a.batch - 1) * a.h + min(k, a.h - 1)];       // (unconditional, masked below)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + i * kThreads, r = e >> 5, k = e & 31;
        gS[r * kLd + k] = (chunk0 + r < a.batch && k < a.h) ? v[i] : 0.f;
      }
    }
    __syncthreads();

    // per wave and row tile: its rows' statistics and the dg1 accumulators (the logits' B operand — lane = row,
    // k = 2·step + hlf — is read from gS at every tile: 17 LDS reads against 34 registers held for the whole kernel,
    // which spilled; measured 54.8 us against 57.2 at five tiles)
    int row[RT];
    float nlse[RT], nbi[RT], invb[RT];
    f32x16 c1[RT];
    const float one_or_zero = hlf == 0 ? 1.0f : 0.f;             // step 16: bias · 1
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int lr0 = (wave + kWaves * q) * kT;
      row[q] = chunk0 + lr0 + j32;
      const float2 st = a.stat[min(row[q], a.batch - 1)];
      nlse[q] = -st.x * kLog2e;
      invb[q] = row[q] < a.batch ? a.inv_batch : 0.f;           // rows beyond the batch: G = 0
      nbi[q] = st.y * invb[q];
#pragma unroll
      for (int r = 0; r < 16; ++r) c1[q][r] = 0.f;
    }

    // one tile step: logits -> G -> dg1 / dW MFMAs of the row tiles `only` (-1: all RT), the 8 waves' partial dW tiles
    // exchanged behind ONE barrier and added in wave order into dWp1 / dbp1 (xdst == NULL) or into xdst
    auto tile_step = [&](const float* W, int t, int only, const uint32_t (&word)[RT], float* xdst) {
      const int par = nbar & 1;
      ++nbar;
      float* my = scratch + (par * kWaves + wave) * kScratch;
      float wA[17];                                  // logits' A operand: lane = item, k = 2·step + hlf
#pragma unroll
      for (int s = 0; s < 16; ++s) wA[s] = W[j32 * kLd + 2 * s + hlf];
      wA[16] = hlf == 0 ? W[j32 * kLd + 32] : 0.f;
      f32x16 c2;
      float cs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) c2[r] = 0.f;
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        if (only >= 0 && only != q) continue;         // workgroup-uniform
        const int lr0 = (wave + kWaves * q) * kT;
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[s], gS[(lr0 + j32) * kLd + 2 * s + hlf], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[16], one_or_zero, c, 0, 0, 0);
        // c[j] = logit(row = lane & 31, item = 32·t + c_row(j, hlf)), bias included.
        // G = (softmax · n_b − x) / B with softmax = 2^(logit·log2e − lse·log2e): one fma + v_exp_f32 per logit.
        // (MFMA time and VALU time ADD on this part — profiles/r04_exp_mfma_valu_overlap.txt — so the 11-instruction
        // full-precision exp cost a tenth of the kernel; the single rounding of the exponent leaves a relative
        // error of 6e-8·|l|·log2e in a probability e^l: 2e-7 at l = -2, and the elements with large |l| are the
        // ones that carry no gradient.)  g = softmax·(n_b/B) − x/B as ONE fma: the addend is -1/B where the
        // positives' bit is set (the bit field sign-extended to a mask over the bits of -1/B), else 0
        float g[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int m = __builtin_amdgcn_sbfe((int)word[q], c_row(j, hlf), 1);            // 0 or -1
          const float sub = __int_as_float(m & __float_as_int(-invb[q]));
          const float pr = exp2_scaled(__builtin_fmaf(c[j], kLog2e, nlse[q]));
          g[j] = __builtin_fmaf(pr, nbi[q], sub);
        }
        // dg1[row][col] += Σ_item G[row][item] · W[item][col]: register j IS contraction step j
#pragma unroll
        for (int s = 0; s < 16; ++s)
          c1[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s], W[c_row(s, hlf) * kLd + j32], c1[q], 0, 0, 0);
        // transpose through the wave's LDS tile: lanes over items, rows become the contraction index
#pragma unroll
        for (int j = 0; j < 16; ++j) my[c_row(j, hlf) * kLd + j32] = g[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float gt[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) gt[s] = my[j32 * kLd + 2 * s + hlf];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          cs += gt[s];
          c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[s], gS[(lr0 + 2 * s + hlf) * kLd + j32], c2, 0, 0, 0);
        }
      }
      // this wave's partial dW tile + column sums
#pragma unroll
      for (int r = 0; r < 16; ++r) my[r * NR_WAVE + lane] = c2[r];
      cs += __shfl_xor(cs, 32, NR_WAVE);
      if (hlf == 0) my[16 * NR_WAVE + j32] = cs;
      __syncthreads();
      // the 8 partial tiles added in wave order — every wave takes two of the 16 accumulator registers (the scratch
      // of this parity is next written two tile steps on, behind the next barrier)
      const float* base = scratch + par * kWaves * kScratch;
      const int item0 = t * kT;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sum += base[w * kScratch + r * NR_WAVE + lane];
        if (xdst) xdst[r * NR_WAVE + lane] = sum;
        else {
          const int it = item0 + c_row(r, hlf);
          if (it < a.cols && j32 < a.h) {
            float* o = a.dWp1 + (int64_t)it * a.h + j32;
            *o = chunk0 ? *o + sum : sum;            // later chunks add to what is there
          }
        }
      }
      if (wave == 0 && hlf == 0) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sum += base[w * kScratch + 16 * NR_WAVE + j32];
        if (xdst) xdst[16 * NR_WAVE + j32] = sum;
        else if (item0 + j32 < a.cols) {
          float* o = a.dbp1 + item0 + j32;
          *o = chunk0 ? *o + sum : sum;
        }
      }
    };

    // the workgroup's steps of this chunk: its whole tiles (W staged kGroup2 tiles at a time), then its share of the
    // left-over tiles — groups (tile, 8 row tiles) dealt round robin, those whose rows lie in this chunk.  ONE loop
    // (one inlined copy of the tile step: two copies cost 40 spilled registers).
    const int n_main = max(t_end - t_begin, 0);
    int n_extra = 0;
    for (int gi = blockIdx.x; gi < a.n_rem * a.groups_per_tile; gi += gridDim.x)
      n_extra += ((gi % a.groups_per_tile) * kWaves * kT / kChunk == chunk0 / kChunk) ? 1 : 0;
    int gi_next = blockIdx.x;                        // cursor over this workgroup's groups
    auto next_group = [&]() {                        // the next group of this chunk (workgroup-uniform)
      while ((gi_next % a.groups_per_tile) * kWaves * kT / kChunk != chunk0 / kChunk) gi_next += gridDim.x;
      const int gi = gi_next;
      gi_next += gridDim.x;
      return gi;
    };
    int gi_cur = n_main == 0 && n_extra > 0 ? next_group() : -1;
    auto tile_of = [&](int it, int gi) { return it < n_main ? t_begin + it : a.main_tiles + gi / a.groups_per_tile; };
    uint32_t word[RT];
    if (n_main + n_extra > 0) {
      const int t0 = tile_of(0, gi_cur);
#pragma unroll
      for (int q = 0; q < RT; ++q) word[q] = a.bitmap[(int64_t)min(row[q], a.batch - 1) * a.words + t0];
    }
    for (int it = 0; it < n_main + n_extra; ++it) {
      const bool main_step = it < n_main;
      const int tt = main_step ? it % kGroup2 : 0;
      if (!main_step || tt == 0) {
        __syncthreads();                             // the previous group's W readers are done
        if (main_step) stage_w_tiles<kGroup2, 4>(a, t_begin + it, min(kGroup2, n_main - it), Wt, tid);
        else stage_w_tiles<kGroup2, 4>(a, tile_of(it, gi_cur), 1, Wt, tid);
        __syncthreads();
      }
      const int t = tile_of(it, gi_cur);
      const int only = main_step ? -1 : ((gi_cur % a.groups_per_tile)) % RT;
      float* xdst = main_step ? nullptr : a.xpart + (int64_t)gi_cur * kScratch;
      // the next step's positives: a tile ahead of their use
      int gi_after = gi_cur;
      if (it + 1 >= n_main && it + 1 < n_main + n_extra) gi_after = next_group();
      uint32_t wnext[RT];
      const int t_after = it + 1 < n_main + n_extra ? tile_of(it + 1, gi_after) : t;
#pragma unroll
      for (int q = 0; q < RT; ++q) wnext[q] = a.bitmap[(int64_t)min(row[q], a.batch - 1) * a.words + t_after];
      tile_step(Wt + tt * kScratch, t, only, word, xdst);
#pragma unroll
      for (int q = 0; q < RT; ++q) word[q] = wnext[q];
      gi_cur = gi_after;
    }

    // the chunk's rows leave: dg1 partials of this workgroup
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int lr0 = chunk0 + (wave + kWaves * q) * kT;
      if (lr0 < a.rows_pad) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.part[((int64_t)blockIdx.x * a.rows_pad + lr0 + c_row(r, hlf)) * kT + j32] = c1[q][r];
      }
    }
  }
}

// the left-over tiles: dW_p1 / db_p1 = the groups' partial tiles added in row order (group 0 = row tiles 0-7, ...);
// runs as the last blocks of vae_dg1_reduce_wg_kernel's launch
__device__ __forceinline__ void vae_dec_xreduce(const FusedArgs& a, int j) {
  const int t = a.main_tiles + j;
  for (int e = threadIdx.x; e < kScratch; e += 256) {
    float sum = 0.f;
    for (int g = 0; g < a.groups_per_tile; ++g) sum += a.xpart[((int64_t)j * a.groups_per_tile + g) * kScratch + e];
    if (e < 16 * NR_WAVE) {
      const int r = e >> 6, ln = e & 63, it = t * kT + c_row(r, ln >> 5), col = ln & 31;
      if (it < a.cols && col < a.h) a.dWp1[(int64_t)it * a.h + col] = sum;
    } else {
      const int it = t * kT + (e - 16 * NR_WAVE);
      if (it < a.cols) a.dbp1[it] = sum;
    }
  }
}

// per row: the workgroups' (max, Σexp) combined — a wave per row; lane w takes workgroups w, w+64, .. in order, the
// lanes' sums meet in a fixed butterfly
__global__ __launch_bounds__(256) void vae_dec_stat_kernel(const float* __restrict__ pstat, int n_wg, int rows_pad,
                                                           int batch, const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ rows,
                                                           const float* __restrict__ posll,
                                                           float2* __restrict__ stat, float* __restrict__ nll) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= batch) return;
  float m = -INFINITY;
  for (int w = lane; w < n_wg; w += NR_WAVE) m = fmaxf(m, pstat[(int64_t)w * 2 * rows_pad + r]);
#pragma unroll
  for (int x = 32; x >= 1; x >>= 1) m = fmaxf(m, __shfl_xor(m, x, NR_WAVE));
  float s = 0.f;
  for (int w = lane; w < n_wg; w += NR_WAVE) {
    const float* p = pstat + (int64_t)w * 2 * rows_pad + r;
    const float pm = p[0];
    if (pm != -INFINITY) s += p[rows_pad] * exp_nonpos(pm - m);
  }
#pragma unroll
  for (int x = 32; x >= 1; x >>= 1) s += __shfl_xor(s, x, NR_WAVE);
  if (lane == 0) {
    const float lse = m + logf(s);
    const int64_t u = rows[r];
    const float n = (float)(indptr[u + 1] - indptr[u]);
    stat[r] = make_float2(lse, n);
    nll[r] = -(posll[r] - n * lse);
  }
}

// dg1[r][j] = Σ_wg part[wg][r][j], workgroups in order: 4 threads per output take a quarter each (in order),
// the quarters are added in order
__global__ __launch_bounds__(256) void vae_dg1_reduce_wg_kernel(const float* __restrict__ part, int n_wg,
                                                                int rows_pad, int batch, int h,
                                                                float* __restrict__ dG1, const FusedArgs a,
                                                                int n_dg1_blocks) {
  __shared__ float s_q[4][64];
  if ((int)blockIdx.x >= n_dg1_blocks) {             // workgroup-uniform: the left-over item tiles' dW / db
    vae_dec_xreduce(a, (int)blockIdx.x - n_dg1_blocks);
    return;
  }
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;                     // over batch x 32 (padded width)
  const int r = idx >> 5, j = idx & 31;
  const int per = (n_wg + 3) / 4, w0 = g * per, w1 = min(n_wg, w0 + per);
  float sum = 0.f;
  if (r < batch) {
    const float* p = part + (int64_t)r * kT + j;
    const int64_t stride = (int64_t)rows_pad * kT;
    for (int k0 = w0; k0 < w1; k0 += 8) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = p[(int64_t)min(k0 + k, w1 - 1) * stride];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k0 + k < w1) sum += t[k];
    }
  }
  s_q[g][o] = sum;
  __syncthreads();
  if (g == 0 && r < batch && j < h) dG1[(int64_t)r * h + j] = ((s_q[0][o] + s_q[1][o]) + s_q[2][o]) + s_q[3][o];
}

int fused_workgroups() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  return cus > 0 ? cus : 256;
}

struct FusedLayout {
  size_t bitmap, pstat, stat, posll, part, xpart, total;
  int words, n_tiles, n_wg, tiles_per_wg, rows_pad, main_tiles, n_rem, groups_per_tile;
};

FusedLayout fused_layout(int batch, int cols) {
  FusedLayout L;
  L.words = (cols + 31) / 32;
  L.n_tiles = L.words;
  const int want = fused_workgroups();                       // one workgroup per CU, contiguous item ranges
  L.n_wg = L.n_tiles < want ? L.n_tiles : want;
  L.tiles_per_wg = L.n_tiles / L.n_wg;                       // whole tiles per workgroup (floor) ...
  L.main_tiles = L.n_wg * L.tiles_per_wg;
  L.n_rem = L.n_tiles - L.main_tiles;                        // ... and the left-over tiles, cut by rows
  L.rows_pad = (batch + kT - 1) / kT * kT;
  L.groups_per_tile = (L.rows_pad / kT + kWaves - 1) / kWaves;
  size_t off = 0;
  L.bitmap = off; off += nr_align_up((size_t)batch * L.words * sizeof(uint32_t), 256);
  L.pstat = off; off += nr_align_up((size_t)L.n_wg * 2 * L.rows_pad * sizeof(float), 256);
  L.stat = off; off += nr_align_up((size_t)batch * sizeof(float2), 256);
  L.posll = off; off += nr_align_up((size_t)batch * sizeof(float), 256);
  L.part = off; off += nr_align_up((size_t)L.n_wg * L.rows_pad * kT * sizeof(float), 256);
  L.xpart = off; off += nr_align_up((size_t)(L.n_rem * L.groups_per_tile + 1) * kScratch * sizeof(float), 256);
  L.total = off;
  return L;
}

template <int RT>
int launch_grad(const FusedArgs& a, int n_wg, hipStream_t st) {
  const size_t lds = (size_t)(kGroup2 * kScratch + 2 * kWaves * kScratch + RT * kWaves * kT * kLd) * sizeof(float);
  static std::atomic<bool> attr[64];                 // the attribute is per device
  int dev = 0;
  NR_CHECK_HIP(hipGetDevice(&dev));
  dev &= 63;
  if (!attr[dev].load()) {
    NR_CHECK_HIP(hipFuncSetAttribute((const void*)vae_dec_grad_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    attr[dev].store(true);
  }
  hipLaunchKernelGGL((vae_dec_grad_kernel<RT>), dim3(n_wg), dim3(kThreads), lds, st, a);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // namespace

extern "C" {

int nrhip_vae_decoder_fused_workspace_bytes(int batch, int cols, size_t* bytes) {
  NR_REQUIRE(bytes && batch >= 1 && cols >= 1, NR_ERR_ARG, "vae_decoder_fused_workspace_bytes: bad arguments");
  *bytes = fused_layout(batch, cols).total;
  return NR_OK;
}

int nrhip_vae_decoder_fused(int batch, int cols, int h, const float* d_G1, const float* d_Wp1,
                            const float* d_bp1, const int64_t* d_indptr, const int32_t* d_indices,
                            const int32_t* d_rows, float* d_nll, float* d_dWp1, float* d_dbp1,
                            float* d_dG1, void* d_ws, size_t ws_bytes, float* d_dbg_logits, void* stream) {
  NR_REQUIRE(d_G1 && d_Wp1 && d_bp1 && d_indptr && d_indices && d_rows && d_nll && d_dWp1 && d_dbp1 && d_dG1 &&
                 d_ws && batch >= 1 && cols >= 1,
             NR_ERR_ARG, "vae_decoder_fused: bad arguments");
  NR_REQUIRE(h >= 1 && h <= kT, NR_ERR_UNSUPPORTED, "vae_decoder_fused: hidden %d > 32", h);
  const FusedLayout L = fused_layout(batch, cols);
  NR_REQUIRE(ws_bytes >= L.total, NR_ERR_WORKSPACE,
             "vae_decoder_fused: workspace too small (nrhip_vae_decoder_fused_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)d_ws;
  FusedArgs a;
  a.G1 = d_G1; a.Wp1 = d_Wp1; a.bp1 = d_bp1; a.bitmap = (const uint32_t*)(ws + L.bitmap);
  a.batch = batch; a.cols = cols; a.h = h; a.words = L.words; a.n_tiles = L.n_tiles;
  a.tiles_per_wg = L.tiles_per_wg; a.rows_pad = L.rows_pad;
  a.main_tiles = L.main_tiles; a.n_rem = L.n_rem; a.groups_per_tile = L.groups_per_tile;
  a.xpart = (float*)(ws + L.xpart);
  a.pstat = (float*)(ws + L.pstat); a.stat = (const float2*)(ws + L.stat);
  a.inv_batch = 1.0f / (float)batch;
  a.dWp1 = d_dWp1; a.dbp1 = d_dbp1; a.part = (float*)(ws + L.part);
  a.dbg_logits = d_dbg_logits;
  a.indptr = d_indptr; a.indices = d_indices; a.rows = d_rows;
  a.bitmap_out = (uint32_t*)(ws + L.bitmap); a.posll = (float*)(ws + L.posll); a.n_wg = L.n_wg;
  // a workgroup's 8 waves take RT row tiles each: 256 rows per RT; larger batches run in chunks inside
  const bool one = batch <= kWaves * kT;
  // (the per-row side kernel under pass 1 on a second stream, forked and joined by events, was measured: the step
  // went from 0.148 to 0.159 ms — the event packets cost more than the 11 us they hide)
  hipLaunchKernelGGL(vae_dec_rows_kernel, dim3(batch), dim3(256), 0, st, a);
  NR_LAUNCH_CHECK();
  const dim3 grid1(L.n_wg, (L.rows_pad / kT + kWaves - 1) / kWaves);
  if (d_dbg_logits) hipLaunchKernelGGL(vae_dec_stats_kernel<2>, grid1, dim3(kThreads), 0, st, a);
  else hipLaunchKernelGGL(vae_dec_stats_kernel<0>, grid1, dim3(kThreads), 0, st, a);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(vae_dec_stat_kernel, dim3((batch + 3) / 4), dim3(256), 0, st, a.pstat, L.n_wg, L.rows_pad,
                     batch, d_indptr, d_rows, (const float*)(ws + L.posll), (float2*)(ws + L.stat), d_nll);
  NR_LAUNCH_CHECK();
  NR_TRY(one ? launch_grad<1>(a, L.n_wg, st) : launch_grad<2>(a, L.n_wg, st));
  const int n_dg1_blocks = (batch * kT + 63) / 64;
  hipLaunchKernelGGL(vae_dg1_reduce_wg_kernel, dim3(n_dg1_blocks + L.n_rem), dim3(256), 0, st, a.part, L.n_wg,
                     L.rows_pad, batch, h, d_dG1, a, n_dg1_blocks);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
